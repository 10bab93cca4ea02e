"""Adam for the executor's parameters on the HIP path (libvqa_amd.so: vqa_adam_tensors, vqa_adam_fc1).

Same update rule, hyper-parameters and skip-if-no-gradient behaviour as the reference's optimizer,
`torch.optim.Adam(ee.parameters(), lr=..., weight_decay=...)` (`Train_model.py:165-167,315`); state
entries carry torch's names (`step`, `exp_avg`, `exp_avg_sq`).  All tensors of a step are updated by
one launch, and when the executor is passed in, the first linear layer's weight (88 % of the
parameters) is updated by a kernel that also writes the bf16 channels-last copy the executor's GEMM
reads, so the executor does not re-pack that tensor on its next forward.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib

_TENSOR = np.dtype([('p', 'u8'), ('g', 'u8'), ('m', 'u8'), ('v', 'u8'), ('numel', 'i8'),
                    ('step_size', 'f4'), ('inv_bc2_sqrt', 'f4')])
_CHUNK = np.dtype([('tensor', 'i4'), ('pad', 'i4'), ('offset', 'i8')])
_CHUNK_ELEMS = 4096


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, executor=None):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError('invalid Adam hyper-parameter')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._executor = executor
        self._host = {}          # device -> (pinned byte buffer, device byte buffer, event)

    def _tables(self, dev, nbytes):
        ent = self._host.get(dev)
        if ent is None or ent[0].numel() < nbytes:
            cap = max(2 * nbytes, 1 << 16)
            ent = [torch.empty(cap, dtype=torch.uint8, pin_memory=True),
                   torch.empty(cap, dtype=torch.uint8, device=dev), None]
            self._host[dev] = ent
        if ent[2] is not None:
            ent[2].synchronize()          # the previous step's upload has left the pinned buffer
        return ent

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        ex = self._executor
        fc1 = None
        if ex is not None and getattr(ex, 'precision', None) == 'bf16':
            fc1 = ex.classifier[ex._fc_idx[0]].weight
        for group in self.param_groups:
            lr, (b1, b2), eps, wd = group['lr'], group['betas'], group['eps'], group['weight_decay']
            rows = []
            for p in group['params']:
                g = p.grad
                if g is None:
                    continue          # unused token this step: no update, no step count (as torch)
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError('vqa_amd.optim.FusedAdam needs contiguous fp32 parameters on the GPU')
                if g.is_sparse or g.dtype != torch.float32:
                    raise RuntimeError('vqa_amd.optim.FusedAdam needs dense fp32 gradients')
                if not g.is_contiguous():
                    g = g.contiguous()
                st = self.state[p]
                if not st:
                    st['step'] = torch.zeros((), dtype=torch.float32)
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['step'] += 1
                t = float(st['step'])
                step_size = lr / (1.0 - b1 ** t)
                inv_bc2_sqrt = 1.0 / math.sqrt(1.0 - b2 ** t)
                if p is fc1:
                    packed = ex._fc1_packed_buffer(p.device)
                    F, P = p.shape[0], ex._dims['p']
                    _lib.check(lib.vqa_adam_fc1(p.data_ptr(), g.data_ptr(), st['exp_avg'].data_ptr(),
                                                st['exp_avg_sq'].data_ptr(), packed.data_ptr(), F, P, wd, b1, b2, eps,
                                                step_size, inv_bc2_sqrt, torch.cuda.current_stream(p.device).cuda_stream))
                    ex._fc1_fresh = True
                    continue
                rows.append((p, g, st['exp_avg'], st['exp_avg_sq'], step_size, inv_bc2_sqrt))
            by_dev = {}
            for r in rows:
                by_dev.setdefault(r[0].device, []).append(r)
            for dev, rs in by_dev.items():
                tt = np.zeros(len(rs), dtype=_TENSOR)
                tt['p'] = [r[0].data_ptr() for r in rs]
                tt['g'] = [r[1].data_ptr() for r in rs]
                tt['m'] = [r[2].data_ptr() for r in rs]
                tt['v'] = [r[3].data_ptr() for r in rs]
                tt['numel'] = [r[0].numel() for r in rs]
                tt['step_size'] = [r[4] for r in rs]
                tt['inv_bc2_sqrt'] = [r[5] for r in rs]
                nch = (tt['numel'] + _CHUNK_ELEMS - 1) // _CHUNK_ELEMS
                ck = np.zeros(int(nch.sum()), dtype=_CHUNK)
                ck['tensor'] = np.repeat(np.arange(len(rs), dtype=np.int32), nch)
                first = np.concatenate([[0], np.cumsum(nch)[:-1]])
                ck['offset'] = (np.arange(len(ck)) - np.repeat(first, nch)) * _CHUNK_ELEMS
                tb, cb = tt.tobytes(), ck.tobytes()
                off = (len(tb) + 255) // 256 * 256
                ent = self._tables(dev, off + len(cb))
                host, devbuf = ent[0], ent[1]
                host[:len(tb)] = torch.frombuffer(bytearray(tb), dtype=torch.uint8)
                host[off:off + len(cb)] = torch.frombuffer(bytearray(cb), dtype=torch.uint8)
                devbuf[:off + len(cb)].copy_(host[:off + len(cb)], non_blocking=True)
                ent[2] = torch.cuda.Event()
                ent[2].record(torch.cuda.current_stream(dev))
                _lib.check(lib.vqa_adam_tensors(devbuf.data_ptr(), devbuf.data_ptr() + off, len(ck), wd, b1, b2, eps,
                                                torch.cuda.current_stream(dev).cuda_stream))
                self._keep = rs          # gradients stay referenced until the next step (kernel still queued)
        return loss
