"""Adam for the executor's parameters on the HIP path (libvqa_amd.so: vqa_adam_tensors, vqa_adam_fc1).

Same update rule, hyper-parameters and skip-if-no-gradient behaviour as the reference's optimizer,
`torch.optim.Adam(ee.parameters(), lr=..., weight_decay=...)` (`Train_model.py:165-167,315`); state
entries carry torch's names (`step`, `exp_avg`, `exp_avg_sq`; the `step` tensors are brought up to date
when the state is read with state_dict(), between steps the counts are kept as host integers).  All tensors of a step are updated by
one launch, and when the executor is passed in, the first linear layer's weight (88 % of the
parameters) is updated by a kernel that also writes the bf16 channels-last copy the executor's GEMM
reads, so the executor does not re-pack that tensor on its next forward.
"""
import ctypes as C
import math
import weakref

import numpy as np
import torch

from . import _lib

_TENSOR = np.dtype([('p', 'u8'), ('g', 'u8'), ('m', 'u8'), ('v', 'u8'), ('numel', 'i8'),
                    ('step_size', 'f4'), ('inv_bc2_sqrt', 'f4')])
_CHUNK = np.dtype([('tensor', 'i4'), ('pad', 'i4'), ('offset', 'i8')])
_CONV = np.dtype([('p', 'u8'), ('g', 'u8'), ('m', 'u8'), ('v', 'u8'), ('pk_fwd', 'u8'), ('pk_bwd', 'u8'),
                  ('step_size', 'f4'), ('inv_bc2_sqrt', 'f4')])
_CHUNK_ELEMS = 4096


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, executor=None):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError('invalid Adam hyper-parameter')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._executor = executor
        if executor is not None:
            # what a data-parallel wrapper asks before it hands this optimizer a slice of the first
            # linear layer's gradient instead of the whole of it (step() applies such a slice)
            executor._sliced_update_optimizer = weakref.ref(self)
        self._host = {}          # device -> two [pinned byte buffer, device byte buffer, event, numpy view] + toggle
        self._recs = {}          # group index -> per-parameter records (see _records)
        self._chunks_key, self._chunks = None, None

    def _tables(self, key, nbytes):
        """Two pinned/device buffer pairs per device (and table kind), used alternately: the host may
        write the next step's tables while the previous upload is still queued."""
        dev = key[0] if isinstance(key, tuple) else key
        ents = self._host.setdefault(key, [None, None, 0])
        k = ents[2] = ents[2] ^ 1
        ent = ents[k]
        if ent is None or ent[0].numel() < nbytes:
            cap = max(2 * nbytes, 1 << 16)
            ent = ents[k] = [torch.empty(cap, dtype=torch.uint8, pin_memory=True),
                             torch.empty(cap, dtype=torch.uint8, device=dev), None, None]
            ent[3] = ent[0].numpy()
        if ent[2] is not None:
            ent[2].synchronize()          # the upload before last has left this pinned buffer
        return ent

    def _records(self, gi, group):
        """Per parameter of a group: [parameter, state dict or None, numel, step count, m ptr, v ptr].
        Step counts are kept as Python ints here and written back to the `step` tensors that
        torch's Adam state carries when the state is read (state_dict)."""
        recs = self._recs.get(gi)
        if recs is None or len(recs) != len(group['params']) or any(r[0] is not p for r, p in zip(recs, group['params'])):
            recs = self._recs[gi] = [[p, None, p.numel(), 0, 0, 0] for p in group['params']]
        return recs

    def _sync_steps(self):
        for recs in self._recs.values():
            for r in recs:
                if r[1] is not None:
                    r[1]['step'].fill_(r[3])

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._recs = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        ex = self._executor
        fc1, targets, shapes = None, None, None
        if ex is not None and getattr(ex, 'precision', None) == 'bf16':
            fc1 = ex.classifier[ex._fc_idx[0]].weight
            tg = ex._conv_pack_targets()
            if tg is not None:
                targets, shapes = tg
        managed = 0              # convolution weights with packed copies that this optimizer owns
        f32 = torch.float32
        for gi, group in enumerate(self.param_groups):
            lr, (b1, b2), eps, wd = group['lr'], group['betas'], group['eps'], group['weight_decay']
            scale = {}           # step count -> (step_size, 1 / sqrt(bias correction 2))
            by_dev = {}          # device -> column lists of the tensor table
            conv = {}            # (device, shape class) -> rows of the convolution-weight table
            keep = []
            for idx, r in enumerate(self._records(gi, group)):
                p = r[0]
                g = p.grad
                tgt = targets.get(p.data_ptr()) if targets is not None else None
                managed += tgt is not None
                shard = None
                if g is None and p is fc1 and getattr(ex, '_fc1_grad_shard', None) is not None:
                    # data parallel with shard_big_update: this rank holds, and updates, its rows only
                    shard, ex._fc1_grad_shard = ex._fc1_grad_shard, None
                    g = shard[0]
                if g is None:
                    continue          # unused token this step: no update, no step count (as torch)
                if g.dtype != f32 or g.is_sparse:
                    raise RuntimeError('vqa_amd.optim.FusedAdam needs dense fp32 gradients')
                if not g.is_contiguous():
                    g = g.contiguous()
                st = r[1]
                if st is None:
                    if not p.is_cuda or p.dtype != f32 or not p.is_contiguous():
                        raise RuntimeError('vqa_amd.optim.FusedAdam needs contiguous fp32 parameters on the GPU')
                    st = self.state[p]
                    if not st:
                        st['step'] = torch.zeros((), dtype=f32)
                        st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                        st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    r[1], r[3] = st, int(st['step'])
                    r[4], r[5] = st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr()
                t = r[3] = r[3] + 1
                sc = scale.get(t)
                if sc is None:
                    sc = scale[t] = (lr / (1.0 - b1 ** t), 1.0 / math.sqrt(1.0 - b2 ** t))
                if p is fc1:
                    packed = ex._fc1_packed_buffer(p.device)
                    F, P = p.shape[0], ex._dims['p']
                    stream = torch.cuda.current_stream(p.device).cuda_stream
                    if shard is None:
                        _lib.check(lib.vqa_adam_fc1(p.data_ptr(), g.data_ptr(), r[4], r[5], packed.data_ptr(), F, P,
                                                    wd, b1, b2, eps, sc[0], sc[1], stream))
                    else:
                        _, rank, world, dp = shard
                        if packed.shape[0] != F:
                            raise RuntimeError('a sharded update needs the layer\'s output width to be a multiple of 128')
                        ex._wait_fc1_gather()      # the previous step's exchange writes the same buffer
                        rows = F // world
                        off = rank * rows * p.shape[1]          # elements in front of this rank's rows
                        _lib.check(lib.vqa_adam_fc1(p.data_ptr() + 4 * off, g.data_ptr(), r[4] + 4 * off, r[5] + 4 * off,
                                                    packed.data_ptr() + 2 * off, rows, P,
                                                    wd, b1, b2, eps, sc[0], sc[1], stream))
                        dp.gather_big_packed(packed)
                    ex._fc1_fresh = True
                    keep.append(g)
                    continue
                if tgt is not None:
                    # a convolution weight: its update also rewrites the packed copies the kernels read
                    conv.setdefault((p.device, tgt[0]), []).append(
                        (p.data_ptr(), g.data_ptr(), r[4], r[5], tgt[1], tgt[2], sc[0], sc[1]))
                    keep.append(g)
                    continue
                cols = by_dev.get(p.device)
                if cols is None:
                    cols = by_dev[p.device] = ([], [], [], [], [], [], [], [])
                cols[0].append(p.data_ptr()); cols[1].append(g.data_ptr()); cols[2].append(r[4]); cols[3].append(r[5])
                cols[4].append(r[2]); cols[5].append(sc[0]); cols[6].append(sc[1]); cols[7].append(idx)
                keep.append(g)
            # one table upload per device: the tensor table and chunk list of vqa_adam_tensors, then the
            # rows of every vqa_adam_conv_pack launch, each part 256-byte aligned
            for dev in set(by_dev) | {d for d, _ in conv}:
                parts = []               # (offset, byte count, fill function, launch function)
                total = 0

                def add(nbytes, fill, launch):
                    nonlocal total
                    parts.append((total, nbytes, fill, launch))
                    total += (nbytes + 255) // 256 * 256
                cols = by_dev.get(dev)
                if cols is not None:
                    n = len(cols[0])
                    ckey = (gi, dev, tuple(cols[7]))
                    if self._chunks_key != ckey:      # the chunk list depends on which tensors take part only
                        numel = np.asarray(cols[4], dtype=np.int64)
                        nch = (numel + _CHUNK_ELEMS - 1) // _CHUNK_ELEMS
                        ck = np.zeros(int(nch.sum()), dtype=_CHUNK)
                        ck['tensor'] = np.repeat(np.arange(n, dtype=np.int32), nch)
                        first = np.concatenate([[0], np.cumsum(nch)[:-1]])
                        ck['offset'] = (np.arange(len(ck)) - np.repeat(first, nch)) * _CHUNK_ELEMS
                        self._chunks_key, self._chunks = ckey, ck.view(np.uint8)
                    cb = self._chunks
                    off = (n * _TENSOR.itemsize + 255) // 256 * 256

                    def fill_tensors(view, cols=cols, n=n, cb=cb, off=off):
                        tt = view[:n * _TENSOR.itemsize].view(_TENSOR)
                        tt['p'], tt['g'], tt['m'], tt['v'] = cols[0], cols[1], cols[2], cols[3]
                        tt['numel'], tt['step_size'], tt['inv_bc2_sqrt'] = cols[4], cols[5], cols[6]
                        view[off:off + len(cb)] = cb

                    def launch_tensors(ptr, stream, cb=cb, off=off):
                        _lib.check(lib.vqa_adam_tensors(ptr, ptr + off, len(cb) // _CHUNK.itemsize,
                                                        wd, b1, b2, eps, stream.cuda_stream))
                    add(off + len(cb), fill_tensors, launch_tensors)
                for (d, kind), rows in conv.items():
                    if d != dev:
                        continue

                    def fill_conv(view, rows=rows):
                        tt = view[:len(rows) * _CONV.itemsize].view(_CONV)
                        for j, name in enumerate(_CONV.names):
                            tt[name] = [row[j] for row in rows]

                    def launch_conv(ptr, stream, rows=rows, kind=kind):
                        cout, cin, taps = shapes[kind]
                        _lib.check(lib.vqa_adam_conv_pack(ptr, len(rows), cout, cin, taps, wd, b1, b2, eps, stream.cuda_stream))
                    add(len(rows) * _CONV.itemsize, fill_conv, launch_conv)
                ent = self._tables(dev, total)
                devbuf, hostnp = ent[1], ent[3]
                for o, nb, fill, _ in parts:
                    fill(hostnp[o:o + nb])
                stream = torch.cuda.current_stream(dev)
                devbuf[:total].copy_(ent[0][:total], non_blocking=True)
                ent[2] = torch.cuda.Event()
                ent[2].record(stream)
                for o, _, _, launch in parts:
                    launch(devbuf.data_ptr() + o, stream)
            self._keep = keep            # gradients stay referenced until the next step (kernels still queued)
        if ex is not None:
            ex._packed_stale = True      # parameters changed without their version counters moving
        if targets is not None and managed == len(targets):
            # every packed convolution weight belongs to this optimizer: the copies now hold this
            # step's values (tensors without a gradient did not change)
            ex._conv_packs_fresh = True
        return loss
