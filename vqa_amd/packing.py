"""Weights in the layouts the kernels read, kept current between optimizer steps: the per-token
parameters stacked into shared [V, ...] arrays, the bf16 MFMA-fragment packs of every convolution
(forward and input-gradient form) and the bf16 channels-last copy of the first linear layer.
A mixin of ModuleNet (module_net.py); the parameters keep the reference's names and shapes
(`Module_Model.py:96-150`), these are derived copies."""
import torch

from . import ops
from ._state import _rup


class _WeightPacking:
    def _stack_module_params(self, device):
        """The per-token modules' parameters as rows of six [V, ...] device arrays (one weight group
        per token, zeros where a token has no such parameter): each nn.Parameter's storage is moved
        into its row, so the arrays ARE the parameters - an optimizer's in-place update, copy_() and
        load_state_dict() write through - and the packing kernels read all tokens of a kind in one
        launch without gathering them first.  Redone when the parameters were given new storage."""
        d = self._dims
        c, nv = d['c'], len(self._tokens)
        f32 = dict(dtype=torch.float32, device=device)
        st = {'device': device,
              'w1': torch.zeros(nv, c, c, 3, 3, **f32), 'w2': torch.zeros(nv, c, c, 3, 3, **f32),
              'wp': torch.zeros(nv, c, 2 * c, 1, 1, **f32),
              'b1': torch.zeros(nv, c, **f32), 'b2': torch.zeros(nv, c, **f32), 'bp': torch.zeros(nv, c, **f32)}
        with torch.no_grad():
            for i, t in enumerate(self._tokens):
                m = self._modules[t]
                pairs = [(m.conv1.weight, st['w1']), (m.conv1.bias, st['b1']),
                         (m.conv2.weight, st['w2']), (m.conv2.bias, st['b2'])]
                if self._binary[i]:
                    pairs += [(m.proj.weight, st['wp']), (m.proj.bias, st['bp'])]
                for prm, arr in pairs:
                    if prm.device != device or prm.dtype != torch.float32:
                        raise RuntimeError('vqa_amd.ModuleNet: parameters must be fp32 tensors on the features\' device')
                    arr[i].copy_(prm.detach())
                    prm.data = arr[i]
        nfixed = 2 * (3 + len(self._fc_idx))        # stem, classifier conv and linear layers come first
        st['sig'] = tuple(p.data_ptr() for _, p in self._param_groups()[nfixed:])
        self._stacks = st
        return st

    def _pack_weights(self, device):
        """bf16 LDS-image copies of all weights; redone only when a parameter changed."""
        plist = self._param_groups()
        ptrs = tuple(p.data_ptr() for _, p in plist)
        stk = self._stacks
        if stk is None or stk['device'] != device or ptrs[len(ptrs) - len(stk['sig']):] != stk['sig']:
            stk = self._stack_module_params(device)          # moves the per-token parameters
            ptrs = tuple(p.data_ptr() for _, p in plist)
        # Version counters catch in-place edits and load_state_dict, but torch's fused optimizers
        # update parameters without bumping them, so a backward pass (after which an optimizer step
        # is to be expected) also marks the packed copies stale.
        key = (ptrs, tuple(p._version for _, p in plist), str(device))
        if key == self._packed_key and not self._packed_stale:
            return self._packed
        same = key == self._packed_key
        fc1_ready = self._fc1_fresh and same and self._fc1_packed is not None
        # vqa_amd.optim.FusedAdam rewrites the packed copies of the convolution weights it updates (all
        # but the stem's first): nothing to redo for those when nothing else touched the parameters
        conv_ready = self._conv_packs_fresh and same
        self._fc1_fresh = self._conv_packs_fresh = False
        self._packed_stale = False
        d = self._dims
        c, p_, cin, cpad = d['c'], d['p'], d['cin'], d['cin_pad']
        nv = len(self._tokens)
        pk = self._packed if same else {}
        bufs = self._pack_bufs

        def pack(name, *args, **kw):
            # persistent destinations: the optimizer kernel writes into them between forwards
            out = bufs.get(name)
            if out is not None and out.device != device:
                out = None
            bufs[name] = pk[name] = ops.pack_conv_weight(*args, out=out, **kw)
        with torch.no_grad():
            f32 = dict(dtype=torch.float32, device=device)
            w1, w2, wp = stk['w1'], stk['w2'], stk['wp']
            pk['b_conv1'], pk['b_conv2'], pk['b_proj'] = stk['b1'], stk['b2'], stk['bp']
            g33 = c * c * 9
            s0, s1, cl = self.stem[0], self.stem[2], self.classifier[0]
            if not conv_ready:
                pack('w_conv1', w1, nv, c, c, 9, g33, c * 9, 9)
                pack('w_conv2', w2, nv, c, c, 9, g33, c * 9, 9)
                pack('wd_conv1', w1, nv, c, c, 9, g33, 9, c * 9, flip=True)
                pack('wd_conv2', w2, nv, c, c, 9, g33, 9, c * 9, flip=True)
                pack('w_proj', wp, nv, c, 2 * c, 1, 2 * c * c, 2 * c, 1)
                pack('wd_proj', wp, nv, 2 * c, c, 1, 2 * c * c, 1, 2 * c)
                pack('w_stem1', s1.weight.contiguous(), 1, c, c, 9, 0, c * 9, 9)
                pack('wd_stem1', s1.weight.contiguous(), 1, c, c, 9, 0, 9, c * 9, flip=True)
                pack('w_cls', cl.weight.contiguous(), 1, p_, c, 1, 0, c, 1)
                pack('wd_cls', cl.weight.contiguous(), 1, c, p_, 1, 0, 1, c)
            pack('w_stem0', s0.weight.contiguous(), 1, c, cin, 9, 0, cin * 9, 9)
            pk['b_stem0'], pk['b_stem1'], pk['b_cls'] = s0.bias, s1.bias, cl.bias
            # linear layers: bf16 [out padded to 128][in]; the first one over the flattened map is
            # re-ordered from c*196+hw (Layers.py:42-44) to the channels-last hw*P+c
            fcw = []
            for li, idx in enumerate(self._fc_idx):
                lin = self.classifier[idx]
                fo, fi = lin.weight.shape
                if li == 0:
                    buf = self._fc1_packed_buffer(device)
                    if not fc1_ready:
                        if self._fc1_master_stale:
                            raise RuntimeError('the first linear layer is updated in slices (shard_big_update): call '
                                               'DataParallelExecutor.gather_sharded_state() on every rank before '
                                               'changing parameters in any other way')
                        ops.fc_weight_to_nhwc(lin.weight.contiguous(), buf, fo, _rup(fo, 128), p_)
                else:
                    buf = self._packed.get('fc_w', [None] * (li + 1))[li] if self._packed_key is not None else None
                    if buf is None or buf.device != device or buf.shape != (_rup(fo, 128), fi):
                        buf = torch.zeros(_rup(fo, 128), fi, dtype=torch.bfloat16, device=device)
                    ops.cast_pad(lin.weight.contiguous(), buf, fo, fi, _rup(fo, 128), fi)
                fcw.append(buf)
            pk['fc_w'] = fcw
        self._packed, self._packed_key = pk, key
        return pk

    def _conv_pack_targets(self):
        """For vqa_amd.optim.FusedAdam: {parameter data_ptr: (shape class, address of its group in the
        forward pack, address in the input-gradient pack)} for every convolution weight whose packed
        copies the optimizer kernel can rewrite (all but the stem's first), plus the shape classes'
        (cout, cin, taps).  None before the first forward on the device has laid the packs out."""
        stk, bufs = self._stacks, self._pack_bufs
        if stk is None or any(k not in bufs for k in ('w_conv1', 'w_conv2', 'wd_conv1', 'wd_conv2', 'w_proj', 'wd_proj',
                                                      'w_stem1', 'wd_stem1', 'w_cls', 'wd_cls')):
            return None
        s1, cl = self.stem[2].weight, self.classifier[0].weight
        sig = (stk['w1'].data_ptr(), s1.data_ptr(), cl.data_ptr()) + tuple(bufs[k].data_ptr() for k in sorted(bufs))
        if self._pack_targets is not None and self._pack_targets[0] == sig:
            return self._pack_targets[1], self._pack_targets[2]
        d = self._dims
        c, p_, nv = d['c'], d['p'], len(self._tokens)
        out = {}
        row33, rowp = c * c * 9, 2 * c * c                 # elements per token: fp32 weights and packs alike
        for i in range(nv):
            for w, f, b in (('w1', 'w_conv1', 'wd_conv1'), ('w2', 'w_conv2', 'wd_conv2')):
                out[stk[w].data_ptr() + 4 * i * row33] = ('c33', bufs[f].data_ptr() + 2 * i * row33, bufs[b].data_ptr() + 2 * i * row33)
            if self._binary[i]:
                out[stk['wp'].data_ptr() + 4 * i * rowp] = ('proj', bufs['w_proj'].data_ptr() + 2 * i * rowp,
                                                            bufs['wd_proj'].data_ptr() + 2 * i * rowp)
        if s1.is_contiguous() and cl.is_contiguous():
            out[s1.data_ptr()] = ('c33', bufs['w_stem1'].data_ptr(), bufs['wd_stem1'].data_ptr())
            out[cl.data_ptr()] = ('cls', bufs['w_cls'].data_ptr(), bufs['wd_cls'].data_ptr())
        shapes = {'c33': (c, c, 9), 'proj': (c, 2 * c, 1), 'cls': (p_, c, 1)}
        self._pack_targets = (sig, out, shapes)
        return out, shapes

    def _fc1_packed_buffer(self, device):
        lin = self.classifier[self._fc_idx[0]]
        fo, fi = lin.weight.shape
        b = self._fc1_packed
        if b is None or b.device != device or b.shape != (_rup(fo, 128), fi):
            b = self._fc1_packed = torch.zeros(_rup(fo, 128), fi, dtype=torch.bfloat16, device=device)
            self._fc1_fresh = False
        return b

    # ------------------------------------------------------------------ forward
