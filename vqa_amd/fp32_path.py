"""The fp32 tolerance path of ModuleNet (`net.precision = 'fp32'`): fp32 maps and weights, every
convolution, weight gradient and linear layer on v_mfma_f32_16x16x4_f32 (conv_f32.hip), layout
changes by vqa_transpose_f32.  Built for the 1e-3 comparison with the reference
(`Module_Model.py:235-248` forward, autograd's backward over it), not for speed.  A mixin of
ModuleNet (module_net.py)."""
import ctypes as C

import torch

from . import _lib, ops
from ._state import HW, _CallState
from .schedule import Plan


class _Fp32Path:
    def _weights_f32(self, dev):
        """Plain fp32 [group][tap][k][n] weight arrays for the fp32 kernels, forward and
        input-gradient (flipped taps, transposed) forms."""
        plist = self._param_groups()
        key = (tuple(p.data_ptr() for _, p in plist), tuple(p._version for _, p in plist), self._param_epoch, str(dev))
        if self._f32_cache is not None and self._f32_cache[0] == key:
            return self._f32_cache[1]
        pk = self._build_weights_f32(dev)
        self._f32_cache = (key, pk)
        return pk

    def _build_weights_f32(self, dev):
        d = self._dims
        c, p_, cin, cpad = d['c'], d['p'], d['cin'], d['cin_pad']
        nv = len(self._tokens)
        with torch.no_grad():
            def fwd(w):      # [..., o, i, 3, 3] -> [..., 9, i, o]
                return w.permute(*range(w.dim() - 4), -2, -1, -3, -4).reshape(*w.shape[:-4], -1, w.shape[-3], w.shape[-4]).contiguous()

            def bwd(w):      # [..., o, i, 3, 3] -> [..., 9 (flipped), o, i]
                return w.flip(-2, -1).permute(*range(w.dim() - 4), -2, -1, -4, -3).reshape(*w.shape[:-4], -1, w.shape[-4], w.shape[-3]).contiguous()
            w1 = torch.stack([self._modules[t].conv1.weight for t in self._tokens]).to(dev, torch.float32)
            w2 = torch.stack([self._modules[t].conv2.weight for t in self._tokens]).to(dev, torch.float32)
            wp = torch.zeros(nv, c, 2 * c, device=dev)
            bp = torch.zeros(nv, c, device=dev)
            for i, t in enumerate(self._tokens):
                if self._binary[i]:
                    wp[i] = self._modules[t].proj.weight.reshape(c, 2 * c)
                    bp[i] = self._modules[t].proj.bias
            s0w = torch.zeros(c, cpad, 3, 3, device=dev)
            s0w[:, :cin] = self.stem[0].weight
            cl = self.classifier[0].weight.reshape(p_, c)
            return {
                'w_conv1': fwd(w1), 'w_conv2': fwd(w2), 'wd_conv1': bwd(w1), 'wd_conv2': bwd(w2),
                'w_proj': wp.permute(0, 2, 1).contiguous(), 'wd_proj': wp.contiguous(),
                'b_conv1': torch.stack([self._modules[t].conv1.bias for t in self._tokens]).to(dev, torch.float32).contiguous(),
                'b_conv2': torch.stack([self._modules[t].conv2.bias for t in self._tokens]).to(dev, torch.float32).contiguous(),
                'b_proj': bp,
                'w_stem0': fwd(s0w), 'w_stem1': fwd(self.stem[2].weight.detach()), 'wd_stem1': bwd(self.stem[2].weight.detach()),
                'w_cls': cl.t().contiguous(), 'wd_cls': cl.contiguous(),
                'b_stem0': self.stem[0].bias.detach(), 'b_stem1': self.stem[2].bias.detach(),
                'b_cls': self.classifier[0].bias.detach(),
            }

    def _run_forward_f32(self, x, program, need_grad):
        d = self._dims
        dev = x.device
        N, c, p_, cin, cpad = x.size(0), d['c'], d['p'], d['cin'], d['cin_pad']
        f32t = torch.float32
        plan = Plan(program.detach().cpu().numpy(), self.vocab, c, self.wgrad_chunk, _lib.PLAN_SKIP_FUSED)
        info = plan.info
        self.used_fns = torch.from_numpy(plan.used_fns()).to(program.device)
        self.last_token_counts = plan.token_counts()
        if self.plan_ready_hook is not None:
            self.plan_ready_hook(self.last_token_counts, dev, need_grad)
        ws = self._pool.acquire()
        st = _CallState(N=N, plan=plan, f32=True, x_dtype=x.dtype, ws=ws)
        try:
            st['pk'] = pk = self._weights_f32(dev)
            blob_host = torch.from_numpy(plan.blob())
            st['blob'] = blob = ws.get('blob32', blob_host.numel(), torch.int32, dev)
            blob.copy_(blob_host)
            slots = info.slots_total if need_grad else info.slots_fwd
            st['arena'] = arena = ws.get('arena32', slots * HW * c, f32t, dev)
            # the loader's NCHW batch (Train_model.py:311) -> channels-last fp32 [N][196][cin_pad]
            xn = ws.bufs.get('x32')
            if xn is None or xn.numel() != N * HW * cpad or xn.device != dev:
                xn = ws.bufs['x32'] = torch.zeros(N * HW * cpad, dtype=f32t, device=dev)   # padding channels stay zero
            ops.transpose_f32(x.detach().to(f32t).contiguous(), xn, N, cin, HW, ld_out=cpad, out_stride=HW * cpad)
            st['x_nhwc'] = xn
            slot = (HW * c, c)
            self._conv(st, xn, arena, pk['w_stem0'], 'STEM0_TILES', 9, cpad // 128, c, (HW * cpad, cpad), slot,
                       bias=pk['b_stem0'], relu=True, f32=True)
            self._conv(st, arena, arena, pk['w_stem1'], 'STEM1_TILES', 9, c // 128, c, slot, slot,
                       bias=pk['b_stem1'], relu=True, f32=True)
            tctx = self._tree_ctx(st, pk)
            tctx.flags = _lib.TREE_F32
            _lib.check(_lib.load().vqa_tree_forward(plan.handle, C.byref(tctx), ops._stream()))
            st['c1'] = c1 = ws.get('c1_32', N * HW * p_, f32t, dev)
            self._conv(st, arena, c1, pk['w_cls'], 'CLS_FWD_TILES', 1, c // 128, p_, slot, (HW * p_, p_),
                       bias=pk['b_cls'], relu=True, f32=True)
            # Flatten (Layers.py:42-44) orders a map c*196+hw: the channels-last map is transposed per
            # sample, so the linear layers use the reference's weights as they are and their weight
            # gradients come out in the reference's layout
            flat = ws.get('flat32', N * HW * p_, f32t, dev)
            ops.transpose_f32(c1, flat, N, HW, p_)
            # linear layers (Module_Model.py:85-92) on fp32 MFMA: split-k slabs, then bias (+ ReLU)
            acts = [flat]
            prev_dim = HW * p_
            fcs = list(d['fcs']) + [d['na']]
            for li, fo in enumerate(fcs):
                lin = self.classifier[self._fc_idx[li]]
                tiles = -(-N // 64) * -(-fo // 64)
                split = max(1, min(prev_dim // 256, -(-1024 // tiles)))
                acc = ws.get('fc32_acc%d' % li, split * N * fo, f32t, dev)
                ops.gemm_f32(acts[-1], lin.weight.detach(), acc, N, fo, prev_dim, lda=prev_dim, ldb=prev_dim, ldc=fo,
                             split_k=split)
                last = li == len(fcs) - 1
                out = torch.empty(N, fo, dtype=f32t, device=dev) if last else ws.get('fc32_act%d' % li, N * fo, f32t, dev)
                ops.bias_act(acc, lin.bias.detach(), out, N, fo, fo, fo, not last, slabs=split, slab_stride=N * fo)
                if last:
                    st['logits'] = out
                else:
                    acts.append(out)
                    prev_dim = fo
            st['acts'] = acts
        except BaseException:
            st.release()
            raise
        if self.keep_state:
            self.last_state = st
        return st

    def _run_backward_f32(self, st, dlogits, want_dx=False):
        d = self._dims
        plan, pk, N = st['plan'], st['pk'], st['N']
        dev = dlogits.device
        lib = _lib.load()
        c, p_, cin, cpad, nv = d['c'], d['p'], d['cin'], d['cin_pad'], len(self._tokens)
        arena, blob, acts = st['arena'], st['blob'], st['acts']
        slot = (HW * c, c)
        fcs = list(d['fcs']) + [d['na']]
        f32 = dict(dtype=torch.float32, device=dev)
        grads = {}
        ws = st['ws']
        dims_in = [HW * p_] + list(d['fcs'])
        g = dlogits.detach().to(torch.float32).contiguous()
        g_ld = fcs[-1]
        for li in range(len(fcs) - 1, -1, -1):
            name = 'classifier.%d' % self._fc_idx[li]
            fo, fi = fcs[li], dims_in[li]
            lin = self.classifier[self._fc_idx[li]]
            # dW[fo][fi] = g^T x input activations (both stored sample-major: k-major operands), in the
            # reference's own layout since the first layer's input is the map in Flatten's order
            gw = torch.empty(fo, fi, **f32)
            ops.gemm_f32(g, acts[li], gw, fo, fi, N, lda=g_ld, ldb=fi, ldc=fi, a_kmajor=True, b_kmajor=True)
            gb = torch.zeros(fo, **f32)
            ops.col_sum(g, gb, N, fo, g_ld)
            grads[name + '.weight'], grads[name + '.bias'] = gw, gb
            # input gradient, times (input > 0): the input came out of a ReLU
            gin = ws.get('fc32_gin%d' % li, N * fi, torch.float32, dev)
            ops.gemm_f32(g, lin.weight.detach(), gin, N, fi, fo, lda=g_ld, ldb=fi, ldc=fi, b_kmajor=True, mask=acts[li])
            g, g_ld = gin, fi
        dc1 = ws.get('dc1_32', N * HW * p_, torch.float32, dev)      # back to channels-last [N][196][P]
        ops.transpose_f32(g, dc1, N, p_, HW)

        it_off, it_cnt = plan.section('SAMPLE_ITEMS')
        sec = lambda n: (blob, plan.section(n)[0])
        wg = dict(src_pix_stride=c, dy_pix_stride=c, src_slot_stride=HW * c, dy_slot_stride=HW * c, f32=True)
        gcw, gcb = torch.zeros(1, 1, c, p_, **f32), torch.zeros(p_, **f32)
        ops.wgrad_tiles(arena, dc1, (blob, it_off), it_cnt, sec('SLOTS_ROOTY'), sec('SLOTS_IDENT'), gcw, gcb,
                        taps=1, k_chunks=c // 128, n_out=p_, dw_group_stride=c * p_, total_nodes=N,
                        **dict(wg, dy_pix_stride=p_, dy_slot_stride=HW * p_))
        gw = torch.empty(p_, c, 1, 1, **f32)
        ops.unpack_conv_grad(gcw, gw, 1, p_, c, 1, 0)
        grads['classifier.0.weight'], grads['classifier.0.bias'] = gw, gcb
        self._conv(st, dc1, arena, pk['wd_cls'], 'CLS_DGRAD_TILES', 1, p_ // 128, c, (HW * p_, p_), slot,
                   mask=arena, f32=True)

        acc = {'g_conv1': torch.zeros(nv, 9, c, c, **f32), 'g_conv2': torch.zeros(nv, 9, c, c, **f32),
               'g_proj': torch.zeros(nv, 2 * c, c, **f32), 'gb_conv1': torch.zeros(nv, c, **f32),
               'gb_conv2': torch.zeros(nv, c, **f32), 'gb_proj': torch.zeros(nv, c, **f32)}
        tctx = self._tree_ctx(st, pk, acc)
        tctx.flags = _lib.TREE_F32
        self._mark('cls_bwd')
        _lib.check(lib.vqa_tree_backward(plan.handle, C.byref(tctx), ops._stream()))
        self._mark('tree_bwd')
        gw1 = torch.empty(nv, c, c, 3, 3, **f32)
        gw2 = torch.empty(nv, c, c, 3, 3, **f32)
        gwp = torch.empty(nv, c, 2 * c, 1, 1, **f32)
        ops.unpack_conv_grad(acc['g_conv1'], gw1, nv, c, c, 9, c * c * 9)
        ops.unpack_conv_grad(acc['g_conv2'], gw2, nv, c, c, 9, c * c * 9)
        ops.unpack_conv_grad(acc['g_proj'], gwp, nv, c, 2 * c, 1, 2 * c * c)
        counts = plan.token_counts()
        for i, tok in enumerate(self._tokens):
            if counts[i] == 0:
                continue
            if self._binary[i]:
                grads[tok + '.proj.weight'], grads[tok + '.proj.bias'] = gwp[i], acc['gb_proj'][i]
            grads[tok + '.conv1.weight'], grads[tok + '.conv1.bias'] = gw1[i], acc['gb_conv1'][i]
            grads[tok + '.conv2.weight'], grads[tok + '.conv2.bias'] = gw2[i], acc['gb_conv2'][i]

        self._conv(st, arena, arena, pk['wd_stem1'], 'STEM1_DGRAD_TILES', 9, c // 128, c, slot, slot, mask=arena, f32=True)
        if want_dx:
            self._input_grad(st, True)
        gs1, gs1b = torch.zeros(1, 9, c, c, **f32), torch.zeros(c, **f32)
        ops.wgrad_tiles(arena, arena, (blob, it_off), it_cnt, sec('SLOTS_S0'), sec('SLOTS_DF'), gs1, gs1b,
                        taps=9, k_chunks=c // 128, n_out=c, dw_group_stride=9 * c * c, total_nodes=N, **wg)
        gs0, gs0b = torch.zeros(1, 9, cpad, c, **f32), torch.zeros(c, **f32)
        ops.wgrad_tiles(st['x_nhwc'], arena, (blob, it_off), it_cnt, sec('SLOTS_IDENT'), sec('SLOTS_DS0'), gs0, gs0b,
                        taps=9, k_chunks=cpad // 128, n_out=c, dw_group_stride=9 * cpad * c, total_nodes=N,
                        **dict(wg, src_pix_stride=cpad, src_slot_stride=HW * cpad))
        gw = torch.empty(c, c, 3, 3, **f32)
        ops.unpack_conv_grad(gs1, gw, 1, c, c, 9, 0)
        grads['stem.2.weight'], grads['stem.2.bias'] = gw, gs1b
        gw = torch.empty(c, cpad, 3, 3, **f32)
        ops.unpack_conv_grad(gs0, gw, 1, c, cpad, 9, 0)
        grads['stem.0.weight'] = gw[:, :cin].contiguous() if cpad != cin else gw
        grads['stem.0.bias'] = gs0b

        out = []
        for name, p in self._param_groups():
            gr = grads.get(name)
            if gr is not None:
                gr = gr.reshape(p.shape)
                if self.grad_ready_hook is not None:
                    self.grad_ready_hook(name, gr)
            out.append(gr if p.requires_grad else None)
        self._params_changed()
        if self.backward_done_hook is not None:
            self.backward_done_hook()
        if not self.keep_state:
            st.release()
        return out
