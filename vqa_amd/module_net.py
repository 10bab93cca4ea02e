"""`ModuleNet`: drop-in for the reference execution engine (`Module_Model.py:95-248`).

Same constructor, same `forward(x, program) -> logits [N, num_answers]`, same `used_fns`, same
state-dict keys; the body is the MI355X path: native scheduler + grouped HIP kernels behind the
C ABI of libvqa_amd.so.  Submodules below only hold parameters under the reference's names
(`stem.0.weight`, `<token>.conv1.weight`, `<token>.proj.weight`, `classifier.3.weight`, ...);
they are never called.  There is no CPU or eager-torch fallback: a CPU tensor or a missing
library raises.
"""
import os
import time
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from . import _lib, ops, vocab as V
from .schedule import Plan
from .feed import FeatureRows

from ._state import HW, _rup, _Workspace, _WorkspacePool, _CallState  # noqa: F401
from .packing import _WeightPacking
from .fp32_path import _Fp32Path


class _ResidualBlockParams(nn.Module):
    """Parameter holder named like `Layers.ResidualBlock` (`Layers.py:11-27`)."""

    def __init__(self, dim):
        super().__init__()
        self.conv1 = nn.Conv2d(dim, dim, kernel_size=3, padding=1)
        self.conv2 = nn.Conv2d(dim, dim, kernel_size=3, padding=1)


class _ConcatBlockParams(nn.Module):
    """Parameter holder named like `ConcatBlock` (`Module_Model.py:15-35`)."""

    def __init__(self, dim):
        super().__init__()
        self.proj = nn.Conv2d(2 * dim, dim, kernel_size=1, padding=0)
        self.conv1 = nn.Conv2d(dim, dim, kernel_size=3, padding=1)
        self.conv2 = nn.Conv2d(dim, dim, kernel_size=3, padding=1)


class _Flatten(nn.Module):
    pass


class _ExecFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, program, need_grad, image_index, *params):
        ctx.net = net
        st = net._run_forward(x, program, need_grad, image_index)
        # the output must not stay reachable from ctx: output -> grad_fn -> ctx -> state -> output is
        # a reference cycle that only the cyclic GC frees, and GBs of arena would pile up meanwhile
        logits = st.pop('logits')
        if need_grad:
            ctx.state = st
        else:
            st.release()
            ctx.state = None
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        st = ctx.state
        if st is None or st.get('released'):
            # the workspace of this call went back to the pool when its backward ran (or the call was
            # made without gradients): its maps may already hold another call's values
            raise RuntimeError('vqa_amd.ModuleNet: backward through a forward call whose buffers were already '
                               'released (a second backward over the same graph is not supported)')
        grads = ctx.net._run_backward(st, dlogits, want_dx=ctx.needs_input_grad[1])
        dx = st.pop('dx', None)
        return (None, dx, None, None, None) + tuple(grads)


class ModuleNet(_WeightPacking, _Fp32Path, nn.Module):
    def __init__(self, info, vocab,
                 feature_dim=(1024, 14, 14),
                 stem_num_layers=2,
                 stem_batchnorm=False,
                 module_dim=128,
                 module_residual=True,
                 module_batchnorm=False,
                 classifier_proj_dim=512,
                 classifier_downsample='maxpool2',
                 classifier_fc_layers=(1024,),
                 classifier_batchnorm=False,
                 classifier_dropout=0):
        super().__init__()
        # Options outside the reference's defaults that would need other kernels: refuse, never
        # silently differ.
        if stem_batchnorm or module_batchnorm or classifier_batchnorm:
            raise NotImplementedError('batch-norm variants are not built for the HIP path')
        if not module_residual:
            raise NotImplementedError('module_residual=False is not built for the HIP path')
        if classifier_dropout:
            raise NotImplementedError('classifier dropout is not built for the HIP path')
        if classifier_downsample in ('maxpool12', 'maxpool4'):
            # only these two strings pool in the reference (Module_Model.py:77-82); its default
            # 'maxpool2' matches neither, so the default network has no pooling
            raise NotImplementedError('pooled classifier is not built for the HIP path')
        if stem_num_layers != 2:
            raise NotImplementedError('stem_num_layers must be 2 for the HIP path')
        if tuple(feature_dim[1:]) != (14, 14):
            raise NotImplementedError('feature maps must be 14 x 14')
        if module_dim % 128 or not classifier_proj_dim or classifier_proj_dim % 128:
            raise NotImplementedError('module_dim and classifier_proj_dim must be multiples of 128')
        if any(f % 128 for f in classifier_fc_layers) or not len(classifier_fc_layers):
            raise NotImplementedError('classifier_fc_layers must be non-empty multiples of 128')

        cin, c, p = feature_dim[0], module_dim, classifier_proj_dim
        self.stem = nn.Sequential(nn.Conv2d(cin, c, kernel_size=3, padding=1), nn.ReLU(inplace=True),
                                  nn.Conv2d(c, c, kernel_size=3, padding=1), nn.ReLU(inplace=True))
        if info:
            print('Here is my stem:')
            print(self.stem)
        num_answers = len(vocab['answer_idx_to_token'])
        layers = [nn.Conv2d(c, p, kernel_size=1), nn.ReLU(inplace=True), _Flatten()]
        prev = p * HW
        for f in classifier_fc_layers:
            layers += [nn.Linear(prev, f), nn.ReLU(inplace=True)]
            prev = f
        layers.append(nn.Linear(prev, num_answers))
        self.classifier = nn.Sequential(*layers)
        if info:
            print('Here is my classifier:')
            print(self.classifier)

        self.stem_times = []
        self.module_times = []
        self.classifier_times = []
        self.timing = False
        self.function_modules_num_inputs = {}
        self.vocab = vocab
        for fn_str in vocab['program_token_to_idx']:
            num_inputs = V.get_num_inputs(fn_str)
            self.function_modules_num_inputs[fn_str] = num_inputs
            if fn_str == 'scene' or num_inputs == 1:
                mod = _ResidualBlockParams(c)
            else:
                mod = _ConcatBlockParams(c)
            self.add_module(fn_str, mod)
        self.save_module_outputs = False

        self._dims = dict(cin=cin, cin_pad=_rup(cin, 128), c=c, p=p, fcs=tuple(classifier_fc_layers),
                          na=num_answers, na_pad=_rup(num_answers, 128))
        i2t = vocab['program_idx_to_token']
        self._tokens = [i2t[i] for i in range(len(i2t))]
        self._binary = [V.get_num_inputs(t) == 2 for t in self._tokens]
        self._fc_idx = [i for i, m in enumerate(self.classifier) if isinstance(m, nn.Linear)]
        self._pool = _WorkspacePool()
        self._side = None              # stream for host<->device traffic that must not queue behind compute
        self._lay = None               # stream of the input layout pass (see _run_forward)
        self.stem_pieces = int(os.environ.get('VQA_STEM_PIECES', '2'))
        self._stem_tile_cache = {}
        self._packed_key = None
        self._packed_stale = False
        self._plist = None            # cached _param_groups()
        self._stacks = None           # the per-token parameters' shared [V, ...] arrays
        self._packed = {}
        self._pack_bufs = {}          # persistent destinations of the packed convolution weights
        self._pack_targets = None
        self._param_epoch = 0         # see _params_changed
        self._f32_cache = None        # (key, weight arrays of the fp32 path)
        self._conv_packs_fresh = False  # set by vqa_amd.optim.FusedAdam: it rewrote those copies with its update
        self._fc1_packed = None       # persistent bf16 copy of the first linear layer's weight
        self.segment_marks = None     # diagnostics: see _mark
        self._fc1_fresh = False       # set by vqa_amd.optim.FusedAdam: that copy already holds the updated weight
        # set by vqa_amd.dp.DataParallelExecutor(shard_big_update=True): this rank's rows of the first
        # linear layer's averaged gradient; the all-gather of the updated bf16 rows still in flight;
        # whether the fp32 master copy is current on its owning ranks only
        self._fc1_grad_shard = None
        self._group_rows = None                  # parameter kind -> {parameter name: row of the per-token gradient array}
        self._sliced_update_optimizer = None     # weak reference to a FusedAdam(executor=self), set by it
        self._fc1_gather = None
        self._fc1_master_stale = False
        self.used_fns = None
        self.wgrad_chunk = 32
        # module width 128: both convolutions of a residual body run in one kernel; False selects
        # the two-launch path (bit-identical results)
        self.fuse_residual_body = True
        # the last linear layer (a hidden layer in front of it, at most 64 answers) forward in one launch and
        # backward in two (vqa_head_forward / vqa_head_backward) instead of GEMM + bias and cast + column sum +
        # two GEMMs + column sum; False selects those (same operands, sums in another order)
        self.fuse_head = True
        # fp32 NCHW features go into the stem's first convolution as they are (vqa_stem_nchw: the layout change
        # inside the kernel, module width 128); False selects the layout pass + vqa_conv_tiles (same bits)
        self.fuse_stem_layout = os.environ.get('VQA_FUSE_STEM', '1') != '0'
        # run nodes that have slack in later tree levels so that launches fill whole rounds of the device
        # (training calls only: in a forward-only call the device waits for the tile lists, and the
        # list scheduling costs the host more than the evener launches give back - 11 % of the call
        # on an untrained sampler's rows, against 1.5 % gained on well-formed programs)
        self.balance_levels = True
        # 'bf16': bf16 storage and MFMA operands, fp32 accumulation (the throughput path);
        # 'fp32': fp32 storage and fp32 MFMA for every convolution and linear layer
        # (the tolerance configuration: results within 1e-3 of the reference's fp32 arithmetic)
        self.precision = 'bf16'
        self.grad_ready_hook = None   # set by the data-parallel wrapper: called as (name, grad tensor)
        self.plan_ready_hook = None   # likewise: (token counts of the batch, device, whether a backward pass will follow), right after the programs are parsed
        self.grad_group_hook = None   # likewise: (array [V, ...], {parameter name: row}) per kind of per-token parameter
        self.backward_done_hook = None
        self.last_token_counts = None
        # programs that are in device memory (a sampler's output) are parsed and bucketed there, with
        # no copy to the host, when no gradient is asked for (bf16, module width 128): same logits bit
        # for bit as with the host scheduler.  used_fns stays a device tensor, and a
        # token id outside the vocabulary cannot raise from an asynchronous launch: it sets
        # schedule_error (see raise_for_schedule_error)
        self.schedule_on_device = False
        self.device_spread_levels = True     # rows with slack run later by differing shares of it: evener launches
        self.device_level_workgroups = 512     # two per CU: all resident at once
        # which of the two weight gradients that feed nothing else in backward run on the side stream
        # (A: the first linear layer's; B: that, then the classifier convolution's; C: only the latter)
        self.backward_order = os.environ.get('VQA_BWD_ORDER', 'A')
        self.schedule_error = None
        self._arity_dev = None
        self.keep_state = False       # tests: keep the last call's buffers in self.last_state
        # list-of-dicts programs: False = the reference's behaviour (AttributeError); True = run them
        self.accept_program_lists = False
        self.last_state = None

    # ------------------------------------------------------------------ parameters
    def _param_groups(self):
        """Ordered (name, parameter) list handed to the autograd function."""
        if self._plist is not None:
            return self._plist
        out = [('stem.0', self.stem[0]), ('stem.2', self.stem[2]), ('classifier.0', self.classifier[0])]
        out += [('classifier.%d' % i, self.classifier[i]) for i in self._fc_idx]
        plist = []
        for name, m in out:
            plist += [(name + '.weight', m.weight), (name + '.bias', m.bias)]
        for tok in self._tokens:
            m = self._modules[tok]
            if isinstance(m, _ConcatBlockParams):
                plist += [(tok + '.proj.weight', m.proj.weight), (tok + '.proj.bias', m.proj.bias)]
            plist += [(tok + '.conv1.weight', m.conv1.weight), (tok + '.conv1.bias', m.conv1.bias),
                      (tok + '.conv2.weight', m.conv2.weight), (tok + '.conv2.bias', m.conv2.bias)]
        self._plist = plist
        return plist

    def _apply(self, fn, *args, **kwargs):
        # .cuda() / .to() / .float(): parameters get new storage; everything derived from it is rebuilt
        out = super()._apply(fn, *args, **kwargs)
        self._plist, self._stacks, self._packed_key = None, None, None
        return out

    def forward(self, x, program):
        N = x.size(0)
        assert N == len(program)
        if type(program) is list or type(program) is tuple:
            # the reference's list-of-dicts path reads self.function_modules, which its __init__
            # never sets (Module_Model.py:136,149,186): it raises AttributeError, and so do we
            if not self.accept_program_lists:
                raise AttributeError("'ModuleNet' object has no attribute 'function_modules'")
            # opt-in: what that path was meant to do (Module_Model_tmp.py:142-167), through the same
            # schedule as token rows
            program = torch.from_numpy(V.lists_to_prefix_rows(program, self.vocab))
        if type(program) is torch.Tensor and program.dim() == 2:
            pass
        else:
            raise ValueError('Unregonized program format')
        if not x.is_cuda:
            raise RuntimeError('vqa_amd.ModuleNet runs on an MI355X only: features must be a CUDA/HIP '
                               'tensor (no CPU fallback)')
        if getattr(self, '_is_replica', False):
            # torch.nn.DataParallel over several GPUs (Functions.py:122-126) hands forward() a shallow
            # copy whose parameters are broadcast views; the executor's packed weights, plans and
            # workspaces belong to one device
            raise RuntimeError('vqa_amd.ModuleNet cannot run as a torch.nn.DataParallel replica: run one process '
                               'per GPU and wrap the executor in vqa_amd.dp.DataParallelExecutor')
        if N == 0:
            # the reference gets as far as torch.cat of no module outputs (Module_Model.py:232)
            raise RuntimeError('torch.cat(): expected a non-empty list of Tensors (empty batch)')
        _lib.load()
        params = [p for _, p in self._param_groups()]
        # the reference's forward is differentiable with respect to the features as well
        # (Module_Model.py:235-248): x.requires_grad alone asks for a backward pass
        need_grad = torch.is_grad_enabled() and (any(p.requires_grad for p in params) or
                                                 (not isinstance(x, FeatureRows) and x.requires_grad))
        if isinstance(x, FeatureRows):
            if self.precision != 'bf16':
                x = x.dense()
        elif not need_grad and N > 1 and x.stride(0) == 0 and self.precision == 'bf16':
            # MAPO candidate scoring (Train_model.py:343-344): feats[i].unsqueeze(0).expand(K, ...) is
            # one image under K programs.  The stem runs once and subtrees the programs share are
            # computed once.
            return self.score_candidates(x[:1], program, torch.zeros(N, dtype=torch.int32))
        return _ExecFn.apply(self, x, program, need_grad, None, *params)

    def score_candidates(self, x, program, image_index):
        """Inference-only scoring of candidate programs that share images: `x` holds the M distinct
        feature maps [M, Cin, 14, 14], `program` the N programs [N, T] and image_index[n] the image
        program n runs on.  Returns logits [N, A], the same values as forward() on x[image_index]
        (same kernels, same order of operations per node).  One call can score the candidates of a
        whole batch of questions; the reference's loop calls the executor once per question."""
        if not (type(program) is torch.Tensor and program.dim() == 2):
            raise ValueError('Unregonized program format')
        if not x.is_cuda:
            raise RuntimeError('vqa_amd.ModuleNet runs on an MI355X only: features must be a CUDA/HIP '
                               'tensor (no CPU fallback)')
        if self.precision != 'bf16':
            raise NotImplementedError('shared-image scoring is built for the bf16 path')
        if (self.schedule_on_device and program.is_cuda and program.device == x.device and self._dims['c'] == 128 and
                self.fuse_residual_body):
            # programs stay on the device (see schedule_on_device): the stem and the scene module run
            # once per image, every program's own nodes run (common subtrees of an image's programs
            # are not searched for on the device); an image index out of range raises from
            # raise_for_schedule_error()
            img = torch.as_tensor(image_index).detach().to(x.device, torch.int32, non_blocking=True).contiguous()
            if img.numel() != program.size(0):
                raise ValueError('image_index must have one entry per program')
            _lib.load()
            with torch.no_grad():
                return _ExecFn.apply(self, x, program, False, img, *[p for _, p in self._param_groups()])
        img = torch.as_tensor(image_index).detach().to('cpu', torch.int32).contiguous()
        if img.numel() != program.size(0):
            raise ValueError('image_index must have one entry per program')
        if img.numel() and (int(img.min()) < 0 or int(img.max()) >= x.size(0)):
            raise IndexError('image_index out of range')
        _lib.load()
        with torch.no_grad():
            return _ExecFn.apply(self, x, program, False, img.numpy(), *[p for _, p in self._param_groups()])

    def _conv(self, st, src, dst, w, tiles_sec, taps, k_chunks, n_out, src_geo, dst_geo, **kw):
        off, cnt = st['plan'].section(tiles_sec)
        ops.conv_tiles(src, dst, w, st['blob'], cnt, taps=taps, k_chunks=k_chunks, n_out=n_out,
                       src_slot_stride=src_geo[0], src_pix_stride=src_geo[1],
                       dst_slot_stride=dst_geo[0], dst_pix_stride=dst_geo[1],
                       w_group_stride=0, tiles_offset=off, **kw)

    def _tree_ctx(self, st, pk, grads=None):
        ctx = _lib.TreeCtx()
        ctx.arena, ctx.blob = st['arena'].data_ptr(), st['blob'].data_ptr()
        for k in ('w_conv1', 'w_conv2', 'w_proj', 'wd_conv1', 'wd_conv2', 'wd_proj',
                  'b_conv1', 'b_conv2', 'b_proj'):
            setattr(ctx, k, pk[k].data_ptr())
        if grads is not None:
            for k in ('g_conv1', 'g_conv2', 'g_proj', 'gb_conv1', 'gb_conv2', 'gb_proj'):
                setattr(ctx, k, grads[k].data_ptr())
        ctx.C, ctx.V = self._dims['c'], len(self._tokens)
        ctx.flags = 0 if self.fuse_residual_body else _lib.TREE_UNFUSED
        return ctx

    def _mark(self, name):
        """With `segment_marks` set to a list, appends (name, timing event on the current stream,
        host time) at fixed points of a step: where the device and the host spend a step
        (scripts/step_segments.py)."""
        if self.segment_marks is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream())
            self.segment_marks.append((name, ev, time.perf_counter()))

    def _stem_input(self, x, ws, M, dev):
        """The stem's input as bf16 [rows][196][cin_pad] plus, when the batch is a set of rows of a
        resident table, their int32 row numbers.  Three forms are read in place, without a layout
        pass: rows of a vqa_amd.feed.ResidentFeatures table, and a bf16 tensor in channels-last
        memory format (what vqa_amd.feed.HostFeed yields).  Anything else is the reference's fp32
        NCHW batch (Train_model.py:311) and goes through vqa_nchw_to_nhwc."""
        d = self._dims
        cin, cpad = d['cin'], d['cin_pad']
        if isinstance(x, FeatureRows):
            if cin != cpad or x.table.size(2) != cin or x.table.dtype != torch.bfloat16 or not x.table.is_contiguous():
                raise ValueError('feature table must be contiguous bf16 [R, 196, %d]' % cpad)
            return x.table, x.index, None
        if (x.dtype == torch.bfloat16 and cin == cpad and x.stride() == (HW * cin, 1, 14 * cin, cin)):
            return x.detach(), None, None
        xs = x.detach().to(torch.float32).contiguous()
        xn = ws.get('x_nhwc', M * HW * cpad, torch.bfloat16, dev)
        return xn, None, xs           # the layout pass is the caller's: it pipelines it with the first convolution

    def _stem_tiles(self, N, dev):
        """Tile lists of the two stem convolutions (image i -> S0 slot N+i -> F slot i) and, behind
        them, of the scene module (F slot i -> H slot 3N+i -> Y slot 2N+i, 12 words a tile): every
        program's leaves are the scene module applied to the stem output (Module_Model.py:115-117,
        171-176), one computation per image.  These depend on the batch size only, so the stem and
        the scene level can be enqueued before the programs are parsed."""
        key = (N, str(dev))
        t = self._stem_tile_cache.get(key)
        if t is None:
            rows = np.full((2, N, 8), -1, dtype=np.int32)
            i = np.arange(N, dtype=np.int32)
            rows[0, :, 0], rows[0, :, 2] = i, N + i
            rows[1, :, 0], rows[1, :, 2] = N + i, i
            rows[:, :, 7] = 0
            scene = np.full((N, 12), -1, dtype=np.int32)
            scene[:, 0], scene[:, 2], scene[:, 3], scene[:, 4] = i, 2 * N + i, 3 * N + i, i
            scene[:, 7] = self.vocab['program_token_to_idx']['scene']
            self._stem_tile_cache.clear()
            t = self._stem_tile_cache[key] = torch.from_numpy(
                np.concatenate([rows.reshape(-1), scene.reshape(-1)])).to(dev)
        return t

    def _run_forward(self, x, program, need_grad, image_index=None):
        """Order matters for the host/device overlap: the programs start their way to the host on a
        side stream, the stem (which does not depend on them) is enqueued, and only then does the
        host wait for the programs, parse them and upload the tile lists - under the stem kernels
        instead of in front of an idle device."""
        d = self._dims
        dev = x.device
        # M feature maps, N programs: M == N unless programs share images (score_candidates)
        M, N, c, p_, cpad = x.size(0), program.size(0), d['c'], d['p'], d['cin_pad']
        if x.dim() != 4 or tuple(x.shape[1:]) != (d['cin'], 14, 14):
            raise ValueError('features must be [N, %d, 14, 14]' % d['cin'])
        if self.precision == 'fp32':
            return self._run_forward_f32(x, program, need_grad)
        if self.precision != 'bf16':
            raise ValueError("precision must be 'bf16' or 'fp32'")
        T = program.size(1)
        ws = self._pool.acquire()
        st = _CallState(N=N, ws=ws, x_dtype=x.dtype if isinstance(x, torch.Tensor) else torch.float32)
        try:
            cur = torch.cuda.current_stream(dev)
            if self._side is None or self._side.device != dev:
                self._side = torch.cuda.Stream(dev)
            side = self._side
            if ws.host_event is not None:
                ws.host_event.synchronize()      # earlier transfers through this set's pinned buffers are over
            hp = ws.bufs.get('prog_host')
            if hp is None or hp.numel() < N * T:
                hp = ws.bufs['prog_host'] = torch.empty(N * T + N * T // 4, dtype=torch.int64, pin_memory=True)
            hprog = hp[:N * T].view(N, T)
            self._mark('fwd_begin')
            # the side stream starts behind everything already enqueued here: whatever produced the
            # programs, and the previous call's kernels that still read this set's tile lists
            ready = torch.cuda.Event()
            ready.record(cur)
            side.wait_event(ready)
            fused = c == 128 and self.fuse_residual_body
            dev_images = isinstance(image_index, torch.Tensor) and image_index.is_cuda
            devs = (self.schedule_on_device and fused and not need_grad and (image_index is None or dev_images) and
                    program.is_cuda and program.device == dev)
            if devs:
                # the programs stay where they are: parsed and bucketed by kernels on the side stream,
                # under the stem
                prog_dev = program.detach().to(torch.int64).contiguous()
                arrived = None
                sched, used = self._device_schedule_buffers(ws, N, T, dev)
                t2i = self.vocab['program_token_to_idx']
                with torch.cuda.stream(side):
                    ops.devsched_build(prog_dev, self._arity_dev, used, sched, t2i['<NULL>'], t2i['<START>'], t2i['scene'],
                                       spread=self.device_spread_levels, image_of_row=image_index if dev_images else None,
                                       num_images=M)
                    sched_done = torch.cuda.Event()
                    sched_done.record(side)
                prog_dev.record_stream(side)
                if dev_images:
                    image_index.record_stream(side)
            elif program.is_cuda:
                prog_dev = program.detach().to(torch.int64).contiguous()
                with torch.cuda.stream(side):
                    hprog.copy_(prog_dev, non_blocking=True)
                    arrived = torch.cuda.Event()
                    arrived.record(side)
            else:
                hprog.copy_(program.detach().to(torch.int64))
                arrived = None

            # refreshing the packed weights (after an optimizer step) is device work the host does
            # not wait for either: it goes behind the program download's starting point
            st['pk'] = pk = self._pack_weights(dev)
            # stem: two conv3x3 + bias + ReLU over the whole batch (Module_Model.py:55-64,239)
            slot = (HW * c, c)
            stem_elems = 4 * M * HW * c      # stem maps S0, F and the scene module's H, Y per image
            # (with the schedule made on the device the host does not learn the node counts: slots
            # have fixed strides, a row has at most T nodes that are not the scene module)
            dev_slots = 4 * M + 3 * N * T
            arena_full = ws.get_capacity('arena', dev_slots * HW * c if devs else stem_elems, torch.bfloat16, dev)
            xn, rows, xs = self._stem_input(x, ws, M, dev)
            st['x_nhwc'], st['x_rows'] = xn, rows
            stem_tiles = self._stem_tiles(M, dev)
            tiles0 = stem_tiles
            if rows is not None:       # maps read in place from a resident table: source slot = table row
                tiles0 = ws.get('stem0_tiles', 8 * M, torch.int32, dev)
                tiles0.copy_(stem_tiles[:8 * M])
                tiles0.view(M, 8)[:, 0] = rows
            geo = dict(taps=9, n_out=c, dst_slot_stride=slot[0], dst_pix_stride=slot[1], w_group_stride=0, relu=True)

            def conv0(lo, hi):
                ops.conv_tiles(xn, arena_full, pk['w_stem0'], tiles0, hi - lo, k_chunks=cpad // 128,
                               src_slot_stride=HW * cpad, src_pix_stride=cpad, bias=pk['b_stem0'],
                               tiles_offset=8 * lo, **geo)
            if xs is None:
                conv0(0, M)
            elif self.fuse_stem_layout and c == 128 and cpad == d['cin'] and cpad % 128 == 0:
                # the layout change happens inside the convolution kernel; the bf16 channels-last copy is
                # written only when a backward pass will read it (the first convolution's weight gradient)
                ops.stem_nchw(xs, xn if need_grad else None, arena_full, pk['w_stem0'], pk['b_stem0'], M, cpad,
                              slot[0], M, slot[1], relu=True)
            else:
                # fp32 NCHW -> bf16 channels-last is HBM-bound, the convolution behind it MFMA-bound:
                # the batch goes through in pieces, the layout pass of piece k + 1 on a stream of its
                # own under the convolution of piece k
                cin = d['cin']
                pieces = self.stem_pieces if M >= 64 * self.stem_pieces else 1
                cuts = [M * k // pieces for k in range(pieces + 1)]
                xn2 = xn.view(M, HW * cpad)
                if pieces > 1 and (self._lay is None or self._lay.device != dev):
                    self._lay = torch.cuda.Stream(dev)
                ops.nchw_to_nhwc(xs[:cuts[1]], xn2[:cuts[1]], cuts[1], cin, cpad)
                for k in range(pieces):
                    if k + 1 < pieces:
                        go = torch.cuda.Event()
                        go.record(cur)                 # the layout of piece k is done: the next one may start
                        with torch.cuda.stream(self._lay):
                            self._lay.wait_event(go)
                            ops.nchw_to_nhwc(xs[cuts[k + 1]:cuts[k + 2]], xn2[cuts[k + 1]:cuts[k + 2]],
                                             cuts[k + 2] - cuts[k + 1], cin, cpad)
                            laid = torch.cuda.Event()
                            laid.record(self._lay)
                    conv0(cuts[k], cuts[k + 1])
                    if k + 1 < pieces:
                        cur.wait_event(laid)
                if pieces > 1:
                    xs.record_stream(self._lay)
            ops.conv_tiles(arena_full, arena_full, pk['w_stem1'], stem_tiles, M, k_chunks=c // 128,
                           src_slot_stride=slot[0], src_pix_stride=slot[1], bias=pk['b_stem1'],
                           tiles_offset=8 * M, **geo)
            if fused:       # the scene level needs no plan either
                ops.resblock_tiles(arena_full, pk['w_conv1'], pk['w_conv2'], stem_tiles, M,
                                   bias_first=pk['b_conv1'], bias_second=pk['b_conv2'], relu=True,
                                   tiles_offset=16 * M)

            self._mark('scene')
            if devs:
                cur.wait_event(sched_done)
                # Module_Model.py:227,231: a float tensor on the program's device
                self.used_fns = used
                self.schedule_error = sched.error[:1].clone()      # the workspace is reused by later calls
                self.last_token_counts = None
                st['plan'] = plan = None
                st['arena'] = arena_full[:dev_slots * HW * c]
            else:
                sched = None
                # programs -> plan -> tile lists on the device
                if arrived is not None:
                    arrived.synchronize()
                # The device is about to run out of work (the stem and the scene level are short), so
                # what stands between the programs and the first level's launch is kept to the
                # scheduler itself: it writes the tile lists straight into pinned memory, their
                # forward part goes up at once, and everything else that the plan feeds waits until
                # the tree's launches are enqueued.  The backward part is written and sent after the
                # forward launches.
                want = Plan.blob_store_words(N, T, len(self.vocab['program_token_to_idx']))
                host = ws.bufs.get('blob_host')
                if host is None or host.numel() < want:
                    host = ws.bufs['blob_host'] = torch.empty(want, dtype=torch.int32, pin_memory=True)
                plan = Plan(hprog.numpy(), self.vocab, c, self.wgrad_chunk,
                            (_lib.PLAN_SKIP_SEPARATE if fused else _lib.PLAN_SKIP_FUSED) |
                            (_lib.PLAN_BALANCE if self.balance_levels and need_grad else 0) | _lib.PLAN_DEFER_BACKWARD,
                            image_index=image_index, num_images=M if image_index is not None else None,
                            # (a plan that outlives the call - keep_state - keeps its tile lists in memory of
                            # its own: the pinned buffer belongs to the workspace set and is written again
                            # two calls later)
                            blob_store=None if self.keep_state else host)
                st['plan'] = plan
                info = plan.info
                nb, bound = info.blob_words, info.blob_words_bound
                if not info.blob_lent:
                    plan.blob_range_into(host, 0, nb)
                st['blob'] = ws.get('blob', bound, torch.int32, dev)
                with torch.cuda.stream(side):
                    st['blob'][:nb].copy_(host[:nb], non_blocking=True)
                    ws.host_event = torch.cuda.Event()
                    ws.host_event.record(side)
                cur.wait_event(ws.host_event)
                self._mark('blob')

                def after_tree():
                    self.last_token_counts = plan.token_counts()
                    if self.plan_ready_hook is not None:
                        self.plan_ready_hook(self.last_token_counts, dev, need_grad)
                    hused = hp[:(N * T + 1) // 2].view(torch.float32)[:N * T].view(N, T)   # programs are parsed: reuse
                    plan.used_fns_into(hused)
                    # Module_Model.py:227,231: a float tensor on the program's device
                    used = torch.empty(N, T, dtype=torch.float32, device=program.device)
                    if used.is_cuda:
                        with torch.cuda.stream(side):
                            used.copy_(hused, non_blocking=True)
                            used.record_stream(side)
                            ws.host_event = torch.cuda.Event()
                            ws.host_event.record(side)
                    else:
                        used.copy_(hused)
                    self.used_fns = used
                slots = info.slots_total if need_grad else info.slots_fwd
                arena_big = ws.get_capacity('arena', slots * HW * c, torch.bfloat16, dev)
                if arena_big.data_ptr() != arena_full.data_ptr():     # grew: carry the stem output over
                    arena_big[:stem_elems].copy_(arena_full[:stem_elems])
                st['arena'] = arena = arena_big[:slots * HW * c]
        except BaseException:
            st.release()
            raise
        # module trees, level by level
        arena = st['arena']
        if devs:
            # the launch reads the level's tile range on the device and its workgroups share it; a row
            # has at most T // L nodes of level L, so more workgroups than that are never of use
            for L in range(1, T + 1):
                ops.resblock_tiles(arena, pk['w_conv1'], pk['w_conv2'], sched.tiles,
                                   min(N * (T // L), self.device_level_workgroups),
                                   bias_first=pk['b_conv1'], bias_second=pk['b_conv2'], relu=True,
                                   w_pre=pk['w_proj'], bias_pre=pk['b_proj'], num_pre=1,
                                   level_info=sched.level_info, level=L)
        else:
            tctx = self._tree_ctx(st, pk)
            if fused:
                tctx.flags |= _lib.TREE_SKIP_LEVEL0
            _lib.check(_lib.load().vqa_tree_forward(plan.handle, C.byref(tctx), ops._stream()))
            after_tree()
        self._mark('tree_fwd')
        # classifier (Module_Model.py:66-93, default: no pooling): conv1x1 + ReLU on the gathered
        # roots, flatten, linear layers
        np128 = _rup(N, 128)
        st['c1'] = c1 = ws.get('c1', np128 * HW * p_, torch.bfloat16, dev)
        if np128 != N:
            c1.view(np128, HW * p_)[N:].zero_()   # padding rows feed GEMM tiles: keep them finite
        if devs:
            ops.conv_tiles(arena, c1, pk['w_cls'], sched.cls_tiles, N, taps=1, k_chunks=c // 128, n_out=p_,
                           src_slot_stride=slot[0], src_pix_stride=slot[1], dst_slot_stride=HW * p_,
                           dst_pix_stride=p_, w_group_stride=0, bias=pk['b_cls'], relu=True)
        else:
            self._conv(st, arena, c1, pk['w_cls'], 'CLS_FWD_TILES', 1, c // 128, p_, slot, (HW * p_, p_),
                       bias=pk['b_cls'], relu=True)
        acts = [c1]
        prev_dim = HW * p_
        fcs = list(d['fcs']) + [d['na']]
        self._wait_fc1_gather()       # the first linear layer's bf16 copy is read from here on
        for li, fo in enumerate(fcs):
            lin = self.classifier[self._fc_idx[li]]
            last = li == len(fcs) - 1
            # split-k so that the launch fills the device: 512 workgroups of the 128 x 128-tile kernel
            # (two per CU), or 256 of the 256 x 256-tile kernel that large aligned problems take
            if N % 256 == 0 and fo % 256 == 0 and (N // 256) * (fo // 256) * 8 >= 128:
                split = max(1, min(prev_dim // 64, 256 // ((N // 256) * (fo // 256))))
            else:
                tiles = (np128 // 128) * (_rup(fo, 128) // 128)
                split = max(1, min(prev_dim // 64, 16, -(-512 // tiles)))
            if last and self._head_fused(fcs, [HW * p_] + list(d['fcs'])):
                logits = torch.empty(N, fo, dtype=torch.float32, device=dev)
                ops.head_forward(acts[-1], prev_dim, pk['fc_w'][li], prev_dim, lin.bias.detach(), logits, N, prev_dim, fo)
                st['logits'] = logits
                break
            acc = ws.get('fc_acc%d' % li, split * N * fo, torch.float32, dev)
            ops.gemm(acts[-1], pk['fc_w'][li], acc, N, fo, prev_dim, lda=prev_dim, ldb=prev_dim, ldc=fo,
                     split_k=split)
            if last:
                logits = torch.empty(N, fo, dtype=torch.float32, device=dev)
                ops.bias_act(acc, lin.bias.detach(), logits, N, fo, fo, fo, False, slabs=split, slab_stride=N * fo)
                st['logits'] = logits
            else:
                a = ws.get('fc_act%d' % li, np128 * fo, torch.bfloat16, dev)
                if np128 != N:
                    a.view(np128, fo)[N:].zero_()
                ops.bias_act(acc, lin.bias.detach(), a, N, fo, fo, fo, True, slabs=split, slab_stride=N * fo)
                acts.append(a)
                prev_dim = fo
        st['acts'] = acts
        self._mark('cls_fwd')
        if need_grad:
            plan.finish()
            total = plan.info.blob_words
            if not plan.info.blob_lent:
                plan.blob_range_into(host, nb, total - nb)
            with torch.cuda.stream(side):
                st['blob'][nb:total].copy_(host[nb:total], non_blocking=True)
                ws.host_event = torch.cuda.Event()
                ws.host_event.record(side)
            st['blob_ready'] = ws.host_event
        if self.keep_state:
            self.last_state = st
        return st

    def _head_fused(self, fcs, dims_in):
        """Whether the last linear layer runs as vqa_head_forward / vqa_head_backward."""
        return (self.fuse_head and len(fcs) >= 2 and fcs[-1] <= 64 and dims_in[-1] % 64 == 0 and
                dims_in[-1] <= 8192)

    def _device_schedule_buffers(self, ws, N, T, dev):
        """Workspace of vqa_devsched_build for N programs of T tokens, the arity table on the device,
        and a fresh used_fns tensor."""
        t2i = self.vocab['program_token_to_idx']
        nv = len(self._tokens)
        if self._arity_dev is None or self._arity_dev.device != dev:
            ar = np.asarray(V.arity_table(self.vocab), dtype=np.int8)
            zero = [i for i in range(nv) if ar[i] == 0 and i != t2i['scene']]
            if zero or ar.min() < 0 or ar.max() > 2:
                raise ValueError('the device scheduler needs arities 1 or 2 for every token but scene')
            self._arity_dev = torch.from_numpy(ar).to(dev)
        sched = ops.DeviceSchedule(ws.get('devsched', ops.DeviceSchedule.words(N, T, nv), torch.int32, dev), N, T, nv)
        return sched, torch.empty(N, T, dtype=torch.float32, device=dev)

    def raise_for_schedule_error(self):
        """After a forward with schedule_on_device: waits for that call's scheduling kernels and raises
        KeyError if a program held a token id outside the vocabulary - what the reference's dict
        lookup (Module_Model.py:200) and the host scheduler raise at once.  The logits of such a
        call are those of a lone scene module for the offending rows."""
        e = self.schedule_error
        flags = int(e.item()) if e is not None else 0
        if flags & 1:
            raise KeyError('program token id outside the vocabulary')
        if flags & 2:
            raise IndexError('image_index out of range')
        if flags & 4:
            raise ValueError('the arity table gives a module other than scene no input or more than two')

    def state_dict(self, *args, **kwargs):
        if self._fc1_master_stale:
            raise RuntimeError('the first linear layer is updated in slices (DataParallelExecutor(shard_big_update=True)): '
                               'its fp32 copy is current on the owning ranks only - call gather_sharded_state() on '
                               'every rank (the wrapper\'s state_dict() does) before reading the state')
        return super().state_dict(*args, **kwargs)

    def _wait_fc1_gather(self):
        """The current stream waits for the exchange of the first linear layer's updated bf16 rows
        (DataParallelExecutor.gather_big_packed) if one is in flight."""
        w = self._fc1_gather
        if w is not None:
            w.wait()
            self._fc1_gather = None

    def _params_changed(self):
        """Parameters are about to change without their version counters moving (a backward pass is
        followed by an optimizer step, and torch's fused optimizers and vqa_amd.optim.FusedAdam do not
        bump versions): every cached copy derived from them is out of date."""
        self._packed_stale = True
        self._param_epoch += 1

    # ------------------------------------------------------------------ fp32 path
    def stored_activation(self, st, kind, index=0):
        """A stored forward activation of call `st` as an fp32 CPU tensor in the reference's NCHW
        layout: kinds 'S0', 'F', 'C1', ('FC', layer) are whole-batch, 'H', 'Y' take a node id and
        'P' a binary-node index (execution order).  For tests and debugging."""
        d, plan, N = self._dims, st['plan'], st['N']
        c = d['c']
        arena = st['arena'].view(-1, HW, c)

        def nchw(t, ch):
            return t.to(torch.float32).cpu().reshape(-1, HW, ch).permute(0, 2, 1).reshape(-1, ch, 14, 14)
        if kind in ('S0', 'F'):
            s0 = plan.slot(kind, 0)
            return nchw(arena[s0:s0 + N], c)
        if kind in ('H', 'Y'):
            return nchw(arena[plan.slot(kind, index)], c)
        if kind == 'P':
            return nchw(arena[plan.slot('P', index)], c)
        if kind == 'C1':
            return nchw(st['c1'].view(-1, HW, d['p'])[:N], d['p'])
        if isinstance(kind, tuple) and kind[0] == 'FC':
            a = st['acts'][kind[1] + 1]
            return a.view(-1, d['fcs'][kind[1]])[:N].to(torch.float32).cpu()
        raise KeyError(kind)

    # ------------------------------------------------------------------ backward
    def _input_grad(self, st, f32):
        """d(loss)/d(features) = dgrad_conv0(dS0): the stem's first convolution run backwards over the
        gradient of its (masked) output, `n_out` = the input channels.  Only computed when the
        caller's features require a gradient (the reference's callers never do: the features come
        from the loader, Train_model.py:311).  Leaves st['dx'], [N, Cin, 14, 14] in x's dtype."""
        d = self._dims
        plan, N = st['plan'], st['N']
        c, cin, cpad = d['c'], d['cin'], d['cin_pad']
        dev = st['arena'].device
        w = self.stem[0].weight.detach()
        if cpad != cin:
            wpad = torch.zeros(c, cpad, 3, 3, dtype=torch.float32, device=dev)
            wpad[:, :cin] = w
            w = wpad
        tiles = np.full((N, 8), -1, dtype=np.int32)
        tiles[:, 0] = [plan.slot('dS0', i) for i in range(N)]
        tiles[:, 2] = np.arange(N, dtype=np.int32)
        tiles[:, 7] = 0
        tiles = torch.from_numpy(tiles.reshape(-1)).to(dev)
        geo = dict(taps=9, k_chunks=c // 128, n_out=cpad, src_slot_stride=HW * c, src_pix_stride=c,
                   dst_slot_stride=HW * cpad, dst_pix_stride=cpad, w_group_stride=0)
        out = torch.empty(N, cpad, 14, 14, dtype=torch.float32, device=dev)
        if f32:
            # [9 taps flipped][k = out channel of the conv][n = its in channel]
            wd = w.flip(-2, -1).permute(2, 3, 0, 1).reshape(9, c, cpad).contiguous()
            dxn = torch.empty(N * HW * cpad, dtype=torch.float32, device=dev)
            ops.conv_tiles(st['arena'], dxn, wd, tiles, N, f32=True, **geo)
            ops.fc_grad_from_nhwc(dxn, out, N, cpad)
        else:
            wd = ops.pack_conv_weight(w.contiguous(), 1, cpad, c, 9, 0, 9, cpad * 9, flip=True)
            dxn = torch.empty(N * HW * cpad, dtype=torch.bfloat16, device=dev)
            ops.conv_tiles(st['arena'], dxn, wd, tiles, N, **geo)
            ops.nhwc_to_nchw(dxn, out, N, cpad)
        st['dx'] = (out[:, :cin] if cpad != cin else out).to(st['x_dtype'])

    def _run_backward(self, st, dlogits, want_dx=False):
        if st.get('f32'):
            return self._run_backward_f32(st, dlogits, want_dx)
        d = self._dims
        plan, pk, N = st['plan'], st['pk'], st['N']
        dev = dlogits.device
        ws, lib = st['ws'], _lib.load()
        self._mark('bwd_begin')
        torch.cuda.current_stream(dev).wait_event(st['blob_ready'])     # backward half of the tile lists
        c, p_, cpad, nv = d['c'], d['p'], d['cin_pad'], len(self._tokens)
        np128, np64 = _rup(N, 128), _rup(N, 64)
        arena, blob, acts = st['arena'], st['blob'], st['acts']
        slot = (HW * c, c)
        fcs = list(d['fcs']) + [d['na']]
        dims_in = [HW * p_] + list(d['fcs'])
        f32 = dict(dtype=torch.float32, device=dev)
        grads = {}
        emitted = set()

        # ---- linear layers, last to first
        dl = dlogits.detach().to(torch.float32).contiguous()
        li_first = len(fcs) - 1
        if self._head_fused(fcs, dims_in):
            # the last layer, and the bias gradient of the layer in front of it, in two launches
            li = li_first
            fo, fi = fcs[li], dims_in[li]
            name = 'classifier.%d' % self._fc_idx[li]
            gw, gb, gb_in = torch.empty(fo, fi, **f32), torch.empty(fo, **f32), torch.empty(fi, **f32)
            gin = ws.get('fc_gin%d' % li, np128 * fi, torch.bfloat16, dev)
            if np128 != N:
                gin.view(np128, fi)[N:].zero_()   # rows past N are the K padding of the next GEMM
            wk = ws.get('head_ws', ops.head_backward_workspace(N, fi, fo), torch.float32, dev)
            ops.head_backward(dl, acts[li], fi, pk['fc_w'][li], fi, gin, fi, gw, gb_in, gb, wk, N, fi, fo)
            if self.grad_ready_hook is not None:
                self.grad_ready_hook(name + '.weight', gw)
                emitted.add(name + '.weight')
            grads[name + '.weight'], grads[name + '.bias'] = gw, gb
            grads['classifier.%d.bias' % self._fc_idx[li - 1]] = gb_in
            g, g_ld = gin, fi
            li_first = li - 1
        else:
            g = ws.get('dlog', np128 * d['na_pad'], torch.bfloat16, dev)
            ops.cast_pad(dl, g, N, d['na'], np128, d['na_pad'])
            gb = torch.zeros(d['na'], **f32)
            ops.col_sum(dl, gb, N, d['na'], d['na'])
            grads['classifier.%d.bias' % self._fc_idx[-1]] = gb
            g_ld = d['na_pad']
        for li in range(li_first, -1, -1):
            fo, fi = fcs[li], dims_in[li]
            name = 'classifier.%d' % self._fc_idx[li]
            # weight gradient: dW[fo][fi] = g^T (rows = samples) x input activations
            gw = torch.empty(fo, fi, **f32)
            if li > 0:
                ops.gemm(g, acts[li], gw, fo, fi, np64, lda=g_ld, ldb=fi, ldc=fi, a_kmajor=True, b_kmajor=True)
                if self.grad_ready_hook is not None:
                    self.grad_ready_hook(name + '.weight', gw)
                    emitted.add(name + '.weight')
            else:
                # The weight gradient of the first linear layer (most of the network's parameters)
                # feeds nothing else in this pass: it runs on the side stream, next to the small
                # classifier and top-of-tree launches that cannot fill the device on their own.  Under
                # data parallelism its all-reduce is issued from there as well, first of all.
                # The GEMM reads a copy of the activations in the reference's flatten order
                # (Layers.py:42-44), so the 411 MB gradient is written once, in its final layout.
                act_t = ws.get('fc0_act_t', np128 * fi, torch.bfloat16, dev)

                def fc1_weight_grad(g=g, g_ld=g_ld, gw=gw, li=li, name=name):
                    ops.transpose_hw_c(acts[li], act_t, np128, p_)
                    ops.gemm(g, act_t, gw, fo, fi, np64, lda=g_ld, ldb=fi, ldc=fi, a_kmajor=True, b_kmajor=True)
                    if self.grad_ready_hook is not None:
                        self.grad_ready_hook(name + '.weight', gw)
                        emitted.add(name + '.weight')
            grads[name + '.weight'] = gw
            # input gradient, masked by the ReLU that produced the input
            gin = ws.get('fc_gin%d' % li, np128 * fi, torch.bfloat16, dev)
            if np128 != N:
                gin.view(np128, fi)[N:].zero_()   # rows past N are the K padding of the next GEMM
            kdim = g_ld if li == len(fcs) - 1 else fo
            ops.gemm(g, pk['fc_w'][li], gin, N, fi, kdim, lda=g_ld, ldb=fi, ldc=fi, b_kmajor=True,
                     out_bf16=True, mask=acts[li])
            g, g_ld = gin, fi
            if li > 0:
                gb = torch.zeros(fi, **f32)
                ops.col_sum(gin, gb, N, fi, fi)
                grads['classifier.%d.bias' % self._fc_idx[li - 1]] = gb
        dc1 = g   # bf16 [N][196][P], already times (C1 > 0)
        self._mark('fc_bwd')

        # ---- classifier conv1x1: weight/bias gradient, and the gradient into the roots' dZ slots
        it_off, it_cnt = plan.section('SAMPLE_ITEMS')
        sec = lambda n: (blob, plan.section(n)[0])
        gcb = torch.empty(p_, **f32)       # (written whole by the reduction)
        info = plan.info
        # partial blocks of the largest launch, with one row of bias sums per item behind them: every
        # gradient is summed in item order, the same bits on every run
        need_blocks = max(info.wgrad_partial_blocks, sum(ops.wgrad_blocks(it_cnt, c, cpad // 128, 9)),
                          sum(ops.wgrad_blocks(plan.section('SAMPLE_ITEMS_FINE')[1], c, c // 128, 9)))
        part = ws.get('wg_partial', need_blocks * 8192, torch.float32, dev)
        # (a buffer of its own: this one launch runs beside the tree's, on the side stream)
        part_cls = ws.get('wg_partial_cls', sum(ops.wgrad_blocks(it_cnt, p_, c // 128, 1)) * 8192, torch.float32, dev)
        gw_cls = torch.empty(p_, c, 1, 1, **f32)
        grads['classifier.0.weight'], grads['classifier.0.bias'] = gw_cls, gcb

        def cls_weight_grad():
            ops.wgrad_tiles(arena, dc1, (blob, it_off), it_cnt, sec('SLOTS_ROOTY'), sec('SLOTS_IDENT'), None, gcb,
                            taps=1, k_chunks=c // 128, n_out=p_, src_slot_stride=HW * c, dy_slot_stride=HW * p_,
                            src_pix_stride=c, dy_pix_stride=p_, dw_group_stride=c * p_, total_nodes=N, partial=part_cls,
                            bias_rows=True)
            ops.wgrad_reduce(part_cls, (blob, it_off), it_cnt, 1, p_, c // 128, 1, gw_cls, 0, db=gcb)

        # Two weight gradients feed nothing else in this pass: the first linear layer's (most of the
        # network's parameters; two device-filling GEMMs side by side only slow each other down, so
        # it starts behind the input-gradient GEMM) and the classifier convolution's (bound by HBM,
        # not by the matrix cores).  What the tree needs is the gradient into the roots.
        order = self.backward_order
        cur, side = torch.cuda.current_stream(dev), self._side
        fork = torch.cuda.Event()
        fork.record(cur)
        with torch.cuda.stream(side):
            side.wait_event(fork)
            if order in ('A', 'B'):
                fc1_weight_grad()
            if order in ('B', 'C'):
                cls_weight_grad()
            side_done = torch.cuda.Event()
            side_done.record(side)
        if order == 'C':
            fc1_weight_grad()
        if order == 'A':
            cls_weight_grad()
        self._conv(st, dc1, arena, pk['wd_cls'], 'CLS_DGRAD_TILES', 1, p_ // 128, c, (HW * p_, p_), slot,
                   mask=arena)

        # ---- module trees: input gradients by level, scene fan-in, all weight gradients
        # (the two 3x3 gradient arrays are the halves of one array: vqa_tree_backward then runs both weight
        # gradients as one launch over a merged item list)
        gw12, gb12 = torch.empty(2 * nv, c, c, 3, 3, **f32), torch.empty(2 * nv, c, **f32)
        gw1, gw2 = gw12[:nv], gw12[nv:]
        gwp = torch.empty(nv, c, 2 * c, 1, 1, **f32)
        # (all six arrays are written whole by the reductions, rows of unused tokens as zeros)
        acc = {'g_conv1': gw1, 'g_conv2': gw2, 'g_proj': gwp, 'gb_conv1': gb12[:nv],
               'gb_conv2': gb12[nv:], 'gb_proj': torch.empty(nv, c, **f32)}
        tctx = self._tree_ctx(st, pk, acc)
        tctx.wgrad_partial = part.data_ptr()
        self._mark('cls_bwd')
        _lib.check(lib.vqa_tree_backward(plan.handle, C.byref(tctx), ops._stream()))
        self._mark('tree_bwd')
        if self.grad_group_hook is not None:
            # the per-token gradients are rows of these arrays (zeros for tokens this batch did not use)
            # (the same name -> row dict objects every step: a receiver may key its own tables on them)
            if self._group_rows is None:
                self._group_rows = {key: {tok + key: i for i, tok in enumerate(self._tokens) if self._binary[i] or not binary_only}
                                    for key, binary_only in (('.conv1.weight', False), ('.conv2.weight', False), ('.proj.weight', True),
                                                             ('.conv1.bias', False), ('.conv2.bias', False), ('.proj.bias', True))}
            for arr, key in ((gw1, '.conv1.weight'), (gw2, '.conv2.weight'), (gwp, '.proj.weight'),
                             (acc['gb_conv1'], '.conv1.bias'), (acc['gb_conv2'], '.conv2.bias'), (acc['gb_proj'], '.proj.bias')):
                rows = self._group_rows[key]
                self.grad_group_hook(arr, rows)
                emitted.update(rows)
        counts = plan.token_counts()
        for i, tok in enumerate(self._tokens):
            if counts[i] == 0:
                continue      # unused in this batch: .grad stays None, as with the reference
            if self._binary[i]:
                grads[tok + '.proj.weight'], grads[tok + '.proj.bias'] = gwp[i], acc['gb_proj'][i]
            grads[tok + '.conv1.weight'], grads[tok + '.conv1.bias'] = gw1[i], acc['gb_conv1'][i]
            grads[tok + '.conv2.weight'], grads[tok + '.conv2.bias'] = gw2[i], acc['gb_conv2'][i]

        # ---- stem: dS0 = dgrad(dF) * (S0 > 0); weight gradients of both convs (no d/dx)
        self._conv(st, arena, arena, pk['wd_stem1'], 'STEM1_DGRAD_TILES', 9, c // 128, c, slot, slot, mask=arena)
        if want_dx:
            self._input_grad(st, False)
        gs1b, gs0b = torch.empty(c, **f32), torch.empty(c, **f32)       # (written whole by the reductions)
        fi_off, fi_cnt = plan.section('SAMPLE_ITEMS_FINE')
        ops.wgrad_tiles(arena, arena, (blob, fi_off), fi_cnt, sec('SLOTS_S0'), sec('SLOTS_DF'), None, gs1b,
                        taps=9, k_chunks=c // 128, n_out=c, src_slot_stride=HW * c, dy_slot_stride=HW * c,
                        src_pix_stride=c, dy_pix_stride=c, dw_group_stride=9 * c * c, total_nodes=N, partial=part,
                        bias_rows=True)
        gw = torch.empty(c, c, 3, 3, **f32)
        ops.wgrad_reduce(part, (blob, fi_off), fi_cnt, 1, c, c // 128, 9, gw, 0, db=gs1b)
        grads['stem.2.weight'], grads['stem.2.bias'] = gw, gs1b
        x_slots = (st['x_rows'], 0) if st.get('x_rows') is not None else sec('SLOTS_IDENT')
        ops.wgrad_tiles(st['x_nhwc'], arena, (blob, it_off), it_cnt, x_slots, sec('SLOTS_DS0'), None, gs0b,
                        taps=9, k_chunks=cpad // 128, n_out=c, src_slot_stride=HW * cpad, dy_slot_stride=HW * c,
                        src_pix_stride=cpad, dy_pix_stride=c, dw_group_stride=9 * cpad * c, total_nodes=N, partial=part,
                        bias_rows=True)
        gw = torch.empty(c, cpad, 3, 3, **f32)
        ops.wgrad_reduce(part, (blob, it_off), it_cnt, 1, c, cpad // 128, 9, gw, 0, db=gs0b)
        grads['stem.0.weight'] = gw[:, :d['cin']].contiguous() if cpad != d['cin'] else gw
        grads['stem.0.bias'] = gs0b

        out = []
        for name, p in self._param_groups():
            gr = grads.get(name)
            if gr is not None:
                gr = gr.reshape(p.shape)
                if self.grad_ready_hook is not None and name not in emitted:
                    self.grad_ready_hook(name, gr)
            out.append(gr if p.requires_grad else None)
        torch.cuda.current_stream(dev).wait_event(side_done)
        self._mark('bwd_end')
        self._params_changed()
        if self.backward_done_hook is not None:
            self.backward_done_hook()
        if not self.keep_state:
            st.release()
        return out
