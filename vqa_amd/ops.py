"""Thin torch-tensor front ends of the C-ABI device entry points.

torch is plumbing here: it owns device memory and the current HIP stream; every computation
below is a hand-written gfx950 kernel inside libvqa_amd.so.  Nothing falls back to torch ops.
"""
import ctypes as C

import torch

from . import _lib

HW = 196


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError('vqa_amd device op called with a CPU tensor: the HIP path has no '
                               'CPU fallback')


def conv_tiles(src, dst, w_packed, tiles, num_tiles, *, taps, k_chunks, n_out, src_slot_stride,
               dst_slot_stride, src_pix_stride, dst_pix_stride, w_group_stride, bias=None,
               res=None, mask=None, pair_chunks=0, pair_out=0, relu=False, tiles_offset=0, f32=False):
    """tiles: int32 device tensor holding vqa_conv_tile records from word `tiles_offset`.
    f32: fp32 maps and plain fp32 [group][tap][k][n] weights (vqa_conv_tiles_f32)."""
    _need_cuda(src, dst, w_packed, tiles)
    p = _lib.ConvParams()
    p.src, p.dst = src.data_ptr(), dst.data_ptr()
    p.res = res.data_ptr() if res is not None else dst.data_ptr()
    p.mask = mask.data_ptr() if mask is not None else dst.data_ptr()
    p.w = w_packed.data_ptr()
    p.bias = bias.data_ptr() if bias is not None else 0
    p.tiles = tiles.data_ptr() + 4 * tiles_offset
    p.src_slot_stride, p.dst_slot_stride, p.w_group_stride = src_slot_stride, dst_slot_stride, w_group_stride
    p.src_pix_stride, p.dst_pix_stride = src_pix_stride, dst_pix_stride
    p.k_chunks, p.n_out, p.pair_chunks, p.pair_out = k_chunks, n_out, pair_chunks, pair_out
    p.flags = (_lib.CONV_BIAS if bias is not None else 0) | (_lib.CONV_RELU if relu else 0)
    p.num_tiles = num_tiles
    fn = _lib.load().vqa_conv_tiles_f32 if f32 else _lib.load().vqa_conv_tiles
    _lib.check(fn(C.byref(p), taps, _stream()))


def resblock_tiles(arena, w_first, w_second, tiles, num_tiles, *, bias_first=None, bias_second=None,
                   relu=False, tiles_offset=0, w_pre=None, bias_pre=None, num_pre=0,
                   phase_stamps=None, level_info=None, level=0):
    """Every convolution of a 128-wide tree node, one workgroup per tile (vqa_resblock_tiles);
    `arena` is bf16 [slots][196][128], `tiles` holds 12-word vqa_resblock_tile records.  With
    `level_info` (device int32 [levels][2] of vqa_devsched_build) the tile range of `level` is read
    on the device and shared by the num_tiles workgroups of the launch."""
    _need_cuda(arena, w_first, w_second, tiles)
    p = _lib.ResblockParams()
    p.arena, p.w_first, p.w_second = arena.data_ptr(), w_first.data_ptr(), w_second.data_ptr()
    p.w_pre = w_pre.data_ptr() if w_pre is not None else 0
    p.bias_first = bias_first.data_ptr() if bias_first is not None else 0
    p.bias_second = bias_second.data_ptr() if bias_second is not None else 0
    p.bias_pre = bias_pre.data_ptr() if bias_pre is not None else 0
    p.tiles = tiles.data_ptr() + 4 * tiles_offset
    p.slot_stride, p.w_group_stride, p.w_pre_group_stride = 196 * 128, 9 * 128 * 128, 2 * 128 * 128
    p.flags = (_lib.CONV_BIAS if bias_first is not None else 0) | (_lib.CONV_RELU if relu else 0)
    p.num_tiles, p.num_pre = num_tiles, num_pre
    p.phase_stamps = phase_stamps.data_ptr() if phase_stamps is not None else 0
    p.level_info = level_info.data_ptr() + 8 * level if level_info is not None else 0
    _lib.check(_lib.load().vqa_resblock_tiles(C.byref(p), _stream()))


class DeviceSchedule:
    """Buffers of one vqa_devsched_build call, carved out of one int32 device tensor: `tiles`
    ([N * T][12] fused tile records, level-major), `level_info` ([T + 1][2]), `cls_tiles` ([N][8]),
    `hist` ([T + 1][V] nodes per (level, token)), `error` ([1])."""

    @staticmethod
    def words(N, T, V):
        return N * (2 * T + 1) * 4 + N * 4 + 3 * (T + 1) * V + N * 2 + (T + 1) * 2 + N * T * 12 + N * 8 + 4 + 4 * 10

    def __init__(self, buf, N, T, V):
        assert buf.dtype == torch.int32 and buf.numel() >= self.words(N, T, V)
        at = [0]

        def take(n):
            v = buf[at[0]:at[0] + n]
            at[0] += (n + 3) // 4 * 4          # 16-byte alignment throughout
            return v
        self.N, self.T, self.V = N, T, V
        self.tiles, self.cls_tiles = take(N * T * 12), take(N * 8)
        self.level_info, self.hist = take((T + 1) * 2), take((T + 1) * V)
        self.error = take(2)          # the flag; the deepest root's level
        self.node, self.counts = take(N * (2 * T + 1) * 4), take(N * 4)
        self.row_base, self.bin_off, self.bin_cursor = take(N * 2), take((T + 1) * V), take((T + 1) * V)


def devsched_build(programs, arity, used, sched, null_id, start_id, scene_id, spread=False, image_of_row=None,
                   num_images=0):
    """Programs in device memory -> tile lists in device memory (vqa_devsched_build): `programs`
    int64 [N][T], `arity` int8 [V], `used` float32 [N][T] out, `sched` a DeviceSchedule; `image_of_row`
    int32 [N] when rows share images (num_images of them)."""
    _need_cuda(programs, arity, used, sched.tiles)
    N, T = programs.shape
    assert programs.dtype == torch.int64 and programs.is_contiguous() and arity.dtype == torch.int8
    assert used.dtype == torch.float32 and used.is_contiguous() and used.shape == (N, T)
    assert (N, T, arity.numel()) == (sched.N, sched.T, sched.V)
    p = _lib.DevschedParams()
    p.programs, p.arity, p.used = programs.data_ptr(), arity.data_ptr(), used.data_ptr()
    for k in ('node', 'counts', 'hist', 'row_base', 'level_info', 'bin_off', 'bin_cursor', 'tiles',
              'cls_tiles', 'error'):
        setattr(p, k, getattr(sched, k).data_ptr())
    p.N, p.T, p.V, p.null_id, p.start_id, p.scene_id = N, T, sched.V, null_id, start_id, scene_id
    p.spread = int(spread)
    if image_of_row is not None:
        _need_cuda(image_of_row)
        assert image_of_row.dtype == torch.int32 and image_of_row.is_contiguous() and image_of_row.numel() == N
        p.image_of_row, p.num_images = image_of_row.data_ptr(), num_images
    _lib.check(_lib.load().vqa_devsched_build(C.byref(p), _stream()))


def pack_conv_weight(w, groups, n_out, k_in, taps, g_stride, n_stride, k_stride, flip=False, out=None):
    """fp32 weights -> bf16 slabs [group][chunk][tap][half][n][64] (swizzled LDS image)."""
    _need_cuda(w)
    kpad = (k_in + 127) // 128 * 128
    if out is None:
        out = torch.empty(groups * kpad * taps * n_out, dtype=torch.bfloat16, device=w.device)
    assert out.numel() == groups * kpad * taps * n_out and w.dtype == torch.float32
    _lib.check(_lib.load().vqa_pack_conv_weight(_ptr(w), _ptr(out), groups, n_out, k_in, taps,
                                                g_stride, n_stride, k_stride, int(flip), _stream()))
    return out


def _wgrad_params(src, dy, items, num_items, src_slots, dy_slots, dw, db, *, k_chunks, n_out,
                  src_slot_stride, dy_slot_stride, src_pix_stride, dy_pix_stride, dw_group_stride,
                  src1_slots=None, pair_chunks=0, total_nodes=0):
    p = _lib.WgradParams()
    p.src, p.dy = src.data_ptr(), dy.data_ptr()
    p.items = items[0].data_ptr() + 4 * items[1]
    p.src_slots = src_slots[0].data_ptr() + 4 * src_slots[1]
    p.src1_slots = (src1_slots[0].data_ptr() + 4 * src1_slots[1]) if src1_slots is not None else 0
    p.dy_slots = dy_slots[0].data_ptr() + 4 * dy_slots[1]
    p.dw = dw.data_ptr() if dw is not None else 0
    p.db = db.data_ptr() if db is not None else 0
    p.src_slot_stride, p.dy_slot_stride, p.dw_group_stride = src_slot_stride, dy_slot_stride, dw_group_stride
    p.src_pix_stride, p.dy_pix_stride = src_pix_stride, dy_pix_stride
    p.k_chunks, p.pair_chunks, p.n_out, p.num_items = k_chunks, pair_chunks, n_out, num_items
    p.total_nodes = total_nodes
    return p


def wgrad_blocks(num_items, n_out, k_chunks, taps):
    """Blocks of 8192 floats of a partial-block weight-gradient launch: (weight-gradient blocks, blocks of
    the per-item bias rows that follow them)."""
    return num_items * (n_out // 64) * k_chunks * taps, (num_items * n_out + 8191) // 8192


def wgrad_tiles(src, dy, items, num_items, src_slots, dy_slots, dw, db, *, taps, partial=None, f32=False,
                bias_rows=False, **kw):
    """items / *_slots are (tensor, word_offset) pairs into an int32 device blob.  With `partial`
    (fp32 workspace of num_items * n_out/64 * k_chunks * taps * 8192 elements) the workgroups store
    partial blocks for wgrad_reduce and `dw` may be None; otherwise they add into `dw` atomically.
    The bias sums are added to `db` atomically, or, with `bias_rows` (and `partial`, sized by
    wgrad_blocks), stored per item behind the partial blocks for wgrad_reduce(..., db=) to add in
    item order."""
    p = _wgrad_params(src, dy, items, num_items, src_slots, dy_slots, dw, None if bias_rows else db, **kw)
    if partial is not None:
        dwb, bb = wgrad_blocks(num_items, kw['n_out'], kw['k_chunks'], taps)
        assert partial.numel() >= (dwb + (bb if bias_rows else 0)) * 8192
        p.partial = partial.data_ptr()
        if bias_rows:
            p.bias_block = dwb
    else:
        assert not bias_rows
    fn = _lib.load().vqa_wgrad_tiles_f32 if f32 else _lib.load().vqa_wgrad_tiles
    _lib.check(fn(C.byref(p), taps, _stream()))


def wgrad_tiles_noalloc(src, dy, items, num_items, src_slots, dy_slots, db, out, *, taps, groups, out_g_stride, **kw):
    """wgrad_tiles + wgrad_reduce with the partial blocks in the library's own static device array:
    allocates nothing (vqa_wgrad_tiles_noalloc).  `out` is fp32 [groups][n_out][k][taps]."""
    p = _wgrad_params(src, dy, items, num_items, src_slots, dy_slots, None, db, **kw)
    _lib.check(_lib.load().vqa_wgrad_tiles_noalloc(C.byref(p), taps, groups, _ptr(out), out_g_stride, _stream()))


def wgrad_reduce(partial, items, num_items, groups, n_out, k_chunks, taps, out, out_g_stride, db=None):
    """out[group][n][k][tap] = sum of the group's partial blocks (see wgrad_tiles); db[group][n] =
    sum of the group's bias rows, when the launch stored them (bias_rows)."""
    bias = 0
    if db is not None:
        assert db.dtype == torch.float32 and db.numel() >= groups * n_out
        bias = partial.data_ptr() + 4 * 8192 * wgrad_blocks(num_items, n_out, k_chunks, taps)[0]
    _lib.check(_lib.load().vqa_wgrad_reduce(_ptr(partial), items[0].data_ptr() + 4 * items[1], num_items, groups,
                                            n_out, k_chunks, taps, _ptr(out), out_g_stride, bias, _ptr(db), _stream()))


def unpack_conv_grad(dw, out, groups, n_out, k_in, taps, out_g_stride):
    _lib.check(_lib.load().vqa_unpack_conv_grad(_ptr(dw), _ptr(out), groups, n_out, k_in, taps,
                                                out_g_stride, _stream()))


def gemm(a, b, c, M, N, K, *, lda, ldb, ldc, a_kmajor=False, b_kmajor=False, out_bf16=False,
         split_k=1, mask=None):
    _need_cuda(a, b, c)
    p = _lib.GemmParams()
    p.a, p.b, p.c = a.data_ptr(), b.data_ptr(), c.data_ptr()
    p.mask = mask.data_ptr() if mask is not None else 0
    p.lda, p.ldb, p.ldc, p.M, p.N, p.K = lda, ldb, ldc, M, N, K
    p.a_kmajor, p.b_kmajor = int(a_kmajor), int(b_kmajor)
    p.out_mode, p.split_k = (2 if out_bf16 else 1), split_k
    _lib.check(_lib.load().vqa_gemm(C.byref(p), _stream()))


def gemm_f32(a, b, c, M, N, K, *, lda, ldb, ldc, a_kmajor=False, b_kmajor=False, split_k=1, mask=None):
    """fp32 operands and result on fp32 MFMA (vqa_gemm_f32); layouts as gemm()."""
    _need_cuda(a, b, c)
    assert a.dtype == b.dtype == c.dtype == torch.float32 and (mask is None or mask.dtype == torch.float32)
    p = _lib.GemmParams()
    p.a, p.b, p.c = a.data_ptr(), b.data_ptr(), c.data_ptr()
    p.mask = mask.data_ptr() if mask is not None else 0
    p.lda, p.ldb, p.ldc, p.M, p.N, p.K = lda, ldb, ldc, M, N, K
    p.a_kmajor, p.b_kmajor, p.out_mode, p.split_k = int(a_kmajor), int(b_kmajor), 1, split_k
    _lib.check(_lib.load().vqa_gemm_f32(C.byref(p), _stream()))


def transpose_f32(x, out, n, rows, cols, ld_out=None, out_stride=None):
    """out[i][c][r] = x[i][r][c], fp32; rows of out ld_out apart (default rows), matrices out_stride apart."""
    _need_cuda(x, out)
    assert x.dtype == out.dtype == torch.float32
    ld_out = rows if ld_out is None else ld_out
    out_stride = cols * ld_out if out_stride is None else out_stride
    _lib.check(_lib.load().vqa_transpose_f32(_ptr(x), _ptr(out), n, rows, cols, ld_out, out_stride, _stream()))


def nchw_to_nhwc(x, out, n, cin, cpad):
    _need_cuda(x, out)
    _lib.check(_lib.load().vqa_nchw_to_nhwc(_ptr(x), _ptr(out), n, cin, cpad, _stream()))


def nhwc_to_nchw(x, out, n, c):
    _lib.check(_lib.load().vqa_nhwc_to_nchw(_ptr(x), _ptr(out), n, c, _stream()))


def cast_pad(w, out, rows, cols, rows_pad, cols_pad):
    _lib.check(_lib.load().vqa_cast_pad(_ptr(w), _ptr(out), rows, cols, rows_pad, cols_pad, _stream()))


def fc_weight_to_nhwc(w, out, f, fpad, p):
    _lib.check(_lib.load().vqa_fc_weight_to_nhwc(_ptr(w), _ptr(out), f, fpad, p, _stream()))


def fc_grad_from_nhwc(g, out, f, p):
    _lib.check(_lib.load().vqa_fc_grad_from_nhwc(_ptr(g), _ptr(out), f, p, _stream()))


def transpose_hw_c(x, out, n, c):
    """bf16 [n][196][c] -> bf16 [n][c][196]"""
    assert x.dtype == out.dtype == torch.bfloat16 and x.numel() >= n * 196 * c and out.numel() >= n * 196 * c
    _lib.check(_lib.load().vqa_transpose_hw_c(_ptr(x), _ptr(out), n, c, _stream()))


def bias_act(acc, bias, out, rows, cols, ld_acc, ld_out, relu, slabs=1, slab_stride=0):
    _lib.check(_lib.load().vqa_bias_act(_ptr(acc), _ptr(bias), _ptr(out), rows, cols, ld_acc, ld_out,
                                        int(relu), slabs, slab_stride,
                                        int(out.dtype == torch.float32), _stream()))


def col_sum(x, out, rows, cols, ld):
    _lib.check(_lib.load().vqa_col_sum(_ptr(x), _ptr(out), rows, cols, ld,
                                       int(x.dtype == torch.bfloat16), _stream()))


def stem_nchw(x, x_nhwc, dst, w_packed, bias, n, cin, dst_slot_stride, dst_slot_first, dst_pix_stride, relu=True):
    """The stem's first convolution (3x3, cin -> 128) from the fp32 NCHW batch x [n][cin][14][14] into slots
    dst_slot_first.. of dst; x_nhwc (or None) receives the bf16 channels-last copy [n][196][cin]"""
    _need_cuda(x, dst, w_packed)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.numel() >= n * cin * 196
    assert dst.dtype == torch.bfloat16 and w_packed.dtype == torch.bfloat16 and w_packed.numel() >= 9 * cin * 128
    assert dst.numel() >= (dst_slot_first + n - 1) * dst_slot_stride + 196 * dst_pix_stride
    assert x_nhwc is None or (x_nhwc.dtype == torch.bfloat16 and x_nhwc.numel() >= n * 196 * cin)
    assert bias is None or (bias.dtype == torch.float32 and bias.numel() >= 128)
    _lib.check(_lib.load().vqa_stem_nchw(_ptr(x), _ptr(x_nhwc), _ptr(dst), _ptr(w_packed), _ptr(bias), dst_slot_stride,
                                         dst_slot_first, dst_pix_stride, n, cin, int(relu), _stream()))


def head_forward(act, ld_act, w, ld_w, bias, logits, n, f, a):
    """logits fp32 [n][a] = act bf16 [n][ld_act] x w bf16 [a][ld_w]^T + bias: the classifier's last linear
    layer (Module_Model.py:92) in one launch"""
    assert act.dtype == w.dtype == torch.bfloat16 and logits.dtype == torch.float32 and logits.is_contiguous()
    assert act.numel() >= (n - 1) * ld_act + f and w.numel() >= (a - 1) * ld_w + f and logits.numel() >= n * a
    _lib.check(_lib.load().vqa_head_forward(_ptr(act), ld_act, _ptr(w), ld_w, _ptr(bias), _ptr(logits), n, f, a, _stream()))


def head_backward_workspace(n, f, a):
    return int(_lib.load().vqa_head_backward_workspace(n, f, a))


def head_backward(dlogits, act, ld_act, w, ld_w, gin, ld_gin, dw, db_in, db, workspace, n, f, a):
    """Backward of head_forward: gin bf16 [n][ld_gin] (times act > 0), dw fp32 [a][f], db fp32 [a], and
    db_in fp32 [f] = column sums of gin; two launches"""
    assert dlogits.dtype == dw.dtype == db_in.dtype == db.dtype == workspace.dtype == torch.float32
    assert act.dtype == w.dtype == gin.dtype == torch.bfloat16 and dlogits.is_contiguous()
    assert dlogits.numel() >= n * a and act.numel() >= (n - 1) * ld_act + f and w.numel() >= (a - 1) * ld_w + f
    assert gin.numel() >= (n - 1) * ld_gin + f and dw.numel() >= a * f and db_in.numel() >= f and db.numel() >= a
    assert workspace.numel() >= head_backward_workspace(n, f, a)
    _lib.check(_lib.load().vqa_head_backward(_ptr(dlogits), _ptr(act), ld_act, _ptr(w), ld_w, _ptr(gin), ld_gin, _ptr(dw),
                                             _ptr(db_in), _ptr(db), _ptr(workspace), n, f, a, _stream()))


def xent(logits, target, loss, dlogits):
    """Mean cross-entropy of fp32 logits [n][a] against int64 targets, and its gradient (two small launches)"""
    n, a = logits.shape
    assert logits.dtype == dlogits.dtype == loss.dtype == torch.float32 and target.dtype == torch.int64
    assert logits.is_contiguous() and dlogits.is_contiguous() and target.is_contiguous()
    assert target.numel() == n and dlogits.numel() == n * a and loss.numel() >= 1
    ws = torch.empty(128, dtype=torch.float32, device=logits.device)
    _lib.check(_lib.load().vqa_xent(_ptr(logits), _ptr(target), _ptr(loss), _ptr(dlogits), _ptr(ws), n, a, _stream()))


def slot_sum_mask(arena, blob, off_offsets, off_src, off_dst, off_mask, n_dst, slot_elems):
    b = blob.data_ptr()
    _lib.check(_lib.load().vqa_slot_sum_mask(_ptr(arena), C.c_void_p(b + 4 * off_offsets),
                                             C.c_void_p(b + 4 * off_src), C.c_void_p(b + 4 * off_dst),
                                             C.c_void_p(b + 4 * off_mask), n_dst, slot_elems, _stream()))


KERNEL_KINDS = ('conv3x3', 'conv1x1', 'wgrad3x3', 'wgrad1x1', 'gemm', 'resblock')


def prof_enable(kinds):
    """Time launches of the given kernel kinds with HIP events: True = all, False = none, or an
    iterable of names from KERNEL_KINDS."""
    if kinds is True:
        mask = (1 << len(KERNEL_KINDS)) - 1
    elif not kinds:
        mask = 0
    else:
        mask = sum(1 << KERNEL_KINDS.index(k) for k in kinds)
    _lib.check(_lib.load().vqa_prof_enable(mask))


def prof_collect():
    """{kind: dict(launches, ms, flops, bytes)} of the launches since the last collect."""
    out = (C.c_double * (4 * len(KERNEL_KINDS)))()
    _lib.check(_lib.load().vqa_prof_collect(out, len(KERNEL_KINDS)))
    return {k: dict(launches=int(out[4 * i]), ms=out[4 * i + 1], flops=out[4 * i + 2], bytes=out[4 * i + 3])
            for i, k in enumerate(KERNEL_KINDS)}
