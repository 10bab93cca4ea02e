"""Time the 3x3 conv kernel on one large launch with parts of it switched off (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, numpy as np, torch
from vqa_amd import ops, _lib
HW, Cc = 196, 128
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2372
G = 30
dev = torch.device('cuda')
arena = (torch.randn(3 * T, HW, Cc, device=dev) * 0.5).to(torch.bfloat16)
w = torch.randn(G, Cc, Cc, 3, 3, device=dev) * 0.05
wp = ops.pack_conv_weight(w, G, Cc, Cc, 9, Cc * Cc * 9, Cc * 9, 9)
bias = torch.randn(G, Cc, device=dev)
groups = np.sort(np.random.default_rng(0).integers(0, G, T))
perm = np.random.default_rng(1).permutation(T)
def run(flags_extra, res, mask, relu, name):
    tiles = torch.tensor([[int(perm[i]), -1, T + i, -1, (2 * T + int(perm[i])) if res else -1, (2 * T + i) if mask else -1, -1, int(groups[i])] for i in range(T)], dtype=torch.int32, device=dev).reshape(-1)
    p = _lib.ConvParams()
    p.src = p.dst = p.res = p.mask = arena.data_ptr(); p.w = wp.data_ptr(); p.bias = bias.data_ptr(); p.tiles = tiles.data_ptr()
    p.src_slot_stride = p.dst_slot_stride = HW * Cc; p.w_group_stride = 9 * Cc * Cc
    p.src_pix_stride = p.dst_pix_stride = Cc; p.k_chunks = 1; p.n_out = Cc
    p.flags = (1 if relu else 0) | (2 if relu else 0) | flags_extra; p.num_tiles = T
    lib = _lib.load(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3): lib.vqa_conv_tiles(C.byref(p), 9, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): lib.vqa_conv_tiles(C.byref(p), 9, st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print('%-44s %8.1f us  %6.0f TF  %.2f us/tile/CU' % (name, us, T * 57.8e6 / us / 1e6, us / (T / 256)))
run(0, False, False, True, 'fwd conv1 (bias relu)')
run(0, True, False, True, 'fwd conv2 (bias res relu)')
run(0, True, True, False, 'dgrad1 (res mask)')
run(0x100, False, False, True, 'no epilogue')
run(0x200, False, False, True, 'no weight streaming (re-read step 0)')
run(0x400, False, False, True, 'no source loads')
run(0x100 | 0x400, False, False, True, 'no epilogue, no source loads')
run(0x100 | 0x200 | 0x400, False, False, True, 'only LDS + MFMA')
