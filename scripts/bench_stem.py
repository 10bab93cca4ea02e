"""Times vqa_stem_nchw alone (1024 samples, 1024 -> 128 channels) against the layout pass + vqa_conv_tiles, and with
parts of the producer waves' work switched off (diagnostic bits of the relu argument)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd import ops, _lib
import ctypes as C
d = torch.device('cuda:0')
N, Cin, Cc, HW = 1024, 1024, 128, 196
x = torch.randn(N, Cin, 14, 14, device=d).clamp_(min=0)
w = torch.randn(1, Cc, Cin, 3, 3, device=d) * 0.02
b = torch.randn(Cc, device=d) * 0.1
wp = ops.pack_conv_weight(w, 1, Cc, Cin, 9, 0, Cin * 9, 9)
dst = torch.zeros(2 * N, HW, Cc, dtype=torch.bfloat16, device=d)
xn = torch.empty(N * HW * Cin, dtype=torch.bfloat16, device=d)
tiles = torch.tensor([[i, -1, N + i, -1, -1, -1, -1, 0] for i in range(N)], dtype=torch.int32).reshape(-1).to(d)
def timeit(fn, name):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    print('%-60s %.1f us' % (name, e0.elapsed_time(e1) * 100))
def old():
    ops.nchw_to_nhwc(x, xn, N, Cin, Cin)
    ops.conv_tiles(xn, dst, wp, tiles, N, taps=9, k_chunks=Cin // 128, n_out=Cc, src_slot_stride=HW * Cin, dst_slot_stride=HW * Cc,
                   src_pix_stride=Cin, dst_pix_stride=Cc, w_group_stride=0, bias=b, relu=True)
def conv_only():
    ops.conv_tiles(xn, dst, wp, tiles, N, taps=9, k_chunks=Cin // 128, n_out=Cc, src_slot_stride=HW * Cin, dst_slot_stride=HW * Cc,
                   src_pix_stride=Cin, dst_pix_stride=Cc, w_group_stride=0, bias=b, relu=True)
def new(flags, copy=True):
    _lib.check(_lib.load().vqa_stem_nchw(ops._ptr(x), ops._ptr(xn if copy else None), ops._ptr(dst), ops._ptr(wp), ops._ptr(b), HW * Cc, N, Cc, N, Cin, flags, ops._stream()))
timeit(old, 'layout pass + vqa_conv_tiles')
timeit(conv_only, 'vqa_conv_tiles alone')
timeit(lambda: new(1), 'vqa_stem_nchw')
timeit(lambda: new(1, False), 'vqa_stem_nchw without the channels-last copy')
timeit(lambda: new(1 | 4), 'producers only load (no strip, no image, no copy)')
timeit(lambda: new(1 | 16, False), 'all samples read the same eight inputs (loads hit in L2), no copy')
timeit(lambda: new(1 | 8, False), 'producers fill the images from registers (no loads, no copy)')
timeit(lambda: new(1 | 2), 'producers idle (barriers only)')
