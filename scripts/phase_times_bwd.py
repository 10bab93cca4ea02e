"""Phase timing of the fused node kernel on backward-shaped tiles (mask on the intermediate map,
residual + mask on the output), from its shader-clock stamps.  Compare with phase_times.py."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd import ops
d = torch.device('cuda:0')
C, HW = 128, 196
for T in (1, 256, 512, 2560):
    arena = torch.randn(6 * T, HW, C, device=d).to(torch.bfloat16)
    w = (torch.randn(4, C, C, 3, 3, device=d) * 0.05)
    p1 = ops.pack_conv_weight(w, 4, C, C, 9, C * C * 9, C * 9, 9)
    p2 = ops.pack_conv_weight(w, 4, C, C, 9, C * C * 9, C * 9, 9)
    # src0 = dZ, dst0 = dX, dst1 = dH, res = dZ, mask0 = X, mask1 = H
    tiles = torch.tensor([[i, -1, 2 * T + i, T + i, i, 3 * T + i, 4 * T + i, i % 4, -1, -1, -1, -1] for i in range(T)],
                         dtype=torch.int32).reshape(-1).to(d)
    stamps = torch.zeros(T, 8, dtype=torch.int64, device=d)
    for _ in range(2):
        ops.resblock_tiles(arena, p1, p2, tiles, T, relu=False, phase_stamps=stamps)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.resblock_tiles(arena, p1, p2, tiles, T, relu=False, phase_stamps=stamps)
    e1.record(); torch.cuda.synchronize()
    s = stamps.cpu().double()
    ph = (s[:, 1:7] - s[:, 0:6])
    tot = s[:, 6] - s[:, 0]
    names = ['stage', 'pre', 'K1', 'epi1', 'K2', 'epi2']
    print('T=%d wall %.1f us' % (T, e0.elapsed_time(e1) * 1e3))
    print('   median ticks: ' + ' '.join('%s %.0f' % (n, ph[:, i].median().item()) for i, n in enumerate(names)) + ' | total %.0f' % tot.median().item())
