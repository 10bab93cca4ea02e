"""Phase timing of the fused node kernel from its shader-clock stamps (diagnostic).
Launches T identical unary tiles and prints, per phase, the median duration in us (100 MHz stamps
are converted with the constant-rate counter frequency)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd import ops
d = torch.device('cuda:0')
C, HW = 128, 196
for T in (1, 256, 512, 2560):
    arena = torch.randn(3 * T, HW, C, device=d).to(torch.bfloat16)
    w = (torch.randn(4, C, C, 3, 3, device=d) * 0.05)
    b = torch.randn(4, C, device=d) * 0.1
    p1 = ops.pack_conv_weight(w, 4, C, C, 9, C * C * 9, C * 9, 9)
    p2 = ops.pack_conv_weight(w, 4, C, C, 9, C * C * 9, C * 9, 9)
    tiles = torch.tensor([[i, -1, 2 * T + i, T + i, i, -1, -1, i % 4, -1, -1, -1, -1] for i in range(T)], dtype=torch.int32).reshape(-1).to(d)
    stamps = torch.zeros(T, 8, dtype=torch.int64, device=d)
    for _ in range(2):
        ops.resblock_tiles(arena, p1, p2, tiles, T, bias_first=b, bias_second=b, relu=True, phase_stamps=stamps)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.resblock_tiles(arena, p1, p2, tiles, T, bias_first=b, bias_second=b, relu=True, phase_stamps=stamps)
    e1.record(); torch.cuda.synchronize()
    s = stamps.cpu().double()
    ph = (s[:, 1:7] - s[:, 0:6])
    tot = s[:, 6] - s[:, 0]
    span = (s[:, 6].max() - s[:, 0].min()).item()
    wall = e0.elapsed_time(e1) * 1e3
    tick = wall / span     # us per tick, roughly (launch overhead included in wall)
    names = ['stage', 'pre', 'K1', 'epi1', 'K2', 'epi2']
    print('T=%d wall %.1f us, span %d ticks (%.4f us/tick incl. launch)' % (T, wall, span, tick))
    print('   median ticks: ' + ' '.join('%s %.0f' % (n, ph[:, i].median().item()) for i, n in enumerate(names)) + ' | total %.0f' % tot.median().item())
