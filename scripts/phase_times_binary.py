"""Fused node kernel, shader-clock phases of binary nodes against unary ones: forward (the 1x1
projection ahead of the body) and backward (a child's half of the transposed projection ahead of its body)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd import ops
d = torch.device('cuda:0')
C, HW = 128, 196
for T in (256, 512):
    arena = torch.randn(10 * T, HW, C, device=d).to(torch.bfloat16)
    w = (torch.randn(4, C, C, 3, 3, device=d) * 0.05)
    wp = (torch.randn(4, C, 2 * C, 1, 1, device=d) * 0.05)
    p1 = ops.pack_conv_weight(w, 4, C, C, 9, C * C * 9, C * 9, 9)
    pre = ops.pack_conv_weight(wp, 4, C, 2 * C, 1, 2 * C * C, 2 * C, 1)
    post = ops.pack_conv_weight(wp, 4, 2 * C, C, 1, 2 * C * C, 1, 2 * C)
    kinds = {
        'forward unary': ([[i, -1, 2 * T + i, T + i, i, -1, -1, i % 4, -1, -1, -1, -1] for i in range(T)], {}),
        'forward binary': ([[i, 5 * T + i, 2 * T + i, T + i, 6 * T + i, -1, -1, i % 4, -1, -1, -1, -1] for i in range(T)],
                           dict(w_pre=pre, num_pre=T)),
        'backward unary': ([[i, -1, 2 * T + i, T + i, i, 3 * T + i, 4 * T + i, i % 4, -1, -1, -1, -1] for i in range(T)], {}),
        'backward child of binary': ([[9 * T + i, -1, 2 * T + i, T + i, i, 3 * T + i, 4 * T + i, i % 4, 2 * ((i + 1) % 4) + i % 2, 5 * T + i, 0, 0]
                                      for i in range(T)], dict(w_pre=post, num_pre=T // 2)),
    }
    for name, (rows, kw) in kinds.items():
        tiles = torch.tensor(rows, dtype=torch.int32).reshape(-1).to(d)
        stamps = torch.zeros(T, 8, dtype=torch.int64, device=d)
        for _ in range(2):
            ops.resblock_tiles(arena, p1, p1, tiles, T, relu=name.startswith('forward'), phase_stamps=stamps, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.resblock_tiles(arena, p1, p1, tiles, T, relu=name.startswith('forward'), phase_stamps=stamps, **kw)
        e1.record(); torch.cuda.synchronize()
        s = stamps.cpu().double()
        last = 6
        ph = s[:, 1:last + 1] - s[:, 0:last]
        names = ['stage', 'pre', 'K1', 'epi1', 'K2', 'epi2', 'post'][:last]
        print('T=%d %-24s wall %5.1f us | ' % (T, name, e0.elapsed_time(e1) * 1e3) +
              ' '.join('%s %.0f' % (n, ph[:, i].median().item()) for i, n in enumerate(names)) +
              ' | total %.0f' % (s[:, last] - s[:, 0]).median().item())
