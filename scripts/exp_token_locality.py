"""How much the fused node kernel owes to the token-sorted tile order: T unary tiles over V weight
groups, sorted by group (what the scheduler emits: an XCD sees V/8 of the groups) against shuffled
(every XCD sees every group, as it would if tiles were queued by program row)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd import ops
d = torch.device('cuda:0')
C, HW, V = 128, 196, 44
w = (torch.randn(V, C, C, 3, 3, device=d) * 0.05)
b = torch.randn(V, C, device=d) * 0.1
p1 = ops.pack_conv_weight(w, V, C, C, 9, C * C * 9, C * 9, 9)
p2 = ops.pack_conv_weight(w * 0.5, V, C, C, 9, C * C * 9, C * 9, 9)
g = torch.Generator().manual_seed(0)
for T in (256, 512, 1024, 2048):
    arena = torch.randn(3 * T, HW, C, device=d).to(torch.bfloat16)
    for order in ('sorted', 'shuffled'):
        tok = torch.arange(T) * V // T
        if order == 'shuffled':
            tok = tok[torch.randperm(T, generator=g)]
        tiles = torch.tensor([[i, -1, 2 * T + i, T + i, i, -1, -1, int(tok[i]), -1, -1, -1, -1] for i in range(T)],
                             dtype=torch.int32).reshape(-1).to(d)
        for _ in range(3):
            ops.resblock_tiles(arena, p1, p2, tiles, T, bias_first=b, bias_second=b, relu=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.resblock_tiles(arena, p1, p2, tiles, T, bias_first=b, bias_second=b, relu=True)
        e1.record(); torch.cuda.synchronize()
        print('T=%d %-9s %.1f us per launch' % (T, order, e0.elapsed_time(e1) * 100))
