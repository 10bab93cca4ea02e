import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd import ops
torch.manual_seed(0)
M, N = 512, 16384
for K in (256, 320, 1024):
    a = (torch.randn(M, K) * 0.25).to(torch.bfloat16)
    b = (torch.randn(K, N) * 0.25).to(torch.bfloat16)
    ref = a.float() @ b.float()
    c = torch.full((M, N), 7.0, device='cuda')
    ops.gemm(a.cuda(), b.cuda(), c, M, N, K, lda=K, ldb=N, ldc=N, b_kmajor=True)
    torch.cuda.synchronize()
    err = (c.cpu() - ref).abs()
    bad = err > 0.05
    print('K', K, 'bad frac', bad.float().mean().item(), 'max err', err.max().item())
    if bad.any():
        rows = bad.any(1).nonzero().flatten()
        cols = bad.any(0).nonzero().flatten()
        print('  bad rows: n=%d first %s' % (len(rows), rows[:40].tolist()))
        print('  bad cols: n=%d first %s' % (len(cols), cols[:40].tolist()))
        t = bad[:256, :256]
        print('  tile(0,0) bad per 16x16 block:')
        print(t.view(16, 16, 16, 16).float().mean(dim=(1, 3)))
    # K-split test: which k contribute wrongly? use one-hot k
    for kk in (0, 31, 32, 63, 64, 200, K - 1):
        a1 = torch.zeros(M, K, dtype=torch.bfloat16); a1[:, kk] = 1
        c.fill_(7.0)
        ops.gemm(a1.cuda(), b.cuda(), c, M, N, K, lda=K, ldb=N, ldc=N, b_kmajor=True)
        torch.cuda.synchronize()
        e = (c.cpu() - b[kk].float()[None, :]).abs().max().item()
        print('   one-hot k=%d max err %.4f' % (kk, e))
