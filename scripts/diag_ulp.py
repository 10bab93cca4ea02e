"""How close is each kernel's output to the correctly rounded float64 result? (diagnostic)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from vqa_amd import ops
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from test_kernels_gpu import rnd, bf, to_slots, from_slots, tiles_tensor, dev, HW

def report(name, got, ref64):
    want = ref64.to(torch.float32).to(torch.bfloat16).to(torch.float32)
    diff = (got != want)
    rel = ((got.double() - ref64).norm() / ref64.norm()).item()
    relw = ((want.double() - ref64).norm() / ref64.norm()).item()
    print('%-28s mismatch vs correctly-rounded %.4f%%  rel err %.3e (rounding floor %.3e)' % (name, 100 * diff.float().mean().item(), rel, relw))

C, S = 128, 4
x = bf(rnd((S, C, 14, 14), 1)); res = bf(rnd((S, C, 14, 14), 2)); w = bf(rnd((1, C, C, 3, 3), 3, 0.05)); b = rnd((1, C), 4, 0.1)
arena = torch.cat([to_slots(x), to_slots(res), torch.zeros(S, HW, C, dtype=torch.bfloat16, device=dev())])
tiles = tiles_tensor([[i, -1, 2 * S + i, -1, S + i, -1, -1, 0] for i in range(S)])
wp = ops.pack_conv_weight(w.to(dev()), 1, C, C, 9, 0, C * 9, 9)
ops.conv_tiles(arena, arena, wp, tiles, S, taps=9, k_chunks=1, n_out=C, src_slot_stride=HW * C, dst_slot_stride=HW * C,
               src_pix_stride=C, dst_pix_stride=C, w_group_stride=9 * C * C, bias=b.to(dev()), res=arena, relu=True)
torch.cuda.synchronize()
ref = F.relu(F.conv2d(x.double(), w[0].double(), b[0].double(), padding=1) + res.double())
report('conv3x3+bias+res+relu', from_slots(arena[2 * S:], C), ref)
# fp32 torch CPU conv for comparison
r32 = F.relu(F.conv2d(x, w[0], b[0], padding=1) + res)
report('  (torch cpu fp32 conv)', r32.to(torch.bfloat16).float(), ref)

# wgrad
dy = bf(rnd((S, C, 14, 14), 21))
arena = torch.cat([to_slots(x), to_slots(dy)])
items = tiles_tensor([[0, 0, S, 0]])
dw = torch.zeros(1, 9, C, C, device=dev()); db = torch.zeros(1, C, device=dev())
ops.wgrad_tiles(arena, arena, (items, 0), 1, (tiles_tensor(list(range(S))), 0), (tiles_tensor([S + i for i in range(S)]), 0), dw, db,
                taps=9, k_chunks=1, n_out=C, src_slot_stride=HW * C, dy_slot_stride=HW * C, src_pix_stride=C, dy_pix_stride=C, dw_group_stride=9 * C * C)
out = torch.zeros(1, C, C, 3, 3, device=dev()); ops.unpack_conv_grad(dw, out, 1, C, C, 9, 0); torch.cuda.synchronize()
wd = torch.zeros(C, C, 3, 3, dtype=torch.float64, requires_grad=True); bd = torch.zeros(C, dtype=torch.float64, requires_grad=True)
F.conv2d(x.double(), wd, bd, padding=1).backward(dy.double())
print('wgrad rel err %.3e  bias %.3e' % (((out[0].cpu().double() - wd.grad).norm() / wd.grad.norm()).item(), ((db[0].cpu().double() - bd.grad).norm() / bd.grad.norm()).item()))

# gemm NT K large
M, N, K = 64, 128, 25088
a, bb = bf(rnd((M, K), 30)), bf(rnd((N, K), 31, 0.01))
ap = torch.zeros(128, K, dtype=torch.bfloat16); ap[:M] = a.to(torch.bfloat16)
c = torch.zeros(8, M, N, device=dev())
ops.gemm(ap.to(dev()), bb.to(torch.bfloat16).to(dev()), c, M, N, K, lda=K, ldb=K, ldc=N, split_k=8); torch.cuda.synchronize()
r = a.double() @ bb.double().t()
print('gemm nt rel err %.3e' % ((c.sum(0).cpu().double() - r).norm() / r.norm()).item())
