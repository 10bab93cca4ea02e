import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd import ops
B, FI, FO = 1024, 100352, 1024
dev = torch.device('cuda')
x = (torch.randn(B, FI, device=dev) * 0.1).to(torch.bfloat16)
w = (torch.randn(FO, FI, device=dev) * 0.01).to(torch.bfloat16)
dy = (torch.randn(B, FO, device=dev) * 0.1).to(torch.bfloat16)
gin = torch.empty(B, FI, dtype=torch.bfloat16, device=dev)
gw = torch.empty(FO, FI, dtype=torch.float32, device=dev)
dyt = dy.t().contiguous()
dgrad = lambda: ops.gemm(dy, w, gin, B, FI, FO, lda=FO, ldb=FI, ldc=FI, b_kmajor=True, out_bf16=True, mask=x)
wgrad_t = lambda: ops.gemm(dyt, x, gw, FO, FI, B, lda=B, ldb=FI, ldc=FI, b_kmajor=True)
def timed(name, fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    print('%-34s %7.1f us  %6.0f TFLOP/s' % (name, ms * 1e3, 2.0 * B * FI * FO / ms / 1e9))
timed('dgrad ' + os.environ.get('VQA_GEMM_DBG', '0'), dgrad)
timed('wgrad (A transposed first) ' + os.environ.get('VQA_GEMM_DBG', '0'), wgrad_t)
