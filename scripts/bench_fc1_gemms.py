"""The three GEMMs of the first linear layer (batch 1024, 100352 -> 1024) timed alone and, for the two
backward ones, side by side on two streams as the executor runs them.  usage: bench_fc1_gemms.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd import ops
B, FI, FO = 1024, 100352, 1024
dev = torch.device('cuda')
x = (torch.randn(B, FI, device=dev) * 0.1).to(torch.bfloat16)
w = (torch.randn(FO, FI, device=dev) * 0.01).to(torch.bfloat16)
dy = (torch.randn(B, FO, device=dev) * 0.1).to(torch.bfloat16)
split = 16
acc = torch.empty(split * B * FO, dtype=torch.float32, device=dev)
gw = torch.empty(FO, FI, dtype=torch.float32, device=dev)
gin = torch.empty(B, FI, dtype=torch.bfloat16, device=dev)
fwd = lambda: ops.gemm(x, w, acc, B, FO, FI, lda=FI, ldb=FI, ldc=FO, split_k=split)
wgrad = lambda: ops.gemm(dy, x, gw, FO, FI, B, lda=FO, ldb=FI, ldc=FI, a_kmajor=True, b_kmajor=True)
dgrad = lambda: ops.gemm(dy, w, gin, B, FI, FO, lda=FO, ldb=FI, ldc=FI, b_kmajor=True, out_bf16=True, mask=x)
side = torch.cuda.Stream()
def both():
    ev = torch.cuda.Event(); ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev); wgrad(); done = torch.cuda.Event(); done.record(side)
    dgrad()
    torch.cuda.current_stream().wait_event(done)
def timed(name, fn, flops, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    print('%-28s %7.1f us  %6.0f TFLOP/s' % (name, ms * 1e3, flops / ms / 1e9))
F = 2.0 * B * FI * FO
timed('forward (split-k 16)', fwd, F)
timed('weight gradient', wgrad, F)
timed('input gradient (masked)', dgrad, F)
timed('both gradients, two streams', both, 2 * F)
# the vendor library on the same shapes (bf16 in, bf16 out; fp32 out where torch offers it)
xt = x  # [B][FI]
timed('torch.mm forward bf16', lambda: torch.mm(x, w.t()), F)
timed('torch.mm wgrad bf16 out', lambda: torch.mm(dy.t(), x), F)
timed('torch.mm dgrad bf16 out', lambda: torch.mm(dy, w), F)
try:
    timed('torch.mm wgrad fp32 out', lambda: torch.mm(dy.t(), x, out_dtype=torch.float32), F)
except Exception as e:
    print('fp32-out mm unavailable:', type(e).__name__, str(e)[:100])
