"""vqa_adam_fc1 alone: bytes moved per second on the default first linear layer (1024 x 100352)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd import _lib
lib = _lib.load()
F, P, HW = 1024, 512, 196
n = F * P * HW
d = 'cuda'
p, g, m, v = (torch.randn(n, device=d) * 0.01 for _ in range(4))
v.abs_()
packed = torch.empty(n, dtype=torch.bfloat16, device=d)
def run():
    _lib.check(lib.vqa_adam_fc1(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), packed.data_ptr(), F, P,
                                5e-5, 0.9, 0.999, 1e-8, 1e-4, 1.0, torch.cuda.current_stream().cuda_stream))
for _ in range(3): run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): run()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 20
print('adam_fc1: %.1f us, %.2f TB/s (30 bytes per parameter)' % (ms * 1e3, n * 30 / ms / 1e9))
