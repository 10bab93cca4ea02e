"""Time vqa_nchw_to_nhwc on the two shapes the executor uses (features, FC1 weight)."""
import torch
from vqa_amd import ops
dev = torch.device('cuda:0')
for name, n, c in (('features', 1024, 1024), ('fc1 weight', 1024, 512)):
    x = torch.randn(n, c, 196, device=dev)
    out = torch.empty(n, 196, c, dtype=torch.bfloat16, device=dev)
    ops.nchw_to_nhwc(x, out, n, c, c)
    ref = x.permute(0, 2, 1).to(torch.bfloat16)
    assert torch.equal(out, ref), name
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(20):
        ops.nchw_to_nhwc(x, out, n, c, c)
    ev[1].record(); torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) / 20 * 1e3
    gb = n * c * 196 * 6 / 1e9
    print('%s: %.1f us, %.2f TB/s' % (name, us, gb / us * 1e-3 * 1e3))
