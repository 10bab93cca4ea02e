"""How far ahead of the device does the host run? Times the host's blocking waits and the plan build
inside the training step loop (diagnostic; monkeypatches, no product code involved)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vqa_amd import synth, vocab as V, schedule
from vqa_amd.module_net import ModuleNet
acc = {}
def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t; return r
    return w
torch.cuda.Event.synchronize = timed('event.synchronize', torch.cuda.Event.synchronize)
schedule.Plan.__init__ = timed('Plan()', schedule.Plan.__init__)
schedule.Plan.blob_into = timed('blob_into', schedule.Plan.blob_into)
vocab = V.make_vocab()
net = ModuleNet(False, vocab); net.load_state_dict(synth.synth_state_dict(17, vocab)); net.cuda().train()
net._run_forward = timed('forward host', net._run_forward)
net._run_backward = timed('backward host', net._run_backward)
from vqa_amd.optim import FusedAdam
opt = FusedAdam(net.parameters(), lr=1e-4, weight_decay=5e-5, executor=net) if os.environ.get('TORCH_ADAM') is None else torch.optim.Adam(net.parameters(), lr=1e-4, weight_decay=5e-5, fused=True)
opt.step = timed('adam host', opt.step)
B = 1024
batches = []
for b in range(4):
    prog = torch.from_numpy(synth.synth_programs(3000 + b, B, vocab)).cuda()
    x = torch.randn(B, 1024, 14, 14, device='cuda').clamp_(min=0)
    ans = torch.from_numpy(synth.synth_answers(b, B)).cuda()
    batches.append((x, prog, ans))
lf = torch.nn.CrossEntropyLoss()
import gc; gc.collect(); gc.freeze()
def step(i):
    x, prog, ans = batches[i % 4]
    opt.zero_grad(set_to_none=True)
    lf(net(x, prog), ans).backward()
    opt.step()
for i in range(3): step(i)
torch.cuda.synchronize(); acc.clear()
t0 = time.perf_counter()
K = 20
for i in range(K): step(i)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('step %.2f ms (host loop %.2f ms/step, final drain %.2f ms)' % ((t2 - t0) / K * 1e3, (t1 - t0) / K * 1e3, (t2 - t1) * 1e3))
for k, v in sorted(acc.items()): print('  %-20s %.3f ms/step' % (k, v / K * 1e3))
