"""cProfile of the host side of the training loop (programs in host memory, so the host never waits
for the device): which Python functions the ~5 ms of host work per step are spent in."""
import sys, os, cProfile, pstats, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd import synth, vocab as V
from vqa_amd.module_net import ModuleNet
from vqa_amd.optim import FusedAdam
vocab = V.make_vocab()
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29544")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
from vqa_amd.dp import DataParallelExecutor
net = ModuleNet(False, vocab); net.load_state_dict(synth.synth_state_dict(17, vocab)); net.cuda().train()
dp = DataParallelExecutor(net)
opt = FusedAdam(net.parameters(), lr=1e-4, weight_decay=5e-5, executor=net)
batches = []
for b in range(4):
    prog = torch.from_numpy(synth.synth_programs(3000 + b, 1024, vocab))
    x = torch.randn(1024, 1024, 14, 14, device='cuda').clamp_(min=0)
    ans = torch.from_numpy(synth.synth_answers(b, 1024)).cuda()
    batches.append((x, prog, ans))
lf = torch.nn.CrossEntropyLoss()
def step(i):
    x, prog, ans = batches[i % 4]
    opt.zero_grad(set_to_none=True)
    lf(dp(x, prog), ans).backward()
    dp.finish_grad_sync()
    opt.step()
for i in range(4): step(i)
torch.cuda.synchronize()
# the backward half runs on autograd's device thread: profile every thread
prof = cProfile.Profile()
threading.setprofile(lambda *a: None)
import torch.autograd
prof.enable()
K = 20
for i in range(K): step(i)
prof.disable()
torch.cuda.synchronize()
st = pstats.Stats(prof)
st.sort_stats('cumulative')
import io
buf = io.StringIO(); st.stream = buf; st.print_stats(45)
print('per-step figures = total / %d' % K)
print(buf.getvalue()[:9000])
