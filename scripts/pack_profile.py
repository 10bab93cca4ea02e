import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd import synth, vocab as V
from vqa_amd.module_net import ModuleNet
vocab = V.make_vocab()
net = ModuleNet(False, vocab); net.cuda()
for it in range(4):
    net._packed_stale = True
    torch.cuda.synchronize(); t0 = time.perf_counter()
    net._pack_weights(torch.device('cuda:0'))
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print('pack host %.2f ms, done %.2f ms' % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
from torch.profiler import profile, ProfilerActivity
net._packed_stale = True
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    net._pack_weights(torch.device('cuda:0')); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=12, max_name_column_width=60))
