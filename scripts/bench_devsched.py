"""Forward-only calls (evaluation, candidate scoring: Functions.py:152-179, Train_model.py:339-354)
with the programs in device memory, as a sampler leaves them: host scheduler (copy to the host,
parse, upload of the tile lists) against the scheduler on the device.  Per call: device time
(events), host time until the call returns, and wall time of back-to-back calls.
usage: python scripts/bench_devsched.py [batch] [reps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from vqa_amd import ops, synth, vocab as V
from vqa_amd.module_net import ModuleNet

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
vocab = V.make_vocab()
net = ModuleNet(False, vocab)
net.load_state_dict(synth.synth_state_dict(17, vocab))
net.cuda().eval()
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(N, 1024, 14, 14, device=dev, generator=g).clamp_(min=0)
x = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)      # read in place: no layout pass
sets = {'wellformed': synth.synth_programs(4000, N, vocab),
        'sampler': synth.synth_sampler_programs(4001, N, len(vocab['program_token_to_idx']), 30)}
out = {'batch': N, 'reps': reps}
for name, prog in sets.items():
    pt = torch.from_numpy(prog).to(dev)
    res = {}
    ref = None
    for mode, (on_dev, balance) in (('host_balanced', (False, True)), ('host', (False, False)),
                                    ('device', (True, False)), ('device_spread', (True, True))):
        net.schedule_on_device, net.balance_levels, net.device_spread_levels = on_dev, balance, balance
        with torch.no_grad():
            for _ in range(3):
                y = net(x, pt)
            torch.cuda.synchronize()
            ref = y.clone() if ref is None else ref
            assert torch.equal(y, ref), mode
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            host = 0.0
            t0 = time.perf_counter()
            e0.record()
            for _ in range(reps):
                h0 = time.perf_counter()
                net(x, pt)
                host += time.perf_counter() - h0
            e1.record()
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
        res[mode] = {'wall_ms': wall / reps * 1e3, 'device_ms': e0.elapsed_time(e1) / reps, 'host_ms_in_call': host / reps * 1e3,
                     'programs_per_s': N * reps / wall}
    # the scheduling kernels alone
    t2i = vocab['program_token_to_idx']
    T = prog.shape[1]
    nv = len(t2i)
    sched = ops.DeviceSchedule(torch.empty(ops.DeviceSchedule.words(N, T, nv), dtype=torch.int32, device=dev), N, T, nv)
    used = torch.empty(N, T, device=dev)
    arity = torch.tensor(V.arity_table(vocab), dtype=torch.int8, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for k in range(reps + 3):
        if k == 3:
            e0.record()
        ops.devsched_build(pt, arity, used, sched, t2i['<NULL>'], t2i['<START>'], t2i['scene'])
    e1.record()
    torch.cuda.synchronize()
    res['devsched_kernels_ms'] = e0.elapsed_time(e1) / reps
    li = sched.level_info.cpu().numpy().reshape(-1, 2)
    res['tiles'] = int(li[:, 1].sum())
    out[name] = res
print(json.dumps(out))
