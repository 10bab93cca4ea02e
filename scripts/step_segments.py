"""Device-side duration of the segments of a train step (timing events at fixed points of the
executor, ModuleNet.segment_marks), for programs handed over on the device and in host memory."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd import synth, vocab as V
from vqa_amd.module_net import ModuleNet
from vqa_amd.optim import FusedAdam
vocab = V.make_vocab()
net = ModuleNet(False, vocab); net.load_state_dict(synth.synth_state_dict(17, vocab)); net.cuda().train()
opt = FusedAdam(net.parameters(), lr=1e-4, weight_decay=5e-5, executor=net)
batches = []
for b in range(4):
    prog = torch.from_numpy(synth.synth_programs(3000 + b, 1024, vocab))
    x = torch.randn(1024, 1024, 14, 14, device='cuda').clamp_(min=0)
    ans = torch.from_numpy(synth.synth_answers(b, 1024)).cuda()
    batches.append((x, prog, ans, prog.cuda()))
lf = torch.nn.CrossEntropyLoss()
import gc; gc.collect(); gc.freeze()
for mode in ('device', 'host'):
    net.segment_marks = []
    for i in range(24):
        x, prog, ans, prog_dev = batches[i % 4]
        if mode == 'device':
            prog = prog_dev
        opt.zero_grad(set_to_none=True)
        lf(net(x, prog), ans).backward()
        opt.step()
        net._mark('adam')
    torch.cuda.synchronize()
    marks = net.segment_marks
    per = len(marks) // 24
    names = [m[0] for m in marks[:per]]
    acc = [0.0] * per
    hacc = [0.0] * per
    steps = range(6, 24)
    for s_ in steps:
        for k in range(per):
            prev = marks[s_ * per + k - 1][1]
            acc[k] += prev.elapsed_time(marks[s_ * per + k][1]) * 1e3 / len(steps)
            hacc[k] += (marks[s_ * per + k][2] - marks[s_ * per + k - 1][2]) * 1e6 / len(steps)
    print('programs on the %s, device: %s  total %.0f us' % (mode, '  '.join('%s %.0f' % (n, a) for n, a in zip(names, acc)), sum(acc)))
    print('programs on the %s, host:   %s  total %.0f us' % (mode, '  '.join('%s %.0f' % (n, a) for n, a in zip(names, hacc)), sum(hacc)))
