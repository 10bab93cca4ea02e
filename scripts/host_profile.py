"""Where does a step's wall time go on the host? (diagnostic)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vqa_amd import synth, vocab as V
from vqa_amd.module_net import ModuleNet
from vqa_amd.schedule import Plan
vocab = V.make_vocab()
net = ModuleNet(False, vocab); net.load_state_dict(synth.synth_state_dict(17, vocab)); net.cuda().train()
opt = torch.optim.Adam(net.parameters(), lr=1e-4, weight_decay=5e-5, fused=True)
B = 1024
prog_np = synth.synth_programs(3000, B, vocab)
x = torch.randn(B, 1024, 14, 14, device='cuda').clamp_(min=0); prog = torch.from_numpy(prog_np).cuda()
ans = torch.from_numpy(synth.synth_answers(1, B)).cuda()
lf = torch.nn.CrossEntropyLoss()
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for it in range(6):
    t0 = sync()
    p = Plan(prog_np, vocab, 128, 16); t1 = time.perf_counter()
    b = torch.from_numpy(p.blob()); t2 = time.perf_counter()
    bd = b.cuda(); t3 = sync()
    opt.zero_grad(set_to_none=True)
    out = net(x, prog); t4 = time.perf_counter(); t4s = sync()
    loss = lf(out, ans); loss.backward(); t5 = time.perf_counter(); t5s = sync()
    opt.step(); t6 = time.perf_counter(); t6s = sync()
    print('plan %.2f blob %.2f upload %.2f | fwd host %.2f gpu-done %.2f | bwd host %.2f gpu-done %.2f | adam host %.2f done %.2f ms'
          % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3, (t4s-t3)*1e3, (t5-t4s)*1e3, (t5s-t4s)*1e3, (t6-t5s)*1e3, (t6s-t5s)*1e3))
