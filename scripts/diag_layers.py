"""Layer-by-layer comparison of HIP intermediates against a bf16-storage torch computation (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from vqa_amd import synth, vocab as V
from vqa_amd.module_net import ModuleNet
from vqa_amd.schedule import Plan
HW = 196
vocab = V.make_vocab(); t2i = vocab['program_token_to_idx']
dims = dict(feature_dim=(96, 14, 14), module_dim=128, classifier_proj_dim=128, classifier_fc_layers=(128,))
prog = np.array([[t2i['scene']] + [0] * 11] * 2, dtype=np.int64)
n = 2
x = synth.synth_features(5, n, dims['feature_dim'])
sd = synth.synth_state_dict(7, vocab, **dims)
q = lambda t: t.to(torch.bfloat16).to(torch.float32)
net = ModuleNet(False, vocab, **dims); net.load_state_dict(sd); net.cuda()
plan = Plan(prog, vocab, 128, 16)
st = net._run_forward(torch.from_numpy(x).cuda(), plan, True); torch.cuda.synchronize()
arena = st['arena'].view(-1, HW, 128).float().cpu()
def slot(which, i): return arena[plan.slot(which, i)].permute(1, 0).reshape(128, 14, 14)
def cmp(name, got, ref):
    want = q(ref)
    print('%-8s mismatch %.4f%%  rel %.3e  (floor %.3e)' % (name, 100 * (got != want).float().mean().item(), ((got - ref).norm() / ref.norm()).item(), ((want - ref).norm() / ref.norm()).item()))
xt = q(torch.from_numpy(x)).double(); W = {k: q(v).double() if k.endswith('weight') else v.double() for k, v in sd.items()}
s0 = F.relu(F.conv2d(xt, W['stem.0.weight'], W['stem.0.bias'], padding=1))
cmp('S0', torch.stack([slot('S0', i) for i in range(n)]), s0.float())
s0 = q(s0.float()).double()
f = F.relu(F.conv2d(s0, W['stem.2.weight'], W['stem.2.bias'], padding=1))
cmp('F', torch.stack([slot('F', i) for i in range(n)]), f.float())
f = q(f.float()).double()
h = F.relu(F.conv2d(f, W['scene.conv1.weight'], W['scene.conv1.bias'], padding=1))
cmp('H', torch.stack([slot('H', i) for i in range(n)]), h.float())
h = q(h.float()).double()
y = F.relu(f + F.conv2d(h, W['scene.conv2.weight'], W['scene.conv2.bias'], padding=1))
cmp('Y', torch.stack([slot('Y', i) for i in range(n)]), y.float())
y = q(y.float()).double()
c1 = F.relu(F.conv2d(y, W['classifier.0.weight'], W['classifier.0.bias']))
got_c1 = st['c1'].view(-1, HW, 128)[:n].float().cpu().permute(0, 2, 1).reshape(n, 128, 14, 14)
cmp('C1', got_c1, c1.float())
c1 = q(c1.float()).double()
f1 = F.relu(F.linear(c1.reshape(n, -1), W['classifier.3.weight'], W['classifier.3.bias']))
cmp('F1', st['acts'][1].view(-1, 128)[:n].float().cpu(), f1.float())
f1 = q(f1.float()).double()
lg = F.linear(f1, W['classifier.5.weight'], W['classifier.5.bias'])
print('logits rel %.3e' % ((st['logits'].cpu().double() - lg).norm() / lg.norm()).item())
# ---- backward
up = torch.from_numpy(synth.counter_normal(3, n * 32).reshape(n, 32)) * 0.1
grads = net._run_backward(st, up.cuda()); torch.cuda.synchronize()
arena = st['arena'].view(-1, HW, 128).float().cpu()
g0 = q(up).double()
Wc, W3, W5 = W['classifier.0.weight'], W['classifier.3.weight'], W['classifier.5.weight']
dF1 = q((g0 @ W5).float()).double() * (f1 > 0)
dC1 = q((dF1 @ W3).float()).double().reshape(n, 128, 14, 14) * (c1 > 0)
dY = F.conv_transpose2d(dC1, Wc)
cmp('dZ', torch.stack([slot('dZ', i) for i in range(n)]), (dY * (y > 0)).float())
dZ = q(dY.float()).double() * (y > 0)
dH = F.conv_transpose2d(dZ, W['scene.conv2.weight'], padding=1)
cmp('dH', torch.stack([slot('dH', i) for i in range(n)]), (dH * (h > 0)).float())
dHp = q(dH.float()).double() * (h > 0)
dF = F.conv_transpose2d(dHp, W['scene.conv1.weight'], padding=1) + dZ
cmp('dF', torch.stack([slot('dF', i) for i in range(n)]), (dF * (f > 0)).float())
dFp = q(dF.float()).double() * (f > 0)
dS0 = F.conv_transpose2d(dFp, W['stem.2.weight'], padding=1)
cmp('dS0', torch.stack([slot('dS0', i) for i in range(n)]), (dS0 * (s0 > 0)).float())
