"""Print per-parameter gradient relative errors of the HIP path vs the CPU oracle (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from oracle.module_net_ref import ModuleNetRef
from vqa_amd import synth, vocab as V
from vqa_amd.module_net import ModuleNet

vocab = V.make_vocab()
t2i = vocab['program_token_to_idx']
dims = dict(feature_dim=(96, 14, 14), module_dim=128, classifier_proj_dim=128, classifier_fc_layers=(128,))
mode = sys.argv[1] if len(sys.argv) > 1 else 'wf'
T = 12
sc, cnt, un, eq, fl = t2i['scene'], t2i['count'], t2i['union'], t2i['equal_color'], t2i['filter_color[red]']
pad = lambda r: (r + [0] * T)[:T]
if mode == 'wf':
    prog = synth.synth_programs(3, 13, vocab, max_len=T, max_nodes=9)
elif mode == 'unary':
    prog = np.array([pad([cnt, fl, sc, 2]), pad([fl, cnt, sc, 2])], dtype=np.int64)
elif mode == 'binary':
    prog = np.array([pad([eq, cnt, fl, sc, cnt, fl, sc, 2]), pad([un, sc, sc, 2])], dtype=np.int64)
elif mode == 'null':
    prog = np.array([pad([]), pad([2])], dtype=np.int64)
elif mode == 'scene':
    prog = np.array([pad([sc]), pad([sc])], dtype=np.int64)
n = prog.shape[0]
x = synth.synth_features(5, n, dims['feature_dim']); ans = synth.synth_answers(5, n)
sd = synth.synth_state_dict(7, vocab, **dims)
ref = ModuleNetRef(vocab, sd, storage='bf16'); rl = ref(torch.from_numpy(x), torch.from_numpy(prog))
up = rl.detach().clone().requires_grad_(True); F.cross_entropy(up, torch.from_numpy(ans)).backward()
rl.backward(up.grad)
net = ModuleNet(False, vocab, **dims); net.load_state_dict(sd); net.cuda()
lg = net(torch.from_numpy(x).cuda(), torch.from_numpy(prog).cuda())
lg.backward(up.grad.cuda()); torch.cuda.synchronize()
rel = lambda a, b: ((a.double().cpu() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
print(mode, 'logits', rel(lg.detach(), rl.detach()))
rows = []
for k, p in net.named_parameters():
    g = ref.grads()[k]
    if g is None: continue
    rows.append((rel(p.grad, g), k, g.norm().item()))
rows.sort(reverse=True)
for r in rows[:12]: print('%.4f %-40s norm %.3e' % r)
print('median', np.median([r[0] for r in rows]), 'min', rows[-1])
