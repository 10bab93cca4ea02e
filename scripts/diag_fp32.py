"""Per-parameter relative error of the fp32 HIP path against the fp32 oracle (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from oracle.module_net_ref import ModuleNetRef
from vqa_amd import synth, vocab as V
from vqa_amd.module_net import ModuleNet
vocab = V.make_vocab()
dims = dict(feature_dim=(96, 14, 14), module_dim=128, classifier_proj_dim=128, classifier_fc_layers=(128,))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
prog = synth.synth_programs(3, n, vocab, max_len=12, max_nodes=9)
x, ans = synth.synth_features(5, n, dims['feature_dim']), synth.synth_answers(5, n)
sd = synth.synth_state_dict(7, vocab, **dims)
xt, pt, at = torch.from_numpy(x), torch.from_numpy(prog), torch.from_numpy(ans)
ref = ModuleNetRef(vocab, sd); rl = ref(xt, pt); F.cross_entropy(rl, at).backward()
net = ModuleNet(False, vocab, **dims); net.load_state_dict(sd); net.cuda().train(); net.precision = 'fp32'
lg = net(xt.cuda(), pt.cuda()); F.cross_entropy(lg, at.cuda()).backward(); torch.cuda.synchronize()
rel = lambda a, b: ((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm().clamp_min(1e-30)).item()
print('logits', rel(lg, rl))
rg = ref.grads()
errs = sorted(((rel(p.grad, rg[k]), k) for k, p in net.named_parameters() if p.grad is not None), reverse=True)
for e, k in errs[:12]: print('%.3e %s' % (e, k))
print('median %.3e' % errs[len(errs) // 2][0])
# ---- stem gradient pieces from the kept state
net2 = ModuleNet(False, vocab, **dims); net2.load_state_dict(sd); net2.cuda().train(); net2.precision = 'fp32'; net2.keep_state = True
lg = net2(xt.cuda(), pt.cuda()); F.cross_entropy(lg, at.cuda()).backward(); torch.cuda.synchronize()
st = net2.last_state; plan = st['plan']; c = 128
arena = st['arena'].view(-1, 196, c)
nodes = plan.nodes(); N = n
sF, sDF = plan.slot('F', 0), plan.slot('dF', 0)
info = plan.info
slot_dFp = info.slots_fwd + 2 * info.num_nodes + info.num_binary
scene = vocab['program_token_to_idx']['scene']
sc_nodes = [v for v in range(len(nodes)) if nodes[v, 1] == scene]
dF_manual = torch.zeros(N, 196, c, device='cuda')
for idx, v in enumerate(sc_nodes):
    dF_manual[nodes[v, 0]] += arena[slot_dFp + idx]
dF_manual *= (arena[sF:sF + N] > 0)
print('dF kernel vs manual', rel(arena[sDF:sDF + N], dF_manual))
# stem.2 weight grad from (S0, dF) with torch
sS0 = plan.slot('S0', 0)
S0 = arena[sS0:sS0 + N].permute(0, 2, 1).reshape(N, c, 14, 14)
dF = arena[sDF:sDF + N].permute(0, 2, 1).reshape(N, c, 14, 14)
gw = torch.nn.grad.conv2d_weight(S0, (c, c, 3, 3), dF, padding=1)
print('stem.2.weight: kernel vs torch-from-arena', rel(net2.stem[2].weight.grad, gw), ' torch-from-arena vs oracle', rel(gw, rg['stem.2.weight']))
# ---- one scene node's input gradient, recomputed with torch from the arena's dH and dZ
sDZ, sDH = plan.slot('dZ', 0), plan.slot('dH', 0)
W1 = net2._modules['scene'].conv1.weight.detach()
worst = 0
for idx, v in enumerate(sc_nodes[:20]):
    dH = arena[sDH + v].t().reshape(1, c, 14, 14)
    dZ = arena[sDZ + v].t().reshape(1, c, 14, 14)
    ref_dfp = F.conv_transpose2d(dH, W1, padding=1) + dZ
    got = arena[slot_dFp + idx].t().reshape(1, c, 14, 14)
    worst = max(worst, rel(got, ref_dfp))
print('dFp kernel vs torch-from-arena, worst of 20:', worst)
# oracle-side gradient at F: rerun the oracle keeping F
ref2 = ModuleNetRef(vocab, sd)
feats = ref2.stem(xt); feats.retain_grad()
lg2 = ref2.classifier(ref2.forward_modules(feats, pt)); F.cross_entropy(lg2, at).backward()
dF_ref = feats.grad.permute(0, 2, 3, 1).reshape(N, 196, c)
dF_hip_unmasked = torch.zeros(N, 196, c, device='cuda')
for idx, v in enumerate(sc_nodes):
    dF_hip_unmasked[nodes[v, 0]] += arena[slot_dFp + idx]
print('d loss / d F (before the ReLU mask): hip vs oracle', rel(dF_hip_unmasked, dF_ref))
per = [(rel(dF_hip_unmasked[i], dF_ref[i]), i, int((nodes[:, 0] == i).sum())) for i in range(N)]
print(sorted(per, reverse=True)[:5])
# ---- stem activations after backward vs a torch recomputation
w0, b0 = sd['stem.0.weight'], sd['stem.0.bias']
w2, b2 = sd['stem.2.weight'], sd['stem.2.bias']
S0_ref = F.relu(F.conv2d(xt, w0, b0, padding=1)); F_ref = F.relu(F.conv2d(S0_ref, w2, b2, padding=1))
nhwc = lambda t: t.permute(0, 2, 3, 1).reshape(N, 196, c)
print('S0 arena vs torch', rel(arena[sS0:sS0 + N], nhwc(S0_ref)), ' F arena vs torch', rel(arena[sF:sF + N], nhwc(F_ref)))
print('mask mismatches F:', int(((arena[sF:sF + N] > 0).cpu() != (nhwc(F_ref) > 0)).sum()), ' S0:', int(((arena[sS0:sS0 + N] > 0).cpu() != (nhwc(S0_ref) > 0)).sum()))
dz1_ref = (dF_ref * (nhwc(F_ref) > 0))
print('dF masked: arena vs oracle', rel(arena[sDF:sDF + N], dz1_ref))
gw_t = torch.nn.grad.conv2d_weight(S0_ref, (c, c, 3, 3), dz1_ref.reshape(N, 14, 14, c).permute(0, 3, 1, 2).contiguous(), padding=1)
print('torch conv2d_weight(S0_ref, dz1_ref) vs oracle stem.2.weight.grad', rel(gw_t, rg['stem.2.weight']))
