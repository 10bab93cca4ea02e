"""Free-running gradient distances, per parameter tensor (diagnostic; writes a JSON table).

For one batch of the default model: the HIP bf16 path, the CPU oracle with bf16 storage points and
the fp32 oracle (pinned to the reference by tests/golden) all run free, with the same upstream
gradient d(loss)/d(logits) taken at the fp32 oracle's logits.  Reports rel(HIP, fp32) and
rel(bf16-storage oracle, fp32) per tensor: the second is what ANY bf16 executor of this network
sits from an fp32 run (ReLU masks of near-zero pre-activations flip), the first must not be worse.

    python scripts/diag_free_running.py [n_samples] [out.json]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

from oracle.module_net_ref import ModuleNetRef
from vqa_amd import synth, vocab as V
from vqa_amd.module_net import ModuleNet

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
out_path = sys.argv[2] if len(sys.argv) > 2 else 'gpurun_out/free_running_parity.json'
vocab = V.make_vocab()
seed = 17
sd = synth.synth_state_dict(seed, vocab)
prog = synth.synth_programs(seed, n, vocab)
x = synth.synth_features(seed, n)
ans = synth.synth_answers(seed, n)
xt, pt, at = torch.from_numpy(x), torch.from_numpy(prog), torch.from_numpy(ans)


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


r32 = ModuleNetRef(vocab, sd)
l32 = r32(xt, pt)
up = l32.detach().clone().requires_grad_(True)
F.cross_entropy(up, at).backward()
up = up.grad
l32.backward(up)
rb = ModuleNetRef(vocab, sd, storage='bf16')
lb = rb(xt, pt)
lb.backward(up)
net = ModuleNet(False, vocab)
net.load_state_dict(sd)
net.cuda().train()
lh = net(xt.cuda(), pt.cuda())
lh.backward(up.cuda())
torch.cuda.synchronize()
rows = {}
for k, p in net.named_parameters():
    g32 = r32.grads()[k]
    if g32 is None:
        continue
    rows[k] = {'hip_vs_fp32': rel(p.grad, g32), 'bf16_oracle_vs_fp32': rel(rb.grads()[k], g32),
               'hip_vs_bf16_oracle': rel(p.grad, rb.grads()[k]), 'norm': float(g32.double().norm())}
table = {'n_samples': n, 'seed': seed, 'logits_hip_vs_fp32': rel(lh, l32), 'logits_bf16_oracle_vs_fp32': rel(lb, l32),
         'tensors': rows}
ratios = np.array([v['hip_vs_fp32'] / max(v['bf16_oracle_vs_fp32'], 1e-12) for v in rows.values()])
table['ratio_hip_over_bf16_oracle'] = {'median': float(np.median(ratios)), 'max': float(ratios.max()),
                                       'argmax': list(rows)[int(ratios.argmax())]}
os.makedirs(os.path.dirname(out_path) or '.', exist_ok=True)
with open(out_path, 'w') as f:
    json.dump(table, f, indent=1)
print('n=%d logits hip %.3e bf16-oracle %.3e; ratio median %.3f max %.3f (%s)' % (
    n, table['logits_hip_vs_fp32'], table['logits_bf16_oracle_vs_fp32'], table['ratio_hip_over_bf16_oracle']['median'],
    table['ratio_hip_over_bf16_oracle']['max'], table['ratio_hip_over_bf16_oracle']['argmax']))
worst = sorted(rows.items(), key=lambda kv: -kv[1]['hip_vs_fp32'] / max(kv[1]['bf16_oracle_vs_fp32'], 1e-12))[:8]
for k, v in worst:
    print('  %-36s hip %.4f bf16-oracle %.4f' % (k, v['hip_vs_fp32'], v['bf16_oracle_vs_fp32']))
