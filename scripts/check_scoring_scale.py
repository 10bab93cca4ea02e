import sys, os
sys.path.insert(0, '.')
import numpy as np, torch
from vqa_amd import synth, vocab as V
from vqa_amd.module_net import ModuleNet
M, K = 512, 8
vocab = V.make_vocab()
net = ModuleNet(False, vocab); net.load_state_dict(synth.synth_state_dict(17, vocab)); net.cuda().eval()
net.keep_state = True
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(4000)
feats = torch.randn(M, 1024, 14, 14, device=dev, generator=g).clamp_(min=0)
base = synth.synth_programs(4000, M, vocab)
prog = np.repeat(base, K, axis=0)
rng = np.random.default_rng(4000)
for i in range(M * K):
    if i % K:
        n = int((prog[i] != 0).sum()); cut = rng.integers(1, max(2, n))
        prog[i, :cut] = rng.integers(4, 44, size=cut)
img = np.repeat(np.arange(M), K)
progs = torch.from_numpy(prog).to(dev)
with torch.no_grad():
    whole = net.score_candidates(feats, progs, img)
    c1_whole = net.last_state['c1'].view(-1, 196 * 512)[:M * K].clone()
    parts, c1_parts = [], []
    for q0 in range(0, M, 64):
        sl = slice(q0 * K, (q0 + 64) * K)
        parts.append(net.score_candidates(feats[q0:q0 + 64], progs[sl], img[sl] - q0))
        c1_parts.append(net.last_state['c1'].view(-1, 196 * 512)[:64 * K].clone())
    torch.cuda.synchronize()
parts, c1_parts = torch.cat(parts), torch.cat(c1_parts)
print('classifier conv outputs equal:', torch.equal(c1_whole, c1_parts))
print('logits max rel diff:', ((whole - parts).abs().max() / parts.abs().max()).item())
