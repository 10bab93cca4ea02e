"""Candidate-program scoring (the executor side of the MAPO rollout configuration): M questions x K
sampled programs each.  Times (a) the reference's call pattern, one executor call per question on an
expand()-ed image (Train_model.py:339-347), (b) one score_candidates call over all M*K programs with the
host scheduler (common subtrees of an image's programs run once), (c) the same call with the programs
scheduled on the device (no copy to the host; every program's own nodes run).
Programs come from an untrained-sampler-like generator (many ill-formed rows), as the LSTM would emit.
usage: python scripts/bench_scoring.py [M] [K]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vqa_amd import synth, vocab as V
from vqa_amd.module_net import ModuleNet
from vqa_amd.schedule import Plan

M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
vocab = V.make_vocab()
net = ModuleNet(False, vocab)
net.load_state_dict(synth.synth_state_dict(17, vocab))
net.cuda().eval()
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(4000)
feats = torch.randn(M, 1024, 14, 14, device=dev, generator=g).clamp_(min=0)
# K candidates per question: a well-formed program and K-1 variants of it with a mutated tail
base = synth.synth_programs(4000, M, vocab)
prog = np.repeat(base, K, axis=0)
rng = np.random.default_rng(4000)
for i in range(M * K):
    if i % K:
        n = int((prog[i] != 0).sum())
        cut = rng.integers(1, max(2, n))
        prog[i, :cut] = rng.integers(4, 44, size=cut)        # new head, shared tail subtrees
img = np.repeat(np.arange(M), K)
p_sep, p_sh = Plan(prog, vocab), Plan(prog, vocab, image_index=img)
progs = torch.from_numpy(prog).to(dev)


def loop():
    out = []
    for i in range(M):
        out.append(net(feats[i].unsqueeze(0).expand(K, 1024, 14, 14), progs[i * K:(i + 1) * K]))
    return torch.cat(out)


def batched():
    return net.score_candidates(feats, progs, img)


with torch.no_grad():
    a, b = loop(), batched()
    torch.cuda.synchronize()
    # tree and classifier conv are bit-identical (scripts/check_scoring_scale.py); the linear layers pick
    # tile size and split-k by batch size, and a different fp32 summation order can move a bf16
    # rounding of their activations: equal at small batches, ~2e-4 apart at 4096 programs
    err = ((a - b).abs().max() / b.abs().max()).item()
    assert err < 1e-3, err
    res = {}
    img_dev = torch.from_numpy(img).to(dev, torch.int32)

    def batched_device():
        net.schedule_on_device = True
        try:
            return net.score_candidates(feats, progs, img_dev)
        finally:
            net.schedule_on_device = False
    c = batched_device()
    torch.cuda.synchronize()
    assert torch.equal(b, c)
    for name, fn in (('per_question_calls', loop), ('one_call', batched), ('one_call_schedule_on_device', batched_device)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        res[name] = {'ms': dt * 1e3, 'programs_per_s': M * K / dt}
print(json.dumps({'max_rel_diff': err, 'questions': M, 'candidates': K, 'nodes_unshared': p_sep.info.num_nodes,
                  'nodes_shared': p_sh.info.num_nodes, **res}))
