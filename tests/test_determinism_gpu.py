"""Run-to-run determinism of a training step on the HIP path: the same weights and the same batch give
the same logits and the same gradient bits on every run - weight and bias gradients of every
convolution and linear layer are summed in a fixed order (no floating-point atomics on this path)."""
import pytest
import torch
import torch.nn.functional as F

from vqa_amd import synth

pytestmark = pytest.mark.gpu


def test_two_runs_of_a_training_step_give_the_same_bits(vocab):
    from vqa_amd.module_net import ModuleNet
    dims = dict(feature_dim=(256, 14, 14), module_dim=128, classifier_proj_dim=256, classifier_fc_layers=(256,))
    net = ModuleNet(False, vocab, **dims)
    net.load_state_dict(synth.synth_state_dict(41, vocab, **dims))
    net.cuda().train()
    n = 160
    prog = torch.from_numpy(synth.synth_programs(42, n, vocab)).cuda()
    x = torch.from_numpy(synth.synth_features(43, n, dims['feature_dim'])).cuda()
    ans = torch.from_numpy(synth.synth_answers(44, n)).cuda()
    runs = []
    for _ in range(3):
        for p in net.parameters():
            p.grad = None
        logits = net(x, prog)
        F.cross_entropy(logits, ans).backward()
        torch.cuda.synchronize()
        runs.append((logits.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}))
    assert len(runs[0][1]) > 40
    for logits, grads in runs[1:]:
        assert torch.equal(logits, runs[0][0])
        assert grads.keys() == runs[0][1].keys()
        for k, g in grads.items():
            assert torch.equal(g, runs[0][1][k]), k

