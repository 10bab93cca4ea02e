"""End to end: the executor trained with FusedAdam on one fixed batch memorises it - the loss falls
from ln(32) to a small fraction of it and every logit stays finite - the reference's training step
(Train_model.py:311-315: forward, cross-entropy, backward, Adam) repeated 80 times."""
import math

import pytest
import torch
import torch.nn.functional as F

from vqa_amd import synth

pytestmark = pytest.mark.gpu


def test_a_fixed_batch_is_memorised(vocab):
    from vqa_amd.module_net import ModuleNet
    from vqa_amd.optim import FusedAdam
    dims = dict(feature_dim=(128, 14, 14), module_dim=128, classifier_proj_dim=128, classifier_fc_layers=(256,))
    net = ModuleNet(False, vocab, **dims)
    net.load_state_dict(synth.synth_state_dict(51, vocab, **dims))
    net.cuda().train()
    opt = FusedAdam(net.parameters(), lr=5e-4, weight_decay=5e-5, executor=net)
    n = 64
    prog = torch.from_numpy(synth.synth_programs(52, n, vocab)).cuda()
    x = torch.from_numpy(synth.synth_features(53, n, dims['feature_dim'])).cuda()
    ans = torch.from_numpy(synth.synth_answers(54, n)).cuda()
    losses = []
    for _ in range(80):
        opt.zero_grad(set_to_none=True)
        logits = net(x, prog)
        loss = F.cross_entropy(logits, ans)
        loss.backward()
        opt.step()
        losses.append(loss.item())
        assert torch.isfinite(logits).all()
    assert losses[0] > 0.8 * math.log(32) and losses[-1] < 0.15 * losses[0], (losses[0], losses[-1])
    with torch.no_grad():
        net.eval()
        assert (net(x, prog).argmax(1) == ans).float().mean().item() > 0.9
