"""vqa_amd.optim.FusedAdam against torch.optim.Adam (the reference's optimizer, Train_model.py:165-167):
same parameters after several steps, tensors without a gradient untouched, and the packed bf16 copy
of the first linear layer's weight that the fused kernel writes equal to a fresh pack."""
import pytest
import torch

from vqa_amd import synth

pytestmark = pytest.mark.gpu


def test_fused_adam_matches_torch_adam(vocab):
    from vqa_amd import ops
    from vqa_amd.module_net import ModuleNet
    from vqa_amd.optim import FusedAdam
    dims = dict(feature_dim=(128, 14, 14), module_dim=128, classifier_proj_dim=128, classifier_fc_layers=(128,))
    sd = synth.synth_state_dict(5, vocab, **dims)
    nets = []
    for _ in range(2):
        n = ModuleNet(False, vocab, **dims)
        n.load_state_dict(sd)
        nets.append(n.cuda())
    a, b = nets
    oa = FusedAdam(a.parameters(), lr=1e-3, weight_decay=5e-5, executor=a)
    ob = torch.optim.Adam(b.parameters(), lr=1e-3, weight_decay=5e-5)
    g = torch.Generator(device='cuda').manual_seed(1)
    names = [k for k, _ in a.named_parameters()]
    for step in range(4):
        skip = set(names[10 + step::7])                  # a changing set of tensors without gradient
        for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
            if k in skip:
                pa.grad = pb.grad = None
            else:
                gr = torch.randn(pa.shape, device='cuda', generator=g) * 0.01
                pa.grad, pb.grad = gr.clone(), gr.clone()
        oa.step()
        ob.step()
    torch.cuda.synchronize()
    # step counts live on the host between steps; reading the state (state_dict) writes them into
    # the `step` tensors of torch's Adam state layout, which torch's own Adam can then load
    sd_a = oa.state_dict()
    assert sd_a['state'].keys() == ob.state_dict()['state'].keys()
    torch.optim.Adam(a.parameters(), lr=1e-3).load_state_dict(sd_a)
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        err = (pa - pb).abs().max().item()
        assert err <= 2e-6 * max(1.0, pb.abs().max().item()), (k, err)
        sa, sb = oa.state[pa], ob.state[pb]
        assert float(sa['step']) == float(sb['step']), k
        torch.testing.assert_close(sa['exp_avg'], sb['exp_avg'], rtol=1e-4, atol=1e-9)
        torch.testing.assert_close(sa['exp_avg_sq'], sb['exp_avg_sq'], rtol=1e-4, atol=1e-12)
    # the bf16 copy written by the fused kernel is the pack of the updated weight
    lin = a.classifier[a._fc_idx[0]]
    fresh = torch.zeros_like(a._fc1_packed)
    ops.fc_weight_to_nhwc(lin.weight.detach().contiguous(), fresh, lin.weight.shape[0], fresh.shape[0], dims['classifier_proj_dim'])
    torch.cuda.synchronize()
    assert a._fc1_fresh and torch.equal(a._fc1_packed, fresh)


def test_training_with_fused_adam_equals_training_with_torch_adam(vocab):
    """Two executors, same batches, one stepped by FusedAdam (which hands the executor its packed
    first-layer weight) and one by torch.optim.Adam (after which the executor re-packs).  After one
    step at a learning rate large enough that a stale packed weight would show (every element moves
    by lr = 7 % of the weight scale), the second forward gives the same logits."""
    import torch.nn.functional as F
    from vqa_amd.module_net import ModuleNet
    from vqa_amd.optim import FusedAdam
    dims = dict(feature_dim=(128, 14, 14), module_dim=128, classifier_proj_dim=128, classifier_fc_layers=(128,))
    sd = synth.synth_state_dict(6, vocab, **dims)
    outs = []
    for fused in (True, False):
        net = ModuleNet(False, vocab, **dims)
        net.load_state_dict(sd)
        net.cuda().train()
        opt = (FusedAdam(net.parameters(), lr=1e-3, executor=net) if fused
               else torch.optim.Adam(net.parameters(), lr=1e-3))
        run = []
        for step in range(2):
            prog = torch.from_numpy(synth.synth_programs(80 + step, 12, vocab, max_len=12, max_nodes=9)).cuda()
            x = torch.from_numpy(synth.synth_features(80 + step, 12, dims['feature_dim'])).cuda()
            ans = torch.from_numpy(synth.synth_answers(80 + step, 12)).cuda()
            opt.zero_grad()
            logits = net(x, prog)
            F.cross_entropy(logits, ans).backward()
            opt.step()
            run.append(logits.detach().clone())
        outs.append(run)
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0], outs[1][0])
    moved = ((outs[0][1] - outs[0][0]).norm() / outs[0][0].norm()).item()
    rel = ((outs[0][1] - outs[1][1]).norm() / outs[1][1].norm()).item()
    assert moved > 0.05 and rel <= 5e-3, (moved, rel)


@pytest.mark.parametrize('width', [128, 256])
def test_fused_adam_rewrites_the_packed_convolution_weights(vocab, width):
    """Once a forward has laid the packed bf16 weight copies out, FusedAdam's update of a convolution
    weight also rewrites its two packed copies (forward and input-gradient form): they must equal a
    fresh pack of the updated weights bit for bit, tensors without a gradient must keep theirs, and
    the next forward (which then skips the re-pack) must give the logits of a freshly loaded executor."""
    from vqa_amd import ops
    from vqa_amd.module_net import ModuleNet
    from vqa_amd.optim import FusedAdam
    dims = dict(feature_dim=(128, 14, 14), module_dim=width, classifier_proj_dim=width, classifier_fc_layers=(128,))
    net = ModuleNet(False, vocab, **dims)
    net.load_state_dict(synth.synth_state_dict(8, vocab, **dims))
    net.cuda().train()
    dev = torch.device('cuda', torch.cuda.current_device())
    prog = torch.from_numpy(synth.synth_programs(90, 6, vocab, max_len=12, max_nodes=9)).cuda()
    x = torch.from_numpy(synth.synth_features(90, 6, dims['feature_dim'])).cuda()
    with torch.no_grad():
        net(x, prog)                                    # lays the packs out
    opt = FusedAdam(net.parameters(), lr=1e-2, weight_decay=5e-5, executor=net)
    g = torch.Generator(device='cuda').manual_seed(2)
    names = [k for k, _ in net.named_parameters()]
    for step in range(2):
        skip = set(names[12 + step::5])
        for k, p in net.named_parameters():
            p.grad = None if k in skip else torch.randn(p.shape, device='cuda', generator=g) * 0.01
        opt.step()
    assert net._conv_packs_fresh
    c, p_, nv = width, width, len(net._tokens)
    stk, bufs = net._stacks, net._pack_bufs
    g33 = c * c * 9
    s1, cl = net.stem[2].weight.detach(), net.classifier[0].weight.detach()
    fresh = {
        'w_conv1': ops.pack_conv_weight(stk['w1'], nv, c, c, 9, g33, c * 9, 9),
        'w_conv2': ops.pack_conv_weight(stk['w2'], nv, c, c, 9, g33, c * 9, 9),
        'wd_conv1': ops.pack_conv_weight(stk['w1'], nv, c, c, 9, g33, 9, c * 9, flip=True),
        'wd_conv2': ops.pack_conv_weight(stk['w2'], nv, c, c, 9, g33, 9, c * 9, flip=True),
        'w_proj': ops.pack_conv_weight(stk['wp'], nv, c, 2 * c, 1, 2 * c * c, 2 * c, 1),
        'wd_proj': ops.pack_conv_weight(stk['wp'], nv, 2 * c, c, 1, 2 * c * c, 1, 2 * c),
        'w_stem1': ops.pack_conv_weight(s1, 1, c, c, 9, 0, c * 9, 9),
        'wd_stem1': ops.pack_conv_weight(s1, 1, c, c, 9, 0, 9, c * 9, flip=True),
        'w_cls': ops.pack_conv_weight(cl, 1, p_, c, 1, 0, c, 1),
        'wd_cls': ops.pack_conv_weight(cl, 1, c, p_, 1, 0, 1, c),
    }
    torch.cuda.synchronize()
    for k, v in fresh.items():
        assert torch.equal(bufs[k], v), k
    with torch.no_grad():
        got = net(x, prog).clone()
    assert not net._conv_packs_fresh                   # consumed by that forward
    other = ModuleNet(False, vocab, **dims)
    other.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()})
    other.cuda().eval()
    with torch.no_grad():
        assert torch.equal(got, other(x, prog))
