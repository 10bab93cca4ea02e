"""The classifier's last linear layer (vqa_head_forward / vqa_head_backward) and the one-launch
cross-entropy (vqa_xent), through the C ABI, against fp32 torch on the same bf16-rounded operands;
and the executor with and without them.  Tolerances: 1e-5 relative for fp32 sums of the same products
in another order; 1 bf16 ulp (2^-8 relative) where the result is rounded to bf16."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def dev():
    return torch.device('cuda:0')


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize('N,Fdim,A', [(1024, 1024, 32), (6, 128, 32), (257, 256, 28), (64, 512, 64), (1000, 1024, 40)])
def test_head_forward_and_backward_against_torch(N, Fdim, A):
    from vqa_amd import ops
    ld_act, ld_w = Fdim + 64, Fdim          # a padded activation stride, as a workspace slice may have
    act = bf(torch.relu(rnd((N, Fdim), 1)))
    w = bf(rnd((A, Fdim), 2, 0.05))
    b = rnd((A,), 3, 0.1)
    act_d = torch.zeros(N, ld_act, dtype=torch.bfloat16, device=dev())
    act_d[:, :Fdim] = act.to(torch.bfloat16)
    w_d = torch.zeros(128, ld_w, dtype=torch.bfloat16, device=dev())
    w_d[:A] = w.to(torch.bfloat16)
    logits = torch.full((N, A), float('nan'), device=dev())
    ops.head_forward(act_d, ld_act, w_d, ld_w, b.to(dev()), logits, N, Fdim, A)
    ref = act.double() @ w.double().t() + b.double()
    assert rel(logits.cpu(), ref) <= 1e-5

    dl = rnd((N, A), 4, 0.01)
    g = bf(dl)
    gin = torch.full((N, Fdim), float('nan'), dtype=torch.bfloat16, device=dev())
    dw = torch.full((A, Fdim), float('nan'), device=dev())
    db_in = torch.full((Fdim,), float('nan'), device=dev())
    db = torch.full((A,), float('nan'), device=dev())
    wk = torch.empty(ops.head_backward_workspace(N, Fdim, A), device=dev())
    for _ in range(2):
        ops.head_backward(dl.to(dev()), act_d, ld_act, w_d, ld_w, gin, Fdim, dw, db_in, db, wk, N, Fdim, A)
    gin_ref = (g.double() @ w.double()) * (act > 0)
    got = gin.float().cpu()
    # bf16 rounding of the result (one ulp of slack where the fp32 sum sits on a rounding boundary) on top of
    # the fp32 summation error, which scales with the terms, not with their sum
    terms = g.abs().double() @ w.abs().double()
    assert ((got.double() - gin_ref).abs() <= gin_ref.abs() * 2.0 ** -7 + terms * 1e-6).all()
    assert rel(dw.cpu(), g.double().t() @ act.double()) <= 1e-5
    assert rel(db.cpu(), dl.double().sum(0)) <= 1e-5
    assert rel(db_in.cpu(), got.double().sum(0)) <= 1e-5     # column sums of what was stored
    # a second call gives the same bits (sums in a fixed order, outputs written whole)
    again = [t.clone() for t in (gin, dw, db_in, db)]
    ops.head_backward(dl.to(dev()), act_d, ld_act, w_d, ld_w, gin, Fdim, dw, db_in, db, wk, N, Fdim, A)
    for t, u in zip((gin, dw, db_in, db), again):
        assert torch.equal(t, u)


@pytest.mark.parametrize('N,A', [(1024, 32), (5, 32), (3000, 28), (1, 7)])
def test_cross_entropy_matches_torch(N, A):
    from vqa_amd.loss import CrossEntropyLoss, cross_entropy
    logits = (rnd((N, A), 5, 3.0)).to(dev()).requires_grad_(True)
    target = torch.randint(0, A, (N,), generator=torch.Generator().manual_seed(6)).to(dev())
    loss = CrossEntropyLoss()(logits, target)
    (loss * 2.5).backward()
    ref_logits = logits.detach().double().cpu().requires_grad_(True)
    ref = F.cross_entropy(ref_logits, target.cpu())
    (ref * 2.5).backward()
    assert abs(loss.item() - ref.item()) <= 1e-5 * abs(ref.item())
    assert rel(logits.grad.cpu(), ref_logits.grad) <= 1e-5
    assert torch.equal(cross_entropy(logits.detach(), target), loss.detach())      # same bits on every run
    with pytest.raises(ValueError):
        cross_entropy(logits.detach().cpu(), target.cpu())


def test_executor_with_and_without_the_fused_head():
    """Same logits to fp32 summation order, same gradients to bf16 rounding of the hidden layer's input
    gradient, with vqa_head_* and with the GEMM path."""
    from vqa_amd import synth, vocab as V
    from vqa_amd.module_net import ModuleNet
    from vqa_amd.loss import cross_entropy
    vocab = V.make_vocab()
    dims = dict(feature_dim=(128, 14, 14), module_dim=128, classifier_proj_dim=128, classifier_fc_layers=(256,))
    n = 24
    prog = torch.from_numpy(synth.synth_programs(11, n, vocab, max_len=12, max_nodes=9)).to(dev())
    x = torch.from_numpy(synth.synth_features(11, n, dims['feature_dim'])).to(dev())
    ans = torch.from_numpy(synth.synth_answers(11, n)).to(dev())
    sd = synth.synth_state_dict(11, vocab, **dims)
    out = {}
    for fused in (True, False):
        net = ModuleNet(False, vocab, **dims)
        net.load_state_dict(sd)
        net.to(dev()).train()
        net.fuse_head = fused
        logits = net(x, prog)
        loss = cross_entropy(logits, ans) if fused else F.cross_entropy(logits, ans)
        loss.backward()
        out[fused] = (logits.detach().cpu(), loss.item(), {k: p.grad.detach().cpu() for k, p in net.named_parameters() if p.grad is not None})
    assert rel(out[True][0], out[False][0]) <= 1e-5
    assert abs(out[True][1] - out[False][1]) <= 1e-5 * abs(out[False][1])
    assert out[True][2].keys() == out[False][2].keys()
    for k, gr in out[False][2].items():
        assert rel(out[True][2][k], gr) <= 5e-3, k
