"""The scheduler on the device (vqa_devsched_build) against three checkers: the oracle's restatement
of forward_modules_ints_helper (Module_Model.py:196-222), row by row; the hook traces captured from
the reference itself (tests/golden/schedule_quirks.json), case by case; and the host scheduler
(levels, histogram, slots).  The same trees per row, the same earliest levels, the same used_fns,
and, through the executor, the same logits bit for bit - on well-formed programs, on an untrained
sampler's rows (LSTM_Model.py:205-235) and on every quirk row."""
import json
import os

import numpy as np
import pytest
import torch

from oracle.module_net_ref import parse_row
from vqa_amd import synth
from vqa_amd import vocab as V
from vqa_amd.schedule import Plan

pytestmark = pytest.mark.gpu

DIMS = dict(feature_dim=(128, 14, 14), module_dim=128, classifier_proj_dim=128, classifier_fc_layers=(128,))


def quirk_rows(t2i, T):
    """The interpreter's quirk cases (Module_Model.py:198-208)."""
    sc, cnt, un, eq = t2i['scene'], t2i['count'], t2i['union'], t2i['equal_color']
    fl, rl = t2i['filter_color[red]'], t2i['relate[left]']
    N_, S_, E_, U_ = V.NULL, V.START, V.END, V.UNK

    def pad(toks):
        return (list(toks) + [N_] * T)[:T]
    return [pad([]), pad([S_, cnt, fl, sc, E_]), pad([cnt, fl, rl, sc, E_]), pad([cnt, E_]), [un] * T,
            pad([sc, cnt, fl, un, E_]), pad([cnt, U_, fl, sc, E_]), pad([cnt, S_, S_, fl, sc, E_]),
            pad([eq, cnt, fl, sc, cnt, rl, fl, sc, E_]), pad([N_] * (T - 2) + [un, sc]), pad([N_] * (T - 1) + [un]),
            pad([N_] * (T - 1) + [S_]), pad([E_]), pad([un, eq, sc, sc, un, sc, fl, sc, E_]), [S_] * T, [fl] * T]


def device_schedule(prog, vocab, spread=False):
    from vqa_amd import ops
    t2i = vocab['program_token_to_idx']
    arity = torch.tensor(V.arity_table(vocab), dtype=torch.int8).cuda()
    N, T = prog.shape
    nv = arity.numel()
    buf = torch.full((ops.DeviceSchedule.words(N, T, nv),), -7, dtype=torch.int32, device='cuda')
    sched = ops.DeviceSchedule(buf, N, T, nv)
    used = torch.full((N, T), -1.0, device='cuda')
    ops.devsched_build(torch.from_numpy(prog).cuda(), arity, used, sched, t2i['<NULL>'], t2i['<START>'], t2i['scene'],
                       spread=spread)
    torch.cuda.synchronize()
    return sched, used.cpu().numpy()


def oracle_trees(prog, vocab):
    """Per row, from the oracle's interpreter: the tree as nested (token id, inputs...) tuples and the
    row of used_fns."""
    t2i, i2t = vocab['program_token_to_idx'], vocab['program_idx_to_token']
    trees, used = [], []
    for row in prog:
        nodes, u = parse_row(row, i2t)

        def nest(v):
            return (t2i[nodes[v][0]],) + tuple(nest(c) for c in nodes[v][2])
        trees.append(nest(len(nodes) - 1))
        used.append(u)
    return trees, np.asarray(used, dtype=np.float32)


def check_against_host(prog, vocab, spread=False, want_trees=None, want_used=None):
    """Rebuild every row's tree from the device's tile records and compare it with the oracle's parse
    of the row (or with `want_trees`, the reference's own trace) and with the host plan's.
    Levels: the host's earliest ones without `spread`; with it, any levels in which inputs run
    before their users, no more of them than the deepest program needs."""
    scene = vocab['program_token_to_idx']['scene']
    N, T = prog.shape
    sched, used = device_schedule(prog, vocab, spread)
    if want_trees is None:
        want_trees, want_used = oracle_trees(prog, vocab)
    np.testing.assert_array_equal(used, want_used)
    plan = Plan(prog, vocab)
    nodes, roots = plan.nodes(), plan.roots()
    np.testing.assert_array_equal(used, plan.used_fns())
    err = sched.error.cpu().numpy()
    assert err[0] == 0 and err[1] == nodes[:, 5].max()
    li = sched.level_info.cpu().numpy().reshape(T + 1, 2)
    hist = sched.hist.cpu().numpy().reshape(T + 1, -1)
    want_hist = np.zeros_like(hist)
    for _s, tok, _c, _a, _b, lvl in nodes:
        if tok != scene:
            want_hist[lvl, tok] += 1
    if spread:
        np.testing.assert_array_equal(hist.sum(0), want_hist.sum(0))
        assert hist[0].sum() == 0 and hist[err[1] + 1:].sum() == 0
    else:
        np.testing.assert_array_equal(hist, want_hist)
    np.testing.assert_array_equal(li[:, 1], hist.sum(1))
    np.testing.assert_array_equal(li[:, 0], np.concatenate([[0], np.cumsum(hist.sum(1))[:-1]]))
    total = int(li[:, 1].sum())
    tiles = sched.tiles.cpu().numpy().reshape(N * T, 12)[:total]
    assert (tiles[:, 5:7] == -1).all() and (tiles[:, 8:] == -1).all()
    nvmax = N * T
    y, h, res = tiles[:, 2], tiles[:, 3], tiles[:, 4]
    assert len(set(y.tolist())) == total and y.min(initial=4 * N) >= 4 * N and y.max(initial=0) < 4 * N + nvmax
    np.testing.assert_array_equal(h, y + nvmax)
    binary = tiles[:, 1] >= 0
    assert len(set(res[binary].tolist())) == int(binary.sum())
    assert ((res[binary] >= 4 * N + 2 * nvmax) & (res[binary] < 4 * N + 3 * nvmax)).all()
    np.testing.assert_array_equal(res[~binary], tiles[~binary, 0])
    # tokens are grouped inside a level (one weight group's tiles are neighbours)
    level_of = {}
    for L in range(T + 1):
        seg = tiles[li[L, 0]:li[L, 0] + li[L, 1]]
        assert (np.diff(seg[:, 7]) >= 0).all()
        for t in seg:
            level_of[int(t[2])] = L
    by_y = {int(t[2]): t for t in tiles}

    def dev_tree(slot, row):
        if 2 * N <= slot < 3 * N:
            assert slot == 2 * N + row         # a row only ever reads its own image's scene output
            return (scene,), 0
        t = by_y[slot]
        kids = [dev_tree(int(s), row) for s in t[:2] if s >= 0]
        lvl = level_of[slot]
        if spread:
            assert lvl > max(k[1] for k in kids)
        else:
            assert lvl == 1 + max(k[1] for k in kids)
        return (int(t[7]),) + tuple(k[0] for k in kids), lvl

    def host_tree(v):
        return (int(nodes[v, 1]),) + tuple(host_tree(c) for c in nodes[v, 3:5] if c >= 0)
    cls = sched.cls_tiles.cpu().numpy().reshape(N, 8)
    np.testing.assert_array_equal(cls[:, 2], np.arange(N))
    assert (cls[:, [1, 3, 4, 5, 6]] == -1).all() and (cls[:, 7] == 0).all()
    for i in range(N):
        got = dev_tree(int(cls[i, 0]), i)[0]
        assert got == want_trees[i], i
        assert got == host_tree(roots[i]), i
    return hist.sum(1), want_hist.sum(1)


@pytest.mark.parametrize('spread', [False, True])
def test_reference_hook_traces(vocab, golden_dir, spread):
    """Every case of the fixture captured from the reference executor with forward hooks (execution
    order, inputs, used_fns; T = 12 and T = 30 rows): the device's trees and used_fns are the
    reference's."""
    with open(os.path.join(golden_dir, 'schedule_quirks.json')) as f:
        cases = json.load(f)['cases']
    by_T = {}
    for c in cases:
        by_T.setdefault(c['T'], []).append(c)
    assert len(by_T) >= 2
    for T, cs in sorted(by_T.items()):
        trees = []
        for c in cs:
            def nest(v, c=c):
                return (c['tokens'][v],) + tuple(nest(k) for k in c['children'][v] if k >= 0)
            trees.append(nest(len(c['tokens']) - 1))
        prog = np.array([c['program'] for c in cs], dtype=np.int64)
        assert prog.shape == (len(cs), T)
        check_against_host(prog, vocab, spread, want_trees=trees,
                           want_used=np.array([c['used_fns'] for c in cs], dtype=np.float32))


@pytest.mark.parametrize('spread', [False, True])
def test_wellformed_sampler_and_quirk_rows_give_the_host_schedule(vocab, spread):
    t2i = vocab['program_token_to_idx']
    check_against_host(synth.synth_programs(1, 257, vocab), vocab, spread)
    check_against_host(synth.synth_sampler_programs(2, 300, len(t2i), 30), vocab, spread)
    check_against_host(synth.synth_sampler_programs(3, 64, len(t2i), 7), vocab, spread)
    check_against_host(synth.synth_programs(4, 32, vocab, max_len=64, max_nodes=60), vocab, spread)
    check_against_host(synth.synth_programs(5, 16, vocab, with_start=True), vocab, spread)
    check_against_host(np.array(quirk_rows(t2i, 14), dtype=np.int64), vocab, spread)
    check_against_host(np.array([[t2i['union']]], dtype=np.int64), vocab, spread)          # T = 1
    got, earliest = check_against_host(synth.synth_programs(6, 1500, vocab), vocab, spread)   # more rows than one scan thread each
    if spread:      # what it is for: the largest launch shrinks, the many small ones grow
        assert got.max() < 0.6 * earliest.max() and (got[1:earliest.nonzero()[0].max()] >= 64).sum() > (earliest >= 64).sum()


def test_token_outside_the_vocabulary_raises_the_flag(vocab):
    t2i = vocab['program_token_to_idx']
    prog = synth.synth_programs(7, 9, vocab, max_len=12, max_nodes=9)
    good = prog.copy()
    prog[4, 0] = len(t2i)
    prog[6, 1] = -1
    sched, used = device_schedule(prog, vocab)
    assert int(sched.error[0].item()) == 1
    assert (used[[4, 6]] == 0).all()
    # the other rows are scheduled as if the bad ones were lone scene modules
    good[[4, 6]] = t2i['<NULL>']
    good[[4, 6], 0] = t2i['scene']
    plan = Plan(good, vocab)
    li = sched.level_info.cpu().numpy().reshape(-1, 2)
    want = np.bincount(plan.nodes()[plan.nodes()[:, 1] != t2i['scene'], 5], minlength=len(li))
    np.testing.assert_array_equal(li[:, 1], want)


def test_arity_table_the_node_records_cannot_hold_raises_the_flag(vocab):
    """A device arity table with 0 (or 3) for a token other than scene: rows that read that token set
    bit 2 of the flags and run as a lone scene module; the others are scheduled as ever."""
    from vqa_amd import ops
    t2i = vocab['program_token_to_idx']
    prog = synth.synth_programs(8, 12, vocab, max_len=12, max_nodes=9)
    prog[3, :3] = [t2i['count'], t2i['exist'], t2i['scene']]
    prog[[0, 1, 2] + list(range(4, 12))] = np.where(prog[[0, 1, 2] + list(range(4, 12))] == t2i['exist'], t2i['count'],
                                                     prog[[0, 1, 2] + list(range(4, 12))])
    for wrong in (0, 3):
        ar = np.asarray(V.arity_table(vocab), dtype=np.int8)
        ar[t2i['exist']] = wrong
        N, T = prog.shape
        buf = torch.full((ops.DeviceSchedule.words(N, T, len(ar)),), -7, dtype=torch.int32, device='cuda')
        sched = ops.DeviceSchedule(buf, N, T, len(ar))
        used = torch.full((N, T), -1.0, device='cuda')
        ops.devsched_build(torch.from_numpy(prog).cuda(), torch.from_numpy(ar).cuda(), used, sched, t2i['<NULL>'],
                           t2i['<START>'], t2i['scene'])
        torch.cuda.synchronize()
        assert int(sched.error[0].item()) == 4
        assert (used[3] == 0).all()
        good = prog.copy()
        good[3] = t2i['<NULL>']
        good[3, 0] = t2i['scene']
        plan = Plan(good, vocab)
        li = sched.level_info.cpu().numpy().reshape(-1, 2)
        want = np.bincount(plan.nodes()[plan.nodes()[:, 1] != t2i['scene'], 5], minlength=len(li))
        np.testing.assert_array_equal(li[:, 1], want)
        np.testing.assert_array_equal(np.delete(used.cpu().numpy(), 3, 0), np.delete(plan.used_fns(), 3, 0))


@pytest.mark.parametrize('balance', [False, True])      # of the host's levels and of the device's
def test_executor_gives_the_same_logits_with_the_schedule_made_on_the_device(vocab, balance):
    from vqa_amd.module_net import ModuleNet
    t2i = vocab['program_token_to_idx']
    T = 14
    prog = np.concatenate([np.array(quirk_rows(t2i, T), dtype=np.int64),
                           synth.synth_sampler_programs(11, 20, len(t2i), T),
                           synth.synth_programs(12, 40, vocab, max_len=T, max_nodes=12)])
    n = prog.shape[0]
    net = ModuleNet(False, vocab, **DIMS)
    net.load_state_dict(synth.synth_state_dict(13, vocab, **DIMS))
    net.cuda().eval()
    net.balance_levels = net.device_spread_levels = balance
    x = torch.from_numpy(synth.synth_features(14, n, DIMS['feature_dim'])).cuda()
    pt = torch.from_numpy(prog).cuda()
    with torch.no_grad():
        want = net(x, pt).clone()
        want_used = net.used_fns.clone()
        net.schedule_on_device = True
        got = net(x, pt).clone()
        assert net.last_token_counts is None and net.used_fns.is_cuda
        net.raise_for_schedule_error()
        assert torch.equal(got, want) and torch.equal(net.used_fns, want_used.to(net.used_fns.device))
        # other integer types and strides of the program tensor (Train_model.py hands int64 over)
        for other in (pt.to(torch.int32), pt.t().contiguous().t(), torch.stack([pt, pt], 2)[:, :, 0]):
            assert torch.equal(net(x, other), want) and torch.equal(net.used_fns, want_used)
        # a second call through the same workspace with fewer, other rows
        got2 = net(x[5:31], pt[5:31]).clone()
        net.schedule_on_device = False
        assert torch.equal(got2, net(x[5:31], pt[5:31]))
        # programs on the host, or a call that needs gradients, take the host scheduler
        net.schedule_on_device = True
        assert torch.equal(net(x, pt.cpu()), want) and net.last_token_counts is not None
    net.train()
    out = net(x[:4], pt[:4])
    out.sum().backward()
    assert net.stem[0].weight.grad is not None
    # an id outside the vocabulary cannot raise from the launch; the flag does on request
    bad = pt.clone()
    bad[3, 0] = len(t2i)
    with torch.no_grad():
        net(x, bad)
    with pytest.raises(KeyError):
        net.raise_for_schedule_error()


def test_sizes_beyond_the_limits_are_refused(vocab):
    """Programs too long for the scan's LDS histogram, or for the packed node records, raise with
    a message instead of launching; one row of one token works."""
    from vqa_amd import _lib
    t2i = vocab['program_token_to_idx']
    for T, msg in ((400, 'too long'), (1100, 'bad argument')):
        with pytest.raises(_lib.VqaError, match=msg):
            device_schedule(np.full((2, T), t2i['scene'], dtype=np.int64), vocab)
    sched, used = device_schedule(np.array([[t2i['scene']]], dtype=np.int64), vocab)
    assert used.tolist() == [[1.0]] and sched.level_info.cpu().numpy()[:, ].reshape(-1, 2)[:, 1].sum() == 0
    assert sched.cls_tiles.cpu().numpy()[0] == 2          # the scene output of image 0 (slot 2N + 0)


def test_candidate_scoring_with_the_schedule_made_on_the_device(vocab):
    """score_candidates (Train_model.py:339-347: K sampled programs per image) with the programs left
    on the device: the stem and the scene module once per image, the logits of the host path (which
    also shares common subtrees) bit for bit, on an untrained sampler's rows and the quirk rows; the
    reference's own call shape (an expand()-ed image) takes the same route; an image index out of
    range raises on request."""
    from vqa_amd.module_net import ModuleNet
    t2i = vocab['program_token_to_idx']
    K, T = 8, 14
    prog = np.concatenate([synth.synth_sampler_programs(62, 2 * K, len(t2i), T),
                           np.array(quirk_rows(t2i, T), dtype=np.int64),
                           synth.synth_programs(63, K, vocab, max_len=T, max_nodes=12)])
    M = prog.shape[0] // K
    img = torch.arange(M).repeat_interleave(K)
    x = torch.from_numpy(synth.synth_features(64, M, DIMS['feature_dim'])).cuda()
    net = ModuleNet(False, vocab, **DIMS)
    net.load_state_dict(synth.synth_state_dict(65, vocab, **DIMS))
    net.cuda().eval()
    pt = torch.from_numpy(prog).cuda()
    with torch.no_grad():
        want = net.score_candidates(x, pt, img).clone()
        want_used = net.used_fns.clone()
        full = net(x[img.cuda()], pt).clone()
        net.schedule_on_device = True
        for index in (img, img.cuda(), img.tolist()):
            got = net.score_candidates(x, pt, index)
            assert net.last_token_counts is None            # the device path ran
            net.raise_for_schedule_error()
            assert torch.equal(got, want) and torch.equal(got, full)
            assert torch.equal(net.used_fns, want_used.to(net.used_fns.device))
        one = net(x[1:2].expand(K, -1, -1, -1), pt[K:2 * K])
        assert net.last_token_counts is None and torch.equal(one, want[K:2 * K])
        bad = img.clone()
        bad[5] = M
        net.score_candidates(x, pt, bad)
        with pytest.raises(IndexError):
            net.raise_for_schedule_error()
