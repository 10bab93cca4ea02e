"""The native scheduler (host code in libvqa_amd.so) against the oracle's restatement of the
reference interpreter, bit-exact: node order, tokens, children, columns, used_fns, level buckets."""
import json
import os

import numpy as np
import pytest

from oracle.module_net_ref import parse_row
from vqa_amd import synth
from vqa_amd.schedule import Plan


def oracle_nodes(prog, vocab):
    """Global node table the way the scheduler numbers it, from the oracle's per-row parse."""
    i2t, t2i = vocab['program_idx_to_token'], vocab['program_token_to_idx']
    rows, used, roots = [], [], []
    for i in range(prog.shape[0]):
        nodes, u = parse_row(prog[i].tolist(), i2t)
        base = len(rows)
        lev = []
        for tok, col, kids in nodes:
            lv = 0 if not kids else 1 + max(lev[k] for k in kids)
            lev.append(lv)
            k = [base + c for c in kids] + [-1, -1]
            rows.append([i, t2i[tok], col, k[0], k[1], lv])
        roots.append(len(rows) - 1)
        used.append(u)
    return np.array(rows, dtype=np.int32), np.array(used, dtype=np.float32), np.array(roots, dtype=np.int32)


def check_against_oracle(prog, vocab):
    plan = Plan(prog, vocab)
    nodes, used, roots = oracle_nodes(prog, vocab)
    np.testing.assert_array_equal(plan.nodes(), nodes)
    np.testing.assert_array_equal(plan.used_fns(), used)
    np.testing.assert_array_equal(plan.roots(), roots)
    assert plan.info.num_nodes == len(nodes)
    assert plan.info.num_levels == nodes[:, 5].max() + 1
    # launch lists: every node but the repeated scene nodes of a sample (all scene nodes of a sample
    # compute the same map: the first one runs, the others share its output slots)
    scene_tok = vocab['program_token_to_idx']['scene']
    first = {}
    for v in range(len(nodes)):
        if nodes[v, 1] == scene_tok:
            first.setdefault(int(nodes[v, 0]), v)
    runs = [v for v in range(len(nodes)) if nodes[v, 1] != scene_tok or first[int(nodes[v, 0])] == v]
    seen = []
    for L in range(plan.info.num_levels):
        lv = plan.level_nodes(L)
        want = sorted([v for v in runs if nodes[v, 5] == L], key=lambda v: (nodes[v, 1], v))
        assert lv.tolist() == want
        seen += lv.tolist()
    assert sorted(seen) == runs and plan.info.num_exec_nodes == len(runs)
    for v in range(len(nodes)):
        if nodes[v, 1] == scene_tok:
            assert plan.slot('Y', v) == plan.slot('Y', first[int(nodes[v, 0])])
            assert plan.slot('H', v) == plan.slot('H', first[int(nodes[v, 0])])
    np.testing.assert_array_equal(plan.token_counts(), np.bincount(nodes[:, 1], minlength=len(vocab['program_token_to_idx'])))
    scene = vocab['program_token_to_idx']['scene']
    assert plan.info.num_scene == int((nodes[:, 1] == scene).sum())
    assert plan.info.num_binary == int((nodes[:, 4] >= 0).sum())
    assert plan.info.num_unary == len(nodes) - plan.info.num_scene - plan.info.num_binary
    return plan, nodes


def test_quirk_rows_match_reference_trace(vocab, golden_dir):
    """Directly against the hook trace captured from the reference executor."""
    with open(os.path.join(golden_dir, 'schedule_quirks.json')) as f:
        cases = json.load(f)['cases']
    scene = vocab['program_token_to_idx']['scene']
    for c in cases:
        plan = Plan(np.array([c['program']], dtype=np.int64), vocab)
        nd = plan.nodes()
        assert nd[:, 1].tolist() == c['tokens'], c['name']
        kids = [[-1] if t == scene else [k for k in (a, b) if k >= 0] for t, a, b in nd[:, [1, 3, 4]].tolist()]
        assert kids == c['children'], c['name']
        assert plan.used_fns()[0].astype(int).tolist() == c['used_fns'], c['name']


def test_wellformed_batch(vocab):
    check_against_oracle(synth.synth_programs(1, 257, vocab), vocab)


def test_sampler_batch_ragged_T(vocab):
    check_against_oracle(synth.synth_sampler_programs(2, 300, 44, 30), vocab)
    check_against_oracle(synth.synth_sampler_programs(3, 64, 44, 7), vocab)
    check_against_oracle(synth.synth_programs(4, 32, vocab, max_len=64, max_nodes=60), vocab)
    check_against_oracle(synth.synth_programs(5, 16, vocab, with_start=True), vocab)


def test_worst_case_sizes(vocab):
    t2i = vocab['program_token_to_idx']
    T = 30
    prog = np.array([[t2i['union']] * T, [0] * T, [t2i['scene']] + [0] * (T - 1)], dtype=np.int64)
    plan, nodes = check_against_oracle(prog, vocab)
    assert (nodes[:, 0] == 0).sum() == 2 * T + 1
    assert (nodes[:, 0] == 1).sum() == T + 1
    assert (nodes[:, 0] == 2).sum() == 1
    assert plan.info.slots_fwd == 4 * 3 + 2 * len(nodes) + T


def test_errors(vocab):
    with pytest.raises(KeyError):
        Plan(np.array([[44, 0, 0]], dtype=np.int64), vocab)
    with pytest.raises(KeyError):
        Plan(np.array([[-1, 0, 0]], dtype=np.int64), vocab)
    with pytest.raises(ValueError):
        Plan(np.zeros(5, dtype=np.int64), vocab)


def test_blob_sections_and_slots(vocab):
    prog = synth.synth_programs(9, 40, vocab)
    plan, nodes = check_against_oracle(prog, vocab)
    blob = plan.blob()
    n, nv = 40, len(nodes)
    off, cnt = plan.section('CLS_FWD_TILES')
    tiles = blob[off:off + cnt * 8].reshape(cnt, 8)
    assert cnt == n
    np.testing.assert_array_equal(tiles[:, 0], [plan.slot('Y', r) for r in plan.roots()])     # Y(root) gathered
    np.testing.assert_array_equal(tiles[:, 2], np.arange(n))
    off, cnt = plan.section('SCENE_OFFSETS')
    offs = blob[off:off + cnt]
    scene = vocab['program_token_to_idx']['scene']
    per_sample = np.bincount(nodes[nodes[:, 1] == scene, 0], minlength=n)
    np.testing.assert_array_equal(np.diff(offs), per_sample)
    off, cnt = plan.section('WG_ITEMS')
    items = blob[off:off + cnt * 4].reshape(cnt, 4)
    assert items[:, 2].sum() == plan.info.num_exec_nodes and items[:, 2].max() <= 16
    assert plan.slot('F', 3) == 3 and plan.slot('S0', 3) == n + 3
    assert plan.slot('Y', 0) == 2 * n + nodes[0, 0] and plan.slot('H', 0) == 3 * n + nodes[0, 0]   # node 0 is a scene node
    nonscene = int(np.nonzero(nodes[:, 1] != vocab['program_token_to_idx']['scene'])[0][0])
    assert plan.slot('Y', nonscene) == 4 * n + nonscene and plan.slot('H', nonscene) == 4 * n + nv + nonscene
    assert plan.info.slots_total > plan.info.slots_fwd


def test_flops_formula(vocab):
    prog = synth.synth_programs(1, 64, vocab)
    plan = Plan(prog, vocab)
    fwd, both = plan.flops()
    i = plan.info
    fixed = 462422016 + 57802752 + 25690112 + 205520896 + 65536
    assert fwd == 64 * fixed + (i.num_scene + i.num_unary) * 115605504 + i.num_binary * 128450560
    assert both == 3 * fwd - 64 * 462422016


def test_balanced_levels_are_a_valid_schedule(vocab):
    """VQA_PLAN_BALANCE only moves nodes with slack to later levels: same nodes, ids and level
    count; every child still runs in an earlier level than its parent; levels come in whole or half
    rounds of 512 tiles except where nodes were due."""
    from vqa_amd import _lib
    prog = synth.synth_programs(77, 300, vocab)
    p0 = Plan(prog, vocab)
    p1 = Plan(prog, vocab, flags=_lib.PLAN_BALANCE)
    n0, n1 = p0.nodes(), p1.nodes()
    np.testing.assert_array_equal(n0[:, :5], n1[:, :5])
    assert p0.info.num_levels == p1.info.num_levels
    lv = n1[:, 5]
    assert (lv >= n0[:, 5]).all()                       # never earlier than the earliest level
    for v in range(len(n1)):
        for c in n1[v, 3:5]:
            if c >= 0:
                assert lv[c] < lv[v]
    sizes = np.bincount(lv, minlength=p1.info.num_levels)
    assert sizes.sum() == len(n1) and (sizes > 0).all()
    assert sizes[1:].max() < np.bincount(n0[:, 5])[1:].max()
    for L in range(p1.info.num_levels):
        ids = p1.level_nodes(L)
        if L == 0:      # scene nodes stay in level 0, one launch entry per sample
            assert (n1[ids, 1] == vocab['program_token_to_idx']['scene']).all()
            assert sorted(n1[ids, 0].tolist()) == sorted(set(n1[lv == 0, 0].tolist()))
        else:
            assert sorted(ids.tolist()) == np.nonzero(lv == L)[0].tolist()
        toks = n1[ids, 1]
        assert (np.diff(toks) >= 0).all()               # still grouped by token inside a level
    np.testing.assert_array_equal(p0.used_fns(), p1.used_fns())


def test_shared_image_plan_interns_common_subtrees(vocab):
    """vqa_schedule_build_shared: programs of one image share nodes with equal (token, inputs);
    used_fns stay per program; roots of identical programs coincide; programs of different images
    share nothing."""
    prog = synth.synth_programs(5, 12, vocab, max_len=14, max_nodes=10)
    prog[1] = prog[0]
    img = np.array([0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2])
    p0, p1 = Plan(prog, vocab), Plan(prog, vocab, image_index=img)
    np.testing.assert_array_equal(p0.used_fns(), p1.used_fns())
    n1, r1 = p1.nodes(), p1.roots()
    assert p1.info.num_images == 3 and p1.info.slots_total == p1.info.slots_fwd
    assert r1[0] == r1[1]
    assert len(n1) < len(p0.nodes())
    # no two nodes describe the same computation; children precede parents; one scene node per image at most
    keys = set()
    scene = vocab['program_token_to_idx']['scene']
    for v, (image, tok, _col, c0, c1, lvl) in enumerate(n1):
        assert (image, tok, c0, c1) not in keys
        keys.add((image, tok, c0, c1))
        for c in (c0, c1):
            if c >= 0:
                assert c < v and n1[c, 0] == image and n1[c, 5] < lvl
    assert sum(1 for r in n1 if r[1] == scene) <= 3
    # expanding the shared plan per program gives back every program's own tree
    n0, r0 = p0.nodes(), p0.roots()

    def tree(nodes, v):
        return (int(nodes[v, 1]),) + tuple(tree(nodes, c) for c in nodes[v, 3:5] if c >= 0)
    for i in range(len(prog)):
        assert tree(n0, r0[i]) == tree(n1, r1[i])
    with pytest.raises(ValueError):
        Plan(prog, vocab, image_index=img[:5])


def test_deferred_backward_part_gives_the_same_blob(vocab):
    """VQA_PLAN_DEFER_BACKWARD: the build returns with the forward part of the blob only;
    vqa_plan_finish appends the rest.  Same blob, same sections as a one-step build."""
    from vqa_amd import _lib
    prog = synth.synth_programs(31, 200, vocab)
    for base in (_lib.PLAN_SKIP_SEPARATE | _lib.PLAN_BALANCE, _lib.PLAN_SKIP_FUSED, 0):
        whole = Plan(prog, vocab, flags=base)
        part = Plan(prog, vocab, flags=base | _lib.PLAN_DEFER_BACKWARD)
        nf = part.info.blob_fwd_words
        assert part.info.blob_words == nf == whole.info.blob_fwd_words < whole.info.blob_words
        assert whole.info.blob_words <= part.info.blob_words_bound
        assert part.section('CLS_FWD_TILES') == whole.section('CLS_FWD_TILES')
        assert part.section('WG_ITEMS')[1] == 0 and part.section('SLOTS_DF')[1] == 0
        np.testing.assert_array_equal(part.blob(), whole.blob()[:nf])
        part.finish()
        part.finish()       # idempotent
        assert part.info.blob_words == whole.info.blob_words
        assert part.info.wgrad_partial_blocks == whole.info.wgrad_partial_blocks
        np.testing.assert_array_equal(part.blob(), whole.blob())
        for name in _lib.SEC:
            assert part.section(name) == whole.section(name)
        import torch
        buf = torch.zeros(whole.info.blob_words, dtype=torch.int32)
        part.blob_range_into(buf, nf, whole.info.blob_words - nf)
        np.testing.assert_array_equal(buf.numpy()[nf:], whole.blob()[nf:])
        assert (buf.numpy()[:nf] == 0).all()


def test_blob_written_into_the_callers_memory(vocab):
    """vqa_schedule_build_into: a store that holds the blob's upper bound receives the blob itself
    (forward part, then what vqa_plan_finish appends), word for word what a plan with memory of its
    own holds; a store that is too small is left alone.  Plans are pooled, so one that lent memory
    must come back clean."""
    import torch
    from vqa_amd import _lib
    prog = synth.synth_programs(77, 150, vocab)
    n, t = prog.shape
    flags = _lib.PLAN_SKIP_SEPARATE | _lib.PLAN_BALANCE | _lib.PLAN_DEFER_BACKWARD
    own = Plan(prog, vocab, flags=flags)
    words = Plan.blob_store_words(n, t, len(vocab['program_token_to_idx']))
    assert words >= own.info.blob_words_bound
    store = torch.full((words,), -7, dtype=torch.int32)
    lent = Plan(prog, vocab, flags=flags, blob_store=store)
    assert lent.info.blob_lent == 1 and own.info.blob_lent == 0
    nf = lent.info.blob_words
    np.testing.assert_array_equal(store[:nf].numpy(), own.blob())
    assert (store[nf:] == -7).all()
    own.finish(); lent.finish()
    total = lent.info.blob_words
    assert total == own.info.blob_words > nf
    np.testing.assert_array_equal(store[:total].numpy(), own.blob())
    np.testing.assert_array_equal(lent.blob(), own.blob())
    assert (store[total:] == -7).all()
    for name in _lib.SEC:
        assert lent.section(name) == own.section(name)
    del lent
    small = torch.full((own.info.blob_words_bound - 1,), -7, dtype=torch.int32)
    p = Plan(prog, vocab, flags=flags, blob_store=small)     # most likely the pooled object that lent
    assert p.info.blob_lent == 0 and (small == -7).all()
    p.finish()
    np.testing.assert_array_equal(p.blob(), own.blob())
    # samples that share images
    img = np.arange(n) // 3
    a = Plan(prog, vocab, flags=_lib.PLAN_SKIP_SEPARATE, image_index=img)
    b = Plan(prog, vocab, flags=_lib.PLAN_SKIP_SEPARATE, image_index=img, blob_store=store)
    assert b.info.blob_lent == 1
    np.testing.assert_array_equal(store[:b.info.blob_words].numpy(), a.blob())
