"""The CPU restatement (oracle/) against outputs captured from the real reference executor.

Fixtures under tests/golden/ were written by oracle/make_golden.py, which imports the reference.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle.module_net_ref import ModuleNetRef, parse_row
from vqa_amd import synth


def _cases(golden_dir):
    with open(os.path.join(golden_dir, 'schedule_quirks.json')) as f:
        return json.load(f)


def test_parse_matches_reference_hooks(vocab, golden_dir):
    g = _cases(golden_dir)
    i2t = vocab['program_idx_to_token']
    t2i = vocab['program_token_to_idx']
    assert len(g['cases']) >= 50
    for c in g['cases']:
        nodes, used = parse_row(c['program'], i2t)
        assert [t2i[n[0]] for n in nodes] == c['tokens'], c['name']
        # a scene node's input is the stem feature map (-1 in the hook trace)
        kids = [list(n[2]) if n[0] != 'scene' else [-1] for n in nodes]
        assert kids == c['children'], c['name']
        assert used == c['used_fns'], c['name']


def test_quirk_counts(vocab, golden_dir):
    by = {(c['name'], c['T']): c for c in _cases(golden_dir)['cases']}
    assert len(by[('all_null', 30)]['tokens']) == 31          # T NULL modules + implicit scene
    assert len(by[('all_binary', 30)]['tokens']) == 61        # 2T+1
    assert by[('all_null', 30)]['used_fns'] == [0] * 30
    assert len(by[('trailing_garbage', 12)]['tokens']) == 1   # root scene, rest never read


def test_error_behaviour(vocab, golden_dir):
    g = _cases(golden_dir)
    assert g['out_of_vocab_error'] == 'KeyError'
    with pytest.raises(KeyError):
        parse_row([99] + [0] * 11, vocab['program_idx_to_token'])


def test_tiny_full_logits_and_grads(vocab, golden_dir):
    d = np.load(os.path.join(golden_dir, 'tiny_full.npz'))
    sd = {k[2:]: torch.from_numpy(d[k]) for k in d.files if k.startswith('w/')}
    m = ModuleNetRef(vocab, sd)
    logits = m(torch.from_numpy(d['x']), torch.from_numpy(d['program']))
    loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(d['answers']))
    loss.backward()
    np.testing.assert_allclose(logits.detach().numpy(), d['logits'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(loss.item(), d['loss'], rtol=1e-6)
    np.testing.assert_array_equal(m.used_fns.numpy(), d['used_fns'])
    gold = {k[2:] for k in d.files if k.startswith('g/')}
    for k, g in m.grads().items():
        assert (g is not None) == (k in gold), k          # unused tokens keep grad None
        if g is not None:
            ref = d['g/' + k]
            assert np.abs(g.numpy() - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), k


def test_full_default_logits_and_grads(vocab, golden_dir):
    d = np.load(os.path.join(golden_dir, 'full_default.npz'))
    seed = int(d['seed'])
    n = d['program'].shape[0]
    x = synth.synth_features(seed, n)
    assert abs(float(x.astype(np.float64).sum()) - float(d['x_checksum'])) < 1e-6 * float(d['x_checksum'])
    m = ModuleNetRef(vocab, synth.synth_state_dict(seed, vocab))
    logits = m(torch.from_numpy(x), torch.from_numpy(d['program']))
    loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(d['answers']))
    loss.backward()
    ref = d['logits']
    assert np.abs(logits.detach().numpy() - ref).max() <= 1e-4 * np.abs(ref).max()
    np.testing.assert_allclose(loss.item(), d['loss'], rtol=1e-5)
    np.testing.assert_array_equal(m.used_fns.numpy(), d['used_fns'])
    from oracle.make_golden_util import sample_slice
    for k, g in m.grads().items():
        if ('gnorm/' + k) not in d.files:
            assert g is None, k
            continue
        assert abs(g.double().norm().item() - float(d['gnorm/' + k])) <= 1e-4 * float(d['gnorm/' + k]), k
        sl = d['gslice/' + k]
        assert np.abs(sample_slice(g) - sl).max() <= 1e-4 * max(np.abs(sl).max(), 1e-6), k
