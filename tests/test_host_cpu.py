"""Host-side checks that need no GPU: the C-ABI library loads and exports every declared symbol,
the drop-in module keeps the reference's constructor / state-dict surface, and it refuses to run
without the HIP path."""
import os
import re

import pytest
import torch

from vqa_amd import _lib, synth


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    hdr = open(os.path.join(os.path.dirname(_lib._HERE), 'include', 'vqa_amd.h')).read()
    declared = set(re.findall(r'\b(vqa_[a-z0-9_]+)\s*\(', hdr))
    assert declared, 'no declarations found'
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.vqa_version() >= 100


def test_state_dict_keys_match_reference_order(vocab):
    from vqa_amd.module_net import ModuleNet
    net = ModuleNet(False, vocab)
    want = synth.state_dict_shapes(vocab)          # asserted equal to the reference in make_golden.py
    got = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    assert got == [(k, tuple(s)) for k, s in want]
    assert sum(p.numel() for p in net.parameters()) == 117471136
    net.load_state_dict({k: torch.zeros(s) for k, s in want})
    # reference kwargs round trip (Functions.py:107-119)
    ModuleNet(False, vocab=vocab, feature_dim=(1024, 14, 14), stem_batchnorm=False, stem_num_layers=2,
              module_dim=128, module_residual=True, module_batchnorm=False, classifier_proj_dim=512,
              classifier_downsample='maxpool2', classifier_fc_layers=(1024,), classifier_batchnorm=False,
              classifier_dropout=0)


def test_unsupported_options_refuse(vocab):
    from vqa_amd.module_net import ModuleNet
    for kw in (dict(module_batchnorm=True), dict(stem_batchnorm=True), dict(classifier_batchnorm=True),
               dict(module_residual=False), dict(classifier_dropout=0.5), dict(classifier_downsample='maxpool4'),
               dict(module_dim=100), dict(feature_dim=(1024, 7, 7))):
        with pytest.raises(NotImplementedError):
            ModuleNet(False, vocab, **kw)


def test_no_cpu_fallback(vocab):
    from vqa_amd.module_net import ModuleNet
    net = ModuleNet(False, vocab, feature_dim=(128, 14, 14), classifier_proj_dim=128, classifier_fc_layers=(128,))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        net(torch.zeros(1, 128, 14, 14), torch.zeros(1, 4, dtype=torch.int64))
    with pytest.raises(ValueError):
        net(torch.zeros(1, 128, 14, 14), torch.zeros(1, 4, 2, dtype=torch.int64))
    with pytest.raises(AttributeError):
        net(torch.zeros(1, 128, 14, 14), [[{'function': 'scene', 'inputs': [], 'value_inputs': []}]])


def test_reference_checkpoint_round_trip(vocab, tmp_path):
    """A checkpoint in the reference's layout (Functions.py:93-101 reads
    `execution_engine_kwargs` / `execution_engine_state`) restores into the drop-in module, and
    what the module saves is what a reference ModuleNet would load: same keys, shapes, values."""
    from vqa_amd.module_net import ModuleNet
    kwargs = dict(vocab=vocab, feature_dim=(128, 14, 14), stem_batchnorm=False, stem_num_layers=2, module_dim=128,
                  module_residual=True, module_batchnorm=False, classifier_proj_dim=128,
                  classifier_downsample='maxpool2', classifier_fc_layers=(128,), classifier_batchnorm=False,
                  classifier_dropout=0)
    dims = {k: kwargs[k] for k in ('feature_dim', 'module_dim', 'classifier_proj_dim', 'classifier_fc_layers')}
    state = synth.synth_state_dict(3, vocab, **dims)
    path = str(tmp_path / 'ee.pt')
    torch.save({'execution_engine_kwargs': kwargs, 'execution_engine_state': state}, path)
    ck = torch.load(path, map_location=lambda storage, loc: storage, weights_only=True)
    net = ModuleNet(False, **ck['execution_engine_kwargs'])
    net.load_state_dict(ck['execution_engine_state'])
    out = net.state_dict()
    assert list(out) == list(state)
    for k in state:
        assert torch.equal(out[k], state[k]), k
    net.share_memory()                      # Train_model.py:224


def test_list_programs_become_the_prefix_rows_the_interpreter_runs():
    """Opt-in list-of-dicts programs (the format Module_Model_tmp.py:142-167 executes): converted rows,
    parsed by the oracle's interpreter, must run the same modules on the same inputs as the list says."""
    import pytest
    from oracle.module_net_ref import parse_row
    from vqa_amd import vocab as V
    vocab = V.make_vocab()
    i2t = vocab['program_idx_to_token']

    def fn(name, inputs, value=None):
        return {'function': name, 'value_inputs': [value] if value else [], 'inputs': inputs}
    prog = [fn('scene', []), fn('filter_color', [0], 'red'), fn('scene', []), fn('filter_shape', [2], 'cube'),
            fn('unique', [3]), fn('unreached', [0]), fn('union', [1, 4]), fn('count', [6])]
    chain = [fn('scene', []), fn('filter_size', [0], 'large'), fn('exist', [1])]
    rows = V.lists_to_prefix_rows([prog, chain], vocab)
    assert rows.shape == (2, 7) and rows.dtype.kind == 'i'       # the unreached entry is dropped
    toks = [i2t[int(t)] for t in rows[0]]
    assert toks == ['count', 'union', 'filter_color[red]', 'scene', 'unique', 'filter_shape[cube]', 'scene']
    nodes, used = parse_row(rows[0].tolist(), i2t)
    # execution order (post-order) and wiring: union gets filter_color[red](scene) and unique(filter_shape[cube](scene))
    assert [n[0] for n in nodes] == ['scene', 'filter_color[red]', 'scene', 'filter_shape[cube]', 'unique', 'union', 'count']
    assert nodes[5][2] == (1, 4) and nodes[6][2] == (5,) and used == [1] * 7
    assert [i2t[int(t)] for t in rows[1]][:4] == ['exist', 'filter_size[large]', 'scene', '<NULL>']
    with pytest.raises(KeyError):
        V.lists_to_prefix_rows([[fn('scene', []), fn('no_such_function', [0])]], vocab)
    with pytest.raises(TypeError):
        V.lists_to_prefix_rows([[fn('scene', []), fn('union', [0])]], vocab)
