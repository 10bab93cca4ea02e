"""Generate tests/golden/* by running the REAL reference executor.  TEST INFRASTRUCTURE ONLY.

Runs only where /root/reference exists (the build container); the fixtures it writes are data
(inputs and the reference's outputs), committed so that the GPU box, which has no reference, can
check against them.  Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

Three fixtures:
  schedule_quirks.json  execution order (token, children) and used_fns of quirk rows, captured
                        with forward hooks on the reference's modules
  tiny_full.npz         a small-width model: weights, inputs, logits, loss, every gradient
  full_default.npz      the default 117 M-parameter model on 4 samples: inputs by seed, logits,
                        loss, per-parameter gradient norms and sampled gradient slices
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')
sys.dont_write_bytecode = True

from Module_Model import ModuleNet  # noqa: E402  (the reference)

from vqa_amd import synth, vocab as V  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')


def trace_reference(model, x, program):
    """Run the reference forward with hooks; per executed module: token, child node ids."""
    events = []
    owner = {}
    keep = []

    def make_hook(tok):
        def hook(mod, inputs, output):
            kids = [owner.get(id(t), -1) for t in inputs]
            owner[id(output)] = len(events)
            keep.append(output)
            events.append((tok, kids))
        return hook

    handles = []
    for tok in model.vocab['program_token_to_idx']:
        handles.append(model._modules[tok].register_forward_hook(make_hook(tok)))
    out = model(x, program)
    for h in handles:
        h.remove()
    return out, events


def quirk_rows(t2i, T):
    sc, cnt, un, eq = t2i['scene'], t2i['count'], t2i['union'], t2i['equal_color']
    fl, rel = t2i['filter_color[red]'], t2i['relate[left]']
    N_, S_, E_, U_ = V.NULL, V.START, V.END, V.UNK

    def pad(toks):
        return (list(toks) + [N_] * T)[:T]

    return {
        'all_null': pad([]),
        'start_prefixed': pad([S_, cnt, fl, sc, E_]),
        'sampled_wellformed': pad([cnt, fl, rel, sc, E_]),
        'unary_then_end': pad([cnt, E_]),
        'all_binary': [un] * T,
        'trailing_garbage': pad([sc, cnt, fl, un, E_]),
        'unk_middle': pad([cnt, U_, fl, sc, E_]),
        'start_middle': pad([cnt, S_, S_, fl, sc, E_]),
        'binary_tree': pad([eq, cnt, fl, sc, cnt, rel, fl, sc, E_]),
        'binary_missing_arg': pad([N_] * (T - 2) + [un, sc]),
        'binary_both_missing': pad([N_] * (T - 1) + [un]),
        'start_last': pad([N_] * (T - 1) + [S_]),
        'end_first': pad([E_]),
        'nested_binary': pad([un, eq, sc, sc, un, sc, fl, sc, E_]),
    }


def small_kwargs():
    return dict(feature_dim=(32, 14, 14), module_dim=16, classifier_proj_dim=16,
                classifier_fc_layers=(32,))


def make_schedule_quirks(vocab):
    t2i = vocab['program_token_to_idx']
    model = ModuleNet(False, vocab, **small_kwargs())
    cases = []
    for T in (12, 30):
        rows = quirk_rows(t2i, T)
        names = sorted(rows)
        prog = torch.tensor([rows[n] for n in names], dtype=torch.int64)
        x = torch.zeros(len(names), 32, 14, 14)
        for i, n in enumerate(names):
            with torch.no_grad():
                _, ev = trace_reference(model, x[i:i + 1], prog[i:i + 1])
            cases.append({'name': n, 'T': T, 'program': rows[n],
                          'tokens': [t2i[tok] for tok, _ in ev],
                          'children': [k for _, k in ev],
                          'used_fns': [int(v) for v in model.used_fns[0].tolist()]})
    # sampler-style random rows (mostly ill-formed)
    prog = synth.synth_sampler_programs(11, 24, vocab_size=len(t2i), max_len=30)
    for i in range(prog.shape[0]):
        with torch.no_grad():
            _, ev = trace_reference(model, torch.zeros(1, 32, 14, 14),
                                    torch.from_numpy(prog[i:i + 1]))
        cases.append({'name': 'sampler_%d' % i, 'T': 30, 'program': prog[i].tolist(),
                      'tokens': [t2i[tok] for tok, _ in ev], 'children': [k for _, k in ev],
                      'used_fns': [int(v) for v in model.used_fns[0].tolist()]})
    # error behaviour: token id outside the vocab
    err = None
    try:
        model(torch.zeros(1, 32, 14, 14), torch.tensor([[len(t2i) + 3] + [0] * 11]))
    except Exception as e:  # noqa: BLE001
        err = type(e).__name__
    err2 = None
    try:
        model(torch.zeros(1, 32, 14, 14), torch.zeros(12, dtype=torch.int64))
    except Exception as e:  # noqa: BLE001
        err2 = type(e).__name__ + ':' + str(e)
    with open(os.path.join(GOLD, 'schedule_quirks.json'), 'w') as f:
        json.dump({'cases': cases, 'out_of_vocab_error': err, 'bad_program_error': err2}, f)
    print('schedule_quirks: %d cases, oov error %s, bad program %s' % (len(cases), err, err2))


def run_fwd_bwd(model, x, prog, ans):
    model.zero_grad()
    logits = model(x, prog)
    loss = torch.nn.CrossEntropyLoss()(logits, ans)      # Train_model.py:185,312
    loss.backward()
    grads = {k: (p.grad.detach().clone() if p.grad is not None else None)
             for k, p in model.named_parameters()}
    return logits.detach(), loss.detach(), grads, model.used_fns.clone()


def make_tiny_full(vocab):
    t2i = vocab['program_token_to_idx']
    dims = small_kwargs()
    sd = synth.synth_state_dict(3, vocab, **dims)
    # widen the init a little so that gradients through deep trees are not vanishing
    model = ModuleNet(False, vocab, **dims)
    assert list(model.state_dict().keys()) == list(sd.keys())
    model.load_state_dict(sd)
    rows = quirk_rows(t2i, 12)
    names = ['binary_tree', 'start_prefixed', 'all_null', 'unk_middle', 'nested_binary',
             'binary_missing_arg', 'trailing_garbage']
    prog = np.array([rows[n] for n in names], dtype=np.int64)
    extra = synth.synth_programs(5, 5, vocab, max_len=12, max_nodes=10)
    prog = np.concatenate([prog, extra], 0)
    n = prog.shape[0]
    x = synth.synth_features(21, n, dims['feature_dim'])
    ans = synth.synth_answers(21, n, len(vocab['answer_idx_to_token']))
    logits, loss, grads, used = run_fwd_bwd(model, torch.from_numpy(x), torch.from_numpy(prog),
                                            torch.from_numpy(ans))
    out = {'x': x, 'program': prog, 'answers': ans, 'logits': logits.numpy(),
           'loss': loss.numpy(), 'used_fns': used.numpy()}
    for k, v in sd.items():
        out['w/' + k] = v.numpy()
    for k, g in grads.items():
        if g is not None:
            out['g/' + k] = g.numpy()
    np.savez_compressed(os.path.join(GOLD, 'tiny_full.npz'), **out)
    print('tiny_full: n=%d loss=%.6f grads=%d/%d' % (n, float(loss),
          sum(g is not None for g in grads.values()), len(grads)))


from oracle.make_golden_util import sample_slice  # noqa: E402


def make_full_default(vocab):
    seed, n = 17, 4
    sd = synth.synth_state_dict(seed, vocab)
    model = ModuleNet(False, vocab)
    assert list(model.state_dict().keys()) == list(sd.keys())
    model.load_state_dict(sd)
    assert sum(p.numel() for p in model.parameters()) == 117471136
    prog = synth.synth_programs(seed, n, vocab, max_len=30, max_nodes=27)
    prog[3] = synth.synth_sampler_programs(seed, 1, len(vocab['program_token_to_idx']), 30)[0]
    x = synth.synth_features(seed, n)
    ans = synth.synth_answers(seed, n, 32)
    logits, loss, grads, used = run_fwd_bwd(model, torch.from_numpy(x), torch.from_numpy(prog),
                                            torch.from_numpy(ans))
    out = {'seed': np.int64(seed), 'program': prog, 'answers': ans, 'logits': logits.numpy(),
           'loss': loss.numpy(), 'used_fns': used.numpy(),
           'x_checksum': np.float64(x.astype(np.float64).sum())}
    for k, g in grads.items():
        if g is not None:
            out['gnorm/' + k] = np.float64(g.double().norm().item())
            out['gslice/' + k] = sample_slice(g)
    np.savez_compressed(os.path.join(GOLD, 'full_default.npz'), **out)
    print('full_default: loss=%.6f logits[0,:4]=%s' % (float(loss), logits[0, :4].tolist()))


if __name__ == '__main__':
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    vocab = V.make_vocab()
    assert len(vocab['program_token_to_idx']) == 44 and len(vocab['answer_idx_to_token']) == 32
    make_schedule_quirks(vocab)
    make_tiny_full(vocab)
    make_full_default(vocab)
