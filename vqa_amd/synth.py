"""Seeded synthetic inputs of CLEVR shape: programs, features, weights.

Everything here is a pure function of its seed and is computed the same way on every box
(integer hashing in numpy, no library RNG streams), so a golden fixture made in one place can be
re-derived in another from the seed alone.
"""
import numpy as np

from . import vocab as V

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def counter_uniform(seed, n, lo=-1.0, hi=1.0):
    """n float32 values in [lo, hi): value k is a hash of (seed, k) only."""
    with np.errstate(over='ignore'):
        base = _splitmix64(np.uint64(seed) * np.uint64(0x2545F4914F6CDD1D) + np.uint64(0x1234567))
        idx = np.arange(n, dtype=np.uint64)
        z = _splitmix64(idx ^ base)
    u = (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))
    return (u * np.float32(hi - lo) + np.float32(lo)).astype(np.float32)


def counter_normal(seed, n):
    """n float32 standard normals (Box-Muller over two counter streams)."""
    u1 = counter_uniform(seed * 2 + 1, n, 0.0, 1.0).astype(np.float64)
    u2 = counter_uniform(seed * 2 + 2, n, 0.0, 1.0).astype(np.float64)
    r = np.sqrt(-2.0 * np.log(1.0 - u1))
    return (r * np.cos(2.0 * np.pi * u2)).astype(np.float32)


def synth_features(seed, n, feature_dim=(1024, 14, 14)):
    """[n, C, H, W] float32: N(0,1) clipped at 0, like post-ReLU ResNet conv4 features."""
    c, h, w = feature_dim
    x = counter_normal(seed, n * c * h * w)
    return np.maximum(x, 0.0).reshape(n, c, h, w)


def _gen_tree(rng, vocab_lists, budget, p_scene, p_binary):
    """Prefix token list of a random well-formed tree with at most `budget` nodes."""
    unary, binary, scene = vocab_lists
    if budget <= 1:
        return [scene]
    r = rng.random()
    if r < p_scene:
        return [scene]
    if r < p_scene + p_binary and budget >= 3:
        tok = binary[int(rng.integers(len(binary)))]
        left_budget = 1 + int(rng.integers(budget - 2))      # 1 .. budget-2
        left = _gen_tree(rng, vocab_lists, left_budget, p_scene, p_binary)
        right = _gen_tree(rng, vocab_lists, budget - 1 - len(left), p_scene, p_binary)
        return [tok] + left + right
    tok = unary[int(rng.integers(len(unary)))]
    return [tok] + _gen_tree(rng, vocab_lists, budget - 1, p_scene, p_binary)


def synth_programs(seed, n, vocab=None, max_len=30, max_nodes=27, p_scene=0.15, p_binary=0.15,
                   with_start=False):
    """[n, max_len] int64 prefix programs: well-formed trees, `<END>`, then `<NULL>` padding.

    This is the sampled-program format of `LSTM_Model.py:205-235` (no `<START>`); with
    `with_start` a `<START>` is put in column 0 as the ground-truth format of
    `Preprocess_questions.py:130-148` has it.
    """
    vocab = vocab or V.make_vocab()
    t2i = vocab['program_token_to_idx']
    unary = [i for t, i in t2i.items() if V.get_num_inputs(t) == 1 and t not in V.SPECIAL_TOKENS]
    binary = [i for t, i in t2i.items() if V.get_num_inputs(t) == 2]
    lists = (sorted(unary), sorted(binary), t2i['scene'])
    rng = np.random.default_rng(seed)
    room = max_len - 1 - (1 if with_start else 0)
    out = np.zeros((n, max_len), dtype=np.int64)
    for i in range(n):
        toks = _gen_tree(rng, lists, min(max_nodes, room), p_scene, p_binary)
        row = ([V.START] if with_start else []) + toks + [V.END]
        out[i, :len(row)] = row
    return out


def synth_sampler_programs(seed, n, vocab_size=44, max_len=30):
    """Programs as an untrained sampler emits them: uniform tokens until `<END>`, then `<NULL>`.

    Rows are mostly ill-formed (specials in the middle, incomplete trees), which is what the
    interpreter's quirk paths (`Module_Model.py:198-208`) exist for.
    """
    rng = np.random.default_rng(seed)
    out = np.zeros((n, max_len), dtype=np.int64)
    for i in range(n):
        for t in range(max_len):
            tok = int(rng.integers(vocab_size))
            out[i, t] = tok
            if tok == V.END:
                break
    return out


def synth_answers(seed, n, num_answers=32):
    return np.random.default_rng(seed + 7919).integers(0, num_answers, size=n).astype(np.int64)


def state_dict_shapes(vocab, feature_dim=(1024, 14, 14), stem_num_layers=2, module_dim=128,
                      classifier_proj_dim=512, classifier_fc_layers=(1024,)):
    """Ordered (key, shape) list of the reference `ModuleNet.state_dict()` for the default
    no-batchnorm options (`Module_Model.py:96-150`): stem, classifier, then one module per token
    in vocab order."""
    cin, h, w = feature_dim
    c = module_dim
    out = []
    prev = cin
    for l in range(stem_num_layers):
        out += [('stem.%d.weight' % (2 * l), (c, prev, 3, 3)), ('stem.%d.bias' % (2 * l), (c,))]
        prev = c
    idx = 0
    prev = c * h * w
    if classifier_proj_dim:
        out += [('classifier.0.weight', (classifier_proj_dim, c, 1, 1)),
                ('classifier.0.bias', (classifier_proj_dim,))]
        idx = 2
        prev = classifier_proj_dim * h * w
    idx += 1  # Flatten
    for f in classifier_fc_layers:
        out += [('classifier.%d.weight' % idx, (f, prev)), ('classifier.%d.bias' % idx, (f,))]
        idx += 2
        prev = f
    na = len(vocab['answer_idx_to_token'])
    out += [('classifier.%d.weight' % idx, (na, prev)), ('classifier.%d.bias' % idx, (na,))]
    for tok in vocab['program_token_to_idx']:
        if V.get_num_inputs(tok) == 2:
            out += [('%s.proj.weight' % tok, (c, 2 * c, 1, 1)), ('%s.proj.bias' % tok, (c,))]
        out += [('%s.conv1.weight' % tok, (c, c, 3, 3)), ('%s.conv1.bias' % tok, (c,)),
                ('%s.conv2.weight' % tok, (c, c, 3, 3)), ('%s.conv2.bias' % tok, (c,))]
    return out


def synth_state_dict(seed, vocab, **dims):
    """Deterministic weights: key k (position p in the list above) comes from counter stream
    seed*1000+p.  Weights are uniform in +-sqrt(6/fan_in) (variance-preserving through ReLU), the
    second conv of every residual body at 0.3x that so deep trees neither vanish nor explode;
    biases are uniform in +-0.05.  With torch's default init the logits would be their bias to
    three digits and a parity check on them would test nothing."""
    import torch
    sd = {}
    for p, (key, shape) in enumerate(state_dict_shapes(vocab, **dims)):
        n = int(np.prod(shape))
        if key.endswith('.bias'):
            bound = 0.05
        else:
            bound = np.sqrt(6.0 / int(np.prod(shape[1:])))
            if key.endswith('.conv2.weight'):
                bound *= 0.3
        sd[key] = torch.from_numpy(counter_uniform(seed * 1000 + p, n, -bound, bound).reshape(shape))
    return sd
