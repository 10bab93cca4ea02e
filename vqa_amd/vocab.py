"""Synthetic CLEVR vocabulary for the module executor.

The reference ships no vocab.json, so the program vocabulary is rebuilt here the way the
reference's preprocessing builds it: four special tokens with fixed ids
(`Preprocess_funcs.py:24-28`, asserted in `Functions.py:41-46`) followed by the sorted CLEVR
function strings (`Preprocess_funcs.py:65-73`), 44 tokens in all (the count hard-coded in
`LSTM_Model.py:123`).  Answers follow the same rule (4 specials + 28 sorted CLEVR answers = 32).
"""

SPECIAL_TOKENS = ('<NULL>', '<START>', '<END>', '<UNK>')
NULL, START, END, UNK = 0, 1, 2, 3

_COLORS = ('blue', 'brown', 'cyan', 'gray', 'green', 'purple', 'red', 'yellow')
_MATERIALS = ('metal', 'rubber')
_SHAPES = ('cube', 'cylinder', 'sphere')
_SIZES = ('large', 'small')
_RELATIONS = ('behind', 'front', 'left', 'right')


def clevr_program_tokens():
    """The 40 CLEVR function strings in `function[value]` form (`Program_funcs.py:97-101`)."""
    toks = ['count', 'exist', 'greater_than', 'intersect', 'less_than', 'scene', 'union', 'unique']
    for attr in ('color', 'integer', 'material', 'shape', 'size'):
        toks.append('equal_%s' % attr)
    for attr in ('color', 'material', 'shape', 'size'):
        toks.append('query_%s' % attr)
        toks.append('same_%s' % attr)
    toks += ['filter_color[%s]' % v for v in _COLORS]
    toks += ['filter_material[%s]' % v for v in _MATERIALS]
    toks += ['filter_shape[%s]' % v for v in _SHAPES]
    toks += ['filter_size[%s]' % v for v in _SIZES]
    toks += ['relate[%s]' % v for v in _RELATIONS]
    return sorted(toks)


def clevr_answer_tokens():
    toks = [str(i) for i in range(11)] + ['yes', 'no']
    toks += list(_COLORS) + list(_MATERIALS) + list(_SHAPES) + list(_SIZES)
    return sorted(toks)


def get_num_inputs(token):
    """Arity of a program token; restates `Program_funcs.py:103-122`.

    `scene` is nullary; a name containing `equal`, or one of union / intersect / less_than /
    greater_than, is binary; everything else (the four special tokens included) is unary.
    """
    name = token.split('[')[0] if '[' in token else token
    if name == 'scene':
        return 0
    if 'equal' in name or name in ('union', 'intersect', 'less_than', 'greater_than'):
        return 2
    return 1


def make_vocab():
    """A vocab dict with the keys `ModuleNet` reads (`Module_Model.py:118,139,200`)."""
    prog = list(SPECIAL_TOKENS) + clevr_program_tokens()
    ans = list(SPECIAL_TOKENS) + clevr_answer_tokens()
    p2i = {t: i for i, t in enumerate(prog)}
    a2i = {t: i for i, t in enumerate(ans)}
    return {
        'program_token_to_idx': p2i,
        'program_idx_to_token': {i: t for t, i in p2i.items()},
        'answer_token_to_idx': a2i,
        'answer_idx_to_token': {i: t for t, i in a2i.items()},
    }


def arity_table(vocab):
    """int list indexed by token id; ids must be dense 0..V-1."""
    i2t = vocab['program_idx_to_token']
    n = len(i2t)
    return [get_num_inputs(i2t[i]) for i in range(n)]


def function_to_str(f):
    """`{'function', 'value_inputs'}` -> token string; restates `Program_funcs.py:97-101`."""
    value_str = ''
    if f['value_inputs']:
        value_str = '[%s]' % ','.join(f['value_inputs'])
    return '%s%s' % (f['function'], value_str)


def lists_to_prefix_rows(programs, vocab):
    """List-of-dicts programs (one list per sample; entry j = {'function', 'value_inputs', 'inputs':
    indices of earlier entries}; the last entry is the root - the format `forward_modules_json`
    of `Module_Model_tmp.py:142-167` executes) as prefix token-id rows, `<NULL>`-padded, the form
    the interpreter (`Module_Model.py:196-222`) runs: root first, then each input's subtree in
    order.  An entry used by two parents is repeated (same values, computed twice); entries the root
    does not reach contribute nothing to the output and are dropped.  A token outside the
    vocabulary raises KeyError (the reference's dict lookup); an entry whose number of inputs is not
    its module's arity raises TypeError (the reference's module call would)."""
    import numpy as np
    t2i = vocab['program_token_to_idx']
    rows = []
    for prog in programs:
        if len(prog) == 0:
            raise IndexError('list index out of range')       # module_outputs[-1] of nothing
        out = []

        def visit(j, depth=0):
            f = prog[j]
            tok = function_to_str(f)
            idx = t2i[tok]
            inputs = [] if tok == 'scene' else list(f['inputs'])
            if len(inputs) != get_num_inputs(tok):
                raise TypeError('%s takes %d inputs, %d given' % (tok, get_num_inputs(tok), len(inputs)))
            if any(i >= j or i < 0 for i in inputs):
                raise IndexError('list index out of range')   # inputs must be earlier entries
            out.append(idx)
            for i in inputs:
                visit(i, depth + 1)
        visit(len(prog) - 1)
        rows.append(out)
    T = max(len(r) for r in rows)
    arr = np.zeros((len(rows), T), dtype=np.int64)
    for i, r in enumerate(rows):
        arr[i, :len(r)] = r
    return arr
