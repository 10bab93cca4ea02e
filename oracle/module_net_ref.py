"""CPU restatement of the reference module executor.  TEST INFRASTRUCTURE ONLY.

This file is the checker for the HIP path: it is imported by tests/, by
`__graft_entry__.smoke()` and by the `cpu_baseline` leg of bench.py, and by nothing that ships.
It restates, in stock torch CPU ops and in the reference's own per-sample order, what
`/root/reference/Module_Model.py` and `Layers.py` compute with the default (no batch-norm,
residual) options.  It is pinned against the reference itself: tests/golden/ holds outputs of the
real `ModuleNet` captured by oracle/make_golden.py, and tests/test_oracle_golden.py checks this
restatement against them.

Each function names the reference lines it follows.
"""
import torch
import torch.nn.functional as F


def get_num_inputs(token):
    """`Program_funcs.py:103-122`: arity from the token string."""
    name = token.split('[')[0] if '[' in token else token
    if name == 'scene':
        return 0
    if 'equal' in name or name in ['union', 'intersect', 'less_than', 'greater_than']:
        return 2
    return 1


def parse_row(row, idx_to_token):
    """Interpret one prefix program row exactly as `forward_modules_ints_helper`
    (`Module_Model.py:196-222`) does, without running any module.

    Returns (nodes, used) where `nodes` is the list of executed modules in execution (post-)
    order, each a tuple (token_string, column_or_minus1, child_node_indices), and `used` is the
    row of `used_fns` (list of 0/1).  The last node is the root.
    """
    T = len(row)
    used = [0] * T
    nodes = []

    def helper(j):
        used_fn_j = True
        col = j
        if j < T:                                   # :198-200
            fn_str = idx_to_token[int(row[j])]      # KeyError for an id outside the vocab
        else:                                       # :201-203 reading past the row
            used_fn_j = False
            fn_str = 'scene'
            col = -1
        if fn_str == '<NULL>':                      # :204-205 not marked, still executed
            used_fn_j = False
        elif fn_str == '<START>':                   # :206-208 skipped
            return helper(j + 1)
        if used_fn_j:                               # :209-210
            used[j] = 1
        j += 1
        children = []
        if fn_str != 'scene':                       # :213-220
            while len(children) < get_num_inputs(fn_str):
                child, j = helper(j)
                children.append(child)
        nodes.append((fn_str, col, tuple(children)))  # :221 module runs here
        return len(nodes) - 1, j

    helper(0)
    return nodes, used


class _RoundAct(torch.autograd.Function):
    """Round an activation to bf16 as a bf16 executor stores it; the gradient that flows back
    through the stored value is rounded to bf16 as well."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


class _RoundGrad(torch.autograd.Function):
    """Identity forward; the incoming gradient is rounded to bf16."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


class _Force(torch.autograd.Function):
    """Stand-in for round(relu(z)) that outputs a GIVEN stored activation instead of the computed
    one and masks the (bf16-rounded) gradient by that stored value's sign.  With every stored
    point forced to the values another executor saved, this oracle's backward is the exact
    vector-Jacobian product at that executor's own activations, so the two backwards can be
    compared without the ReLU-mask flips that two independently rounded forwards always have."""

    @staticmethod
    def forward(ctx, z, forced, round_grad=True):
        ctx.save_for_backward(forced > 0)
        ctx.round_grad = round_grad
        return forced.to(z.dtype).clone()

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        if ctx.round_grad:
            g = g.to(torch.bfloat16).to(torch.float32)
        return g * mask, None, None


class _RoundWeight(torch.autograd.Function):
    """Use a weight at bf16 precision; its gradient stays fp32 (accumulated in fp32)."""

    @staticmethod
    def forward(ctx, w):
        return w.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


class _Params(dict):
    """state-dict keyed access that rounds `.weight` entries on read when storage is bf16."""

    def __init__(self, d, bf16):
        super().__init__(d)
        self.bf16 = bf16

    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        return _RoundWeight.apply(v) if (self.bf16 and k.endswith('.weight')) else v


class ModuleNetRef:
    """The reference executor on plain tensors keyed like its state dict.

    storage='bf16' keeps the reference's algorithm and fp32 arithmetic but rounds to bf16 at the
    points where a bf16 executor stores values: the input features, the weights, the output of
    every conv / linear (+bias, +residual, ReLU) except the logits, and the gradients flowing
    back through those stored values.  It is what the reference computes when its tensors are
    bf16 with fp32 accumulation; ReLU masks are then taken from the same rounded values as the
    HIP path's, which a comparison of gradients needs (a mask that flips on a near-zero
    pre-activation changes the gradient by O(1) in that unit, whatever the arithmetic precision).

    `forward` follows `ModuleNet.forward` (`Module_Model.py:235-248`): stem on the whole batch,
    per-sample interpretation with batch-1 module calls, `torch.cat` of the roots, classifier on
    the whole batch.  Parameters are leaf tensors with `requires_grad`, so autograd gives the
    reference's gradients (a token unused in the batch keeps `.grad is None`).
    """

    def __init__(self, vocab, state_dict, dtype=torch.float32, storage='fp32', forced=None):
        """forced: optional callable (kind, index) -> stored activation tensor, kinds 'S0', 'F',
        'C1', ('FC', layer) for whole-batch tensors and 'H', 'Y', 'P' per node (index = global
        node id in execution order) / per binary node.  With storage='fp32' the forced values only
        pin the ReLU masks (no rounding of values or gradients)."""
        assert storage in ('fp32', 'bf16')
        self.vocab = vocab
        self.bf16 = storage == 'bf16'
        self.forced = forced
        self._node = self._bin = 0
        self.q = _RoundAct.apply if self.bf16 else (lambda t: t)
        self.p = _Params({k: v.detach().clone().to(dtype).requires_grad_(True)
                          for k, v in state_dict.items()}, self.bf16)
        self.used_fns = None
        self.num_stem = len([k for k in self.p if k.startswith('stem.') and k.endswith('.weight')])
        self.fc_keys = sorted({int(k.split('.')[1]) for k in self.p
                               if k.startswith('classifier.') and k.endswith('.weight')})

    def _act(self, z, kind, index=0):
        """relu, then the storage rounding (or the forced stored value)."""
        if self.forced is not None:
            return _Force.apply(z, self.forced(kind, index), self.bf16)
        return self.q(F.relu(z))

    # Layers.py:29-39 (no batch-norm, proj None) and Module_Model.py:43-53
    def _res_body(self, tok, x):
        p = self.p
        node = self._node
        self._node += 1
        h = self._act(F.conv2d(x, p[tok + '.conv1.weight'], p[tok + '.conv1.bias'], padding=1), 'H', node)
        out = F.conv2d(h, p[tok + '.conv2.weight'], p[tok + '.conv2.bias'], padding=1)
        return self._act(x + out, 'Y', node)

    def _module(self, tok, inputs):
        if len(inputs) == 2:                         # ConcatBlock.forward, Module_Model.py:37-41
            p = self.p
            out = torch.cat(inputs, 1)
            out = self._act(F.conv2d(out, p[tok + '.proj.weight'], p[tok + '.proj.bias']), 'P', self._bin)
            self._bin += 1
            return self._res_body(tok, out)
        return self._res_body(tok, inputs[0])       # ResidualBlock.forward

    def stem(self, x):                               # build_stem, Module_Model.py:55-64
        x = self.q(x)
        for l in range(self.num_stem):
            x = self._act(F.conv2d(x, self.p['stem.%d.weight' % (2 * l)],
                                   self.p['stem.%d.bias' % (2 * l)], padding=1),
                          'F' if l == self.num_stem - 1 else 'S%d' % l)
        return x

    def classifier(self, x):                         # build_classifier, :66-93 (no pooling)
        ks = self.fc_keys
        if self.p['classifier.%d.weight' % ks[0]].dim() == 4:
            x = self._act(F.conv2d(x, self.p['classifier.%d.weight' % ks[0]],
                                   self.p['classifier.%d.bias' % ks[0]]), 'C1')
            ks = ks[1:]
        x = x.reshape(x.size(0), -1)                 # Flatten, Layers.py:42-44
        for li, k in enumerate(ks[:-1]):
            x = self._act(F.linear(x, self.p['classifier.%d.weight' % k],
                                   self.p['classifier.%d.bias' % k]), ('FC', li))
        # logits stay fp32; their incoming gradient is stored in bf16 by a bf16 executor
        out = F.linear(x, self.p['classifier.%d.weight' % ks[-1]],
                       self.p['classifier.%d.bias' % ks[-1]])
        return _RoundGrad.apply(out) if self.bf16 else out

    def forward_modules(self, feats, program):       # forward_modules_ints, :224-233
        N, T = program.shape
        i2t = self.vocab['program_idx_to_token']
        self._node = self._bin = 0
        used = torch.zeros(N, T)
        roots = []
        for i in range(N):
            nodes, u = parse_row(program[i].tolist(), i2t)
            used[i] = torch.tensor(u, dtype=torch.float32)
            outs = []
            for tok, _col, children in nodes:
                if tok == 'scene':
                    outs.append(self._module(tok, [feats[i:i + 1]]))
                else:
                    outs.append(self._module(tok, [outs[c] for c in children]))
            roots.append(outs[-1])
        self.used_fns = used
        return torch.cat(roots, 0)

    def forward(self, x, program):                   # :235-248
        assert x.size(0) == len(program)
        if not (type(program) is torch.Tensor and program.dim() == 2):
            raise ValueError('Unregonized program format')
        feats = self.stem(x)
        return self.classifier(self.forward_modules(feats, program))

    __call__ = forward

    def grads(self):
        return {k: dict.__getitem__(self.p, k).grad for k in self.p}
