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


class ModuleNetRef:
    """The reference executor on plain tensors keyed like its state dict.

    `forward` follows `ModuleNet.forward` (`Module_Model.py:235-248`): stem on the whole batch,
    per-sample interpretation with batch-1 module calls, `torch.cat` of the roots, classifier on
    the whole batch.  Parameters are leaf tensors with `requires_grad`, so autograd gives the
    reference's gradients (a token unused in the batch keeps `.grad is None`).
    """

    def __init__(self, vocab, state_dict, dtype=torch.float32):
        self.vocab = vocab
        self.p = {k: v.detach().clone().to(dtype).requires_grad_(True)
                  for k, v in state_dict.items()}
        self.used_fns = None
        self.num_stem = len([k for k in self.p if k.startswith('stem.') and k.endswith('.weight')])
        self.fc_keys = sorted({int(k.split('.')[1]) for k in self.p
                               if k.startswith('classifier.') and k.endswith('.weight')})

    # Layers.py:29-39 (no batch-norm, proj None) and Module_Model.py:43-53
    def _res_body(self, tok, x):
        p = self.p
        out = F.conv2d(F.relu(F.conv2d(x, p[tok + '.conv1.weight'], p[tok + '.conv1.bias'],
                                       padding=1)),
                       p[tok + '.conv2.weight'], p[tok + '.conv2.bias'], padding=1)
        return F.relu(x + out)

    def _module(self, tok, inputs):
        if len(inputs) == 2:                         # ConcatBlock.forward, Module_Model.py:37-41
            p = self.p
            out = torch.cat(inputs, 1)
            out = F.relu(F.conv2d(out, p[tok + '.proj.weight'], p[tok + '.proj.bias']))
            return self._res_body(tok, out)
        return self._res_body(tok, inputs[0])       # ResidualBlock.forward

    def stem(self, x):                               # build_stem, Module_Model.py:55-64
        for l in range(self.num_stem):
            x = F.relu(F.conv2d(x, self.p['stem.%d.weight' % (2 * l)],
                                self.p['stem.%d.bias' % (2 * l)], padding=1))
        return x

    def classifier(self, x):                         # build_classifier, :66-93 (no pooling)
        ks = self.fc_keys
        if self.p['classifier.%d.weight' % ks[0]].dim() == 4:
            x = F.relu(F.conv2d(x, self.p['classifier.%d.weight' % ks[0]],
                                self.p['classifier.%d.bias' % ks[0]]))
            ks = ks[1:]
        x = x.reshape(x.size(0), -1)                 # Flatten, Layers.py:42-44
        for k in ks[:-1]:
            x = F.relu(F.linear(x, self.p['classifier.%d.weight' % k],
                                self.p['classifier.%d.bias' % k]))
        return F.linear(x, self.p['classifier.%d.weight' % ks[-1]],
                        self.p['classifier.%d.bias' % ks[-1]])

    def forward_modules(self, feats, program):       # forward_modules_ints, :224-233
        N, T = program.shape
        i2t = self.vocab['program_idx_to_token']
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
        return {k: v.grad for k, v in self.p.items()}
