"""Edge cases of the executor's boundary, run by hand on a GPU box: odd program dtypes and strides,
one column, an empty batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd import synth, vocab as V
from vqa_amd.module_net import ModuleNet
vocab = V.make_vocab()
dims = dict(feature_dim=(128, 14, 14), module_dim=128, classifier_proj_dim=128, classifier_fc_layers=(128,))
net = ModuleNet(False, vocab, **dims)
net.load_state_dict(synth.synth_state_dict(3, vocab, **dims))
net.cuda().eval()
n = 6
prog = torch.from_numpy(synth.synth_programs(9, n, vocab, max_len=12, max_nodes=9)).cuda()
x = torch.from_numpy(synth.synth_features(9, n, dims['feature_dim'])).cuda()
with torch.no_grad():
    ref = net(x, prog)
    for name, p in (('int32', prog.to(torch.int32)), ('int16', prog.to(torch.int16)),
                    ('transposed storage', prog.t().contiguous().t()),
                    ('strided columns', torch.stack([prog, prog], 2)[:, :, 0]),
                    ('cpu tensor', prog.cpu())):
        try:
            out = net(x, p)
            print('%-20s %s' % (name, 'same' if torch.equal(out, ref) else 'DIFFERENT'))
        except Exception as e:
            print('%-20s raised %s: %s' % (name, type(e).__name__, str(e)[:100]))
    for name, xx in (('x channels_last', x.contiguous(memory_format=torch.channels_last)),
                     ('x strided batch', torch.stack([x, x], 1)[:, 0]), ('x fp64', x.double()), ('x bf16', x.bfloat16())):
        try:
            out = net(xx, prog)
            print('%-20s max diff %.3g' % (name, (out - ref).abs().max().item()))
        except Exception as e:
            print('%-20s raised %s: %s' % (name, type(e).__name__, str(e)[:100]))
    try:
        out = net(x[:0], prog[:0])
        print('empty batch          ->', tuple(out.shape))
    except Exception as e:
        print('empty batch          raised %s: %s' % (type(e).__name__, str(e)[:100]))
    try:
        net(x, prog[:3])
    except AssertionError:
        print('length mismatch      AssertionError, as the reference')
    bad = prog.clone(); bad[0, 0] = 99
    try:
        net(x, bad)
    except Exception as e:
        print('token 99             raised %s: %s' % (type(e).__name__, str(e)[:80]))
torch.cuda.synchronize()
