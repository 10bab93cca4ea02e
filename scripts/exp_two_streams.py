"""Experiment: the tree forward of a 1024-program batch as one plan on one stream, against two
512-program plans on two streams (their level launches interleave, so the tail of one level and the
in-phase start of the next are covered by the other half's tiles)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vqa_amd import synth, vocab as V, _lib, ops
from vqa_amd.schedule import Plan
from vqa_amd.module_net import ModuleNet
HW, c = 196, 128
vocab = V.make_vocab()
net = ModuleNet(False, vocab); net.load_state_dict(synth.synth_state_dict(17, vocab)); net.cuda()
dev = torch.device('cuda', 0)
pk = net._pack_weights(dev)
prog = synth.synth_programs(3000, 1024, vocab)
flags = _lib.PLAN_SKIP_SEPARATE | _lib.PLAN_BALANCE
lib = _lib.load()

def setup(rows):
    plan = Plan(prog[rows], vocab, c, 32, flags)
    blob = torch.from_numpy(plan.blob()).cuda()
    arena = (torch.randn(plan.info.slots_fwd * HW * c, device=dev) * 0.5).clamp_(min=0).to(torch.bfloat16)
    ctx = _lib.TreeCtx()
    ctx.arena, ctx.blob = arena.data_ptr(), blob.data_ptr()
    for k in ('w_conv1', 'w_conv2', 'w_proj', 'wd_conv1', 'wd_conv2', 'wd_proj', 'b_conv1', 'b_conv2', 'b_proj'):
        setattr(ctx, k, pk[k].data_ptr())
    ctx.C, ctx.V = c, len(net._tokens)
    ctx.flags = 0
    return plan, blob, arena, ctx

def run(parts, streams, reps=20):
    def once():
        ev = torch.cuda.Event(); ev.record()
        dones = []
        for (plan, blob, arena, ctx), s in zip(parts, streams):
            with torch.cuda.stream(s):
                s.wait_event(ev)
                _lib.check(lib.vqa_tree_forward(plan.handle, C.byref(ctx), s.cuda_stream))
                d = torch.cuda.Event(); d.record(s); dones.append(d)
        for d in dones:
            torch.cuda.current_stream().wait_event(d)
    for _ in range(3): once()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): once()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

whole = [setup(np.arange(1024))]
halves = [setup(np.arange(0, 512)), setup(np.arange(512, 1024))]
quarters = [setup(np.arange(i * 256, (i + 1) * 256)) for i in range(4)]
s = [torch.cuda.Stream() for _ in range(4)]
print('one plan, one stream        %.0f us  (levels %d, nodes %d)' % (run(whole, s[:1]), whole[0][0].info.num_levels, whole[0][0].info.num_exec_nodes))
print('two halves, one stream      %.0f us' % run(halves, [s[0], s[0]]))
print('two halves, two streams     %.0f us' % run(halves, s[:2]))
print('four quarters, four streams %.0f us' % run(quarters, s))
