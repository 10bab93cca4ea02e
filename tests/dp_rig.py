"""Rig for data-parallel tests on a one-GPU box: RCCL refuses two ranks on one device, so the ranks
share cuda:0 and talk over gloo.

gloo's own handling of device tensors (worker threads copying on streams of their own) stalls now
and then when two processes share the GPU: with VQA_DP_RIG_DIRECT=1 (that path, unpatched) one run
in two of tests/test_dp_gpu.py ended with BOTH ranks parked in `work.wait()` after both had issued
the same five collectives of the same sizes in the same order (the wrapper's debug log,
VQA_DP_DEBUG=1) - nothing the wrapper could still do for either of them.  The transport is not
what these tests are about, so device tensors cross through the host here, but with the timing of
a real asynchronous backend: at issue time the tensor's value is captured on the ISSUING stream
(a copy into pinned memory behind whatever produced it there), nothing is reduced until
`wait()`, and the result lands on the stream current at `wait()`.  A wrapper that issued a
collective on the wrong stream, read a gradient before waiting for it, or waited in different
orders on different ranks fails these tests.  A rank that is stuck for 90 s dumps its stack and
exits, so a hang fails the test instead of stopping the suite."""
import faulthandler
import os

import torch
import torch.distributed as dist


class _StagedWork:
    """One collective on device tensors, carried through the host: `run(host_in, host_out)` is the
    CPU collective; `src` is read at issue time, `dst` written at wait time."""

    def __init__(self, src, dst, run):
        self.dst, self.run = dst, run
        self.host_in = torch.empty(src.shape, dtype=src.dtype, pin_memory=True)
        self.host_in.copy_(src.detach(), non_blocking=True)     # on the issuing stream
        self.host_out = self.host_in if dst is src else torch.empty(dst.shape, dtype=dst.dtype, pin_memory=True)
        self.event = torch.cuda.Event()
        self.event.record(torch.cuda.current_stream())

    def wait(self):
        self.event.synchronize()
        self.run(self.host_in, self.host_out)
        self.dst.copy_(self.host_out, non_blocking=True)        # on the waiting stream
        torch.cuda.current_stream().synchronize()               # the pinned buffers die with this object
        return True


def join(rank, world, port):
    faulthandler.dump_traceback_later(90, exit=True)
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    if os.environ.get('VQA_DP_RIG_DIRECT'):      # gloo's own device-tensor path (see the module docstring)
        return
    patch_collectives()


def patch_collectives():
    """Routes the three collectives the wrapper uses on device tensors through the host (see above)."""
    all_reduce, reduce_scatter, all_gather = dist.all_reduce, dist.reduce_scatter_tensor, dist.all_gather_into_tensor

    def finish(w, async_op):
        if async_op:
            return w
        w.wait()
        return None

    def staged_all_reduce(t, op=dist.ReduceOp.SUM, group=None, async_op=False):
        if not t.is_cuda:
            return all_reduce(t, op=op, group=group, async_op=async_op)
        return finish(_StagedWork(t, t, lambda hi, ho: all_reduce(hi, op=op, group=group)), async_op)

    def staged_reduce_scatter(output, input, op=dist.ReduceOp.SUM, group=None, async_op=False):
        if not input.is_cuda:
            return reduce_scatter(output, input, op=op, group=group, async_op=async_op)
        return finish(_StagedWork(input, output, lambda hi, ho: reduce_scatter(ho, hi, op=op, group=group)), async_op)

    def staged_all_gather(output, input, group=None, async_op=False):
        if not input.is_cuda:
            return all_gather(output, input, group=group, async_op=async_op)
        return finish(_StagedWork(input, output, lambda hi, ho: all_gather(ho, hi, group=group)), async_op)
    dist.all_reduce, dist.reduce_scatter_tensor, dist.all_gather_into_tensor = staged_all_reduce, staged_reduce_scatter, staged_all_gather


def leave():
    faulthandler.cancel_dump_traceback_later()
    dist.destroy_process_group()
