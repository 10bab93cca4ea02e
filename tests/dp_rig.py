"""Rig for data-parallel tests on a one-GPU box: RCCL refuses two ranks on one device, so the ranks
share cuda:0 and talk over gloo.  gloo's own handling of device tensors (a worker thread copying on
streams of its own) stalled now and then with two processes on one GPU; the transport is not what
these tests are about, so device tensors are reduced through the host, in issue order, on the calling
thread.  A rank that is stuck for 90 s dumps its stack and exits, so a hang fails the test instead
of stopping the suite."""
import faulthandler
import os

import torch
import torch.distributed as dist


class _Done:
    def wait(self):
        return True


def join(rank, world, port):
    faulthandler.dump_traceback_later(90, exit=True)
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    whole = dist.all_reduce

    def staged(t, op=dist.ReduceOp.SUM, group=None, async_op=False):
        if not t.is_cuda:
            return whole(t, op=op, group=group, async_op=async_op)
        torch.cuda.current_stream().synchronize()
        h = t.detach().cpu()
        whole(h, op=op, group=group)
        t.copy_(h)
        return _Done() if async_op else None
    dist.all_reduce = staged


def leave():
    faulthandler.cancel_dump_traceback_later()
    dist.destroy_process_group()
