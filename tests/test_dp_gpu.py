"""Data parallelism end to end on the GPU: two ranks (both on cuda:0 with gloo between them,
tests/dp_rig.py, since RCCL refuses two ranks on one device) each run the HIP executor on their half
of a batch; the synchronised gradients must equal those of one process on the whole batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from vqa_amd import synth, vocab as V

pytestmark = pytest.mark.gpu

DIMS = dict(feature_dim=(128, 14, 14), module_dim=128, classifier_proj_dim=128, classifier_fc_layers=(128,))


def _worker(rank, world, port, n):
    import dp_rig
    dp_rig.join(rank, world, port)
    try:
        from vqa_amd.dp import DataParallelExecutor, shard_rows
        from vqa_amd.module_net import ModuleNet
        vocab = V.make_vocab()
        sd = synth.synth_state_dict(5, vocab, **DIMS)
        prog = synth.synth_programs(6, n, vocab, max_len=10, max_nodes=8)
        prog[0, :] = 0                      # an all-<NULL> row: a token only rank 0 sees
        x = torch.from_numpy(synth.synth_features(6, n, DIMS['feature_dim'])).cuda()
        ans = torch.from_numpy(synth.synth_answers(6, n)).cuda()
        prog = torch.from_numpy(prog).cuda()
        # single process, whole batch
        full = ModuleNet(False, vocab, **DIMS)
        full.load_state_dict(sd)
        full.cuda()
        F.cross_entropy(full(x, prog), ans).backward()
        # this rank's shard
        net = ModuleNet(False, vocab, **DIMS)
        net.load_state_dict(sd)
        net.cuda()
        dp = DataParallelExecutor(net, direct_bytes=1 << 20)
        lo, hi = shard_rows(n, rank, world)
        F.cross_entropy(dp(x[lo:hi], prog[lo:hi]), ans[lo:hi]).backward()
        dp.finish_grad_sync()
        torch.cuda.synchronize()
        for (k, p), (_, q) in zip(net.named_parameters(), full.named_parameters()):
            if q.grad is None:
                assert p.grad is None, k
                continue
            assert p.grad is not None, k
            err = ((p.grad - q.grad).norm() / q.grad.norm().clamp_min(1e-20)).item()
            assert err <= 2e-3, '%s %g' % (k, err)
    finally:
        dp_rig.leave()


def test_two_rank_gradients_equal_single_process():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, 8), nprocs=2, join=True)
