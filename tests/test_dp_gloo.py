"""Data-parallel gradient sync on 2 ranks over gloo (CPU): the wrapper's bucketing, the early
in-place all-reduce of the large gradient, and the rule that a token unused on every rank keeps
grad None while one used anywhere is averaged over all ranks."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from vqa_amd.dp import DataParallelExecutor, shard_rows


class _FakeNet(nn.Module):
    """Stands in for ModuleNet on the CPU: same hook protocol, grads supplied by the test."""

    def __init__(self):
        super().__init__()
        self._tokens = ['<NULL>', 'scene', 'count']
        self.big = nn.Parameter(torch.zeros(64, 64))
        self.mods = nn.ParameterDict({t.replace('<', '').replace('>', ''): nn.Parameter(torch.zeros(4, 3)) for t in self._tokens})
        self.grad_ready_hook = self.backward_done_hook = None
        self.last_token_counts = None

    def _param_groups(self):
        return [('classifier.3.weight', self.big)] + [(t + '.conv1.weight', self.mods[t.replace('<', '').replace('>', '')]) for t in self._tokens]

    def forward(self, x, program):
        return x.sum() * 0 + sum(p.sum() * 0 for p in self.parameters())

    def fake_backward(self, grads):
        for n, _ in self._param_groups():
            if grads.get(n) is not None:
                self.grad_ready_hook(n, grads[n])
        self.backward_done_hook()


def _worker(rank, world, port):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        net = _FakeNet()
        dp = DataParallelExecutor(net, direct_bytes=64 * 64 * 4)
        # rank 0 uses <NULL> and scene; rank 1 uses scene only; nobody uses count
        net.last_token_counts = [2, 1, 0] if rank == 0 else [0, 3, 0]
        dp(torch.zeros(1), None)
        g = {'classifier.3.weight': torch.full((64, 64), float(rank + 1)),
             'scene.conv1.weight': torch.full((4, 3), 10.0 * (rank + 1))}
        if rank == 0:
            g['<NULL>.conv1.weight'] = torch.full((4, 3), 7.0)
        net.fake_backward(g)
        dp.finish_grad_sync()
        assert torch.allclose(net.big.grad, torch.full((64, 64), 1.5))
        assert torch.allclose(net.mods['scene'].grad, torch.full((4, 3), 15.0))
        assert torch.allclose(net.mods['NULL'].grad, torch.full((4, 3), 3.5))      # zeros from rank 1
        assert net.mods['count'].grad is None
        # second step with other values: buffers are reused
        net.last_token_counts = [0, 1, 1]
        dp(torch.zeros(1), None)
        net.fake_backward({'classifier.3.weight': torch.ones(64, 64), 'scene.conv1.weight': torch.ones(4, 3),
                           'count.conv1.weight': torch.full((4, 3), 2.0 * rank)})
        dp.finish_grad_sync()
        assert torch.allclose(net.mods['count'].grad, torch.full((4, 3), 1.0))
        assert net.mods['NULL'].grad is None
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_grad_sync_two_ranks_gloo():
    mp.spawn(_worker, args=(2, _free_port()), nprocs=2, join=True)


def test_shard_rows_matches_dataparallel_scatter():
    for n, w in ((1024, 8), (10, 4), (7, 2), (3, 4)):
        chunks = [c.numel() for c in torch.arange(n).chunk(w)]
        got = [shard_rows(n, r, w) for r in range(w)]
        assert [b - a for a, b in got if b > a] == chunks
        assert got[0][0] == 0 and max(b for _, b in got) == n
