"""Data-parallel gradient sync on 2 ranks over gloo (CPU): the wrapper's bucketing, the early
in-place all-reduce of the large gradient, and the rule that a token unused on every rank keeps
grad None while one used anywhere is averaged over all ranks."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from vqa_amd.dp import DataParallelExecutor, loss_weight, shard_rows


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


class _FakeGroupedNet(nn.Module):
    """Per-token gradients handed over as rows of [V, ...] arrays (grad_group_hook), as ModuleNet does."""

    def __init__(self):
        super().__init__()
        self._tokens = ['<NULL>', 'scene', 'count']
        self.stem = nn.Parameter(torch.zeros(5, 5))
        self.w = nn.ParameterList([nn.Parameter(torch.zeros(600, 600)) for _ in self._tokens])
        self.b = nn.ParameterList([nn.Parameter(torch.zeros(4)) for _ in self._tokens])
        self.grad_ready_hook = self.grad_group_hook = self.backward_done_hook = None
        self.last_token_counts = None

    def _param_groups(self):
        out = [('stem.0.weight', self.stem)]
        for i, t in enumerate(self._tokens):
            out += [(t + '.conv1.weight', self.w[i]), (t + '.conv1.bias', self.b[i])]
        return out

    def forward(self, x, program):
        return x.sum() * 0

    def fake_backward(self, stem, gw, gb):
        self.grad_ready_hook('stem.0.weight', stem)
        self.grad_group_hook(gw, {t + '.conv1.weight': i for i, t in enumerate(self._tokens)})
        self.grad_group_hook(gb, {t + '.conv1.bias': i for i, t in enumerate(self._tokens)})
        # autograd would install views of the arrays for the tokens this rank used
        for i, c in enumerate(self.last_token_counts):
            self.w[i].grad = gw[i] if c else None
            self.b[i].grad = gb[i] if c else None
        self.backward_done_hook()


def _worker_grouped(rank, world, port):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        net = _FakeGroupedNet()
        dp = DataParallelExecutor(net)
        for step in range(2):
            # rank 0 uses <NULL> and scene, rank 1 scene only, nobody count; the arrays hold zeros for unused tokens
            net.last_token_counts = [2, 1, 0] if rank == 0 else [0, 3, 0]
            dp(torch.zeros(1), None)
            gw = torch.zeros(3, 600, 600)
            gb = torch.zeros(3, 4)
            gw[1], gb[1] = 10.0 * (rank + 1) + step, rank + 1.0
            if rank == 0:
                gw[0], gb[0] = 7.0, 3.0
            net.fake_backward(torch.full((5, 5), float(rank)), gw, gb)
            dp.finish_grad_sync()
            assert torch.allclose(net.stem.grad, torch.full((5, 5), 0.5))
            assert torch.allclose(net.w[1].grad, torch.full((600, 600), 15.0 + step))
            assert torch.allclose(net.b[1].grad, torch.full((4,), 1.5))
            assert torch.allclose(net.w[0].grad, torch.full((600, 600), 3.5))       # rank 1 contributed zeros
            assert torch.allclose(net.b[0].grad, torch.full((4,), 1.5))
            assert net.w[2].grad is None and net.b[2].grad is None                  # unused everywhere
            assert net.w[1].grad.data_ptr() == gw[1].data_ptr()                     # reduced in place
    finally:
        dist.destroy_process_group()


def test_grouped_gradients_two_ranks_gloo():
    mp.spawn(_worker_grouped, args=(2, _free_port()), nprocs=2, join=True)


def test_loss_weight_makes_mean_of_shard_means_the_global_mean():
    g = torch.Generator().manual_seed(4)
    for n, world in ((10, 4), (16, 4), (7, 8), (1000, 3)):
        losses = torch.rand(n, generator=g, dtype=torch.float64)
        acc = 0.0
        for r in range(world):
            lo, hi = shard_rows(n, r, world)
            if hi > lo:
                acc += loss_weight(n, r, world) * losses[lo:hi].mean().item() / world
            else:
                assert loss_weight(n, r, world) == 0.0
        assert abs(acc - losses.mean().item()) < 1e-12
    assert loss_weight(16, 2, 4) == 1.0


def _worker_big_modes(rank, world, port):
    """The large gradient through every transport/algorithm pair: the averaged result must be the
    mean of the ranks' gradients - to fp32 rounding on fp32 transport, to bf16 rounding (2^-8
    relative per element) on bf16 transport - and a second step must reuse the staging buffers."""
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        for transport in ('fp32', 'bf16'):
            for algo in ('all_reduce', 'reduce_scatter'):
                net = _FakeNet()
                dp = DataParallelExecutor(net, direct_bytes=64 * 64 * 4, big_transport=transport, big_algo=algo)
                for step in range(2):
                    net.last_token_counts = [1, 1, 0]
                    dp(torch.zeros(1), None)
                    gens = [torch.Generator().manual_seed(100 * step + r) for r in range(world)]
                    every = [torch.randn(64, 64, generator=g) for g in gens]
                    want = sum(every) / world
                    net.fake_backward({'classifier.3.weight': every[rank].clone(),
                                       'scene.conv1.weight': torch.full((4, 3), float(rank)),
                                       '<NULL>.conv1.weight': torch.ones(4, 3)})
                    dp.finish_grad_sync()
                    got = net.big.grad
                    assert got.dtype == torch.float32 and got.shape == (64, 64)
                    err = ((got - want).norm() / want.norm()).item()
                    assert err <= (1e-6 if transport == 'fp32' else 1e-2), (transport, algo, step, err)
                    if transport == 'bf16':
                        assert err > 1e-5          # it did cross as bf16
                    assert torch.allclose(net.mods['scene'].grad, torch.full((4, 3), 0.5))
    finally:
        dist.destroy_process_group()


def test_large_gradient_transports_and_algorithms_two_ranks_gloo():
    mp.spawn(_worker_big_modes, args=(2, _free_port()), nprocs=2, join=True)
