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


class _FakeShardNet(_FakeNet):
    """What DataParallelExecutor(shard_big_update=True) looks at on the executor: the bf16 path, the
    index of the first linear layer, and an optimizer that said it can apply a slice of rows."""

    def __init__(self, with_optimizer=True, precision='bf16'):
        super().__init__()
        self._fc_idx = [3]
        self.precision = precision
        holder = nn.Module()
        holder.weight = self.big
        self.classifier = {3: holder}
        self._fc1_grad_shard = None
        self._fc1_gather = None
        self._fc1_master_stale = False
        import weakref

        class _Opt:
            pass
        self._opt = _Opt()          # kept alive here: the executor holds a weak reference only
        self._sliced_update_optimizer = weakref.ref(self._opt) if with_optimizer else None


def _worker_sharded(rank, world, port):
    """The first linear layer's update split over 8 ranks (F / 8 rows each, the split of a whole node):
    every rank ends up with the average of ITS rows only and no .grad on the parameter; the updated
    bf16 rows of all ranks are exchanged; a slice that no optimizer applied, a second backward pass
    before the sync, an optimizer that cannot apply slices and the fp32 path are all handled loudly
    or by falling back to the replicated all-reduce."""
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        F = 64
        rows = F // world
        base = torch.arange(1, F + 1, dtype=torch.float32)[:, None].expand(F, 64)
        net = _FakeShardNet()
        dp = DataParallelExecutor(net, direct_bytes=64 * 64 * 4, shard_big_update=True)
        for step in range(2):
            net.last_token_counts = [1, 1, 0]
            dp(torch.zeros(1), None)
            net.fake_backward({'classifier.3.weight': (base * (rank + 1 + step)).contiguous(),
                               'scene.conv1.weight': torch.full((4, 3), float(rank)),
                               '<NULL>.conv1.weight': torch.ones(4, 3)})
            dp.finish_grad_sync()
            assert net.big.grad is None
            shard, r, w, owner = net._fc1_grad_shard
            assert (r, w, owner) == (rank, world, dp) and shard.numel() == rows * 64
            mean_factor = sum(k + 1 + step for k in range(world)) / world
            want = base[rank * rows:(rank + 1) * rows] * mean_factor
            assert torch.allclose(shard.view(rows, 64), want, rtol=1e-6)
            assert torch.allclose(net.mods['scene'].grad, torch.full((4, 3), (world - 1) / 2.0))
            # the optimizer's part: take the slice, update these rows of the bf16 copy, exchange
            net._fc1_grad_shard = None
            packed = torch.zeros(F, 64, dtype=torch.bfloat16)
            packed[rank * rows:(rank + 1) * rows] = float(10 * step + rank + 1)
            dp.gather_big_packed(packed)
            net._fc1_gather.wait()
            want_rows = torch.arange(world, dtype=torch.float32).repeat_interleave(rows) + 10 * step + 1
            assert torch.equal(packed.float(), want_rows[:, None].expand(F, 64))
            assert net._fc1_master_stale
        # the fp32 copy of the layer is current on the owning rank only until it is gathered
        net.big.data.zero_()
        net.big.data[rank * rows:(rank + 1) * rows] = float(rank + 1)
        dp.gather_sharded_state()
        assert torch.equal(net.big.data[:, 0], torch.arange(world, dtype=torch.float32).repeat_interleave(rows) + 1)
        assert not net._fc1_master_stale

        # a slice nobody applied: the next backward pass refuses to replace it
        net.last_token_counts = [1, 1, 0]
        dp(torch.zeros(1), None)
        net.fake_backward({'classifier.3.weight': base.contiguous(), 'scene.conv1.weight': torch.ones(4, 3)})
        dp.finish_grad_sync()
        dp(torch.zeros(1), None)
        try:
            net.fake_backward({'classifier.3.weight': base.contiguous(), 'scene.conv1.weight': torch.ones(4, 3)})
            raise AssertionError('an unapplied slice was overwritten')
        except RuntimeError as e:
            assert 'never applied' in str(e)

        # no optimizer that applies slices, or the fp32 path: the replicated all-reduce, .grad on the parameter
        for kw in (dict(with_optimizer=False), dict(precision='fp32')):
            net2 = _FakeShardNet(**kw)
            dp2 = DataParallelExecutor(net2, direct_bytes=64 * 64 * 4, shard_big_update=True)
            net2.last_token_counts = [1, 1, 0]
            dp2(torch.zeros(1), None)
            net2.fake_backward({'classifier.3.weight': (base * (rank + 1)).contiguous(),
                                'scene.conv1.weight': torch.ones(4, 3)})
            dp2.finish_grad_sync()
            assert net2._fc1_grad_shard is None
            assert torch.allclose(net2.big.grad, base * ((world + 1) / 2.0), rtol=1e-6)

        # two backward passes before the sync: refused before anything of the second is issued
        net3 = _FakeNet()
        dp3 = DataParallelExecutor(net3, direct_bytes=64 * 64 * 4)
        net3.last_token_counts = [1, 1, 0]
        dp3(torch.zeros(1), None)
        g = {'classifier.3.weight': torch.ones(64, 64), 'scene.conv1.weight': torch.ones(4, 3)}
        net3.fake_backward(g)
        try:
            net3.fake_backward(g)
            raise AssertionError('a second backward pass went through')
        except RuntimeError as e:
            assert 'second backward pass' in str(e)
        dp3.finish_grad_sync()            # the first pass is intact
        assert torch.allclose(net3.big.grad, torch.ones(64, 64))
        net3.last_token_counts = [1, 1, 0]
        dp3(torch.zeros(1), None)
        net3.fake_backward(g)             # and the next step runs
        dp3.finish_grad_sync()
    finally:
        dist.destroy_process_group()


def test_sharded_update_of_the_large_layer_eight_ranks_gloo():
    mp.spawn(_worker_sharded, args=(8, _free_port()), nprocs=8, join=True)
