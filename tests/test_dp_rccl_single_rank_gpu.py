"""The data-parallel wrapper on the real RCCL backend ("nccl") with a single rank: every collective
the wrapper issues (presence MAX on the auxiliary stream, in-place AVG of the first linear layer's
gradient from the side stream, the per-kind gradient arrays, the flat bucket) goes through RCCL, and
the training trajectory must equal the unwrapped executor's.  (More ranks need more GPUs; the
multi-rank logic is covered on gloo.)"""
import os
import socket

import pytest
import torch
import torch.nn.functional as F

from vqa_amd import synth

pytestmark = pytest.mark.gpu


def test_single_rank_rccl_matches_plain_training(vocab):
    import torch.distributed as dist
    from vqa_amd.dp import DataParallelExecutor
    from vqa_amd.module_net import ModuleNet
    from vqa_amd.optim import FusedAdam
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dev = torch.device('cuda', 0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        dims = dict(feature_dim=(128, 14, 14), module_dim=128, classifier_proj_dim=128, classifier_fc_layers=(128,))
        sd = synth.synth_state_dict(41, vocab, **dims)
        runs = []
        for wrapped in (False, True, ('bf16', 'reduce_scatter'), 'sharded'):
            net = ModuleNet(False, vocab, **dims)
            net.load_state_dict(sd)
            net.to(dev).train()
            if wrapped is True:
                step_net = DataParallelExecutor(net, direct_bytes=1 << 20)
            elif wrapped == 'sharded':   # reduce-scatter of the large gradient, Adam on this rank's rows, all-gather of the bf16 rows
                step_net = DataParallelExecutor(net, direct_bytes=1 << 20, shard_big_update=True)
            elif wrapped:      # the large gradient as a bf16 copy through RCCL's reduce-scatter + all-gather
                step_net = DataParallelExecutor(net, direct_bytes=1 << 20, big_transport=wrapped[0], big_algo=wrapped[1])
            else:
                step_net = net
            opt = FusedAdam(net.parameters(), lr=1e-3, weight_decay=5e-5, executor=net)
            losses = []
            for step in range(3):
                prog = torch.from_numpy(synth.synth_programs(90 + step, 16, vocab, max_len=12, max_nodes=9)).to(dev)
                x = torch.from_numpy(synth.synth_features(90 + step, 16, dims['feature_dim'])).to(dev)
                ans = torch.from_numpy(synth.synth_answers(90 + step, 16)).to(dev)
                opt.zero_grad(set_to_none=True)
                loss = F.cross_entropy(step_net(x, prog), ans)
                loss.backward()
                if wrapped:
                    step_net.finish_grad_sync()
                unused = [n for n, p in net.named_parameters() if p.grad is None]
                opt.step()
                losses.append(loss.item())
            if wrapped == 'sharded':
                step_net.gather_sharded_state(opt)
            torch.cuda.synchronize()
            runs.append((losses, unused, {k: v.detach().clone() for k, v in net.state_dict().items()}))
        (la, ua, sa), (lb, ub, sb), (lc, uc, sc), (ld, ud, sd_) = runs
        assert sorted(ud) == sorted(ua + ['classifier.3.weight'])   # that gradient lives as this rank's rows
        for a, b in zip(la, ld):
            assert abs(a - b) <= 1e-4 * max(1.0, abs(a)), (la, ld)
        for k in sa:
            torch.testing.assert_close(sa[k], sd_[k], rtol=2e-3, atol=2e-5, msg=k)
        assert uc == ua
        for a, c in zip(la, lc):       # one bf16 rounding of one gradient: the trajectory moves by less than bf16 noise
            assert abs(a - c) <= 5e-3 * max(1.0, abs(a)), (la, lc)
        assert ua == ub and len(ua) > 0                      # tokens no program used keep grad None either way
        for a, b in zip(la, lb):
            assert abs(a - b) <= 1e-4 * max(1.0, abs(a)), (la, lb)
        for k in sa:
            torch.testing.assert_close(sa[k], sb[k], rtol=2e-3, atol=2e-5, msg=k)
    finally:
        dist.destroy_process_group()
