#!/usr/bin/env python3
"""Benchmark of the module-executor train step on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 runs one process per GPU: either launched by torch.distributed.run (WORLD_SIZE etc. in the
environment), or started plainly as above, in which case this process - before it touches a GPU -
starts torch.distributed.run on itself as a child, relays the child's result line and exit code.

One step = one pass of the hot path over one synthetic CLEVR-shape batch per GPU: features
NCHW fp32 -> executor forward -> cross-entropy -> executor backward -> gradient all-reduce (N > 1)
-> Adam step (Train_model.py:306-315).  Inputs are resident in HBM before the timed region.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

_T_START = time.perf_counter()
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from vqa_amd import ops, synth, vocab as V  # noqa: E402
from vqa_amd.schedule import Plan  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0     # dense bf16 MFMA, MI355X
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', type=int, choices=[3, 5], default=3,
                    help="BASELINE.json configuration: 3 = the headline (default model, batch 1024/GPU, programs of <= 27 nodes); "
                         "5 = the deep-program stress (module width 256, programs of up to 60 nodes in 64 columns, batch 512/GPU); "
                         "sets the defaults of --batch, --module-dim, --max-len and --max-nodes")
    ap.add_argument('--batch', type=int, default=None, help='programs per GPU')
    ap.add_argument('--module-dim', type=int, default=None)
    ap.add_argument('--max-len', type=int, default=None)
    ap.add_argument('--max-nodes', type=int, default=None)
    ap.add_argument('--batches', type=int, default=4, help='distinct resident synthetic batches cycled through')
    ap.add_argument('--no-optimizer', action='store_true')
    ap.add_argument('--dp-sharded-update', action='store_true',
                    help="N > 1: the first linear layer's gradient is reduce-scattered by rows, each rank updates its rows and the "
                         'updated bf16 rows are all-gathered (same arithmetic; ignores --dp-transport/--dp-algo).  Default: every rank '
                         "all-reduces that gradient and updates the whole layer - the path with the most multi-rank coverage")
    ap.add_argument('--dp-replicated-update', action='store_true', help='the default; accepted for older command lines')
    ap.add_argument('--no-fwd-bwd-only', action='store_true', help='skip the extra steps without the optimizer that give executor_fwd_bwd')
    ap.add_argument('--torch-loss', action='store_true', help='torch.nn.CrossEntropyLoss (seven launches) instead of vqa_amd.loss.CrossEntropyLoss (one)')
    ap.add_argument('--torch-adam', action='store_true', help="torch.optim.Adam(fused=True) instead of vqa_amd.optim.FusedAdam")
    ap.add_argument('--cpu-programs', action='store_true', help='hand the programs over as host tensors (the data loader\'s), '
                    'so that the tile lists of a step are built while the device still runs the step before')
    ap.add_argument('--features', choices=['fp32', 'resident'], default='fp32',
                    help="fp32: the loader's [N,1024,14,14] fp32 batches (default, the reference's input); resident: "
                         'rows of a bf16 feature table held in HBM (vqa_amd.feed.ResidentFeatures), read in place')
    ap.add_argument('--dp-transport', choices=['fp32', 'bf16'], default='fp32',
                    help="how the first linear layer's 411 MB gradient crosses the links (N > 1)")
    ap.add_argument('--dp-algo', choices=['all_reduce', 'reduce_scatter'], default='all_reduce')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-batch', type=int, default=64, help='batch of the CPU baseline (BASELINE config 1: 64)')
    ap.add_argument('--cpu-reps', type=int, default=3, help='timed repetitions of the CPU baseline')
    a = ap.parse_args()
    if a.dp_sharded_update and (a.dp_replicated_update or a.dp_transport != 'fp32' or a.dp_algo != 'all_reduce'):
        ap.error('--dp-sharded-update excludes --dp-replicated-update, --dp-transport and --dp-algo '
                 "(a sharded layer's gradient is reduce-scattered in fp32)")
    preset = {3: dict(batch=1024, module_dim=128, max_len=30, max_nodes=27),
              5: dict(batch=512, module_dim=256, max_len=64, max_nodes=60)}[a.config]
    for k, v in preset.items():
        if getattr(a, k) is None:
            setattr(a, k, v)
    return a


def cpu_baseline(vocab, batch, reps):
    """The CPU restatement of the reference executor (oracle/: per-sample reference order, batch-1
    stock torch fp32 convolutions, the op sequence of Module_Model.py:196-222) on BASELINE.json's
    configuration 1: batch 64, fp32, [64,1024,14,14] features, random CLEVR-style programs of <= 27
    nodes from the generator the GPU run uses.  One warm-up pass, then `reps` timed repetitions each
    of forward only (no_grad) and of forward + cross-entropy + backward; medians.  Every host core
    this process may use (the GPU box gives one GPU's share of its cores and says so through
    OMP_NUM_THREADS)."""
    from oracle.module_net_ref import ModuleNetRef
    cores = os.cpu_count() or 1
    if os.environ.get('OMP_NUM_THREADS', '').isdigit():
        cores = max(1, min(cores, int(os.environ['OMP_NUM_THREADS'])))
    torch.set_num_threads(cores)
    sd = synth.synth_state_dict(17, vocab)
    ref = ModuleNetRef(vocab, sd)
    prog = torch.from_numpy(synth.synth_programs(1001, batch, vocab))
    x = torch.from_numpy(synth.synth_features(1001, batch))
    ans = torch.from_numpy(synth.synth_answers(1001, batch))
    torch.nn.functional.cross_entropy(ref(x[:4], prog[:4]), ans[:4]).backward()        # warm-up
    fwd, both = [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        with torch.no_grad():
            ref(x, prog)
        fwd.append(time.perf_counter() - t0)
        for p in ref.p.values():
            p.grad = None
        t0 = time.perf_counter()
        torch.nn.functional.cross_entropy(ref(x, prog), ans).backward()
        both.append(time.perf_counter() - t0)
    tf, tb = float(np.median(fwd)), float(np.median(both))
    return {'value': batch / tb, 'unit': 'programs/s', 'cores': cores, 'kind': 'port',
            'forward_only': batch / tf,
            'sample': 'BASELINE config 1: batch %d, fp32, default model, programs of <= 27 nodes; 1 warm-up + %d '
                      'repetitions each of forward only (median %.2f s) and forward + cross-entropy + backward '
                      '(median %.2f s, `value`)' % (batch, reps, tf, tb)}


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: one rank per GPU is started
    through torch.distributed.run as a child process of this one, which has not touched a GPU (no
    exec: the child is a fresh process), and whose only job is to pass on the result line and the
    exit code.  Rendezvous on 127.0.0.1."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(a.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print('bench.py: starting %d ranks: %s' % (a.gpus, ' '.join(cmd)), file=sys.stderr, flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE)
    lines = [ln for ln in r.stdout.decode(errors='replace').splitlines() if ln.startswith('{')]
    for ln in r.stdout.decode(errors='replace').splitlines():
        if not ln.startswith('{'):
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    elif r.returncode == 0:
        print('bench.py: the ranks ended without a result line', file=sys.stderr)
        return 1
    return r.returncode


def main():
    a = parse()
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(a))
    # stdout carries exactly one line, the result: whatever libraries print there (RCCL's version
    # banner at communicator set-up, for one) is sent to stderr instead
    result_fd = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != a.gpus:
        print('bench.py: --gpus %d but the launcher started %d ranks: running %d' % (a.gpus, world, world), file=sys.stderr)
        a.gpus = world
    # diagnostics: all ranks on GPU 0 with gloo between them (a one-GPU box cannot host two RCCL
    # ranks) - exercises this script's multi-rank path; the figures it prints mean nothing
    rehearse = world > 1 and os.environ.get('VQA_BENCH_REHEARSE_SHARED_GPU') is not None
    if rehearse:
        local = 0
    ndev = torch.cuda.device_count()
    if ndev and local >= ndev:
        local %= ndev            # a launcher that shows each rank only its own GPU(s)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist = None
    force_dp = world == 1 and os.environ.get('VQA_BENCH_FORCE_DP') is not None    # diagnostics: the wrapper's own cost
    if world > 1 or force_dp:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if force_dp:
            os.environ.setdefault('MASTER_PORT', '29533')
            dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
        elif rehearse:
            # device tensors cross through the host, with the timing of an asynchronous backend
            # (tests/dp_rig.py): gloo's own device-tensor path stalls now and then with two processes on one GPU
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tests'))
            import dp_rig
            import faulthandler
            faulthandler.dump_traceback_later(150, exit=True)
            dist.init_process_group('gloo')
            dp_rig.patch_collectives()
        else:
            dist.init_process_group('nccl', device_id=dev)

    from vqa_amd.module_net import ModuleNet
    vocab = V.make_vocab()
    net = ModuleNet(False, vocab, module_dim=a.module_dim)
    net.load_state_dict(synth.synth_state_dict(17, vocab, module_dim=a.module_dim))   # identical replicas
    net.to(dev).train()
    if os.environ.get('VQA_WGRAD_CHUNK'):
        net.wgrad_chunk = int(os.environ['VQA_WGRAD_CHUNK'])
    step_net = net
    if world > 1 or force_dp:
        from vqa_amd.dp import DataParallelExecutor
        step_net = DataParallelExecutor(net, big_transport=a.dp_transport, big_algo=a.dp_algo,
                                        shard_big_update=a.dp_sharded_update and not a.torch_adam and not a.no_optimizer)
    opt = None
    if not a.no_optimizer:
        # Train_model.py:165-167: Adam, lr 1e-4, weight_decay 5e-5
        if a.torch_adam:
            opt = torch.optim.Adam(net.parameters(), lr=1e-4, weight_decay=5e-5, fused=True)
        else:
            from vqa_amd.optim import FusedAdam
            opt = FusedAdam(net.parameters(), lr=1e-4, weight_decay=5e-5, executor=net)
    # Train_model.py:185: CrossEntropyLoss, mean over the batch
    if a.torch_loss:
        loss_fn = torch.nn.CrossEntropyLoss()
    else:
        from vqa_amd.loss import CrossEntropyLoss
        loss_fn = CrossEntropyLoss()

    # resident synthetic batches (seed = 1000*config + rank, config 3)
    batches, flops_step, flops_exec_step, nodes_step = [], [], [], []
    for b in range(max(1, a.batches)):
        seed = 3000 + rank + 100 * b
        prog = synth.synth_programs(seed, a.batch, vocab, max_len=a.max_len, max_nodes=a.max_nodes)
        g = torch.Generator(device=dev).manual_seed(seed)
        x = torch.randn(a.batch, 1024, 14, 14, device=dev, generator=g).clamp_(min=0)
        ans = torch.from_numpy(synth.synth_answers(seed, a.batch)).to(dev)
        plan = Plan(prog, vocab, a.module_dim)
        flops_step.append(plan.flops()[1])                  # what the reference's per-node execution does
        flops_exec_step.append(plan.flops(executed=True)[1])  # repeated scene nodes of a sample run once here
        nodes_step.append(plan.info.num_nodes)
        batches.append((x, torch.from_numpy(prog) if a.cpu_programs else torch.from_numpy(prog).to(dev), ans))
    if a.features == 'resident':
        from vqa_amd.feed import ResidentFeatures
        table = ResidentFeatures(torch.cat([b[0] for b in batches]), dev)
        batches = [(table.rows(np.arange(i * a.batch, (i + 1) * a.batch)), p_, an) for i, (_, p_, an) in enumerate(batches)]

    def step(i, with_opt=True):
        x, prog, ans = batches[i % len(batches)]
        if opt is not None:
            opt.zero_grad(set_to_none=True)
        else:
            for p in net.parameters():
                p.grad = None
        loss = loss_fn(step_net(x, prog), ans)
        loss.backward()
        if world > 1 or force_dp:
            step_net.finish_grad_sync()
        if opt is not None and with_opt:
            opt.step()
        return loss

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # A full (generation-2) Python garbage collection over the torch-sized heap takes ~90 ms and would
    # land in some timed step; moving the set-up objects to the permanent generation makes later
    # collections cheap.  Nothing is skipped: the collector stays enabled.
    import gc
    gc.collect()
    gc.freeze()
    for i in range(a.warmup):
        step(i)
    barrier()
    # The roofline figure needs the dominant kernel's launches timed with HIP events inside the
    # timed region.  An event pair keeps what it brackets from overlapping its neighbours and
    # costs about 5 us: the library brackets a tree pass's run of level launches with one pair
    # (four pairs per step), every fifth timed step is sampled, and only that kind; the other
    # kinds are timed in a few extra steps after the timed region.
    dominant = 'resblock' if a.module_dim == 128 else 'conv3x3'
    prof_on = os.environ.get('VQA_BENCH_NO_KERNEL_TIMING') is None
    sampled_steps = 0
    ops.prof_enable(False)
    ops.prof_collect()
    t0 = time.perf_counter()
    marks = []
    for i in range(a.steps):
        if prof_on:
            ops.prof_enable([dominant] if i % 5 == 0 else False)
            sampled_steps += i % 5 == 0
        loss = step(a.warmup + i)
        marks.append(time.perf_counter())
        if os.environ.get('VQA_BENCH_VERBOSE') == '2':
            import gc
            st = torch.cuda.memory_stats()
            print('step %d reserved %.2f GB segments %d device_alloc %d gc %s' % (
                i, st['reserved_bytes.all.current'] / 2**30, st['segment.all.current'],
                st['num_device_alloc'], gc.get_count()), file=sys.stderr)
    barrier()
    dt = time.perf_counter() - t0
    if os.environ.get('VQA_BENCH_VERBOSE'):
        print('timed region starts %.2f s after import' % (t0 - _T_START), file=sys.stderr)
        print('host step marks (ms): ' + ' '.join('%.1f' % ((m - t0) * 1e3) for m in marks), file=sys.stderr)
    ops.prof_enable(False)
    prof = ops.prof_collect()
    extra_steps = 0
    if prof_on and rank == 0 and world == 1:
        extra_steps = min(a.steps, 4)
        ops.prof_enable([k for k in ops.KERNEL_KINDS if k != dominant])
        for i in range(extra_steps):
            step(a.warmup + a.steps + i)
        torch.cuda.synchronize()
        ops.prof_enable(False)
        for k, v in ops.prof_collect().items():
            if k != dominant:
                prof[k] = v
    # beside the headline (whole training step): the executor alone, forward + loss + backward without
    # the optimizer, over a few steps after the timed region - the quantity the 40 % target names
    fb_ms = None
    if rank == 0 and world == 1 and opt is not None and not a.no_fwd_bwd_only:
        nfb = min(a.steps, 10)
        step(a.warmup + a.steps, with_opt=False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(nfb):
            step(a.warmup + a.steps + 1 + i, with_opt=False)
        torch.cuda.synchronize()
        fb_ms = (time.perf_counter() - t1) / nfb * 1e3
        fb_flops = sum(flops_exec_step[(a.warmup + a.steps + 1 + i) % len(batches)] for i in range(nfb)) / nfb
    rank_ms = [dt / a.steps * 1e3]
    if dist is not None:
        every = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(every, torch.tensor([dt], device=dev, dtype=torch.float64))
        rank_ms = [t.item() / a.steps * 1e3 for t in every]
        dt = max(t.item() for t in every)

    if rank == 0:
        idx = [(a.warmup + i) % len(batches) for i in range(a.steps)]
        flops = sum(flops_step[j] for j in idx) / a.steps            # per rank per step (rank 0's batches)
        flops_exec = sum(flops_exec_step[j] for j in idx)              # per rank, all timed steps
        k = prof[dominant]
        kname = ('resblock_tiles_kernel (fused residual body: two grouped 3x3 implicit-GEMM convolutions per tile, '
                 'forward and input-gradient launches)' if dominant == 'resblock' else
                 'conv_tiles_kernel<9> (grouped 3x3 implicit-GEMM conv: forward and input-gradient launches)')
        ach = k['flops'] / (k['ms'] * 1e-3) / 1e12 if k['ms'] > 0 else 0.0
        # HBM bytes per launch of the dominant kernel cannot be counted from inside this process: they
        # come from the committed rocprofv3 --pmc passes over this same default workload
        # (scripts/refresh_profiles.sh, scripts/pmc_traffic.py), and only for that workload
        traffic, traffic_note, traffic_src = None, None, None
        prof_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles')

        def latest(suffix):
            """Newest round's committed profile file r<round>_<suffix>, or None."""
            import glob
            import re
            best = None
            for f in glob.glob(os.path.join(prof_dir, 'r*_' + suffix)):
                m = re.match(r'r(\d+)_', os.path.basename(f))
                if m and (best is None or int(m.group(1)) > best[0]):
                    best = (int(m.group(1)), f)
            return best[1] if best else None

        def source_of(path, j):
            return 'committed profile profiles/%s (measured at git %s), not counted in this run' % (
                os.path.basename(path), j.get('commit', 'unknown'))
        tpath = latest('pmc_resblock_traffic.json')
        if (dominant == 'resblock' and a.batch == 1024 and a.max_nodes == 27 and a.max_len == 30 and
                a.features == 'fp32' and tpath):
            with open(tpath) as f:
                tj = json.load(f)
            traffic = tj['hbm_bytes_per_launch']
            traffic_src = source_of(tpath, tj)
            traffic_note = ('bytes per launch, mean over %d launches; %s; algorithmic: 2-3 maps of 50 176 B read and 2 written '
                            'per tile plus 0.59 MB of weights per token and launch' % (tj['launches'], tj['method']))
        # achieved HBM rate of the whole step: bytes of one step from the committed PMC passes over
        # this same default workload (scripts/pmc_step_traffic.py) over this run's step time
        hbm = None
        spath = latest('pmc_step_traffic.json')
        if traffic is not None and world == 1 and not a.no_optimizer and not a.torch_adam and spath:
            with open(spath) as f:
                sj = json.load(f)
            hbm = {'bytes_per_step': sj['hbm_bytes_per_step'], 'achieved_gbps': sj['hbm_bytes_per_step'] / (dt / a.steps) / 1e9,
                   'peak_gbps': 8000.0, 'frac': sj['hbm_bytes_per_step'] / (dt / a.steps) / 8e12,
                   'source': source_of(spath, sj), 'method': sj['method']}
        sharded = world > 1 and a.dp_sharded_update and not a.torch_adam and not a.no_optimizer
        out = {
            'metric': 'programs/sec (module-net fwd+bwd, whole node)',
            'value': a.batch * world * a.steps / dt,
            'unit': 'programs/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': dt / a.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': 'executor train step (fwd + cross-entropy + bwd%s%s), bf16 MFMA with fp32 '
                                   'accumulation, batch %d/GPU, %s ModuleNet, random '
                                   'well-formed CLEVR-style programs (<=%d nodes, T=%d, mean %.1f nodes), '
                                   'features %s, programs %s'
                                   % (('' if world == 1 else " + RCCL gradient exchange (first linear layer's gradient reduce-scattered by rows)"
                                       if sharded else ' + RCCL gradient all-reduce (large gradient: %s, %s)' % (a.dp_transport, a.dp_algo)),
                                      '' if a.no_optimizer else (' + torch fused Adam' if a.torch_adam else ' + Adam (vqa_amd.optim.FusedAdam)'), a.batch,
                                      'default 117.5 M-parameter' if a.module_dim == 128 else 'module width %d' % a.module_dim,
                                      a.max_nodes, a.max_len,
                                      float(np.mean(nodes_step)) / a.batch,
                                      '[N,1024,14,14] fp32 resident in HBM' if a.features == 'fp32' else
                                      'rows of a bf16 [R,196,1024] table resident in HBM, read in place',
                                      'in host memory' if a.cpu_programs else 'resident in HBM'),
                       'global_batch': a.batch * world,
                       'parallelism': 'dp%d' % world + ('' if world == 1 else (
                           ', every rank updates all parameters' if not sharded else
                           ", first linear layer's gradient reduce-scattered by rows, its Adam update and bf16 copy "
                           'sharded over the ranks (same arithmetic as the replicated update)'))},
            # executed: what the device runs (the scene nodes of a sample are one computation);
            # model: what the reference's per-node execution of the same programs would do
            'executed_tflops': flops_exec * world / dt / 1e12,
            'model_tflops': flops * world * a.steps / dt / 1e12,
            'mfma_peak_frac': flops_exec / dt / 1e12 / MFMA_BF16_PEAK_TFLOPS,
            'model_flops_frac': flops * a.steps / dt / 1e12 / MFMA_BF16_PEAK_TFLOPS,
            'roofline': {'bound': 'mfma', 'kernel': kname,
                         'achieved': ach, 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': ach / MFMA_BF16_PEAK_TFLOPS, 'traffic': traffic, 'traffic_source': traffic_src,
                         'traffic_note': traffic_note,
                         'launches': k['launches'], 'avg_launch_ms': k['ms'] / max(k['launches'], 1),
                         'sampled_steps': sampled_steps},
            'kernels': {n: {'launches': v['launches'],
                            'ms_per_step': v['ms'] / max(sampled_steps if n == dominant else extra_steps, 1),
                            'tflops': (v['flops'] / (v['ms'] * 1e-3) / 1e12) if v['ms'] > 0 else 0.0}
                        for n, v in prof.items()},
            'hbm': hbm,
            'loss': float(loss.detach()),
        }
        if fb_ms is not None:
            out['executor_fwd_bwd'] = {
                'ms_per_step': fb_ms, 'programs_per_s': a.batch / (fb_ms * 1e-3),
                'executed_tflops': fb_flops / (fb_ms * 1e-3) / 1e12,
                'mfma_peak_frac': fb_flops / (fb_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                'note': 'forward + cross-entropy + backward without the optimizer step, %d steps after the timed region; '
                        'not the headline: `value` is the whole training step' % min(a.steps, 10)}
        if world > 1 or force_dp:
            # what ran between the ranks, so that a first multi-GPU line can be read without the code
            out['ranks'] = {'n': world, 'ms_per_step_min': min(rank_ms), 'ms_per_step_max': max(rank_ms),
                            'ms_per_step': rank_ms, 'backend': dist.get_backend(),
                            'large_gradient': 'reduce-scatter by rows + sharded Adam + all-gather of bf16 rows' if sharded else
                                              '%s %s, every rank updates the whole layer' % (a.dp_transport, a.dp_algo)}
        if rehearse:
            out['rehearsal'] = 'all ranks on one GPU, gloo through the host: not a measurement'
        if not a.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(vocab, a.cpu_batch, a.cpu_reps)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + '\n').encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
