"""bench.py's output contract: exactly one JSON line on stdout with the agreed keys, whatever the
libraries print; a bounded CPU baseline beside it.  Small batch, two steps."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_json_line_with_the_agreed_keys():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1', '--batch', '64',
                        '--batches', '1', '--cpu-batch', '4', '--cpu-reps', '1'], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1 and d['higher_is_better'] is True
    assert d['unit'] == 'programs/s' and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert abs(d['value'] - 64 / (d['ms_per_step'] * 1e-3)) <= 1e-6 * d['value']
    assert 'workload' in d['config'] and 'model' not in d['config']
    rf = d['roofline']
    assert rf['bound'] == 'mfma' and rf['unit'] == 'TFLOP/s' and rf['peak'] == 2500.0
    assert rf['achieved'] > 0 and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-9
    assert rf['traffic'] is None                 # the committed counter passes are of the default workload only
    cb = d['cpu_baseline']
    assert cb['kind'] == 'port' and cb['cores'] >= 1 and cb['value'] > 0 and cb['unit'] == 'programs/s' and cb['sample']
    assert cb['forward_only'] > cb['value']       # forward alone is reported beside forward + backward
    assert d['mfma_peak_frac'] <= d['model_flops_frac']      # executed work never exceeds the reference-equivalent count
    fb = d['executor_fwd_bwd']                                # the executor alone, beside the whole-step headline
    assert 0 < fb['ms_per_step'] and fb['mfma_peak_frac'] > 0 and 'not the headline' in fb['note']


def _rehearse(cmd):
    env = dict(os.environ, VQA_BENCH_REHEARSE_SHARED_GPU='1')
    env.pop('WORLD_SIZE', None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=400, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith('{'), r.stdout
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['config']['global_batch'] == 128
    assert abs(d['value'] - 128 / (d['ms_per_step'] * 1e-3)) <= 1e-6 * d['value']
    assert 'rehearsal' in d
    assert 'cpu_baseline' not in d and 'executor_fwd_bwd' not in d          # rank 0 at N = 1 only
    rk = d['ranks']
    assert rk['n'] == 2 and len(rk['ms_per_step']) == 2 and rk['backend'] == 'gloo'
    assert rk['ms_per_step_min'] <= rk['ms_per_step_max'] == d['ms_per_step']
    return d


def test_two_rank_launch_path_rehearsed_on_one_gpu():
    """The script's multi-rank path as the driver launches it (torch.distributed.run, one process per
    rank, RANK/LOCAL_RANK/WORLD_SIZE from the environment): both ranks on this box's one GPU with the
    collectives staged through the host (VQA_BENCH_REHEARSE_SHARED_GPU; the figures mean nothing).
    One JSON line from rank 0, whole-job values; by default every rank updates all parameters, and the
    sharded update of the first linear layer is named when it is asked for."""
    import socket
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    launch = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
              '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'bench.py'),
              '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '64', '--batches', '1']
    d = _rehearse(launch)
    assert 'every rank updates all parameters' in d['config']['parallelism']
    assert 'fp32 all_reduce' in d['ranks']['large_gradient'] and 'all-reduce (large gradient: fp32, all_reduce)' in d['config']['workload']
    d = _rehearse(launch + ['--dp-sharded-update'])
    assert 'sharded over the ranks' in d['config']['parallelism'] and 'reduce-scatter' in d['ranks']['large_gradient']
    assert 'large gradient:' not in d['config']['workload']


def test_plain_start_with_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the form a one-GPU run is started in): the
    script starts torch.distributed.run on itself as a child process, relays the one result line and
    the exit code."""
    _rehearse([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
               '--batch', '64', '--batches', '1'])
