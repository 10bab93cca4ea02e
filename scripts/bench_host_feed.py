"""Training steps fed from a feature shard in HOST memory (vqa_amd.feed.HostFeed: gather into pinned
buffers on a background thread, copy over PCIe on a stream of its own, three batches ahead), against
the same steps on rows of the HBM-resident table.  This is the PCIe-inclusive rate of the default
workload; bench.py's figure is the resident one.  usage: bench_host_feed.py [rows] [steps]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from vqa_amd import synth, vocab as V
from vqa_amd.feed import FeatureShard, HostFeed, ResidentFeatures, ShardWriter
from vqa_amd.module_net import ModuleNet
from vqa_amd.optim import FusedAdam

ROWS = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 40
BATCH, WARM = 1024, 5
dev = torch.device('cuda', 0)
vocab = V.make_vocab()
net = ModuleNet(False, vocab)
net.load_state_dict(synth.synth_state_dict(17, vocab))
net.to(dev).train()
opt = FusedAdam(net.parameters(), lr=1e-4, weight_decay=5e-5, executor=net)
loss_fn = torch.nn.CrossEntropyLoss()

tmp = tempfile.mkdtemp()
path = os.path.join(tmp, 'features.vqaf')
t0 = time.perf_counter()
with ShardWriter(path) as w:
    g = torch.Generator().manual_seed(5)
    for r0 in range(0, ROWS, 256):
        n = min(256, ROWS - r0)
        w.add_batch(['img%d' % i for i in range(r0, r0 + n)], torch.randn(n, 1024, 14, 14, generator=g).clamp_(min=0))
shard = FeatureShard(path)
print('shard: %d rows, %.2f GB, written in %.1f s' % (shard.count, os.path.getsize(path) / 1e9, time.perf_counter() - t0), flush=True)

rng = np.random.default_rng(3)
index = [rng.integers(0, ROWS, BATCH) for _ in range(WARM + STEPS)]
progs = [torch.from_numpy(synth.synth_programs(3000 + 100 * (i % 4), BATCH, vocab, max_len=30, max_nodes=27)).to(dev) for i in range(4)]
answers = [torch.from_numpy(synth.synth_answers(3000 + 100 * (i % 4), BATCH)).to(dev) for i in range(4)]


def run(batches, label):
    t = None
    for i, x in enumerate(batches):
        if i == WARM:
            torch.cuda.synchronize()
            t = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        loss_fn(net(x, progs[i % 4]), answers[i % 4]).backward()
        opt.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / STEPS
    print('%-34s %7.2f ms per step  %8.0f programs/s' % (label, dt * 1e3, BATCH / dt), flush=True)


_ = shard.rows[:].sum(dtype=np.int64)       # the file is in the page cache from here on
from vqa_amd import _lib
lib = _lib.load()
src = torch.from_numpy(shard.rows)
pinned = torch.empty(BATCH, 196, 1024, dtype=torch.int16).pin_memory()
devbuf = torch.empty(BATCH, 196, 1024, dtype=torch.int16, device=dev)
ix = torch.as_tensor(index[0].astype(np.int64))
for threads in (1, 4, 8, 16, 32):
    t = time.perf_counter()
    for _ in range(5):
        _lib.check(lib.vqa_gather_rows(src.data_ptr(), shard.count, 196 * 1024 * 2, ix.data_ptr(), BATCH, pinned.data_ptr(), threads))
    dt = (time.perf_counter() - t) / 5
    print('gather of one batch, %2d threads: %6.2f ms  %5.1f GB/s' % (threads, dt * 1e3, pinned.numel() * 2 / dt / 1e9), flush=True)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    devbuf.copy_(pinned, non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 5
print('pinned -> HBM copy of one batch:  %6.2f ms  %5.1f GB/s' % (dt * 1e3, pinned.numel() * 2 / dt / 1e9), flush=True)
feed = HostFeed(shard, index, dev, depth=3)
run(feed, 'host shard -> pinned -> PCIe -> HBM')
feed.close()
table = ResidentFeatures(shard, dev)
run((table.rows(ix) for ix in index), 'rows of the HBM-resident table')
os.remove(path)
os.rmdir(tmp)
