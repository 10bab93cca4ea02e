"""Feature feed (vqa_amd.feed): shard files, the HBM-resident table and the pinned-memory prefetcher.
The reference's loader reads one fp32 file per sample (DataLoader.py:118-119); a run fed from a
shard must compute exactly what a run fed with those fp32 tensors computes."""
import os
import numpy as np
import pytest
import torch

from vqa_amd import feed, synth, vocab as V


def _files(tmp_path, n, c=1024, seed=5):
    """Per-image files as the reference stores them: a saved [C, 14, 14] fp32 tensor each."""
    d = str(tmp_path) + '/feats_'
    g = torch.Generator().manual_seed(seed)
    names, maps = [], []
    for i in range(n):
        t = torch.randn(c, 14, 14, generator=g).clamp_(min=0)
        name = 'CLEVR_train_%06d.png' % (i * 7)
        torch.save(t, d + name)
        names.append(name)
        maps.append(t)
    return d, names, torch.stack(maps)


def test_shard_round_trip(tmp_path):
    d, names, maps = _files(tmp_path, 5, c=256)
    shard = feed.pack_reference_features(d, names, str(tmp_path / 'f.vqaf'), channels=256)
    assert shard.count == 5 and shard.channels == 256 and shard.names == names
    assert os.path.getsize(shard.path) == feed.HEADER_BYTES + 5 * 196 * 256 * 2 + len('\n'.join(names))
    assert shard.row_of(names[3]) == 3 and shard.rows_of([names[4], names[0]]).tolist() == [4, 0]
    with pytest.raises(KeyError):
        shard.row_of('nope')
    # rows are the round-to-nearest-even bf16 values, pixel-major
    want = maps.to(torch.bfloat16).reshape(5, 256, 196).permute(0, 2, 1)
    got = torch.from_numpy(np.ascontiguousarray(shard.rows[:])).view(torch.bfloat16)
    assert torch.equal(got, want)
    assert torch.equal(shard.batch_fp32([2, 2, 0]), maps.to(torch.bfloat16).to(torch.float32)[[2, 2, 0]])
    # incremental writer, batches
    with feed.ShardWriter(str(tmp_path / 'g.vqaf'), channels=256) as w:
        w.add_batch(names[:3], maps[:3])
        w.add_batch(names[3:], maps[3:])
    again = feed.FeatureShard(str(tmp_path / 'g.vqaf'))
    assert np.array_equal(again.rows[:], shard.rows[:]) and again.names == names


def test_shard_errors(tmp_path):
    p = tmp_path / 'bad.vqaf'
    p.write_bytes(b'not a shard' + b'\0' * 5000)
    with pytest.raises(ValueError):
        feed.FeatureShard(str(p))
    with pytest.raises(ValueError):
        with feed.ShardWriter(str(tmp_path / 'h.vqaf'), channels=64) as w:
            w.add('a', torch.zeros(32, 14, 14))
    with pytest.raises(ValueError):
        feed._to_rows(torch.zeros(3, 8, 7, 7))


def test_host_gather_matches_fancy_indexing():
    """vqa_gather_rows is host code (the pinned-buffer gather of HostFeed): any thread count gives
    table[index], an empty batch is fine, a row outside the table is refused before any write."""
    from vqa_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(2)
    for row_bytes, rows in ((6, 40), (401408, 9), (3 << 20, 3)):
        table = rng.integers(0, 256, (rows, row_bytes), dtype=np.uint8)
        for n, threads in ((1, 1), (7, 3), (25, 16), (0, 4)):
            idx = rng.integers(0, rows, n).astype(np.int64)
            out = np.full((max(n, 1), row_bytes), 0xAB, dtype=np.uint8)
            _lib.check(lib.vqa_gather_rows(table.ctypes.data, rows, row_bytes, idx.ctypes.data, n, out.ctypes.data, threads))
            assert np.array_equal(out[:n], table[idx])
            if n == 0:
                assert (out == 0xAB).all()
    table = np.zeros((4, 8), dtype=np.uint8)
    out = np.full((2, 8), 7, dtype=np.uint8)
    for bad in (-1, 4):
        idx = np.array([1, bad], dtype=np.int64)
        assert lib.vqa_gather_rows(table.ctypes.data, 4, 8, idx.ctypes.data, 2, out.ctypes.data, 2) == 1      # VQA_EINVAL
        assert b'outside the table' in lib.vqa_last_error()
        assert (out == 7).all()


@pytest.mark.gpu
def test_resident_rows_and_channels_last_match_fp32_batches(tmp_path):
    """Same logits and same weight gradients, bit for bit, whether the executor gets the loader's fp32
    NCHW batch, rows of the resident table, or the prefetcher's bf16 channels-last tensor."""
    from vqa_amd.module_net import ModuleNet
    vocab = V.make_vocab()
    d, names, maps = _files(tmp_path, 12)
    shard = feed.pack_reference_features(d, names, str(tmp_path / 'f.vqaf'))
    dev = torch.device('cuda')
    net = ModuleNet(False, vocab)
    net.load_state_dict(synth.synth_state_dict(17, vocab))
    net.to(dev).train()
    rows = [7, 0, 3, 3, 11, 5, 1, 9]
    prog = torch.from_numpy(synth.synth_programs(21, len(rows), vocab)).to(dev)
    ans = torch.from_numpy(synth.synth_answers(21, len(rows))).to(dev)
    res = feed.ResidentFeatures(shard, dev)
    assert res.count == 12 and torch.equal(res.table.cpu().view(torch.int16), torch.from_numpy(np.ascontiguousarray(shard.rows[:])))
    with pytest.raises(IndexError):
        res.rows([0, 12])

    def run(x):
        for p in net.parameters():
            p.grad = None
        out = net(x, prog)
        torch.nn.functional.cross_entropy(out, ans).backward()
        return out.detach().cpu(), {n: p.grad.detach().cpu().clone() for n, p in net.named_parameters() if p.grad is not None}

    base, gbase = run(maps[rows].to(dev))
    variants = {
        'resident rows': res.rows(rows),
        'resident rows by name': res.rows_of([names[r] for r in rows]),
        'resident rows, device index': res.rows(torch.tensor(rows, device=dev)),
        'channels-last bf16': next(feed.HostFeed(shard, [rows], dev)),
    }
    assert variants['channels-last bf16'].shape == (len(rows), 1024, 14, 14)
    for name, x in variants.items():
        out, g = run(x)
        assert torch.equal(out, base), name
        assert g.keys() == gbase.keys()
        for k in g:
            if k.endswith('.bias'):     # bias gradients are fp32 atomic sums: order varies run to run
                torch.testing.assert_close(g[k], gbase[k], rtol=1e-4, atol=1e-6, msg=str((name, k)))
            else:
                assert torch.equal(g[k], gbase[k]), (name, k)
    assert torch.equal(res.rows(rows).dense().cpu(), shard.batch_fp32(rows))
    with torch.no_grad():
        net.eval()
        assert torch.equal(net(res.rows(rows), prog).cpu(), net(maps[rows].to(dev), prog).cpu())


@pytest.mark.gpu
def test_host_feed_order_reuse_and_errors(tmp_path):
    d, names, maps = _files(tmp_path, 9, c=128, seed=9)
    shard = feed.pack_reference_features(d, names, str(tmp_path / 's.vqaf'), channels=128)
    dev = torch.device('cuda')
    rng = np.random.RandomState(3)
    batches = [rng.randint(0, 9, size=rng.randint(1, 7)).tolist() for _ in range(11)]
    want = maps.to(torch.bfloat16)
    got = []
    hf = feed.HostFeed(shard, batches, dev, depth=3)
    for b, x in zip(batches, hf):
        assert x.dtype == torch.bfloat16 and x.shape == (len(b), 128, 14, 14)
        assert x.stride() == (196 * 128, 1, 14 * 128, 128)
        got.append(x.cpu())          # a copy on the consumer's stream, before the buffer is reused
    assert len(got) == len(batches)
    for b, x in zip(batches, got):
        assert torch.equal(x, want[b])
    with pytest.raises(StopIteration):
        next(hf)
    hf.close()
    bad = feed.HostFeed(shard, [[0, 1], [0, 99]], dev)
    next(bad)
    with pytest.raises(IndexError):
        next(bad)
    bad.close()
