"""Feature feed for the executor: packed bf16 feature shards, an HBM-resident table, and a
pinned-memory prefetcher.

The reference's loader reads one 0.8 MB fp32 file per sample with torch.load (DataLoader.py:118-119)
and collates [N, 1024, 14, 14] fp32 batches (DataLoader.py:217-219) that the training loop copies to
the device (Train_model.py:311); the executor then has to turn them into the bf16 channels-last maps
its stem reads.  Here the features are stored once in that final form:

  shard file   rows of bf16 [196 pixels][C channels] (401 408 B at C = 1024, half of the fp32
               file), round-to-nearest-even from the fp32 values - the same rounding the device's
               layout pass applies, so a run fed from a shard computes bit for bit what a run fed
               with the fp32 tensors computes;
  ResidentFeatures   the whole shard in HBM (CLEVR's 70 000 training images are 28 GB of the 288):
               a batch is a list of row numbers, the stem reads the rows in place - no copy, no
               layout pass, no PCIe traffic in the training loop;
  HostFeed     for tables that do not fit: a background thread gathers each batch's rows into pinned
               memory and copies them to the device on its own stream, a few batches ahead; batches
               come out as bf16 channels-last [N, C, 14, 14] tensors, which the executor also reads
               in place.

File layout: 4096-byte header (magic, JSON description, zero padding), the rows, then the image
names, one per line (the reference keys its files by the `image_idxs` strings, DataLoader.py:58,114).
"""
import json
import os
import queue
import threading

import numpy as np
import torch

from . import _lib

MAGIC = b'VQAFEAT1'
HEADER_BYTES = 4096
HW = 196


def _to_rows(feats):
    """fp32 [n, C, 14, 14] (or [C, 14, 14]) -> bf16 [n, 196, C], round to nearest even."""
    t = torch.as_tensor(feats)
    if t.dim() == 3:
        t = t.unsqueeze(0)
    if t.dim() != 4 or t.size(2) * t.size(3) != HW:
        raise ValueError('features must be [n, C, 14, 14]')
    n, c = t.size(0), t.size(1)
    return t.to(torch.float32).reshape(n, c, HW).permute(0, 2, 1).to(torch.bfloat16).contiguous()


class ShardWriter:
    """Writes a shard incrementally: add(name, feats) per image or add_batch(names, feats)."""

    def __init__(self, path, channels=1024):
        self.path, self.channels = path, channels
        self.names = []
        self._f = open(path, 'wb')
        self._f.write(b'\0' * HEADER_BYTES)

    def add_batch(self, names, feats):
        rows = _to_rows(feats)
        if rows.size(2) != self.channels or rows.size(0) != len(names):
            raise ValueError('expected %d maps of %d channels' % (len(names), self.channels))
        self._f.write(rows.view(torch.int16).numpy().tobytes())
        self.names.extend(str(n) for n in names)

    def add(self, name, feats):
        self.add_batch([name], feats)

    def close(self):
        if any('\n' in n for n in self.names):
            raise ValueError('image names must not contain newlines')
        names = '\n'.join(self.names).encode()
        self._f.write(names)
        head = {'count': len(self.names), 'channels': self.channels, 'pixels': HW, 'dtype': 'bf16',
                'layout': 'NHWC', 'names_bytes': len(names)}
        blob = MAGIC + json.dumps(head).encode()
        if len(blob) > HEADER_BYTES:
            raise ValueError('header too long')
        self._f.seek(0)
        self._f.write(blob)
        self._f.close()
        self._f = None

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if self._f is not None:
            if et is None:
                self.close()
            else:
                self._f.close()


def pack_reference_features(feature_path, image_names, out_path, channels=1024):
    """Converts the reference's per-image feature files (`feature_path + name`, each a saved
    [C, 14, 14] tensor, DataLoader.py:118-119) into one shard.  The files are read with
    torch.load(weights_only=True): nothing in them is executed."""
    with ShardWriter(out_path, channels) as w:
        for name in image_names:
            w.add(name, torch.load(feature_path + str(name), map_location='cpu', weights_only=True))
    return FeatureShard(out_path)


class FeatureShard:
    """Memory-mapped reader of a shard: `rows` is an int16 view [count, 196, C] of the bf16 data
    (reinterpret with .view(torch.bfloat16) after torch.from_numpy), `row_of(name)` the row of an
    image."""

    def __init__(self, path):
        with open(path, 'rb') as f:
            head = f.read(HEADER_BYTES)
            if head[:len(MAGIC)] != MAGIC:
                raise ValueError('%s is not a feature shard' % path)
            d = json.loads(head[len(MAGIC):].rstrip(b'\0').decode())
            if d['dtype'] != 'bf16' or d['layout'] != 'NHWC' or d['pixels'] != HW:
                raise ValueError('unsupported shard format: %r' % d)
            self.count, self.channels = d['count'], d['channels']
            f.seek(HEADER_BYTES + self.count * HW * self.channels * 2)
            names = f.read(d['names_bytes']).decode()
        self.path = path
        self.names = names.split('\n') if self.count else []
        if len(self.names) != self.count:
            raise ValueError('shard is truncated')
        self._index = {n: i for i, n in enumerate(self.names)}
        # copy-on-write mapping: nothing is ever written back, and torch can wrap it without a copy
        self.rows = np.memmap(path, dtype=np.int16, mode='c', offset=HEADER_BYTES,
                              shape=(self.count, HW, self.channels))

    def row_of(self, name):
        return self._index[str(name)]

    def rows_of(self, names):
        return np.fromiter((self._index[str(n)] for n in names), dtype=np.int64, count=len(names))

    def batch_fp32(self, rows):
        """The fp32 [n, C, 14, 14] batch the reference's loader would have produced, but holding the
        bf16-rounded values (CPU; for tests and for the fp32 path)."""
        t = torch.from_numpy(np.ascontiguousarray(self.rows[np.asarray(rows, dtype=np.int64)]))
        return t.view(torch.bfloat16).to(torch.float32).permute(0, 2, 1).reshape(-1, self.channels, 14, 14).contiguous()


def _device(device):
    device = torch.device(device)
    if device.type != 'cuda':
        raise RuntimeError('the feature feed delivers to an MI355X: device must be a CUDA/HIP device')
    return device if device.index is not None else torch.device('cuda', torch.cuda.current_device())


class FeatureRows:
    """A batch of feature maps given as rows of a device-resident table: what ResidentFeatures.rows()
    returns and ModuleNet.forward() accepts in place of the [N, C, 14, 14] tensor.  Looks like that
    tensor where the executor's callers look at it (size, shape, is_cuda, device)."""

    def __init__(self, table, index):
        self.table, self.index = table, index     # bf16 [R, 196, C] ; int32 [N], both on the device
        self.shape = torch.Size((index.numel(), table.size(2), 14, 14))
        self.is_cuda, self.device, self.dtype = table.is_cuda, table.device, table.dtype

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def dim(self):
        return 4

    def __len__(self):
        return self.shape[0]

    def dense(self):
        """The batch as an fp32 [N, C, 14, 14] tensor (what the reference's loop would hold)."""
        r = self.table[self.index.long()]
        return r.to(torch.float32).permute(0, 2, 1).reshape(self.shape).contiguous()


class ResidentFeatures:
    """A whole shard (or any [R, C, 14, 14] tensor) held in HBM as bf16 [R, 196, C]."""

    def __init__(self, source, device, chunk_rows=512):
        device = _device(device)
        if isinstance(source, FeatureShard):
            self.names = source.names
            self._index = source._index
            table = torch.empty(source.count, HW, source.channels, dtype=torch.bfloat16, device=device)
            stage = [torch.empty(chunk_rows, HW, source.channels, dtype=torch.int16).pin_memory() for _ in range(2)]
            done = [None, None]
            for k, r0 in enumerate(range(0, source.count, chunk_rows)):
                r1 = min(source.count, r0 + chunk_rows)
                b = k & 1
                if done[b] is not None:
                    done[b].synchronize()
                stage[b][:r1 - r0].copy_(torch.from_numpy(source.rows[r0:r1]))
                table[r0:r1].view(torch.int16).copy_(stage[b][:r1 - r0], non_blocking=True)
                done[b] = torch.cuda.Event()
                done[b].record()
            torch.cuda.current_stream(device).synchronize()
        else:
            self.names, self._index = None, None
            table = _to_rows(torch.as_tensor(source).to(device))
        self.table = table
        self.count = table.size(0)

    def rows(self, index):
        """`index`: row numbers (sequence, numpy array or CPU tensor; checked here, on the host) or
        an int32/int64 device tensor (clamped on the device: the kernels must never see a row
        outside the table)."""
        if torch.is_tensor(index) and index.is_cuda:
            idx = index.to(torch.int32).clamp(0, self.count - 1)
        else:
            a = np.asarray(index, dtype=np.int64).reshape(-1)
            if a.size and (a.min() < 0 or a.max() >= self.count):
                raise IndexError('feature row out of range')
            idx = torch.from_numpy(a.astype(np.int32)).pin_memory().to(self.table.device, non_blocking=True)
        return FeatureRows(self.table, idx.contiguous())

    def rows_of(self, names):
        if self._index is None:
            raise ValueError('this table has no image names')
        return self.rows([self._index[str(n)] for n in names])


class HostFeed:
    """Iterator over batches of a shard that stays in host memory: a background thread gathers the
    rows of the next batches into pinned buffers and copies them to the device on a stream of its
    own, `depth` batches ahead.  Each item is a bf16 [n, C, 14, 14] tensor in channels-last memory
    format, ready on the consumer's current stream, valid until `depth - 1` further items have been
    taken (its buffer is then reused)."""

    def __init__(self, shard, batches, device, depth=3, threads=None):
        self.shard, self.device, self.depth = shard, _device(device), max(2, depth)
        # host threads of the gather: one thread copies 6 GB/s, PCIe moves 50
        self.threads = threads if threads is not None else max(1, min(8, (os.cpu_count() or 2) // 2))
        self._lib = _lib.load()
        self._batches = iter(batches)
        self._src = torch.from_numpy(shard.rows)            # int16 view of the memory map
        self._stream = torch.cuda.Stream(self.device)
        self._free = queue.Queue()
        self._ready = queue.Queue(maxsize=self.depth)
        self._bufs = []
        self._taken = []
        self._stop = False
        for _ in range(self.depth + 1):
            self._free.put(len(self._bufs))
            self._bufs.append({'host': None, 'dev': None, 'consumed': None, 'arrived': None})
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()

    def _run(self):
        try:
            torch.cuda.set_device(self.device)
            for rows in self._batches:
                if self._stop:
                    break
                idx = torch.as_tensor(np.asarray(rows, dtype=np.int64).reshape(-1))
                n = idx.numel()
                if n and (int(idx.min()) < 0 or int(idx.max()) >= self.shard.count):
                    raise IndexError('feature row out of range')
                b = self._free.get()
                if b is None:
                    break
                buf = self._bufs[b]
                if buf['host'] is None or buf['host'].size(0) < n:
                    buf['host'] = torch.empty(n, HW, self.shard.channels, dtype=torch.int16).pin_memory()
                    buf['dev'] = torch.empty(n, HW, self.shard.channels, dtype=torch.int16, device=self.device)
                if buf['arrived'] is not None:
                    buf['arrived'].synchronize()      # the pinned buffer's previous copy has left it
                row_bytes = HW * self.shard.channels * 2
                _lib.check(self._lib.vqa_gather_rows(self._src.data_ptr(), self.shard.count, row_bytes, idx.data_ptr(), n,
                                                     buf['host'].data_ptr(), self.threads))
                with torch.cuda.stream(self._stream):
                    if buf['consumed'] is not None:
                        self._stream.wait_event(buf['consumed'])   # the consumer's kernels are done with it
                    buf['dev'][:n].copy_(buf['host'][:n], non_blocking=True)
                    buf['arrived'] = torch.cuda.Event()
                    buf['arrived'].record(self._stream)
                # not waited for here: the next batch is gathered while this one crosses PCIe
                self._ready.put((b, n, None))
            self._ready.put((None, 0, None))
        except BaseException as e:          # surfaces in the consumer
            self._ready.put((None, 0, e))

    def __iter__(self):
        return self

    def __next__(self):
        b, n, err = self._ready.get()
        if err is not None:
            raise err
        if b is None:
            self._ready.put((None, 0, None))
            raise StopIteration
        # buffers handed out earlier go back once the consumer's stream has passed this point
        self._taken.append(b)
        while len(self._taken) > self.depth - 1:
            old = self._taken.pop(0)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._bufs[old]['consumed'] = ev
            self._free.put(old)
        torch.cuda.current_stream(self.device).wait_event(self._bufs[b]['arrived'])
        dev = self._bufs[b]['dev'][:n].view(torch.bfloat16)
        return dev.view(n, 14, 14, self.shard.channels).permute(0, 3, 1, 2)

    def close(self):
        self._stop = True
        self._free.put(None)
        while self._thread.is_alive():
            try:
                self._ready.get(timeout=0.05)
            except queue.Empty:
                pass
        self._thread.join()
