"""Data-parallel training of the executor: one process per GPU, RCCL gradient all-reduce.

Replaces `torch.nn.DataParallel(ee)` (`Functions.py:122-126`), which re-broadcasts all 117 M
parameters to every replica on every forward and reduces gradients to device 0.  Here every rank
keeps a persistent replica, owns rows [r*N/G, (r+1)*N/G) of the batch (the same split
DataParallel's scatter makes), and the only collective on the data path is the gradient
all-reduce:

  * the first linear layer's gradient (411 MB of the 470 MB) is produced first in backward; its
    all-reduce is launched the moment its kernel is enqueued and runs on RCCL's stream under the
    whole module-tree backward
  * the per-token modules' gradients leave the executor as rows of a few [V, ...] arrays (one per
    parameter kind): the large ones are reduced in place, array by array, as soon as they are
    complete; a token that no program of this rank used has a row of zeros, so every rank issues
    identical collectives whatever its programs were
  * what is left (stem, classifier conv, biases: a dozen tensors) goes into one flat fp32 bucket
    with a fixed layout, reduced when backward ends
  * a parameter keeps `grad is None` only if NO rank used its token, which is what the
    reference's reduce-add over replicas gives (Adam then skips it, weight decay included)

The wrapper works on any backend torch.distributed offers (tests run it on gloo/CPU with a
stand-in module); on MI355X the backend is "nccl", i.e. RCCL over xGMI.
"""
import os
import sys

import torch
import torch.distributed as dist


def shard_rows(n, rank, world):
    """Row range of rank `rank` when `n` samples are split over `world` ranks as
    torch.nn.DataParallel's scatter does (ceil-sized chunks)."""
    per = -(-n // world)
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def loss_weight(n, rank, world):
    """Factor for a rank's mean loss so that the averaged gradients are those of the mean loss over
    all `n` samples, as the reference computes it on DataParallel's gathered outputs
    (Train_model.py:312-314): n_rank * world / n.  It is 1 when `world` divides `n`."""
    lo, hi = shard_rows(n, rank, world)
    return (hi - lo) * world / float(n)


class DataParallelExecutor:
    def __init__(self, net, process_group=None, direct_bytes=32 << 20, big_transport='fp32', big_algo='all_reduce',
                 shard_big_update=False):
        """shard_big_update (with vqa_amd.optim.FusedAdam(executor=net)): the first linear layer's
        gradient is reduce-scattered by rows instead of all-reduced, each rank runs Adam on its 1/N of
        that layer (N ranks updating the same 103 M parameters N times over is the replicated step's
        largest non-MFMA cost), and what is all-gathered is the updated bf16 copy the kernels read -
        under the next forward, which only needs it at the classifier.  Same arithmetic as the
        replicated update.  The fp32 master copy, its gradient and its Adam state are then current on
        the owning rank only: gather_sharded_state() before reading or saving them.

        big_transport / big_algo apply to gradients of at least `direct_bytes` (the first linear
        layer's 411 MB weight gradient, 87 % of all gradient bytes) when they are not sharded:
          big_transport 'fp32' (default: N ranks give the gradients of one rank on the whole batch to
              fp32 rounding) or 'bf16' (the gradient crosses the links as bf16 - half the wire time,
              within the 2e-2 gradient tolerance of the bf16 path, no longer bit-near to one rank);
          big_algo 'all_reduce' (default) or 'reduce_scatter': an explicit reduce-scatter into this
              rank's 1/N slice followed by an all-gather, the two halves of a ring all-reduce as
              separate collectives (on the xGMI mesh each moves (N-1)/N of the bytes over all links
              at once; the split is also where a sharded optimizer step would go)."""
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        if big_transport not in ('fp32', 'bf16') or big_algo not in ('all_reduce', 'reduce_scatter'):
            raise ValueError('big_transport must be fp32 or bf16, big_algo all_reduce or reduce_scatter')
        self.big_transport, self.big_algo = big_transport, big_algo
        self.shard_big_update = bool(shard_big_update)
        self._shard_bufs = {}        # name -> persistent fp32 buffer of this rank's slice of the gradient
        self._sharded = {}           # name -> that buffer, for gradients reduce-scattered this step
        self._big_bufs = {}          # name -> persistent (bf16 staging, shard) buffers
        self._big_after = []         # (fp32 gradient, bf16 staging) to cast back once the collectives are done
        self.net = net
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.direct_bytes = direct_bytes
        self._params = list(net._param_groups())
        self._names = [n for n, _ in self._params]
        self._token_of = {}
        toks = getattr(net, '_tokens', [])
        for ti, tok in enumerate(toks):
            for n in self._names:
                if n.startswith(tok + '.'):
                    self._token_of[n] = ti
        self._by_name = dict(self._params)
        self._flat = None
        self._layout = None
        self._zeros = None
        self._groups = []            # (array, {name: row}) handed over by the executor this step
        self._grouped_names = set()  # names that arrive as rows of such arrays (fixed after the first step)
        self._row_tables = {}        # id(name -> row dict) -> (that dict, [(parameter, token index, row)])
        self._rows_checked = {}      # id(name -> row dict) -> (array address, tokens used here, tokens used anywhere) last checked
        self._ungrouped = None       # (len(_grouped_names) it was made for, [(name, parameter)] outside the arrays)
        self._small_groups = []
        self._works = []
        self._direct = {}
        self._small = {}
        self._presence = None
        self._pending = False
        self._backward_done = False  # a backward pass has run since the last finish_grad_sync()
        net.grad_ready_hook = self._on_grad
        net.grad_group_hook = self._on_group
        net.plan_ready_hook = self._on_plan
        self._aux, self._presence_host, self._presence_event, self._presence_launched = None, None, None, False
        self._keep_big = None
        self._moments_stale = False
        net.backward_done_hook = self._on_backward_done
        self._avg = dist.get_backend(process_group) == 'nccl'

    # ---------------------------------------------------------------- forward
    def __call__(self, x, program):
        return self.forward(x, program)

    def forward(self, x, program):
        # (_presence_launched belongs to the pending backward: an eval / no_grad forward between a
        # training forward and finish_grad_sync() - the MAPO scoring pattern, Train_model.py:339-369 -
        # leaves it alone; finish_grad_sync() clears it)
        out = self.net(x, program)
        counts = getattr(self.net, 'last_token_counts', None)
        if counts is not None and torch.is_grad_enabled() and not self._presence_launched:
            self._presence = torch.as_tensor(counts, dtype=torch.int32).clamp(max=1).to(out.device)
            self._works.append(dist.all_reduce(self._presence, op=dist.ReduceOp.MAX, group=self.pg, async_op=True))
        return out

    def _on_plan(self, counts, device, need_grad):
        """Called by the executor as soon as the batch's programs are parsed (before its forward
        launches): which tokens any rank uses is agreed on here, on a stream of its own, and is in
        host memory long before finish_grad_sync() asks for it - the host never waits for the
        device's backward pass to learn it."""
        if not need_grad or device.type != 'cuda':      # (grad mode is off inside the executor's autograd function)
            return
        if self._aux is None:
            self._aux = torch.cuda.Stream(device)
            self._presence_host = torch.zeros(len(counts), dtype=torch.int32).pin_memory()
        if self._presence_event is not None:
            self._presence_event.synchronize()       # the pinned buffer's previous use
        with torch.cuda.stream(self._aux):
            self._presence_host.copy_(torch.as_tensor(counts, dtype=torch.int32).clamp(max=1))
            self._presence = self._presence_host.to(device, non_blocking=True)
            work = dist.all_reduce(self._presence, op=dist.ReduceOp.MAX, group=self.pg, async_op=True)
            work.wait()                              # the aux stream waits, not the host
            self._presence_host.copy_(self._presence, non_blocking=True)
            self._presence_event = torch.cuda.Event()
            self._presence_event.record(self._aux)
        self._presence_launched = True

    # ---------------------------------------------------------------- backward hooks
    def _reduce(self, t):
        if os.environ.get('VQA_DP_DEBUG'):
            print('[dp rank %d] issue collective %d: %d elements on stream %x' % (
                self.rank, len(self._works), t.numel(), torch.cuda.current_stream().cuda_stream if t.is_cuda else 0),
                file=sys.stderr, flush=True)
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        self._works.append(dist.all_reduce(t, op=op, group=self.pg, async_op=True))

    def _reduce_big(self, name, grad):
        """The collective(s) for one large gradient, issued on the current stream: optionally through
        a bf16 copy, optionally as reduce-scatter + all-gather.  The result is in `grad` (fp32) once
        finish_grad_sync() has run."""
        if self.big_transport == 'fp32' and self.big_algo == 'all_reduce':
            return self._reduce(grad)
        flat = grad.view(-1)
        n = flat.numel()
        bufs = self._big_bufs.get(name)
        if bufs is None or bufs[0] != (n, grad.device):
            stage = torch.empty(n, dtype=torch.bfloat16, device=grad.device) if self.big_transport == 'bf16' else None
            shard = (torch.empty(n // self.world, dtype=torch.bfloat16 if stage is not None else torch.float32, device=grad.device)
                     if self.big_algo == 'reduce_scatter' and n % self.world == 0 else None)
            bufs = self._big_bufs[name] = ((n, grad.device), stage, shard)
        _, stage, shard = bufs
        src = flat
        if stage is not None:
            stage.copy_(flat)                # fp32 -> bf16 on the current stream
            src = stage
            self._big_after.append((flat, stage))
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        if shard is None:
            self._works.append(dist.all_reduce(src, op=op, group=self.pg, async_op=True))
            return
        w = dist.reduce_scatter_tensor(shard, src, op=op, group=self.pg, async_op=True)
        w.wait()       # RCCL: the current stream waits, the host does not; the all-gather reads the slice
        self._works.append(dist.all_gather_into_tensor(src, shard, group=self.pg, async_op=True))

    def _has_slicing_optimizer(self):
        ref = getattr(self.net, '_sliced_update_optimizer', None)
        return ref is not None and ref() is not None

    def _shardable(self, name, grad):
        """Only the executor's first linear layer on the bf16 path, and only when an optimizer that
        can run its update on a slice of rows has said so (vqa_amd.optim.FusedAdam(executor=net)
        registers itself on the executor): with any other optimizer the slice would never be
        applied, so the gradient takes the replicated all-reduce instead."""
        net = self.net
        idx = getattr(net, '_fc_idx', None)
        return (self.shard_big_update and idx is not None and
                getattr(net, 'precision', None) == 'bf16' and self._has_slicing_optimizer() and
                name == 'classifier.%d.weight' % idx[0] and grad.dim() == 2 and grad.shape[0] % self.world == 0 and
                grad.dtype == torch.float32)

    def _one_backward_per_sync(self):
        """torch.nn.DataParallel lets a caller run several backward passes before the optimizer step and
        sums them; here the gradients of a pass live in the executor's own arrays and are reduced in
        place while the pass runs, so a second pass would overwrite the first one's.  Refused loudly,
        before anything of the second pass is issued (every rank takes the same branch)."""
        if self._backward_done:
            raise RuntimeError('DataParallelExecutor: a second backward pass before finish_grad_sync(); gradient '
                               'accumulation over several passes is not supported - call finish_grad_sync() and the '
                               'optimizer step after every backward pass, or sum the losses and run one pass')

    def _on_grad(self, name, grad):
        self._one_backward_per_sync()
        self._pending = True
        if grad.numel() * grad.element_size() >= self.direct_bytes and grad.is_contiguous():
            if self._shardable(name, grad):
                if getattr(self.net, '_fc1_grad_shard', None) is not None:
                    raise RuntimeError("DataParallelExecutor(shard_big_update=True): the previous step's rows of the first "
                                       "linear layer's gradient were never applied - FusedAdam(executor=net).step() has "
                                       'to run between finish_grad_sync() and the next backward pass')
                n = grad.numel() // self.world
                shard = self._shard_bufs.get(name)
                if shard is None or shard.numel() != n or shard.device != grad.device:
                    shard = self._shard_bufs[name] = torch.empty(n, dtype=torch.float32, device=grad.device)
                op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
                self._works.append(dist.reduce_scatter_tensor(shard, grad.view(-1), op=op, group=self.pg, async_op=True))
                self._sharded[name] = shard
                self._keep_big = grad            # read by the collective until finish_grad_sync() has waited
                return
            self._direct[name] = grad
            self._reduce_big(name, grad)     # in place, overlapped with the rest of backward
        else:
            self._small[name] = grad

    def _on_group(self, array, rows):
        """`array` [V, ...] holds the gradients of one parameter kind of every token (zeros where a
        token was not used); rows maps parameter name -> row."""
        self._one_backward_per_sync()
        self._pending = True
        if self._flat is not None and not self._grouped_names.issuperset(rows):
            self._flat = None                # first sight of these names after the bucket was laid out
        self._grouped_names.update(rows)
        if array.numel() * array.element_size() >= (1 << 20) and array.is_contiguous():
            self._groups.append((array, rows))
            self._reduce(array)              # in place; the parameters' .grad are views of it
        else:
            self._small_groups.append((array, rows))

    def _ensure_flat(self, device):
        if self._flat is not None and self._flat.device == device:
            return
        layout, off = {}, 0
        for n, p in self._params:
            if p.numel() * 4 >= self.direct_bytes or n in self._grouped_names:
                continue
            layout[n] = (off, p.numel())
            off += p.numel()
        for k, (array, _) in enumerate(self._small_groups):
            layout[('group', k)] = (off, array.numel())
            off += array.numel()
        self._layout = layout
        self._flat = torch.zeros(off, dtype=torch.float32, device=device)

    def _on_backward_done(self):
        self._backward_done = True
        dev = next(iter(self._small.values())).device if self._small else self._params[0][1].device
        self._ensure_flat(dev)
        # the bucket in one launch: the pieces in layout order, zeros where this step has no gradient
        parts = []
        for key, (off, cnt) in self._layout.items():
            if isinstance(key, tuple):
                g = self._small_groups[key[1]][0]
            else:
                g = self._small.get(key)
            if g is None:
                z = self._zeros
                if z is None or z.numel() < cnt or z.device != dev:
                    z = self._zeros = torch.zeros(max(cnt, 1 << 16), dtype=torch.float32, device=dev)
                g = z[:cnt]
            parts.append(g.reshape(-1))
        if parts:
            torch.cat(parts, out=self._flat)
        self._reduce(self._flat)

    # ---------------------------------------------------------------- after loss.backward()
    def finish_grad_sync(self):
        """Wait for the collectives and install the averaged gradients on the parameters."""
        if not self._pending:
            return
        dbg = os.environ.get('VQA_DP_DEBUG')
        for i, w in enumerate(self._works):
            if dbg:
                print('[dp rank %d] waiting for collective %d of %d' % (self.rank, i, len(self._works)), file=sys.stderr, flush=True)
            w.wait()
        if dbg:
            print('[dp rank %d] all collectives done' % self.rank, file=sys.stderr, flush=True)
        self._works = []
        for flat, stage in self._big_after:
            flat.copy_(stage)                # bf16 -> fp32, behind the waits above on this stream
        self._big_after = []
        present = None
        if self._presence_launched:
            self._presence_event.synchronize()
            present = self._presence_host.tolist()
        elif self._presence is not None:
            present = self._presence.cpu().tolist()
        scale = 1.0 if self._avg else 1.0 / self.world
        if not self._avg:
            self._flat.mul_(scale)
        groups = list(self._groups)
        if not self._avg:
            for array, _ in groups:
                array.mul_(scale)            # reduced in place as sums
        for k, (array, rows) in enumerate(self._small_groups):
            # these travelled in the flat bucket: the averages go back into the array, whose rows the
            # parameters already have as .grad (one small copy instead of a new view per parameter)
            off, cnt = self._layout[('group', k)]
            array.copy_(self._flat[off:off + cnt].view(array.shape))
            groups.append((array, rows))
        # A parameter whose token this rank used already has its row of the array as .grad (autograd
        # installed it; the reduction was in place): only the others are touched.  The executor hands
        # over the same name -> row dict every step, so the (parameter, token, row) table of a group is
        # made once; per parameter and step this loop is then an attribute read and a pointer compare.
        by_name, token_of = self._by_name, self._token_of
        counts = getattr(self.net, 'last_token_counts', None)
        local = tuple(1 if c else 0 for c in counts) if counts is not None else None
        present = tuple(present) if present is not None else None
        for array, rows in groups:
            ent = self._row_tables.get(id(rows))
            if ent is None or ent[0] is not rows:
                ent = self._row_tables[id(rows)] = (rows, [(by_name[n], token_of[n], row) for n, row in rows.items()])
            base, row_bytes = array.data_ptr(), array[0].numel() * array.element_size()
            # Same array address, same tokens present on this rank and anywhere as in a step whose
            # .grad tensors were checked one by one: autograd installs the rows the same way (whether it
            # keeps the row view it is handed depends on the code path, not on the data), nothing to do.
            # (only when every token used anywhere is used here: a parameter whose token only another rank
            # used has no .grad from autograd and gets its row below, every step)
            key = (base, local, present)
            if local is not None and local == present and self._rows_checked.get(id(rows)) == key:
                continue
            self._rows_checked[id(rows)] = key
            for p, ti, row in ent[1]:
                if present is not None and not present[ti]:
                    if p.grad is not None:
                        p.grad = None         # unused on every rank
                    continue
                g = p.grad
                if g is None or g.data_ptr() != base + row * row_bytes:
                    p.grad = array[row].view(p.shape)
        if self._ungrouped is None or self._ungrouped[0] != len(self._grouped_names):
            self._ungrouped = (len(self._grouped_names), [(n, p) for n, p in self._params if n not in self._grouped_names])
        for n, p in self._ungrouped[1]:
            if n in self._sharded:
                # this rank's rows of the averaged gradient, for FusedAdam's update of the same rows
                shard = self._sharded[n]
                if not self._avg:
                    shard.mul_(scale)
                p.grad = None
                self.net._fc1_grad_shard = (shard, self.rank, self.world, self)
                continue
            if n in self._direct:
                g = self._direct[n]
                if not self._avg:
                    g.mul_(scale)
                p.grad = g.reshape(p.shape)
                continue
            if n not in self._layout:
                continue
            ti = self._token_of.get(n)
            if ti is not None and present is not None and not present[ti]:
                p.grad = None                 # unused on every rank
                continue
            off, cnt = self._layout[n]
            p.grad = self._flat[off:off + cnt].view(p.shape)
        self._direct, self._small, self._presence, self._pending = {}, {}, None, False
        self._sharded, self._keep_big = {}, None
        self._groups, self._small_groups = [], []
        self._presence_launched = False
        self._backward_done = False

    # ---------------------------------------------------------------- sharded update of the first linear layer
    def gather_big_packed(self, packed):
        """Called by FusedAdam after it updated this rank's rows: every rank's rows of the bf16 copy
        [rows][...] are exchanged.  Not waited for here: the executor waits before its next use of
        that copy (ModuleNet._wait_fc1_gather), with the stem and the module tree in between."""
        flat = packed.view(-1)
        n = flat.numel() // self.world
        piece = flat[self.rank * n:(self.rank + 1) * n]
        self.net._fc1_gather = dist.all_gather_into_tensor(flat, piece, group=self.pg, async_op=True)
        self.net._fc1_master_stale = self._moments_stale = True

    def gather_sharded_state(self, optimizer=None):
        """All ranks: bring the fp32 master copy of the sharded layer (and, given the optimizer, its
        Adam moments) up to date everywhere - before state_dict(), a checkpoint, or any other reader."""
        net = self.net
        tensors = []
        if getattr(net, '_fc1_master_stale', False):
            p = net.classifier[net._fc_idx[0]].weight
            tensors.append(p.data)
            net._fc1_master_stale = False
        if optimizer is not None and self._moments_stale:
            p = net.classifier[net._fc_idx[0]].weight
            if p in optimizer.state:
                tensors += [optimizer.state[p]['exp_avg'], optimizer.state[p]['exp_avg_sq']]
            self._moments_stale = False
        for t in tensors:
            flat = t.view(-1)
            n = flat.numel() // self.world
            dist.all_gather_into_tensor(flat, flat[self.rank * n:(self.rank + 1) * n].clone(), group=self.pg)

    # convenience passthroughs
    def parameters(self):
        return self.net.parameters()

    def state_dict(self):
        self.gather_sharded_state()
        return self.net.state_dict()

    def train(self, mode=True):
        self.net.train(mode)
        return self

    def eval(self):
        self.net.eval()
        return self
