"""What one executor call keeps between its forward and its backward, and the caller-owned device
buffers of the C ABI that calls reuse: shared by the bf16 path (module_net.py) and the fp32 path
(fp32_path.py)."""
import torch

HW = 196


def _rup(x, m):
    return (x + m - 1) // m * m


class _Workspace:
    """One set of the caller-owned device buffers of the C ABI: grow-only, reused call after call,
    so a steady training loop allocates nothing here."""

    def __init__(self):
        self.bufs = {}
        self.busy = False
        self.host_event = None    # last transfer through this set's pinned host buffers

    def get(self, name, numel, dtype, device, zero=False):
        t = self.bufs.get(name)
        if t is None or t.numel() < numel or t.dtype != dtype or t.device != device:
            t = torch.empty(numel + numel // 8, dtype=dtype, device=device)   # headroom: sizes vary per batch
            self.bufs[name] = t
        t = t[:numel]
        if zero:
            t.zero_()
        return t

    def get_capacity(self, name, numel, dtype, device):
        """The whole grow-only buffer `name`, at least `numel` long (contents are lost on growth)."""
        t = self.bufs.get(name)
        if t is None or t.numel() < numel or t.dtype != dtype or t.device != device:
            t = self.bufs[name] = torch.empty(numel + numel // 8, dtype=dtype, device=device)
        return t


class _WorkspacePool:
    """A forward whose backward is still pending keeps its workspace; any other forward in between
    (MAPO scoring runs eval forwards between a train forward and its backward,
    Train_model.py:339-369) takes another set.  Steady state: one set."""

    def __init__(self):
        self.sets = []

    def acquire(self):
        for w in self.sets:
            if not w.busy:
                w.busy = True
                return w
        w = _Workspace()
        w.busy = True
        self.sets.append(w)
        return w


class _CallState(dict):
    """Everything one forward leaves for its backward.  Gives the workspace back when backward has
    run, or when the graph is dropped without one."""

    def release(self):
        ws = self.pop('ws', None)
        if ws is not None:
            ws.busy = False
        self['released'] = True

    def __del__(self):
        self.release()
