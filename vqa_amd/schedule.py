"""Host wrapper of the program scheduler in libvqa_amd.so (vqa_schedule_build).

The scheduler replaces the per-sample recursion of `forward_modules_ints_helper`
(`Module_Model.py:196-222`): it parses every prefix program with the reference's rules and buckets
the tree nodes of the whole batch by (level, token).
"""
import ctypes as C

import numpy as np

from . import _lib, vocab as V


# arity table per vocabulary object (rebuilt if its token table is replaced or grows)
_ARITY = {}


class Plan:
    """Parsed + bucketed programs of one batch.  Owns the native plan handle."""

    def __init__(self, programs, vocab, module_dim=128, wgrad_chunk=16, flags=0, image_index=None, num_images=None,
                 blob_store=None):
        """image_index: optional int array [N], the image each sample runs on; samples of one image
        then share common subtrees and the plan is forward-only (vqa_schedule_build_shared).
        blob_store: optional int32 host tensor (pinned, say) that the tile lists are written into if
        it is large enough (`blob_store_words`); `info.blob_lent` tells, and the tensor must then
        outlive the plan."""
        lib = _lib.load()
        prog = np.ascontiguousarray(np.asarray(programs), dtype=np.int64)
        if prog.ndim != 2:
            raise ValueError('Unregonized program format')      # Module_Model.py:245
        t2i = vocab['program_token_to_idx']
        cached = _ARITY.get(id(vocab))
        if cached is None or cached[0] is not t2i or len(t2i) != len(cached[1]):
            cached = _ARITY[id(vocab)] = (t2i, np.asarray(V.arity_table(vocab), dtype=np.int8))
        arity = cached[1]
        self._lib = lib
        self._h = C.c_void_p()
        n, t = prog.shape
        img, m = None, n
        if image_index is not None:
            img = np.ascontiguousarray(np.asarray(image_index), dtype=np.int32)
            if img.shape != (n,):
                raise ValueError('image_index must have one entry per program')
            m = int(num_images) if num_images is not None else int(img.max()) + 1
        store, store_words = None, 0
        if blob_store is not None:
            assert blob_store.dtype.itemsize == 4 and blob_store.is_contiguous()
            store, store_words = C.cast(blob_store.data_ptr(), _lib.i32p), blob_store.numel()
        self._store = blob_store
        _lib.check(lib.vqa_schedule_build_into(
            prog.ctypes.data_as(_lib.i64p), n, t, arity.ctypes.data_as(_lib.i8p), len(arity),
            t2i['<NULL>'], t2i['<START>'], t2i['scene'], module_dim, wgrad_chunk, flags,
            img.ctypes.data_as(_lib.i32p) if img is not None else None, m, store, store_words,
            C.byref(self._h)))
        self.info = _lib.PlanInfo()
        _lib.check(lib.vqa_plan_get_info(self._h, C.byref(self.info)))
        self.N, self.T, self.C = n, t, module_dim
        self.V = len(arity)

    def __del__(self):
        h = getattr(self, '_h', None)
        if h is not None and h.value:
            self._lib.vqa_plan_free(h)
            self._h = C.c_void_p()

    @property
    def handle(self):
        return self._h

    @staticmethod
    def blob_store_words(n, t, vocab_size):
        """Words that a blob_store needs for any n programs of t tokens."""
        return n * (t + 1) * 88 + n * 64 + n * (t + 1) // 2 * 32 + vocab_size * 32 + 1024

    def used_fns(self):
        out = np.empty((self.N, self.T), dtype=np.float32)
        _lib.check(self._lib.vqa_plan_used_fns(self._h, out.ctypes.data_as(_lib.f32p)))
        return out

    def used_fns_into(self, host_tensor):
        """Write used_fns into a caller-provided float32 host tensor (e.g. pinned memory)."""
        assert host_tensor.numel() >= self.N * self.T and host_tensor.dtype.is_floating_point
        _lib.check(self._lib.vqa_plan_used_fns(self._h, C.cast(host_tensor.data_ptr(), _lib.f32p)))

    def nodes(self):
        """int32 [num_nodes, 6]: sample, token, column, child0, child1, level."""
        out = np.empty((self.info.num_nodes, 6), dtype=np.int32)
        _lib.check(self._lib.vqa_plan_nodes(self._h, out.ctypes.data_as(_lib.i32p)))
        return out

    def level_nodes(self, level):
        cnt = self._lib.vqa_plan_level_nodes(self._h, level, None, 0)
        out = np.empty(max(cnt, 1), dtype=np.int32)
        self._lib.vqa_plan_level_nodes(self._h, level, out.ctypes.data_as(_lib.i32p), cnt)
        return out[:cnt]

    def token_counts(self):
        out = np.empty(self.V, dtype=np.int32)
        _lib.check(self._lib.vqa_plan_token_counts(self._h, out.ctypes.data_as(_lib.i32p)))
        return out

    def roots(self):
        out = np.empty(self.N, dtype=np.int32)
        _lib.check(self._lib.vqa_plan_roots(self._h, out.ctypes.data_as(_lib.i32p)))
        return out

    def blob(self):
        out = np.empty(self.info.blob_words, dtype=np.int32)
        _lib.check(self._lib.vqa_plan_blob(self._h, out.ctypes.data_as(_lib.i32p)))
        return out

    def blob_into(self, host_tensor):
        """Write the blob into a caller-provided int32 host tensor (e.g. pinned memory)."""
        assert host_tensor.numel() >= self.info.blob_words
        _lib.check(self._lib.vqa_plan_blob(self._h, C.cast(host_tensor.data_ptr(), _lib.i32p)))

    def finish(self):
        """Builds the backward part of a plan made with PLAN_DEFER_BACKWARD and refreshes `info`."""
        _lib.check(self._lib.vqa_plan_finish(self._h))
        _lib.check(self._lib.vqa_plan_get_info(self._h, C.byref(self.info)))

    def blob_range_into(self, host_tensor, first, count):
        """Words [first, first + count) of the blob into the same positions of an int32 host tensor."""
        assert host_tensor.numel() >= first + count and host_tensor.dtype.itemsize == 4
        _lib.check(self._lib.vqa_plan_blob_range(self._h, first, count,
                                                 C.cast(host_tensor.data_ptr() + 4 * first, _lib.i32p)))

    def section(self, name):
        off, cnt = C.c_int32(), C.c_int32()
        _lib.check(self._lib.vqa_plan_section(self._h, _lib.SEC[name], C.byref(off), C.byref(cnt)))
        return off.value, cnt.value

    def slot(self, which, index):
        names = {'F': 0, 'S0': 1, 'Y': 2, 'H': 3, 'dZ': 4, 'dH': 5, 'dF': 6, 'dS0': 7, 'P': 8}
        return self._lib.vqa_plan_slot(self._h, names[which], index)

    def flops(self, feature_dim=(1024, 14, 14), proj_dim=512, fc_dims=(1024,), num_answers=32, executed=False):
        """Algorithmic FLOPs (2 per multiply-add) of forward and forward+backward for this
        batch: stem + parsed tree nodes + classifier; elementwise work is not counted and the
        input features get no gradient.  executed=True counts what the device runs instead of what
        the reference's per-node execution does (repeated scene nodes of a sample run once)."""
        hw, c, cin = feature_dim[1] * feature_dim[2], self.C, feature_dim[0]
        i = self.info
        f_stem0 = 2 * hw * 9 * cin * c
        f_stem1 = 2 * hw * 9 * c * c
        f_u = 36 * hw * c * c
        f_b = f_u + 4 * hw * c * c
        prev = c * hw
        f_cls = 0
        if proj_dim:
            f_cls += 2 * hw * c * proj_dim
            prev = proj_dim * hw
        for f in fc_dims:
            f_cls += 2 * prev * f
            prev = f
        f_cls += 2 * prev * num_answers
        scenes = i.num_scene
        if executed:        # the scene nodes of a sample are one computation
            scenes -= i.num_nodes - i.num_exec_nodes
        fwd = self.N * (f_stem0 + f_stem1 + f_cls) + (scenes + i.num_unary) * f_u + i.num_binary * f_b
        return fwd, 3 * fwd - self.N * f_stem0
