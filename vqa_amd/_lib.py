"""ctypes binding of libvqa_amd.so (declared in include/vqa_amd.h).

The library is built in-tree by `python -m vqa_amd.build` (hipcc, gfx950 only).  There is no
fallback: if the shared object is missing or a symbol is absent, loading raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libvqa_amd.so')

i32, i64, f32p, vp = C.c_int32, C.c_int64, C.POINTER(C.c_float), C.c_void_p
i32p, i64p, i8p = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int8)


class PlanInfo(C.Structure):
    _fields_ = [(n, i32) for n in ('N', 'T', 'C', 'num_nodes', 'num_scene', 'num_unary',
                                   'num_binary', 'num_levels', 'slots_fwd', 'slots_total',
                                   'blob_words', 'wgrad_items', 'max_level_tiles',
                                   'wgrad_partial_blocks', 'num_images', 'num_exec_nodes',
                                   'blob_fwd_words', 'blob_words_bound', 'blob_lent')]


class ConvParams(C.Structure):
    _fields_ = [('src', vp), ('dst', vp), ('res', vp), ('mask', vp), ('w', vp), ('bias', vp),
                ('tiles', vp), ('src_slot_stride', i64), ('dst_slot_stride', i64),
                ('w_group_stride', i64), ('src_pix_stride', i32), ('dst_pix_stride', i32),
                ('k_chunks', i32), ('n_out', i32), ('pair_chunks', i32), ('pair_out', i32),
                ('flags', i32), ('num_tiles', i32)]


class WgradParams(C.Structure):
    _fields_ = [('src', vp), ('dy', vp), ('items', vp), ('src_slots', vp), ('src1_slots', vp),
                ('dy_slots', vp), ('dw', vp), ('db', vp), ('partial', vp), ('src_slot_stride', i64),
                ('dy_slot_stride', i64), ('dw_group_stride', i64), ('src_pix_stride', i32),
                ('dy_pix_stride', i32), ('k_chunks', i32), ('pair_chunks', i32), ('n_out', i32),
                ('num_items', i32), ('total_nodes', i32), ('bias_block', i32)]


class GemmParams(C.Structure):
    _fields_ = [('a', vp), ('b', vp), ('c', vp), ('mask', vp), ('lda', i64), ('ldb', i64),
                ('ldc', i64), ('M', i32), ('N', i32), ('K', i32), ('a_kmajor', i32),
                ('b_kmajor', i32), ('out_mode', i32), ('split_k', i32)]


class TreeCtx(C.Structure):
    _fields_ = [('arena', vp), ('blob', vp),
                ('w_conv1', vp), ('w_conv2', vp), ('w_proj', vp),
                ('wd_conv1', vp), ('wd_conv2', vp), ('wd_proj', vp),
                ('b_conv1', vp), ('b_conv2', vp), ('b_proj', vp),
                ('g_conv1', vp), ('g_conv2', vp), ('g_proj', vp),
                ('gb_conv1', vp), ('gb_conv2', vp), ('gb_proj', vp), ('wgrad_partial', vp),
                ('C', i32), ('V', i32), ('flags', i32), ('reserved', i32)]


class ResblockParams(C.Structure):
    _fields_ = [('arena', vp), ('w_first', vp), ('w_second', vp), ('w_pre', vp), ('w_unused', vp),
                ('bias_first', vp), ('bias_second', vp), ('bias_pre', vp), ('tiles', vp),
                ('slot_stride', i64), ('w_group_stride', i64), ('w_pre_group_stride', i64),
                ('flags', i32), ('num_tiles', i32), ('num_pre', i32), ('unused0', i32),
                ('phase_stamps', vp), ('level_info', vp)]


class DevschedParams(C.Structure):
    _fields_ = [(n, vp) for n in ('programs', 'arity', 'used', 'node', 'counts', 'hist', 'row_base',
                                  'level_info', 'bin_off', 'bin_cursor', 'tiles', 'cls_tiles', 'error', 'image_of_row')] + \
               [(n, i32) for n in ('N', 'T', 'V', 'null_id', 'start_id', 'scene_id', 'spread', 'num_images')]


SEC = {name: i for i, name in enumerate((
    'STEM0_TILES', 'STEM1_TILES', 'STEM1_DGRAD_TILES', 'CLS_FWD_TILES', 'CLS_DGRAD_TILES',
    'SLOTS_IDENT', 'SLOTS_S0', 'SLOTS_F', 'SLOTS_DF', 'SLOTS_DS0', 'SLOTS_ROOTY', 'SAMPLE_ITEMS',
    'SCENE_OFFSETS', 'SCENE_SRC', 'WG_ITEMS', 'WG_SRC_C1', 'WG_DY_C1', 'WG_SRC_C2', 'WG_DY_C2',
    'WG_PITEMS', 'WG_PSRC0', 'WG_PSRC1', 'WG_PDY', 'SAMPLE_ITEMS_FINE', 'SLOTS_SCDZ', 'SLOTS_SCY',
    'SCENE_DPROJ_TILES'))}

CONV_BIAS, CONV_RELU = 1, 2
TREE_UNFUSED, TREE_F32, TREE_SKIP_LEVEL0 = 1, 2, 4
PLAN_SKIP_SEPARATE, PLAN_SKIP_FUSED, PLAN_BALANCE, PLAN_DEFER_BACKWARD = 1, 2, 4, 8

# name -> (restype, argtypes); every symbol include/vqa_amd.h declares
SIGNATURES = {
    'vqa_version': (i32, []),
    'vqa_last_error': (C.c_char_p, []),
    'vqa_schedule_build': (i32, [i64p, i32, i32, i8p, i32, i32, i32, i32, i32, i32, i32,
                                 C.POINTER(vp)]),
    'vqa_schedule_build_shared': (i32, [i64p, i32, i32, i8p, i32, i32, i32, i32, i32, i32, i32, i32p, i32,
                                        C.POINTER(vp)]),
    'vqa_schedule_build_into': (i32, [i64p, i32, i32, i8p, i32, i32, i32, i32, i32, i32, i32, i32p, i32,
                                      i32p, i64, C.POINTER(vp)]),
    'vqa_plan_free': (None, [vp]),
    'vqa_plan_get_info': (i32, [vp, C.POINTER(PlanInfo)]),
    'vqa_plan_used_fns': (i32, [vp, f32p]),
    'vqa_plan_nodes': (i32, [vp, i32p]),
    'vqa_plan_level_nodes': (i32, [vp, i32, i32p, i32]),
    'vqa_plan_token_counts': (i32, [vp, i32p]),
    'vqa_plan_roots': (i32, [vp, i32p]),
    'vqa_plan_blob': (i32, [vp, i32p]),
    'vqa_plan_blob_range': (i32, [vp, i32, i32, i32p]),
    'vqa_plan_finish': (i32, [vp]),
    'vqa_plan_section': (i32, [vp, i32, i32p, i32p]),
    'vqa_plan_slot': (i32, [vp, i32, i32]),
    'vqa_conv_tiles': (i32, [C.POINTER(ConvParams), i32, vp]),
    'vqa_resblock_tiles': (i32, [C.POINTER(ResblockParams), vp]),
    'vqa_stem_nchw': (i32, [vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, vp]),
    'vqa_devsched_build': (i32, [C.POINTER(DevschedParams), vp]),
    'vqa_conv_tiles_f32': (i32, [C.POINTER(ConvParams), i32, vp]),
    'vqa_wgrad_tiles_f32': (i32, [C.POINTER(WgradParams), i32, vp]),
    'vqa_slot_sum_mask_f32': (i32, [vp, vp, vp, vp, vp, i32, i64, vp]),
    'vqa_pack_conv_weight': (i32, [vp, vp, i32, i32, i32, i32, i64, i64, i64, i32, vp]),
    'vqa_wgrad_tiles': (i32, [C.POINTER(WgradParams), i32, vp]),
    'vqa_wgrad_reduce': (i32, [vp, vp, i32, i32, i32, i32, i32, vp, i64, vp, vp, vp]),
    'vqa_wgrad_tiles_noalloc': (i32, [C.POINTER(WgradParams), i32, i32, vp, i64, vp]),
    'vqa_unpack_conv_grad': (i32, [vp, vp, i32, i32, i32, i32, i64, vp]),
    'vqa_gemm': (i32, [C.POINTER(GemmParams), vp]),
    'vqa_gemm_f32': (i32, [C.POINTER(GemmParams), vp]),
    'vqa_transpose_f32': (i32, [vp, vp, i32, i32, i32, i64, i64, vp]),
    'vqa_nchw_to_nhwc': (i32, [vp, vp, i32, i32, i32, vp]),
    'vqa_nhwc_to_nchw': (i32, [vp, vp, i32, i32, vp]),
    'vqa_cast_pad': (i32, [vp, vp, i32, i32, i32, i32, vp]),
    'vqa_fc_weight_to_nhwc': (i32, [vp, vp, i32, i32, i32, vp]),
    'vqa_fc_grad_from_nhwc': (i32, [vp, vp, i32, i32, vp]),
    'vqa_transpose_hw_c': (i32, [vp, vp, i32, i32, vp]),
    'vqa_bias_act': (i32, [vp, vp, vp, i32, i32, i64, i64, i32, i32, i64, i32, vp]),
    'vqa_col_sum': (i32, [vp, vp, i32, i32, i64, i32, vp]),
    'vqa_slot_sum_mask': (i32, [vp, vp, vp, vp, vp, i32, i64, vp]),
    'vqa_head_forward': (i32, [vp, i64, vp, i64, vp, vp, i32, i32, i32, vp]),
    'vqa_head_backward_workspace': (i64, [i32, i32, i32]),
    'vqa_head_backward': (i32, [vp, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, i32, i32, i32, vp]),
    'vqa_xent': (i32, [vp, vp, vp, vp, vp, i32, i32, vp]),
    'vqa_gather_rows': (i32, [vp, i64, i64, vp, i64, vp, i32]),
    'vqa_adam_tensors': (i32, [vp, vp, i32, C.c_float, C.c_float, C.c_float, C.c_float, vp]),
    'vqa_adam_fc1': (i32, [vp, vp, vp, vp, vp, i32, i32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                         C.c_float, vp]),
    'vqa_adam_conv_pack': (i32, [vp, i32, i32, i32, i32, C.c_float, C.c_float, C.c_float, C.c_float, vp]),
    'vqa_prof_enable': (i32, [i32]),
    'vqa_prof_collect': (i32, [C.POINTER(C.c_double), i32]),
    'vqa_tree_forward': (i32, [vp, C.POINTER(TreeCtx), vp]),
    'vqa_tree_backward': (i32, [vp, C.POINTER(TreeCtx), vp]),
}

_lib = None


class VqaError(RuntimeError):
    pass


def load():
    """Load the shared library (once) and bind every declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: it carries its own HIP runtime, and the library must bind to that one (two
    # runtimes in one process do not share devices or streams)
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError('%s is missing: build it with `python -m vqa_amd.build` '
                          '(there is no CPU fallback)' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    """Raise for a nonzero status: KeyError for a token outside the vocabulary (what the
    reference's dict lookup raises, Module_Model.py:200), RuntimeError otherwise."""
    if rc == 0:
        return
    msg = load().vqa_last_error().decode()
    if rc == 3:
        raise KeyError(msg)
    raise VqaError('libvqa_amd status %d: %s' % (rc, msg))
