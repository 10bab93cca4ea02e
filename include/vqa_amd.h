/*
 * vqa_amd.h - C ABI of libvqa_amd.so, the MI355X (gfx950) module executor.
 *
 * The reference (Jacobjohansen90/VQA) has no FFI: its executor is the Python class ModuleNet
 * (Module_Model.py:95-248) calling torch.nn ops.  This library is what a binding for that hot
 * path binds instead; every entry point names the reference lines it replaces.  The Python host
 * (vqa_amd/module_net.py) loads it with ctypes.
 *
 * Conventions
 *   - plain pointers and sizes only; device pointers are raw HIP device addresses
 *   - the caller owns every device buffer, workspaces included; nothing here allocates device
 *     memory (the scheduler allocates host memory for its plan)
 *   - device entry points enqueue work on `stream` and return without waiting
 *   - return 0 on success, nonzero VQA_E* otherwise; vqa_last_error() gives the message of the
 *     calling thread's last failure
 *   - activations are bf16, channels-last: a feature map is [196 pixels][C channels]; a "slot"
 *     is one such map inside the arena; C is a multiple of 128; the image is 14 x 14
 */
#ifndef VQA_AMD_H
#define VQA_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VQA_OK 0
#define VQA_EINVAL 1   /* bad argument (shape, alignment, null pointer) */
#define VQA_ELAUNCH 2  /* HIP reported an error at launch */
#define VQA_EVOCAB 3   /* program token id outside the vocabulary (the reference raises KeyError) */

#define VQA_HW 196     /* 14 x 14 pixels per feature map */

int vqa_version(void);
const char* vqa_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Program scheduler (host).  Replaces the per-sample recursive interpreter
 * forward_modules_ints_helper / forward_modules_ints (Module_Model.py:196-233): parses every
 * prefix program row with the reference's exact rules, then buckets the tree nodes of the whole
 * batch by (tree level, token) so that each level runs as a few grouped launches.
 * ------------------------------------------------------------------------------------------ */
typedef struct vqa_plan vqa_plan;

typedef struct {
    int32_t N, T, C;
    int32_t num_nodes, num_scene, num_unary, num_binary; /* scene incl. implicit; unary incl. specials */
    int32_t num_levels;
    int32_t slots_fwd;      /* arena slots needed for forward only */
    int32_t slots_total;    /* forward + backward */
    int32_t blob_words;     /* int32 words of the device descriptor blob */
    int32_t wgrad_items;    /* work items of the grouped weight-gradient launch */
    int32_t max_level_tiles;
    int32_t wgrad_partial_blocks; /* vqa_tree_backward's workspace, in blocks of 8192 floats */
    int32_t num_images;     /* N, or the image count of vqa_schedule_build_shared */
    int32_t num_exec_nodes; /* nodes that run: all the scene nodes of an image compute the same map, one does */
    int32_t blob_fwd_words; /* leading part of the blob that the forward pass reads */
    int32_t blob_words_bound; /* upper bound of blob_words once the plan is finished (VQA_PLAN_DEFER_BACKWARD) */
    int32_t blob_lent;      /* 1: the blob is in the memory given to vqa_schedule_build_into */
} vqa_plan_info;

/* programs: host int64 [N][T].  arity: int8 [V] (0 scene, 1 unary, 2 binary), indexed by token id
 * (Program_funcs.py:114-122).  null_id / start_id: ids of <NULL> and <START> (Functions.py:44-45).
 * scene_id: id of 'scene'.  C: module width.  wgrad_chunk: nodes per weight-gradient work item.
 * flags: the blob holds, per tree level, one tile list per convolution pass (for vqa_conv_tiles) and
 * the fused per-node lists (for vqa_resblock_tiles); a caller that will only run one of the two
 * paths can leave the other's lists out, which halves the blob. */
#define VQA_PLAN_SKIP_SEPARATE 1
#define VQA_PLAN_SKIP_FUSED 2
/* levels: by default a node's level is 1 + its deepest child's (the earliest it can run).  With
 * VQA_PLAN_BALANCE nodes that have slack are moved to later levels so that the launches of a level
 * come in whole rounds of the device (256 CUs x 2 workgroups); the level count, node ids, slots and
 * results stay the same. */
#define VQA_PLAN_BALANCE 4
/* With VQA_PLAN_DEFER_BACKWARD the build returns once the forward part of the blob (blob_fwd_words,
 * then equal to blob_words) is written: the caller uploads it and enqueues the forward pass, and
 * vqa_plan_finish appends the backward tile lists and sections while the device is busy.  Until
 * then the backward sections have count 0, vqa_tree_backward refuses the plan, and blob_words /
 * wgrad_items / wgrad_partial_blocks describe the forward part only. */
#define VQA_PLAN_DEFER_BACKWARD 8
int vqa_schedule_build(const int64_t* programs, int32_t N, int32_t T, const int8_t* arity,
                       int32_t V, int32_t null_id, int32_t start_id, int32_t scene_id, int32_t C,
                       int32_t wgrad_chunk, int32_t flags, vqa_plan** out);
/* The same for samples that share images (candidate-program scoring, Train_model.py:339-347: one
 * question's K programs run on K copies of one image).  image_of_sample: int32 [N], values in
 * [0, num_images).  Nodes are interned per (image, token, inputs): a subtree that several programs
 * of one image have in common is a single node, computed once; the stem runs once per image
 * (F and S0 slots, the stem tile lists and the node table's `sample` field are per image).
 * Forward only: vqa_tree_backward refuses such a plan. */
int vqa_schedule_build_shared(const int64_t* programs, int32_t N, int32_t T, const int8_t* arity,
                              int32_t V, int32_t null_id, int32_t start_id, int32_t scene_id, int32_t C,
                              int32_t wgrad_chunk, int32_t flags, const int32_t* image_of_sample,
                              int32_t num_images, vqa_plan** out);
/* Either of the two above (image_of_sample NULL and num_images = N for the first), with the blob
 * written straight into memory of the caller's: blob_store, of blob_store_words int32 words, e.g.
 * pinned host memory that an asynchronous copy to the device can read without a staging copy.  The
 * memory is used if it holds the finished blob's upper bound (88 words per node and 32 more per
 * two-input node, of at most N * (T + 1) nodes, plus 64 N + 32 V + 1024), and then belongs to the
 * plan until vqa_plan_free:
 * vqa_plan_finish appends to it, vqa_plan_get_info reports blob_lent = 1 and the blob's words are
 * blob_store[0 .. blob_words).  If it is NULL or too small the plan keeps the blob in memory of its
 * own (blob_lent = 0, vqa_plan_blob / vqa_plan_blob_range copy it out).  No counterpart in the
 * reference, whose helper (Module_Model.py:196-222) runs the modules as it parses. */
int vqa_schedule_build_into(const int64_t* programs, int32_t N, int32_t T, const int8_t* arity,
                            int32_t V, int32_t null_id, int32_t start_id, int32_t scene_id, int32_t C,
                            int32_t wgrad_chunk, int32_t flags, const int32_t* image_of_sample,
                            int32_t num_images, int32_t* blob_store, int64_t blob_store_words,
                            vqa_plan** out);
void vqa_plan_free(vqa_plan* plan);
int vqa_plan_get_info(const vqa_plan* plan, vqa_plan_info* out);
/* used_fns as ModuleNet.used_fns (Module_Model.py:227,231): float [N][T] of 0/1 */
int vqa_plan_used_fns(const vqa_plan* plan, float* out);
/* node table, execution (post-)order per sample, samples in order: int32 [num_nodes][6] =
 * {sample, token, column (-1 implicit scene), child0, child1 (global node ids, -1 none), level} */
int vqa_plan_nodes(const vqa_plan* plan, int32_t* out);
/* nodes of a level in launch order (sorted by token, then node id); returns count */
int vqa_plan_level_nodes(const vqa_plan* plan, int32_t level, int32_t* out, int32_t cap);
/* per-token node counts: int32 [V] */
int vqa_plan_token_counts(const vqa_plan* plan, int32_t* out);
/* root node id of each sample: int32 [N] */
int vqa_plan_roots(const vqa_plan* plan, int32_t* out);
/* the descriptor blob to copy to the device before running the tree (blob_words int32) */
int vqa_plan_blob(const vqa_plan* plan, int32_t* out);
/* words [first, first + count) of the blob (the part vqa_plan_finish appended, say) */
int vqa_plan_blob_range(const vqa_plan* plan, int32_t first, int32_t count, int32_t* out);
/* completes a plan built with VQA_PLAN_DEFER_BACKWARD (no-op on a complete one) */
int vqa_plan_finish(vqa_plan* plan);
/* Sections of the blob that the host needs by name (the per-level tile lists are used by
 * vqa_tree_forward / vqa_tree_backward internally).  offset is in int32 words. */
enum {
    VQA_SEC_STEM0_TILES = 0,   /* conv tiles: x[sample] -> S0(sample) */
    VQA_SEC_STEM1_TILES,       /* S0 -> F */
    VQA_SEC_STEM1_DGRAD_TILES, /* dF -> dS0, mask S0 */
    VQA_SEC_CLS_FWD_TILES,     /* Y(root of sample) -> row `sample` of the classifier input: the
                                  torch.cat of Module_Model.py:232 becomes a gather */
    VQA_SEC_CLS_DGRAD_TILES,   /* row `sample` -> dZ(root), mask Y(root) */
    VQA_SEC_SLOTS_IDENT,       /* 0..N-1 */
    VQA_SEC_SLOTS_S0, VQA_SEC_SLOTS_F, VQA_SEC_SLOTS_DF, VQA_SEC_SLOTS_DS0, VQA_SEC_SLOTS_ROOTY,
    VQA_SEC_SAMPLE_ITEMS,      /* vqa_wgrad_item list cutting 0..N-1 into chunks, group 0 */
    VQA_SEC_SCENE_OFFSETS, VQA_SEC_SCENE_SRC,
    VQA_SEC_WG_ITEMS, VQA_SEC_WG_SRC_C1, VQA_SEC_WG_DY_C1, VQA_SEC_WG_SRC_C2, VQA_SEC_WG_DY_C2,
    VQA_SEC_WG_PITEMS, VQA_SEC_WG_PSRC0, VQA_SEC_WG_PSRC1, VQA_SEC_WG_PDY,
    VQA_SEC_SAMPLE_ITEMS_FINE, /* as SAMPLE_ITEMS in chunks of max(1, min(wgrad_chunk, N/128)) samples */
    VQA_SEC_SLOTS_SCDZ, VQA_SEC_SLOTS_SCY, /* per sample: dZ slot of its scene node that runs; that node's output slot */
    VQA_SEC_SCENE_DPROJ_TILES, /* conv tiles, fused path only: dP(binary node) -> dZ of those of its inputs that
                                  are scene nodes (the other destination is -1), masks = their outputs */
VQA_SEC_WG_ITEMS_12, VQA_SEC_WG_SRC_12, VQA_SEC_WG_DY_12, /* both 3x3 weight gradients of the tree as ONE item list
                                  (longest first; items of the second convolution carry group V + token and index the
                                  second half of the two slot lists): one launch instead of two half-filled tails */
        VQA_SEC_COUNT
};
int vqa_plan_section(const vqa_plan* plan, int32_t id, int32_t* offset, int32_t* count);
/* arena slot ids (slot s is the map at arena + s * 196 * C elements): which = 0 F(sample) 1 S0(sample) 2 Y(node) 3 H(node)
 * 4 dZ(node) 5 dH(node) 6 dF(sample) 7 dS0(sample) 8 P(binary node index); returns slot or -1.
 * Scene nodes of one image share their Y / H / dH slots (one of them runs, see num_exec_nodes). */
int32_t vqa_plan_slot(const vqa_plan* plan, int32_t which, int32_t index);

/* ------------------------------------------------------------------------------------------
 * Grouped convolution over tiles.  One tile = one 14x14 feature map x 128 output channels.
 * Replaces every batch-1 nn.Conv2d call of the interpreter (Layers.py:33-36,
 * Module_Model.py:38-40,47-50), the batched stem convs (Module_Model.py:59-62) and the classifier
 * 1x1 conv (:72-75), and, with flipped weights, their input gradients.
 *   out = epilogue( sum_{chunk, tap} W[group][chunk][tap] * src[chunk](pixel + tap) )
 * epilogue: + bias[group], + res, ReLU, or multiply by (mask > 0), by flags / per-tile slots.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t src0, src1;   /* source slots */
    int32_t dst0, dst1;   /* destination slots */
    int32_t res;          /* residual slot or -1 */
    int32_t mask0, mask1; /* mask slots or -1 */
    int32_t wgroup;       /* weight group (token id) */
} vqa_conv_tile;

#define VQA_CONV_BIAS 1
#define VQA_CONV_RELU 2

typedef struct {
    const void* src;        /* bf16 base of source slots */
    void* dst;              /* bf16 base of destination slots */
    const void* res;        /* bf16 base of residual slots (geometry of dst) */
    const void* mask;       /* bf16 base of mask slots (geometry of dst) */
    const void* w;          /* packed bf16 weights, see vqa_pack_conv_weight */
    const float* bias;      /* fp32 [groups][n_out] or null */
    const vqa_conv_tile* tiles; /* device */
    int64_t src_slot_stride;/* elements between source slots */
    int64_t dst_slot_stride;
    int64_t w_group_stride; /* elements between weight groups */
    int32_t src_pix_stride; /* elements between pixels of a source map */
    int32_t dst_pix_stride;
    int32_t k_chunks;       /* 128-channel source chunks; K = k_chunks * taps * 128 */
    int32_t n_out;          /* output channels, multiple of 128 */
    int32_t pair_chunks;    /* 0: chunk c is channels [128c,128c+128) of src0.  >0: chunk c is
                               channels [128(c%pair),..) of (c/pair ? src1 : src0) */
    int32_t pair_out;       /* same rule for the 128-wide output tiles and dst0/dst1, mask0/mask1 */
    int32_t flags;
    int32_t num_tiles;
} vqa_conv_params;

int vqa_conv_tiles(const vqa_conv_params* p, int32_t taps /* 9 or 1 */, void* stream);

/* The stem's first convolution straight from the reference's input layout (Module_Model.py:59-60,239 on
 * the fp32 NCHW batch of Train_model.py:311): x fp32 [N][Cin][14][14], 3x3, Cin -> 128 channels, bias,
 * [ReLU]; sample n goes to slot dst_slot_first + n of dst (bf16 [slot][196][dst_pix_stride]).  The layout
 * change rides along inside the kernel (no vqa_nchw_to_nhwc pass); when x_nhwc is non-null the bf16
 * channels-last copy [N][196][Cin] that the weight-gradient pass reads is written as well.  w: the
 * convolution's weights packed by vqa_pack_conv_weight (one group, 128 outputs).  Cin % 128 == 0.
 * Same bits as vqa_nchw_to_nhwc followed by vqa_conv_tiles. */
int vqa_stem_nchw(const float* x, void* x_nhwc, void* dst, const void* w, const float* bias,
                  int64_t dst_slot_stride, int32_t dst_slot_first, int32_t dst_pix_stride, int32_t N,
                  int32_t Cin, int32_t flags /* bit 0: ReLU; bits 1-4 switch parts of the kernel off
                  for timing (scripts/bench_stem.py) and give wrong results: leave them zero */,
                  void* stream);

/* Fused module body for module width 128: every convolution of one tree node in one workgroup
 * (Layers.py:33-36 / Module_Model.py:38-40,47-50 forward; their input gradients backward).  All
 * maps are slots of one arena, 256-byte pixels.
 *   [pre, if src1 >= 0:  slot res = [relu](pre([src0; src1]) + bias_pre), and the input is that map;
 *    pre, if pre_group >= 0:  slot res = pre_half(src0) * (pre_mask > 0), likewise the input, where
 *      pre_half is outputs [128 h, 128 h + 128) of 1x1 weight group g, pre_group = 2 g + h, K = 128;
 *    otherwise the input is src0]
 *   slot dst1 = [relu](first(input) + bias_first) * (mask1 > 0)
 *   slot dst0 = [relu](second(dst1 map) + bias_second + res) * (mask0 > 0)
 * Forward of a ConcatBlock uses the first form (the 1x1 projection of the concatenated children,
 * Module_Model.py:38-39).  Backward, a ConcatBlock's tile ends with dP in dst0 and each of its two
 * children starts with the second form: its half of the transposed projection of the parent's dP,
 * masked by its own output, which is its dZ. */
typedef struct {
    int32_t src0, src1;
    int32_t dst0, dst1;
    int32_t res;                /* -1: none */
    int32_t mask0, mask1;       /* -1: none */
    int32_t wgroup;             /* weight group (token id) */
    int32_t pre_group;          /* -1: none; else 2 * (weight group of w_pre) + output half */
    int32_t pre_mask;           /* -1: none */
    int32_t reserved0, reserved1;
} vqa_resblock_tile;

typedef struct {
    void* arena;
    const void* w_first;        /* packed 3x3 weights of the first convolution, one group per token */
    const void* w_second;
    const void* w_pre;          /* packed 1x1 weights: K = 256 (two sources), 128 outputs for tiles with
                                   src1 >= 0; K = 128, 256 outputs for tiles with pre_group >= 0; or null */
    const void* w_unused;       /* keeps the layout of earlier versions; ignored */
    const float* bias_first;    /* fp32 [groups][128] or null */
    const float* bias_second;
    const float* bias_pre;
    const vqa_resblock_tile* tiles; /* device */
    int64_t slot_stride;        /* elements between slots (196 * 128) */
    int64_t w_group_stride;     /* elements between 3x3 weight groups (9 * 128 * 128) */
    int64_t w_pre_group_stride; /* elements between 1x1 weight groups (2 * 128 * 128) */
    int32_t flags;              /* VQA_CONV_BIAS | VQA_CONV_RELU, applied to every convolution but the
                                   pre_group form of pre */
    int32_t num_tiles;
    int32_t num_pre;            /* how many K = 256 1x1 passes the tiles hold in front (two pre_group
                                   tiles count as one; work accounting and argument checks: the
                                   kernel decides per tile) */
    int32_t unused0;
    unsigned long long* phase_stamps; /* diagnostics: null, or device [num_tiles][8] shader-clock
                                   timestamps of each tile's phase boundaries */
    const int32_t* level_info;  /* null, or device {first tile, tiles}: the tile range is read on the
                                   device (`tiles` + first tile, that many) and shared by the
                                   num_tiles workgroups of the launch, whatever its length; for tile
                                   lists made by vqa_devsched_build, whose sizes the host never
                                   learns.  num_pre then only says whether such tiles may
                                   occur */
} vqa_resblock_params;

int vqa_resblock_tiles(const vqa_resblock_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * Program scheduler on the device, forward only: the parse and (level, token) bucketing of
 * vqa_schedule_build for programs that are in device memory (the sampler builds them there,
 * LSTM_Model.py:205-235), without the copy to the host that forward_modules_ints_helper
 * (Module_Model.py:196-222) and its caller (Train_model.py:339-354) start with.  Same parse rules,
 * hence the same trees and the same logits bit for bit as the host scheduler's.  Slots have fixed strides (NVmax = N * T): stem maps and scene module as in
 * vqa_schedule_build (F i, S0 N+i, scene output 2N+i, scene intermediate 3N+i), other nodes
 * Y = 4N + v, H = 4N + NVmax + v, projections P = 4N + 2 NVmax + b; the arena needs 4N + 3 NVmax
 * slots.  Level L (1..T) runs as vqa_resblock_tiles with level_info + 2L over `tiles` (a node of
 * level L has L nodes of its own below and including it, so a row holds at most T / L of them: no
 * more than N * (T / L) workgroups are of use).  A token id
 * outside [0, V) sets bit 0 of *error (the host scheduler's VQA_EVOCAB) and the row runs as a lone scene.
 * Limits: V <= 255, T <= 1023, N * T < 2^26, and programs short enough for a row's parse state
 * (52 T + 16 bytes) to fit LDS; VQA_EINVAL beyond them.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const int64_t* programs;  /* device [N][T] token ids */
    const int8_t* arity;      /* device [V]; 1 or 2 for every token but scene (checked per row read, see error) */
    float* used;              /* device out [N][T]: Module_Model.py:227,231 used_fns */
    int32_t* node;            /* device workspace [N][2T+1][4], 16-byte aligned */
    int32_t* counts;          /* device workspace [N][4], 16-byte aligned */
    int32_t* hist;            /* device out [T+1][V]: nodes per (level, token) */
    int32_t* row_base;        /* device workspace [N][2] */
    int32_t* level_info;      /* device out [T+1][2]: first tile, tiles of each level */
    int32_t* bin_off;         /* device workspace [T+1][V] */
    int32_t* bin_cursor;      /* device workspace [T+1][V] */
    int32_t* tiles;           /* device out [N*T][12]: vqa_resblock_tile records, level-major; 16-byte aligned */
    int32_t* cls_tiles;       /* device out [N][8]: vqa_conv_tile records of the classifier's gather; 16-byte aligned */
    int32_t* error;           /* device out [2]: flags (1: a token id outside [0, V); 2: an image index
                                 outside [0, num_images); 4: a token other than scene whose arity is not 1
                                 or 2 - such a row runs as a lone scene too); the deepest root's level */
    const int32_t* image_of_row; /* null: row i runs on image i.  Else device [N]: the image of each row
                                 (candidate scoring, Train_model.py:339-347: K programs on one image):
                                 stem and scene module run once per image, the rows of an image all read
                                 its scene output (slots F i, S0 M + i, scene 2M + i / 3M + i, then
                                 Y = 4M + v ... with M = num_images in place of N) */
    int32_t N, T, V, null_id, start_id, scene_id;
    int32_t spread;           /* 0: every node in its earliest level (1 + its deepest input's), the host
                                 scheduler's levels without VQA_PLAN_BALANCE.  1: rows with slack run
                                 some levels later, a different share of the slack per row, which evens
                                 the launches out */
    int32_t num_images;       /* M, with image_of_row; otherwise ignored (M = N) */
} vqa_devsched_params;

int vqa_devsched_build(const vqa_devsched_params* p, void* stream);

/* fp32 variants (v_mfma_f32_16x16x4_f32, fp32 maps): same descriptors and epilogue semantics.
 * Weights are not packed: `w` is fp32 [group][tap][k][n] (n contiguous, k over all source chunks in
 * order), w_group_stride = taps * K * n_out.  The input gradient uses the flipped, transposed array.
 * vqa_wgrad_tiles_f32 always accumulates atomically into dw [group][tap][k][n]. */
int vqa_conv_tiles_f32(const vqa_conv_params* p, int32_t taps, void* stream);

/* Weight packing: fp32 tensor -> bf16 image the conv kernel DMA-copies into LDS.
 * Source element (n, k, t) is at w[group*g_stride + n*n_stride + k*k_stride + (flip ? taps-1-t : t)]
 * (forward: n = out channel, k = in channel; input-gradient: swapped and flipped). */
int vqa_pack_conv_weight(const float* w, void* packed, int32_t groups, int32_t n_out, int32_t k_in,
                         int32_t taps, int64_t g_stride, int64_t n_stride, int64_t k_stride,
                         int32_t flip, void* stream);

/* ------------------------------------------------------------------------------------------
 * Grouped weight gradient: dW[group][tap][i][o] += sum over the group's nodes and pixels of
 * src(pixel + tap)[i] * dy(pixel)[o]; db[group][o] += sum dy(pixel)[o].  Replaces autograd's
 * per-call conv weight gradients and their accumulation into the shared per-token weights
 * (Train_model.py:313 over Module_Model.py:221).  Two ways out for dW: fp32 atomics into `dw`
 * (zero it first), or, when `partial` is set, one block of taps * 8192 floats per workgroup
 * (num_items * n_out/64 * k_chunks blocks), which vqa_wgrad_reduce sums per group into torch's
 * layout; the second is several times cheaper and deterministic.  db likewise: fp32 atomics, or,
 * in partial-block mode with `bias_block` set, one row of n_out sums per item that
 * vqa_wgrad_reduce adds per group in item order (the same bits on every run).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t group;   /* weight group (token id) */
    int32_t first;   /* first entry of src_slots / dy_slots */
    int32_t count;   /* nodes in this item */
    int32_t pad;
} vqa_wgrad_item;

typedef struct {
    const void* src;            /* bf16 base of source slots */
    const void* dy;             /* bf16 base of gradient slots */
    const vqa_wgrad_item* items;/* device */
    const int32_t* src_slots;   /* device: per node, source slot (pair: src1_slots too) */
    const int32_t* src1_slots;  /* device or null */
    const int32_t* dy_slots;    /* device */
    float* dw;                  /* fp32 [groups][taps][k_in][n_out] accumulators (atomic mode) */
    float* db;                  /* fp32 [groups][n_out] or null */
    float* partial;             /* non-null: partial-block mode, dw is not used */
    int64_t src_slot_stride, dy_slot_stride;
    int64_t dw_group_stride;    /* elements */
    int32_t src_pix_stride, dy_pix_stride;
    int32_t k_chunks;           /* 128-channel chunks of the source (i) dimension */
    int32_t pair_chunks;        /* as in vqa_conv_params */
    int32_t n_out;              /* channels of dy (o), multiple of 64 */
    int32_t num_items;
    int32_t total_nodes;        /* sum of the items' counts (work accounting only) */
    int32_t bias_block;         /* 0: the bias sums are added to db with atomics.  > 0 (partial-block
                                   mode): they are stored, [item][n_out], at partial + bias_block *
                                   8192 floats - behind the num_items * n_out/64 * k_chunks * taps
                                   blocks of the launch - and db is not touched; vqa_wgrad_reduce
                                   sums them */
} vqa_wgrad_params;

int vqa_wgrad_tiles(const vqa_wgrad_params* p, int32_t taps, void* stream);
int vqa_wgrad_tiles_f32(const vqa_wgrad_params* p, int32_t taps, void* stream);

/* out[group][n][k][tap] (torch's OIHW, groups apart by out_g_stride) = sum of the partial blocks
 * of the group's items; a group without items gets zeros.  items / num_items / n_out / k_chunks as
 * in the vqa_wgrad_tiles call that wrote `partial`.  bias_partial / db: null, or the [item][n_out]
 * sums that call stored (its partial + bias_block * 8192) and the fp32 [groups][n_out] array that
 * receives their per-group sums (written whole; n_out <= 2048). */
int vqa_wgrad_reduce(const float* partial, const vqa_wgrad_item* items, int32_t num_items,
                     int32_t groups, int32_t n_out, int32_t k_chunks, int32_t taps, float* out,
                     int64_t out_g_stride, const float* bias_partial, float* db, void* stream);

/* vqa_wgrad_tiles in partial-block mode followed by vqa_wgrad_reduce, for callers that cannot
 * allocate device memory or bring a workspace: code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  No counterpart in the reference.  The partial blocks live in a
 * 32 MiB device array that is part of the library (each GPU the library runs on holds its own copy
 * until the process ends); the call allocates nothing and frees nothing.
 *   out[group][n][k][tap] = sum over the group's items, as vqa_wgrad_reduce writes it (`groups`
 *   groups, out_g_stride apart; all of them are written); db is accumulated into as by
 *   vqa_wgrad_tiles; p->partial and p->dw are ignored.
 * Limit: num_items * (n_out / 64) * k_chunks * taps <= 1024 partial blocks of 32 KiB, beyond which
 * the call fails with VQA_EINVAL and a message that says so.  All calls on one GPU share that GPU's
 * array (it keeps the last call's contents; nothing in it needs clearing between calls), so two
 * calls that overlap on one GPU (different streams) race: serialise them.  The first call on a
 * device looks the array's address up (hipGetSymbolAddress): make it outside graph capture. */
int vqa_wgrad_tiles_noalloc(const vqa_wgrad_params* p, int32_t taps, int32_t groups, float* out,
                            int64_t out_g_stride, void* stream);

/* dW accumulators [groups][taps][k][n] fp32 -> torch layout [n][k][taps] (OIHW), groups apart by
 * out_g_stride */
int vqa_unpack_conv_grad(const float* dw, float* out, int32_t groups, int32_t n_out, int32_t k_in,
                         int32_t taps, int64_t out_g_stride, void* stream);

/* ------------------------------------------------------------------------------------------
 * Dense GEMM for the classifier's linear layers (Module_Model.py:85-92) and their gradients.
 * C[M][N] (+)= A * B with bf16 operands; a_kmajor = 0: A is [M][K] (lda = row stride), 1: A is
 * stored [K][M]; b_kmajor = 0: B is stored [N][K] (weights [out][in]), 1: B is stored [K][N].
 * Operand buffers must be allocated (and zero filled) out to multiples of 128 rows/cols and K must
 * be a multiple of 64.  out_mode 1: fp32 store; with split_k > 1 slab z (the partial sum over its
 * share of K) goes to c + z*M*ldc and the caller adds the slabs (vqa_bias_act does);
 * out_mode 2: bf16 store, times (mask > 0) when mask is non-null (mask has C's geometry, bf16).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const void* a; const void* b; void* c; const void* mask;
    int64_t lda, ldb, ldc;
    int32_t M, N, K;
    int32_t a_kmajor, b_kmajor, out_mode, split_k;
} vqa_gemm_params;

int vqa_gemm(const vqa_gemm_params* p, void* stream);

/* The same product at fp32 (operands, accumulation on v_mfma_f32_16x16x4_f32, result) for the linear
 * layers of the tolerance configuration (Module_Model.py:85-92 and their gradients): out_mode must
 * be 1; no padding or alignment requirements; `mask`, when set, is fp32 with C's geometry (result
 * times (mask > 0), unsplit only); split_k > 1 writes slabs as vqa_gemm does. */
int vqa_gemm_f32(const vqa_gemm_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * Layout and elementwise helpers
 * ------------------------------------------------------------------------------------------ */
/* x fp32 [N][Cin][196] (NCHW, the loader's layout, Train_model.py:311) -> bf16 [N][196][Cpad] */
int vqa_nchw_to_nhwc(const float* x, void* out, int32_t N, int32_t Cin, int32_t Cpad, void* stream);
/* bf16 [N][196][C] -> fp32 [N][C][196] */
int vqa_nhwc_to_nchw(const void* x, float* out, int32_t N, int32_t C, void* stream);
/* fp32 [rows][cols] -> bf16 [rows_pad][cols_pad] (zero padded) */
int vqa_cast_pad(const float* w, void* out, int32_t rows, int32_t cols, int32_t rows_pad,
                 int32_t cols_pad, void* stream);
/* FC weight over the flattened [P][196] map (Layers.py:42-44 flatten order c*196+hw) <-> the
 * channels-last order hw*P+c.  to_nhwc: fp32 [F][P*196] -> bf16 [Fpad][196*P];
 * from_nhwc: fp32 [F][196*P] -> fp32 [F][P*196] */
int vqa_fc_weight_to_nhwc(const float* w, void* out, int32_t F, int32_t Fpad, int32_t P, void* stream);
int vqa_fc_grad_from_nhwc(const float* g, float* out, int32_t F, int32_t P, void* stream);
/* Per-sample transposition bf16 [N][196][C] -> bf16 [N][C][196]: a channels-last activation in the
 * flatten order of Layers.py:42-44, so that the weight gradient of the linear layer behind it
 * (Module_Model.py:86-88) comes out of its GEMM in the reference's own layout.  C % 64 == 0. */
int vqa_transpose_hw_c(const void* x, void* out, int32_t N, int32_t C, void* stream);
/* Batched fp32 transposition out[i][c][r] = x[i][r][c] for i < n, r < rows, c < cols; x is dense
 * [n][rows][cols], rows of out are ld_out floats apart and its matrices out_stride apart (the rest
 * of out is left alone).  The fp32 path's layout changes: the loader's NCHW batch (Train_model.py:311)
 * -> channels-last maps, and channels-last <-> the flatten order of Layers.py:42-44. */
int vqa_transpose_f32(const float* x, float* out, int32_t n, int32_t rows, int32_t cols, int64_t ld_out,
                      int64_t out_stride, void* stream);
/* out[rows][ld_out] = act(sum over `slabs` split-K slabs of acc fp32 [rows][ld_acc] + bias[col]) for
 * col < cols; out is bf16, or fp32 when out_f32 */
int vqa_bias_act(const float* acc, const float* bias, void* out, int32_t rows, int32_t cols,
                 int64_t ld_acc, int64_t ld_out, int32_t relu, int32_t slabs, int64_t slab_stride,
                 int32_t out_f32, void* stream);
/* out[col] += sum over rows of x[row][col]; x bf16 (is_bf16) or fp32 */
int vqa_col_sum(const void* x, float* out, int32_t rows, int32_t cols, int64_t ld, int32_t is_bf16,
                void* stream);
/* ------------------------------------------------------------------------------------------
 * The classifier's last linear layer and the loss behind it, in few launches (their cost is
 * launches, not arithmetic).  Replaces nn.Linear(fc, num_answers) of Module_Model.py:92 and
 * autograd's gradients of it, and torch.nn.CrossEntropyLoss of Train_model.py:185,312.
 * Operands are bf16 where the GEMM path has bf16 (act [N][ld_act], w [A][ld_w], the gradient of
 * the logits is rounded on the way in); sums are fp32 in a fixed order.  A <= 64, F % 64 == 0.
 * ------------------------------------------------------------------------------------------ */
/* logits[n][o] = sum_i act[n][i] * w[o][i] + bias[o]   (fp32 [N][A]; bias may be null) */
int vqa_head_forward(const void* act, int64_t ld_act, const void* w, int64_t ld_w, const float* bias,
                     float* logits, int32_t N, int32_t F, int32_t A, void* stream);
/* floats of workspace vqa_head_backward needs */
int64_t vqa_head_backward_workspace(int32_t N, int32_t F, int32_t A);
/* From dlogits fp32 [N][A]: gin bf16 [N][ld_gin] = (dlogits x w) * (act > 0), the gradient into the
 * layer's input times the ReLU that produced it; dw fp32 [A][F], db fp32 [A] (this layer's weight and
 * bias gradients) and db_in fp32 [F] = column sums of gin (the bias gradient of the layer in front).
 * All four are written whole. */
int vqa_head_backward(const float* dlogits, const void* act, int64_t ld_act, const void* w, int64_t ld_w,
                      void* gin, int64_t ld_gin, float* dw, float* db_in, float* db, float* workspace,
                      int32_t N, int32_t F, int32_t A, void* stream);
/* loss[0] = mean over n of (logsumexp(logits[n]) - logits[n][target[n]]) and its gradient
 * dlogits = (softmax(logits) - onehot(target)) / N, fp32 [N][A], in two launches (rows, then the sum of
 * per-workgroup partial losses in a fixed order); targets int64 in [0, A) (a row whose target is out of
 * range adds nothing to either); workspace: 128 floats */
int vqa_xent(const float* logits, const int64_t* target, float* loss, float* dlogits, float* workspace, int32_t N,
             int32_t A, void* stream);

/* dst slot <- (sum of listed source slots) * (mask slot > 0): scene-node gradient fan-in
 * (feats[i:i+1] is read by every scene node of sample i, Module_Model.py:214) */
int vqa_slot_sum_mask(void* arena, const int32_t* offsets, const int32_t* src_slots,
                      const int32_t* dst_slots, const int32_t* mask_slots, int32_t n_dst,
                      int64_t slot_elems, void* stream);
int vqa_slot_sum_mask_f32(float* arena, const int32_t* offsets, const int32_t* src_slots,
                          const int32_t* dst_slots, const int32_t* mask_slots, int32_t n_dst,
                          int64_t slot_elems, void* stream);

/* ------------------------------------------------------------------------------------------
 * Adam (Train_model.py:165-167,315: torch.optim.Adam, L2 weight decay) for the executor's
 * parameters.  g = grad + wd p;  m += (1-b1)(g-m);  v = b2 v + (1-b2) g^2;
 * p -= step_size * m / (sqrt(v) * inv_bc2_sqrt + eps), with step_size = lr / (1 - b1^t) and
 * inv_bc2_sqrt = 1 / sqrt(1 - b2^t) per tensor (a token unused in a batch has no gradient, is left
 * out of the table and keeps its own step count, as torch skips `grad is None`).
 * vqa_adam_tensors: one launch over `num_chunks` chunks of 4096 elements (device tables).
 * vqa_adam_fc1: the first linear layer's weight [F][P][196] (Module_Model.py:85-88), which also
 * writes the bf16 [F][196][P] copy the executor's GEMM reads.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    float* p; const float* g; float* m; float* v;
    int64_t numel;
    float step_size, inv_bc2_sqrt;
} vqa_adam_tensor;
typedef struct { int32_t tensor; int32_t pad; int64_t offset; } vqa_adam_chunk;

int vqa_adam_tensors(const vqa_adam_tensor* tensors, const vqa_adam_chunk* chunks, int32_t num_chunks,
                     float weight_decay, float beta1, float beta2, float eps, void* stream);
int vqa_adam_fc1(float* p, const float* g, float* m, float* v, void* packed, int32_t F, int32_t P,
                 float weight_decay, float beta1, float beta2, float eps, float step_size,
                 float inv_bc2_sqrt, void* stream);

/* The same update for convolution weights [cout][cin][taps] (torch's OIHW, taps flattened; cout and
 * cin multiples of 128) that also rewrites, per tensor, the packed bf16 copies the convolution
 * kernels read (vqa_pack_conv_weight's layout): pk_fwd = the forward copy of this tensor's weight
 * group (n = output channel, k = input channel), pk_bwd = its input-gradient copy (n = input
 * channel, k = output channel, taps flipped); either may be null.  One launch updates `num_tensors`
 * tensors of one shape from a device table; saves re-packing every weight after every optimizer step. */
typedef struct {
    float* p; const float* g; float* m; float* v;
    void* pk_fwd; void* pk_bwd;
    float step_size, inv_bc2_sqrt;
} vqa_adam_conv_tensor;
int vqa_adam_conv_pack(const vqa_adam_conv_tensor* tensors, int32_t num_tensors, int32_t cout, int32_t cin,
                       int32_t taps, float weight_decay, float beta1, float beta2, float eps, void* stream);

/* ------------------------------------------------------------------------------------------
 * Feature feed, host side.  The reference's loader reads one saved [1024][14][14] tensor per image
 * and stacks a batch in its collate step (DataLoader.py:118-119, DataLoader.py:141-166).  Here the
 * features of a split sit in one table of rows (bf16 [196][C] each) in host memory, and a batch is
 * gathered into a contiguous (pinned) buffer by `threads` host threads:
 * out[i] = table[index[i]], rows of row_bytes bytes.  Host pointers only; returns when the copy is
 * complete.  An index outside [0, table_rows) fails with VQA_EINVAL before anything is written
 * (the reference raises on a missing file).
 * ------------------------------------------------------------------------------------------ */
int vqa_gather_rows(const void* table, int64_t table_rows, int64_t row_bytes, const int64_t* index,
                    int64_t n, void* out, int32_t threads);

/* ------------------------------------------------------------------------------------------
 * Per-kernel timing for the benchmark (no counterpart in the reference).  Every launch of an MFMA
 * kernel whose kind is enabled is bracketed by HIP events on its stream (two event records per
 * launch cost host time: enable only the kinds wanted).  vqa_prof_collect waits for
 * them and writes, per kind k (0 conv3x3, 1 conv1x1, 2 wgrad3x3, 3 wgrad1x1, 4 gemm,
 * 5 fused residual body; `kinds` >= 6):
 * out[4k] launches, out[4k+1] total ms, out[4k+2] algorithmic FLOPs, out[4k+3] algorithmic bytes.
 * ------------------------------------------------------------------------------------------ */
int vqa_prof_enable(int32_t kind_mask);   /* bit k set: time launches of kind k; 0: off */
int vqa_prof_collect(double* out, int32_t kinds);

/* ------------------------------------------------------------------------------------------
 * Whole module tree.  Replaces forward_modules_ints (Module_Model.py:224-233) and autograd's
 * backward over it.  `blob` is the device copy of vqa_plan_blob.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    void* arena;                       /* bf16 [slots][196][C] */
    const int32_t* blob;               /* device */
    const void *w_conv1, *w_conv2, *w_proj;     /* packed forward weights, one group per token */
    const void *wd_conv1, *wd_conv2, *wd_proj;  /* packed input-gradient weights */
    const float *b_conv1, *b_conv2, *b_proj;    /* fp32 [V][C] */
    float *g_conv1, *g_conv2, *g_proj;          /* out: fp32 weight gradients in torch's layout,
                                                   [V][C][C][3][3] and [V][C][2C][1][1] (all written) */
    float *gb_conv1, *gb_conv2, *gb_proj;       /* out: fp32 bias gradients [V][C]; required (gb_proj when the plan
                                                   has binary nodes).  bf16 path: written whole, rows of unused
                                                   tokens as zeros.  VQA_TREE_F32: accumulated into - zero them first */
    float* wgrad_partial;                       /* workspace of vqa_tree_backward:
                                                   vqa_plan_info.wgrad_partial_blocks * 8192 floats */
    int32_t C, V;
    int32_t flags;                     /* VQA_TREE_* */
    int32_t reserved;
} vqa_tree_ctx;

/* vqa_tree_ctx.flags: run the residual body as two vqa_conv_tiles launches even where the fused
 * vqa_resblock_tiles kernel applies (C == 128); same results, for comparison and timing. */
#define VQA_TREE_UNFUSED 1
/* fp32 maps and weights (the tolerance configuration): the arena holds fp32 [slots][196][C], the
 * weight pointers are plain fp32 [V][tap][k][n] arrays (see vqa_conv_tiles_f32), the separate tile
 * lists are used, and g_conv1 / g_conv2 / g_proj are fp32 accumulators [V][tap][k][n] that the
 * caller zeroes first and converts with vqa_unpack_conv_grad (wgrad_partial is not used). */
#define VQA_TREE_F32 2
/* vqa_tree_forward: level 0 (the scene modules, one tile per image, independent of the programs)
 * was already run by the caller (vqa_resblock_tiles on its own tile list, before the plan existed) */
#define VQA_TREE_SKIP_LEVEL0 4

int vqa_tree_forward(const vqa_plan* plan, const vqa_tree_ctx* ctx, void* stream);
/* input gradients level by level, scene fan-in, then all weight gradients */
int vqa_tree_backward(const vqa_plan* plan, const vqa_tree_ctx* ctx, void* stream);

#ifdef __cplusplus
}
#endif
#endif
