// Level-by-level execution of the whole batch's module trees: the replacement for the per-sample
// recursion of Module_Model.py:224-233 and for autograd's walk back over it.  Host code; it only
// fills parameter blocks and enqueues the grouped kernels of conv.hip / wgrad.hip / misc.hip.
#include <hip/hip_runtime.h>

#include "plan.h"
#include "prof.h"

namespace {

vqa_conv_params base_conv(const vqa_plan* pl, const vqa_tree_ctx* c) {
    vqa_conv_params p = {};
    p.src = p.res = p.mask = c->arena;
    p.dst = c->arena;
    const int64_t slot = (int64_t)VQA_HW * pl->C;
    p.src_slot_stride = p.dst_slot_stride = slot;
    p.src_pix_stride = p.dst_pix_stride = pl->C;
    p.k_chunks = pl->C / 128;
    p.n_out = pl->C;
    return p;
}

int conv(const vqa_tree_ctx* c, const vqa_conv_params* p, int taps, void* stream) {
    return (c->flags & VQA_TREE_F32) ? vqa_conv_tiles_f32(p, taps, stream) : vqa_conv_tiles(p, taps, stream);
}

const vqa_conv_tile* tiles_at(const vqa_tree_ctx* c, int word_off) {
    return (const vqa_conv_tile*)(c->blob + word_off);
}

}  // namespace

extern "C" int vqa_tree_forward(const vqa_plan* pl, const vqa_tree_ctx* c, void* stream) {
    if (!pl || !c || !c->arena || !c->blob || !c->w_conv1 || !c->w_conv2 || !c->b_conv1 || !c->b_conv2 ||
        c->C != pl->C || (pl->num_binary > 0 && (!c->w_proj || !c->b_proj))) {
        vqa_set_error("vqa_tree_forward: bad argument");
        return VQA_EINVAL;
    }
    const int cc = pl->C / 128;
    const int64_t w33 = (int64_t)9 * pl->C * pl->C, w11 = (int64_t)2 * pl->C * pl->C;
    const bool fused = pl->C == 128 && !(c->flags & (VQA_TREE_UNFUSED | VQA_TREE_F32));
    if (fused ? !pl->has_fused : !pl->has_separate) {
        vqa_set_error("vqa_tree_forward: the plan was built without the tile lists of this path");
        return VQA_EINVAL;
    }
    // (timing, when it is on: the levels' launches follow each other with nothing between them - one pair of
    // events around the run, not one per launch)
    VqaProfRun prof_run(VQA_K_RESBLOCK, (hipStream_t)stream);
    for (int L = (c->flags & VQA_TREE_SKIP_LEVEL0) ? 1 : 0; L < pl->num_levels; L++) {
        const PlanLevel& lv = pl->levels[L];
        int rc;
        if (fused) {   // the whole node in one kernel
            vqa_resblock_params p = {};
            p.arena = c->arena;
            p.w_first = c->w_conv1; p.w_second = c->w_conv2; p.w_pre = c->w_proj;
            p.bias_first = c->b_conv1; p.bias_second = c->b_conv2; p.bias_pre = c->b_proj;
            p.tiles = (const vqa_resblock_tile*)(c->blob + lv.off_fused_fwd); p.num_tiles = lv.count;
            p.num_pre = lv.bcount;
            p.slot_stride = (int64_t)VQA_HW * 128; p.w_group_stride = w33; p.w_pre_group_stride = w11;
            p.flags = VQA_CONV_BIAS | VQA_CONV_RELU;
            if ((rc = vqa_resblock_tiles(&p, stream))) return rc;
            continue;
        }
        if (lv.bcount > 0) {   // P = relu(proj([x; y]) + b)   Module_Model.py:38-39
            vqa_conv_params p = base_conv(pl, c);
            p.w = c->w_proj; p.bias = c->b_proj; p.w_group_stride = w11;
            p.tiles = tiles_at(c, lv.off_proj); p.num_tiles = lv.bcount;
            p.k_chunks = 2 * cc; p.pair_chunks = cc;
            p.flags = VQA_CONV_BIAS | VQA_CONV_RELU;
            if ((rc = conv(c, &p, 1, stream))) return rc;
        }
        {                      // H = relu(conv1(in) + b1)     Layers.py:33 / Module_Model.py:47
            vqa_conv_params p = base_conv(pl, c);
            p.w = c->w_conv1; p.bias = c->b_conv1; p.w_group_stride = w33;
            p.tiles = tiles_at(c, lv.off_conv1); p.num_tiles = lv.count;
            p.flags = VQA_CONV_BIAS | VQA_CONV_RELU;
            if ((rc = conv(c, &p, 9, stream))) return rc;
        }
        {                      // Y = relu(in + conv2(H) + b2) Layers.py:33-36
            vqa_conv_params p = base_conv(pl, c);
            p.w = c->w_conv2; p.bias = c->b_conv2; p.w_group_stride = w33;
            p.tiles = tiles_at(c, lv.off_conv2); p.num_tiles = lv.count;
            p.flags = VQA_CONV_BIAS | VQA_CONV_RELU;
            if ((rc = conv(c, &p, 9, stream))) return rc;
        }
    }
    return VQA_OK;
}

extern "C" int vqa_tree_backward(const vqa_plan* pl, const vqa_tree_ctx* c, void* stream) {
    if (!pl || !c || !c->arena || !c->blob || !c->wd_conv1 || !c->wd_conv2 || !c->g_conv1 || !c->g_conv2 ||
        (!c->wgrad_partial && !(c->flags & VQA_TREE_F32)) || c->C != pl->C ||
        (pl->num_binary > 0 && (!c->wd_proj || !c->g_proj))) {
        vqa_set_error("vqa_tree_backward: bad argument");
        return VQA_EINVAL;
    }
    if (!c->gb_conv1 || !c->gb_conv2 || (pl->num_binary > 0 && !c->gb_proj)) {
        vqa_set_error("vqa_tree_backward: the bias-gradient arrays gb_conv1, gb_conv2 (and gb_proj for plans with binary "
                      "nodes) are required");
        return VQA_EINVAL;
    }
    if (pl->shared) {
        vqa_set_error("vqa_tree_backward: a plan built with shared images is forward-only");
        return VQA_EINVAL;
    }
    if (!pl->finished) {
        vqa_set_error("vqa_tree_backward: the plan's backward part is not built yet (vqa_plan_finish)");
        return VQA_EINVAL;
    }
    const int cc = pl->C / 128;
    const int64_t w33 = (int64_t)9 * pl->C * pl->C, w11 = (int64_t)2 * pl->C * pl->C;
    const int64_t slot = (int64_t)VQA_HW * pl->C;
    const bool fused = pl->C == 128 && !(c->flags & (VQA_TREE_UNFUSED | VQA_TREE_F32));
    if (fused ? !pl->has_fused : !pl->has_separate) {
        vqa_set_error("vqa_tree_backward: the plan was built without the tile lists of this path");
        return VQA_EINVAL;
    }
    int rc;
    const bool f32 = (c->flags & VQA_TREE_F32) != 0;
    VqaProfRun prof_run(VQA_K_RESBLOCK, (hipStream_t)stream);      // levels n - 1 .. 1: back to back
    for (int L = pl->num_levels - 1; L >= 0; L--) {
        const PlanLevel& lv = pl->levels[L];
        if (L == 0) {
            prof_run.end();
            const int32_t* bl = c->blob;
            if (fused && pl->sec_cnt[VQA_SEC_SCENE_DPROJ_TILES] > 0) {
                // binary nodes' dP -> the dZ of their inputs that are scene nodes (every other input
                // takes its half of the transposed projection inside its own tile)
                vqa_conv_params p = base_conv(pl, c);
                p.w = c->wd_proj; p.w_group_stride = w11;
                p.tiles = tiles_at(c, pl->sec_off[VQA_SEC_SCENE_DPROJ_TILES]);
                p.num_tiles = pl->sec_cnt[VQA_SEC_SCENE_DPROJ_TILES];
                p.n_out = 2 * pl->C; p.pair_out = cc;
                if ((rc = vqa_conv_tiles(&p, 1, stream))) return rc;
            }
            // every scene node of a sample is the same computation, run once: the gradients that its
            // users wrote for each of them are summed into the one that ran, times (its output > 0)
            rc = f32 ? vqa_slot_sum_mask_f32((float*)c->arena, bl + pl->sec_off[VQA_SEC_SCENE_OFFSETS],
                                             bl + pl->sec_off[VQA_SEC_SCENE_SRC], bl + pl->sec_off[VQA_SEC_SLOTS_SCDZ],
                                             bl + pl->sec_off[VQA_SEC_SLOTS_SCY], pl->N, slot, stream)
                     : vqa_slot_sum_mask(c->arena, bl + pl->sec_off[VQA_SEC_SCENE_OFFSETS],
                                         bl + pl->sec_off[VQA_SEC_SCENE_SRC], bl + pl->sec_off[VQA_SEC_SLOTS_SCDZ],
                                         bl + pl->sec_off[VQA_SEC_SLOTS_SCY], pl->N, slot, stream);
            if (rc) return rc;
        }
        if (fused) {
            vqa_resblock_params p = {};
            p.arena = c->arena;
            p.w_first = c->wd_conv2; p.w_second = c->wd_conv1; p.w_pre = c->wd_proj;
            p.tiles = (const vqa_resblock_tile*)(c->blob + lv.off_fused_bwd); p.num_tiles = lv.count;
            p.num_pre = lv.half_count / 2;
            p.slot_stride = slot; p.w_group_stride = w33; p.w_pre_group_stride = w11;
            if ((rc = vqa_resblock_tiles(&p, stream))) return rc;
            continue;
        }
        {   // dH = dgrad_conv2(dZ) * (H > 0)
            vqa_conv_params p = base_conv(pl, c);
            p.w = c->wd_conv2; p.w_group_stride = w33;
            p.tiles = tiles_at(c, lv.off_dgrad2); p.num_tiles = lv.count;
            if ((rc = conv(c, &p, 9, stream))) return rc;
        }
        {   // d(in) = (dgrad_conv1(dH) + dZ) * (in > 0)
            vqa_conv_params p = base_conv(pl, c);
            p.w = c->wd_conv1; p.w_group_stride = w33;
            p.tiles = tiles_at(c, lv.off_dgrad1); p.num_tiles = lv.count;
            if ((rc = conv(c, &p, 9, stream))) return rc;
        }
        if (lv.bcount > 0) {   // [dx; dy] = dP * Wproj, each masked by its own child's output
            vqa_conv_params p = base_conv(pl, c);
            p.w = c->wd_proj; p.w_group_stride = w11;
            p.tiles = tiles_at(c, lv.off_dproj); p.num_tiles = lv.bcount;
            p.n_out = 2 * pl->C; p.pair_out = cc;
            if ((rc = conv(c, &p, 1, stream))) return rc;
        }
    }
    // weight gradients of every module, grouped by token
    vqa_wgrad_params w = {};
    w.src = w.dy = c->arena;
    w.src_slot_stride = w.dy_slot_stride = slot;
    w.src_pix_stride = w.dy_pix_stride = pl->C;
    w.n_out = pl->C;
    w.items = (const vqa_wgrad_item*)(c->blob + pl->sec_off[VQA_SEC_WG_ITEMS]);
    w.num_items = pl->sec_cnt[VQA_SEC_WG_ITEMS];
    w.total_nodes = pl->num_exec;
    w.k_chunks = cc;
    if (f32) {     // atomic accumulation into [V][tap][k][n] arrays the caller zeroed
        w.dw_group_stride = w33;
        w.src_slots = c->blob + pl->sec_off[VQA_SEC_WG_SRC_C1];
        w.dy_slots = c->blob + pl->sec_off[VQA_SEC_WG_DY_C1];
        w.dw = c->g_conv1; w.db = c->gb_conv1;
        if ((rc = vqa_wgrad_tiles_f32(&w, 9, stream))) return rc;
        w.src_slots = c->blob + pl->sec_off[VQA_SEC_WG_SRC_C2];
        w.dy_slots = c->blob + pl->sec_off[VQA_SEC_WG_DY_C2];
        w.dw = c->g_conv2; w.db = c->gb_conv2;
        if ((rc = vqa_wgrad_tiles_f32(&w, 9, stream))) return rc;
        if (pl->sec_cnt[VQA_SEC_WG_PITEMS] > 0) {
            w.items = (const vqa_wgrad_item*)(c->blob + pl->sec_off[VQA_SEC_WG_PITEMS]);
            w.num_items = pl->sec_cnt[VQA_SEC_WG_PITEMS];
            w.total_nodes = pl->num_binary;
            w.src_slots = c->blob + pl->sec_off[VQA_SEC_WG_PSRC0];
            w.src1_slots = c->blob + pl->sec_off[VQA_SEC_WG_PSRC1];
            w.dy_slots = c->blob + pl->sec_off[VQA_SEC_WG_PDY];
            w.k_chunks = 2 * cc; w.pair_chunks = cc;
            w.dw_group_stride = w11;
            w.dw = c->g_proj; w.db = c->gb_proj;
            if ((rc = vqa_wgrad_tiles_f32(&w, 1, stream))) return rc;
        }
        return VQA_OK;
    }
    // partial blocks, and behind them one row of bias sums per item: both summed per token by
    // vqa_wgrad_reduce in item order, so every gradient comes out the same on every run
    w.partial = c->wgrad_partial;
    const float* bias_rows;
    if (c->g_conv2 == c->g_conv1 + (int64_t)c->V * w33 && c->gb_conv2 == c->gb_conv1 + (int64_t)c->V * pl->C &&
        pl->sec_cnt[VQA_SEC_WG_ITEMS_12] == 2 * w.num_items) {
        // the caller's two gradient arrays are one [2 V][...] array: both convolutions' weight gradients as one launch
        // over the merged item list (groups V.. are the second convolution's), and one reduction
        w.items = (const vqa_wgrad_item*)(c->blob + pl->sec_off[VQA_SEC_WG_ITEMS_12]);
        w.num_items = pl->sec_cnt[VQA_SEC_WG_ITEMS_12];
        w.total_nodes = 2 * pl->num_exec;
        w.bias_block = w.num_items * (pl->C / 64) * cc * 9;
        bias_rows = w.partial + (int64_t)w.bias_block * 8192;
        w.src_slots = c->blob + pl->sec_off[VQA_SEC_WG_SRC_12];
        w.dy_slots = c->blob + pl->sec_off[VQA_SEC_WG_DY_12];
        if ((rc = vqa_wgrad_tiles(&w, 9, stream))) return rc;
        if ((rc = vqa_wgrad_reduce(w.partial, w.items, w.num_items, 2 * c->V, pl->C, cc, 9, c->g_conv1, w33, bias_rows, c->gb_conv1, stream))) return rc;
    } else {
        w.bias_block = w.num_items * (pl->C / 64) * cc * 9;
        bias_rows = w.partial + (int64_t)w.bias_block * 8192;
        w.src_slots = c->blob + pl->sec_off[VQA_SEC_WG_SRC_C1];
        w.dy_slots = c->blob + pl->sec_off[VQA_SEC_WG_DY_C1];
        if ((rc = vqa_wgrad_tiles(&w, 9, stream))) return rc;
        if ((rc = vqa_wgrad_reduce(w.partial, w.items, w.num_items, c->V, pl->C, cc, 9, c->g_conv1, w33, bias_rows, c->gb_conv1, stream))) return rc;
        w.src_slots = c->blob + pl->sec_off[VQA_SEC_WG_SRC_C2];
        w.dy_slots = c->blob + pl->sec_off[VQA_SEC_WG_DY_C2];
        if ((rc = vqa_wgrad_tiles(&w, 9, stream))) return rc;
        if ((rc = vqa_wgrad_reduce(w.partial, w.items, w.num_items, c->V, pl->C, cc, 9, c->g_conv2, w33, bias_rows, c->gb_conv2, stream))) return rc;
    }
    if (pl->sec_cnt[VQA_SEC_WG_PITEMS] > 0) {
        w.items = (const vqa_wgrad_item*)(c->blob + pl->sec_off[VQA_SEC_WG_PITEMS]);
        w.num_items = pl->sec_cnt[VQA_SEC_WG_PITEMS];
        w.total_nodes = pl->num_binary;
        w.src_slots = c->blob + pl->sec_off[VQA_SEC_WG_PSRC0];
        w.src1_slots = c->blob + pl->sec_off[VQA_SEC_WG_PSRC1];
        w.dy_slots = c->blob + pl->sec_off[VQA_SEC_WG_PDY];
        w.k_chunks = 2 * cc; w.pair_chunks = cc;
        w.bias_block = w.num_items * (pl->C / 64) * 2 * cc;
        bias_rows = w.partial + (int64_t)w.bias_block * 8192;
        if ((rc = vqa_wgrad_tiles(&w, 1, stream))) return rc;
        if ((rc = vqa_wgrad_reduce(w.partial, w.items, w.num_items, c->V, pl->C, 2 * cc, 1, c->g_proj, w11, bias_rows, c->gb_proj, stream))) return rc;
    } else if (c->g_proj) {
        hipError_t e = hipMemsetAsync(c->g_proj, 0, (size_t)c->V * w11 * sizeof(float), (hipStream_t)stream);
        if (e == hipSuccess && c->gb_proj) e = hipMemsetAsync(c->gb_proj, 0, (size_t)c->V * pl->C * sizeof(float), (hipStream_t)stream);
        if (e != hipSuccess) { vqa_set_error("vqa_tree_backward: memset failed"); return VQA_ELAUNCH; }
    }
    return VQA_OK;
}
