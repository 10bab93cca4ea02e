// Program scheduler: parse prefix programs with the reference interpreter's rules, bucket the tree
// nodes of the whole batch by (level, token), and lay out the int32 descriptor blob the device
// kernels index.  Host code only.
//
// Follows Module_Model.py:196-222 (forward_modules_ints_helper):
//   * a column past the end of the row is an implicit `scene` that is not marked used (:201-203)
//   * <START> is skipped wherever it appears (:206-208)
//   * <NULL> is not marked used but still runs as a unary module (:204-205, then :215-221)
//   * every other token, <END> and <UNK> included, is marked used and runs with the arity of
//     Program_funcs.py:114-122
//   * tokens after the root's subtree is complete are never read
#include "plan.h"

#include <algorithm>
#include <cstring>
#include <mutex>
#include <new>

thread_local std::string g_vqa_error;

void vqa_set_error(const std::string& s) { g_vqa_error = s; }

extern "C" const char* vqa_last_error(void) { return g_vqa_error.c_str(); }
extern "C" int vqa_version(void) { return 100; }

namespace {

struct Frame { int node; int need; };

// Freed plans are kept (a few of them) and reused by the next build: their node table and blob are
// megabytes, and getting fresh pages for them from the kernel on every batch costs more than the
// scheduling itself.
std::mutex g_pool_mu;
std::vector<vqa_plan*> g_pool;

vqa_plan* plan_acquire() {
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (!g_pool.empty()) {
            vqa_plan* p = g_pool.back();
            g_pool.pop_back();
            return p;
        }
    }
    return new (std::nothrow) vqa_plan();
}

void plan_release(vqa_plan* p) {
    if (!p) return;
    // back to the state of a new plan, capacities kept
    p->N = p->T = p->C = p->V = p->M = 0;
    p->shared = false;
    p->interned.clear();
    p->nodes.clear(); p->pending.clear(); p->roots.clear(); p->used.clear();
    p->bin_index.clear(); p->scene_index.clear(); p->token_counts.clear(); p->rep.clear(); p->first_scene.clear();
    p->num_exec = 0;
    for (auto& lv : p->level_nodes) lv.clear();
    p->levels.clear(); p->blob.unlend();
    p->num_scene = p->num_unary = p->num_binary = p->num_levels = p->max_level_tiles = 0;
    p->slot_F = p->slot_S0 = p->slot_Ysc = p->slot_Hsc = p->slot_Y = p->slot_H = p->slot_P = p->slots_fwd = 0;
    p->slot_dHsc = p->slot_dZ = p->slot_dH = p->slot_dP = p->slot_dFp = p->slot_dF = p->slot_dS0 = p->slots_total = 0;
    p->has_separate = p->has_fused = true;
    p->scene_id = p->wgrad_chunk = p->flags = p->fwd_words = 0;
    p->finished = false;
    p->blob_bound = 0;
    std::memset(p->sec_off, 0, sizeof(p->sec_off));
    std::memset(p->sec_cnt, 0, sizeof(p->sec_cnt));
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (g_pool.size() < 4) g_pool.push_back(p);
    else delete p;
}

// A finished node becomes a row of the node table - or, when samples share images, the row that
// already holds the same computation: same image, same token, same (already interned) inputs give
// the same feature map, so it is computed once (common subtrees of the candidate programs of one
// question, Train_model.py:339-347).
int add_node(vqa_plan* pl, const PlanNode& nd) {
    if (!pl->shared) {
        pl->nodes.push_back(nd);
        return (int)pl->nodes.size() - 1;
    }
    const uint64_t key = ((uint64_t)(uint32_t)nd.token << 56) | ((uint64_t)(uint32_t)nd.sample << 38) |
                         ((uint64_t)(uint32_t)(nd.child0 + 1) << 19) | (uint64_t)(uint32_t)(nd.child1 + 1);
    auto it = pl->interned.find(key);
    if (it != pl->interned.end()) return it->second;
    pl->nodes.push_back(nd);
    const int id = (int)pl->nodes.size() - 1;
    pl->interned.emplace(key, id);
    return id;
}

// Iterative form of the recursive helper: a stack of open nodes waiting for inputs.
int parse_row(const int64_t* row, int T, const int8_t* arity, int V, int null_id, int start_id,
              int scene_id, int sample, int image, vqa_plan* pl, float* used, std::vector<Frame>& stack) {
    stack.clear();
    int j = 0;
    const int base = (int)pl->nodes.size();
    for (;;) {
        int tok, col;
        // read one token (skipping <START>) or fall off the row
        for (;;) {
            if (j < T) {
                int64_t v = row[j];
                if (v < 0 || v >= V) {
                    vqa_set_error("program token id " + std::to_string((long long)v) +
                                  " outside the vocabulary (sample " + std::to_string(sample) +
                                  ", column " + std::to_string(j) + ")");
                    return VQA_EVOCAB;
                }
                tok = (int)v;
                if (tok == start_id) { j++; continue; }
                col = j;
                if (tok != null_id) used[j] = 1.0f;
                j++;
            } else {
                tok = scene_id;
                col = -1;
            }
            break;
        }
        PlanNode nd;
        nd.sample = image; nd.token = tok; nd.col = col; nd.child0 = nd.child1 = -1; nd.level = 0;
        const int ar = (tok == scene_id) ? 0 : arity[tok];
        // post-order numbering: a node gets its id when it completes, so reserve via the stack
        if (ar == 0) {
            int id = add_node(pl, nd);
            // deliver to parents
            for (;;) {
                if (stack.empty()) { pl->roots.push_back(id); (void)base; return VQA_OK; }
                Frame& f = stack.back();
                PlanNode& par = pl->pending[f.node];
                if (par.child0 < 0) par.child0 = id; else par.child1 = id;
                par.level = std::max(par.level, pl->nodes[id].level + 1);
                if (--f.need > 0) break;
                // parent complete: it becomes a real node now
                int pidx = f.node;
                stack.pop_back();
                id = add_node(pl, pl->pending[pidx]);
            }
        } else {
            pl->pending.push_back(nd);
            stack.push_back({(int)pl->pending.size() - 1, ar});
        }
    }
}

inline void push_tile(PlanBlob& b, int s0, int s1, int d0, int d1, int res, int m0, int m1,
                      int wg) {
    const size_t at = b.size();
    b.resize(at + 8);            // capacity is reserved up front: no reallocation here
    int32_t* t = b.data() + at;
    t[0] = s0; t[1] = s1; t[2] = d0; t[3] = d1; t[4] = res; t[5] = m0; t[6] = m1; t[7] = wg;
}

// arena slots of a node's maps; all scene nodes of an image share the image's scene slots
inline int y_slot(const vqa_plan* pl, int v) {
    const PlanNode& nd = pl->nodes[v];
    return nd.token == pl->scene_id ? pl->slot_Ysc + nd.sample : pl->slot_Y + v;
}
inline int h_slot(const vqa_plan* pl, int v) {
    const PlanNode& nd = pl->nodes[v];
    return nd.token == pl->scene_id ? pl->slot_Hsc + nd.sample : pl->slot_H + v;
}
inline int dh_slot(const vqa_plan* pl, int v) {
    const PlanNode& nd = pl->nodes[v];
    return nd.token == pl->scene_id ? pl->slot_dHsc + nd.sample : pl->slot_dH + v;
}
inline int in_slot(const vqa_plan* pl, int v) {   // slot the residual body of node v reads
    const PlanNode& nd = pl->nodes[v];
    if (nd.token == pl->scene_id) return pl->slot_F + nd.sample;
    if (nd.child1 >= 0) return pl->slot_P + pl->bin_index[v];
    return y_slot(pl, nd.child0);
}

// Forward half of the blob: what vqa_tree_forward and the classifier's gather read.  It comes first
// so that a caller can upload it and start the forward launches while the backward half is still
// being written (VQA_PLAN_DEFER_BACKWARD, vqa_plan_finish).
void build_forward_part(vqa_plan* pl) {
    PlanBlob& b = pl->blob;
    const int N = pl->N, M = pl->M;
    pl->levels.assign(pl->num_levels, {});
    int max_tiles = 0;
    for (int L = 0; L < pl->num_levels; L++) {
        PlanLevel& lv = pl->levels[L];
        const std::vector<int>& nodes = pl->level_nodes[L];
        lv.count = (int)nodes.size();
        max_tiles = std::max(max_tiles, lv.count);
        // binary nodes of the level keep the same (token, id) order
        lv.bcount = 0;
        for (int v : nodes) lv.bcount += pl->nodes[v].child1 >= 0;
        if (pl->has_separate) {   // one tile list per convolution pass
            lv.off_proj = (int)b.size();
            for (int v : nodes) {
                const PlanNode& nd = pl->nodes[v];
                if (nd.child1 < 0) continue;
                push_tile(b, y_slot(pl, nd.child0), y_slot(pl, nd.child1),
                          pl->slot_P + pl->bin_index[v], -1, -1, -1, -1, nd.token);
            }
            lv.off_conv1 = (int)b.size();
            for (int v : nodes)
                push_tile(b, in_slot(pl, v), -1, h_slot(pl, v), -1, -1, -1, -1, pl->nodes[v].token);
            lv.off_conv2 = (int)b.size();
            for (int v : nodes)
                push_tile(b, h_slot(pl, v), -1, y_slot(pl, v), -1, in_slot(pl, v), -1, -1,
                          pl->nodes[v].token);
        }
        if (pl->has_fused) {
            // the same passes as one fused tile per node (vqa_resblock_tile, 12 words):
            // [children -> P] -> H -> Y
            lv.off_fused_fwd = (int)b.size();
            b.resize(b.size() + (size_t)12 * nodes.size());   // within the reserved capacity
            int32_t* w = b.data() + lv.off_fused_fwd;
            const PlanNode* N_ = pl->nodes.data();
            const int sid = pl->scene_id;
            for (int v : nodes) {
                const PlanNode& nd = N_[v];
                const int c0 = nd.child0 < 0 ? 0 : nd.child0, c1 = nd.child1 < 0 ? 0 : nd.child1;
                const bool sc = nd.token == sid, bin = nd.child1 >= 0;
                const int y0 = N_[c0].token == sid ? pl->slot_Ysc + N_[c0].sample : pl->slot_Y + c0;
                const int y1 = N_[c1].token == sid ? pl->slot_Ysc + N_[c1].sample : pl->slot_Y + c1;
                const int in = sc ? pl->slot_F + nd.sample : bin ? pl->slot_P + pl->bin_index[v] : y0;
                w[0] = bin ? y0 : in; w[1] = bin ? y1 : -1;
                w[2] = sc ? pl->slot_Ysc + nd.sample : pl->slot_Y + v;
                w[3] = sc ? pl->slot_Hsc + nd.sample : pl->slot_H + v;
                w[4] = in; w[5] = -1; w[6] = -1; w[7] = nd.token;
                w[8] = w[9] = w[10] = w[11] = -1;
                w += 12;
            }
        }
    }
    pl->max_level_tiles = max_tiles;
    auto sec = [&](int id) { pl->sec_off[id] = (int)b.size(); };
    auto endsec = [&](int id, int count) { pl->sec_cnt[id] = count; };
    sec(VQA_SEC_STEM0_TILES);
    for (int i = 0; i < M; i++) push_tile(b, i, -1, pl->slot_S0 + i, -1, -1, -1, -1, 0);
    endsec(VQA_SEC_STEM0_TILES, M);
    sec(VQA_SEC_STEM1_TILES);
    for (int i = 0; i < M; i++) push_tile(b, pl->slot_S0 + i, -1, pl->slot_F + i, -1, -1, -1, -1, 0);
    endsec(VQA_SEC_STEM1_TILES, M);
    sec(VQA_SEC_CLS_FWD_TILES);
    for (int i = 0; i < N; i++) push_tile(b, y_slot(pl, pl->roots[i]), -1, i, -1, -1, -1, -1, 0);
    endsec(VQA_SEC_CLS_FWD_TILES, N);
}

// Backward half: input-gradient tile lists per level, per-sample sections, weight-gradient work
// items.  Appended behind the forward half; nothing written earlier moves.
void build_backward_part(vqa_plan* pl) {
    PlanBlob& b = pl->blob;
    const int N = pl->N, NV = (int)pl->nodes.size(), V = pl->V, wgrad_chunk = pl->wgrad_chunk;
    const int scene_id = pl->scene_id;
    pl->finished = true;
    std::vector<int> parent(NV, -1);     // plans with a backward part share no nodes: one user each
    for (int v = 0; v < NV; v++)
        for (int c : {pl->nodes[v].child0, pl->nodes[v].child1}) if (c >= 0) parent[c] = v;
    for (int L = 0; L < pl->num_levels; L++) {
        PlanLevel& lv = pl->levels[L];
        const std::vector<int>& nodes = pl->level_nodes[L];
        auto target = [&](int v, int& dst, int& mask) {
            // d(input) lands masked by (input > 0): a scene node's input is the image's stem map
            const PlanNode& nd = pl->nodes[v];
            if (nd.token == scene_id) { dst = pl->slot_dF + nd.sample; mask = pl->slot_F + nd.sample; }
            else if (nd.child1 >= 0) { dst = pl->slot_dP + pl->bin_index[v]; mask = pl->slot_P + pl->bin_index[v]; }
            else { dst = pl->slot_dZ + nd.child0; mask = y_slot(pl, nd.child0); }
        };
        if (pl->has_separate) {
            // dH = dgrad_conv2(dZ) * (H > 0)
            lv.off_dgrad2 = (int)b.size();
            for (int v : nodes)
                push_tile(b, pl->slot_dZ + v, -1, dh_slot(pl, v), -1, -1, h_slot(pl, v), -1, pl->nodes[v].token);
            // d(input) = (dgrad_conv1(dH) + dZ) * (input > 0)
            lv.off_dgrad1 = (int)b.size();
            for (int v : nodes) {
                int dst, mask;
                target(v, dst, mask);
                push_tile(b, dh_slot(pl, v), -1, dst, -1, pl->slot_dZ + v, mask, -1, pl->nodes[v].token);
            }
            lv.off_dproj = (int)b.size();
            for (int v : nodes) {
                const PlanNode& nd = pl->nodes[v];
                if (nd.child1 < 0) continue;
                push_tile(b, pl->slot_dP + pl->bin_index[v], -1, pl->slot_dZ + nd.child0,
                          pl->slot_dZ + nd.child1, -1, y_slot(pl, nd.child0), y_slot(pl, nd.child1), nd.token);
            }
        }
        if (pl->has_fused) {   // [dP of the parent -> dZ] -> dH -> dX as one tile
            // A binary node's dX is its dP; each input of it that is not a scene node takes its half of
            // the transposed projection in front of its own body (pre_group = 2 * parent token + half,
            // masked by its own output), so src0 is the parent's dP and the dZ slot is written, not read.
            lv.off_fused_bwd = (int)b.size();
            lv.half_count = 0;
            for (int v : nodes) {
                const PlanNode& nd = pl->nodes[v];
                int dst, mask;
                target(v, dst, mask);
                const int par = nd.token == scene_id ? -1 : parent[v];
                const bool half = par >= 0 && pl->nodes[par].child1 >= 0;
                lv.half_count += half;
                const int32_t w[12] = {half ? pl->slot_dP + pl->bin_index[par] : pl->slot_dZ + v, -1, dst, dh_slot(pl, v),
                                       pl->slot_dZ + v, mask, h_slot(pl, v), nd.token,
                                       half ? 2 * pl->nodes[par].token + (pl->nodes[par].child1 == v) : -1,
                                       half ? y_slot(pl, v) : -1, 0, 0};
                b.insert(b.end(), w, w + 12);
            }
        }
    }

    auto sec = [&](int id) { pl->sec_off[id] = (int)b.size(); };
    auto endsec = [&](int id, int count) { pl->sec_cnt[id] = count; };
    sec(VQA_SEC_STEM1_DGRAD_TILES);
    for (int i = 0; i < N; i++)
        push_tile(b, pl->slot_dF + i, -1, pl->slot_dS0 + i, -1, -1, pl->slot_S0 + i, -1, 0);
    endsec(VQA_SEC_STEM1_DGRAD_TILES, N);
    sec(VQA_SEC_CLS_DGRAD_TILES);
    for (int i = 0; i < N; i++)
        push_tile(b, i, -1, pl->slot_dZ + pl->roots[i], -1, -1, y_slot(pl, pl->roots[i]), -1, 0);
    endsec(VQA_SEC_CLS_DGRAD_TILES, N);
    // Scene inputs of binary nodes: a scene node runs once per image for all of its users, so nobody's
    // tile can take the projection half in front; one 1x1 launch ahead of the scene level does it for
    // every such pair of the batch.
    sec(VQA_SEC_SCENE_DPROJ_TILES);
    int npairs = 0;
    if (pl->has_fused) {
        for (int v = 0; v < NV; v++) {
            const PlanNode& nd = pl->nodes[v];
            if (nd.child1 < 0) continue;
            const bool s0 = pl->nodes[nd.child0].token == scene_id, s1 = pl->nodes[nd.child1].token == scene_id;
            if (!s0 && !s1) continue;
            push_tile(b, pl->slot_dP + pl->bin_index[v], -1, s0 ? pl->slot_dZ + nd.child0 : -1,
                      s1 ? pl->slot_dZ + nd.child1 : -1, -1, s0 ? y_slot(pl, nd.child0) : -1,
                      s1 ? y_slot(pl, nd.child1) : -1, nd.token);
            npairs++;
        }
    }
    endsec(VQA_SEC_SCENE_DPROJ_TILES, npairs);

    auto slot_list = [&](int id, int base_slot) {
        sec(id);
        for (int i = 0; i < N; i++) b.push_back(base_slot + i);
        endsec(id, N);
    };
    slot_list(VQA_SEC_SLOTS_IDENT, 0);
    slot_list(VQA_SEC_SLOTS_S0, pl->slot_S0);
    slot_list(VQA_SEC_SLOTS_F, pl->slot_F);
    slot_list(VQA_SEC_SLOTS_DF, pl->slot_dF);
    slot_list(VQA_SEC_SLOTS_DS0, pl->slot_dS0);
    sec(VQA_SEC_SLOTS_ROOTY);
    for (int i = 0; i < N; i++) b.push_back(y_slot(pl, pl->roots[i]));
    endsec(VQA_SEC_SLOTS_ROOTY, N);

    // sample-chunk work items (one weight group) for the stem / classifier weight gradients
    sec(VQA_SEC_SAMPLE_ITEMS);
    int nitems = 0;
    for (int s = 0; s < N; s += wgrad_chunk, nitems++) {
        const int32_t it[4] = {0, s, std::min(wgrad_chunk, N - s), 0};
        b.insert(b.end(), it, it + 4);
    }
    endsec(VQA_SEC_SAMPLE_ITEMS, nitems);
    // the same in finer chunks (at least 128 items when N allows), for launches whose grid is
    // items x 2 only (stem conv 2: 128 source channels) and would otherwise leave most CUs idle
    sec(VQA_SEC_SAMPLE_ITEMS_FINE);
    const int fine = std::max(1, std::min(wgrad_chunk, N / 128));
    nitems = 0;
    for (int s = 0; s < N; s += fine, nitems++) {
        const int32_t it[4] = {0, s, std::min(fine, N - s), 0};
        b.insert(b.end(), it, it + 4);
    }
    endsec(VQA_SEC_SAMPLE_ITEMS_FINE, nitems);

    // scene fan-in (CSR by sample): the dZ slots of a sample's scene nodes (node order) ...
    {
        std::vector<int> cnt(N + 1, 0);
        for (int v = 0; v < NV; v++) if (pl->scene_index[v] >= 0) cnt[pl->nodes[v].sample + 1]++;
        for (int i = 0; i < N; i++) cnt[i + 1] += cnt[i];
        sec(VQA_SEC_SCENE_OFFSETS);
        b.insert(b.end(), cnt.begin(), cnt.end());
        endsec(VQA_SEC_SCENE_OFFSETS, N + 1);
        sec(VQA_SEC_SCENE_SRC);
        const size_t at = b.size();
        b.resize(at + cnt[N]);
        for (int v = 0; v < NV; v++)
            if (pl->scene_index[v] >= 0) b[at + cnt[pl->nodes[v].sample]++] = pl->slot_dZ + v;
        endsec(VQA_SEC_SCENE_SRC, (int)(b.size() - at));
    }
    // ... summed into the representative's dZ slot, times (its output > 0)
    sec(VQA_SEC_SLOTS_SCDZ);
    for (int i = 0; i < N; i++) b.push_back(pl->slot_dZ + pl->first_scene[i]);
    endsec(VQA_SEC_SLOTS_SCDZ, N);
    sec(VQA_SEC_SLOTS_SCY);
    for (int i = 0; i < N; i++) b.push_back(pl->slot_Ysc + i);
    endsec(VQA_SEC_SLOTS_SCY, N);

    // module weight gradients: nodes grouped by token, cut into items of <= wgrad_chunk nodes
    std::vector<std::vector<int>> by_tok(V);
    for (int v = 0; v < NV; v++) if (pl->rep[v] == v) by_tok[pl->nodes[v].token].push_back(v);
    std::vector<int32_t> items, src_c1, dy_c1, src_c2, dy_c2;
    std::vector<int32_t> pitems, psrc0, psrc1, pdy;
    src_c1.reserve(NV); dy_c1.reserve(NV); src_c2.reserve(NV); dy_c2.reserve(NV);
    for (int t = 0; t < V; t++) {
        const std::vector<int>& vs = by_tok[t];
        if (vs.empty()) continue;
        for (size_t s = 0; s < vs.size(); s += wgrad_chunk) {
            const int32_t it[4] = {t, (int32_t)src_c1.size(),
                                   (int32_t)std::min<size_t>(wgrad_chunk, vs.size() - s), 0};
            items.insert(items.end(), it, it + 4);
            for (size_t k = s; k < s + it[2]; k++) {
                int v = vs[k];
                src_c1.push_back(in_slot(pl, v)); dy_c1.push_back(dh_slot(pl, v));
                src_c2.push_back(h_slot(pl, v)); dy_c2.push_back(pl->slot_dZ + v);
            }
        }
        if (pl->nodes[vs[0]].child1 >= 0) {
            for (size_t s = 0; s < vs.size(); s += wgrad_chunk) {
                const int32_t it[4] = {t, (int32_t)psrc0.size(),
                                       (int32_t)std::min<size_t>(wgrad_chunk, vs.size() - s), 0};
                pitems.insert(pitems.end(), it, it + 4);
                for (size_t k = s; k < s + it[2]; k++) {
                    int v = vs[k];
                    psrc0.push_back(y_slot(pl, pl->nodes[v].child0));
                    psrc1.push_back(y_slot(pl, pl->nodes[v].child1));
                    pdy.push_back(pl->slot_dP + pl->bin_index[v]);
                }
            }
        }
    }
    // longest items first: workgroups are dispatched in index order as CUs free up, so this is the
    // greedy longest-processing-time schedule and the launch does not end on a late long item
    auto by_count = [](std::vector<int32_t>& its) {
        struct It { int32_t w[4]; };
        It* a = reinterpret_cast<It*>(its.data());
        std::stable_sort(a, a + its.size() / 4, [](const It& x, const It& y) { return x.w[2] > y.w[2]; });
    };
    by_count(items);
    by_count(pitems);
    auto put = [&](int id, const std::vector<int32_t>& v, int count) {
        sec(id); b.insert(b.end(), v.begin(), v.end()); endsec(id, count);
    };
    put(VQA_SEC_WG_ITEMS, items, (int)items.size() / 4);
    put(VQA_SEC_WG_SRC_C1, src_c1, (int)src_c1.size());
    put(VQA_SEC_WG_DY_C1, dy_c1, (int)dy_c1.size());
    put(VQA_SEC_WG_SRC_C2, src_c2, (int)src_c2.size());
    put(VQA_SEC_WG_DY_C2, dy_c2, (int)dy_c2.size());
    put(VQA_SEC_WG_PITEMS, pitems, (int)pitems.size() / 4);
    put(VQA_SEC_WG_PSRC0, psrc0, (int)psrc0.size());
    put(VQA_SEC_WG_PSRC1, psrc1, (int)psrc1.size());
    put(VQA_SEC_WG_PDY, pdy, (int)pdy.size());
    // Both 3x3 weight gradients as one launch: a launch of 300-odd items x 4 workgroups is two and a half rounds of
    // the device and ends on a round that is less than half full; one launch of twice the items ends on one such
    // round instead of two.  Items of the second convolution: group V + token, slots in the second half of the lists.
    {
        std::vector<int32_t> items12(items), src12(src_c1), dy12(dy_c1);
        const int32_t half = (int32_t)src_c1.size();
        for (size_t k = 0; k < items.size(); k += 4) {
            const int32_t it[4] = {items[k] + V, items[k + 1] + half, items[k + 2], 0};
            items12.insert(items12.end(), it, it + 4);
        }
        src12.insert(src12.end(), src_c2.begin(), src_c2.end());
        dy12.insert(dy12.end(), dy_c2.begin(), dy_c2.end());
        by_count(items12);
        put(VQA_SEC_WG_ITEMS_12, items12, (int)items12.size() / 4);
        put(VQA_SEC_WG_SRC_12, src12, (int)src12.size());
        put(VQA_SEC_WG_DY_12, dy12, (int)dy12.size());
    }
}

}  // namespace

// Re-assign tree levels so that launches come in whole rounds of the device.
//
// The earliest-possible level of a node (1 + its deepest child) makes the first levels huge and the
// last dozen tiny, and a level of n tiles costs about ceil(n / round) passes over the device
// (round = 256 CUs x 2 resident workgroups): 538 tiles cost two passes, 20 tiles still cost a full
// tile latency.  Only the deepest programs are critical, though; every other node has slack up to
// its latest level (its parent's latest level - 1, the last level for a root).  List scheduling
// over the same number of levels: each level takes every node that is due, then the ready nodes
// with the least slack until it holds the even share of what is left, rounded up to whole rounds
// (half a round once the share is that small).  Node ids, slots and results do not change; only
// the launch a node runs in.
static void balance_levels(vqa_plan* pl, int round, int scene_id) {
    const int NV = (int)pl->nodes.size(), L = pl->num_levels;
    if (L <= 2 || round <= 0) return;
    // Scene nodes stay at level 0 (one launch over the images, before anything else); the rest is
    // scheduled over levels 1..L-1.  A node may have several users (interned nodes of a shared
    // image): users as a CSR list.  Which inputs a node has is as good as random to a branch
    // predictor, so the passes over the nodes select instead of branching: an input that is absent
    // or a scene node counts as input NV, a row of the tables that nothing reads.  The tables live in
    // one scratch array that the plan keeps from build to build.
    PlanNode* nodes = pl->nodes.data();
    std::vector<int32_t>& w = pl->work;
    w.resize((size_t)11 * NV + 2 * L + 16);
    int32_t* e0 = w.data();              // [NV] first input that is scheduled here, or NV
    int32_t* e1 = e0 + NV;               // [NV] second
    int32_t* nuse = e1 + NV;             // [NV + 2] CSR row starts
    int32_t* pending = nuse + NV + 2;    // [NV + 1] inputs not yet placed
    int32_t* latest = pending + NV + 1;  // [NV + 1] last level the node may run in
    int32_t* fill = latest + NV + 1;     // [NV + 1] CSR fill cursors
    int32_t* users = fill + NV + 1;      // [2 NV] every node files itself twice, under NV if need be
    int32_t* take = users + 2 * NV;      // [NV]
    int32_t* next = take + NV;           // [NV + L] ready nodes: one first-in first-out chain per latest
                                         // level, entry NV + b being the head of chain b
    int32_t* tail = next + NV + L;       // [L]
    std::fill(nuse, nuse + NV + 2, 0);
    for (int v = 0; v < NV; v++) {
        const PlanNode& nd = nodes[v];
        const int c0 = nd.child0, c1 = nd.child1;
        const int ok0 = (c0 >= 0) & (nodes[c0 < 0 ? 0 : c0].token != scene_id);
        const int ok1 = (c1 >= 0) & (nodes[c1 < 0 ? 0 : c1].token != scene_id);
        const int a = ok0 ? c0 : NV, b = ok1 ? c1 : NV;
        e0[v] = a; e1[v] = b;
        nuse[a + 1]++; nuse[b + 1]++;
        pending[v] = ok0 + ok1;
        latest[v] = L - 1;
    }
    latest[NV] = L - 1;
    for (int v = 0; v <= NV; v++) { fill[v] = nuse[v]; nuse[v + 1] += nuse[v]; }
    for (int v = 0; v < NV; v++) { users[fill[e0[v]]++] = v; users[fill[e1[v]]++] = v; }
    for (int v = NV - 1; v >= 0; v--) {          // a user's id is larger than its inputs'
        const int lv = latest[v] - 1;
        latest[e0[v]] = std::min(latest[e0[v]], lv);
        latest[e1[v]] = std::min(latest[e1[v]], lv);
    }
    for (int b = 0; b < L; b++) { next[NV + b] = -1; tail[b] = NV + b; }
    auto push = [&](int v) {
        const int b = latest[v];
        next[v] = -1;
        next[tail[b]] = v;
        tail[b] = v;
    };
    int total = 0, nready = 0;
    for (int v = 0; v < NV; v++) {
        const int mine = nodes[v].token != scene_id;
        total += mine;
        take[nready] = v;
        nready += mine & (pending[v] == 0);
    }
    for (int i = 0; i < nready; i++) push(take[i]);
    int done = 0, lowest = 1;
    for (int t = 1; t < L; t++) {
        const int remaining = total - done, share = (remaining + (L - t) - 1) / (L - t);
        const int target = share <= round / 2 ? round / 2 : (share + round - 1) / round * round;
        int ntake = 0;
        while (lowest < L) {
            const int v = next[NV + lowest];
            if (v < 0) { lowest++; continue; }
            if (lowest > t && ntake >= target) break;
            next[NV + lowest] = next[v];
            tail[lowest] = next[v] < 0 ? NV + lowest : tail[lowest];
            take[ntake++] = v;
        }
        for (int i = 0; i < ntake; i++) {
            const int v = take[i];
            nodes[v].level = t;
            for (int k = nuse[v]; k < nuse[v + 1]; k++) {
                const int p = users[k];
                if (--pending[p] == 0) {
                    push(p);
                    lowest = std::min(lowest, latest[p]);
                }
            }
        }
        done += ntake;
    }
}

extern "C" int vqa_schedule_build(const int64_t* programs, int32_t N, int32_t T,
                                  const int8_t* arity, int32_t V, int32_t null_id,
                                  int32_t start_id, int32_t scene_id, int32_t C,
                                  int32_t wgrad_chunk, int32_t flags, vqa_plan** out) {
    return vqa_schedule_build_into(programs, N, T, arity, V, null_id, start_id, scene_id, C, wgrad_chunk,
                                   flags, nullptr, N, nullptr, 0, out);
}

extern "C" int vqa_schedule_build_shared(const int64_t* programs, int32_t N, int32_t T,
                                         const int8_t* arity, int32_t V, int32_t null_id,
                                         int32_t start_id, int32_t scene_id, int32_t C,
                                         int32_t wgrad_chunk, int32_t flags, const int32_t* image_of_sample,
                                         int32_t num_images, vqa_plan** out) {
    return vqa_schedule_build_into(programs, N, T, arity, V, null_id, start_id, scene_id, C, wgrad_chunk,
                                   flags, image_of_sample, num_images, nullptr, 0, out);
}

extern "C" int vqa_schedule_build_into(const int64_t* programs, int32_t N, int32_t T,
                                       const int8_t* arity, int32_t V, int32_t null_id,
                                       int32_t start_id, int32_t scene_id, int32_t C,
                                       int32_t wgrad_chunk, int32_t flags, const int32_t* image_of_sample,
                                       int32_t num_images, int32_t* blob_store, int64_t blob_store_words,
                                       vqa_plan** out) {
    if (!programs || !arity || !out || N <= 0 || T <= 0 || V <= 0 || V > 255 || scene_id < 0 ||
        scene_id >= V || C <= 0 || C % 128 != 0 || wgrad_chunk <= 0 ||
        (image_of_sample ? (num_images <= 0 || num_images >= (1 << 18)) : num_images != N) ||
        (blob_store && blob_store_words < 0)) {
        vqa_set_error("vqa_schedule_build: bad argument");
        return VQA_EINVAL;
    }
    if (image_of_sample) {
        for (int i = 0; i < N; i++)
            if (image_of_sample[i] < 0 || image_of_sample[i] >= num_images) {
                vqa_set_error("vqa_schedule_build_shared: image index out of range");
                return VQA_EINVAL;
            }
        if ((int64_t)N * (2 * T + 1) >= (1 << 19) - 1) {
            vqa_set_error("vqa_schedule_build_shared: batch too large for shared-image scheduling");
            return VQA_EINVAL;
        }
    }
    vqa_plan* pl = plan_acquire();
    if (!pl) { vqa_set_error("out of host memory"); return VQA_EINVAL; }
    pl->N = N; pl->T = T; pl->C = C; pl->V = V;
    pl->M = num_images; pl->shared = image_of_sample != nullptr;
    pl->used.assign((size_t)N * T, 0.0f);
    pl->nodes.reserve((size_t)N * (T + 1));
    pl->roots.reserve(N);
    std::vector<Frame> stack;
    stack.reserve(2 * T + 2);
    for (int i = 0; i < N; i++) {
        pl->pending.clear();
        int rc = parse_row(programs + (size_t)i * T, T, arity, V, null_id, start_id, scene_id, i,
                           image_of_sample ? image_of_sample[i] : i, pl, pl->used.data() + (size_t)i * T, stack);
        if (rc) { plan_release(pl); return rc; }
    }
    pl->pending.clear();
    pl->interned.clear();
    const int M = pl->M;

    const int NV = (int)pl->nodes.size();
    // Kind indices, and the scene aliases: every scene node of an image computes the same map (same
    // module, same input F), so the first one runs and the others stand for it.  Node ids, and with
    // them everything a caller indexes by node, stay as the reference's execution order numbers them.
    pl->bin_index.resize(NV);
    pl->scene_index.resize(NV);
    pl->rep.resize(NV);
    pl->first_scene.assign(M, -1);
    pl->token_counts.assign(V, 0);
    int nb = 0, ns = 0, nu = 0, maxlev = 0, nexec = NV;
    {
        int32_t* bin_index = pl->bin_index.data();
        int32_t* scene_index = pl->scene_index.data();
        int32_t* rep = pl->rep.data();
        int32_t* first_scene = pl->first_scene.data();
        int32_t* token_counts = pl->token_counts.data();
        for (int v = 0; v < NV; v++) {     // selects, not branches: the kinds come in no order
            const PlanNode& nd = pl->nodes[v];
            token_counts[nd.token]++;
            const int sc = nd.token == scene_id, bin = nd.child1 >= 0;
            const int fs = first_scene[nd.sample], nfs = (sc & (fs < 0)) ? v : fs;
            first_scene[nd.sample] = nfs;
            const int r = sc ? nfs : v;
            scene_index[v] = sc ? ns : -1;
            bin_index[v] = bin ? nb : -1;
            rep[v] = r;
            ns += sc; nb += bin; nu += 1 - sc - bin;
            nexec -= r != v;
            maxlev = std::max(maxlev, nd.level);
        }
    }
    pl->num_scene = ns; pl->num_unary = nu; pl->num_binary = nb; pl->num_levels = maxlev + 1;
    pl->num_exec = nexec;

    if (flags & VQA_PLAN_BALANCE) balance_levels(pl, 512 / (C / 128), scene_id);

    // arena slots
    // F, S0 and the scene modules' maps come first, at places that depend on the image count only:
    // the stem and the scene level can run before the programs are parsed
    pl->slot_F = 0; pl->slot_S0 = M; pl->slot_Ysc = 2 * M; pl->slot_Hsc = 3 * M;
    pl->slot_Y = 4 * M; pl->slot_H = 4 * M + NV;
    pl->slot_P = 4 * M + 2 * NV; pl->slots_fwd = 4 * M + 2 * NV + nb;
    pl->slot_dZ = pl->slots_fwd; pl->slot_dH = pl->slot_dZ + NV; pl->slot_dHsc = pl->slot_dH + NV;
    pl->slot_dP = pl->slot_dHsc + M;
    pl->slot_dFp = 0; pl->slot_dF = pl->slot_dP + nb; pl->slot_dS0 = pl->slot_dF + M;
    pl->slots_total = pl->shared ? pl->slots_fwd : pl->slot_dS0 + M;
    pl->scene_id = scene_id; pl->wgrad_chunk = wgrad_chunk; pl->flags = flags;

    // level buckets: nodes of a level sorted by (token, node id) - one counting sort over
    // (level, token) keys, which ascending node ids keep stable; the aliased scene nodes go to a key
    // past the last level's
    if ((int)pl->level_nodes.size() < pl->num_levels) pl->level_nodes.resize(pl->num_levels);
    {
        const int NL = pl->num_levels, NK = NL * V;
        std::vector<int32_t>& w = pl->work;
        w.resize((size_t)NK + 1 + 2 * NV);
        int32_t* pos = w.data();       // [NK + 1]
        int32_t* key = pos + NK + 1;   // [NV]
        int32_t* flat = key + NV;      // [NV]
        std::fill(pos, pos + NK + 1, 0);
        const int32_t* rep = pl->rep.data();
        for (int v = 0; v < NV; v++) {
            const PlanNode& nd = pl->nodes[v];
            const int k = rep[v] == v ? nd.level * V + nd.token : NK;
            key[v] = k;
            pos[k]++;
        }
        int at = 0;
        for (int k = 0; k <= NK; k++) { const int n = pos[k]; pos[k] = at; at += n; }
        for (int L = 0; L < NL; L++) pl->level_nodes[L].resize((L + 1 < NL ? pos[(L + 1) * V] : pos[NK]) - pos[L * V]);
        const int first_alias = pos[NK];
        for (int v = 0; v < NV; v++) flat[pos[key[v]]++] = v;
        at = 0;
        for (int L = 0; L < NL; L++) {
            std::vector<int>& lv = pl->level_nodes[L];
            if (!lv.empty()) std::memcpy(lv.data(), flat + at, lv.size() * sizeof(int));
            at += (int)lv.size();
        }
        (void)first_alias;
    }
    pl->has_separate = !(flags & VQA_PLAN_SKIP_SEPARATE);
    pl->has_fused = !(flags & VQA_PLAN_SKIP_FUSED);
    // upper bound of the finished blob, so that a caller can size its buffers after the forward part
    pl->blob_bound = (size_t)NV * 88 + (size_t)N * 64 + (size_t)nb * 32 + (size_t)V * 32 + 1024;
    if (blob_store && (size_t)blob_store_words >= pl->blob_bound) pl->blob.lend(blob_store, (size_t)blob_store_words);
    else pl->blob.reserve(pl->blob_bound);
    build_forward_part(pl);
    pl->fwd_words = (int)pl->blob.size();
    pl->finished = pl->shared;      // forward-only plans have no backward part
    if (!pl->shared && !(flags & VQA_PLAN_DEFER_BACKWARD)) build_backward_part(pl);
    if (pl->blob.empty()) pl->blob.push_back(0);
    *out = pl;
    return VQA_OK;
}

extern "C" int vqa_plan_finish(vqa_plan* pl) {
    if (!pl) { vqa_set_error("null plan"); return VQA_EINVAL; }
    if (!pl->finished) build_backward_part(pl);
    return VQA_OK;
}

extern "C" void vqa_plan_free(vqa_plan* plan) { plan_release(plan); }

extern "C" int vqa_plan_get_info(const vqa_plan* pl, vqa_plan_info* o) {
    if (!pl || !o) { vqa_set_error("null plan"); return VQA_EINVAL; }
    std::memset(o, 0, sizeof(*o));
    o->N = pl->N; o->T = pl->T; o->C = pl->C;
    o->num_nodes = (int)pl->nodes.size();
    o->num_scene = pl->num_scene; o->num_unary = pl->num_unary; o->num_binary = pl->num_binary;
    o->num_levels = pl->num_levels;
    o->slots_fwd = pl->slots_fwd; o->slots_total = pl->slots_total;
    o->blob_words = (int)pl->blob.size();
    o->blob_lent = pl->blob.lent() ? 1 : 0;
    o->wgrad_items = pl->sec_cnt[VQA_SEC_WG_ITEMS];
    o->max_level_tiles = pl->max_level_tiles;
    o->num_images = pl->M;
    o->num_exec_nodes = pl->num_exec;
    o->blob_fwd_words = pl->fwd_words;
    o->blob_words_bound = (int)pl->blob_bound;
    {   // one block of 8192 floats per tap and workgroup of the largest weight-gradient launch, and
        // behind them one row of C bias sums per item
        const int cc = pl->C / 128, os = pl->C / 64;
        const int it = pl->sec_cnt[VQA_SEC_WG_ITEMS], pit = pl->sec_cnt[VQA_SEC_WG_PITEMS];
        auto rows = [&](int items) { return (int)(((int64_t)items * pl->C + 8191) / 8192); };
        // (the two 3x3 weight gradients of the tree run as one launch: twice the items)
        o->wgrad_partial_blocks = std::max(2 * it * os * cc * 9 + rows(2 * it), pit * os * 2 * cc + rows(pit));
    }
    return VQA_OK;
}

extern "C" int vqa_plan_used_fns(const vqa_plan* pl, float* out) {
    if (!pl || !out) { vqa_set_error("null argument"); return VQA_EINVAL; }
    std::memcpy(out, pl->used.data(), pl->used.size() * sizeof(float));
    return VQA_OK;
}

extern "C" int vqa_plan_nodes(const vqa_plan* pl, int32_t* out) {
    if (!pl || !out) { vqa_set_error("null argument"); return VQA_EINVAL; }
    for (size_t v = 0; v < pl->nodes.size(); v++) {
        const PlanNode& n = pl->nodes[v];
        int32_t* o = out + v * 6;
        o[0] = n.sample; o[1] = n.token; o[2] = n.col; o[3] = n.child0; o[4] = n.child1; o[5] = n.level;
    }
    return VQA_OK;
}

extern "C" int vqa_plan_level_nodes(const vqa_plan* pl, int32_t level, int32_t* out, int32_t cap) {
    if (!pl || level < 0 || level >= pl->num_levels) return -1;
    const std::vector<int>& lv = pl->level_nodes[level];
    if (out) for (int i = 0; i < (int)lv.size() && i < cap; i++) out[i] = lv[i];
    return (int)lv.size();
}

extern "C" int vqa_plan_token_counts(const vqa_plan* pl, int32_t* out) {
    if (!pl || !out) { vqa_set_error("null argument"); return VQA_EINVAL; }
    std::memcpy(out, pl->token_counts.data(), pl->token_counts.size() * sizeof(int32_t));
    return VQA_OK;
}

extern "C" int vqa_plan_roots(const vqa_plan* pl, int32_t* out) {
    if (!pl || !out) { vqa_set_error("null argument"); return VQA_EINVAL; }
    std::memcpy(out, pl->roots.data(), pl->roots.size() * sizeof(int32_t));
    return VQA_OK;
}

extern "C" int vqa_plan_blob_range(const vqa_plan* pl, int32_t first, int32_t count, int32_t* out) {
    if (!pl || !out || first < 0 || count < 0 || (size_t)first + (size_t)count > pl->blob.size()) {
        vqa_set_error("vqa_plan_blob_range: bad argument");
        return VQA_EINVAL;
    }
    std::memcpy(out, pl->blob.data() + first, (size_t)count * sizeof(int32_t));
    return VQA_OK;
}

extern "C" int vqa_plan_blob(const vqa_plan* pl, int32_t* out) {
    if (!pl || !out) { vqa_set_error("null argument"); return VQA_EINVAL; }
    std::memcpy(out, pl->blob.data(), pl->blob.size() * sizeof(int32_t));
    return VQA_OK;
}

extern "C" int vqa_plan_section(const vqa_plan* pl, int32_t id, int32_t* offset, int32_t* count) {
    if (!pl || id < 0 || id >= VQA_SEC_COUNT || !offset || !count) {
        vqa_set_error("vqa_plan_section: bad argument");
        return VQA_EINVAL;
    }
    *offset = pl->sec_off[id]; *count = pl->sec_cnt[id];
    return VQA_OK;
}

extern "C" int32_t vqa_plan_slot(const vqa_plan* pl, int32_t which, int32_t index) {
    if (!pl || index < 0) return -1;
    const int NV = (int)pl->nodes.size();
    switch (which) {
        case 0: return index < pl->N ? pl->slot_F + index : -1;
        case 1: return index < pl->N ? pl->slot_S0 + index : -1;
        case 2: case 3: case 5: {
            if (index >= NV) return -1;
            const PlanNode& nd = pl->nodes[index];
            const bool sc = pl->rep[index] != index || pl->first_scene[nd.sample] == index;   // a scene node
            if (which == 2) return sc ? pl->slot_Ysc + nd.sample : pl->slot_Y + index;
            if (which == 3) return sc ? pl->slot_Hsc + nd.sample : pl->slot_H + index;
            return sc ? pl->slot_dHsc + nd.sample : pl->slot_dH + index;
        }
        case 4: return index < NV ? pl->slot_dZ + index : -1;
        case 8: return index < pl->num_binary ? pl->slot_P + index : -1;
        case 6: return index < pl->N ? pl->slot_dF + index : -1;
        case 7: return index < pl->N ? pl->slot_dS0 + index : -1;
        default: return -1;
    }
}
