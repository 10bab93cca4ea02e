// Program scheduler on the device (gfx950): the parse of forward_modules_ints_helper
// (Module_Model.py:196-222) and the (level, token) bucketing of the whole batch as four small
// kernels, so that programs a sampler left in device memory (LSTM_Model.py:205-235 builds them on
// the GPU and copies them to the host every step, :228-235) become tile lists without a round
// trip through the host.  Forward only (candidate scoring and evaluation: Train_model.py:339-354,
// Functions.py:152-179).  Which launch a node runs in does not change its value: results are
// bit-identical to the host scheduler's, with or without either side's level balancing.
//
// The host never learns the node counts: the arena uses fixed strides (a row has at most T nodes
// that are not the scene module), the launch of a level has a fixed number of workgroups which
// share the level's tile range, read from device memory (vqa_resblock_tiles with `level_info`), and
// a token id outside the vocabulary - a KeyError in the reference and in the host scheduler - raises
// a flag in device memory instead (the row then runs as a lone scene module).
//
// A parse is a chain of dependent reads and writes, one short step per token: it runs one row per
// lane with all of its state in LDS (a round trip to HBM per step would cost 100 times the step).
#include <string>

#include "vqa_dev.h"
#include "plan.h"

#define USED_BIT (1 << 30)

// node record, 4 words: token | level << 8, input 0, input 1 (indices into the row's node list, -1
// none), rank among the row's nodes that are not the scene module | rank among its binary nodes << 16
//
// One lane per program row.  Iterative form of the recursive helper: a stack of open nodes that
// wait for inputs, finished nodes appended in execution (post-)order.  LDS, lane-minor:
//   toks [T][B]     the rows' tokens (-1: outside the vocabulary), USED_BIT once marked used
//   pend [T][3][B]  open nodes: token | level << 8, input 0, input 1
//   stack[T][B]     open node | inputs still missing << 16
//   node [2T+1][4][B]
//   arity[V]
__global__ __launch_bounds__(64) void devsched_parse_kernel(vqa_devsched_params p) {
    extern __shared__ int32_t lds[];
    const int T = p.T, B = blockDim.x, t = threadIdx.x;
    int32_t* toks = lds;
    int32_t* pend = toks + T * B;
    int32_t* stack = pend + 3 * T * B;
    int32_t* node = stack + T * B;
    int32_t* arity = node + (2 * T + 1) * 4 * B;     // [V]
    for (int i = t; i < p.V; i += B) arity[i] = p.arity[i];
    const int row0 = blockIdx.x * B, row = row0 + t;
    for (int idx = t; idx < B * T; idx += B) {
        const int r = idx / T, j = idx - r * T;
        int tok = p.null_id;
        if (row0 + r < p.N) {
            const int64_t v = p.programs[(int64_t)(row0 + r) * T + j];
            tok = (v < 0 || v >= p.V) ? -1 : (int)v;
        }
        toks[j * B + r] = tok;
    }
    __syncthreads();
    if (row < p.N) {
        int j = 0, n = 0, np = 0, sp = 0, root = 0;
        bool bad = false, bad_arity = false;
        for (;;) {
            int tok;
            for (;;) {
                if (j < T) {
                    tok = toks[j * B + t];
                    if (tok < 0) { bad = true; break; }
                    if (tok == p.start_id) { j++; continue; }                    // skipped, wherever it stands
                    if (tok != p.null_id) toks[j * B + t] = tok | USED_BIT;      // <NULL> runs as a module but is not marked
                    j++;
                } else {
                    tok = p.scene_id;                                            // reading past the row: an implicit scene
                }
                break;
            }
            if (bad) break;
            const int ar = tok == p.scene_id ? 0 : arity[tok];
            // a table that gives a module other than scene no input, or more than two: the node records
            // (two inputs, leaves are scene outputs) cannot hold that row
            if (tok != p.scene_id && (ar < 1 || ar > 2)) { bad = bad_arity = true; break; }
            if (ar == 0) {
                int id = n++;
                int lvl = 0;
                int32_t* nd = node + id * 4 * B + t;
                nd[0] = tok; nd[B] = -1; nd[2 * B] = -1;
                bool done = false;
                for (;;) {
                    if (sp == 0) { root = id; done = true; break; }
                    const int f = stack[(sp - 1) * B + t];
                    const int pi = f & 0xffff, need = (f >> 16) - 1;
                    int32_t* par = pend + pi * 3 * B + t;
                    if (par[B] < 0) par[B] = id; else par[2 * B] = id;
                    const int ptok = par[0] & 255, plvl = max(par[0] >> 8, lvl + 1);
                    par[0] = ptok | plvl << 8;
                    if (need > 0) { stack[(sp - 1) * B + t] = pi | need << 16; break; }
                    sp--;
                    id = n++;
                    lvl = plvl;
                    nd = node + id * 4 * B + t;
                    nd[0] = par[0]; nd[B] = par[B]; nd[2 * B] = par[2 * B];
                }
                if (done) break;
            } else {
                int32_t* q = pend + np * 3 * B + t;
                q[0] = tok; q[B] = -1; q[2 * B] = -1;
                stack[sp * B + t] = np | ar << 16;
                np++; sp++;
            }
        }
        if (bad) {
            atomicOr(p.error, bad_arity ? 4 : 1);
            for (int c = 0; c < T; c++) toks[c * B + t] = 0;
            n = 1; root = 0;
            node[t] = p.scene_id; node[B + t] = -1; node[2 * B + t] = -1;
        }
        int nv = 0, nb = 0;
        int4* out = (int4*)p.node + (int64_t)row * (2 * T + 1);
        for (int k = 0; k < n; k++) {
            const int32_t* nd = node + k * 4 * B + t;
            const int w0 = nd[0], c0 = nd[B], c1 = nd[2 * B];
            out[k] = make_int4(w0, c0, c1, nv | nb << 16);
            if ((w0 & 255) == p.scene_id) continue;
            nv++;
            nb += c1 >= 0;
        }
        const int root_level = node[root * 4 * B + t] >> 8;
        atomicMax(p.error + 1, root_level);
        *((int4*)p.counts + row) = make_int4(n, nv, nb, root | root_level << 16);
    }
    __syncthreads();
    for (int idx = t; idx < B * T; idx += B) {
        const int r = idx / T, j = idx - r * T;
        if (row0 + r < p.N) p.used[(int64_t)(row0 + r) * T + j] = (toks[j * B + r] & USED_BIT) ? 1.f : 0.f;
    }
}

// The level a node runs in.  Its earliest one is 1 + the deepest input's; with `spread`, a row whose
// root sits below the batch's deepest root runs as a whole some levels later, by a share of its
// slack that differs from row to row (a van der Corput sequence over the row number): the earliest
// levels alone make the first launches huge and the last dozen tiny, and every launch, however
// small, costs a tile's latency.  Inputs still run before their users, results do not change.
__device__ __forceinline__ int final_level(const vqa_devsched_params& p, int row, int root_level, int earliest) {
    if (!p.spread) return earliest;
    const int slack = p.error[1] - root_level;
    const float u = (float)__brev((unsigned)(row + 1)) * 2.3283064365386963e-10f;      // [0, 1)
    return earliest + min(slack, (int)(u * (float)(slack + 1)));
}

// One thread per (row, node): the (level, token) histogram of the nodes that are not the scene module.
__global__ __launch_bounds__(256) void devsched_hist_kernel(vqa_devsched_params p) {
    const int R = 2 * p.T + 1;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (int64_t)p.N * R) return;
    const int row = (int)(gid / R), k = (int)(gid - (int64_t)row * R);
    const int4 cnt = *((const int4*)p.counts + row);
    if (k >= cnt.x) return;
    const int w0 = ((const int4*)p.node + (int64_t)row * R + k)->x;
    if ((w0 & 255) == p.scene_id) return;
    atomicAdd(p.hist + final_level(p, row, cnt.w >> 16, w0 >> 8) * p.V + (w0 & 255), 1);
}

// exclusive scan of one value per thread over the workgroup (1024 threads)
__device__ __forceinline__ int block_exclusive_scan(int v, int* buf) {
    const int tid = threadIdx.x;
    buf[tid] = v;
    __syncthreads();
    int acc = v;
    for (int d = 1; d < 1024; d <<= 1) {
        const int other = tid >= d ? buf[tid - d] : 0;
        __syncthreads();
        acc += other;
        buf[tid] = acc;
        __syncthreads();
    }
    return acc - v;
}

// One workgroup: exclusive scans.  Rows -> first global index of the row's non-scene nodes and of
// its binary nodes; (level, token) bins in level-major order -> first tile of each bin, and per
// level {first tile, tiles} for the launches.  Also clears the bin cursors.
__global__ __launch_bounds__(1024) void devsched_scan_kernel(vqa_devsched_params p) {
    extern __shared__ int32_t lds[];      // [(T+1) * V] histogram, then [T+1] level sums
    __shared__ int buf[1024];
    const int tid = threadIdx.x;
    const int per = (p.N + 1023) / 1024;
    const int lo = min(p.N, tid * per), hi = min(p.N, lo + per);
    int sv = 0, sb = 0;
    for (int r = lo; r < hi; r++) { sv += p.counts[r * 4 + 1]; sb += p.counts[r * 4 + 2]; }
    int av = block_exclusive_scan(sv, buf);
    __syncthreads();
    int ab = block_exclusive_scan(sb, buf);
    for (int r = lo; r < hi; r++) {
        p.row_base[r * 2] = av; p.row_base[r * 2 + 1] = ab;
        av += p.counts[r * 4 + 1]; ab += p.counts[r * 4 + 2];
    }
    const int levels = p.T + 1, bins = levels * p.V;
    int32_t* lsum = lds + bins;
    for (int i = tid; i < bins; i += 1024) lds[i] = p.hist[i];
    __syncthreads();
    if (tid < levels) {
        int s = 0;
        for (int t = 0; t < p.V; t++) s += lds[tid * p.V + t];
        lsum[tid] = s;
    }
    __syncthreads();
    if (tid == 0) {
        int a = 0;
        for (int L = 0; L < levels; L++) { const int c = lsum[L]; lsum[L] = a; p.level_info[L * 2] = a; p.level_info[L * 2 + 1] = c; a += c; }
    }
    __syncthreads();
    if (tid < levels) {
        int a = lsum[tid];
        for (int t = 0; t < p.V; t++) { const int c = lds[tid * p.V + t]; lds[tid * p.V + t] = a; a += c; }
    }
    __syncthreads();
    for (int i = tid; i < bins; i += 1024) { p.bin_off[i] = lds[i]; p.bin_cursor[i] = 0; }
}

// One thread per (row, node): arena slots and the tile record.  Slots with fixed strides (NVmax =
// N * T): F(i) = i, S0(i) = N + i, scene output 2N + i, scene intermediate 3N + i, Y(v) = 4N + v,
// H(v) = 4N + NVmax + v, P(b) = 4N + 2 NVmax + b.  A node's tile goes into its (level, token) bin
// at a position taken with an atomic: the order inside a bin varies from run to run, the tiles of
// a launch are independent of each other, so results do not.
__global__ __launch_bounds__(256) void devsched_emit_kernel(vqa_devsched_params p) {
    const int R = 2 * p.T + 1;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (int64_t)p.N * R) return;
    const int row = (int)(gid / R), k = (int)(gid - (int64_t)row * R);
    const int4 cnt = *((const int4*)p.counts + row);
    if (k >= cnt.x) return;
    const int4* nodes = (const int4*)p.node + (int64_t)row * R;
    const int4 nd = nodes[k];
    // (slot bases count images: M of them when several rows share one, else one per row)
    const int N = p.image_of_row ? p.num_images : p.N, nvmax = p.N * p.T;
    int img = row;
    if (p.image_of_row) {
        img = p.image_of_row[row];
        if (img < 0 || img >= N) { atomicOr(p.error, 2); img = 0; }
    }
    const int tok = nd.x & 255, L = final_level(p, row, cnt.w >> 16, nd.x >> 8);
    const int vbase = p.row_base[row * 2], bbase = p.row_base[row * 2 + 1];
    const bool scene = tok == p.scene_id;
    const int y = scene ? 2 * N + img : 4 * N + vbase + (nd.w & 0xffff);
    if (k == (cnt.w & 0xffff)) {
        // the classifier's gather: root map of the row -> row of the classifier input (the torch.cat
        // of Module_Model.py:232)
        int4* ct = (int4*)p.cls_tiles + (int64_t)row * 2;
        ct[0] = make_int4(y, -1, row, -1);
        ct[1] = make_int4(-1, -1, -1, 0);
    }
    if (scene) return;
    const int4 k0 = nodes[nd.y];
    const int y0 = (k0.x & 255) == p.scene_id ? 2 * N + img : 4 * N + vbase + (k0.w & 0xffff);
    const bool bin = nd.z >= 0;
    int y1 = -1;
    if (bin) {
        const int4 k1 = nodes[nd.z];
        y1 = (k1.x & 255) == p.scene_id ? 2 * N + img : 4 * N + vbase + (k1.w & 0xffff);
    }
    const int in = bin ? 4 * N + 2 * nvmax + bbase + (nd.w >> 16) : y0;
    const int pos = p.bin_off[L * p.V + tok] + atomicAdd(p.bin_cursor + L * p.V + tok, 1);
    int4* t = (int4*)p.tiles + (int64_t)pos * 3;
    t[0] = make_int4(y0, y1, y, y + nvmax);
    t[1] = make_int4(in, -1, -1, tok);
    t[2] = make_int4(-1, -1, -1, -1);
}

extern "C" int vqa_devsched_build(const vqa_devsched_params* p, void* stream) {
    if (!p || !p->programs || !p->arity || !p->used || !p->node || !p->counts || !p->hist || !p->row_base ||
        !p->level_info || !p->bin_off || !p->bin_cursor || !p->tiles || !p->cls_tiles || !p->error || p->N <= 0 ||
        p->T <= 0 || p->V <= 0 || p->V > 255 || (p->image_of_row && p->num_images <= 0) || p->T + 1 > 1024 || (int64_t)p->N * p->T >= (1 << 26) ||
        (((uintptr_t)p->node | (uintptr_t)p->counts | (uintptr_t)p->tiles | (uintptr_t)p->cls_tiles) & 15)) {
        vqa_set_error("vqa_devsched_build: bad argument (sizes, null pointer, or node / counts / tiles / cls_tiles not 16-byte aligned)");
        return VQA_EINVAL;
    }
    // rows per workgroup of the parse: as many as LDS holds, 64 at most
    const size_t per_row = ((size_t)5 * p->T + 4 * (2 * (size_t)p->T + 1)) * sizeof(int32_t);
    int rows = 64;
    const size_t fixed = 256 * sizeof(int32_t);      // the arity table
    while (rows > 1 && rows * per_row + fixed > 144 * 1024) rows >>= 1;
    const size_t scan_lds = ((size_t)(p->T + 1) * p->V + p->T + 1) * sizeof(int32_t);
    if (rows * per_row + fixed > 144 * 1024 || scan_lds > 60 * 1024) {
        vqa_set_error("vqa_devsched_build: programs too long for the device scheduler");
        return VQA_EINVAL;
    }
    if (!vqa_allow_lds((const void*)devsched_parse_kernel, 144 * 1024, "vqa_devsched_build") ||
        !vqa_allow_lds((const void*)devsched_scan_kernel, 60 * 1024, "vqa_devsched_build")) return VQA_ELAUNCH;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(p->hist, 0, (size_t)(p->T + 1) * p->V * sizeof(int32_t), s);
    if (e == hipSuccess) e = hipMemsetAsync(p->error, 0, 2 * sizeof(int32_t), s);
    if (e != hipSuccess) { vqa_set_error(std::string("vqa_devsched_build: ") + hipGetErrorString(e)); return VQA_ELAUNCH; }
    hipLaunchKernelGGL(devsched_parse_kernel, dim3((p->N + rows - 1) / rows), dim3(rows), rows * per_row + fixed, s, *p);
    const int64_t threads = (int64_t)p->N * (2 * p->T + 1);
    hipLaunchKernelGGL(devsched_hist_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, *p);
    hipLaunchKernelGGL(devsched_scan_kernel, dim3(1), dim3(1024), scan_lds, s, *p);
    hipLaunchKernelGGL(devsched_emit_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, *p);
    e = hipGetLastError();
    if (e != hipSuccess) { vqa_set_error(std::string("vqa_devsched_build: ") + hipGetErrorString(e)); return VQA_ELAUNCH; }
    return VQA_OK;
}
