// Program scheduler on the device (gfx950): the parse of forward_modules_ints_helper
// (Module_Model.py:196-222) and the (level, token) bucketing of the whole batch as three small
// kernels, so that programs a sampler left in device memory (LSTM_Model.py:205-235 builds them on
// the GPU and copies them to the host every step, :228-235) become tile lists without a round
// trip through the host.  Forward only (candidate scoring and evaluation: Train_model.py:339-354,
// Functions.py:152-179); levels are the earliest ones (1 + the deepest input), which is what the
// host scheduler produces without VQA_PLAN_BALANCE, so results are bit-identical to it.
//
// The host never learns the node counts: the arena uses fixed strides (a row has at most T nodes
// that are not the scene module), the launch of level L has N * (T / L) workgroups - a row's upper
// bound - which take the level's tile range from device memory and exit when they are beyond it
// (vqa_resblock_tiles with `level_info`), and a token id
// outside the vocabulary - a KeyError in the reference and in the host scheduler - raises a flag in
// device memory instead (the row then runs as a lone scene module).
#include <string>

#include "vqa_dev.h"
#include "plan.h"

// One thread per program row.  Iterative form of the recursive helper: a stack of open nodes that
// wait for inputs (in `work`), finished nodes appended in execution (post-)order.
//   node  [N][2T+1][4] : token, input 0, input 1 (indices into the row's node list, -1 none), level
//   counts[N][4]       : nodes, nodes that are not the scene module, binary nodes, root's index
//   hist  [T+1][V]     : nodes that are not the scene module per (level, token), whole batch (atomic)
//   work  [N][T][6]    : scratch (pending nodes and the stack)
__global__ __launch_bounds__(256) void devsched_parse_kernel(vqa_devsched_params p) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= p.N) return;
    const int T = p.T, R = 2 * T + 1;
    const int64_t* prog = p.programs + (int64_t)row * T;
    float* used = p.used + (int64_t)row * T;
    int32_t* node = p.node + (int64_t)row * R * 4;
    int32_t* pend = p.work + (int64_t)row * T * 6;          // [T][4] pending nodes, then [T][2] stack frames
    int32_t* stack = pend + T * 4;
    for (int j = 0; j < T; j++) used[j] = 0.f;
    int j = 0, n = 0, np = 0, sp = 0, root = 0;
    bool bad = false;
    for (;;) {
        int tok;
        for (;;) {
            if (j < T) {
                const int64_t v = prog[j];
                if (v < 0 || v >= p.V) { bad = true; break; }
                tok = (int)v;
                if (tok == p.start_id) { j++; continue; }          // skipped, wherever it stands
                if (tok != p.null_id) used[j] = 1.f;               // <NULL> runs as a module but is not marked
                j++;
            } else {
                tok = p.scene_id;                                  // reading past the row: an implicit scene
            }
            break;
        }
        if (bad) break;
        const int ar = tok == p.scene_id ? 0 : p.arity[tok];
        if (ar == 0) {
            int id = n++;
            node[id * 4 + 0] = tok; node[id * 4 + 1] = -1; node[id * 4 + 2] = -1; node[id * 4 + 3] = 0;
            bool done = false;
            for (;;) {
                if (sp == 0) { root = id; done = true; break; }
                int32_t* f = stack + (sp - 1) * 2;
                int32_t* par = pend + f[0] * 4;
                if (par[1] < 0) par[1] = id; else par[2] = id;
                par[3] = max(par[3], node[id * 4 + 3] + 1);
                if (--f[1] > 0) break;
                sp--;
                id = n++;
                node[id * 4 + 0] = par[0]; node[id * 4 + 1] = par[1]; node[id * 4 + 2] = par[2]; node[id * 4 + 3] = par[3];
            }
            if (done) break;
        } else {
            int32_t* q = pend + np * 4;
            q[0] = tok; q[1] = -1; q[2] = -1; q[3] = 0;
            stack[sp * 2] = np++; stack[sp * 2 + 1] = ar;
            sp++;
        }
    }
    if (bad) {
        atomicOr(p.error, 1);
        for (int c = 0; c < T; c++) used[c] = 0.f;
        n = 1; root = 0;
        node[0] = p.scene_id; node[1] = -1; node[2] = -1; node[3] = 0;
    }
    int nv = 0, nb = 0;
    for (int k = 0; k < n; k++) {
        const int tok = node[k * 4];
        if (tok == p.scene_id) continue;
        nv++;
        nb += node[k * 4 + 2] >= 0;
        atomicAdd(p.hist + node[k * 4 + 3] * p.V + tok, 1);
    }
    int32_t* c = p.counts + row * 4;
    c[0] = n; c[1] = nv; c[2] = nb; c[3] = root;
}

// One workgroup: exclusive scans.  Rows -> first global index of the row's non-scene nodes and of
// its binary nodes; (level, token) bins in level-major order -> first tile of each bin, and per
// level {first tile, tiles} for the launches.  Also clears the bin cursors.
__global__ __launch_bounds__(1024) void devsched_scan_kernel(vqa_devsched_params p) {
    __shared__ int part[1024];
    __shared__ int part_b[1024];
    const int tid = threadIdx.x;
    // rows: each thread owns a contiguous run
    const int per = (p.N + 1023) / 1024;
    int sv = 0, sb = 0;
    for (int r = tid * per; r < min(p.N, (tid + 1) * per); r++) { sv += p.counts[r * 4 + 1]; sb += p.counts[r * 4 + 2]; }
    part[tid] = sv; part_b[tid] = sb;
    __syncthreads();
    if (tid == 0) {
        int av = 0, ab = 0;
        for (int i = 0; i < 1024; i++) { const int v = part[i], b = part_b[i]; part[i] = av; part_b[i] = ab; av += v; ab += b; }
    }
    __syncthreads();
    int av = part[tid], ab = part_b[tid];
    for (int r = tid * per; r < min(p.N, (tid + 1) * per); r++) {
        p.row_base[r * 2] = av; p.row_base[r * 2 + 1] = ab;
        av += p.counts[r * 4 + 1]; ab += p.counts[r * 4 + 2];
    }
    // bins: level-major; thread L sums its level, thread 0 scans the levels, thread L scans its bins
    const int levels = p.T + 1;
    __syncthreads();
    if (tid < levels) {
        int s = 0;
        for (int t = 0; t < p.V; t++) s += p.hist[tid * p.V + t];
        part[tid] = s;
    }
    __syncthreads();
    if (tid == 0) {
        int a = 0;
        for (int L = 0; L < levels; L++) { const int c = part[L]; p.level_info[L * 2] = a; p.level_info[L * 2 + 1] = c; a += c; }
    }
    __syncthreads();
    if (tid < levels) {
        int a = p.level_info[tid * 2];
        for (int t = 0; t < p.V; t++) { const int c = p.hist[tid * p.V + t]; p.bin_off[tid * p.V + t] = a; p.bin_cursor[tid * p.V + t] = 0; a += c; }
    }
}

// One thread per row again: arena slots and the tile records.  Slots with fixed strides (NVmax =
// N * T): F(i) = i, S0(i) = M + i, Ysc(i) = 2M + i, Hsc(i) = 3M + i, Y(v) = 4M + v,
// H(v) = 4M + NVmax + v, P(b) = 4M + 2 NVmax + b.  A node's tile goes into its (level, token) bin
// at a position taken with an atomic: the order inside a bin varies from run to run, the tiles of
// a launch are independent of each other, so results do not.
__global__ __launch_bounds__(256) void devsched_emit_kernel(vqa_devsched_params p) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= p.N) return;
    const int T = p.T, R = 2 * T + 1, M = p.N;
    const int nvmax = p.N * T;
    const int32_t* node = p.node + (int64_t)row * R * 4;
    int32_t* yslot = p.work + (int64_t)row * T * 6;         // the parse scratch is free now; 2T + 1 <= 6T
    const int n = p.counts[row * 4];
    int v = p.row_base[row * 2], b = p.row_base[row * 2 + 1];
    for (int k = 0; k < n; k++) {
        const int tok = node[k * 4], c0 = node[k * 4 + 1], c1 = node[k * 4 + 2], L = node[k * 4 + 3];
        if (tok == p.scene_id) { yslot[k] = 2 * M + row; continue; }
        const int y = 4 * M + v, h = 4 * M + nvmax + v;
        yslot[k] = y;
        const bool bin = c1 >= 0;
        const int in = bin ? 4 * M + 2 * nvmax + b : yslot[c0];
        const int pos = p.bin_off[L * p.V + tok] + atomicAdd(p.bin_cursor + L * p.V + tok, 1);
        int32_t* t = p.tiles + (int64_t)pos * 12;
        t[0] = bin ? yslot[c0] : in; t[1] = bin ? yslot[c1] : -1; t[2] = y; t[3] = h; t[4] = in;
        t[5] = -1; t[6] = -1; t[7] = tok; t[8] = -1; t[9] = -1; t[10] = -1; t[11] = -1;
        v++; b += bin;
    }
    // the classifier's gather: root map of the row -> row of the classifier input (the torch.cat of
    // Module_Model.py:232)
    int32_t* ct = p.cls_tiles + (int64_t)row * 8;
    ct[0] = yslot[p.counts[row * 4 + 3]]; ct[1] = -1; ct[2] = row; ct[3] = -1; ct[4] = -1; ct[5] = -1; ct[6] = -1; ct[7] = 0;
}

extern "C" int vqa_devsched_build(const vqa_devsched_params* p, void* stream) {
    if (!p || !p->programs || !p->arity || !p->used || !p->node || !p->counts || !p->hist || !p->work || !p->row_base ||
        !p->level_info || !p->bin_off || !p->bin_cursor || !p->tiles || !p->cls_tiles || !p->error || p->N <= 0 ||
        p->T <= 0 || p->V <= 0 || p->T + 1 > 1024) {
        vqa_set_error("vqa_devsched_build: bad argument");
        return VQA_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(p->hist, 0, (size_t)(p->T + 1) * p->V * sizeof(int32_t), s);
    if (e == hipSuccess) e = hipMemsetAsync(p->error, 0, sizeof(int32_t), s);
    if (e != hipSuccess) { vqa_set_error(std::string("vqa_devsched_build: ") + hipGetErrorString(e)); return VQA_ELAUNCH; }
    const int blocks = (p->N + 255) / 256;
    hipLaunchKernelGGL(devsched_parse_kernel, dim3(blocks), dim3(256), 0, s, *p);
    hipLaunchKernelGGL(devsched_scan_kernel, dim3(1), dim3(1024), 0, s, *p);
    hipLaunchKernelGGL(devsched_emit_kernel, dim3(blocks), dim3(256), 0, s, *p);
    e = hipGetLastError();
    if (e != hipSuccess) { vqa_set_error(std::string("vqa_devsched_build: ") + hipGetErrorString(e)); return VQA_ELAUNCH; }
    return VQA_OK;
}
