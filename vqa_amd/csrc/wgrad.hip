// Grouped convolution weight gradient (gfx950, bf16 MFMA; per-workgroup partial sums + a reduction
// pass, or fp32 atomic accumulation).
//
//   dW[group][tap][i][o] += sum_{node in group} sum_{pixel} src_node(pixel + tap)[i] * dy_node(pixel)[o]
//
// The contraction runs over pixels, which is the slow axis of the channels-last maps, so both MFMA
// operands are read from LDS with the transposing read (ds_read_b64_tr_b16).  One workgroup
// (4 waves) owns a work item (<= wgrad_chunk nodes of one weight group) x 64 output channels x
// 64 source channels; wave w owns source channels [16w, 16w+16) of that half x 4 tiles of 16 output
// channels x every tap, so its accumulators (taps x 4 tiles) stay in registers across all nodes of
// the item and leave once.  The quarter-of-a-node shape keeps LDS at 76 KB, so two independent
// workgroups share a CU and one stages its next node while the other is in its MFMA phase (with
// one 8-wave workgroup per CU every wave staged at the same time and the matrix pipe idled).  Leaving as fp32 atomics keeps a CU busy for ~60 us per workgroup (one 256-byte
// atomic instruction per ~50 ns per CU) with nothing else resident to hide it; leaving as plain
// 16-byte stores into a per-workgroup partial block costs a few us, and vqa_wgrad_reduce then sums
// the blocks of each group straight into torch's weight layout (which also makes the result
// independent of the order workgroups finish in).  A k-step of 32 is two image rows of 16 columns; columns 14, 15
// of the dy image are zero, the source image has a zero border.
//
// Replaces autograd's per-call conv weight/bias gradients and their sum over every use of a
// token's module (Train_model.py:313 over Module_Model.py:221).
#include "vqa_dev.h"
#include "plan.h"
#include "prof.h"
#include <mutex>
#include <type_traits>
#include <string>

#define XH_STRIDE 160                        // 64 channels + 32 bytes
#define XH_POS 264
#define XH_BYTES (XH_POS * XH_STRIDE)        // 42240
#define DY_STRIDE 160
#define DY_POS 224
#define DY_BYTES (DY_POS * DY_STRIDE)        // 35840
#define WGRAD_LDS (XH_BYTES + DY_BYTES)

template <int TAPS>
__global__ __launch_bounds__(256, 2) void wgrad_tiles_kernel(vqa_wgrad_params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xh = smem;
    char* dyt = smem + XH_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4, q = l15 >> 2, pp = lane & 3;
    // Workgroup -> (item, output slice, source half).  The workgroups of one item read the same maps
    // (every source half once per output slice and the other way round): dispatched far apart they
    // each fetched them from HBM (measured: 4.6 GB per step against 2.2 GB algorithmic).  Consecutive
    // workgroup ids go round-robin over the 8 XCDs, so XCD x takes items x, x + 8, ... and runs the
    // siblings of an item back to back: the second to last reader of a map finds it in that XCD's L2.
    const int oslices = p.n_out >> 6, zs = p.k_chunks * 2, sibs = oslices * zs;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int item = (j / sibs) * 8 + xcd, sib = j % sibs;
    if (item >= p.num_items) return;             // grid is padded to whole groups of 8 items
    const vqa_wgrad_item it = p.items[item];
    const int oslice = sib % oslices, kchunk = (sib / oslices) >> 1, khalf = (sib / oslices) & 1;

    for (int i = tid; i < WGRAD_LDS / 16; i += 256) ((uint4*)smem)[i] = make_uint4(0, 0, 0, 0);

    int which = 0, choff = kchunk * 128;
    if (p.pair_chunks > 0) { which = kchunk / p.pair_chunks; choff = (kchunk % p.pair_chunks) * 128; }
    choff += khalf * 64;
    const int32_t* sslots = which ? p.src1_slots : p.src_slots;

    // Staging registers are named scalars (not arrays): loop-carried arrays end up in scratch.  Two
    // sets (a, b); the second one is used by the single-tap kernel only.
    uint4 x0a, x1a, x2a, x3a, x4a, x5a, x6a, d0a, d1a, d2a, d3a, d4a, d5a, d6a;
    uint4 x0b, x1b, x2b, x3b, x4b, x5b, x6b, d0b, d1b, d2b, d3b, d4b, d5b, d6b;
#define XLOAD(i, var)                                                                          \
    {                                                                                          \
        const int e = tid + 256 * i, pix = min(e >> 3, VQA_HW - 1), pc = e & 7;                \
        var = *(const uint4*)(sp + (int64_t)pix * p.src_pix_stride * 2 + pc * 16);             \
    }
#define DLOAD(i, var)                                                                          \
    {                                                                                          \
        const int e = tid + 256 * i, pix = min(e >> 3, VQA_HW - 1), pc = e & 7;                \
        var = *(const uint4*)(dp + (int64_t)pix * p.dy_pix_stride * 2 + pc * 16);              \
    }
#define LOAD_NODE(n, S)                                                                        \
    {                                                                                          \
        const int ss = sslots[it.first + (n)], ds = p.dy_slots[it.first + (n)];                \
        const char* sp = (const char*)p.src + ((int64_t)ss * p.src_slot_stride + choff) * 2;   \
        const char* dp = (const char*)p.dy + ((int64_t)ds * p.dy_slot_stride + oslice * 64) * 2; \
        XLOAD(0, x0##S) XLOAD(1, x1##S) XLOAD(2, x2##S) XLOAD(3, x3##S) XLOAD(4, x4##S) XLOAD(5, x5##S) XLOAD(6, x6##S) \
        DLOAD(0, d0##S) DLOAD(1, d1##S) DLOAD(2, d2##S) DLOAD(3, d3##S) DLOAD(4, d4##S) DLOAD(5, d5##S) DLOAD(6, d6##S) \
    }
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define XSTORE(i, var)                                                                         \
    {                                                                                          \
        const int e = tid + 256 * i, pix = e >> 3, pc = e & 7;                                 \
        if (pix < VQA_HW) {                                                                    \
            const int y = pix / 14, x = pix - y * 14;                                          \
            *(uint4*)(xh + ((y + 1) * 16 + x + 1) * XH_STRIDE + pc * 16) = var;                \
        }                                                                                      \
    }
#define DSTORE(i, var)                                                                         \
    {                                                                                          \
        const int e = tid + 256 * i, pix = e >> 3, pc = e & 7;                                 \
        if (pix < VQA_HW) {                                                                    \
            const int y = pix / 14, x = pix - y * 14;                                          \
            *(uint4*)(dyt + (y * 16 + x) * DY_STRIDE + pc * 16) = var;                         \
            bsum[0] += __uint_as_float(var.x << 16); bsum[1] += __uint_as_float(var.x & 0xFFFF0000u); \
            bsum[2] += __uint_as_float(var.y << 16); bsum[3] += __uint_as_float(var.y & 0xFFFF0000u); \
            bsum[4] += __uint_as_float(var.z << 16); bsum[5] += __uint_as_float(var.z & 0xFFFF0000u); \
            bsum[6] += __uint_as_float(var.w << 16); bsum[7] += __uint_as_float(var.w & 0xFFFF0000u); \
        }                                                                                      \
    }
#define STORE_NODE(S)                                                                          \
    {                                                                                          \
        XSTORE(0, x0##S) XSTORE(1, x1##S) XSTORE(2, x2##S) XSTORE(3, x3##S) XSTORE(4, x4##S) XSTORE(5, x5##S) XSTORE(6, x6##S) \
        DSTORE(0, d0##S) DSTORE(1, d1##S) DSTORE(2, d2##S) DSTORE(3, d3##S) DSTORE(4, d4##S) DSTORE(5, d5##S) DSTORE(6, d6##S) \
    }

    f32x4 acc[TAPS][4];
#pragma unroll
    for (int t = 0; t < TAPS; t++)
#pragma unroll
        for (int n = 0; n < 4; n++) acc[t][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // lane's transposed-read addresses at k-step 0, read r = 0: position (row g>>1, col 4*(g&1)+q)
    const int prow = g >> 1, pcol = 4 * (g & 1) + q;
    const int xoff = (prow * 16 + pcol) * XH_STRIDE + (wave * 16 + 4 * pp) * 2;   // + tap, + kk, + r
    const int doff = (prow * 16 + pcol) * DY_STRIDE + (4 * pp) * 2;               // + 32*ot, + kk, + r

    // the MFMA phase over the staged node: 7 k-steps of two image rows
    // (the k-step's two base addresses are made opaque: every read is then that register plus a constant that fits
    // the instruction's offset field; left to itself the compiler picks one base for the whole loop and spends a
    // vector add per read on offsets that do not fit)
    auto load_first = [&](bf16x8 (&bf)[4], bf16x8 (&ar)[2], int kk) {
        int xo = xoff + kk * 32 * XH_STRIDE, dq = doff + kk * 32 * DY_STRIDE;
        asm volatile("" : "+v"(xo), "+v"(dq));
        const char* xb = xh + xo;
        const char* db = dyt + dq;
#pragma unroll
        for (int o = 0; o < 4; o++)
            bf[o] = frag_from_tr(lds_tr4(db + o * 32), lds_tr4(db + o * 32 + 8 * DY_STRIDE));
#pragma unroll
        for (int t = 0; t < (TAPS > 1 ? 2 : 1); t++) {
            const int tapoff = (TAPS == 9) ? ((t / 3) * 16 + (t % 3)) : 17;
            ar[t] = frag_from_tr(lds_tr4(xb + tapoff * XH_STRIDE), lds_tr4(xb + (tapoff + 8) * XH_STRIDE));
        }
    };
    // One k-step: the taps run through a 3-deep ring of source fragments fetched two taps ahead, the issue order
    // is pinned (2 LDS reads : 4 MFMAs; otherwise every tap waits for its own reads), and - NEXT - the first
    // fragments of the following k-step (its four gradient fragments and first two source fragments: 12 reads, into
    // registers the staging no longer needs) are asked for under this one's last taps: a k-step that begins by
    // asking for them waits out an LDS round trip with the matrix pipe idle, seven times per node.
    auto kstep = [&](const bf16x8 (&bf)[4], const bf16x8 (&a01)[2], int kk, auto next_tag, bf16x8 (&bfn)[4], bf16x8 (&an)[2]) {
        constexpr bool NEXT = decltype(next_tag)::value;
        constexpr int PF = TAPS > 4 ? TAPS - 4 : 0;            // the tap in front of which the next k-step's reads go
        int xo = xoff + kk * 32 * XH_STRIDE;
        asm volatile("" : "+v"(xo));
        const char* xb = xh + xo;
        auto afrag = [&](int t) {
            const int tapoff = (TAPS == 9) ? ((t / 3) * 16 + (t % 3)) : 17;
            return frag_from_tr(lds_tr4(xb + tapoff * XH_STRIDE), lds_tr4(xb + (tapoff + 8) * XH_STRIDE));
        };
        bf16x8 ar[3];
        ar[0] = a01[0];
        if (TAPS > 1) ar[1] = a01[1];
#pragma unroll
        for (int t = 0; t < TAPS; t++) {
            if (NEXT && t == PF) load_first(bfn, an, kk + 1);
            if (t + 2 < TAPS) ar[(t + 2) % 3] = afrag(t + 2);
#pragma unroll
            for (int o = 0; o < 4; o++) acc[t][o] = mma16(ar[t % 3], bf[o], acc[t][o]);
        }
#pragma unroll
        for (int t = 0; t < TAPS; t++) {
            if (NEXT && t == PF) __builtin_amdgcn_sched_group_barrier(0x100, TAPS > 1 ? 12 : 10, 0);
            if (t + 2 < TAPS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
    };
    auto mfma_phase = [&]() {
        bf16x8 bfa[4], bfb[4], aa[2], ab[2];
        load_first(bfa, aa, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, TAPS > 1 ? 12 : 10, 0);
#pragma unroll 1
        for (int kk = 0; kk < 6; kk += 2) {
            kstep(bfa, aa, kk, std::true_type{}, bfb, ab);
            kstep(bfb, ab, kk + 1, std::true_type{}, bfa, aa);
        }
        kstep(bfa, aa, 6, std::false_type{}, bfb, ab);
    };

    if constexpr (TAPS == 1) {
        // One tap: a node is 28 MFMAs per wave against ~75 KB of staging, so the loop is all load
        // latency; with only 16 accumulator registers there is room to keep the next TWO nodes'
        // loads in flight while one is staged and multiplied.
        LOAD_NODE(0, a)
        if (it.count > 1) LOAD_NODE(1, b)
        for (int n = 0; n < it.count; n += 2) {
            __syncthreads();
            STORE_NODE(a)
            if (n + 2 < it.count) LOAD_NODE(n + 2, a)
            __syncthreads();
            mfma_phase();
            if (n + 1 < it.count) {
                __syncthreads();
                STORE_NODE(b)
                if (n + 3 < it.count) LOAD_NODE(n + 3, b)
                __syncthreads();
                mfma_phase();
            }
        }
    } else {
        for (int n = 0; n < it.count; n++) {
            // no register prefetch across the MFMA phase: the partner workgroup on this CU covers the
            // round trip, and 14 more live registers would spill
            LOAD_NODE(n, a)
            __syncthreads();          // zero fill done / previous node's reads done
            STORE_NODE(a)
            __syncthreads();
            mfma_phase();
        }
    }

    // accumulators out: row = source channel, col = output channel
    if (p.partial) {
        // block of this workgroup: [tap][64 output channels][128 source channels] fp32
        float* part = p.partial + ((int64_t)(item * oslices + oslice) * p.k_chunks + kchunk) * (TAPS * 8192) + khalf * 64;
#pragma unroll
        for (int t = 0; t < TAPS; t++)
#pragma unroll
            for (int o = 0; o < 4; o++)
                *(f32x4*)(part + (t * 64 + o * 16 + l15) * 128 + wave * 16 + g * 4) = acc[t][o];
    } else {
        const int k_in = p.k_chunks * 128;
        float* dw = p.dw + (int64_t)it.group * p.dw_group_stride;
#pragma unroll
        for (int t = 0; t < TAPS; t++)
#pragma unroll
            for (int o = 0; o < 4; o++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int i = kchunk * 128 + khalf * 64 + wave * 16 + g * 4 + r;
                    const int oc = oslice * 64 + o * 16 + l15;
                    atomicAdd(dw + ((int64_t)t * k_in + i) * p.n_out + oc, acc[t][o][r]);
                }
    }

    const bool bias_rows = p.partial && p.bias_block > 0;
    if ((p.db || bias_rows) && kchunk == 0 && khalf == 0) {
        __syncthreads();
        float* red = (float*)smem;       // [256][8]
#pragma unroll
        for (int j = 0; j < 8; j++) red[tid * 8 + j] = bsum[j];
        __syncthreads();
        if (tid < 64) {
            const int pc = tid >> 3, j = tid & 7;
            float s = 0.f;
            for (int u = pc; u < 256; u += 8) s += red[u * 8 + j];
            if (bias_rows) p.partial[(int64_t)p.bias_block * 8192 + (int64_t)item * p.n_out + oslice * 64 + tid] = s;
            else atomicAdd(p.db + (int64_t)it.group * p.n_out + oslice * 64 + tid, s);
        }
    }
}

extern "C" int vqa_wgrad_tiles(const vqa_wgrad_params* p, int32_t taps, void* stream) {
    if (!p || !p->src || !p->dy || !p->items || !p->src_slots || !p->dy_slots || (!p->dw && !p->partial) ||
        (taps != 9 && taps != 1) || p->k_chunks <= 0 || p->n_out <= 0 || p->n_out % 64 != 0 ||
        p->src_pix_stride % 8 != 0 || p->dy_pix_stride % 8 != 0 || p->num_items < 0 ||
        (p->pair_chunks > 0 && !p->src1_slots)) {
        vqa_set_error("vqa_wgrad_tiles: bad argument");
        return VQA_EINVAL;
    }
    if (p->num_items == 0) return VQA_OK;
    if (!vqa_allow_lds((const void*)wgrad_tiles_kernel<9>, WGRAD_LDS, "vqa_wgrad_tiles") ||
        !vqa_allow_lds((const void*)wgrad_tiles_kernel<1>, WGRAD_LDS, "vqa_wgrad_tiles")) return VQA_ELAUNCH;
    const int64_t blocks = (int64_t)((p->num_items + 7) / 8) * 8 * (p->n_out / 64) * (p->k_chunks * 2);
    if (blocks > 0x7fffffff) { vqa_set_error("vqa_wgrad_tiles: too many work items"); return VQA_EINVAL; }
    dim3 grid((unsigned)blocks), block(256);
    VqaProfScope prof(taps == 9 ? VQA_K_WGRAD9 : VQA_K_WGRAD1,
                      2.0 * VQA_HW * taps * p->k_chunks * 128.0 * p->n_out * p->total_nodes,
                      (double)p->total_nodes * VQA_HW * 2.0 * (p->k_chunks * 128 + p->n_out), (hipStream_t)stream);
    if (taps == 9) hipLaunchKernelGGL(wgrad_tiles_kernel<9>, grid, block, WGRAD_LDS, (hipStream_t)stream, *p);
    else hipLaunchKernelGGL(wgrad_tiles_kernel<1>, grid, block, WGRAD_LDS, (hipStream_t)stream, *p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { vqa_set_error(std::string("vqa_wgrad_tiles: ") + hipGetErrorString(e)); return VQA_ELAUNCH; }
    return VQA_OK;
}

// Sum the partial blocks of vqa_wgrad_tiles per weight group, writing torch's [n][k][tap] layout.
// A thread owns one output channel x 4 consecutive source channels x all taps: 16-byte loads from
// every block, TAPS x 16 contiguous bytes of output.  The pass is all load latency (each partial is
// read once, from HBM), so the blocks of the group are first listed in LDS - in item order, which
// fixes the order of the fp32 sums whatever the launch looks like - and then read U at a time.
// With SEG > 1 (few groups: the stem and classifier convolutions have one, and a grid of
// outputs / 1024 workgroups would leave the device idle behind long chains of dependent loads) the
// workgroup covers 256 / SEG columns and its SEG thread rows each take every SEG-th batch of U
// blocks; their sums are added through LDS in row order.
template <int TAPS, int U, int SEG>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial,
                                                           const vqa_wgrad_item* __restrict__ items, int num_items,
                                                           int n_out, int k_chunks, float* __restrict__ out,
                                                           int64_t out_g_stride, const float* __restrict__ bias_partial,
                                                           float* __restrict__ db) {
    typedef float f32x4r __attribute__((ext_vector_type(4)));
    constexpr int COLS = 256 / SEG;
    __shared__ int list[256];
    __shared__ int wcount[4];
    __shared__ f32x4r segsum[SEG > 1 ? (SEG - 1) * COLS * TAPS : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = tid % COLS, seg = tid / COLS;
    const int grp = blockIdx.y, k_in = k_chunks * 128;
    const int e = (blockIdx.x * COLS + col) * 4;             // oc * k_in + i, i a multiple of 4
    const int oc = e / k_in, i = e - oc * k_in;
    const int oslice = oc >> 6, ocl = oc & 63, kchunk = i >> 7, il = i & 127;
    const int oslices = n_out >> 6;
    const int64_t in_block = (int64_t)(oslice * k_chunks + kchunk) * (TAPS * 8192) + ocl * 128 + il;
    const int64_t item_stride = (int64_t)oslices * k_chunks * (TAPS * 8192);
    f32x4r acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; t++) acc[t] = (f32x4r){0.f, 0.f, 0.f, 0.f};
    // the group's bias sums: one more workgroup per group, next to the ones that sum weight blocks
    // (inside one of those it would be the launch's tail); thread = output channel (two of them when
    // n_out > 256), the items' rows added in list order
    const bool do_bias = db != nullptr && blockIdx.x == gridDim.x - 1;
    float bacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};        // n_out <= 2048
    for (int base = 0; base < num_items; base += 256) {
        const int j = base + tid;
        const bool mine = j < num_items && items[j].group == grp;
        const unsigned long long b = __ballot(mine);
        if (lane == 0) wcount[wave] = __popcll(b);
        __syncthreads();
        int off = 0, n = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) { if (w < wave) off += wcount[w]; n += wcount[w]; }
        if (mine) list[off + __popcll(b & ((1ull << lane) - 1))] = j;
        __syncthreads();
        for (int k = seg * U; k < n && !do_bias; k += SEG * U) {
            f32x4r v[U][TAPS];
#pragma unroll
            for (int u = 0; u < U; u++)
                if (k + u < n) {
                    const float* part = partial + list[k + u] * item_stride + in_block;
#pragma unroll
                    for (int t = 0; t < TAPS; t++) v[u][t] = __builtin_nontemporal_load((const f32x4r*)(part + t * 8192));
                }
#pragma unroll
            for (int u = 0; u < U; u++)
                if (k + u < n) {
#pragma unroll
                    for (int t = 0; t < TAPS; t++) acc[t] += v[u][t];
                }
        }
        if (do_bias) {
#pragma unroll
            for (int h = 0; h < 8; h++) {
                const int o = tid + 256 * h;
                if (o < n_out)
                    for (int k = 0; k < n; k += 8) {
                        float v[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) v[u] = k + u < n ? bias_partial[(int64_t)list[k + u] * n_out + o] : 0.f;
#pragma unroll
                        for (int u = 0; u < 8; u++) bacc[h] += v[u];
                    }
            }
        }
        __syncthreads();       // the list is rewritten by the next window
    }
    if (do_bias) {
#pragma unroll
        for (int h = 0; h < 8; h++)
            if (tid + 256 * h < n_out) db[(int64_t)grp * n_out + tid + 256 * h] = bacc[h];
        return;
    }
    if (SEG > 1) {
        if (seg > 0) {
#pragma unroll
            for (int t = 0; t < TAPS; t++) segsum[((seg - 1) * TAPS + t) * COLS + col] = acc[t];
        }
        __syncthreads();
        if (seg > 0) return;
#pragma unroll
        for (int s_ = 0; s_ < SEG - 1; s_++)
#pragma unroll
            for (int t = 0; t < TAPS; t++) acc[t] += segsum[(s_ * TAPS + t) * COLS + col];
    }
    // (source channel q, tap t) of this thread's 4 x TAPS contiguous outputs is acc[t][q]
    float* o = out + (int64_t)grp * out_g_stride + (int64_t)e * TAPS;
#pragma unroll
    for (int f = 0; f < TAPS; f++) {
        f32x4r w;
#pragma unroll
        for (int r = 0; r < 4; r++) w[r] = acc[(4 * f + r) % TAPS][(4 * f + r) / TAPS];
        *(f32x4r*)(o + 4 * f) = w;
    }
}

extern "C" int vqa_wgrad_reduce(const float* partial, const vqa_wgrad_item* items, int32_t num_items,
                                int32_t groups, int32_t n_out, int32_t k_chunks, int32_t taps, float* out,
                                int64_t out_g_stride, const float* bias_partial, float* db, void* stream) {
    if (!partial || !items || !out || num_items < 0 || groups <= 0 || n_out <= 0 || n_out % 64 != 0 ||
        k_chunks <= 0 || (taps != 9 && taps != 1) || ((db != nullptr) != (bias_partial != nullptr)) ||
        (db && n_out > 2048)) {
        vqa_set_error("vqa_wgrad_reduce: bad argument");
        return VQA_EINVAL;
    }
    const int cols4 = n_out * k_chunks * 128 / 4;                  // 16-byte columns per group; n_out % 64 == 0
    dim3 block(256);
    hipStream_t s = (hipStream_t)stream;
    // (single-tap blocks are small: there the plain form is as fast, measured)
    const int extra = db ? 1 : 0;                                  // the bias workgroup of each group
    if (taps == 1 || (int64_t)(cols4 / 256) * groups >= 512) {
        dim3 grid(cols4 / 256 + extra, groups);
        if (taps == 9) hipLaunchKernelGGL((wgrad_reduce_kernel<9, 3, 1>), grid, block, 0, s, partial, items, num_items, n_out, k_chunks, out, out_g_stride, bias_partial, db);
        else hipLaunchKernelGGL((wgrad_reduce_kernel<1, 8, 1>), grid, block, 0, s, partial, items, num_items, n_out, k_chunks, out, out_g_stride, bias_partial, db);
    } else {
        dim3 grid(cols4 / 32 + extra, groups);
        hipLaunchKernelGGL((wgrad_reduce_kernel<9, 2, 8>), grid, block, 0, s, partial, items, num_items, n_out, k_chunks, out, out_g_stride, bias_partial, db);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { vqa_set_error(std::string("vqa_wgrad_reduce: ") + hipGetErrorString(e)); return VQA_ELAUNCH; }
    return VQA_OK;
}

// ---------------------------------------------------------------------------------------------
// Weight gradient for callers that cannot allocate (HIP graph capture, tight serving loops): the
// partial blocks of vqa_wgrad_tiles live in a file-scope device array instead of a caller
// workspace, and the reduction into torch's layout follows in the same call.  Every GPU the
// library runs on holds its own copy of the array until the process ends; its address on a device
// is looked up once, with that device current (the array's name is only a host-side handle).
// ---------------------------------------------------------------------------------------------
#define WG_SCRATCH_BLOCKS 1024                                    // blocks of 8192 floats: 32 MiB
static __device__ __align__(256) float scratch[(size_t)WG_SCRATCH_BLOCKS * 8192];
#define WG_MAX_DEVICES 64
static float* g_scratch_ptr[WG_MAX_DEVICES];                      // per device, null until looked up
static std::mutex g_scratch_mu;

static int wgrad_scratch_on_current_device(float** out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= WG_MAX_DEVICES) {
        vqa_set_error("vqa_wgrad_tiles_noalloc: no current device (or a device index beyond 63)");
        return VQA_ELAUNCH;
    }
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* ptr = nullptr;
        hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) {
            vqa_set_error(std::string("vqa_wgrad_tiles_noalloc: hipGetSymbolAddress: ") + hipGetErrorString(e));
            return VQA_ELAUNCH;
        }
        g_scratch_ptr[dev] = (float*)ptr;
    }
    *out = g_scratch_ptr[dev];
    return VQA_OK;
}

extern "C" int vqa_wgrad_tiles_noalloc(const vqa_wgrad_params* p, int32_t taps, int32_t groups, float* out,
                                       int64_t out_g_stride, void* stream) {
    if (!p || !out || groups <= 0 || (taps != 9 && taps != 1) || p->num_items < 0 || p->n_out <= 0 ||
        p->n_out % 64 != 0 || p->k_chunks <= 0) {
        vqa_set_error("vqa_wgrad_tiles_noalloc: bad argument");
        return VQA_EINVAL;
    }
    // one block of taps * 8192 floats per (item, 64 output channels, 128-channel source chunk)
    const int64_t need = (int64_t)p->num_items * (p->n_out / 64) * p->k_chunks * taps;
    if (need > WG_SCRATCH_BLOCKS) {
        vqa_set_error("vqa_wgrad_tiles_noalloc: " + std::to_string(need) + " partial blocks of 32 KiB needed (items x n_out/64 x "
                      "k_chunks x taps), the static scratch holds " + std::to_string(WG_SCRATCH_BLOCKS) +
                      ": split the item list over several calls or use vqa_wgrad_tiles with a workspace");
        return VQA_EINVAL;
    }
    float* sc = nullptr;
    int rc = wgrad_scratch_on_current_device(&sc);
    if (rc) return rc;
    vqa_wgrad_params q = *p;
    q.partial = sc;          // every block the reduction reads is written whole by this launch: nothing to clear
    q.dw = nullptr;
    q.bias_block = 0;        // db by atomics: the scratch holds weight-gradient blocks only
    if ((rc = vqa_wgrad_tiles(&q, taps, stream))) return rc;
    return vqa_wgrad_reduce(sc, q.items, q.num_items, groups, q.n_out, q.k_chunks, taps, out, out_g_stride, nullptr, nullptr, stream);
}

// dW accumulators [group][tap][k][n] fp32 -> torch's [n][k][tap] per group
__global__ void unpack_conv_grad_kernel(const float* __restrict__ dw, float* __restrict__ out,
                                        int n_out, int k_in, int taps, int64_t out_g_stride) {
    const int g = blockIdx.y;
    const int64_t per = (int64_t)n_out * k_in * taps;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < per;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int t = idx % taps;
        const int64_t r = idx / taps;
        const int k = r % k_in, n = r / k_in;
        out[g * out_g_stride + idx] = dw[g * per + ((int64_t)t * k_in + k) * n_out + n];
    }
}

extern "C" int vqa_unpack_conv_grad(const float* dw, float* out, int32_t groups, int32_t n_out,
                                    int32_t k_in, int32_t taps, int64_t out_g_stride, void* stream) {
    if (!dw || !out || groups <= 0 || n_out <= 0 || k_in <= 0 || taps <= 0) {
        vqa_set_error("vqa_unpack_conv_grad: bad argument");
        return VQA_EINVAL;
    }
    const int64_t per = (int64_t)n_out * k_in * taps;
    int blocks = (int)std::min<int64_t>((per + 255) / 256, 2048);
    hipLaunchKernelGGL(unpack_conv_grad_kernel, dim3(blocks, groups), dim3(256), 0, (hipStream_t)stream,
                       dw, out, n_out, k_in, taps, out_g_stride);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { vqa_set_error(std::string("vqa_unpack_conv_grad: ") + hipGetErrorString(e)); return VQA_ELAUNCH; }
    return VQA_OK;
}
