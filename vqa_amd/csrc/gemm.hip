// Dense bf16 GEMM for the classifier's linear layers and their gradients (gfx950 MFMA).
//
// C[M][N] = A * B.  Each operand is either k-contiguous in memory ([rows][K]) or k-major
// ([K][rows]); k-major operands are fed to the MFMA through the transposing LDS read.  128 x 128
// output tile, K step 64, 4 waves in a 2 x 2 grid (64 x 64 per wave), both operands DMA-copied
// global->LDS, double buffered, one barrier per K step.  Bank conflicts are avoided with XOR
// swizzles applied to the per-lane DMA source address and undone in the fragment reads (the LDS
// image itself is linear, as the DMA requires).
//
// Replaces nn.Linear of Module_Model.py:85,92 and autograd's gradients of it.
#include "vqa_dev.h"
#include "plan.h"
#include "prof.h"

#define GT_BYTES 16384   // one operand stage: 128 x 64 bf16
#define GT_STAGES 2      // operand stages in LDS (with the epilogue tile 68 KB: two workgroups per CU; 4 stages at one per CU ran 1.7x slower)
#define GT_STAGE_LD 132  // row stride (floats) of the epilogue's fp32 staging tile: 132 % 32 = 4 keeps the 16-byte accumulator writes of 8 rows on 32 different banks (136 made them 2-way conflicts)

// k-contiguous tile [128 rows][64 k]: 16-B chunk j of row r lives at chunk position j ^ (r & 7)
__device__ __forceinline__ void stage_kc(const char* base, int64_t ld_bytes, char* lds, int lane, int wave) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int piece = wave * 4 + i;
        const int row = piece * 8 + (lane >> 3), cp = lane & 7;
        dma16(base + row * ld_bytes + ((cp ^ (row & 7)) << 4), lds + piece * 1024);
    }
}
// k-major tile [64 k][128 cols]: 16-B chunk j of row k lives at chunk position j ^ sw(k)
__device__ __forceinline__ int sw_km(int row) { return ((row & 3) | (((row >> 3) & 1) << 2)) << 1; }
__device__ __forceinline__ void stage_km(const char* base, int64_t ld_bytes, char* lds, int lane, int wave) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int piece = wave * 4 + i;
        const int row = piece * 4 + (lane >> 4), cp = lane & 15;
        dma16(base + row * ld_bytes + ((cp ^ sw_km(row)) << 4), lds + piece * 1024);
    }
}
// fragment of 16 rows (row0..row0+15) x 32 k (k-substep ks) from a k-contiguous tile
__device__ __forceinline__ bf16x8 frag_kc(const char* lds, int row0, int ks, int lane) {
    const int row = row0 + (lane & 15), kc = ks * 4 + (lane >> 4);
    return *(const bf16x8*)(lds + row * 128 + ((kc ^ (row & 7)) << 4));
}
// same fragment from a k-major tile (columns col0..col0+15)
__device__ __forceinline__ bf16x8 frag_km(const char* lds, int col0, int ks, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int j = (col0 >> 3) + (pp >> 1), sub = (pp & 1) * 8;
    const int r0 = ks * 32 + 8 * g + q, r1 = r0 + 4;
    const s16x4 lo = lds_tr4(lds + r0 * 256 + ((j ^ sw_km(r0)) << 4) + sub);
    const s16x4 hi = lds_tr4(lds + r1 * 256 + ((j ^ sw_km(r1)) << 4) + sub);
    return frag_from_tr(lo, hi);
}

// The same two fragment reads as inline assembly, for loops that keep LDS-DMA writes in flight while
// they read (see gemm256_kernel): `lds` is the tile's LDS byte address.
__device__ __forceinline__ uint32_t lds_off(const void* p) { return (uint32_t)(uintptr_t)(LDS_AS const char*)p; }
__device__ __forceinline__ bf16x8 frag_kc_asm(uint32_t lds, int row0, int ks, int lane) {
    const int row = row0 + (lane & 15), kc = ks * 4 + (lane >> 4);
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(lds + row * 128 + ((kc ^ (row & 7)) << 4)));
    return v;
}
__device__ __forceinline__ bf16x8 frag_km_asm(uint32_t lds, int col0, int ks, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int j = (col0 >> 3) + (pp >> 1), sub = (pp & 1) * 8;
    const int r0 = ks * 32 + 8 * g + q, r1 = r0 + 4;
    s16x4 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(lds + r0 * 256 + ((j ^ sw_km(r0)) << 4) + sub));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(lds + r1 * 256 + ((j ^ sw_km(r1)) << 4) + sub));
    return frag_from_tr(lo, hi);
}
// s_waitcnt lgkmcnt(N), tied to the registers it is for: nothing that uses them can be scheduled ahead of it
template <int N>
__device__ __forceinline__ void wait_lgkm_for1(bf16x8& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N)); }
template <int N>
__device__ __forceinline__ void wait_lgkm_for5(bf16x8& a, bf16x8& b, bf16x8& c, bf16x8& d, bf16x8& e) {
    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : "n"(N));
}

template <bool A_KM, bool B_KM, int STAGES>
__global__ __launch_bounds__(256) void gemm_kernel(vqa_gemm_params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int l15 = lane & 15, lq = lane >> 4;
    // Workgroup -> tile.  Consecutive workgroup ids go round-robin over the 8 XCDs, each with its own
    // L2.  Tiles are taken in groups of up to 8 x 8 (one k-split): a group is the 64 workgroups one
    // XCD holds at a time (32 CUs x 2), so the row and column panels the group shares come from HBM
    // once and from that XCD's L2 seven times, instead of once per tile from HBM.
    const int tm = (p.M + 127) >> 7, tn = (p.N + 127) >> 7;
    const int gm = min(tm, 8), gn = min(tn, 8);
    const int mgroups = (tm + gm - 1) / gm, ngroups = (tn + gn - 1) / gn;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int g = (j >> 6) * 8 + xcd, w = j & 63;
    if (g >= mgroups * ngroups * p.split_k) return;
    const int kz = g / (mgroups * ngroups), gr = g - kz * (mgroups * ngroups);
    const int mt = (gr % mgroups) * gm + w % gm, nt = (gr / mgroups) * gn + w / gm;
    if (w >= gm * gn || mt >= tm || nt >= tn) return;
    const int n0 = nt * 128, m0 = mt * 128;
    const int ksteps = p.K / 64;
    const int per = (ksteps + p.split_k - 1) / p.split_k;
    const int kbeg = kz * per, kend = min(ksteps, kbeg + per);

    const char* abase = (const char*)p.a;
    const char* bbase = (const char*)p.b;
    const int64_t lda = p.lda * 2, ldb = p.ldb * 2;
    auto issue = [&](int kstep, int buf) {
        char* la = smem + buf * 2 * GT_BYTES;
        char* lb = la + GT_BYTES;
        const int64_t k0 = (int64_t)kstep * 64;
        if (A_KM) stage_km(abase + k0 * lda + (int64_t)m0 * 2, lda, la, lane, wave);
        else stage_kc(abase + (int64_t)m0 * lda + k0 * 2, lda, la, lane, wave);
        if (B_KM) stage_km(bbase + k0 * ldb + (int64_t)n0 * 2, ldb, lb, lane, wave);
        else stage_kc(bbase + (int64_t)n0 * ldb + k0 * 2, ldb, lb, lane, wave);
    };

    f32x4 acc[4][4];   // [ni][mi]: lane holds C[m = mi*16 + l15][n = ni*16 + 4*lq + r]
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // STAGES - 1 K-steps are in flight ahead of the one being multiplied
#pragma unroll
    for (int s_ = 0; s_ < STAGES - 1; s_++)
        if (kbeg + s_ < kend) issue(kbeg + s_, s_);
    for (int k = kbeg; k < kend; k++) {
        const int buf = (k - kbeg) % STAGES;
        // stage k has landed once at most the STAGES - 2 younger ones (8 DMA instructions each) are pending
        if (kend - 1 - k >= STAGES - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (STAGES - 2)) : "memory");
        else wait_vm0();
        __syncthreads();
        if (k + STAGES - 1 < kend) issue(k + STAGES - 1, (buf + STAGES - 1) % STAGES);
        const char* la = smem + buf * 2 * GT_BYTES;
        const char* lb = la + GT_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                af[i] = A_KM ? frag_km(la, wm * 64 + i * 16, ks, lane) : frag_kc(la, wm * 64 + i * 16, ks, lane);
                bfr[i] = B_KM ? frag_km(lb, wn * 64 + i * 16, ks, lane) : frag_kc(lb, wn * 64 + i * 16, ks, lane);
            }
#pragma unroll
            for (int ni = 0; ni < 4; ni++)
#pragma unroll
                for (int mi = 0; mi < 4; mi++) acc[ni][mi] = mma16(bfr[ni], af[mi], acc[ni][mi]);
        }
    }

    // Epilogue through LDS: the accumulator layout (a lane holds 4 consecutive columns of 16 different
    // rows) would store 32- or 64-byte pieces per row; staged as an fp32 [128][136] tile, every wave
    // stores whole 512-byte (fp32) or 256-byte (bf16) row segments, 16 bytes per lane.
    __syncthreads();                   // every wave is done with the operand stages
    float* stage = (float*)smem;
#pragma unroll
    for (int mi = 0; mi < 4; mi++)
#pragma unroll
        for (int ni = 0; ni < 4; ni++)
            *(f32x4*)(stage + (wm * 64 + mi * 16 + l15) * GT_STAGE_LD + wn * 64 + ni * 16 + 4 * lq) = acc[ni][mi];
    __syncthreads();
    if (p.out_mode == 2) {
        const int cg = (lane & 15) * 8;            // 8 columns per lane, 16 lanes per row
        const int n = n0 + cg;
#pragma unroll 4
        for (int it = 0; it < 8; it++) {
            const int r = it * 16 + wave * 4 + (lane >> 4);
            const int m = m0 + r;
            if (m >= p.M || n >= p.N) continue;
            const float* sp = stage + r * GT_STAGE_LD + cg;
            const f32x4 lo = *(const f32x4*)sp, hi = *(const f32x4*)(sp + 4);
            float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            const int64_t off = (int64_t)m * p.ldc + n;
            bf16_t* c = (bf16_t*)p.c + off;
            const bf16_t* mk = p.mask ? (const bf16_t*)p.mask + off : nullptr;
            if (n + 7 < p.N) {
                if (mk) {
                    const u16x8 k8 = *(const u16x8*)mk;
#pragma unroll
                    for (int q = 0; q < 8; q++) v[q] = ((int16_t)k8[q] > 0) ? v[q] : 0.f;
                }
                uint4 o;
                o.x = cvt_pk_bf16((f32x2){v[0], v[1]}); o.y = cvt_pk_bf16((f32x2){v[2], v[3]});
                o.z = cvt_pk_bf16((f32x2){v[4], v[5]}); o.w = cvt_pk_bf16((f32x2){v[6], v[7]});
                *(uint4*)c = o;
            } else {
                for (int q = 0; q < 8 && n + q < p.N; q++) {
                    float x = v[q];
                    if (mk && !((int16_t)mk[q] > 0)) x = 0.f;
                    c[q] = f2bf(x);
                }
            }
        }
    } else {
        const int cg = (lane & 31) * 4;            // 4 columns per lane, 32 lanes per row
        const int n = n0 + cg;
        float* cbase = (float*)p.c + (int64_t)kz * p.M * p.ldc;   // split-k slab
#pragma unroll 4
        for (int it = 0; it < 16; it++) {
            const int r = it * 8 + wave * 2 + (lane >> 5);
            const int m = m0 + r;
            if (m >= p.M || n >= p.N) continue;
            const f32x4 v = *(const f32x4*)(stage + r * GT_STAGE_LD + cg);
            float* c = cbase + (int64_t)m * p.ldc + n;
            if (n + 3 < p.N) *(float4*)c = make_float4(v[0], v[1], v[2], v[3]);
            else for (int q = 0; q < 4 && n + q < p.N; q++) c[q] = v[q];
        }
    }
}

// The same GEMM on 256 x 256 output tiles: 8 waves in a 2 x 4 grid (128 x 64 per wave), one
// workgroup per CU.  A 128 x 128 tile moves 1 operand byte from L2 to LDS per 64 FLOP, and the
// device sustains about 10 TB/s there (measured: 4 GB through L2 in 370 us), which caps that kernel
// near 650 TFLOP/s whatever else is done; a 256 x 256 tile needs half the bytes.  Each operand tile
// is two of the 128-row images above, staged by one half of the waves each.  M and N must be
// multiples of 256 (the operand loads are not guarded).
#define GB_STAGE_LD 260     // row stride (floats) of the epilogue's fp32 [128][256] staging tile (260 % 32 = 4: conflict-free 16-byte writes)
template <bool A_KM, bool B_KM>
__global__ __launch_bounds__(512) void gemm256_kernel(vqa_gemm_params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int l15 = lane & 15, lq = lane >> 4;
    // groups of up to 32 tiles (one k-split) per XCD at a time, as in gemm_kernel
    const int tm = p.M >> 8, tn = p.N >> 8;
    const int gm = min(tm, 4), gn = min(tn, 32 / gm);
    const int mgroups = (tm + gm - 1) / gm, ngroups = (tn + gn - 1) / gn;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int g = (j >> 5) * 8 + xcd, w = j & 31;
    if (g >= mgroups * ngroups * p.split_k) return;
    const int kz = g / (mgroups * ngroups), gr = g - kz * (mgroups * ngroups);
    const int mt = (gr % mgroups) * gm + w % gm, nt = (gr / mgroups) * gn + w / gm;
    if (w >= gm * gn || mt >= tm || nt >= tn) return;
    const int n0 = nt * 256, m0 = mt * 256;
    const int ksteps = p.K / 64;
    const int per = (ksteps + p.split_k - 1) / p.split_k;
    const int kbeg = kz * per, kend = min(ksteps, kbeg + per);

    const char* abase = (const char*)p.a;
    const char* bbase = (const char*)p.b;
    const int64_t lda = p.lda * 2, ldb = p.ldb * 2;
    const int half = wave >> 2, w4 = wave & 3;     // which 128-row image this wave stages
    auto issue = [&](int kstep, int buf) {
        char* la = smem + buf * 4 * GT_BYTES + half * GT_BYTES;
        char* lb = la + 2 * GT_BYTES;
        const int64_t k0 = (int64_t)kstep * 64;
        const int ma = m0 + half * 128, nb = n0 + half * 128;
        if (A_KM) stage_km(abase + k0 * lda + (int64_t)ma * 2, lda, la, lane, w4);
        else stage_kc(abase + (int64_t)ma * lda + k0 * 2, lda, la, lane, w4);
        if (B_KM) stage_km(bbase + k0 * ldb + (int64_t)nb * 2, ldb, lb, lane, w4);
        else stage_kc(bbase + (int64_t)nb * ldb + k0 * 2, ldb, lb, lane, w4);
    };

    f32x4 acc[4][8];   // [ni][mi]: lane holds C[m = wm*128 + mi*16 + l15][n = wn*64 + ni*16 + 4*lq + r]
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 8; b++) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (kbeg < kend) issue(kbeg, 0);
    for (int k = kbeg; k < kend; k++) {
        const int buf = (k - kbeg) & 1;
        wait_vm0();
        __syncthreads();
        if (k + 1 < kend) issue(k + 1, buf ^ 1);
        const char* la = smem + buf * 4 * GT_BYTES + wm * GT_BYTES;                 // this wave's 128 rows of A
        const char* lb = smem + buf * 4 * GT_BYTES + (2 + (wn >> 1)) * GT_BYTES;    // the image holding its 64 columns
        const int cb = (wn & 1) * 64;
        // The 8 waves read 96 KB of fragments per 32-deep half of the stage, three quarters of the
        // time its MFMAs take: read first and multiplied afterwards, the LDS and the MFMA pipe take
        // turns.  So the second half's fragments are read under the first half's MFMAs: row block mi
        // of A is multiplied with all four column blocks, then its registers take the second half's
        // fragment; only B's second half needs registers of its own.
        //
        // The fragment reads are inline assembly with their own two waits per K step.  Left to the
        // compiler, which cannot tell a read of this stage from the LDS-DMA writes of the next one
        // (issued a few lines up), every first read got an s_waitcnt vmcnt(0) in front: the next stage
        // was waited for before this one was multiplied, and a K step lasted a trip to HBM plus its
        // MFMA time.  A wait is tied to the registers it is for, so nothing that uses them can be
        // scheduled ahead of it; the registers go from the reads straight into the MFMAs.
        const uint32_t la_o = lds_off(la), lb_o = lds_off(lb);
        // LDS instructions per fragment, and the waits counted in them (reads return in order; the counter
        // holds 15 at most): B's first half goes out first, then A's eight row blocks; in front of row
        // block mi the reads issued behind its fragment may still be pending - the later row blocks and
        // the second-half fragments read so far.
        constexpr int UA = A_KM ? 2 : 1, UB = B_KM ? 2 : 1;
        bf16x8 af[8], bfr[4], bfr1[4];
#pragma unroll
        for (int i = 0; i < 4; i++) bfr[i] = B_KM ? frag_km_asm(lb_o, cb + i * 16, 0, lane) : frag_kc_asm(lb_o, cb + i * 16, 0, lane);
#pragma unroll
        for (int i = 0; i < 8; i++) af[i] = A_KM ? frag_km_asm(la_o, i * 16, 0, lane) : frag_kc_asm(la_o, i * 16, 0, lane);
        __builtin_amdgcn_sched_barrier(0);
#define GEMM_ROWBLOCK(mi)                                                                                      \
        {                                                                                                      \
            constexpr int pend = 7 * UA + ((mi) < 4 ? (mi) : 4) * UB;                                          \
            if ((mi) == 0) wait_lgkm_for5<(pend < 15 ? pend : 15)>(bfr[0], bfr[1], bfr[2], bfr[3], af[0]);     \
            else wait_lgkm_for1<(pend < 15 ? pend : 15)>(af[mi]);                                              \
            _Pragma("unroll") for (int ni = 0; ni < 4; ni++) acc[ni][mi] = mma16(bfr[ni], af[mi], acc[ni][mi]); \
            af[mi] = A_KM ? frag_km_asm(la_o, (mi) * 16, 1, lane) : frag_kc_asm(la_o, (mi) * 16, 1, lane);     \
            if ((mi) < 4) bfr1[(mi) & 3] = B_KM ? frag_km_asm(lb_o, cb + ((mi) & 3) * 16, 1, lane) : frag_kc_asm(lb_o, cb + ((mi) & 3) * 16, 1, lane); \
            __builtin_amdgcn_sched_barrier(0);                                                                 \
        }
        GEMM_ROWBLOCK(0) GEMM_ROWBLOCK(1) GEMM_ROWBLOCK(2) GEMM_ROWBLOCK(3)
        GEMM_ROWBLOCK(4) GEMM_ROWBLOCK(5) GEMM_ROWBLOCK(6) GEMM_ROWBLOCK(7)
#undef GEMM_ROWBLOCK
        // second half: its fragments went out in the order A'0 B'0 A'1 B'1 A'2 B'2 A'3 B'3 A'4 .. A'7
#define GEMM_ROWBLOCK2(mi)                                                                                     \
        {                                                                                                      \
            constexpr int pend = ((mi) < 4 ? 4 : 7 - (mi)) * UA;                                               \
            if ((mi) == 0) wait_lgkm_for5<pend>(bfr1[0], bfr1[1], bfr1[2], bfr1[3], af[0]);                    \
            else wait_lgkm_for1<pend>(af[mi]);                                                                 \
            _Pragma("unroll") for (int ni = 0; ni < 4; ni++) acc[ni][mi] = mma16(bfr1[ni], af[mi], acc[ni][mi]); \
            __builtin_amdgcn_sched_barrier(0);                                                                 \
        }
        GEMM_ROWBLOCK2(0) GEMM_ROWBLOCK2(1) GEMM_ROWBLOCK2(2) GEMM_ROWBLOCK2(3)
        GEMM_ROWBLOCK2(4) GEMM_ROWBLOCK2(5) GEMM_ROWBLOCK2(6) GEMM_ROWBLOCK2(7)
#undef GEMM_ROWBLOCK2
    }

    // epilogue: two passes of 128 rows through an fp32 [128][264] LDS tile, whole-row stores
    float* stage = (float*)smem;
    float* cbase = (float*)p.c + (int64_t)kz * p.M * p.ldc;   // split-k slab (fp32 results)
#pragma unroll 1
    for (int pass = 0; pass < 2; pass++) {
        // the mask rows this thread needs in this pass: all eight loads are in flight across the two
        // barriers and the staging writes (one at a time in front of each store, their latency was
        // four HBM round trips per pass)
        u16x8 mk8[8];
        if (p.out_mode == 2 && p.mask) {
#pragma unroll
            for (int it = 0; it < 8; it++) {
                const int r = it * 16 + wave * 2 + (lane >> 5);
                mk8[it] = __builtin_nontemporal_load((const u16x8*)((const bf16_t*)p.mask +
                                                     (int64_t)(m0 + pass * 128 + r) * p.ldc + n0 + (lane & 31) * 8));
            }
        }
        __syncthreads();                   // operand stages / the previous pass are done with
        if (wm == pass) {
#pragma unroll
            for (int mi = 0; mi < 8; mi++)
#pragma unroll
                for (int ni = 0; ni < 4; ni++)
                    *(f32x4*)(stage + (mi * 16 + l15) * GB_STAGE_LD + wn * 64 + ni * 16 + 4 * lq) = acc[ni][mi];
        }
        __syncthreads();
        const int mbase = m0 + pass * 128;
        if (p.out_mode == 2) {
            const int cg = (lane & 31) * 8;            // 8 columns per lane, 32 lanes per row
#pragma unroll
            for (int it = 0; it < 8; it++) {
                const int r = it * 16 + wave * 2 + (lane >> 5);
                const float* sp = stage + r * GB_STAGE_LD + cg;
                const f32x4 lo = *(const f32x4*)sp, hi = *(const f32x4*)(sp + 4);
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                const int64_t off = (int64_t)(mbase + r) * p.ldc + n0 + cg;
                if (p.mask) {
                    const u16x8 k8 = mk8[it];
#pragma unroll
                    for (int q = 0; q < 8; q++) v[q] = ((int16_t)k8[q] > 0) ? v[q] : 0.f;
                }
                uint4 o;
                o.x = cvt_pk_bf16((f32x2){v[0], v[1]}); o.y = cvt_pk_bf16((f32x2){v[2], v[3]});
                o.z = cvt_pk_bf16((f32x2){v[4], v[5]}); o.w = cvt_pk_bf16((f32x2){v[6], v[7]});
                *(uint4*)((bf16_t*)p.c + off) = o;
            }
        } else {
            const int cg = lane * 4;                   // 4 columns per lane, a wave per row
#pragma unroll 4
            for (int it = 0; it < 16; it++) {
                const int r = it * 8 + wave;
                const f32x4 v = *(const f32x4*)(stage + r * GB_STAGE_LD + cg);
                float* cp = cbase + (int64_t)(mbase + r) * p.ldc + n0 + cg;
                // a result that is not split over k is final and far larger than the caches (the first
                // linear layer's 411 MB weight gradient): stream it, whole 1 KiB rows per instruction
                if (p.split_k == 1) __builtin_nontemporal_store(v, (f32x4*)cp);
                else *(float4*)cp = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

extern "C" int vqa_gemm(const vqa_gemm_params* p, void* stream) {
    if (!p || !p->a || !p->b || !p->c || p->M <= 0 || p->N <= 0 || p->K <= 0 || p->K % 64 != 0 ||
        p->split_k < 1 || (p->out_mode != 1 && p->out_mode != 2) || (p->out_mode == 2 && p->split_k != 1) ||
        p->lda % 8 != 0 || p->ldb % 8 != 0 || p->ldc % (p->out_mode == 2 ? 8 : 4) != 0) {
        vqa_set_error("vqa_gemm: bad argument (K must be a multiple of 64, leading dimensions of 8; 4 for an fp32 result)");
        return VQA_EINVAL;
    }
    constexpr int ST = GT_STAGES;
    const int lds = ST * 2 * GT_BYTES > 128 * GT_STAGE_LD * 4 ? ST * 2 * GT_BYTES : 128 * GT_STAGE_LD * 4;
    if (!vqa_allow_lds((const void*)gemm_kernel<false, false, ST>, lds, "vqa_gemm") ||
        !vqa_allow_lds((const void*)gemm_kernel<false, true, ST>, lds, "vqa_gemm") ||
        !vqa_allow_lds((const void*)gemm_kernel<true, true, ST>, lds, "vqa_gemm") ||
        !vqa_allow_lds((const void*)gemm_kernel<true, false, ST>, lds, "vqa_gemm")) return VQA_ELAUNCH;
    hipStream_t s = (hipStream_t)stream;
    if (p->M % 256 == 0 && p->N % 256 == 0 && (int64_t)(p->M / 256) * (p->N / 256) * p->split_k >= 128) {
        // large problems: 256 x 256 tiles
        const int lds256 = 128 * GB_STAGE_LD * 4 > 8 * GT_BYTES ? 128 * GB_STAGE_LD * 4 : 8 * GT_BYTES;
        if (!vqa_allow_lds((const void*)gemm256_kernel<false, false>, lds256, "vqa_gemm") ||
            !vqa_allow_lds((const void*)gemm256_kernel<false, true>, lds256, "vqa_gemm") ||
            !vqa_allow_lds((const void*)gemm256_kernel<true, true>, lds256, "vqa_gemm") ||
            !vqa_allow_lds((const void*)gemm256_kernel<true, false>, lds256, "vqa_gemm")) return VQA_ELAUNCH;
        const int tm = p->M / 256, tn = p->N / 256;
        const int gm = tm < 4 ? tm : 4, gn = tn < 32 / gm ? tn : 32 / gm;
        const int64_t groups = (int64_t)((tm + gm - 1) / gm) * ((tn + gn - 1) / gn) * p->split_k;
        const int64_t blocks = (groups + 7) / 8 * 8 * 32;
        if (blocks > 0x7fffffff) { vqa_set_error("vqa_gemm: problem too large"); return VQA_EINVAL; }
        dim3 grid((unsigned)blocks), block(512);
        VqaProfScope prof(VQA_K_GEMM, 2.0 * p->M * (double)p->N * p->K,
                          2.0 * ((double)p->M * p->K + (double)p->N * p->K) + (p->out_mode == 2 ? 2.0 : 4.0) * p->M * (double)p->N, s);
        if (!p->a_kmajor && !p->b_kmajor) hipLaunchKernelGGL((gemm256_kernel<false, false>), grid, block, lds256, s, *p);
        else if (!p->a_kmajor && p->b_kmajor) hipLaunchKernelGGL((gemm256_kernel<false, true>), grid, block, lds256, s, *p);
        else if (p->a_kmajor && p->b_kmajor) hipLaunchKernelGGL((gemm256_kernel<true, true>), grid, block, lds256, s, *p);
        else hipLaunchKernelGGL((gemm256_kernel<true, false>), grid, block, lds256, s, *p);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { vqa_set_error(std::string("vqa_gemm: ") + hipGetErrorString(e)); return VQA_ELAUNCH; }
        return VQA_OK;
    }
    const int tm = (p->M + 127) / 128, tn = (p->N + 127) / 128;
    const int gm = tm < 8 ? tm : 8, gn = tn < 8 ? tn : 8;
    const int64_t groups = (int64_t)((tm + gm - 1) / gm) * ((tn + gn - 1) / gn) * p->split_k;
    const int64_t blocks = (groups + 7) / 8 * 8 * 64;
    if (blocks > 0x7fffffff) { vqa_set_error("vqa_gemm: problem too large"); return VQA_EINVAL; }
    dim3 grid((unsigned)blocks), block(256);
    VqaProfScope prof(VQA_K_GEMM, 2.0 * p->M * (double)p->N * p->K,
                      2.0 * ((double)p->M * p->K + (double)p->N * p->K) + (p->out_mode == 2 ? 2.0 : 4.0) * p->M * (double)p->N, s);
    if (!p->a_kmajor && !p->b_kmajor) hipLaunchKernelGGL((gemm_kernel<false, false, ST>), grid, block, lds, s, *p);
    else if (!p->a_kmajor && p->b_kmajor) hipLaunchKernelGGL((gemm_kernel<false, true, ST>), grid, block, lds, s, *p);
    else if (p->a_kmajor && p->b_kmajor) hipLaunchKernelGGL((gemm_kernel<true, true, ST>), grid, block, lds, s, *p);
    else hipLaunchKernelGGL((gemm_kernel<true, false, ST>), grid, block, lds, s, *p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { vqa_set_error(std::string("vqa_gemm: ") + hipGetErrorString(e)); return VQA_ELAUNCH; }
    return VQA_OK;
}
