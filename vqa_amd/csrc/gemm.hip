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

template <bool A_KM, bool B_KM>
__global__ __launch_bounds__(256) void gemm_kernel(vqa_gemm_params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int l15 = lane & 15, lq = lane >> 4;
    const int n0 = blockIdx.x * 128, m0 = blockIdx.y * 128;
    const int ksteps = p.K / 64;
    const int per = (ksteps + p.split_k - 1) / p.split_k;
    const int kbeg = blockIdx.z * per, kend = min(ksteps, kbeg + per);

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

    if (kbeg < kend) issue(kbeg, 0);
    for (int k = kbeg; k < kend; k++) {
        const int buf = (k - kbeg) & 1;
        wait_vm0();
        __syncthreads();
        if (k + 1 < kend) issue(k + 1, buf ^ 1);
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

#pragma unroll
    for (int mi = 0; mi < 4; mi++) {
        const int m = m0 + wm * 64 + mi * 16 + l15;
        if (m >= p.M) continue;
#pragma unroll
        for (int ni = 0; ni < 4; ni++) {
            const int n = n0 + wn * 64 + ni * 16 + 4 * lq;
            if (n >= p.N) continue;
            const int64_t off = (int64_t)m * p.ldc + n;
            float v[4] = {acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]};
            if (p.out_mode == 2) {
                bf16_t* c = (bf16_t*)p.c + off;
                const bf16_t* mk = p.mask ? (const bf16_t*)p.mask + off : nullptr;
                if (n + 3 < p.N) {
                    if (mk) {
                        const u16x4 k4 = *(const u16x4*)mk;
#pragma unroll
                        for (int r = 0; r < 4; r++) v[r] = ((int16_t)k4[r] > 0) ? v[r] : 0.f;
                    }
                    u16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
                    *(u16x4*)c = o;
                } else {
                    for (int r = 0; r < 4 && n + r < p.N; r++) {
                        float x = v[r];
                        if (mk && !((int16_t)mk[r] > 0)) x = 0.f;
                        c[r] = f2bf(x);
                    }
                }
            } else {
                float* c = (float*)p.c + (int64_t)blockIdx.z * p.M * p.ldc + off;   // split-k slab
                if (n + 3 < p.N) *(float4*)c = make_float4(v[0], v[1], v[2], v[3]);
                else for (int r = 0; r < 4 && n + r < p.N; r++) c[r] = v[r];
            }
        }
    }
}

extern "C" int vqa_gemm(const vqa_gemm_params* p, void* stream) {
    if (!p || !p->a || !p->b || !p->c || p->M <= 0 || p->N <= 0 || p->K <= 0 || p->K % 64 != 0 ||
        p->split_k < 1 || (p->out_mode != 1 && p->out_mode != 2) || (p->out_mode == 2 && p->split_k != 1) ||
        p->lda % 8 != 0 || p->ldb % 8 != 0 || p->ldc % 4 != 0) {
        vqa_set_error("vqa_gemm: bad argument (K must be a multiple of 64, leading dimensions of 8)");
        return VQA_EINVAL;
    }
    static bool attr_set = false;
    const int lds = 4 * GT_BYTES;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void*)gemm_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void*)gemm_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void*)gemm_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    dim3 grid((p->N + 127) / 128, (p->M + 127) / 128, p->split_k), block(256);
    hipStream_t s = (hipStream_t)stream;
    VqaProfScope prof(VQA_K_GEMM, 2.0 * p->M * (double)p->N * p->K,
                      2.0 * ((double)p->M * p->K + (double)p->N * p->K) + (p->out_mode == 2 ? 2.0 : 4.0) * p->M * (double)p->N, s);
    if (!p->a_kmajor && !p->b_kmajor) hipLaunchKernelGGL((gemm_kernel<false, false>), grid, block, lds, s, *p);
    else if (!p->a_kmajor && p->b_kmajor) hipLaunchKernelGGL((gemm_kernel<false, true>), grid, block, lds, s, *p);
    else if (p->a_kmajor && p->b_kmajor) hipLaunchKernelGGL((gemm_kernel<true, true>), grid, block, lds, s, *p);
    else hipLaunchKernelGGL((gemm_kernel<true, false>), grid, block, lds, s, *p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { vqa_set_error(std::string("vqa_gemm: ") + hipGetErrorString(e)); return VQA_ELAUNCH; }
    return VQA_OK;
}
