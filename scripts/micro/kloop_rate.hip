// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 kloop_rate.hip -o kloop_rate
// What MFMA issue rate does one wave per SIMD (or two) reach with the K-loop shapes of the convolution
// kernels?  Cycles per MFMA (ideal: 16 for v_mfma_f32_16x16x32_bf16) for
//   V0  MFMAs only, operands in registers
//   V1  + one ds_read_b128 per two MFMAs through an 8-deep ring (wave tile: 2 weight x 14 pixel fragments)
//   V2  V1 + the two weight-fragment loads per K-step from memory (4-slot ring)
//   V3  wave tile 4 weight x 7 pixel fragments: one ds_read_b128 per four MFMAs, four loads per K-step
//   V4  V3 without the loads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define STRIDE 288
#define LDSB (264 * STRIDE)

__device__ __forceinline__ f32x4 mma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

template <int V>
__global__ __launch_bounds__(256, 2) void k(const char* w, float* out, long* cyc, int ksteps) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
    for (int i = tid; i < LDSB / 16; i += 256) ((uint4*)lds)[i] = make_uint4(0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80);
    __syncthreads();
    const char* wb = w + (size_t)(blockIdx.x % 40) * 294912 + wave * 2048 + lane * 16;
    long t0 = 0, t1 = 0;
    if constexpr (V <= 2) {
        f32x4 acc[14][2];
        for (int m = 0; m < 14; m++) acc[m][0] = acc[m][1] = (f32x4){0, 0, 0, 0};
        bf16x8 wr[4][2];
        for (int i = 0; i < 4; i++) { wr[i][0] = *(const bf16x8*)(wb + i * 8192); wr[i][1] = *(const bf16x8*)(wb + i * 8192 + 1024); }
        int xaddr[14];
        for (int m = 0; m < 14; m++) xaddr[m] = (m * 16 + l15) * STRIDE + lq * 16;
        bf16x8 xq[8];
        for (int q = 0; q < 7; q++) xq[q] = *(const bf16x8*)(lds + xaddr[q]);
        t0 = clock64();
        int s = 0;
#pragma unroll 1
        for (int it = 0; it < ksteps / 4; it++) {
            const char* xb = lds + (it % 9) * STRIDE;
#pragma unroll
            for (int ks = 0; ks < 4; ks++, s++) {
#pragma unroll
                for (int m = 0; m < 14; m++) {
                    const int q = ks * 14 + m, qa = q + 7, ka = (qa / 14) & 3, ma = qa % 14;
                    if (V >= 1) xq[qa & 7] = *(const bf16x8*)(xb + xaddr[ma] + ka * 64);
                    acc[m][0] = mma(wr[ks][0], xq[q & 7], acc[m][0]);
                    acc[m][1] = mma(wr[ks][1], xq[q & 7], acc[m][1]);
                }
                if (V >= 1) {
#pragma unroll
                    for (int m = 0; m < 14; m++) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    }
                }
                if (V >= 2) {
                    const int sn = min(s + 4, 35);
                    wr[ks][0] = *(const bf16x8*)(wb + sn * 8192);
                    wr[ks][1] = *(const bf16x8*)(wb + sn * 8192 + 1024);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        t1 = clock64();
        float sum = 0;
        for (int m = 0; m < 14; m++) sum += acc[m][0][0] + acc[m][1][1];
        if (sum == 12345.f) out[tid] = sum;
    } else {
        // wave = (64 output channels: 4 weight fragments) x (7 pixel tiles)
        const int ph = wave & 1;
        f32x4 acc[7][4];
        for (int m = 0; m < 7; m++) for (int f = 0; f < 4; f++) acc[m][f] = (f32x4){0, 0, 0, 0};
        bf16x8 wr[4][4];
        const char* wb2 = w + (size_t)(blockIdx.x % 40) * 294912 + (wave >> 1) * 4096 + lane * 16;
        for (int i = 0; i < 4; i++) for (int f = 0; f < 4; f++) wr[i][f] = *(const bf16x8*)(wb2 + i * 8192 + f * 1024);
        int xaddr[7];
        for (int m = 0; m < 7; m++) xaddr[m] = ((7 * ph + m) * 16 + l15) * STRIDE + lq * 16;
        bf16x8 xq[4];
        for (int q = 0; q < 3; q++) xq[q] = *(const bf16x8*)(lds + xaddr[q]);
        t0 = clock64();
        int s = 0;
#pragma unroll 1
        for (int it = 0; it < ksteps / 4; it++) {
            const char* xb = lds + (it % 9) * STRIDE;
#pragma unroll
            for (int ks = 0; ks < 4; ks++, s++) {
#pragma unroll
                for (int m = 0; m < 7; m++) {
                    const int q = ks * 7 + m, qa = q + 3, ka = (qa / 7) & 3, ma = qa % 7;
                    xq[qa & 3] = *(const bf16x8*)(xb + xaddr[ma] + ka * 64);
#pragma unroll
                    for (int f = 0; f < 4; f++) acc[m][f] = mma(wr[ks][f], xq[q & 3], acc[m][f]);
                }
#pragma unroll
                for (int m = 0; m < 7; m++) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                }
                if (V == 3) {
                    const int sn = min(s + 4, 35);
#pragma unroll
                    for (int f = 0; f < 4; f++) wr[ks][f] = *(const bf16x8*)(wb2 + sn * 8192 + f * 1024);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        t1 = clock64();
        float sum = 0;
        for (int m = 0; m < 7; m++) sum += acc[m][0][0] + acc[m][1][2] + acc[m][2][3] + acc[m][3][1];
        if (sum == 12345.f) out[tid] = sum;
    }
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V>
void run(const char* name, int nb, const char* w, float* out, long* cyc, long* hcyc) {
    const int ksteps = 72;     // two 3x3 convolutions of 128 channels
    hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 10; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<V>, dim3(nb), dim3(256), LDSB, 0, w, out, cyc, ksteps);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 3 && ms < best) best = ms;
    }
    hipMemcpy(hcyc, cyc, nb * sizeof(long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < nb; i++) mean += hcyc[i];
    mean /= nb;
    printf("%-44s workgroups %4d: %7.0f cycles per workgroup = %5.2f per MFMA and wave, launch %.1f us\n", name, nb, mean,
           mean / (ksteps * 28.0), best * 1e3);
}

int main() {
    char* w; float* out; long* cyc;
    hipMalloc(&w, 40 * 294912 + 65536); hipMemset(w, 0x3c, 40 * 294912 + 65536);
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 4096 * sizeof(long));
    long* hcyc = new long[4096];
    for (int nb : {1, 256, 512}) {
        run<0>("V0 MFMA only", nb, w, out, cyc, hcyc);
        run<1>("V1 + 1 LDS read : 2 MFMA", nb, w, out, cyc, hcyc);
        run<2>("V2 + weight loads (the kernels' K loop)", nb, w, out, cyc, hcyc);
        run<4>("V4 4 x 7 wave tile, 1 LDS read : 4 MFMA", nb, w, out, cyc, hcyc);
        run<3>("V3 + weight loads", nb, w, out, cyc, hcyc);
    }
    return 0;
}
