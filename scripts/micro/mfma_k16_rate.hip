// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 mfma_k16_rate.hip -o mfma_k16_rate
// Cycles per MFMA, one wave per SIMD, 16 independent accumulators, operands in registers:
//   v_mfma_f32_16x16x32_bf16 (K = 32) against v_mfma_f32_16x16x16_bf16 (K = 16).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int K>
__global__ __launch_bounds__(256) void k(float* out, long* cyc, int iters) {
    f32x4 acc[16];
    for (int i = 0; i < 16; i++) acc[i] = (f32x4){0, 0, 0, 0};
    bf16x8 a8, b8;
    s16x4 a4, b4;
    for (int i = 0; i < 8; i++) { a8[i] = (__bf16)(float)(threadIdx.x + i); b8[i] = (__bf16)(float)(threadIdx.x * 3 + i); }
    for (int i = 0; i < 4; i++) { a4[i] = (short)(0x3f80 + threadIdx.x + i); b4[i] = (short)(0x3f80 + i); }
    long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (K == 32) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
        }
    }
    long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][3];
    if (s == 12345.f) out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float* out; long* cyc; long h[256];
    hipMalloc(&out, 4096); hipMalloc(&cyc, 256 * sizeof(long));
    const int iters = 2000;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k<32>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        printf("K=32: %.2f cycles per MFMA\n", (double)h[0] / (iters * 16.0));
        hipLaunchKernelGGL(k<16>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        printf("K=16: %.2f cycles per MFMA\n", (double)h[0] / (iters * 16.0));
    }
    return 0;
}
