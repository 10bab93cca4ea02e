// build: hipcc -O3 --offload-arch=gfx950 launch_ramp.hip -o launch_ramp
// How long does the device take to get N workgroups of the fused node kernel's shape going?
// Kernels that do nothing but touch LDS once, with 256 threads, varying LDS size and register count.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int REGS>
__global__ __launch_bounds__(256, 2) void k(float* out, int spin) {
    extern __shared__ float lds[];
    float v[REGS];
#pragma unroll
    for (int i = 0; i < REGS; i++) v[i] = threadIdx.x * (i + 1.f);
    long t0 = clock64();
    while (clock64() - t0 < spin) {}
#pragma unroll
    for (int i = 1; i < REGS; i++) v[0] += v[i] * v[(i * 7) % REGS];
    lds[threadIdx.x] = v[0];
    __syncthreads();
    if (lds[(threadIdx.x + 1) & 255] == 12345.f) out[blockIdx.x] = v[0];
}

template <int REGS>
void run(const char* name, int nb, int ldsb, int spin, float* out) {
    hipFuncSetAttribute((const void*)k<REGS>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 20; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<REGS>, dim3(nb), dim3(256), ldsb, 0, out, spin);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 3 && ms < best) best = ms;
    }
    printf("%-28s workgroups %4d lds %5d B spin %6d cycles: %.1f us\n", name, nb, ldsb, spin, best * 1e3);
}

int main() {
    float* out; hipMalloc(&out, 1 << 20);
    for (int nb : {256, 512, 1024}) {
        run<8>("few registers", nb, 1024, 0, out);
        run<8>("few registers, 74 KB LDS", nb, 74 * 1024, 0, out);
        run<200>("200 live registers", nb, 1024, 0, out);
        run<200>("200 registers, 74 KB LDS", nb, 74 * 1024, 0, out);
        run<200>("same + 20 us of work", nb, 74 * 1024, 42000, out);
    }
    return 0;
}
