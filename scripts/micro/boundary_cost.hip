// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 boundary_cost.hip -o boundary_cost
// What does a kernel boundary cost behind a kernel that leaves dirty lines in the L2s?  256 workgroups write
// `kb` KiB each (plain or streaming stores), then spin for a fixed number of cycles; 40 launches back to back in
// one stream, time per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(u32x4* out, int kb, int spin, int shift) {
    const long t0 = clock64();
    u32x4* o = out + ((size_t)((blockIdx.x + shift) % gridDim.x) * kb * 64);
    const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    for (int i = threadIdx.x; i < kb * 64; i += 256) {
        if (MODE == 1) o[i] = v;
        if (MODE == 2) __builtin_nontemporal_store(v, o + i);
    }
    while (clock64() - t0 < spin) {}
}

template <int MODE>
void run(const char* name, int nb, int kb, u32x4* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < 40; i++) hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(256), 0, 0, out, kb, 40000, i);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 1 && ms < best) best = ms;
    }
    printf("%-20s %4d workgroups x %4d KiB: %.2f us per launch\n", name, nb, kb, best * 1e3 / 40);
}

int main() {
    u32x4* out; hipMalloc(&out, (size_t)1024 * 1024 * 1024);
    for (int nb : {256, 512})
        for (int kb : {0, 100, 400}) {
            if (kb == 0) { run<0>("no stores", nb, kb, out); continue; }
            run<1>("plain stores", nb, kb, out);
            run<2>("streaming stores", nb, kb, out);
        }
    return 0;
}
