// build: hipcc -O3 --offload-arch=gfx950 grid_sync.hip -o grid_sync
// Cost of a grid-wide barrier on MI355X: 512 workgroups x 256 threads (2 per CU, the residency of the
// fused node kernel: 74 KB of LDS each), cooperative launch, cooperative_groups grid sync.
// Between barriers each workgroup writes a line another workgroup (on another XCD) reads afterwards,
// and the values are checked, so the barrier's memory ordering is exercised too.
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <vector>
namespace cg = cooperative_groups;

__global__ __launch_bounds__(256, 2) void k(int* buf, int iters, int* bad) {
    extern __shared__ char lds[];
    cg::grid_group g = cg::this_grid();
    const int nb = gridDim.x, b = blockIdx.x;
    int errors = 0;
    for (int it = 0; it < iters; it++) {
        if (threadIdx.x < 32) buf[b * 32 + threadIdx.x] = it * 1000 + b;
        g.sync();
        const int other = (b * 7 + 3 + it) % nb;
        if (threadIdx.x < 32 && buf[other * 32 + threadIdx.x] != it * 1000 + other) errors++;
        g.sync();
    }
    if (errors) atomicAdd(bad, errors);
    if (threadIdx.x == 0) lds[0] = 1;
}

int main() {
    const int nb = 512, iters = 200;
    int *buf, *bad;
    hipMalloc(&buf, nb * 32 * sizeof(int));
    hipMalloc(&bad, sizeof(int));
    hipMemset(bad, 0, sizeof(int));
    const int ldsb = 74 * 1024;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    int maxb = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&maxb, k, 256, ldsb);
    printf("max active blocks per CU: %d\n", maxb);
    int it = iters;
    void* args[] = {&buf, &it, &bad};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0, 0);
        hipError_t err = hipLaunchCooperativeKernel((const void*)k, dim3(nb), dim3(256), args, ldsb, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        int hb = -1;
        hipMemcpy(&hb, bad, sizeof(int), hipMemcpyDeviceToHost);
        printf("launch %s: %d iterations x 2 barriers in %.1f us -> %.2f us per barrier, stale reads %d\n",
               hipGetErrorString(err), iters, ms * 1e3, ms * 1e3 / (2 * iters), hb);
    }
    return 0;
}
