// What v_permlane16_swap_b32 does on a 64-lane wave: prints, per 16-lane row, which (operand, row) each result holds.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* a, unsigned* b) {
    unsigned x = 0x100 + threadIdx.x, y = 0x200 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    a[threadIdx.x] = r[0]; b[threadIdx.x] = r[1];
}
int main() {
    unsigned *a, *b, ha[64], hb[64];
    if (hipMalloc(&a, 256) != hipSuccess || hipMalloc(&b, 256) != hipSuccess) return 1;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b);
    if (hipMemcpy(ha, a, 256, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (hipMemcpy(hb, b, 256, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    for (int r = 0; r < 4; r++)
        printf("row %d: first result = operand %x row %d lane0 %x | second result = operand %x row %d lane0 %x\n", r,
               ha[16 * r] >> 8, (ha[16 * r] & 0xff) / 16, ha[16 * r] & 0xf, hb[16 * r] >> 8, (hb[16 * r] & 0xff) / 16, hb[16 * r] & 0xf);
    return 0;
}
