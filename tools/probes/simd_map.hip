// Which SIMD does each wave of a 512-thread workgroup land on?  Prints SIMD_ID (HW_ID bits 5:4) per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(512) void probe(unsigned* out) {
    extern __shared__ char smem[];
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
    if (threadIdx.x == 1000) smem[0] = 1;
}
int main() {
    unsigned* d; hipMalloc(&d, 64 * 8 * 4);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(probe, dim3(64), dim3(512), 150 * 1024, 0, d);
    unsigned h[64 * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 12; ++b) { printf("wg %2d: simd", b); for (int w = 0; w < 8; ++w) printf(" %u", (h[b * 8 + w] >> 4) & 3); printf("   cu %u se %u\n", (h[b*8] >> 8) & 15, (h[b*8] >> 13) & 7); }
    return 0;
}
