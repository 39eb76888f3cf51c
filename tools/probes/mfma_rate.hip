// Cycles one wave needs to issue 16 v_mfma_f32_32x32x16_bf16 (4 accumulators) on its SIMD, in an 8-wave workgroup:
//   mode 0: waves 0-3 compute, waves 4-7 wait at the barrier
//   mode 1: waves 0-3 compute, waves 4-7 each read 16 x ds_read_b128 meanwhile
//   mode 2: all 8 waves compute
//   mode 3: mode 1 plus 4 global_load_lds per loading wave
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void probe(unsigned long long* out, const int* src, int mode, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 64 * 1024 / 4; i += 512) ((int*)smem)[i] = src[i & 1023];
    __syncthreads();
    const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    i32x4 fa[8], fb[8];
    for (int i = 0; i < 8; ++i) { fa[i] = *(const i32x4*)(smem + (lane * 144 + i * 32) % 60000); fb[i] = *(const i32x4*)(smem + 4096 + (lane * 144 + i * 32) % 60000); }
    unsigned long long t0 = 0, t1 = 0;
    for (int it = 0; it < 20; ++it) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (it == 10) t0 = __builtin_amdgcn_s_memtime();
        if (grp == 0 || mode == 2) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8, fa[2 * s + (j & 1)]), __builtin_bit_cast(s16x8, fb[2 * s + (j >> 1)]), acc[j], 0, 0, 0);
        } else if (mode == 1 || mode == 3) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                fa[i] = *(const i32x4*)(smem + ((lane & 31) * 144 + (lane >> 5) * 16 + i * 32 + it * 4608) % 60000);
                fb[i] = *(const i32x4*)(smem + 8192 + ((lane & 31) * 144 + (lane >> 5) * 16 + i * 32 + it * 4608) % 60000);
            }
            if (mode == 3) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane * 4 + i * 256 + (wave & 3) * 1024),
                                                     (__attribute__((address_space(3))) void*)(smem + 61440 + ((wave & 3) * 4 + i) * 1024), 16, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (it == 18) t1 = __builtin_amdgcn_s_memtime();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += (float)(fa[i][0] + fb[i][1]);
    sink[blockIdx.x * 512 + tid] = s;
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}
int main() {
    unsigned long long* d; int* src; float* sink;
    hipMalloc(&d, 256 * 8 * 8); hipMalloc(&src, 1 << 20); hipMalloc(&sink, 256 * 512 * 4);
    hipMemset(src, 0x3c, 1 << 20);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(256), dim3(512), 150 * 1024, 0, d, src, mode, sink);
        unsigned long long h[256 * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d: cycles per phase (8 phases timed) wave0 %.0f wave4 %.0f  (wg 100: %.0f %.0f)\n", mode, h[0] / 8.0, h[4] / 8.0, h[800] / 8.0, h[804] / 8.0);
    }
    return 0;
}
