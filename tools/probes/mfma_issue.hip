// Issue rate of bf16 MFMAs from ONE wave's instruction stream, no barriers and no memory traffic inside the timed
// region: 64 MFMAs in a row over NACC accumulators, with 1 wave per SIMD (256 threads) or 2 (512 threads, both computing).
// Prints shader cycles per MFMA per wave (s_memtime) and the clock (s_memtime ticks per s_memrealtime tick x 100 MHz).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE, int NACC, int NTH, int BAR, int HALF>
__global__ __launch_bounds__(NTH) void probe(unsigned long long* out, const s16x8* src, float* sink) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    s16x8 fa[4], fb[4];
    for (int i = 0; i < 4; ++i) { fa[i] = src[(tid * 8 + i) & 4095]; fb[i] = src[(tid * 8 + 4 + i) & 4095]; }
    f32x16 acc[NACC];
    f32x4 acc4[NACC];
    for (int i = 0; i < NACC; ++i) { for (int r = 0; r < 16; ++r) acc[i][r] = 0.f; for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f; }
    __syncthreads();
    unsigned long long t0 = 0, t1 = 0, r0 = 0, r1 = 0;
    for (int rep = 0; rep < 6; ++rep) {
        __builtin_amdgcn_sched_barrier(0);
        if (rep == 2) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
        __builtin_amdgcn_sched_barrier(0);
        const bool mine = !HALF || __builtin_amdgcn_readfirstlane(wave) < 4;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (BAR) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
            if (mine) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int i = g * 16 + j;
                    if (SHAPE == 32) acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i & 3], fb[(i >> 2) & 3], acc[i % NACC], 0, 0, 0);
                    else acc4[i % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i & 3], fb[(i >> 2) & 3], acc4[i % NACC], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) { for (int r = 0; r < 16; ++r) s += acc[i][r]; for (int r = 0; r < 4; ++r) s += acc4[i][r]; }
    sink[blockIdx.x * NTH + tid] = s;
    if (lane == 0) { out[(blockIdx.x * 8 + wave) * 2] = t1 - t0; out[(blockIdx.x * 8 + wave) * 2 + 1] = r1 - r0; }
}

template <int SHAPE, int NACC, int NTH, int BAR = 0, int HALF = 0>
static void run(unsigned long long* d, const s16x8* src, float* sink) {
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((probe<SHAPE, NACC, NTH, BAR, HALF>), dim3(256), dim3(NTH), 0, 0, d, src, sink);
    static unsigned long long h[256 * 16];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const int wg = 100;
    const double cyc = h[(wg * 8) * 2] / 256.0, real = (double)h[(wg * 8) * 2 + 1];
    printf("%dx%d  %d acc  %d waves/SIMD bar%d half%d : %.1f cycles per MFMA per wave (%.1f per SIMD)   clock %.2f GHz\n", SHAPE, SHAPE, NACC, NTH / 256, BAR, HALF,
           cyc, cyc / (NTH / 256), h[(wg * 8) * 2] / real * 0.1);
}

int main() {
    unsigned long long* d; s16x8* src; float* sink;
    hipMalloc(&d, 256 * 16 * 8); hipMalloc(&src, 4096 * 16); hipMalloc(&sink, 256 * 512 * 4);
    static unsigned short hs[4096 * 8];
    unsigned x = 12345u;
    for (int i = 0; i < 4096 * 8; ++i) { x = x * 1664525u + 1013904223u; hs[i] = (unsigned short)(0x3c00u + ((x >> 16) & 0x3ffu) + ((x >> 31) << 15)); }
    hipMemcpy(src, hs, sizeof(hs), hipMemcpyHostToDevice);
    run<32, 1, 256>(d, src, sink);
    run<32, 4, 256>(d, src, sink);
    run<32, 4, 512>(d, src, sink);
    run<32, 4, 256, 1, 0>(d, src, sink);
    run<32, 4, 512, 1, 1>(d, src, sink);
    run<32, 4, 512, 1, 0>(d, src, sink);
    run<32, 4, 512, 0, 1>(d, src, sink);
    run<16, 1, 256>(d, src, sink);
    run<16, 4, 256>(d, src, sink);
    run<16, 16, 256>(d, src, sink);
    run<16, 4, 512>(d, src, sink);
    return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
