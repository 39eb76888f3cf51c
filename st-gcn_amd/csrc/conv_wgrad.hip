// conv_wgrad: weight gradient of conv_rows' forward form on MFMA.
//
//   dW[tap][co][ci] = sum_{b, to, v}  dout[b, to, v, co] * pro(in[b, stride*to - pad + tap*dil, v, ci])
//
// The contraction runs over output rows, which are NOT the contiguous axis of either operand,
// so both MFMA operands are read from LDS images kept in their global [row][channel] form:
// bf16 through ds_read_b64_tr_b16 (a 4-row x 16-column block delivered column-major), f32
// through per-element ds_read_b32 (the f32 MFMA takes one element per lane).
// One workgroup owns a (co-tile, ci-tile) pair for all taps and walks a strided subset of the
// 128-row tiles; per tile it stages dout[128 rows][co-tile] and the input window (all frames the
// taps of those rows touch, prologue applied once, frames outside [0,Tin) zero-filled).
// Partial sums go to a slab [part][tap][Cout][Cin] (f32) that conv_wgrad_reduce sums in a fixed
// order, so results are bitwise reproducible.
#include "stgcn_common.h"

struct ConvWgradArgs {
    const void* in; const void* dout; float* slab;
    const float* pro_scale; const float* pro_shift;
    int B, Tin, Tout, V, Cin, Cout;
    int ntaps, stride, pad, dil;
    int win_rows_max, tiles_per_sample, nseg;
};

constexpr int CW_RS = 144;

template <typename T, int MAXT>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(ConvWgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ES = ET<T>::ES;
    constexpr int CT = 128 / ES;            // channels per tile side: 64 (bf16) / 32 (f32)
    constexpr int VEC = ET<T>::VEC;
    constexpr bool BF = ET<T>::IS_BF16;
    char* dtile = smem;                                  // [128][CW_RS]
    int* rb = (int*)(smem + 128 * CW_RS);                // [128] window byte offset of each row
    char* win = smem + 128 * CW_RS + 128 * 4;            // [win_rows_max][CW_RS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int V = a.V;
    const int Rout = a.Tout * V;
    const int co0 = blockIdx.y * CT, ci0 = blockIdx.z * CT;
    const int coh = wave & 1, cih = wave >> 1;           // bf16: wave's 32x32 quadrant

    f32x16 acc[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // tr-read lane roles (bf16): block row q, column quad p, 16-column half g1
    const int q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = (lane >> 4) & 1;

    const int ntiles = a.B * a.tiles_per_sample;
    for (int g = blockIdx.x; g < ntiles; g += a.nseg) {
        const int b = g / a.tiles_per_sample;
        const int tile = g - b * a.tiles_per_sample;
        const int r0 = tile * 128, r1 = min(r0 + 128, Rout);
        const int tlo = r0 / V, thi = (r1 - 1) / V;
        const int flo = a.stride * tlo - a.pad;                              // may be negative
        const int fhi = a.stride * thi - a.pad + (a.ntaps - 1) * a.dil;      // may exceed Tin-1
        const int wrows = (fhi - flo + 1) * V;

        __syncthreads();     // previous tile fully consumed
        if (tid < 128) {
            const int r = min(r0 + tid, r1 - 1);
            const int to = r / V;
            rb[tid] = ((a.stride * (to - tlo)) * V + (r - to * V)) * CW_RS;
        }
        // dout tile, zero beyond the sample
        for (int p = tid; p < 128 * 8; p += 256) {
            const int row = p >> 3, seg = p & 7;
            i32x4 v = zero16();
            const int co = co0 + seg * VEC;
            if (r0 + row < r1 && co < a.Cout)
                v = *(const i32x4*)((const char*)a.dout + (((size_t)b * Rout + r0 + row) * a.Cout + co) * ES);
            *(i32x4*)(dtile + row * CW_RS + seg * 16) = v;
        }
        // input window
        const int npieces = wrows * 8;
        for (int p0 = 0; p0 < npieces; p0 += 256 * 4) {
            i32x4 v[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = p0 + u * 256 + tid;
                ok[u] = false;
                v[u] = zero16();
                if (p < npieces) {
                    const int row = p >> 3, seg = p & 7;
                    const int f = flo + row / V;
                    if (f >= 0 && f < a.Tin) {
                        ok[u] = true;
                        v[u] = *(const i32x4*)((const char*)a.in + ((((size_t)b * a.Tin + f) * V + (row % V)) * a.Cin + ci0 + seg * VEC) * ES);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = p0 + u * 256 + tid;
                if (p < npieces) {
                    const int row = p >> 3, seg = p & 7;
                    if (ok[u] && a.pro_scale) {
                        float f[VEC];
                        unpack16<T>(v[u], f);
                        const int c0 = ci0 + seg * VEC;
#pragma unroll
                        for (int j = 0; j < VEC; ++j) f[j] = fmaxf(fmaf(f[j], a.pro_scale[c0 + j], a.pro_shift[c0 + j]), 0.f);
                        v[u] = pack16<T>(f);
                    }
                    *(i32x4*)(win + (size_t)row * CW_RS + seg * 16) = v[u];
                }
            }
        }
        __syncthreads();

        if constexpr (BF) {
            // wave (coh, cih): all 128 rows, 8 k-steps of 16 rows
            const int acol = (coh * 32 + 16 * g1 + 4 * p4) * 2;
            const int bcol = (cih * 32 + 16 * g1 + 4 * p4) * 2;
#pragma unroll 2
            for (int s = 0; s < 8; ++s) {
                const int rowa = 16 * s + 8 * h + q4;       // this lane's address row, first read
                const int rowb = rowa + 4;                  // second read
                const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(dtile + rowa * CW_RS + acol));
                const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(dtile + rowb * CW_RS + acol));
                const s16x8 af = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                const int wa = rb[rowa] + bcol, wb = rb[rowb] + bcol;
#pragma unroll
                for (int t = 0; t < MAXT; ++t) {
                    if (t < a.ntaps) {
                        const int sh = t * a.dil * V * CW_RS;
                        const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(win + wa + sh));
                        const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(win + wb + sh));
                        const s16x8 bfv = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfv, acc[t], 0, 0, 0);
                    }
                }
            }
        } else {
            // wave owns rows [32*wave, 32*wave+32); 16 steps of 2 rows
#pragma unroll 4
            for (int s = 0; s < 16; ++s) {
                const int row = 32 * wave + 2 * s + h;
                const float av = *(const float*)(dtile + row * CW_RS + l31 * 4);
                const int wo = rb[row] + l31 * 4;
#pragma unroll
                for (int t = 0; t < MAXT; ++t) {
                    if (t < a.ntaps) {
                        const float bv = *(const float*)(win + wo + t * a.dil * V * CW_RS);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
                    }
                }
            }
        }
    }

    // partial slab: acc[t] row = co (acc_row), col = ci (lane)
    const int part = BF ? blockIdx.x : blockIdx.x * 4 + wave;
    const int cob = co0 + (BF ? coh * 32 : 0);
    const int ci = ci0 + (BF ? cih * 32 : 0) + l31;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        if (t < a.ntaps) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = cob + acc_row(r, h);
                if (co < a.Cout && ci < a.Cin)
                    a.slab[(((size_t)part * a.ntaps + t) * a.Cout + co) * a.Cin + ci] = acc[t][r];
            }
        }
    }
}

// dW[co][ci][tap] = sum_part slab[part][tap][co][ci]   (torch conv weight layout [Cout][Cin][ntaps][1])
__global__ void conv_wgrad_reduce_kernel(const float* slab, float* dW, int nparts, int ntaps, int Cout, int Cin) {
    const size_t n = (size_t)ntaps * Cout * Cin;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int p = 0; p < nparts; ++p) s += slab[(size_t)p * n + i];
        const int ci = (int)(i % Cin);
        const size_t rest = i / Cin;
        const int co = (int)(rest % Cout);
        const int tap = (int)(rest / Cout);
        dW[((size_t)co * Cin + ci) * ntaps + tap] = s;
    }
}

template <typename T>
static int conv_wgrad_run(ConvWgradArgs& a, float* dW, size_t ws_bytes, hipStream_t stream) {
    constexpr int CT = 128 / ET<T>::ES;
    if (a.Cin % CT != 0 || a.Cout <= 0 || a.ntaps < 1 || a.ntaps > 9 || a.V <= 0 || a.V > 64) return STGCN_ERR_ARG;
    const int nf = (128 - 2) / a.V + 2;
    a.win_rows_max = (a.stride * (nf - 1) + (a.ntaps - 1) * a.dil + 1) * a.V;
    const size_t lds = 128 * CW_RS + 128 * 4 + (size_t)a.win_rows_max * CW_RS;
    if (lds > 160 * 1024) return STGCN_ERR_ARG;
    a.tiles_per_sample = ceil_div_i(a.Tout * a.V, 128);
    const int ncot = ceil_div_i(a.Cout, CT), ncit = a.Cin / CT;
    const int ntiles = a.B * a.tiles_per_sample;
    int nseg = 1024 / (ncot * ncit);
    if (nseg < 1) nseg = 1;
    if (nseg > ntiles) nseg = ntiles;
    const int wave_parts = ET<T>::IS_BF16 ? 1 : 4;
    const size_t per_part = (size_t)a.ntaps * a.Cout * a.Cin * sizeof(float);
    while (nseg > 1 && (size_t)nseg * wave_parts * per_part > ws_bytes) nseg /= 2;
    if ((size_t)nseg * wave_parts * per_part > ws_bytes) return STGCN_ERR_WORKSPACE;
    a.nseg = nseg;
    dim3 grid((unsigned)nseg, (unsigned)ncot, (unsigned)ncit);
    int rc;
    if (a.ntaps == 1) {
        auto kern = conv_wgrad_kernel<T, 1>;
        static bool set1 = false;
        if (!set1) { if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return STGCN_ERR_LAUNCH; set1 = true; }
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a);
    } else {
        auto kern = conv_wgrad_kernel<T, 9>;
        static bool set9 = false;
        if (!set9) { if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return STGCN_ERR_LAUNCH; set9 = true; }
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a);
    }
    rc = hipGetLastError() == hipSuccess ? STGCN_OK : STGCN_ERR_LAUNCH;
    if (rc) return rc;
    const size_t n = (size_t)a.ntaps * a.Cout * a.Cin;
    const int left = fold_parts(a.slab, nseg * wave_parts, n, stream);
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       a.slab, dW, left, a.ntaps, a.Cout, a.Cin);
    return hipGetLastError() == hipSuccess ? STGCN_OK : STGCN_ERR_LAUNCH;
}

extern "C" int stgcn_conv_wgrad(int dtype, const void* in, const void* dout, float* dW,
                                const float* pro_scale, const float* pro_shift,
                                int B, int Tin, int Tout, int V, int Cin, int Cout,
                                int ntaps, int stride, int pad, int dil,
                                void* workspace, size_t workspace_bytes, void* stream) {
    if (!in || !dout || !dW || !workspace) return STGCN_ERR_ARG;
    if ((pro_scale == nullptr) != (pro_shift == nullptr)) return STGCN_ERR_ARG;
    ConvWgradArgs a;
    a.in = in; a.dout = dout; a.slab = (float*)workspace; a.pro_scale = pro_scale; a.pro_shift = pro_shift;
    a.B = B; a.Tin = Tin; a.Tout = Tout; a.V = V; a.Cin = Cin; a.Cout = Cout;
    a.ntaps = ntaps; a.stride = stride; a.pad = pad; a.dil = dil;
    a.win_rows_max = 0; a.tiles_per_sample = 0; a.nseg = 0;
    if (dtype == STGCN_BF16) return conv_wgrad_run<bf16>(a, dW, workspace_bytes, (hipStream_t)stream);
    if (dtype == STGCN_F32) return conv_wgrad_run<float>(a, dW, workspace_bytes, (hipStream_t)stream);
    return STGCN_ERR_ARG;
}
