// conv1s: the 1-tap temporal convolution in bf16 (the strided residual branch of a block and its data gradient) as a
// streaming product.  One tap means no halo and little arithmetic per byte (K = Cin <= 256): the kernel is bound by
// HBM, so it is built for many independent loads and stores in flight, not for MFMA rate.
//
//   forward,    stride s : out[b, j, v, :]     = bias + W . in[b, s*j, v, :]            j < Tout
//   transposed, stride s : out[b, s*j, v, :]   = W . in[b, j, v, :]                     j < Tin
//                          out[b, s*j+1 .. s*j+s-1, v, :] = 0                           (frames no input frame maps to)
//
// Both are rows-times-matrix over VIRTUAL rows u = (b, j, v): input row (b*Tin + fs*j)*V + v, output row
// (b*Tout + os*j)*V + v.
//   - The whole weight matrix [Cout][Cin] sits in LDS for the life of the workgroup.
//   - Each of the 8 waves streams its own blocks of 32 virtual rows and never meets the others at a barrier: its A
//     fragments come straight from global memory into registers (the next block's are in flight while this one
//     multiplies), the B fragments from LDS; the 32 x 128 (or 32 x 64) result goes through a wave-private LDS patch
//     so that it leaves as 16-byte pieces of whole rows, and the per-channel sums of the BatchNorm that follows are
//     gathered from those rounded pieces (fixed channels per lane, fixed block order per wave, one fixed-order fold
//     per workgroup, then one double atomic per channel).
#include "stgcn_common.h"

struct Conv1sArgs {
    const void* in; const void* w; void* out;
    const float* bias; double* stats;
    int Cout, V, Tv, Tin, Tout, fs, os;
    int nfill;            // real frames after os*j that are zero-filled (transposed: os - 1)
    int nrows, nblocks;   // virtual rows, blocks of 32 of them
};

constexpr int C1_WAVES = 8;

// KS = Cin / 16 k-steps, NT = 32-column tiles per pass (4: 128 columns, 2: 64 columns)
template <int KS, int NT>
__global__ __launch_bounds__(C1_WAVES * 64) void conv1s_kernel(Conv1sArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CIN = KS * 16;
    constexpr int WRS = CIN * 2 + 16;                    // weight row in LDS: padded against bank conflicts
    constexpr int PC = NT * 32;                          // columns per pass
    constexpr int SRS = PC * 2 + 16;                     // staging row
    constexpr int P = PC / 8;                            // 16-byte pieces per staged row: 16 or 8
    constexpr int R = 64 / P;                            // rows per store instruction: 4 or 8
    constexpr int PATCH = 32 * SRS + 256;                // per wave: 32 staged rows + 32 output rows + 32 fill flags
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int Cout = a.Cout, V = a.V;
    const int npass = Cout / PC;
    char* wl = smem;
    char* patch = smem + (size_t)Cout * WRS + (size_t)wave * PATCH;
    int* orow_l = (int*)(patch + 32 * SRS);              // [32] output row of block row i, -1 = none
    int* fill_l = orow_l + 32;                           // [32] frames to zero-fill after it

    // weights -> LDS (16-byte pieces, rows padded)
    for (int i = tid; i < Cout * (CIN / 8); i += C1_WAVES * 64) {
        const int row = i / (CIN / 8), pc = i - row * (CIN / 8);
        *(i32x4*)(wl + (size_t)row * WRS + pc * 16) = *(const i32x4*)((const char*)a.w + ((size_t)row * CIN + pc * 8) * 2);
    }
    __syncthreads();

    float s1[2][8], s2[2][8];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int j = 0; j < 8; ++j) { s1[p][j] = 0.f; s2[p][j] = 0.f; }

    const int rows_per_sample = a.Tv * V;
    // this lane's virtual row of block blk: input row (or -1), output row, frames to fill
    auto geo = [&](int blk, int& irow, int& orow, int& fill) {
        const int u = blk * 32 + l31;
        irow = -1; orow = -1; fill = 0;
        if (u < a.nrows) {
            const int b = u / rows_per_sample, rem = u - b * rows_per_sample;
            const int j = rem / V, v = rem - j * V;
            irow = (b * a.Tin + a.fs * j) * V + v;
            orow = (b * a.Tout + a.os * j) * V + v;
            fill = min(a.nfill, a.Tout - 1 - a.os * j);
        }
    };
    // Lanes without a row (past the last block) read row 0 and their results are never stored: the loads are issued
    // on every path, so that the compiler's wait counts see exactly what is in flight (see the loop below).
    auto load_a = [&](i32x4 (&af)[KS], int irow) {
        const char* src = (const char*)a.in + ((size_t)max(irow, 0) * CIN + h * 8) * 2;
#pragma unroll
        for (int s = 0; s < KS; ++s) af[s] = *(const i32x4*)(src + s * 32);
    };
    const int piece = lane % P, prow = lane / P;
    auto block = [&](const i32x4 (&af)[KS], i32x4 (&nxt)[KS], int orow, int fill) {
        if (h == 0) { orow_l[l31] = orow; fill_l[l31] = fill; }
#pragma unroll
        for (int p = 0; p < 2; ++p) {                      // at most two passes (Cout <= 256); unrolled: s1/s2 stay in registers
            if (p >= npass) break;
            f32x16 acc[NT];
            float biasv[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                biasv[n] = a.bias ? a.bias[p * PC + n * 32 + l31] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
            }
            const char* wb = wl + (size_t)(p * PC + l31) * WRS + h * 16;
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const i32x4 bf = *(const i32x4*)(wb + (size_t)n * 32 * WRS + s * 32);
                    mma_step<bf16>(acc[n], af[s], bf);
                }
            if (p == 0) {
                // the one wait for the next block's fragments: behind this block's first MFMA run and before its
                // stores, so that no later wait for them is also a wait for those stores
#pragma unroll
                for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(nxt[s]));
            }
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    *(uint16_t*)(patch + acc_row(r, h) * SRS + (n * 32 + l31) * 2) = (uint16_t)(pack_bf16x2(acc[n][r] + biasv[n], 0.f) & 0xFFFFu);
            // one wave, LDS operations in order: the pieces read below see the writes above
#pragma unroll
            for (int i = 0; i < 32 / R; ++i) {
                const int row = i * R + prow;
                const i32x4 v = *(const i32x4*)(patch + row * SRS + piece * 16);
                const int orow_i = orow_l[row], fill_i = fill_l[row];
                if (orow_i >= 0) {
                    char* dst = (char*)a.out + ((size_t)orow_i * Cout + p * PC + piece * 8) * 2;
                    *(i32x4*)dst = v;
                    for (int f = 1; f <= fill_i; ++f) *(i32x4*)(dst + (size_t)f * V * Cout * 2) = zero16();
                    float fv[8];
                    unpack16<bf16>(v, fv);
#pragma unroll
                    for (int j = 0; j < 8; ++j) { s1[p][j] += fv[j]; s2[p][j] = fmaf(fv[j], fv[j], s2[p][j]); }
                }
            }
        }
    };

    const int gw = blockIdx.x * C1_WAVES + wave, gstep = gridDim.x * C1_WAVES;
    i32x4 af0[KS], af1[KS];
    int ir, or0, fl0, or1, fl1;
    int blk = gw;
    geo(blk, ir, or0, fl0);                                // geo gives "no row" (-1) to every lane of a block past the end
    load_a(af0, ir);
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(af0[s]));
    while (blk < a.nblocks) {
        const int b1 = blk + gstep;
        geo(b1, ir, or1, fl1);
        load_a(af1, ir);
        block(af0, af1, or0, fl0);
        if (b1 >= a.nblocks) break;
        const int b2 = b1 + gstep;
        geo(b2, ir, or0, fl0);
        load_a(af0, ir);
        block(af1, af0, or1, fl1);
        blk = b2;
    }

    if (a.stats) {
        __syncthreads();
        float* red = (float*)smem;                         // [512][16]
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            if (p >= npass) break;
#pragma unroll
            for (int j = 0; j < 8; ++j) { red[tid * 16 + j] = s1[p][j]; red[tid * 16 + 8 + j] = s2[p][j]; }
            __syncthreads();
            if (tid < 2 * PC) {
                const int which = tid / PC, c = tid % PC;
                const int pc = c / 8, j = c % 8;
                float t = 0.f;
                for (int w = 0; w < C1_WAVES; ++w)
                    for (int q = 0; q < R; ++q) t += red[(w * 64 + q * P + pc) * 16 + which * 8 + j];
                atomicAdd(&a.stats[which * Cout + p * PC + c], (double)t);
            }
            __syncthreads();
        }
    }
}

template <int KS, int NT>
static int launch_conv1s(Conv1sArgs& a, hipStream_t stream) {
    constexpr int CIN = KS * 16, WRS = CIN * 2 + 16, PC = NT * 32, SRS = PC * 2 + 16, PATCH = 32 * SRS + 256;
    size_t lds = (size_t)a.Cout * WRS + (size_t)C1_WAVES * PATCH;
    if (lds < (size_t)C1_WAVES * 64 * 16 * sizeof(float)) lds = (size_t)C1_WAVES * 64 * 16 * sizeof(float);
    if (lds > 160 * 1024) return -1;
    auto kern = conv1s_kernel<KS, NT>;
    static std::atomic<uint64_t> lds_ok{0};
    if (stgcn_allow_big_lds((const void*)kern, lds_ok) != STGCN_OK) return STGCN_ERR_LAUNCH;
    int gx = min(stgcn_num_cus(), ceil_div_i(a.nblocks, C1_WAVES));
    hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3(C1_WAVES * 64), lds, stream, a);
    return hipGetLastError() == hipSuccess ? STGCN_OK : STGCN_ERR_LAUNCH;
}

// Returns -1 when the shape is outside this kernel's domain (the caller then uses the tiled kernels).
int conv1s_try(const void* in, const void* w, void* out, const float* bias, double* stats,
               int B, int Tin, int Tout, int V, int Cin, int Cout, int stride, int transposed, hipStream_t stream) {
    if (stride < 1 || stride > 2 || V < 1 || B < 1 || Tin < 1 || Tout < 1) return -1;
    if ((Cin != 64 && Cin != 128 && Cin != 256) || Cout % 64 != 0 || Cout < 64 || Cout > 256) return -1;
    Conv1sArgs a;
    a.in = in; a.w = w; a.out = out; a.bias = bias; a.stats = stats;
    a.Cout = Cout; a.V = V; a.Tin = Tin; a.Tout = Tout;
    if (!transposed) {
        if ((long long)(Tout - 1) * stride > Tin - 1) return -1;
        a.Tv = Tout; a.fs = stride; a.os = 1; a.nfill = 0;
    } else {
        // every output frame must be s*j or one of the s-1 frames after it, for an input frame j
        if ((long long)(Tin - 1) * stride > Tout - 1 || (long long)Tin * stride < Tout) return -1;
        a.Tv = Tin; a.fs = 1; a.os = stride; a.nfill = stride - 1;
    }
    const long long nrows = (long long)B * a.Tv * V;
    if (nrows > (1ll << 30) || (long long)B * max(Tin, Tout) * V > (1ll << 30)) return -1;     // row indices are ints
    a.nrows = (int)nrows; a.nblocks = (int)((nrows + 31) / 32);
    const bool wide = Cout % 128 == 0;
    if (!wide && Cout > 128) return -1;                   // 64-column passes: the kernel unrolls two of them
    switch (Cin) {
    case 64:  return wide ? launch_conv1s<4, 4>(a, stream) : launch_conv1s<4, 2>(a, stream);
    case 128: return wide ? launch_conv1s<8, 4>(a, stream) : launch_conv1s<8, 2>(a, stream);
    default:  return Cout <= 128 ? launch_conv1s<16, 2>(a, stream) : -1;     // 64-column passes: two sets of 16 A fragments fill the registers
    }
}

// ------------------------------------------------------------------------------------------
// Weight gradient of the same convolution:  dW[co][ci] = sum over virtual rows u of dout[orow(u)][co] * in[irow(u)][ci].
// The tiled weight-gradient kernels give a workgroup one (64 co x 64 ci) pair, so every pair re-reads its rows; with
// one tap the whole Cout x Cin result (256 x 128 or 128 x 64) fits the accumulators of ONE workgroup: wave
// (wco, wci) owns TCO x TCI tiles of 32 x 32, the workgroup streams batches of 64 rows of both tensors through LDS
// once (double-buffered, the next batch in registers while this one multiplies), and both MFMA operands come out of
// the [row][channel] images through ds_read_b64_tr_b16, the contraction running over the rows.  One partial per
// workgroup goes to the slab; the fixed-order slab reduction of the tiled kernels sums them.
struct Conv1sWgArgs {
    const void* in; const void* dout; float* slab;
    int V, Tv, Tin, fs;
    int nrows, nbatches;
};

template <int TCO, int WCO, int TCI, int WCI>
__global__ __launch_bounds__(512) void conv1s_wgrad_kernel(Conv1sWgArgs a) {
    static_assert(WCO * WCI == 8, "eight waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int COUT = 32 * TCO * WCO, CIN = 32 * TCI * WCI;
    constexpr int DRS = COUT * 2 + 16, XRS = CIN * 2 + 16, RB = 64;
    constexpr int DPP = COUT / 8, XPP = CIN / 8;          // 16-byte pieces per row
    constexpr int ND = RB * DPP / 512, NX = RB * XPP / 512;
    static_assert(ND >= 1 && NX >= 1, "a batch is at least one piece per thread");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = (lane >> 4) & 1;
    const int wco = wave % WCO, wci = wave / WCO;
    char* dimg = smem;                                    // [2][RB][DRS]
    char* ximg = smem + 2 * RB * DRS;                     // [2][RB][XRS]
    const int V = a.V, Rv = a.Tv * V;

    f32x16 acc[TCO][TCI];
#pragma unroll
    for (int i = 0; i < TCO; ++i)
#pragma unroll
        for (int j = 0; j < TCI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    i32x4 dreg[ND], xreg[NX];
    auto load_batch = [&](int bi) {
        const int u0 = bi * RB;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int p = tid + 512 * i, row = p / DPP, seg = p - row * DPP;
            const int u = u0 + row;
            dreg[i] = zero16();
            if (u < a.nrows) dreg[i] = *(const i32x4*)((const char*)a.dout + ((size_t)u * COUT + seg * 8) * 2);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int p = tid + 512 * i, row = p / XPP, seg = p - row * XPP;
            const int u = u0 + row;
            xreg[i] = zero16();
            if (u < a.nrows) {
                const int b = u / Rv, rem = u - b * Rv;
                const int j = rem / V, v = rem - j * V;
                const size_t irow = ((size_t)b * a.Tin + a.fs * j) * V + v;
                xreg[i] = *(const i32x4*)((const char*)a.in + (irow * CIN + seg * 8) * 2);
            }
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int p = tid + 512 * i, row = p / DPP, seg = p - row * DPP;
            *(i32x4*)(dimg + ((size_t)buf * RB + row) * DRS + seg * 16) = dreg[i];
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int p = tid + 512 * i, row = p / XPP, seg = p - row * XPP;
            *(i32x4*)(ximg + ((size_t)buf * RB + row) * XRS + seg * 16) = xreg[i];
        }
    };
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
    auto compute = [&](int buf) {
        const char* di = dimg + (size_t)buf * RB * DRS;
        const char* xi = ximg + (size_t)buf * RB * XRS;
#pragma unroll
        for (int s = 0; s < RB / 16; ++s) {
            const int row = 16 * s + 8 * h + q4;
            s16x8 af[TCO], bf[TCI];
#pragma unroll
            for (int t = 0; t < TCO; ++t) {
                const int col = ((wco * TCO + t) * 32 + 16 * g1 + 4 * p4) * 2;
                const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(di + row * DRS + col));
                const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(di + (row + 4) * DRS + col));
                af[t] = s16x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            }
#pragma unroll
            for (int t = 0; t < TCI; ++t) {
                const int col = ((wci * TCI + t) * 32 + 16 * g1 + 4 * p4) * 2;
                const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(xi + row * XRS + col));
                const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(xi + (row + 4) * XRS + col));
                bf[t] = s16x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
            }
#pragma unroll
            for (int i = 0; i < TCO; ++i)
#pragma unroll
                for (int j = 0; j < TCI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };

    int bi = blockIdx.x;
    if (bi < a.nbatches) { load_batch(bi); commit(0); }
    lds_barrier();
    for (int it = 0; bi < a.nbatches; ++it, bi += gridDim.x) {
        const int nb = bi + gridDim.x;
        if (nb < a.nbatches) load_batch(nb);
        compute(it & 1);
        if (nb < a.nbatches) commit((it + 1) & 1);
        lds_barrier();
    }

    float* part = a.slab + (size_t)blockIdx.x * COUT * CIN;
#pragma unroll
    for (int i = 0; i < TCO; ++i)
#pragma unroll
        for (int j = 0; j < TCI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (wco * TCO + i) * 32 + acc_row(r, h), ci = (wci * TCI + j) * 32 + l31;
                part[(size_t)co * CIN + ci] = acc[i][j][r];
            }
}

template <int TCO, int WCO, int TCI, int WCI>
static int launch_conv1s_wgrad(Conv1sWgArgs& a, void* ws, size_t ws_bytes, int* nparts, hipStream_t stream) {
    constexpr int COUT = 32 * TCO * WCO, CIN = 32 * TCI * WCI;
    const size_t lds = 2 * 64 * (size_t)(COUT * 2 + 16) + 2 * 64 * (size_t)(CIN * 2 + 16);
    const size_t per_part = (size_t)COUT * CIN * sizeof(float);
    int gx = min(stgcn_num_cus(), a.nbatches);
    while (gx > 1 && (size_t)gx * per_part > ws_bytes) gx /= 2;
    if ((size_t)gx * per_part > ws_bytes) return STGCN_ERR_WORKSPACE;
    a.slab = (float*)ws;
    auto kern = conv1s_wgrad_kernel<TCO, WCO, TCI, WCI>;
    static std::atomic<uint64_t> lds_ok{0};
    if (stgcn_allow_big_lds((const void*)kern, lds_ok) != STGCN_OK) return STGCN_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3(512), lds, stream, a);
    *nparts = gx;
    return hipGetLastError() == hipSuccess ? STGCN_OK : STGCN_ERR_LAUNCH;
}

// Launches the partial sums; -1 when the shape is outside this kernel's domain.  On STGCN_OK *nparts partials
// [Cout][Cin] (float32) lie at the start of the workspace for the caller's fixed-order reduction.
int conv1s_wgrad_try(const void* in, const void* dout, int B, int Tin, int Tout, int V, int Cin, int Cout, int stride,
                     void* ws, size_t ws_bytes, int* nparts, hipStream_t stream) {
    if (stride < 1 || stride > 2 || V < 1 || B < 1 || Tin < 1 || Tout < 1 || (long long)(Tout - 1) * stride > Tin - 1) return -1;
    const long long nrows = (long long)B * Tout * V;
    if (nrows > (1ll << 30) || (long long)B * Tin * V > (1ll << 30)) return -1;
    Conv1sWgArgs a;
    a.in = in; a.dout = dout; a.slab = nullptr; a.V = V; a.Tv = Tout; a.Tin = Tin; a.fs = stride;
    a.nrows = (int)nrows; a.nbatches = (int)((nrows + 63) / 64);
    if (Cout == 256 && Cin == 128) return launch_conv1s_wgrad<2, 4, 2, 2>(a, ws, ws_bytes, nparts, stream);
    if (Cout == 128 && Cin == 64) return launch_conv1s_wgrad<1, 4, 1, 2>(a, ws, ws_bytes, nparts, stream);
    return -1;
}
