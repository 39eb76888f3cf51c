// conv_t9: the stride-1 temporal convolution (any odd tap count up to 9, pad = (taps-1)/2) of a bf16
// channels-last activation with 64 or 128 input channels, as a persistent LDS-ring kernel.
//
// Same arithmetic as conv_rows' forward / stride-1 transposed form (see conv_rows.hip); this is the
// fast path for the shapes the st_gcn blocks spend their time in.
//
//   - A workgroup (4 waves) owns 64 output channels and a contiguous run of 256-row tiles of ONE sample
//     and walks them in order.  Input rows live in an LDS ring of 512 row slots (slot = row & 511; a tile
//     needs rows [p0 - 4V, p0 + 256 + 4V), V <= 32), so every input row is fetched from HBM once per
//     workgroup: when the walk advances, only the 256 new rows are loaded.  They are fetched into
//     registers BEFORE the tile's MFMA phase and written to the ring (BatchNorm+ReLU prologue applied
//     once per element) after it, so the fetch latency hides under ~4.6k MFMA cycles.
//   - Rows outside the sample are stored as zeros: a tap shift is a uniform row shift of (tap-4)*V, and
//     the temporal zero padding falls out of the zero rows.  No per-tap validity logic.
//   - Weights: 64 input channels -> all 9 taps resident in LDS for the whole walk (read from L2 once per
//     workgroup); 128 input channels -> per (chunk, tap) tiles double-buffered, prefetched through registers.
//   - Each wave computes 64 rows x 64 channels (2x2 accumulator tiles): 4 ds_read_b128 per 4 MFMAs.
//   - Epilogue: accumulators (+bias) are rounded to bf16 into the ring slots that are dead after the
//     MFMA phase, then re-read as 16-byte row pieces: coalesced 128-byte stores, coalesced loads of the
//     auxiliary tensor for the fused ReLU/BatchNorm backward, per-thread (fixed channels) statistics that
//     reach memory as one double atomic per channel per workgroup.
#include "stgcn_common.h"

struct ConvT9Args {
    const void* in; const void* w; void* out;
    const float* bias; const float* pro_scale; const float* pro_shift;
    const void* aux; const float* ep_scale; const float* ep_shift; const float* ep_mean; const float* ep_rstd;
    double* stats;
    int B, R, V, Cin, Cout, ntaps, transposed;
    int tiles_per_seg, segs_per_sample;
};

// scheduling pattern for a block of 8 MFMAs and 8 LDS reads: (MFMA, 2 reads) x 4, then 4 MFMAs, so the last read
// has four MFMAs of time before its consumer
#define INTERLEAVE_8() _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); } __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
constexpr int T9_BM = 256;
constexpr int T9_WR = 512;         // ring slots

template <int NCH, int EPI>
__global__ __launch_bounds__(256) void conv_t9_kernel(ConvT9Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool WRES = (NCH == 1);
    constexpr int RS = NCH * 128 + 16;          // ring row stride (bytes)
    constexpr int WS = 144;                     // weight tile row stride
    constexpr int PPR = NCH * 8;                // 16-byte pieces per input row
    constexpr int NPF = T9_BM * PPR / 256;      // prefetch pieces per thread per tile
    char* ring = smem;
    char* wl = smem + (size_t)T9_WR * RS;       // WRES: [ntaps][64][WS] ; else 2 x [64][WS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int V = a.V, R = a.R, Cin = a.Cin, Cout = a.Cout, ntaps = a.ntaps;
    const int half = (ntaps - 1) / 2;
    const int b = blockIdx.x / a.segs_per_sample;
    const int seg = blockIdx.x - b * a.segs_per_sample;
    const int co0 = blockIdx.y * 64;
    const int ntiles_total = (R + T9_BM - 1) / T9_BM;
    const int tile_beg = seg * a.tiles_per_seg;
    const int tile_end = min(ntiles_total, tile_beg + a.tiles_per_seg);
    if (tile_beg >= tile_end) return;

    const uint16_t* inb = (const uint16_t*)a.in + (size_t)b * R * Cin;
    uint16_t* outb = (uint16_t*)a.out + (size_t)b * R * Cout;
    const uint16_t* auxb = (EPI == 1) ? (const uint16_t*)a.aux + (size_t)b * R * Cout : nullptr;

    // staging role: fixed 8-channel group per thread
    const int sseg = tid % PPR;                 // which 16-byte piece of the input row
    const int srow = tid / PPR;                 // first row handled
    constexpr int SROWS = 256 / PPR;            // rows covered per pass of the 256 threads
    float psc[8], psh[8];
    const bool has_pro = a.pro_scale != nullptr;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        psc[j] = has_pro ? a.pro_scale[sseg * 8 + j] : 1.f;
        psh[j] = has_pro ? a.pro_shift[sseg * 8 + j] : 0.f;
    }
    auto transform = [&](i32x4 v) -> i32x4 {
        if (!has_pro) return v;
        float f[8];
        unpack16<bf16>(v, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = fmaxf(fmaf(f[j], psc[j], psh[j]), 0.f);
        return pack16<bf16>(f);
    };
    // load `T9_BM` consecutive input rows starting at sample row p_first into registers (zeros outside [0, R))
    i32x4 pf[NPF];
    auto fetch_rows = [&](int p_first) {
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int p = p_first + srow + i * SROWS;
            pf[i] = zero16();
            if (p >= 0 && p < R) pf[i] = *(const i32x4*)(inb + (size_t)p * Cin + sseg * 8);
        }
    };
    auto commit_rows = [&](int p_first) {
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int p = p_first + srow + i * SROWS;
            i32x4 v = pf[i];
            if (p >= 0 && p < R) v = transform(v);
            *(i32x4*)(ring + (size_t)(p & (T9_WR - 1)) * RS + sseg * 16) = v;
        }
    };

    // weights
    i32x4 wreg[2];
    auto load_w = [&](int step) {               // streamed form: step = chunk * ntaps + tap
        const int chunk = step / ntaps, tap = step - chunk * ntaps;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = tid + 256 * i;
            const int n = p >> 3, sg = p & 7;
            const int co = co0 + n;
            wreg[i] = zero16();
            if (co < Cout) wreg[i] = *(const i32x4*)((const uint16_t*)a.w + ((size_t)tap * Cout + co) * Cin + chunk * 64 + sg * 8);
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = tid + 256 * i;
            *(i32x4*)(wl + (size_t)buf * 64 * WS + (p >> 3) * WS + (p & 7) * 16) = wreg[i];
        }
    };
    if constexpr (WRES) {
        for (int p = tid; p < ntaps * 64 * 8; p += 256) {
            const int sg = p & 7, n = (p >> 3) & 63, tap = p >> 9;
            const int co = co0 + n;
            i32x4 v = zero16();
            if (co < Cout) v = *(const i32x4*)((const uint16_t*)a.w + ((size_t)tap * Cout + co) * Cin + sg * 8);
            *(i32x4*)(wl + ((size_t)tap * 64 + n) * WS + sg * 16) = v;
        }
    }

    // initial window: rows [p0 - half*V, p0 + BM + half*V)  (at most 512)
    int p0 = tile_beg * T9_BM;
    {
        const int first = p0 - half * V;
        const int nrows = T9_BM + 2 * half * V;
        for (int base = 0; base < nrows; base += T9_BM) {
            fetch_rows(first + base);
            // rows past the window end would alias live slots only if nrows > 512 (excluded by V <= 32)
#pragma unroll
            for (int i = 0; i < NPF; ++i) {
                const int rel = base + srow + i * SROWS;
                if (rel < nrows) {
                    const int p = first + rel;
                    i32x4 v = pf[i];
                    if (p >= 0 && p < R) v = transform(v);
                    *(i32x4*)(ring + (size_t)(p & (T9_WR - 1)) * RS + sseg * 16) = v;
                }
            }
        }
    }

    // epilogue role: fixed 8 output channels per thread
    const int eseg = tid & 7, erow = tid >> 3;
    float e_sc[8], e_sh[8], e_mu[8], e_rs[8], s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        s1[j] = 0.f; s2[j] = 0.f;
        const int co = co0 + eseg * 8 + j;
        if (EPI == 1 && co < Cout) { e_sc[j] = a.ep_scale[co]; e_sh[j] = a.ep_shift[co]; e_mu[j] = a.ep_mean[co]; e_rs[j] = a.ep_rstd[co]; }
        else { e_sc[j] = e_sh[j] = e_mu[j] = e_rs[j] = 0.f; }
    }
    float biasv[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int co = co0 + n * 32 + l31;
        biasv[n] = (a.bias && co < Cout) ? a.bias[co] : 0.f;
    }

    for (int tile = tile_beg; tile < tile_end; ++tile, p0 += T9_BM) {
        const bool more = tile + 1 < tile_end;
        const int pnew = p0 + T9_BM + half * V;          // first row the NEXT tile adds to the window
        if (more) fetch_rows(pnew);                       // in flight during the MFMA phase
        if constexpr (!WRES) load_w(0);

        f32x16 acc[2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

        const int prow0 = p0 + wave * 64 + l31;           // this lane's row of sub-tile m is prow0 + 32*m
        if constexpr (WRES) {
            lds_barrier();                              // window (and, first time, weights) visible
            // fragments of one half tap (two k-steps): a0/a1 = the two row sub-tiles, b0/b1 = the two column tiles
            struct Half { i32x4 a0[2], a1[2], b0[2], b1[2]; };
            auto load_half = [&](Half& f, int tap, int hs) {
                const int sh = (a.transposed ? (half - tap) : (tap - half)) * V;
                const char* a0 = ring + (size_t)((prow0 + sh) & (T9_WR - 1)) * RS + h * 16 + hs * 64;
                const char* a1 = ring + (size_t)((prow0 + 32 + sh) & (T9_WR - 1)) * RS + h * 16 + hs * 64;
                const char* wb = wl + ((size_t)tap * 64 + l31) * WS + h * 16 + hs * 64;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    f.a0[s] = *(const i32x4*)(a0 + s * 32);
                    f.a1[s] = *(const i32x4*)(a1 + s * 32);
                    f.b0[s] = *(const i32x4*)(wb + s * 32);
                    f.b1[s] = *(const i32x4*)(wb + 32 * WS + s * 32);
                }
            };
            auto mma_half = [&](const Half& f) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    mma_step<bf16>(acc[0][0], f.a0[s], f.b0[s]);
                    mma_step<bf16>(acc[0][1], f.a0[s], f.b1[s]);
                    mma_step<bf16>(acc[1][0], f.a1[s], f.b0[s]);
                    mma_step<bf16>(acc[1][1], f.a1[s], f.b1[s]);
                }
            };
            // software pipeline: the next half's LDS reads are in flight while this half's 8 MFMAs run
            Half fa, fb;
            load_half(fa, 0, 0);
            for (int tap = 0; tap < ntaps; ++tap) {
                // the reads of the next half ride in the shadows of this half's MFMAs
                __builtin_amdgcn_sched_barrier(0);
                load_half(fb, tap, 1);
                mma_half(fa);
                INTERLEAVE_8()
                __builtin_amdgcn_sched_barrier(0);
                load_half(fa, tap + 1 < ntaps ? tap + 1 : tap, 0);
                mma_half(fb);
                INTERLEAVE_8()
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            const int nsteps = NCH * ntaps;
            for (int step = 0; step < nsteps; ++step) {
                const int chunk = step / ntaps, tap = step - chunk * ntaps;
                store_w(step & 1);
                lds_barrier();
                if (step + 1 < nsteps) load_w(step + 1);
                const int sh = (a.transposed ? (half - tap) : (tap - half)) * V;
                const char* a0 = ring + (size_t)((prow0 + sh) & (T9_WR - 1)) * RS + chunk * 128 + h * 16;
                const char* a1 = ring + (size_t)((prow0 + 32 + sh) & (T9_WR - 1)) * RS + chunk * 128 + h * 16;
                const char* wb = wl + (size_t)(step & 1) * 64 * WS + l31 * WS + h * 16;
                // two halves of two k-steps: the second half's reads are in flight under the first half's MFMAs
                i32x4 fa0[2], fa1[2], fb0[2], fb1[2], ga0[2], ga1[2], gb0[2], gb1[2];
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    fa0[s] = *(const i32x4*)(a0 + s * 32);
                    fa1[s] = *(const i32x4*)(a1 + s * 32);
                    fb0[s] = *(const i32x4*)(wb + s * 32);
                    fb1[s] = *(const i32x4*)(wb + 32 * WS + s * 32);
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    ga0[s] = *(const i32x4*)(a0 + 64 + s * 32);
                    ga1[s] = *(const i32x4*)(a1 + 64 + s * 32);
                    gb0[s] = *(const i32x4*)(wb + 64 + s * 32);
                    gb1[s] = *(const i32x4*)(wb + 32 * WS + 64 + s * 32);
                }
                __builtin_amdgcn_sched_barrier(0);      // keep all 16 reads ahead of the MFMAs
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    mma_step<bf16>(acc[0][0], fa0[s], fb0[s]);
                    mma_step<bf16>(acc[0][1], fa0[s], fb1[s]);
                    mma_step<bf16>(acc[1][0], fa1[s], fb0[s]);
                    mma_step<bf16>(acc[1][1], fa1[s], fb1[s]);
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    mma_step<bf16>(acc[0][0], ga0[s], gb0[s]);
                    mma_step<bf16>(acc[0][1], ga0[s], gb1[s]);
                    mma_step<bf16>(acc[1][0], ga1[s], gb0[s]);
                    mma_step<bf16>(acc[1][1], ga1[s], gb1[s]);
                }
            }
        }
        lds_barrier();                                  // every wave is done reading the window

        // accumulators (+bias) -> bf16 -> dead ring slots (those of the rows the next tile brings in)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = wave * 64 + m * 32 + acc_row(r, h);          // row within the tile
                    const int slot = (pnew + rl) & (T9_WR - 1);
                    const uint32_t pk = pack_bf16x2(acc[m][n][r] + biasv[n], 0.f);
                    *(uint16_t*)(ring + (size_t)slot * RS + (n * 32 + l31) * 2) = (uint16_t)(pk & 0xFFFFu);
                }
        lds_barrier();

        // 16-byte row pieces: 256 rows x 8 pieces, 8 per thread
        {
            i32x4 av[8];
            if constexpr (EPI == 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = p0 + erow + 32 * i;
                    av[i] = zero16();
                    if (row < R && co0 + eseg * 8 < Cout) av[i] = *(const i32x4*)(auxb + (size_t)row * Cout + co0 + eseg * 8);
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int rl = erow + 32 * i;
                const int row = p0 + rl;
                if (row < R && co0 + eseg * 8 < Cout) {
                    const int slot = (pnew + rl) & (T9_WR - 1);
                    i32x4 v = *(const i32x4*)(ring + (size_t)slot * RS + eseg * 16);
                    float f[8];
                    unpack16<bf16>(v, f);
                    if constexpr (EPI == 1) {
                        float y[8];
                        unpack16<bf16>(av[i], y);
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            f[j] = (fmaf(y[j], e_sc[j], e_sh[j]) > 0.f) ? f[j] : 0.f;
                            s1[j] += f[j];
                            s2[j] += f[j] * ((y[j] - e_mu[j]) * e_rs[j]);
                        }
                        v = pack16<bf16>(f);
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) { s1[j] += f[j]; s2[j] += f[j] * f[j]; }
                    }
                    *(i32x4*)(outb + (size_t)row * Cout + co0 + eseg * 8) = v;
                }
            }
        }
        lds_barrier();                                  // staging slots free again
        if (more) commit_rows(pnew);
        // the __syncthreads at the top of the next tile's MFMA phase publishes the new rows
    }

    if (a.stats) {
        // per-thread sums (fixed 8 channels) -> LDS -> one double atomic per channel per workgroup
        __syncthreads();
        float* red = (float*)smem;                        // [256][16]
#pragma unroll
        for (int j = 0; j < 8; ++j) { red[tid * 16 + j] = s1[j]; red[tid * 16 + 8 + j] = s2[j]; }
        __syncthreads();
        if (tid < 128) {
            const int which = tid >> 6, c = tid & 63;     // which: 0 = s1, 1 = s2 ; c = channel within the tile
            const int sg = c >> 3, j = c & 7;
            float t = 0.f;
            for (int r = 0; r < 32; ++r) t += red[(r * 8 + sg) * 16 + which * 8 + j];
            if (co0 + c < Cout) atomicAdd(&a.stats[which * Cout + co0 + c], (double)t);
        }
    }
}

template <int NCH, int EPI>
static int launch_conv_t9(ConvT9Args& a, hipStream_t stream) {
    constexpr int RS = NCH * 128 + 16;
    const size_t wbytes = (NCH == 1) ? (size_t)a.ntaps * 64 * 144 : (size_t)2 * 64 * 144;
    const size_t lds = (size_t)T9_WR * RS + wbytes;
    if (lds > 160 * 1024) return STGCN_ERR_ARG;
    const int ntiles = ceil_div_i(a.R, T9_BM);
    const int gy = ceil_div_i(a.Cout, 64);
    // one workgroup per CU is resident (LDS): aim at one round of 256 workgroups, so the per-workgroup start-up
    // (weight load, first window) is paid once per CU
    int segs = (256 + a.B * gy - 1) / (a.B * gy);
    if (segs < 1) segs = 1;
    if (segs > ntiles) segs = ntiles;
    int tps = ceil_div_i(ntiles, segs);
    if (tps < 4 && ntiles >= 4) tps = 4;
    segs = ceil_div_i(ntiles, tps);
    a.tiles_per_seg = tps;
    a.segs_per_sample = segs;
    auto kern = conv_t9_kernel<NCH, EPI>;
    static bool set = false;
    if (!set) { if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return STGCN_ERR_LAUNCH; set = true; }
    hipLaunchKernelGGL(kern, dim3((unsigned)(a.B * segs), (unsigned)gy), dim3(256), lds, stream, a);
    return hipGetLastError() == hipSuccess ? STGCN_OK : STGCN_ERR_LAUNCH;
}

// Returns -1 when the shape is outside this kernel's domain (the caller then uses conv_rows).
int conv_t9_try(const void* in, const void* w, void* out, const float* bias, const float* pro_scale, const float* pro_shift,
                int epilogue, const void* aux, const float* ep_scale, const float* ep_shift, const float* ep_mean, const float* ep_rstd,
                double* stats, int B, int T, int V, int Cin, int Cout, int ntaps, int transposed, hipStream_t stream) {
    if (!(Cin == 64 || Cin == 128) || V > 32 || V < 1 || ntaps > 9 || (ntaps & 1) == 0 || Cout % 8 != 0) return -1;
    ConvT9Args a;
    a.in = in; a.w = w; a.out = out; a.bias = bias; a.pro_scale = pro_scale; a.pro_shift = pro_shift;
    a.aux = aux; a.ep_scale = ep_scale; a.ep_shift = ep_shift; a.ep_mean = ep_mean; a.ep_rstd = ep_rstd; a.stats = stats;
    a.B = B; a.R = T * V; a.V = V; a.Cin = Cin; a.Cout = Cout; a.ntaps = ntaps; a.transposed = transposed;
    a.tiles_per_seg = 0; a.segs_per_sample = 0;
    if (Cin == 64) return epilogue == 1 ? launch_conv_t9<1, 1>(a, stream) : launch_conv_t9<1, 0>(a, stream);
    return epilogue == 1 ? launch_conv_t9<2, 1>(a, stream) : launch_conv_t9<2, 0>(a, stream);
}
