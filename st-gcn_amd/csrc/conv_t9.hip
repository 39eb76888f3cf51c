// conv_t9: the stride-1 temporal convolution (any odd tap count up to 9, pad = (taps-1)/2) of a bf16
// channels-last activation with 64 input channels and at most 64 output channels per workgroup, as a
// persistent LDS-ring kernel.
//
// Same arithmetic as conv_rows' forward / stride-1 transposed form (see conv_rows.hip); this is the
// fast path for the 64-channel st_gcn blocks, whose temporal convolutions are the most memory-bound.
//
//   - A workgroup (8 waves, two per SIMD) owns 64 output channels and a contiguous run of 256-row tiles of ONE
//     sample and walks them in order.  Input rows live in an LDS ring of 512 row slots (slot = row & 511; a tile
//     needs rows [p0 - 4V, p0 + 256 + 4V), V <= 32), so every input row is fetched from HBM once per
//     workgroup: when the walk advances, only the 256 new rows are loaded.  They are fetched into
//     registers BEFORE the tile's MFMA phase and written to the ring (BatchNorm+ReLU prologue applied
//     once per element) after it, so the fetch latency hides under the MFMAs.
//   - Rows outside the sample are stored as zeros: a tap shift is a uniform row shift of (tap-4)*V, and
//     the temporal zero padding falls out of the zero rows.  No per-tap validity logic.
//   - All taps' weights are resident in LDS for the whole walk (read from L2 once per workgroup).
//   - Each wave computes 32 rows x 64 channels (1x2 accumulator tiles); the LDS reads of the next half tap are
//     slotted behind the current half's MFMAs.
//   - Epilogue: accumulators (+bias) are rounded to bf16 into the ring slots that are dead after the
//     MFMA phase, then re-read as 16-byte row pieces: coalesced 128-byte stores, coalesced loads of the
//     auxiliary tensor for the fused ReLU/BatchNorm backward, per-thread (fixed channels) statistics that
//     reach memory as one double atomic per channel per workgroup.
//   - Barriers order LDS traffic only (lds_barrier), so the register prefetch and the epilogue stores stay in
//     flight across them.
#include "stgcn_common.h"

struct ConvT9Args {
    const void* in; const void* w; void* out;
    const float* bias; const float* pro_scale; const float* pro_shift; BnPrologue pbn;
    const void* aux; const float* ep_scale; const float* ep_shift; const float* ep_mean; const float* ep_rstd;
    double* stats;
    int B, R, V, Cout, ntaps, transposed;
    int tiles_per_seg, segs_per_sample;
};

// scheduling pattern for a half tap: (MFMA, 2 LDS reads) x 3, then the last MFMA
#define T9_INTERLEAVE() _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); } __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
constexpr int T9_BM = 256;         // rows per tile
constexpr int T9_WR = 512;         // ring slots
constexpr int T9_NT = 512;         // threads per workgroup
constexpr int T9_CIN = 64;
constexpr int T9_RS = T9_CIN * 2 + 16;   // ring / weight row stride (bytes)

template <int EPI>
__global__ __launch_bounds__(T9_NT) void conv_t9_kernel(ConvT9Args a) {
    bn_prologue_preamble(a.pbn);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RS = T9_RS;
    constexpr int NPF = T9_BM * 8 / T9_NT;      // 16-byte pieces per thread per tile (input rows, output rows)
    constexpr int SROWS = T9_NT / 8;            // rows covered per pass of the workgroup
    char* ring = smem;
    char* wl = smem + (size_t)T9_WR * RS;       // [ntaps][64][RS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int V = a.V, R = a.R, Cout = a.Cout, ntaps = a.ntaps;
    constexpr int Cin = T9_CIN;
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
    const uint16_t* auxb = (EPI != 0 && a.aux) ? (const uint16_t*)a.aux + (size_t)b * R * Cout : nullptr;

    // staging / epilogue role: fixed 8-channel group per thread
    const int sseg = tid & 7, srow = tid >> 3;
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
    // load T9_BM consecutive input rows starting at sample row p_first into registers (zeros outside [0, R))
    i32x4 pf[NPF];
    auto fetch_rows = [&](int p_first) {
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int p = p_first + srow + i * SROWS;
            pf[i] = zero16();
            if (p >= 0 && p < R) pf[i] = *(const i32x4*)(inb + (size_t)p * Cin + sseg * 8);
        }
    };
    auto commit_rows = [&](int p_first, int nrows) {
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int rel = srow + i * SROWS;
            const int p = p_first + rel;
            if (rel < nrows) {
                i32x4 v = pf[i];
                if (p >= 0 && p < R) v = transform(v);
                *(i32x4*)(ring + (size_t)(p & (T9_WR - 1)) * RS + sseg * 16) = v;
            }
        }
    };

    // weights: all taps resident
    for (int p = tid; p < ntaps * 64 * 8; p += T9_NT) {
        const int sg = p & 7, n = (p >> 3) & 63, tap = p >> 9;
        const int co = co0 + n;
        i32x4 v = zero16();
        if (co < Cout) v = *(const i32x4*)((const uint16_t*)a.w + ((size_t)tap * Cout + co) * Cin + sg * 8);
        *(i32x4*)(wl + ((size_t)tap * 64 + n) * RS + sg * 16) = v;
    }

    // initial window: rows [p0 - half*V, p0 + BM + half*V)  (at most 512: V <= 32)
    int p0 = tile_beg * T9_BM;
    {
        const int first = p0 - half * V;
        const int nrows = T9_BM + 2 * half * V;
        for (int base = 0; base < nrows; base += T9_BM) {
            fetch_rows(first + base);
            commit_rows(first + base, nrows - base);
        }
    }

    float e_sc[8], e_sh[8], s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        s1[j] = 0.f; s2[j] = 0.f;
        const int co = co0 + sseg * 8 + j;
        if (EPI == 1 && co < Cout) { e_sc[j] = a.ep_scale[co]; e_sh[j] = a.ep_shift[co]; }
        else { e_sc[j] = e_sh[j] = 0.f; }
    }
    float biasv[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int co = co0 + n * 32 + l31;
        biasv[n] = (a.bias && co < Cout) ? a.bias[co] : 0.f;
    }
    const bool cok = co0 + sseg * 8 < Cout;

    for (int tile = tile_beg; tile < tile_end; ++tile, p0 += T9_BM) {
        const bool more = tile + 1 < tile_end;
        const int pnew = p0 + T9_BM + half * V;          // first row the NEXT tile adds to the window
        if (more) fetch_rows(pnew);                       // in flight during the MFMA phase
        i32x4 av[NPF];                                    // auxiliary rows of this tile, fetched ahead of their use
        if constexpr (EPI != 0) {
#pragma unroll
            for (int i = 0; i < NPF; ++i) {
                const int row = p0 + srow + SROWS * i;
                av[i] = zero16();
                if (row < R && cok && auxb) av[i] = *(const i32x4*)(auxb + (size_t)row * Cout + co0 + sseg * 8);
            }
        }

        f32x16 acc[2];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

        const int prow0 = p0 + wave * 32 + l31;           // this lane's row
        lds_barrier();                                    // window (and, first time, weights) visible
        {
            // fragments of one half tap (two k-steps): a = the wave's 32 rows, b0/b1 = the two column tiles
            struct Half { i32x4 a[2], b0[2], b1[2]; };
            auto load_half = [&](Half& f, int tap, int hs) {
                const int sh = (a.transposed ? (half - tap) : (tap - half)) * V;
                const char* ap = ring + (size_t)((prow0 + sh) & (T9_WR - 1)) * RS + h * 16 + hs * 64;
                const char* wb = wl + ((size_t)tap * 64 + l31) * RS + h * 16 + hs * 64;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    f.a[s] = *(const i32x4*)(ap + s * 32);
                    f.b0[s] = *(const i32x4*)(wb + s * 32);
                    f.b1[s] = *(const i32x4*)(wb + 32 * RS + s * 32);
                }
            };
            auto mma_half = [&](const Half& f) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    mma_step<bf16>(acc[0], f.a[s], f.b0[s]);
                    mma_step<bf16>(acc[1], f.a[s], f.b1[s]);
                }
            };
            // software pipeline: the next half's 6 LDS reads are in flight while this half's 4 MFMAs run
            Half fa, fb;
            load_half(fa, 0, 0);
            for (int tap = 0; tap < ntaps; ++tap) {
                __builtin_amdgcn_sched_barrier(0);
                load_half(fb, tap, 1);
                mma_half(fa);
                T9_INTERLEAVE()
                __builtin_amdgcn_sched_barrier(0);
                load_half(fa, tap + 1 < ntaps ? tap + 1 : tap, 0);
                mma_half(fb);
                T9_INTERLEAVE()
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        lds_barrier();                                    // every wave is done reading the window
        // The one wait for the rows fetched at the top of the tile, HERE, a whole MFMA phase after their issue and
        // before the epilogue's stores: the commit at the end of the tile then needs none, and a wait placed there
        // by the compiler would also be a wait for the stores issued just before it (waits retire in issue order).
#pragma unroll
        for (int i = 0; i < NPF; ++i) asm volatile("" : "+v"(pf[i]));
        if constexpr (EPI != 0) {                         // the auxiliary rows as well: one wait, not one per row behind the row before's store
#pragma unroll
            for (int i = 0; i < NPF; ++i) asm volatile("" : "+v"(av[i]));
        }

        // accumulators (+bias) -> bf16 -> dead ring slots (those of the rows the next tile brings in)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = wave * 32 + acc_row(r, h);                  // row within the tile
                const int slot = (pnew + rl) & (T9_WR - 1);
                const uint32_t pk = pack_bf16x2(acc[n][r] + biasv[n], 0.f);
                *(uint16_t*)(ring + (size_t)slot * RS + (n * 32 + l31) * 2) = (uint16_t)(pk & 0xFFFFu);
            }
        lds_barrier();

        // 16-byte row pieces: 256 rows x 8 pieces
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int rl = srow + SROWS * i;
            const int row = p0 + rl;
            if (row < R && cok) {
                const int slot = (pnew + rl) & (T9_WR - 1);
                i32x4 v = *(const i32x4*)(ring + (size_t)slot * RS + sseg * 16);
                float f[8];
                unpack16<bf16>(v, f);
                if constexpr (EPI == 1) {
                    float y[8];
                    unpack16<bf16>(av[i], y);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        f[j] = (fmaf(y[j], e_sc[j], e_sh[j]) > 0.f) ? f[j] : 0.f;
                        s1[j] += f[j];
                        s2[j] = fmaf(f[j], y[j], s2[j]);        // raw moment; centred once at the end of the kernel
                    }
                    v = pack16<bf16>(f);
                } else if constexpr (EPI == 2) {                  // + residual, ReLU (BatchNorm folded into weights and bias)
                    float y[8];
                    unpack16<bf16>(av[i], y);
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j] + y[j], 0.f);
                    v = pack16<bf16>(f);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { s1[j] += f[j]; s2[j] += f[j] * f[j]; }
                }
                *(i32x4*)(outb + (size_t)row * Cout + co0 + sseg * 8) = v;
            }
        }
        lds_barrier();                                    // staging slots free again
        if (more) commit_rows(pnew, T9_BM);
        // the barrier at the top of the next tile's MFMA phase publishes the new rows
    }

    if (a.stats) {
        // per-thread sums (fixed 8 channels) -> LDS -> one double atomic per channel per workgroup
        __syncthreads();
        float* red = (float*)smem;                        // [512][16]
#pragma unroll
        for (int j = 0; j < 8; ++j) { red[tid * 16 + j] = s1[j]; red[tid * 16 + 8 + j] = s2[j]; }
        __syncthreads();
        if (tid < 128) {
            const int which = tid >> 6, c = tid & 63;     // which: 0 = s1, 1 = s2 ; c = channel within the tile
            const int sg = c >> 3, j = c & 7;
            float t = 0.f, t1 = 0.f;
            for (int r = 0; r < SROWS; ++r) t += red[(r * 8 + sg) * 16 + which * 8 + j];
            double tv = (double)t;
            if (EPI == 1 && which == 1 && co0 + c < Cout) {
                // the epilogue gathered sum g*y; the statistic is sum g*yhat = rstd * (sum g*y - mean * sum g)
                for (int r = 0; r < SROWS; ++r) t1 += red[(r * 8 + sg) * 16 + j];
                tv = (double)a.ep_rstd[co0 + c] * (tv - (double)a.ep_mean[co0 + c] * (double)t1);
            }
            if (co0 + c < Cout) atomicAdd(&a.stats[which * Cout + co0 + c], tv);
        }
    }
}

template <int EPI>
static int launch_conv_t9(ConvT9Args& a, hipStream_t stream) {
    const size_t lds = (size_t)T9_WR * T9_RS + (size_t)a.ntaps * 64 * T9_RS;
    if (lds > 160 * 1024) return STGCN_ERR_ARG;
    const int ntiles = ceil_div_i(a.R, T9_BM);
    const int gy = ceil_div_i(a.Cout, 64);
    // one workgroup per CU is resident (LDS): aim at one round of 256 workgroups, so the per-workgroup start-up
    // (weight load, first window) is paid once per CU
    int segs = (stgcn_num_cus() + a.B * gy - 1) / (a.B * gy);
    if (segs < 1) segs = 1;
    if (segs > ntiles) segs = ntiles;
    int tps = ceil_div_i(ntiles, segs);
    if (tps < 4 && ntiles >= 4) tps = 4;
    segs = ceil_div_i(ntiles, tps);
    a.tiles_per_seg = tps;
    a.segs_per_sample = segs;
    auto kern = conv_t9_kernel<EPI>;
    static std::atomic<uint64_t> lds_ok{0};
    if (stgcn_allow_big_lds((const void*)kern, lds_ok) != STGCN_OK) return STGCN_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3((unsigned)(a.B * segs), (unsigned)gy), dim3(T9_NT), lds, stream, a);
    return hipGetLastError() == hipSuccess ? STGCN_OK : STGCN_ERR_LAUNCH;
}

// Returns -1 when the shape is outside this kernel's domain (the caller then tries conv_big, then conv_rows).
int conv_t9_try(const void* in, const void* w, void* out, const float* bias, const float* pro_scale, const float* pro_shift,
                int epilogue, const void* aux, const float* ep_scale, const float* ep_shift, const float* ep_mean, const float* ep_rstd,
                double* stats, int B, int T, int V, int Cin, int Cout, int ntaps, int transposed, hipStream_t stream, const BnPrologue* pbn) {
    if (Cin != T9_CIN || Cout > 64 || V > 32 || V < 1 || ntaps > 9 || (ntaps & 1) == 0 || Cout % 8 != 0) return -1;
    ConvT9Args a;
    a.in = in; a.w = w; a.out = out; a.bias = bias; a.pro_scale = pro_scale; a.pro_shift = pro_shift; a.pbn = pbn ? *pbn : BnPrologue{};
    a.aux = aux; a.ep_scale = ep_scale; a.ep_shift = ep_shift; a.ep_mean = ep_mean; a.ep_rstd = ep_rstd; a.stats = stats;
    a.B = B; a.R = T * V; a.V = V; a.Cout = Cout; a.ntaps = ntaps; a.transposed = transposed;
    a.tiles_per_seg = 0; a.segs_per_sample = 0;
    return epilogue == 1 ? launch_conv_t9<1>(a, stream) : epilogue == 2 ? launch_conv_t9<2>(a, stream) : launch_conv_t9<0>(a, stream);
}
