// conv_wide: bf16 temporal convolutions with 128 or 256 output channels and an input stride of one frame (stride-1
// forward and data-gradient forms, and the per-parity launches of the stride-2 data gradient), as a persistent tiled
// implicit GEMM whose whole output width belongs to ONE workgroup.
//
// Same arithmetic and the same virtual-frame formulation as conv_big (see conv_big.hip): virtual output frame j of a
// sample is real output frame os * j + opar, effective tap i reads input frame j + foff0 + i * dfoff with the weights
// of tap wk0 + i * dwk.  What differs is the shape of the work:
//
//   - Tile = BM rows x BN channels with BN = Cout: 256 x 256 (8 waves as 4 x 2) or 512 x 128 (8 x 1); a wave owns
//     64 rows x 128 channels.  The input window of a chunk is staged once for every output channel, and a k-step of 32
//     channels reads 4 + 8 fragments for 32 v_mfma_f32_16x16x32_bf16 (3/8 of a fragment per MFMA-equivalent of
//     32x32x16 work, against 1 in conv_big).
//   - Chunks are 32 input channels (64-byte rows in LDS), so TWO windows fit beside the weight tiles: the window of
//     chunk c+1 arrives while the taps of chunk c run.  Without a prologue it arrives by LDS-DMA (global_load_lds: no
//     registers, no VALU, no ds_write); with the BatchNorm+ReLU prologue it passes through registers, fetched at the
//     first tap of a chunk and written two taps later.  No phase of the main loop is without MFMAs.
//   - Rows are unpadded (an LDS-DMA instruction writes 1 KiB contiguously = 16 rows).  A fragment read is
//     ds_read_b128 by lane (r = lane & 15, q = lane >> 4) of slot q of row R + r; with the 16-byte slot s of row r
//     stored at slot s ^ 2 * ((r >> 2) & 1) every 16-lane group of such a read covers all 64 banks once, for any R.
//     The swizzle is applied on the SOURCE address of the DMA / on the ds_write address.
//   - The product is formed transposed, D'[channel][row] = W frag x X frag: a lane's four accumulator registers are
//     four consecutive CHANNELS of one row, which leave through a wave-private LDS patch as 8-byte writes (a quarter of
//     the 2-byte writes the untransposed layout needs) and go to memory as 16-byte pieces of whole 256-byte rows.
//   - Weight tiles (BN x 32 channels of one tap = 16 or 8 KiB) cycle through four LDS buffers: tile g+3 is loaded into
//     registers in step g-1, written to LDS in step g, published by the barriers of steps g+1 and g+2; its first four
//     fragments are read during the second half of step g+2, the other four during the first half of step g+3, each
//     into the registers whose MFMAs have just been issued.  One barrier per step (32 MFMAs per wave).
#include "stgcn_common.h"
#include <stdlib.h>
#include <stdio.h>

struct ConvWideArgs {
    const void* in; const void* w; void* out;
    const float* bias; const float* pro_scale; const float* pro_shift; BnPrologue pbn;
    const void* aux; const float* ep_scale; const float* ep_shift; const float* ep_mean; const float* ep_rstd;
    double* stats;
    int B, Tin, V, Cin, Cout;
    int Tv, Tout_real;
    int os, opar, ntaps;
    int foff0, dfoff, wk0, dwk;
    int foff_min, foff_max;
    int tiles_per_sample;
    long long* dbg;      // CWD_STAMP builds only: per-wave cycle sums of workgroup 0
};

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

__device__ __forceinline__ f32x4 mma16(const i32x4& a, const i32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// Workgroup barrier over LDS traffic, as lds_barrier(), that also tells the compiler's wait-count bookkeeping that no LDS
// read is outstanding (it does not look inside inline asm: without the builtin it closes every later wait for a
// fragment with lgkmcnt(0), i.e. also waits for the reads issued just before).
__device__ __forceinline__ void lds_barrier_known() {
    __builtin_amdgcn_s_waitcnt(0xC07F);                 // lgkmcnt(0), every other counter left alone
    asm volatile("s_barrier" ::: "memory");
}

// One LDS-DMA load: every lane's 16 bytes at gsrc go to LDS at lds_dst + 16 * lane (lds_dst wave-uniform).  Issued from
// inline asm on purpose: the compiler models the builtin form (a FLAT instruction with a global and an LDS memory
// operand) as "may complete out of order with everything" and from then on closes each of its own waits -- for a
// fragment read, for a register load -- with a zero count, i.e. waits for whatever was issued last.  What it cannot see
// it does not wait for; the kernel counts these loads itself (vmcnt before the barrier that publishes them).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

#ifdef CWD_STAMP
// stamps sit next to barriers only: s_memtime returns through the scalar cache, and a pending scalar load turns every
// later LDS wait into a wait for everything
#define CWD_T(x) long long x; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(x) :: "memory")
#define CWD_TB(x) CWD_T(x)      /* never leave the scalar load in flight: its late write lands in whatever the register holds by then */
#define CWD_ACC(i, v) dbgv[i] += (v)
#else
#define CWD_T(x)
#define CWD_TB(x)
#define CWD_ACC(i, v)
#endif

constexpr int CWD_ROW = 64;            // bytes of an LDS row: one 32-channel chunk
constexpr int CWD_STG = 272;           // staging row: 128 channels + 16 bytes
constexpr int CWD_NWT = 4;             // weight-tile buffers

template <int WR, int WC, int EPI, bool PRO>
__global__ __launch_bounds__(512) void conv_wide_kernel(ConvWideArgs a) {
    bn_prologue_preamble(a.pbn);
    static_assert(WR * WC == 8, "eight waves");
    constexpr int BM = WR * 64, BN = WC * 128;
    constexpr int WIN_ROWS = WR == 4 ? 576 : 832;
    constexpr int WIN_BYTES = WIN_ROWS * CWD_ROW;
    constexpr int NWP = WIN_ROWS / 16;                   // 1-KiB pieces of a window
    constexpr int WPW = (NWP + 7) / 8;                   // window pieces per wave (LDS-DMA form)
    constexpr int NRP = (WIN_ROWS * 4 + 511) / 512;      // 16-byte pieces per thread (register form)
    constexpr int WT_BYTES = BN * CWD_ROW;
    constexpr int WTP = BN / 16 / 8;                     // weight pieces per wave per tile
    constexpr int WT0 = 2 * WIN_BYTES;
    constexpr int ZOFF = WT0 + CWD_NWT * WT_BYTES;       // the all-zero row
    constexpr int STAT0 = ZOFF + CWD_ROW;                // [8 waves][2][128] floats
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q = lane >> 4;
    const int wr = wave % WR, wc = wave / WR;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int V = a.V, Cin = a.Cin, Cout = a.Cout, Tin = a.Tin, ntaps = a.ntaps;
    const int co0 = blockIdx.y * BN;
    const int Rv = a.Tv * V, Rreal = a.Tout_real * V;
    const int nchunks = Cin / 32;
    const int ntile_total = a.B * a.tiles_per_sample;
    float* stat = (float*)(smem + STAT0) + wave * 256;

#ifdef CWD_STAMP
    long long dbgv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    CWD_T(t_kernel0);
#endif
    if (tid < CWD_ROW / 16) *(i32x4*)(smem + ZOFF + tid * 16) = zero16();
    for (int i = lane; i < 256; i += 64) stat[i] = 0.f;

    // ---- per-lane constants of the loads ------------------------------------------------------------------------
    const int swz_l = ((lane >> 4) & 1) << 1;            // slot flip of a DMA lane's row: row = 16 * piece + (lane >> 2)
    // weight tile: piece p = rows 16 p .. 16 p + 15 of the tile; lane -> row (lane >> 2), physical slot (lane & 3).  The
    // lane's part of the address is one 32-bit offset; everything else is uniform and rides in the scalar base.
    const unsigned wl = (unsigned)((lane >> 2) * Cin * 2 + (((lane & 3) ^ swz_l) << 4));
    // Weight tiles travel through registers: an LDS-DMA instruction costs the issuing wave 60-180 cycles (measured: this
    // kernel without its weight DMA, two instructions per wave and step, ran 16 % shorter), a global_load_dwordx4 and a
    // ds_write_b128 a fraction of that.  wreg holds the wave's share of ONE tile: loaded in step g-1, written to LDS in
    // step g (same lane-linear 1-KiB pieces as a DMA would write), then reloaded.
    i32x4 wreg[WTP];                                     // the tile loaded one step ago (a second set, for two steps of flight,
                                                         // measured no faster and costs eight registers)
    int wi_chunk = 0, wi_tap = 0;                        // the next weight tile to request (the sequence wraps per tile)
    auto load_tile = [&](i32x4 (&dst)[WTP]) {
        const char* src = (const char*)a.w + ((size_t)(a.wk0 + wi_tap * a.dwk) * Cout * Cin + (size_t)wi_chunk * 32 +
                                              (size_t)(co0 + wave * WTP * 16) * Cin) * 2;
#pragma unroll
        for (int i = 0; i < WTP; ++i) dst[i] = *(const i32x4*)(src + (size_t)i * 16 * Cin * 2 + wl);
        if (++wi_tap == ntaps) { wi_tap = 0; if (++wi_chunk == nchunks) wi_chunk = 0; }
    };
    auto load_w_next = [&]() { load_tile(wreg); };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int i = 0; i < WTP; ++i) *(i32x4*)(smem + WT0 + buf * WT_BYTES + (wave * WTP + i) * 1024 + lane * 16) = wreg[i];
    };
    // window, LDS-DMA form: wave w brings pieces w, w + 8, ... (clamped: the count per wave is fixed, see the waits)
    auto dma_win = [&](int b, int flo, int wrows, int chunk, int buf) {
        const char* base = (const char*)a.in + (((size_t)b * Tin + flo) * V * (size_t)Cin + (size_t)chunk * 32 + (((lane & 3) ^ swz_l) << 3)) * 2;
        const int last = max(wrows - 1, 0);
#pragma unroll
        for (int j = 0; j < WPW; ++j) {
            const int piece = min(wave + 8 * j, NWP - 1);
            const int row = min(piece * 16 + (lane >> 2), last);
            glds16(base + (size_t)row * Cin * 2, lds0 + buf * WIN_BYTES + piece * 1024);
        }
    };
    // window, register form (prologue): thread -> logical slot (tid & 3) of rows (tid >> 2) + 128 i
    i32x4 pf[PRO ? NRP : 1];
    auto fetch_win = [&](int b, int flo, int wrows, int chunk) {
        const char* base = (const char*)a.in + (((size_t)b * Tin + flo) * V * (size_t)Cin + (size_t)chunk * 32 + (tid & 3) * 8) * 2;
        const int last = max(wrows - 1, 0);
#pragma unroll
        for (int i = 0; i < NRP; ++i) {
            const int row = min((tid >> 2) + 128 * i, last);
            pf[i] = *(const i32x4*)(base + (size_t)row * Cin * 2);
        }
    };
    auto commit_win = [&](int chunk, int buf) {
        float psc[8], psh[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { psc[j] = a.pro_scale[chunk * 32 + (tid & 3) * 8 + j]; psh[j] = a.pro_shift[chunk * 32 + (tid & 3) * 8 + j]; }
#pragma unroll
        for (int i = 0; i < NRP; ++i) {
            const int row = (tid >> 2) + 128 * i;
            if (row < WIN_ROWS) {
                float f[8];
                unpack16<bf16>(pf[i], f);
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = fmaxf(fmaf(f[j], psc[j], psh[j]), 0.f);
                *(i32x4*)(smem + buf * WIN_BYTES + row * CWD_ROW + ((((tid & 3) ^ (((row >> 2) & 1) << 1))) << 4)) = pack16<bf16>(f);
            }
        }
    };

    auto tile_geo = [&](int ti, int& b, int& r0, int& r1, int& flo, int& wrows) {
        b = ti / a.tiles_per_sample;
        const int tile = ti - b * a.tiles_per_sample;
        r0 = tile * BM;
        r1 = min(r0 + BM, Rv);
        const int tlo = r0 / V, thi = (r1 - 1) / V;
        const int f0 = max(tlo + a.foff_min, 0), f1 = min(thi + a.foff_max, Tin - 1);
        flo = f0;
        wrows = max(0, (f1 - f0 + 1) * V);
    };

    // XCD-aware tile walk (as conv_big): XCD q takes the q-th eighth of the tile sequence
    const bool xcd_walk = (gridDim.x & 7) == 0 && ntile_total >= (int)gridDim.x;
    const int wq = blockIdx.x & 7, wr_ = blockIdx.x >> 3, wpx = gridDim.x >> 3;
    const int t_hi = xcd_walk ? (int)(((long long)ntile_total * (wq + 1)) >> 3) : ntile_total;
    const int t_step = xcd_walk ? wpx : (int)gridDim.x;
    int ti = xcd_walk ? (int)(((long long)ntile_total * wq) >> 3) + wr_ : (int)blockIdx.x;
    if (ti >= t_hi) return;

    int b, r0, r1, flo, wrows;
    tile_geo(ti, b, r0, r1, flo, wrows);
#ifdef CWD_SETPRIO
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);        // the younger wave of each SIMD loses every arbitration otherwise
#endif
    int cc = 0;                                          // chunks done so far: the current window is buffer cc & 1
    int wcur = 0;                                        // buffer of the current step's weight tile
    if constexpr (PRO) { fetch_win(b, flo, wrows, 0); commit_win(0, 0); }
    else dma_win(b, flo, wrows, 0, 0);
    for (int t = 0; t < 3; ++t) { load_tile(wreg); store_w(t); }
    load_tile(wreg);                                     // tile 3: written in step 0
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // B fragment of column subtile n: row wc * 128 + 16 n + l15 of the weight tile, slot q
    const int boff = (wc * 128 + l15) * CWD_ROW + ((q ^ (((l15 >> 2) & 1) << 1)) << 4);

    for (; ti < t_hi; ti += t_step) {
        CWD_T(t_tile0);
        // this lane's rows (one per 16-row subtile) as window rows for frame offset 0
        int wbase[4];
        int tfr[4];                                      // frame of the row, or a value that fails every range check
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int u = r0 + wr * 64 + m * 16 + l15;
            const bool ok = u < r1;
            const int uu = ok ? u : r0;
            const int t = uu / V;
            tfr[m] = ok ? t : -(1 << 20);
            wbase[m] = (t - flo) * V + (uu - t * V);
        }
        f32x4 acc[4][8];
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (a.bias) {
                const float* bp = a.bias + co0 + wc * 128 + n * 16 + 4 * q;
                bv[0] = bp[0]; bv[1] = bp[1]; bv[2] = bp[2]; bv[3] = bp[3];
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m][n] = bv;
        }
        const int tn = ti + t_step;
        int nb = 0, nr0 = 0, nr1 = 0, nflo = 0, nwrows = 0;
        if (tn < t_hi) tile_geo(tn, nb, nr0, nr1, nflo, nwrows);

        auto a_offsets = [&](int tapi, int (&aoff)[4]) {
            const int fo = a.foff0 + tapi * a.dfoff;
            const int foV = fo * V;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int row = wbase[m] + foV;
                const int off = (row << 6) | ((q ^ ((row >> 1) & 2)) << 4);
                // one unsigned compare covers both ends of the sample (and the rows past the tile, whose frame is far
                // below zero): a select, not a branch
                aoff[m] = (unsigned)(tfr[m] + fo) < (unsigned)Tin ? off : ZOFF;
            }
        };
        // Fragments.  X (rows): am[] is this step's, nam[] the next step's, read during this step's second half.
        // W (channels): bq[0..3] arrive during the previous step's second half, bq[4..7] during this step's first half
        // (each register set is reloaded right after the MFMAs that consumed it: eight fragments, no second set).
        i32x4 am[4], bq[8], nam[4];
#ifdef CWD_AB_READS_UNUSED
        i32x4 dz = zero16(), dz2 = zero16();
#endif
        int aoff[4];
        // first step of the tile: its fragments are read here, behind a barrier of their own
        lds_barrier();
        a_offsets(0, aoff);
#pragma unroll
        for (int m = 0; m < 4; ++m) am[m] = *(const i32x4*)(smem + (cc & 1) * WIN_BYTES + aoff[m]);
#pragma unroll
        for (int n = 0; n < 4; ++n) bq[n] = *(const i32x4*)(smem + WT0 + wcur * WT_BYTES + boff + n * 1024);

        CWD_T(t_steps0);
        CWD_ACC(0, t_steps0 - t_tile0);
        for (int chunk = 0; chunk < nchunks; ++chunk, ++cc) {
            const bool more = chunk + 1 < nchunks || tn < t_hi;        // a window follows this one
            for (int tapi = 0; tapi < ntaps; ++tapi) {
                CWD_TB(t_b0);
#ifndef CWD_AB_NO_BARRIER
                lds_barrier_known();
#endif
                CWD_T(t_b1);
                CWD_ACC(1, t_b1 - t_b0);
                const bool newwin = tapi + 1 == ntaps;
                const int wnext = (wcur + 1) & (CWD_NWT - 1);
                const char* bc = smem + WT0 + wcur * WT_BYTES + boff;
                const char* bn = smem + WT0 + wnext * WT_BYTES + boff;
                const char* wn = smem + ((newwin ? cc + 1 : cc) & 1) * WIN_BYTES;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int n = 0; n < 8; ++n) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc[m][n] = mma16(bq[n], am[m], acc[m][n]);
                    // (the last step of a tile reads ahead like every other: what it fetches is dropped, the next tile's
                    // first step reads its own fragments; every address stays inside the buffers)
#ifdef CWD_AB_READS_UNUSED
                    {   // timing-only: the same reads, results dropped (one register set, consumed once per step)
                        if (n < 4) dz = *(const i32x4*)(bc + (n + 4) * 1024);
                        else if (n < 6) {
#pragma unroll
                            for (int i = 2 * (n - 4); i < 2 * (n - 4) + 2; ++i) { dz = *(const i32x4*)(bn + i * 1024); dz2 = *(const i32x4*)(wn + aoff[i]); }
                        }
                        if (n == 7) asm volatile("" :: "v"(dz), "v"(dz2));
                        if (n >= 4) nam[n - 4] = am[n - 4];
                    }
#elif !defined(CWD_AB_NO_READS)
                    // The next step's fragments are all requested in blocks 4 and 5: what a wave requests in its last
                    // blocks it waits for at the barrier, with its SIMD partner doing the same and the matrix pipe empty.
                    if (n < 4) bq[n + 4] = *(const i32x4*)(bc + (n + 4) * 1024);
                    else if (n < 6) {
#pragma unroll
                        for (int i = 2 * (n - 4); i < 2 * (n - 4) + 2; ++i) {
                            bq[i] = *(const i32x4*)(bn + i * 1024);
                            nam[i] = *(const i32x4*)(wn + aoff[i]);
                        }
                    }
#else
                    if (n >= 4) nam[n - 4] = am[n - 4];
#endif
#ifndef CWD_AB_NO_DMA
                    if (n == 1) {
                        // tile g+3 (in registers since the step before) goes to the buffer of tile g-1, whose last reads
                        // were in the first half of step g-1; then the requests of this step: the next window at the
                        // first tap, the next weight tile always
#ifndef CWD_AB_NO_WSTORE
                        store_w((wcur + 3) & (CWD_NWT - 1));
#endif
#ifndef CWD_AB_NO_WIN
                        if (tapi == 0 && more) {
                            const bool same = chunk + 1 < nchunks;
                            if constexpr (PRO) fetch_win(same ? b : nb, same ? flo : nflo, same ? wrows : nwrows, same ? chunk + 1 : 0);
                            else dma_win(same ? b : nb, same ? flo : nflo, same ? wrows : nwrows, same ? chunk + 1 : 0, (cc + 1) & 1);
                        }
#endif
#ifndef CWD_AB_NO_WLOAD
                        load_w_next();
#endif
                    }
#endif
                    if constexpr (PRO) { if (n == 3 && tapi == 2 && more) commit_win(chunk + 1 < nchunks ? chunk + 1 : 0, (cc + 1) & 1); }
#ifndef CWD_AB_NO_AOFF
                    if (n == 2) {
                        // pure arithmetic floats to the top of the step unless something pins it behind the first MFMAs
                        int tnext = newwin ? 0 : tapi + 1;
                        asm volatile("" : "+v"(tnext));
                        a_offsets(tnext, aoff);
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) am[m] = nam[m];
                wcur = wnext;
                // (no wait here: the store of the weight registers in the next step waits for their load, the youngest
                // request of this step, and with it for the window pieces requested before it)
                CWD_TB(t_w1);
                CWD_ACC(3, t_w1 - t_b1);
                CWD_ACC(6, 1);
            }
        }
        CWD_T(t_e0);
        lds_barrier();                                   // every wave is done with the last window

        // ---- epilogue: wave-private patch (16 rows x 128 channels) in the window buffer just consumed ------------
        {
            char* stg = smem + ((cc - 1) & 1) * WIN_BYTES + wave * (16 * CWD_STG);
            const int piece = lane & 15, prow = lane >> 4;
            const int ch0 = co0 + wc * 128 + piece * 8;
            float e_sc[8], e_sh[8], s1[8], s2[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s1[j] = 0.f; s2[j] = 0.f;
                if (EPI == 1) { e_sc[j] = a.ep_scale[ch0 + j]; e_sh[j] = a.ep_shift[ch0 + j]; }
                else { e_sc[j] = 0.f; e_sh[j] = 0.f; }
            }
            const bf16* auxb = (const bf16*)a.aux + (size_t)b * Rreal * Cout + ch0;
            bf16* outb = (bf16*)a.out + (size_t)b * Rreal * Cout + ch0;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                int rrow[4];
                i32x4 av[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int u = r0 + wr * 64 + m * 16 + i * 4 + prow;
                    int rr = -1;
                    if (u < r1) {
                        if (a.os == 1) rr = u;
                        else { const int t = u / V; rr = (a.os * t + a.opar) * V + (u - t * V); }
                    }
                    rrow[i] = rr;
                    if constexpr (EPI != 0) {
                        av[i] = zero16();
                        if (rr >= 0 && a.aux) av[i] = *(const i32x4*)(auxb + (size_t)rr * Cout);
                    }
                }
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    const f32x4 v = acc[m][n];
                    i32x2 p = {(int)pack_bf16x2(v[0], v[1]), (int)pack_bf16x2(v[2], v[3])};
                    *(i32x2*)(stg + l15 * CWD_STG + (n * 16 + 4 * q) * 2) = p;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if constexpr (EPI != 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(av[i]));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    i32x4 v = *(const i32x4*)(stg + (i * 4 + prow) * CWD_STG + piece * 16);
                    if (rrow[i] >= 0) {
                        float f[8];
                        unpack16<bf16>(v, f);
                        if constexpr (EPI == 1) {
                            float y[8];
                            unpack16<bf16>(av[i], y);
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                f[j] = (fmaf(y[j], e_sc[j], e_sh[j]) > 0.f) ? f[j] : 0.f;
                                s1[j] += f[j];
                                s2[j] = fmaf(f[j], y[j], s2[j]);
                            }
                            v = pack16<bf16>(f);
                        } else if constexpr (EPI == 2) {
                            float y[8];
                            unpack16<bf16>(av[i], y);
#pragma unroll
                            for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j] + y[j], 0.f);
                            v = pack16<bf16>(f);
                        } else {
#pragma unroll
                            for (int j = 0; j < 8; ++j) { s1[j] += f[j]; s2[j] += f[j] * f[j]; }
                        }
                        *(i32x4*)(outb + (size_t)rrow[i] * Cout) = v;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if (EPI != 2 && a.stats) {
                // the four lanes that share a piece index fold their sums in a fixed order; lanes 0-15 add them to
                // the wave's own table (no atomics: one wave, one owner per entry)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    s1[j] += __shfl_xor(s1[j], 16); s1[j] += __shfl_xor(s1[j], 32);
                    s2[j] += __shfl_xor(s2[j], 16); s2[j] += __shfl_xor(s2[j], 32);
                }
                if (prow == 0) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { stat[piece * 8 + j] += s1[j]; stat[128 + piece * 8 + j] += s2[j]; }
                }
            }
        }
        b = nb; r0 = nr0; r1 = nr1; flo = nflo; wrows = nwrows;
        CWD_T(t_e1);
        CWD_ACC(4, t_e1 - t_e0);
    }
#ifdef CWD_STAMP
    CWD_T(t_kernel1);
    dbgv[5] = t_kernel1 - t_kernel0;
    if (a.dbg && blockIdx.x == 0 && lane == 0)
        for (int i = 0; i < 8; ++i) a.dbg[wave * 8 + i] = dbgv[i];
#endif

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the weight tile requested for a step that never came
    if (EPI != 2 && a.stats) {
        __syncthreads();
        if (tid < 2 * BN) {
            const int which = tid / BN, c = tid % BN;     // which: 0 = s1, 1 = s2 ; c = channel within the tile
            const float* tb = (const float*)(smem + STAT0) + (c / 128) * WR * 256 + which * 128 + (c & 127);
            float t = 0.f;
            for (int r = 0; r < WR; ++r) t += tb[r * 256];
            double tv = (double)t;
            if (EPI == 1 && which == 1) {
                // the epilogue gathered sum g*y; the statistic is sum g*yhat = rstd * (sum g*y - mean * sum g)
                float t1 = 0.f;
                for (int r = 0; r < WR; ++r) t1 += tb[r * 256 - 128];
                tv = (double)a.ep_rstd[co0 + c] * (tv - (double)a.ep_mean[co0 + c] * (double)t1);
            }
            atomicAdd(&a.stats[which * Cout + co0 + c], tv);
        }
    }
}

template <int WR, int WC, int EPI, bool PRO>
static int launch_conv_wide(ConvWideArgs& a, hipStream_t stream) {
    constexpr int BM = WR * 64, BN = WC * 128;
    constexpr int WIN_ROWS = WR == 4 ? 576 : 832;
    a.tiles_per_sample = ceil_div_i(a.Tv * a.V, BM);
    const size_t lds = 2 * (size_t)WIN_ROWS * CWD_ROW + CWD_NWT * (size_t)BN * CWD_ROW + CWD_ROW + 8 * 256 * sizeof(float);
    int gx = stgcn_num_cus();
    if (gx > a.B * a.tiles_per_sample) gx = a.B * a.tiles_per_sample;
    auto kern = conv_wide_kernel<WR, WC, EPI, PRO>;
    static std::atomic<uint64_t> lds_ok{0};
    if (stgcn_allow_big_lds((const void*)kern, lds_ok) != STGCN_OK) return STGCN_ERR_LAUNCH;
#ifdef CWD_STAMP
    static long long* dbg = nullptr;
    if (!dbg) { hipMalloc(&dbg, 64 * sizeof(long long)); }
    hipMemsetAsync(dbg, 0, 64 * sizeof(long long), stream);
    a.dbg = dbg;
#endif
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)(a.Cout / BN)), dim3(512), lds, stream, a);
#ifdef CWD_STAMP
    if (getenv("STGCN_CW_DBG")) {
        long long h[64];
        hipStreamSynchronize(stream);
        hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
        for (int w = 0; w < 8; ++w)
            fprintf(stderr, "wave %d: kernel %lld  tile-start %lld  barrier %lld  (unused %lld)  step-after-barrier %lld  epilogue %lld  steps %lld\n",
                    w, h[w * 8 + 5], h[w * 8 + 0], h[w * 8 + 1], h[w * 8 + 2], h[w * 8 + 3], h[w * 8 + 4], h[w * 8 + 6]);
    }
#endif
    return hipGetLastError() == hipSuccess ? STGCN_OK : STGCN_ERR_LAUNCH;
}

template <int WR, int WC>
static int conv_wide_launch_form(ConvWideArgs& a, int epi, bool dry, hipStream_t stream) {
    constexpr int BM = WR * 64;
    constexpr int WIN_ROWS = WR == 4 ? 576 : 832;
    // frames a tile of BM rows can touch, and the rows its window then holds
    const int nf = (BM - 2) / a.V + 2;
    int fr = nf - 1 + (a.foff_max - a.foff_min) + 1;
    if (fr > a.Tin) fr = a.Tin;
    if (fr * a.V > WIN_ROWS) return -1;
    if (dry) return 0;
    const bool pro = a.pro_scale != nullptr;
    if (epi == 1) return pro ? -1 : launch_conv_wide<WR, WC, 1, false>(a, stream);
    if (epi == 2) return pro ? launch_conv_wide<WR, WC, 2, true>(a, stream) : launch_conv_wide<WR, WC, 2, false>(a, stream);
    return pro ? launch_conv_wide<WR, WC, 0, true>(a, stream) : launch_conv_wide<WR, WC, 0, false>(a, stream);
}

static int conv_wide_launch_any(ConvWideArgs& a, int epi, bool dry, hipStream_t stream) {
    a.foff_min = a.dfoff > 0 ? a.foff0 : a.foff0 + (a.ntaps - 1) * a.dfoff;
    a.foff_max = a.dfoff > 0 ? a.foff0 + (a.ntaps - 1) * a.dfoff : a.foff0;
    if (a.Cout == 256) return conv_wide_launch_form<4, 2>(a, epi, dry, stream);
    if (a.Cout == 128) return conv_wide_launch_form<8, 1>(a, epi, dry, stream);
    return -1;
}

// Returns -1 when the shape is outside this kernel's domain (the caller then tries conv_big).
int conv_wide_try(const void* in, const void* w, void* out, const float* bias, const float* pro_scale, const float* pro_shift,
                  int epilogue, const void* aux, const float* ep_scale, const float* ep_shift, const float* ep_mean, const float* ep_rstd,
                  double* stats, int B, int Tin, int Tout, int V, int Cin, int Cout, int ntaps, int stride, int pad, int dil,
                  int transposed, hipStream_t stream, const BnPrologue* pbn) {
    static const bool off = getenv("STGCN_NO_CONV_WIDE") != nullptr;
    if (off) return -1;
    if ((Cout != 128 && Cout != 256) || Cin % 32 != 0 || Cin < 64 || V < 1 || V > 32 || dil != 1) return -1;
    if (epilogue == 1 && pro_scale) return -1;
    // One workgroup per compute unit walks the tiles; with fewer tiles than compute units part of the chip would
    // idle, and conv_big's tiles (a quarter or half the size) fill it better: an 8-clip step's 128 tiles run 1.5x
    // faster there.  STGCN_CONV_WIDE_ANY_GRID (read on every call) lifts the condition: tests run this kernel on small
    // shapes with it.
    {
        const int bm = Cout == 256 ? 256 : 512;
        const int tv = stride == 1 ? Tout : (Tout + 1) / 2;
        const long tiles = (long)B * ceil_div_i(tv * V, bm);
        if (tiles < stgcn_num_cus() && getenv("STGCN_CONV_WIDE_ANY_GRID") == nullptr) return -1;
    }
    ConvWideArgs a;
    a.in = in; a.w = w; a.out = out; a.bias = bias; a.pro_scale = pro_scale; a.pro_shift = pro_shift; a.pbn = pbn ? *pbn : BnPrologue{};
    a.aux = aux; a.ep_scale = ep_scale; a.ep_shift = ep_shift; a.ep_mean = ep_mean; a.ep_rstd = ep_rstd; a.stats = stats;
    a.B = B; a.Tin = Tin; a.V = V; a.Cin = Cin; a.Cout = Cout; a.Tout_real = Tout;
    a.tiles_per_sample = 0;
    a.dbg = nullptr;
    if (stride == 1) {
        if (ntaps < 4 || ntaps > 9) return -1;
        a.os = 1; a.opar = 0; a.Tv = Tout; a.ntaps = ntaps;
        a.foff0 = transposed ? pad : -pad; a.dfoff = transposed ? -1 : 1; a.wk0 = 0; a.dwk = 1;
        return conv_wide_launch_any(a, epilogue, false, stream);
    }
    if (stride != 2 || !transposed) return -1;
    // transposed, stride 2: one launch per output-frame parity, each a stride-1 convolution over the taps of its parity
    for (int pass = 0; pass < 2; ++pass) {               // pass 0 checks that both parities fit, pass 1 launches
        for (int par = 0; par < 2; ++par) {
            a.os = 2; a.opar = par; a.Tv = (Tout - par + 1) / 2;
            if (a.Tv <= 0) continue;
            const int k0 = (par + pad) & 1;               // taps k = k0, k0 + 2, ...: (par + pad - k) even
            a.ntaps = k0 < ntaps ? (ntaps - k0 + 1) / 2 : 0;
            if (a.ntaps < 4) return -1;
            a.foff0 = (par + pad - k0) / 2; a.dfoff = -1; a.wk0 = k0; a.dwk = 2;
            const int rc = conv_wide_launch_any(a, epilogue, pass == 0, stream);
            if (rc != STGCN_OK) return rc;
        }
    }
    return STGCN_OK;
}
