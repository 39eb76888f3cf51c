// conv_big: bf16 temporal convolutions that the LDS-ring kernel (conv_t9) does not take -- 256 input channels,
// stride 2 in either direction, the 1-tap strided residual convolution -- as a persistent tiled implicit GEMM.
//
// Same arithmetic as conv_rows (see conv_rows.hip).  Every supported form is expressed over VIRTUAL output
// frames j of one sample:
//     real output frame = os * j + opar,      input frame of effective tap i = fs * j + foff[i],   weights wk[i]
//   forward, stride s        : fs = s, os = 1,            foff[i] = i - pad,             wk[i] = i
//   transposed, stride 1     : fs = 1, os = 1,            foff[i] = pad - i,             wk[i] = i
//   transposed, stride 2     : one launch per output-frame parity p (os = 2, opar = p, fs = 1) with only the taps k
//                              of matching parity: foff = (p + pad - k) / 2.  No MFMA is spent on the zeros a
//                              strided transposed convolution is full of; a parity without taps writes zeros.
//
//   - A workgroup (8 or 4 waves) owns one 128-channel tile of Cout and walks (sample, row-tile) pairs; a row tile is
//     BM = 256 (or 128) virtual output rows.  Per 64-channel chunk of Cin the window of input rows all taps need is
//     staged once in LDS (BatchNorm+ReLU prologue applied once per element); taps read their A fragments from it at
//     per-lane row offsets (a zero row stands in for frames outside the sample).
//   - GRP (forward, stride 2): the frames a tile's taps touch are twice the tile's own, too many rows for the window
//     at 256-row tiles.  The taps are then taken in two groups by the parity of their frame offset; a group's window
//     holds every second frame (the ones of its parity), and a chunk is two phases: stage group 0, run its taps,
//     stage group 1, run its taps.
//   - The NEXT window (next chunk, or the first chunk of the next tile) is fetched into registers while the current
//     chunk's taps run, so global latency hides under 9 x 16 (or 32) MFMAs per wave; weight tiles (128 x 64 per
//     tap) are double-buffered the same way.
//   - Per wave 64 rows x 64 channels (2 x 2 accumulator tiles); BM = 256 runs 8 waves (two per SIMD, so one wave's
//     LDS and barrier waits hide under the other's MFMAs), BM = 128 runs 4.
//   - Epilogue: accumulators (+bias) are rounded to bf16 into LDS, then leave as 16-byte row pieces: coalesced
//     stores, coalesced loads of the auxiliary tensor of the fused ReLU/BatchNorm backward, per-thread statistics
//     (fixed channels) that reach memory as one double atomic per channel per workgroup.
#include "stgcn_common.h"

struct ConvBigArgs {
    const void* in; const void* w; void* out;
    const float* bias; const float* pro_scale; const float* pro_shift; BnPrologue pbn;
    const void* aux; const float* ep_scale; const float* ep_shift; const float* ep_mean; const float* ep_rstd;
    double* stats;
    int B, Tin, V, Cin, Cout;
    int Tv, Tout_real;
    int fs, os, opar, ntaps;
    int foff[9], wk[9];
    int foff_min, foff_max;
    int tiles_per_sample, zero_row;
    int gt[3], gmin[2], gmax[2];      // GRP: taps [gt[g], gt[g+1]) form group g, its least and largest frame offset
};

// scheduling pattern for one k-step: NR LDS reads paired two-by-two behind the first MFMAs, then the remaining MFMAs
#define CB_INTERLEAVE(NR, NM) _Pragma("unroll") for (int i_ = 0; i_ < (NR) / 2; ++i_) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); } __builtin_amdgcn_sched_group_barrier(0x008, (NM) - (NR) / 2, 0);
constexpr int CB_RS = 144;
constexpr int CB_WIN_ROWS = 544;    // window capacity in rows
constexpr int CB_WRS = 128;         // weight-tile row: one unpadded 128-byte chunk, its 16-byte slots XOR-swizzled
constexpr int CB_NWB = 4;           // weight-tile buffers

template <typename T, int WR, int WC, int MT, int NT, int EPI, bool GRP = false>
__global__ __launch_bounds__(WR * WC * 64) void conv_big_kernel(ConvBigArgs a) {
    bn_prologue_preamble(a.pbn);
    constexpr int NG = GRP ? 2 : 1;                      // tap groups = window phases per chunk
    constexpr int WFS = GRP ? 2 : 1;                     // frame step inside a window
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ES = ET<T>::ES;                        // bytes per element
    constexpr int VEC = ET<T>::VEC;                      // elements per 16-byte piece
    constexpr int CK = 128 / ES;                         // channels per 128-byte chunk of Cin: 64 (bf16) / 32 (f32)
    constexpr int BM = WR * MT * 32;
    constexpr int NTH = WR * WC * 64;                    // threads per workgroup (256 or 512)
    constexpr int RPP = NTH / 8;                         // window rows staged per pass
    constexpr int NPF = (CB_WIN_ROWS + RPP - 1) / RPP;   // window pieces per thread
    constexpr int BN = WC * NT * 32;                     // column tile: 128 or 64 output channels
    constexpr int EPR = BN / VEC;                        // 16-byte pieces per output row of the tile
    constexpr int WPT = BN * 8 / NTH;                    // global->LDS loads (8 rows each) per wave per weight tile
    constexpr int ERP = NTH / EPR;                       // epilogue rows per pass
    constexpr int EPP = BM / ERP;                        // epilogue passes per tile
    constexpr int EG = EPP < 8 ? EPP : 8;                // passes whose auxiliary loads are issued together
    static_assert(BN == 128 || BN == 64, "column tile is 128 or 64");
    constexpr int SRS = BN * ES + 16;                  // epilogue staging row stride
    char* win = smem;
    char* wbuf = smem + (size_t)CB_WIN_ROWS * CB_RS;     // CB_NWB x [BN][CB_WRS]
    const int ZOFF = a.zero_row * CB_RS;                 // byte offset of the all-zero row

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wr = wave % WR, wc = wave / WR;
    const int V = a.V, Cin = a.Cin, Cout = a.Cout, Tin = a.Tin, ntaps = a.ntaps;
    const int co0 = blockIdx.y * BN;
    const int Rv = a.Tv * V, Rreal = a.Tout_real * V;
    const int nchunks = Cin / CK;
    const int ntile_total = a.B * a.tiles_per_sample;

    if (tid < CB_RS / 16) *(i32x4*)(smem + ZOFF + tid * 16) = zero16();

    const int sseg = tid & 7, srow = tid >> 3;           // window staging: RPP rows per pass
    const int eseg = tid % EPR, erow = tid / EPR;          // epilogue: ERP rows per pass, fixed 8 channels
    const bool has_pro = a.pro_scale != nullptr;

    float e_sc[VEC], e_sh[VEC], s1[VEC], s2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        s1[j] = 0.f; s2[j] = 0.f;
        const int co = co0 + eseg * VEC + j;
        if (EPI == 1 && co < Cout) { e_sc[j] = a.ep_scale[co]; e_sh[j] = a.ep_shift[co]; }
        else { e_sc[j] = e_sh[j] = 0.f; }
    }
    float biasv[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int co = co0 + (wc * NT + n) * 32 + l31;
        biasv[n] = (a.bias && co < Cout) ? a.bias[co] : 0.f;
    }

    // tile geometry
    auto tile_geo = [&](int ti, int& b, int& r0, int& r1, int (&flo)[NG], int (&wrows)[NG]) {
        b = ti / a.tiles_per_sample;
        const int tile = ti - b * a.tiles_per_sample;
        r0 = tile * BM;
        r1 = min(r0 + BM, Rv);
        const int tlo = r0 / V, thi = (r1 - 1) / V;
        if constexpr (!GRP) {
            int f0 = a.fs * tlo + a.foff_min, f1 = a.fs * thi + a.foff_max;
            f0 = max(f0, 0);
            f1 = min(f1, Tin - 1);
            flo[0] = f0;
            wrows[0] = max(0, (f1 - f0 + 1) * V);
        } else {
#pragma unroll
            for (int g = 0; g < NG; ++g) {               // clamped to the sample in steps of two frames: the parity stays
                int f0 = a.fs * tlo + a.gmin[g], f1 = a.fs * thi + a.gmax[g];
                if (f0 < 0) f0 += ((1 - f0) >> 1) << 1;
                if (f1 > Tin - 1) f1 -= ((f1 - Tin + 2) >> 1) << 1;
                flo[g] = f0;
                wrows[g] = f1 >= f0 ? (((f1 - f0) >> 1) + 1) * V : 0;
            }
        }
    };
    i32x4 pf[NPF];
    const float inv_v = 1.0f / (float)V;
    auto fetch_win = [&](int b, int flo, int wrows, int chunk) {
        const char* base = (const char*)a.in + (((size_t)b * Tin + flo) * V * (size_t)Cin + (size_t)chunk * CK + sseg * VEC) * ES;
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int row = srow + RPP * i;
            int grow = row;                                // row of the sample, counted from frame flo
            if constexpr (GRP) grow += (int)(((float)row + 0.5f) * inv_v) * V;      // window frame f is sample frame flo + 2 f
            if (row < wrows) pf[i] = *(const i32x4*)(base + (size_t)grow * Cin * ES);
        }
    };
    auto commit_win = [&](int wrows, int chunk) {
        float psc[VEC], psh[VEC];
        if (has_pro) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) { psc[j] = a.pro_scale[chunk * CK + sseg * VEC + j]; psh[j] = a.pro_shift[chunk * CK + sseg * VEC + j]; }
        }
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int row = srow + RPP * i;
            if (row < wrows) {
                i32x4 v = pf[i];
                if (has_pro) {
                    float f[VEC];
                    unpack16<T>(v, f);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) f[j] = fmaxf(fmaf(f[j], psc[j], psh[j]), 0.f);
                    v = pack16<T>(f);
                }
                *(i32x4*)(win + (size_t)row * CB_RS + sseg * 16) = v;
            }
        }
    };
    // Weight tiles (BN rows x one 128-byte chunk of Cin) go from global memory straight into LDS, 8 rows per wave
    // instruction, without passing through registers.  Such a load writes the wave's 64 x 16 bytes contiguously, so
    // rows are unpadded; to keep the fragment reads (32 rows, one 16-byte slot each) off each other's banks, slot q
    // of row r holds the row's piece q ^ ((r >> 1) & 7) -- applied here on the global address of each lane.
    // Columns past Cout repeat row Cout-1: their accumulators are never stored.
    int wg_off[WPT];
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const int row = (wave * WPT + i) * 8 + (lane >> 3);
        const int piece = (lane & 7) ^ ((row >> 1) & 7);
        wg_off[i] = (min(co0 + row, Cout - 1) * Cin + piece * VEC) * ES;
    }
    auto glds_w = [&](int chunk, int tapi, int buf) {
        const char* src = (const char*)a.w + ((size_t)a.wk[tapi] * Cout * Cin + (size_t)chunk * CK) * ES;
#pragma unroll
        for (int i = 0; i < WPT; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + wg_off[i]),
                                             (__attribute__((address_space(3))) void*)(wbuf + (size_t)buf * BN * CB_WRS + (wave * WPT + i) * 8 * CB_WRS),
                                             16, 0, 0);
    };
    // Tile g of the (chunk-major, then tap) sequence -- the same for every row tile of this workgroup, so it just
    // wraps around -- lives in buffer g % 4.  Its loads are issued early in step g-3 (the buffer was last read
    // during step g-4), waited for by each wave at the end of step g-2 (a counted wait that leaves the loads of step
    // g-2 itself in flight), published by the barrier that opens step g-1, and read during step g; its first
    // fragments are already read at the end of step g-1.
    int wi_chunk = 0, wi_tap = 0;                          // iterator of the next tile to fetch
    auto glds_w_next = [&](int buf) {
        glds_w(wi_chunk, wi_tap, buf);
        if (++wi_tap == ntaps) { wi_tap = 0; if (++wi_chunk == nchunks) wi_chunk = 0; }
    };
    // Tile walk.  Workgroups go to the 8 XCDs round-robin by linear id, and each XCD has its own L2: XCD q takes the
    // q-th eighth of the (sample-major) tile sequence and its workgroups walk it side by side, so the tiles in flight
    // on one L2 are neighbours -- their windows overlap by the taps' halo (8 of ~20 frames) and the other column
    // tile of the same rows has the same blockIdx.x.  Grids that are not a multiple of 8 walk the plain stride.
    const bool xcd_walk = (gridDim.x & 7) == 0 && ntile_total >= (int)gridDim.x;
    const int wq = blockIdx.x & 7, wr_ = blockIdx.x >> 3, wpx = gridDim.x >> 3;
    const int t_hi = xcd_walk ? (int)(((long long)ntile_total * (wq + 1)) >> 3) : ntile_total;
    const int t_step = xcd_walk ? wpx : (int)gridDim.x;
    int ti = xcd_walk ? (int)(((long long)ntile_total * wq) >> 3) + wr_ : (int)blockIdx.x;
    if (ti >= t_hi) return;
    int b, r0, r1, flo[NG], wrows[NG];
    tile_geo(ti, b, r0, r1, flo, wrows);
    if (ntaps > 0) {
        fetch_win(b, flo[0], wrows[0], 0);
        glds_w_next(0);
        glds_w_next(1);
        glds_w_next(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    int wcur = 0;                                          // buffer of the current step's weight tile

    for (; ti < t_hi; ti += t_step) {
        // this lane's output rows (one per row subtile)
        int t_out[MT], vj[MT];
        bool rvalid[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int u = r0 + (wr * MT + m) * 32 + l31;
            rvalid[m] = u < r1;
            const int uu = rvalid[m] ? u : r0;
            t_out[m] = uu / V;
            vj[m] = uu - t_out[m] * V;
        }
        f32x16 acc[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

        // geometry of the next tile (its first window is prefetched during this tile's last chunk)
        const int tn = ti + t_step;
        int nb = 0, nr0 = 0, nr1 = 0, nflo[NG], nwrows[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) { nflo[g] = 0; nwrows[g] = 0; }
        if (tn < t_hi) tile_geo(tn, nb, nr0, nr1, nflo, nwrows);

        if (ntaps > 0) {
            struct Frag { i32x4 af[MT]; i32x4 bf[NT]; };
            // this lane's B-fragment row and (swizzled) slot of k-step 0; k-step s flips bits 5-6 of the offset
            const int wb_off = ((wc * NT) * 32 + l31) * CB_WRS + ((h ^ ((l31 >> 1) & 7)) * 16);
            int gflo = flo[0];                             // first frame of the window in LDS
            auto a_offsets = [&](int tapi, int (&aoff)[MT]) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int fr = a.fs * t_out[m] + a.foff[tapi];
                    const bool ok = rvalid[m] && fr >= 0 && fr < Tin;
                    aoff[m] = (ok ? (((fr - gflo) / WFS) * V + vj[m]) * CB_RS : ZOFF) + h * 16;
                }
            };
            auto load_frag = [&](Frag& f, const int (&aoff)[MT], int wbi, int s) {
                const char* wb = wbuf + wbi * (BN * CB_WRS) + (wb_off ^ (s * 32));
#pragma unroll
                for (int m = 0; m < MT; ++m) f.af[m] = *(const i32x4*)(win + aoff[m] + s * 32);
#pragma unroll
                for (int n = 0; n < NT; ++n) f.bf[n] = *(const i32x4*)(wb + n * 32 * CB_WRS);
            };
            auto mma_frag = [&](const Frag& f) {
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int m = 0; m < MT; ++m) mma_step<T>(acc[m][n], f.af[m], f.bf[n]);
            };
            Frag fa, fb;
            int aoff[MT], aoff_n[MT];
            for (int ph = 0; ph < nchunks * NG; ++ph) {   // a phase: one window (a chunk; with GRP one tap group of it)
                const int chunk = GRP ? ph >> 1 : ph;
                const bool g1 = GRP && (ph & 1);
                const int tap0 = GRP ? (g1 ? a.gt[1] : a.gt[0]) : 0, tap1 = GRP ? (g1 ? a.gt[2] : a.gt[1]) : ntaps;
                gflo = g1 ? flo[NG - 1] : flo[0];
                lds_barrier();                             // previous window / epilogue staging fully consumed
                commit_win(g1 ? wrows[NG - 1] : wrows[0], chunk);
                if (GRP && !g1) fetch_win(b, flo[NG - 1], wrows[NG - 1], chunk);
                else if (chunk + 1 < nchunks) fetch_win(b, flo[0], wrows[0], chunk + 1);
                else if (tn < t_hi) fetch_win(nb, nflo[0], nwrows[0], 0);
                for (int tapi = tap0; tapi < tap1; ++tapi) {
                    lds_barrier();                         // weight tile of this step (and a new window) visible
                    const int wnext = (wcur + 1) & (CB_NWB - 1);
                    if (tapi == tap0) {                // first step on a new window: nothing was prefetched
                        a_offsets(tapi, aoff);
                        load_frag(fa, aoff, wcur, 0);
                    }
                    const bool chain = tapi + 1 < tap1;    // the next step reads the same window
                    // 4 k-steps; the fragments of the next one (or of the next step's first) ride under the MFMAs.
                    // The next tap's row offsets are computed behind MFMAs, not in front of the first: both waves
                    // of a SIMD leave the barrier together, and whatever precedes the first MFMA idles the pipe.
                    __builtin_amdgcn_sched_barrier(0);
                    load_frag(fb, aoff, wcur, 1);
                    mma_frag(fa);
                    CB_INTERLEAVE(MT + NT, MT * NT)
                    __builtin_amdgcn_sched_barrier(0);
                    glds_w_next((wcur + 3) & (CB_NWB - 1));  // tile of step +3 into the buffer step -1 read; behind
                    load_frag(fa, aoff, wcur, 2);            // the first MFMA group: nothing should delay that one
                    mma_frag(fb);
                    CB_INTERLEAVE(MT + NT, MT * NT)
                    __builtin_amdgcn_sched_barrier(0);
                    if (chain) a_offsets(tapi + 1, aoff_n);
                    load_frag(fb, aoff, wcur, 3);
                    mma_frag(fa);
                    CB_INTERLEAVE(MT + NT, MT * NT)
                    __builtin_amdgcn_sched_barrier(0);
                    if (chain) load_frag(fa, aoff_n, wnext, 0);
                    mma_frag(fb);
                    CB_INTERLEAVE(MT + NT, MT * NT)
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < MT; ++m) aoff[m] = aoff_n[m];
                    wcur = wnext;
                    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WPT) : "memory");   // tile of step +2 has landed
                }
            }
        }
        lds_barrier();                                   // every wave is done with the window and weight tiles

        // accumulators (+bias) -> bf16 -> staging rows
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = (wr * MT + m) * 32 + acc_row(r, h);
                    const int col = (wc * NT + n) * 32 + l31;
                    if constexpr (ET<T>::IS_BF16)
                        *(uint16_t*)(smem + (size_t)rl * SRS + col * 2) = (uint16_t)(pack_bf16x2(acc[m][n][r] + biasv[n], 0.f) & 0xFFFFu);
                    else
                        *(float*)(smem + (size_t)rl * SRS + col * 4) = acc[m][n][r] + biasv[n];
                }
        lds_barrier();
        {
            const bool cok = co0 + eseg * VEC < Cout;
            const T* auxb = (const T*)a.aux + (size_t)b * Rreal * Cout + co0 + eseg * VEC;
            T* outb = (T*)a.out + (size_t)b * Rreal * Cout + co0 + eseg * VEC;
#pragma unroll 1
            for (int g = 0; g < EPP / EG; ++g) {
                int rrow[EG];
                i32x4 av[EG];
#pragma unroll
                for (int i = 0; i < EG; ++i) {
                    const int u = r0 + erow + ERP * (g * EG + i);
                    int rr = -1;
                    if (u < r1 && cok) {
                        if (a.os == 1) rr = u;
                        else { const int t = u / V; rr = (a.os * t + a.opar) * V + (u - t * V); }
                    }
                    rrow[i] = rr;
                    if constexpr (EPI != 0) {
                        av[i] = zero16();
                        if (rr >= 0 && a.aux) av[i] = *(const i32x4*)(auxb + (size_t)rr * Cout);
                    }
                }
                if constexpr (EPI != 0) {
                    // ONE wait for the group's auxiliary rows, before its first store: the loads sit behind branches, so
                    // the compiler would wait for each row with vmcnt(0) -- from the second row on that is also a wait
                    // for the store of the row before (waits retire in issue order), one trip to memory per row
#pragma unroll
                    for (int i = 0; i < EG; ++i) asm volatile("" : "+v"(av[i]));
                }
#pragma unroll
                for (int i = 0; i < EG; ++i) {
                    if (rrow[i] >= 0) {
                        const int rl = erow + ERP * (g * EG + i);
                        i32x4 v = *(const i32x4*)(smem + (size_t)rl * SRS + eseg * 16);
                        float f[VEC];
                        unpack16<T>(v, f);
                        if constexpr (EPI == 1) {
                            float y[VEC];
                            unpack16<T>(av[i], y);
#pragma unroll
                            for (int j = 0; j < VEC; ++j) {
                                f[j] = (fmaf(y[j], e_sc[j], e_sh[j]) > 0.f) ? f[j] : 0.f;
                                s1[j] += f[j];
                                s2[j] = fmaf(f[j], y[j], s2[j]);        // raw moment; centred once at the end of the kernel
                            }
                            v = pack16<T>(f);
                        } else if constexpr (EPI == 2) {          // + residual, ReLU (BatchNorm folded into weights and bias)
                            float y[VEC];
                            unpack16<T>(av[i], y);
#pragma unroll
                            for (int j = 0; j < VEC; ++j) f[j] = fmaxf(f[j] + y[j], 0.f);
                            v = pack16<T>(f);
                        } else {
#pragma unroll
                            for (int j = 0; j < VEC; ++j) { s1[j] += f[j]; s2[j] += f[j] * f[j]; }
                        }
                        *(i32x4*)(outb + (size_t)rrow[i] * Cout) = v;
                    }
                }
            }
        }
        b = nb; r0 = nr0; r1 = nr1;
#pragma unroll
        for (int g = 0; g < NG; ++g) { flo[g] = nflo[g]; wrows[g] = nwrows[g]; }
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // weight tiles fetched ahead for steps that never came
    if (a.stats) {
        __syncthreads();
        float* red = (float*)smem;                        // [NTH][16]
#pragma unroll
        for (int j = 0; j < VEC; ++j) { red[tid * 16 + j] = s1[j]; red[tid * 16 + 8 + j] = s2[j]; }
        __syncthreads();
        if (tid < 2 * BN) {
            const int which = tid / BN, c = tid % BN;     // which: 0 = s1, 1 = s2 ; c = channel within the tile
            const int sg = c / VEC, j = c % VEC;
            float t = 0.f, t1 = 0.f;
            for (int r = 0; r < ERP; ++r) t += red[(r * EPR + sg) * 16 + which * 8 + j];
            double tv = (double)t;
            if (EPI == 1 && which == 1 && co0 + c < Cout) {
                // the epilogue gathered sum g*y; the statistic is sum g*yhat = rstd * (sum g*y - mean * sum g)
                for (int r = 0; r < ERP; ++r) t1 += red[(r * EPR + sg) * 16 + j];
                tv = (double)a.ep_rstd[co0 + c] * (tv - (double)a.ep_mean[co0 + c] * (double)t1);
            }
            if (co0 + c < Cout) atomicAdd(&a.stats[which * Cout + co0 + c], tv);
        }
    }
}

template <typename T, int WR, int WC, int MT, int NT, int EPI, bool GRP = false>
static int launch_conv_big(ConvBigArgs& a, hipStream_t stream) {
    constexpr int BM = WR * MT * 32;
    a.tiles_per_sample = ceil_div_i(a.Tv * a.V, BM);
    constexpr int BN = WC * NT * 32;
    const size_t body = (size_t)CB_WIN_ROWS * CB_RS + CB_NWB * (size_t)BN * CB_WRS;
    a.zero_row = (int)((body + CB_RS - 1) / CB_RS);
    const size_t lds = (size_t)(a.zero_row + 1) * CB_RS;
    const int nct = ceil_div_i(a.Cout, BN);
    int gx = max(1, stgcn_num_cus() / nct);
    if (gx > a.B * a.tiles_per_sample) gx = a.B * a.tiles_per_sample;
    auto kern = conv_big_kernel<T, WR, WC, MT, NT, EPI, GRP>;
    static std::atomic<uint64_t> lds_ok{0};
    if (stgcn_allow_big_lds((const void*)kern, lds_ok) != STGCN_OK) return STGCN_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)nct), dim3(WR * WC * 64), lds, stream, a);
    return hipGetLastError() == hipSuccess ? STGCN_OK : STGCN_ERR_LAUNCH;
}

// window rows a tile of BM virtual rows can need
static int conv_big_win_rows(const ConvBigArgs& a, int BM) {
    const int nf = (BM - 2) / a.V + 2;
    int fr = a.fs * (nf - 1) + (a.foff_max - a.foff_min) + 1;
    if (fr > a.Tin) fr = a.Tin;
    return fr * a.V;
}

// Forward, stride 2, 256-row tiles: the taps sorted into two groups by the parity of their frame offset, each with a
// window of every second frame.  Returns false when that does not apply or does not fit.
static bool conv_big_group_taps(ConvBigArgs& a) {
    if (a.fs != 2 || a.os != 1 || a.ntaps < 2) return false;
    int foff[9], wk[9], n = 0;
    for (int g = 0; g < 2; ++g) {
        a.gt[g] = n;
        for (int i = 0; i < a.ntaps; ++i)
            if ((a.foff[i] & 1) == g) {
                if (n == a.gt[g] || a.foff[i] < a.gmin[g]) a.gmin[g] = a.foff[i];
                if (n == a.gt[g] || a.foff[i] > a.gmax[g]) a.gmax[g] = a.foff[i];
                foff[n] = a.foff[i]; wk[n] = a.wk[i]; ++n;
            }
    }
    a.gt[2] = n;
    if (a.gt[1] == a.gt[0] || a.gt[2] == a.gt[1]) return false;          // one parity only: nothing to split
    const int nf = (256 - 2) / a.V + 2;
    for (int g = 0; g < 2; ++g)
        if (((a.fs * (nf - 1) + a.gmax[g] - a.gmin[g]) / 2 + 1) * a.V > CB_WIN_ROWS) return false;
    for (int i = 0; i < a.ntaps; ++i) { a.foff[i] = foff[i]; a.wk[i] = wk[i]; }
    return true;
}

// dry = true: only report whether the shape fits (0) or not (-1)
template <typename T>
static int conv_big_launch_any(ConvBigArgs& a, int epi, hipStream_t stream, bool dry) {
    a.foff_min = 0; a.foff_max = 0;
    for (int i = 0; i < a.ntaps; ++i) {
        if (i == 0 || a.foff[i] < a.foff_min) a.foff_min = a.foff[i];
        if (i == 0 || a.foff[i] > a.foff_max) a.foff_max = a.foff[i];
    }
    for (int g = 0; g < 2; ++g) { a.gt[g] = 0; a.gmin[g] = 0; a.gmax[g] = 0; }
    a.gt[2] = 0;
    if constexpr (ET<T>::IS_BF16) {
        if (!dry && a.Cout > 64 && epi != 1 && conv_big_group_taps(a))
            return epi == 2 ? launch_conv_big<T, 4, 2, 2, 2, 2, true>(a, stream) : launch_conv_big<T, 4, 2, 2, 2, 0, true>(a, stream);
    }
    const int Rv = a.Tv * a.V;
    // 256-row tiles read fewer fragments per MFMA; 128-row tiles waste less on the ragged last tile of a sample
    const double waste256 = 1.0 - (double)Rv / (ceil_div_i(Rv, 256) * 256.0);
    const double waste128 = 1.0 - (double)Rv / (ceil_div_i(Rv, 128) * 128.0);
    const bool fit256 = conv_big_win_rows(a, 256) <= CB_WIN_ROWS;
    const bool fit128 = conv_big_win_rows(a, 128) <= CB_WIN_ROWS;
    if (dry) return (fit256 || fit128) ? 0 : -1;
    // 64-channel column tiles (8 waves of 32 rows x 64 channels) for narrow outputs, and always in f32, whose
    // epilogue staging rows are twice as wide
    const bool narrow = a.Cout <= 64 || !ET<T>::IS_BF16;
    if (fit256 && (waste256 <= waste128 + 0.10 || !fit128)) {
        if (narrow) return epi == 1 ? launch_conv_big<T, 8, 1, 1, 2, 1>(a, stream) : epi == 2 ? launch_conv_big<T, 8, 1, 1, 2, 2>(a, stream) : launch_conv_big<T, 8, 1, 1, 2, 0>(a, stream);
        if constexpr (ET<T>::IS_BF16)
            return epi == 1 ? launch_conv_big<T, 4, 2, 2, 2, 1>(a, stream) : epi == 2 ? launch_conv_big<T, 4, 2, 2, 2, 2>(a, stream) : launch_conv_big<T, 4, 2, 2, 2, 0>(a, stream);
    }
    if (fit128) {
        if (narrow) return epi == 1 ? launch_conv_big<T, 4, 1, 1, 2, 1>(a, stream) : epi == 2 ? launch_conv_big<T, 4, 1, 1, 2, 2>(a, stream) : launch_conv_big<T, 4, 1, 1, 2, 0>(a, stream);
        if constexpr (ET<T>::IS_BF16)
            return epi == 1 ? launch_conv_big<T, 4, 2, 1, 2, 1>(a, stream) : epi == 2 ? launch_conv_big<T, 4, 2, 1, 2, 2>(a, stream) : launch_conv_big<T, 4, 2, 1, 2, 0>(a, stream);
    }
    return -1;
}

// Returns -1 when the shape is outside this kernel's domain (the caller then uses conv_rows).
template <typename T>
static int conv_big_try_t(const void* in, const void* w, void* out, const float* bias, const float* pro_scale, const float* pro_shift,
                          int epilogue, const void* aux, const float* ep_scale, const float* ep_shift, const float* ep_mean, const float* ep_rstd,
                          double* stats, int B, int Tin, int Tout, int V, int Cin, int Cout, int ntaps, int stride, int pad, int dil,
                          int transposed, hipStream_t stream, const BnPrologue* pbn) {
    if (Cin % (128 / ET<T>::ES) != 0 || Cin <= 0 || Cout % ET<T>::VEC != 0 || V < 1 || V > 32 || ntaps < 1 || ntaps > 9 || dil != 1 || (stride != 1 && stride != 2)) return -1;
    ConvBigArgs a;
    a.in = in; a.w = w; a.out = out; a.bias = bias; a.pro_scale = pro_scale; a.pro_shift = pro_shift; a.pbn = pbn ? *pbn : BnPrologue{};
    a.aux = aux; a.ep_scale = ep_scale; a.ep_shift = ep_shift; a.ep_mean = ep_mean; a.ep_rstd = ep_rstd; a.stats = stats;
    a.B = B; a.Tin = Tin; a.V = V; a.Cin = Cin; a.Cout = Cout; a.Tout_real = Tout;
    a.tiles_per_sample = 0; a.zero_row = 0;
    if (!transposed || stride == 1) {
        a.fs = transposed ? 1 : stride; a.os = 1; a.opar = 0; a.Tv = Tout; a.ntaps = ntaps;
        for (int i = 0; i < ntaps; ++i) { a.foff[i] = transposed ? pad - i : i - pad; a.wk[i] = i; }
        return conv_big_launch_any<T>(a, epilogue, stream, false);
    }
    // transposed, stride 2: one launch per output-frame parity
    for (int pass = 0; pass < 2; ++pass) {               // pass 0 checks both parities fit, pass 1 launches
        for (int par = 0; par < 2; ++par) {
            a.fs = 1; a.os = 2; a.opar = par; a.Tv = (Tout - par + 1) / 2; a.ntaps = 0;
            if (a.Tv <= 0) continue;
            for (int k = 0; k < ntaps; ++k) {
                const int num = par + pad - k;
                if ((num & 1) == 0) { a.foff[a.ntaps] = num / 2; a.wk[a.ntaps] = k; ++a.ntaps; }
            }
            const int rc = conv_big_launch_any<T>(a, epilogue, stream, pass == 0);
            if (rc != STGCN_OK) return rc;
        }
    }
    return STGCN_OK;
}

int conv_big_try(int dtype, const void* in, const void* w, void* out, const float* bias, const float* pro_scale, const float* pro_shift,
                 int epilogue, const void* aux, const float* ep_scale, const float* ep_shift, const float* ep_mean, const float* ep_rstd,
                 double* stats, int B, int Tin, int Tout, int V, int Cin, int Cout, int ntaps, int stride, int pad, int dil,
                 int transposed, hipStream_t stream, const BnPrologue* pbn) {
    if (dtype == STGCN_BF16)
        return conv_big_try_t<bf16>(in, w, out, bias, pro_scale, pro_shift, epilogue, aux, ep_scale, ep_shift, ep_mean, ep_rstd,
                                    stats, B, Tin, Tout, V, Cin, Cout, ntaps, stride, pad, dil, transposed, stream, pbn);
    if (dtype == STGCN_F32)
        return conv_big_try_t<float>(in, w, out, bias, pro_scale, pro_shift, epilogue, aux, ep_scale, ep_shift, ep_mean, ep_rstd,
                                     stats, B, Tin, Tout, V, Cin, Cout, ntaps, stride, pad, dil, transposed, stream, pbn);
    return -1;
}
