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

// NPW = window pieces (16 bytes) per thread: the window of the NEXT tile is fetched into registers while the current
// tile's MFMAs run, so its load latency is hidden.
// EXACT: ntaps == MAXT, so the tap loops carry no run-time tests (the LDS reads of a k-step are then issued together).
template <typename T, int MAXT, int NPW, bool EXACT>
__global__ __launch_bounds__(512) void conv_wgrad_kernel(ConvWgradArgs a) {
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
    const int coh = wave & 1, cih = (wave >> 1) & 1;     // bf16: wave's 32x32 quadrant ...
    const int kg = wave >> 2;                            // ... and k-step parity (8 waves = 4 quadrants x 2)

    f32x16 acc[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // tr-read lane roles (bf16): block row q, column quad p, 16-column half g1
    const int q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = (lane >> 4) & 1;

    // staging role: fixed 16-byte segment per thread; rows tid>>3 + 64*i
    const int sseg = tid & 7, srow = tid >> 3;
    float psc[VEC], psh[VEC];
    const bool has_pro = a.pro_scale != nullptr;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        psc[j] = has_pro ? a.pro_scale[ci0 + sseg * VEC + j] : 1.f;
        psh[j] = has_pro ? a.pro_shift[ci0 + sseg * VEC + j] : 0.f;
    }
    const bool co_ok = co0 + sseg * VEC < a.Cout;

    const int ntiles = a.B * a.tiles_per_sample;
    i32x4 dreg[2], wreg[NPW];
    unsigned wok = 0;                                    // bit i: wreg[i] holds a row inside [0, Tin)
    auto geom = [&](int g, int& b, int& r0, int& r1, int& tlo, int& flo, int& wrows) {
        b = g / a.tiles_per_sample;
        const int tile = g - b * a.tiles_per_sample;
        r0 = tile * 128; r1 = min(r0 + 128, Rout);
        tlo = r0 / V;
        const int thi = (r1 - 1) / V;
        flo = a.stride * tlo - a.pad;                                        // may be negative
        const int fhi = a.stride * thi - a.pad + (a.ntaps - 1) * a.dil;      // may exceed Tin-1
        wrows = (fhi - flo + 1) * V;
    };
    auto fetch = [&](int g) {
        int b, r0, r1, tlo, flo, wrows;
        geom(g, b, r0, r1, tlo, flo, wrows);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = srow + 64 * i;
            dreg[i] = zero16();
            if (r0 + row < r1 && co_ok)
                dreg[i] = *(const i32x4*)((const char*)a.dout + (((size_t)b * Rout + r0 + row) * a.Cout + co0 + sseg * VEC) * ES);
        }
        wok = 0;
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int row = srow + 64 * i;
            wreg[i] = zero16();
            if (row < wrows) {
                const int fr = row / V;
                const int f = flo + fr;
                if (f >= 0 && f < a.Tin) {
                    wok |= 1u << i;
                    wreg[i] = *(const i32x4*)((const char*)a.in + ((((size_t)b * a.Tin + f) * V + (row - fr * V)) * a.Cin + ci0 + sseg * VEC) * ES);
                }
            }
        }
    };
    auto commit = [&](int wrows) {
#pragma unroll
        for (int i = 0; i < 2; ++i) *(i32x4*)(dtile + (srow + 64 * i) * CW_RS + sseg * 16) = dreg[i];
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int row = srow + 64 * i;
            if (row < wrows) {
                i32x4 v = wreg[i];
                if (has_pro && ((wok >> i) & 1u)) {
                    float f[VEC];
                    unpack16<T>(v, f);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) f[j] = fmaxf(fmaf(f[j], psc[j], psh[j]), 0.f);
                    v = pack16<T>(f);
                }
                *(i32x4*)(win + (size_t)row * CW_RS + sseg * 16) = v;
            }
        }
    };

    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int g = blockIdx.x; g < ntiles; g += a.nseg) {
        int b, r0, r1, tlo, flo, wrows;
        geom(g, b, r0, r1, tlo, flo, wrows);

        lds_barrier();       // previous tile fully consumed
        if (tid < 128) {
            const int r = min(r0 + tid, r1 - 1);
            const int to = r / V;
            rb[tid] = ((a.stride * (to - tlo)) * V + (r - to * V)) * CW_RS;
        }
        commit(wrows);
        lds_barrier();
        if (g + a.nseg < ntiles) fetch(g + a.nseg);      // in flight during this tile's MFMAs

        if constexpr (BF) {
            // wave (coh, cih, kg): the k-steps (16 rows each) of parity kg among the 8 of the tile
            const int acol = (coh * 32 + 16 * g1 + 4 * p4) * 2;
            const int bcol = (cih * 32 + 16 * g1 + 4 * p4) * 2;
#pragma unroll 2
            for (int sj = 0; sj < 4; ++sj) {
                const int s = 2 * sj + kg;
                const int rowa = 16 * s + 8 * h + q4;       // this lane's address row, first read
                const int rowb = rowa + 4;                  // second read
                const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(dtile + rowa * CW_RS + acol));
                const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(dtile + rowb * CW_RS + acol));
                const s16x8 af = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                const int wa = rb[rowa] + bcol, wb = rb[rowb] + bcol;
#pragma unroll
                for (int t = 0; t < MAXT; ++t) {
                    if (EXACT || t < a.ntaps) {
                        const int sh = t * a.dil * V * CW_RS;
                        const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(win + wa + sh));
                        const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(win + wb + sh));
                        const s16x8 bfv = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfv, acc[t], 0, 0, 0);
                    }
                }
            }
        } else {
            // wave owns rows [16*wave, 16*wave+16) of the 32 x 32 (co, ci) tile's contraction; 8 steps of 2 rows
#pragma unroll 4
            for (int s = 0; s < 8; ++s) {
                const int row = 16 * wave + 2 * s + h;
                const float av = *(const float*)(dtile + row * CW_RS + l31 * 4);
                const int wo = rb[row] + l31 * 4;
#pragma unroll
                for (int t = 0; t < MAXT; ++t) {
                    if (EXACT || t < a.ntaps) {
                        const float bv = *(const float*)(win + wo + t * a.dil * V * CW_RS);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
                    }
                }
            }
        }
    }

    // Fold the waves that hold partial sums of the same tile through LDS, one tap at a time: f32 -- all 8 waves;
    // bf16 -- the two k-step parities of each quadrant.  One partial per workgroup goes to the slab.
    float* red = (float*)smem;                            // [8 waves][16][64] floats = 32 KiB
    const int part = blockIdx.x;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        if (EXACT || t < a.ntaps) {
            lds_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[t][r];
            lds_barrier();
            if constexpr (BF) {
                // 4 quadrants x 1024 values: 8 per thread
                for (int e = tid; e < 4 * 1024; e += 512) {
                    const int quad = e >> 10, idx = e & 1023;
                    const float v = red[quad * 1024 + idx] + red[(quad + 4) * 1024 + idx];
                    const int r = idx >> 6, ln = idx & 63;
                    const int co = co0 + (quad & 1) * 32 + acc_row(r, ln >> 5);
                    const int ci = ci0 + (quad >> 1) * 32 + (ln & 31);
                    if (co < a.Cout && ci < a.Cin)
                        a.slab[(((size_t)part * a.ntaps + t) * a.Cout + co) * a.Cin + ci] = v;
                }
            } else {
                for (int idx = tid; idx < 1024; idx += 512) {
                    float v = 0.f;
#pragma unroll
                    for (int w = 0; w < 8; ++w) v += red[w * 1024 + idx];
                    const int r = idx >> 6, ln = idx & 63;
                    const int co = co0 + acc_row(r, ln >> 5);
                    const int ci = ci0 + (ln & 31);
                    if (co < a.Cout && ci < a.Cin)
                        a.slab[(((size_t)part * a.ntaps + t) * a.Cout + co) * a.Cin + ci] = v;
                }
            }
        }
    }
}

// The partials' fixed-order sum, dW[co][ci][tap] = sum_part slab[part][tap][co][ci] (torch conv weight layout
// [Cout][Cin][ntaps][1]): launched at once, or left on the caller's list.
static int conv_wgrad_sum(StgcnSumList* list, const float* slab, float* dW, int nparts, int ntaps, int Cout, int Cin, hipStream_t stream) {
    StgcnPendingSum s{};
    s.slab = slab; s.out1 = dW; s.out2 = nullptr;
    s.n1 = (unsigned long long)ntaps * Cout * Cin; s.n2 = 0;
    s.nparts = nparts; s.kind = STGCN_SUM_CONV_W; s.ntaps = ntaps; s.cout = Cout; s.cin = Cin;
    return stgcn_sum_or_defer(list, s, stream);
}

// ------------------------------------------------------------------------------------------
// bf16 fast path for "same"-padded temporal kernels with stride 1 or 2: persistent workgroups with an LDS ring,
// as in conv_t9.  A workgroup owns one (64 co x 64 ci) tile pair for all taps and walks runs of consecutive
// output-row tiles of one sample.  Its 8 waves (two per SIMD) are 4 quadrants (32 x 32, 9 accumulators each) x 2
// k-step parities; the two parities are summed through LDS at the end.  Input rows (64-channel slice,
// prologue applied, zeros outside the sample) sit in a 512-slot ring (slot = row & 511) so each is fetched once
// per run; dout tiles are double-buffered.  The next tile's rows are fetched into registers under the MFMA phase.
//   stride 1: 256 output rows per tile (16 k-steps), input advances 256 rows per tile.
//   stride 2: OT whole output frames per tile (OT*V <= 128 rows, 8 k-steps), input advances 2*OT*V <= 256 rows;
//             output row j of the tile pairs with input row  first + j + (j / V) * V + tap * V.
struct ConvWgradRingArgs {
    const void* in; const void* dout; float* slab;
    const float* pro_scale; const float* pro_shift;
    int B, Rin, Rout, V, Cin, Cout, ntaps;
    int nruns, segs_per_sample, tiles_per_seg, tiles_per_sample;
    int tile_rows;       // output rows one tile advances by (256, or OT*V for stride 2)
};

// NT = number of taps (compile-time: the tap loops carry no run-time tests, so a k-step's LDS reads issue together)
template <int STRIDE, int NT>
__global__ __launch_bounds__(512) void conv_wgrad_ring_kernel(ConvWgradRingArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS rows are 128 bytes, unpadded; the ring is 64 KiB at the start of the LDS allocation, so ring addresses
    // wrap as 16-bit sums.  The 64-byte half of a row is flipped on rows with bit 1 set -- in the dout tiles always,
    // in the ring for stride 1 -- so the four consecutive rows one transposed read touches per 32-lane half land on
    // four different quarters of the bank row (conflict-free).  Stride 1: each lane keeps one ring address per tap
    // and advances it by 32 rows per k-step pair.  Stride 2: the lane's row offsets of its k-steps are fixed for
    // the whole kernel (registers); a read adds the tap's scalar base for the tile to them.
    constexpr int RS = 128;
    constexpr int KS = STRIDE == 1 ? 16 : 8;              // k-steps (16 output rows each) per tile
    constexpr int DROWS = KS * 16;                        // dout rows per tile buffer
    constexpr int DPASS = DROWS / 64;                     // staging passes for a dout tile (64 rows per pass)
    constexpr int KW = KS / 2;                            // k-steps per wave: waves 0-3 take the even ones, 4-7 the odd
    char* ring = smem;                                   // [512 + 4][RS]: slots 512..515 mirror slots 0..3
    char* dt = smem + 516 * RS;                          // 2 x [DROWS][RS]
    auto swz_d = [](int row) -> int { return (row & 2) << 5; };                       // dout tiles
    auto swz = [](int row) -> int { return STRIDE == 1 ? (row & 2) << 5 : 0; };       // ring
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int V = a.V, Rin = a.Rin, Rout = a.Rout, Cin = a.Cin, Cout = a.Cout, ntaps = a.ntaps;
    const int half = (ntaps - 1) / 2;
    const int co0 = blockIdx.y * 64, ci0 = blockIdx.z * 64;
    const int coh = wave & 1, cih = (wave >> 1) & 1, kg = wave >> 2;
    const int q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = (lane >> 4) & 1;
    const int acol = ((coh * 32 + 16 * g1 + 4 * p4) * 2) ^ swz_d(q4); // dout rows of a read are q4 (+4) mod 8
    const int bcol = (cih * 32 + 16 * g1 + 4 * p4) * 2;
    const int tile_rows = a.tile_rows;
    const int in_adv = STRIDE * tile_rows;                // input rows one tile advances by
    const int span = STRIDE == 1 ? 256 + 2 * half * V : in_adv + (2 * half - 1) * V;   // input rows one tile touches

    // input-row offset (from the span start) that pairs with output row j of the tile.  Stride 2: j + (j / V) * V;
    // rows past the tile pair with zero dout rows and are pointed at row 0 of the span, which always holds finite
    // data.  j < 256 and V <= 32, so the float quotient is exact.
    const float rV = 1.0f / (float)V;
    auto in_off = [&](int j) -> int {
        if constexpr (STRIDE == 1) return j;
        if (j >= tile_rows) return 0;
        const int q = (int)(((float)j + 0.5f) * rV);
        return j + q * V;
    };

    // stride 2: ring byte offsets (row offset * RS + column) of the two reads of each of the wave's k-steps
    uint16_t rofs[KW][2];
    if constexpr (STRIDE == 2) {
#pragma unroll
        for (int i = 0; i < KW; ++i) {
            const int j = 16 * (2 * i + kg) + 8 * h + q4;
            rofs[i][0] = (uint16_t)(in_off(j) * RS + bcol);
            rofs[i][1] = (uint16_t)(in_off(j + 4) * RS + bcol);
        }
    }

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int sseg = tid & 7, srow = tid >> 3;           // staging: 64 rows per pass
    float psc[8], psh[8];
    const bool has_pro = a.pro_scale != nullptr;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        psc[j] = has_pro ? a.pro_scale[ci0 + sseg * 8 + j] : 1.f;
        psh[j] = has_pro ? a.pro_shift[ci0 + sseg * 8 + j] : 0.f;
    }
    auto transform = [&](i32x4 v) -> i32x4 {
        if (!has_pro) return v;
        float f[8];
        unpack16<bf16>(v, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = fmaxf(fmaf(f[j], psc[j], psh[j]), 0.f);
        return pack16<bf16>(f);
    };
    const bool co_ok = co0 + sseg * 8 < Cout;

    for (int run = blockIdx.x; run < a.nruns; run += gridDim.x) {
        const int b = run / a.segs_per_sample;
        const int seg = run - b * a.segs_per_sample;
        const int tile_beg = seg * a.tiles_per_seg;
        const int tile_end = min(a.tiles_per_sample, tile_beg + a.tiles_per_seg);
        if (tile_beg >= tile_end) continue;
        const uint16_t* inb = (const uint16_t*)a.in + (size_t)b * Rin * Cin + ci0;
        const uint16_t* dob = (const uint16_t*)a.dout + (size_t)b * Rout * Cout + co0;
        // Rows in flight: pf holds input rows on their way to the ring, pd the next dout tile.  Stride 2 (8 k-steps a
        // tile: half a tile is shorter than a trip to memory) fetches both at the top of a tile, from clamped addresses
        // on every path (rows outside the sample become zeros when they are committed): loads behind a branch are
        // "maybe issued" to the compiler's wait-count pass, and its wait for pd in the middle of the tile would then
        // also be a wait for pf, issued just after it.  Stride 1 has no registers left for a second set (it spills
        // with one, and runs 6-8 % slower): there pd IS pf, the dout tile travels in the first half of a tile and the
        // input rows in the second.
        constexpr bool TWO_SETS = STRIDE == 2;
        i32x4 pf[4], pd2[TWO_SETS ? DPASS : 1];
        i32x4* const pd = TWO_SETS ? pd2 : pf;
        auto fetch_in = [&](int p_first) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int p = p_first + srow + 64 * i;
                if constexpr (TWO_SETS) {
                    pf[i] = *(const i32x4*)(inb + (size_t)min(max(p, 0), Rin - 1) * Cin + sseg * 8);
                } else {
                    pf[i] = zero16();
                    if (p >= 0 && p < Rin) pf[i] = *(const i32x4*)(inb + (size_t)p * Cin + sseg * 8);
                }
            }
        };
        auto commit_in = [&](int p_first, int nrows) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rel = srow + 64 * i;
                const int p = p_first + rel;
                if (rel < nrows) {
                    i32x4 v = pf[i];
                    if (p >= 0 && p < Rin) v = transform(v);
                    else if (TWO_SETS) v = zero16();
                    const int slot = p & 511;
                    const int off = slot * RS + ((sseg * 16) ^ swz(slot));
                    *(i32x4*)(ring + off) = v;
                    if (slot < 4) *(i32x4*)(ring + off + 512 * RS) = v;
                }
            }
        };
        const int dseg = co_ok ? sseg * 8 : 0;
        auto fetch_do = [&](int p_first) {
#pragma unroll
            for (int i = 0; i < DPASS; ++i) {
                const int rel = srow + 64 * i;
                const int p = p_first + rel;
                if constexpr (TWO_SETS) {
                    pd[i] = *(const i32x4*)(dob + (size_t)min(p, Rout - 1) * Cout + dseg);
                } else {
                    pd[i] = zero16();
                    if (rel < tile_rows && p < Rout && co_ok) pd[i] = *(const i32x4*)(dob + (size_t)p * Cout + sseg * 8);
                }
            }
        };
        auto commit_do = [&](int buf, int p_first) {
#pragma unroll
            for (int i = 0; i < DPASS; ++i) {
                const int rel = srow + 64 * i;
                i32x4 v = pd[i];
                if (TWO_SETS && !(rel < tile_rows && p_first + rel < Rout && co_ok)) v = zero16();
                *(i32x4*)(dt + (size_t)buf * DROWS * RS + rel * RS + ((sseg * 16) ^ swz_d(srow))) = v;
            }
        };

        int p0 = tile_beg * tile_rows;                    // first output row of the tile
        lds_barrier();                                  // previous run fully consumed
        {
            const int first = STRIDE * p0 - half * V;
            for (int base = 0; base < span; base += 256) {
                fetch_in(first + base);
                commit_in(first + base, span - base);
            }
            fetch_do(p0);
            commit_do(0, p0);
        }
        int cur = 0;
        // stride 1: ring byte offset of this lane's first operand row of the coming k-step, per tap
        uint16_t badr[NT];
        if constexpr (STRIDE == 1) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int slot = (p0 - half * V + 16 * kg + 8 * h + q4 + t * V) & 511;
                badr[t] = (uint16_t)(slot * RS + (bcol ^ swz(slot)));
            }
        }
        for (int tile = tile_beg; tile < tile_end; ++tile, p0 += tile_rows) {
            const bool more = tile + 1 < tile_end;
            const int first = STRIDE * p0 - half * V;     // first input row of this tile's span
            const int pnew = first + span;                // first input row the next tile adds
            lds_barrier();                              // ring and dt[cur] visible
            if (more) {
                fetch_do(p0 + tile_rows);
                if constexpr (TWO_SETS) fetch_in(pnew);   // a whole tile's arithmetic to land, half of one for pd
            }
            const char* dcur = dt + (size_t)cur * DROWS * RS;
            // Fragment reads run one MFMA ahead: the tap t+1 operand (or the next k-step's dout fragment and
            // tap-0 operand) is requested before the tap t MFMA issues, so the LDS latency hides under the MFMA pipe.
            auto tr = [&](const char* p) {
                return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
            };
            auto lda = [&](int s) -> s16x8 {
                const int rowa = 16 * s + 8 * h + q4;
                const s16x4 a0 = tr(dcur + rowa * RS + acol), a1 = tr(dcur + (rowa + 4) * RS + acol);
                return s16x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            };
            // stride 2: ring byte offset of row `first + t*V` (the tap's base for this tile), uniform
            uint16_t tapb[NT];
            if constexpr (STRIDE == 2) {
#pragma unroll
                for (int t = 0; t < NT; ++t) tapb[t] = (uint16_t)(((first + t * V) & 511) * RS);
            }
            // i: index of the wave's k-step within the tile.  adv (stride 1): bytes the lane's tap address moves on
            // after the read (0 for the unused read after the last k-step)
            auto ldb = [&](int i, int t, int adv) -> s16x8 {
                const char* p0;
                const char* p1;
                if constexpr (STRIDE == 1) {
                    p0 = ring + (uint32_t)badr[t];
                    p1 = p0 + 4 * RS;                       // rows +4: the mirror slots cover the wrap
                    badr[t] = (uint16_t)(badr[t] + (uint16_t)adv);
                } else {
                    p0 = ring + (uint32_t)(uint16_t)(rofs[i][0] + tapb[t]);
                    p1 = ring + (uint32_t)(uint16_t)(rofs[i][1] + tapb[t]);
                }
                const s16x4 b0 = tr(p0), b1 = tr(p1);
                return s16x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
            };
            constexpr int ADV = 32 * RS;                  // a wave's next k-step is 32 rows on
            constexpr int STEP_UNROLL = (STRIDE == 2 || NT == 1) ? KW : 1;
            s16x8 af = lda(kg), bc = ldb(0, 0, ADV);
#pragma unroll STEP_UNROLL
            for (int i = 0; i < KW; ++i) {
                // the step after the tile's last wraps to the first: its reads are valid and unused
                const int adv0 = i == KW - 1 ? 0 : ADV;
                const int in = (i + 1) & (KW - 1);
                s16x8 afn = af;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    s16x8 bn;
                    if (t + 1 < NT) bn = ldb(i, t + 1, ADV);
                    else { afn = lda(2 * in + kg); bn = ldb(in, 0, adv0); }
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bc, acc[t], 0, 0, 0);
                    if (t + 1 < NT) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    bc = bn;
                }
                af = afn;
                if (i == KW / 2 - 1 && more) {
                    commit_do(cur ^ 1, p0 + tile_rows);   // last read a tile ago: every wave has passed a barrier since
                    if constexpr (!TWO_SETS) fetch_in(pnew);
                }
            }
            lds_barrier();                              // every wave is done reading the ring
            if (more) commit_in(pnew, 256);
            cur ^= 1;
        }
    }

    // fold the odd-k-step waves into their even partners through LDS, one tap at a time
    float* red = (float*)smem;                            // [4][16][64] floats
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        {
            lds_barrier();
            if (kg == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((wave & 3) * 16 + r) * 64 + lane] = acc[t][r];
            }
            lds_barrier();
            if (kg == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] += red[(wave * 16 + r) * 64 + lane];
            }
        }
    }
    if (kg != 0) return;
    const int part = blockIdx.x;
    const int cob = co0 + coh * 32;
    const int ci = ci0 + cih * 32 + l31;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = cob + acc_row(r, h);
                if (co < Cout && ci < Cin)
                    a.slab[(((size_t)part * ntaps + t) * Cout + co) * Cin + ci] = acc[t][r];
            }
        }
    }
}

// -1: shape not covered
static int conv_wgrad_ring_try(const void* in, const void* dout, float* dW, const float* pro_scale, const float* pro_shift,
                               int B, int Tin, int Tout, int V, int Cin, int Cout, int ntaps, int stride,
                               void* ws, size_t ws_bytes, StgcnSumList* list, hipStream_t stream) {
    if (Cin % 64 != 0 || Cout % 8 != 0 || V > 32 || V < 1 || (ntaps != 9 && ntaps != 1)) return -1;
    if (stride != 1 && stride != 2) return -1;
    ConvWgradRingArgs a;
    a.in = in; a.dout = dout; a.slab = (float*)ws; a.pro_scale = pro_scale; a.pro_shift = pro_shift;
    a.B = B; a.Rin = Tin * V; a.Rout = Tout * V; a.V = V; a.Cin = Cin; a.Cout = Cout; a.ntaps = ntaps;
    int ntiles;
    if (stride == 1) { a.tile_rows = 256; ntiles = ceil_div_i(a.Rout, 256); }
    else { const int ot = 128 / V; a.tile_rows = ot * V; ntiles = ceil_div_i(Tout, ot); }
    a.tiles_per_sample = ntiles;
    const int ncot = ceil_div_i(Cout, 64), ncit = Cin / 64;
    int nwg = stgcn_num_cus() / (ncot * ncit);
    if (nwg < 1) nwg = 1;
    const size_t per_part = (size_t)ntaps * Cout * Cin * sizeof(float);
    while (nwg > 1 && (size_t)nwg * per_part > ws_bytes) nwg /= 2;
    if ((size_t)nwg * per_part > ws_bytes) return STGCN_ERR_WORKSPACE;
    int segs = ceil_div_i(nwg, B);                        // runs = B * segs >= nwg
    if (segs > ntiles) segs = ntiles;
    int tps = ceil_div_i(ntiles, segs);
    if (tps < 4 && ntiles >= 4) tps = 4;
    segs = ceil_div_i(ntiles, tps);
    a.tiles_per_seg = tps; a.segs_per_sample = segs; a.nruns = B * segs;
    if (nwg > a.nruns) nwg = a.nruns;
    const size_t lds = 516 * 128 + 2 * (stride == 1 ? 256 : 128) * 128;
    const int variant = (stride - 1) * 2 + (ntaps == 9 ? 1 : 0);
    const void* fns[4] = {(const void*)conv_wgrad_ring_kernel<1, 1>, (const void*)conv_wgrad_ring_kernel<1, 9>,
                          (const void*)conv_wgrad_ring_kernel<2, 1>, (const void*)conv_wgrad_ring_kernel<2, 9>};
    static std::atomic<uint64_t> lds_ok[4];
    if (stgcn_allow_big_lds(fns[variant], lds_ok[variant]) != STGCN_OK) return STGCN_ERR_LAUNCH;
    const dim3 grid((unsigned)nwg, (unsigned)ncot, (unsigned)ncit);
    switch (variant) {
        case 0: hipLaunchKernelGGL((conv_wgrad_ring_kernel<1, 1>), grid, dim3(512), lds, stream, a); break;
        case 1: hipLaunchKernelGGL((conv_wgrad_ring_kernel<1, 9>), grid, dim3(512), lds, stream, a); break;
        case 2: hipLaunchKernelGGL((conv_wgrad_ring_kernel<2, 1>), grid, dim3(512), lds, stream, a); break;
        default: hipLaunchKernelGGL((conv_wgrad_ring_kernel<2, 9>), grid, dim3(512), lds, stream, a); break;
    }
    if (hipGetLastError() != hipSuccess) return STGCN_ERR_LAUNCH;
    return conv_wgrad_sum(list, a.slab, dW, nwg, ntaps, Cout, Cin, stream);
}

template <typename T>
static int conv_wgrad_run(ConvWgradArgs& a, float* dW, size_t ws_bytes, StgcnSumList* list, hipStream_t stream) {
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
    const int wave_parts = 1;                             // the waves of a workgroup are folded in LDS
    const size_t per_part = (size_t)a.ntaps * a.Cout * a.Cin * sizeof(float);
    while (nseg > 1 && (size_t)nseg * wave_parts * per_part > ws_bytes) nseg /= 2;
    if ((size_t)nseg * wave_parts * per_part > ws_bytes) return STGCN_ERR_WORKSPACE;
    a.nseg = nseg;
    dim3 grid((unsigned)nseg, (unsigned)ncot, (unsigned)ncit);
    int rc;
    const int npw = ceil_div_i(a.win_rows_max, 64);      // window pieces per thread (64 rows per staging pass)
    if (npw > 12) return STGCN_ERR_ARG;
#define CW_LAUNCH(MT_, NP_, EX_) do { \
        auto kern = conv_wgrad_kernel<T, MT_, NP_, EX_>; \
        static std::atomic<uint64_t> lds_ok_{0}; \
        if (stgcn_allow_big_lds((const void*)kern, lds_ok_) != STGCN_OK) return STGCN_ERR_LAUNCH; \
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, stream, a); } while (0)
    if (a.ntaps == 1) {
        if (npw <= 6) CW_LAUNCH(1, 6, true); else CW_LAUNCH(1, 12, true);
    } else if (a.ntaps == 9) {
        if (npw <= 6) CW_LAUNCH(9, 6, true); else if (npw <= 9) CW_LAUNCH(9, 9, true); else CW_LAUNCH(9, 12, true);
    } else {
        CW_LAUNCH(9, 12, false);
    }
#undef CW_LAUNCH
    rc = hipGetLastError() == hipSuccess ? STGCN_OK : STGCN_ERR_LAUNCH;
    if (rc) return rc;
    return conv_wgrad_sum(list, a.slab, dW, nseg * wave_parts, a.ntaps, a.Cout, a.Cin, stream);
}

// streaming path for the 1-tap convolution of the residual branch (conv1s.hip); -1 = shape not covered
int conv1s_wgrad_try(const void* in, const void* dout, int B, int Tin, int Tout, int V, int Cin, int Cout, int stride,
                     void* ws, size_t ws_bytes, int* nparts, hipStream_t stream);

static int conv_wgrad_any(int dtype, const void* in, const void* dout, float* dW,
                          const float* pro_scale, const float* pro_shift,
                          int B, int Tin, int Tout, int V, int Cin, int Cout,
                          int ntaps, int stride, int pad, int dil,
                          void* workspace, size_t workspace_bytes, StgcnSumList* list, hipStream_t stream) {
    if (!in || !dout || !dW || !workspace) return STGCN_ERR_ARG;
    if ((pro_scale == nullptr) != (pro_shift == nullptr)) return STGCN_ERR_ARG;
    ConvWgradArgs a;
    a.in = in; a.dout = dout; a.slab = (float*)workspace; a.pro_scale = pro_scale; a.pro_shift = pro_shift;
    a.B = B; a.Tin = Tin; a.Tout = Tout; a.V = V; a.Cin = Cin; a.Cout = Cout;
    a.ntaps = ntaps; a.stride = stride; a.pad = pad; a.dil = dil;
    a.win_rows_max = 0; a.tiles_per_sample = 0; a.nseg = 0;
    if (dtype == STGCN_BF16 && ntaps == 1 && pad == 0 && dil == 1 && !pro_scale) {
        int nparts = 0;
        const int rc = conv1s_wgrad_try(in, dout, B, Tin, Tout, V, Cin, Cout, stride, workspace, workspace_bytes, &nparts, stream);
        if (rc == STGCN_OK) return conv_wgrad_sum(list, (const float*)workspace, dW, nparts, 1, Cout, Cin, stream);
        if (rc > 0) return rc;
    }
    if (dtype == STGCN_BF16 && dil == 1 && pad == (ntaps - 1) / 2 && B > 0 && Tin > 0 &&
        Tout == (Tin - 1) / stride + 1) {
        const int rc = conv_wgrad_ring_try(in, dout, dW, pro_scale, pro_shift, B, Tin, Tout, V, Cin, Cout, ntaps, stride,
                                           workspace, workspace_bytes, list, stream);
        if (rc >= 0) return rc;
    }
    if (dtype == STGCN_BF16) return conv_wgrad_run<bf16>(a, dW, workspace_bytes, list, stream);
    if (dtype == STGCN_F32) return conv_wgrad_run<float>(a, dW, workspace_bytes, list, stream);
    return STGCN_ERR_ARG;
}

extern "C" int stgcn_conv_wgrad(int dtype, const void* in, const void* dout, float* dW,
                                const float* pro_scale, const float* pro_shift,
                                int B, int Tin, int Tout, int V, int Cin, int Cout,
                                int ntaps, int stride, int pad, int dil,
                                void* workspace, size_t workspace_bytes, void* stream) {
    return conv_wgrad_any(dtype, in, dout, dW, pro_scale, pro_shift, B, Tin, Tout, V, Cin, Cout, ntaps, stride, pad, dil,
                          workspace, workspace_bytes, nullptr, (hipStream_t)stream);
}

// The same partial sums with the final sum left on the caller's list (stgcn_sum_list_run launches it together with
// the other sums waiting there); *workspace_used = bytes of the workspace that must stay untouched until then.
extern "C" int stgcn_conv_wgrad_deferred(int dtype, const void* in, const void* dout, float* dW,
                                         const float* pro_scale, const float* pro_shift,
                                         int B, int Tin, int Tout, int V, int Cin, int Cout,
                                         int ntaps, int stride, int pad, int dil,
                                         void* workspace, size_t workspace_bytes, void* sum_list, size_t* workspace_used, void* stream) {
    StgcnSumList* list = (StgcnSumList*)sum_list;
    if (!list || !workspace_used) return STGCN_ERR_ARG;
    const int first = list->n;
    const int rc = conv_wgrad_any(dtype, in, dout, dW, pro_scale, pro_shift, B, Tin, Tout, V, Cin, Cout, ntaps, stride, pad, dil,
                                  workspace, workspace_bytes, list, (hipStream_t)stream);
    if (rc != STGCN_OK) { list->n = first; return rc; }
    *workspace_used = stgcn_sum_list_extent(list, first, workspace);
    return STGCN_OK;
}
