// Graph convolution (1x1 K*C projection fused with the partitioned-adjacency aggregation)
// and its parameter gradients, on MFMA, one skeleton frame per wave.
//
// gcn_fused:   out[f, a, n] = sum_k sum_b adj[k][a][b] * ( sum_c in[f, b, c] * w[k][n][c] + bias[k][n] )
//   Forward uses adj[k][a=w][b=v] = A_eff[k][v][w] (A_eff itself with the transposed-adjacency flag),
//   w[k][n=co][c=ci] = W[k*Cout+co][ci].
//   Input gradient is the same kernel with adj[k][a=v][b=w] = A_eff[k][v][w], w[k][n=ci][c=co]
//   = W[k*Cout+co][ci] and no bias.
//   Per (frame, 32-column tile): the projection z_k (joints x columns, 32x32 accumulator) is
//   computed with the frame's rows as the MFMA A operand straight from global memory (channels
//   are contiguous), then each z_k accumulator is converted in registers into the B operand of the
//   aggregation MFMA (the accumulator's row index -- the joint -- is the contraction index), so the
//   K*C intermediate never leaves the register file.
// gcn_agrad:   dA[k][v][w] = sum_{f, co} ( sum_ci W[k][co][ci] x[f,v,ci] + bias[k][co] ) * dy[f, w, co]
// gcn_wgrad:   dW[k][co][ci] = sum_{f, v} (sum_w A_eff[k][v][w] dy[f,w,co]) * x[f, v, ci];  db[k][co] likewise.
#include "stgcn_common.h"
#include <type_traits>

#define GCN_KMAX 3

// ------------------------------------------------------------------------------------------
struct GcnFusedArgs {
    const void* in; const void* w; const float* adj; const float* bias; void* out; const void* add;
    const uint16_t* add_mask;    // optional: 16 bits per 16-byte piece of add; bit j clear = element j counts as zero
    double* stats;
    int NF, V, Cin, Nout, K, ntw, adj_t;
};

// Adjacency operand fragments for the chained aggregation MFMA, built in LDS once per workgroup.
//  bf16: entry (k, s2, lane) = 8 bf16: adj[k][a = lane&31][b = acc_row(8*s2 + j, lane>>5)], j = 0..7
//  f32 : entry (k, r, lane)  = adj[k][lane&31][acc_row(r, lane>>5)]
template <typename T>
__device__ __forceinline__ void build_adj_chain_table(char* tab, const float* adj, int K, int V, int tid, int nthreads, bool adj_t) {
    if constexpr (ET<T>::IS_BF16) {
        for (int e = tid; e < K * 2 * 64; e += nthreads) {
            const int lane = e & 63, s2 = (e >> 6) & 1, k = e >> 7;
            const int a = lane & 31, h = lane >> 5;
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int b = acc_row(8 * s2 + j, h);
                f[j] = (a < V && b < V) ? adj[adj_t ? ((size_t)k * V + b) * V + a : ((size_t)k * V + a) * V + b] : 0.f;
            }
            *(i32x4*)(tab + (size_t)e * 16) = pack16<bf16>(f);
        }
    } else {
        for (int e = tid; e < K * 16 * 64; e += nthreads) {
            const int lane = e & 63, r = (e >> 6) & 15, k = e >> 10;
            const int a = lane & 31, h = lane >> 5;
            const int b = acc_row(r, h);
            ((float*)tab)[e] = (a < V && b < V) ? adj[adj_t ? ((size_t)k * V + b) * V + a : ((size_t)k * V + a) * V + b] : 0.f;
        }
    }
}
template <typename T> static inline size_t adj_chain_table_bytes(int K) {
    return ET<T>::IS_BF16 ? (size_t)K * 2 * 64 * 16 : (size_t)K * 16 * 64 * 4;
}

// acc (rows = contraction index) -> B operand of the next MFMA; a_tab supplies the matching A operand
template <typename T>
__device__ __forceinline__ void chain_mma(f32x16& accy, const f32x16& accz, const char* tab_k, int lane) {
    if constexpr (ET<T>::IS_BF16) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            i32x4 bfr;
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = (int)pack_bf16x2(accz[8 * s2 + 2 * j], accz[8 * s2 + 2 * j + 1]);
            const i32x4 afr = *(const i32x4*)(tab_k + ((size_t)s2 * 64 + lane) * 16);
            accy = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8, afr), __builtin_bit_cast(s16x8, bfr), accy, 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float av = ((const float*)tab_k)[r * 64 + lane];
            accy = __builtin_amdgcn_mfma_f32_32x32x2f32(av, accz[r], accy, 0, 0, 0);
        }
    }
}

// NA = 32-byte k-steps per input row (Cin / KSTEP) as a compile-time constant, so the projection is a straight run
// of MFMAs with its LDS reads issued ahead; NA = 0 is the any-width fallback (runtime loop, operands fetched per
// step).  With NA <= 16 the next frame's rows are fetched into a second register set while this frame computes.
// ADD: the launch has an addend; without one the prefetch registers of the addend are not allocated, which lets the
// narrow forms (NA <= 4) run three workgroups per CU.
// NWV: waves (frames in flight) per workgroup.  Wide inputs whose weight image for two column tiles takes most of a
// CU's LDS run ONE workgroup of 8 waves instead of two of 4 with one column tile each: the same two waves per SIMD,
// but every frame fetched from L2 feeds twice the MFMAs.
template <typename T, int NA, bool ADD, int NWV>
__global__ __launch_bounds__(NWV * 64, NWV == 8 ? 1 : (NA >= 1 && NA <= 4 && (!ADD || ET<T>::IS_BF16)) ? 3 : 2) void gcn_fused_kernel(GcnFusedArgs a) {
    constexpr int NTHR = NWV * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ES = ET<T>::ES;
    constexpr int KSTEP = ET<T>::KSTEP;
    constexpr int VEC = ET<T>::VEC;
    constexpr int NAP = NA == 0 ? 1 : (NA < 16 ? NA : 16);     // fragments held per pass
    constexpr int NPASS = NA == 0 ? 0 : (NA + 15) / 16;
    constexpr bool PF = NPASS == 1 && NA <= 8;                  // prefetch the next frame (register budget; at 256
                                                                // channels a second set measured no gain without an addend)
    constexpr bool ONCE = NPASS == 1 && !PF;                    // fragments loaded once per frame, no prefetch
    // frames fetched ahead: two where a third register set does not cost a resident workgroup (with an addend it
    // would; at 128 input channels two measured 8 % slower than one)
    constexpr int DEPTH = (PF && ET<T>::IS_BF16 && NA <= 4 && !ADD) ? 2 : 1;
    constexpr int NADD = ET<T>::IS_BF16 ? 4 : 8;                // addend pieces prefetched per lane (covers V * NC / VEC <= 64 * NADD)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int V = a.V, K = a.K, Cin = a.Cin, Nout = a.Nout, NF = a.NF;
    const int NTW = a.ntw;                       // 32-column tiles per workgroup
    const int nbase = blockIdx.y * NTW * 32;
    const int RSW = Cin * ES + 16;               // LDS row stride of the weight image
    char* wl = smem;                             // [K][NTW*32][RSW]
    char* tab = smem + (size_t)K * NTW * 32 * RSW;
    const size_t tab_k_bytes = ET<T>::IS_BF16 ? 2 * 64 * 16 : 16 * 64 * 4;
    float* bl = (float*)(tab + (size_t)K * tab_k_bytes);        // [K][NTW*32] bias
    const int NC = NTW * 32;

    // stage weights (rows beyond Nout are zero), bias and the adjacency fragments
    {
        const int ppr = Cin * ES / 16;           // 16-byte pieces per row
        const int npieces = K * NC * ppr;
        for (int p = tid; p < npieces; p += NTHR) {
            const int seg = p % ppr, row = p / ppr;
            const int k = row / NC, nl = row - k * NC;
            const int n = nbase + nl;
            i32x4 v = zero16();
            if (n < Nout) v = *(const i32x4*)((const char*)a.w + (((size_t)k * Nout + n) * Cin) * ES + seg * 16);
            *(i32x4*)(wl + (size_t)row * RSW + seg * 16) = v;
        }
        for (int e = tid; e < K * NC; e += NTHR) {
            const int k = e / NC, n = nbase + (e - k * NC);
            bl[e] = (a.bias && n < Nout) ? a.bias[(size_t)k * Nout + n] : 0.f;
        }
        build_adj_chain_table<T>(tab, a.adj, K, V, tid, NTHR, a.adj_t != 0);
    }
    __syncthreads();

    // per-wave output staging: [32 joints][NTW*32 channels] in the storage dtype (+16 B pad per row)
    const int SRS = NC * ES + 16;
    char* stage = (char*)(bl + (size_t)K * NC) + (size_t)wave * 32 * SRS;
    const int ppr = NC / VEC;                    // 16-byte pieces per staged row (power of two)
    const int pshift = 31 - __builtin_clz(ppr);
    const int eseg = lane & (ppr - 1);           // this lane's fixed channel group in the store phase
    const int ncols = min(NC, Nout - nbase);
    const int npieces = V << pshift;
    const bool col_ok = eseg * VEC < ncols;
    const bool add_pf = ADD && npieces <= NADD * 64;
    float st1[VEC], st2[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { st1[i] = 0.f; st2[i] = 0.f; }
    const int brow = min(l31, V - 1);            // rows >= V duplicate the last joint; adj zeroes them
    auto frame_ptr = [&](int f) { return (const char*)a.in + (((size_t)f * V + brow) * Cin) * ES + h * 16; };

    // cur: this frame's fragments; pre: the register set that takes the fragments of frame fpre (DEPTH frames ahead);
    // nxt: the set of the next frame (pre itself at depth 1)
    auto process = [&](int f, i32x4 (&cur)[NAP], i32x4 (&pre)[NAP], int fpre, i32x4 (&nxt)[NAP]) {
        const char* ap = frame_ptr(f);
        if constexpr (PF) {
            // Issued on every path (past the last frame: the last frame again).  Behind a branch the loads would be
            // "maybe issued" to the compiler's wait-count pass, which then waits for the CURRENT fragments as if they
            // were the youngest loads in flight -- with the prefetch in flight that is a wait for the prefetch.
            const char* np = frame_ptr(min(fpre, NF - 1));
#pragma unroll
            for (int s = 0; s < NAP; ++s) pre[s] = *(const i32x4*)(np + s * 32);
        }
        if constexpr (ONCE) {
#pragma unroll
            for (int s = 0; s < NAP; ++s) cur[s] = *(const i32x4*)(ap + s * 32);
        }
        i32x4 addv[ADD ? NADD : 1];
        uint32_t addm[ADD ? NADD : 1];           // keep bits of the addend pieces (all ones without a mask)
        if (add_pf) {
            // branch-free for the same reason: lanes without a piece read a valid one and never use it
            const int esafe = col_ok ? eseg * VEC : 0;
#pragma unroll
            for (int i = 0; i < NADD; ++i) {
                const int p = min(lane + 64 * i, npieces - 1);
                const size_t o = ((size_t)f * V + (p >> pshift)) * Nout + nbase + esafe;
                addv[i] = *(const i32x4*)((const T*)a.add + o);
                addm[i] = 0xFFu;
                if (a.add_mask) addm[i] = a.add_mask[o / VEC];
            }
        }
        for (int nt = 0; nt < NTW; ++nt) {
            if (nbase + nt * 32 >= Nout) break;
            f32x16 accy;
#pragma unroll
            for (int r = 0; r < 16; ++r) accy[r] = 0.f;
            for (int k = 0; k < K; ++k) {
                const float bv = bl[k * NC + nt * 32 + l31];
                f32x16 accz;
#pragma unroll
                for (int r = 0; r < 16; ++r) accz[r] = bv;
                const char* wrow = wl + ((size_t)(k * NTW + nt) * 32 + l31) * RSW + h * 16;
                if constexpr (NA == 0) {
                    const int nar = Cin / KSTEP;
                    for (int s = 0; s < nar; ++s) {
                        const i32x4 afr = *(const i32x4*)(ap + s * 32);
                        const i32x4 bfr = *(const i32x4*)(wrow + s * 32);
                        mma_step<T>(accz, afr, bfr);
                    }
                } else {
#pragma unroll
                    for (int pass = 0; pass < NPASS; ++pass) {
                        if constexpr (NPASS > 1) {
#pragma unroll
                            for (int s = 0; s < NAP; ++s)
                                if (pass * 16 + s < NA) cur[s] = *(const i32x4*)(ap + (pass * 16 + s) * 32);
                        }
#pragma unroll
                        for (int s = 0; s < NAP; ++s) {
                            if (pass * 16 + s < NA) {
                                const i32x4 bfr = *(const i32x4*)(wrow + (pass * 16 + s) * 32);
                                mma_step<T>(accz, cur[s], bfr);
                            }
                        }
                    }
                }
                chain_mma<T>(accy, accz, tab + (size_t)k * tab_k_bytes, lane);
            }
            // lane = column n, register r = joint acc_row(r, h): park the tile in the wave's staging rows
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int aj = acc_row(r, h);
                if constexpr (ET<T>::IS_BF16)
                    *(uint16_t*)(stage + aj * SRS + (nt * 32 + l31) * 2) = (uint16_t)(pack_bf16x2(accy[r], 0.f) & 0xFFFFu);
                else
                    *(float*)(stage + aj * SRS + (nt * 32 + l31) * 4) = accy[r];
            }
        }
        __builtin_amdgcn_wave_barrier();
        if constexpr (PF) {
            // The one wait for the prefetch, a frame's arithmetic after its issue and BEFORE this frame's stores: the
            // next frame then starts on fragments the compiler knows to be complete, and no wait for them is also a
            // wait for stores issued just before.
#pragma unroll
            for (int s = 0; s < NAP; ++s) asm volatile("" : "+v"(nxt[s]));
        }
        // store phase: 16-byte pieces, whole channel rows coalesced; optional addend and statistics
        auto emit = [&](int p, bool have, i32x4 addval, uint32_t addbits) {
            const int row = p >> pshift;
            i32x4 v = *(const i32x4*)(stage + row * SRS + eseg * 16);
            const size_t o = ((size_t)f * V + row) * Nout + nbase + eseg * VEC;
            if (ADD || a.stats) {
                float fv[VEC];
                unpack16<T>(v, fv);
                if constexpr (ADD) {
                    float av[VEC];
                    unpack16<T>(have ? addval : *(const i32x4*)((const T*)a.add + o), av);
                    const uint32_t bits = have ? addbits : (a.add_mask ? (uint32_t)a.add_mask[o / VEC] : 0xFFu);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) fv[j] += (bits >> j) & 1u ? av[j] : 0.f;
                    v = pack16<T>(fv);
                    if constexpr (ET<T>::IS_BF16) unpack16<T>(v, fv);      // statistics of the stored values
                }
#pragma unroll
                for (int j = 0; j < VEC; ++j) { st1[j] += fv[j]; st2[j] += fv[j] * fv[j]; }
            }
            *(i32x4*)((T*)a.out + o) = v;
        };
        if (col_ok) {
            if (npieces <= NADD * 64) {
#pragma unroll
                for (int i = 0; i < NADD; ++i) {
                    const int p = lane + 64 * i;
                    if (p < npieces) emit(p, add_pf, addv[ADD ? i : 0], addm[ADD ? i : 0]);
                }
            } else {
                for (int p = lane; p < npieces; p += 64) emit(p, false, zero16(), 0xFFu);
            }
        }
        __builtin_amdgcn_wave_barrier();
    };

    {
        const int fstep = gridDim.x * NWV;
        int f = blockIdx.x * NWV + wave;
        i32x4 fa[NAP], fb[NAP], fc[NAP];
        if constexpr (PF) {
            if (f < NF) {
                const char* ap = frame_ptr(f);
#pragma unroll
                for (int s = 0; s < NAP; ++s) fa[s] = *(const i32x4*)(ap + s * 32);
                if constexpr (DEPTH == 2) {
                    const char* bp = frame_ptr(min(f + fstep, NF - 1));
#pragma unroll
                    for (int s = 0; s < NAP; ++s) fb[s] = *(const i32x4*)(bp + s * 32);
                }
            }
            // complete before the loop, as every later frame's fragments are when its turn comes (see process)
#pragma unroll
            for (int s = 0; s < NAP; ++s) asm volatile("" : "+v"(fa[s]));
        }
        while (f < NF) {
            if constexpr (DEPTH == 2) {
                process(f, fa, fc, f + 2 * fstep, fb);
                f += fstep;
                if (f >= NF) break;
                process(f, fb, fa, f + 2 * fstep, fc);
                f += fstep;
                if (f >= NF) break;
                process(f, fc, fb, f + 2 * fstep, fa);
                f += fstep;
            } else {
                process(f, fa, fb, f + fstep, fb);
                f += fstep;
                if (f >= NF) break;
                if constexpr (PF) process(f, fb, fa, f + fstep, fa);
                else process(f, fa, fb, f + fstep, fb);
                f += fstep;
            }
        }
    }
    if (a.stats) {
        // lanes with equal eseg hold partial sums of the same channels: fold them with shuffles, then the four
        // waves through LDS, so that each workgroup issues ONE atomic per channel with consecutive lanes on
        // consecutive addresses (scattered / repeated double atomics serialise at the memory side)
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            for (int d = 32; d >= ppr; d >>= 1) {
                st1[j] += __shfl_xor(st1[j], d);
                st2[j] += __shfl_xor(st2[j], d);
            }
        }
        __syncthreads();                               // every wave is done with its staging rows
        float* red = (float*)(bl + (size_t)K * NC);    // [NWV waves][2][NC]
        if (lane < ppr) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                red[(wave * 2 + 0) * NC + eseg * VEC + j] = st1[j];
                red[(wave * 2 + 1) * NC + eseg * VEC + j] = st2[j];
            }
        }
        __syncthreads();
        for (int t = tid; t < 2 * NC; t += NTHR) {
            const int which = t / NC, c = t - which * NC;
            if (c < ncols) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < NWV; ++w) v += red[(w * 2 + which) * NC + c];
                atomicAdd(&a.stats[which * Nout + nbase + c], (double)v);
            }
        }
    }
}

template <typename T, int NA, bool ADD, int NWV>
static int gcn_fused_launch3(GcnFusedArgs& a, dim3 grid, size_t lds, hipStream_t stream) {
    auto kern = gcn_fused_kernel<T, NA, ADD, NWV>;
    static std::atomic<uint64_t> lds_ok{0};
    if (stgcn_allow_big_lds((const void*)kern, lds_ok) != STGCN_OK) return STGCN_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, grid, dim3(NWV * 64), lds, stream, a);
    return hipGetLastError() == hipSuccess ? STGCN_OK : STGCN_ERR_LAUNCH;
}
template <typename T, int NA, bool ADD>
static int gcn_fused_launch2(GcnFusedArgs& a, dim3 grid, size_t lds, int nwv, hipStream_t stream) {
    if constexpr (NA >= 8) {
        if (nwv == 8) return gcn_fused_launch3<T, NA, ADD, 8>(a, grid, lds, stream);
    }
    return gcn_fused_launch3<T, NA, ADD, 4>(a, grid, lds, stream);
}
template <typename T, int NA>
static int gcn_fused_launch(GcnFusedArgs& a, dim3 grid, size_t lds, int nwv, hipStream_t stream) {
    return a.add ? gcn_fused_launch2<T, NA, true>(a, grid, lds, nwv, stream) : gcn_fused_launch2<T, NA, false>(a, grid, lds, nwv, stream);
}

template <typename T>
static int gcn_fused_run(GcnFusedArgs& a, hipStream_t stream) {
    constexpr int ES = ET<T>::ES, KSTEP = ET<T>::KSTEP;
    if (a.Cin % KSTEP != 0 || a.Cin <= 0 || a.Cin > 32 * KSTEP || a.Nout <= 0 || a.Nout % ET<T>::VEC != 0 || a.K < 1 || a.K > 8 || a.V < 1 || a.V > 32 || a.NF < 1) return STGCN_ERR_ARG;
    const size_t rsw = (size_t)a.Cin * ES + 16;
    // column tiles per workgroup (1, 2 or 4: the store phase wants a power-of-two piece count per row): the widest
    // that keeps the LDS image within half a CU, so two workgroups (two waves per SIMD) stay resident
    auto lds_for = [&](int n, int nwv) {
        return (size_t)a.K * n * 32 * rsw + adj_chain_table_bytes<T>(a.K) + (size_t)a.K * n * 32 * 4 + (size_t)nwv * 32 * (n * 32 * ES + 16);
    };
    int ntw = 4, nwv = 4;
    while (ntw > 1 && (lds_for(ntw, 4) > 80 * 1024 || (ntw / 2) * 32 >= a.Nout)) ntw /= 2;
    const int na = a.Cin / KSTEP;
    // one column tile per 4-wave workgroup is all that half a CU's LDS holds: take the whole CU with 8 waves and two
    if (ntw == 1 && a.Nout > 32 && na >= 8 && (na & (na - 1)) == 0 && lds_for(2, 8) <= 160 * 1024) { ntw = 2; nwv = 8; }
    const size_t lds = lds_for(ntw, nwv);
    if (lds > 160 * 1024) return STGCN_ERR_ARG;
    a.ntw = ntw;
    const int gy = ceil_div_i(a.Nout, ntw * 32);
    // persistent waves: as many workgroups as stay resident (2 per CU when the LDS image allows it)
    const int per_cu = ((!a.add || ET<T>::IS_BF16) && na >= 1 && na <= 4 && 3 * lds <= 160 * 1024) ? 3 : lds <= 80 * 1024 ? 2 : 1;
    const int resident = per_cu * stgcn_num_cus();
    int gx = max(1, resident / gy);
    if (gx > ceil_div_i(a.NF, nwv)) gx = ceil_div_i(a.NF, nwv);
    const dim3 grid((unsigned)gx, (unsigned)gy);
    switch (a.Cin / KSTEP) {
        case 1: return gcn_fused_launch<T, 1>(a, grid, lds, nwv, stream);
        case 2: return gcn_fused_launch<T, 2>(a, grid, lds, nwv, stream);
        case 4: return gcn_fused_launch<T, 4>(a, grid, lds, nwv, stream);
        case 8: return gcn_fused_launch<T, 8>(a, grid, lds, nwv, stream);
        case 16: return gcn_fused_launch<T, 16>(a, grid, lds, nwv, stream);
        case 32: return gcn_fused_launch<T, 32>(a, grid, lds, nwv, stream);
        default: return gcn_fused_launch<T, 0>(a, grid, lds, nwv, stream);
    }
}

extern "C" int stgcn_gcn_fused(int dtype, const void* in, const void* w_packed, const float* adj, int adj_transposed, const float* bias,
                               void* out, const void* add, const void* add_mask, double* stats,
                               int NF, int V, int Cin, int Nout, int K, void* stream) {
    if (!in || !w_packed || !adj || !out) return STGCN_ERR_ARG;
    GcnFusedArgs a;
    a.in = in; a.w = w_packed; a.adj = adj; a.bias = bias; a.out = out; a.add = add; a.add_mask = (const uint16_t*)add_mask; a.stats = stats;
    a.NF = NF; a.V = V; a.Cin = Cin; a.Nout = Nout; a.K = K; a.ntw = 1; a.adj_t = adj_transposed;
    if (dtype == STGCN_BF16) return gcn_fused_run<bf16>(a, (hipStream_t)stream);
    if (dtype == STGCN_F32) return gcn_fused_run<float>(a, (hipStream_t)stream);
    return STGCN_ERR_ARG;
}

// ------------------------------------------------------------------------------------------
// gcn_agrad
struct GcnAgradArgs {
    const void* x; const void* w; const float* bias; const void* dy; float* slab;
    int NF, V, Cin, Cout, K, ntw;
};

// NA as in gcn_fused_kernel.  Each wave keeps the K 32x32 partial dA tiles in accumulators over all its frames;
// the waves' tiles are summed through LDS at the end and the workgroup writes one partial to the slab.  NWV (waves
// per workgroup) as in gcn_fused_kernel.
// RAG: Cout is not a multiple of 32 (the last column tile is ragged).
template <typename T, int NA, int NWV, bool RAG>
__global__ __launch_bounds__(NWV * 64, NWV == 8 ? 1 : 2) void gcn_agrad_kernel(GcnAgradArgs a) {
    constexpr int NTHR = NWV * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ES = ET<T>::ES;
    constexpr int KSTEP = ET<T>::KSTEP;
    constexpr bool BF = ET<T>::IS_BF16;
    constexpr int NAP = NA == 0 ? 1 : (NA < 16 ? NA : 16);
    constexpr int NPASS = NA == 0 ? 0 : (NA + 15) / 16;
    constexpr bool PF = NPASS == 1 && NA <= 8;
    constexpr bool ONCE = NPASS == 1 && !PF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int V = a.V, K = a.K, Cin = a.Cin, Cout = a.Cout, NF = a.NF;
    const int NTW = a.ntw;
    const int NC = NTW * 32;
    const int cbase = blockIdx.y * NC;
    const int RSW = Cin * ES + 16;
    char* wl = smem;                             // [K][NC][RSW]
    float* bl = (float*)(smem + (size_t)K * NC * RSW);          // [K][NC] bias
    {
        const int ppr = Cin * ES / 16;
        const int npieces = K * NC * ppr;
        for (int p = tid; p < npieces; p += NTHR) {
            const int seg = p % ppr, row = p / ppr;
            const int k = row / NC, nl = row - k * NC;
            const int co = cbase + nl;
            i32x4 v = zero16();
            if (co < Cout) v = *(const i32x4*)((const char*)a.w + (((size_t)k * Cout + co) * Cin) * ES + seg * 16);
            *(i32x4*)(wl + (size_t)row * RSW + seg * 16) = v;
        }
        for (int e = tid; e < K * NC; e += NTHR) {
            const int k = e / NC, co = cbase + (e - k * NC);
            bl[e] = (a.bias && co < Cout) ? a.bias[(size_t)k * Cout + co] : 0.f;
        }
    }
    __syncthreads();

    f32x16 accd[GCN_KMAX];
#pragma unroll
    for (int k = 0; k < GCN_KMAX; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) accd[k][r] = 0.f;
    const int jrow = min(l31, V - 1);
    auto frame_ptr = [&](int f) { return (const char*)a.x + (((size_t)f * V + jrow) * Cin) * ES + h * 16; };
    typedef typename std::conditional<BF, i32x2, f32x4>::type dyraw_t;   // 4 consecutive channels of dy

    auto process = [&](int f, i32x4 (&cur)[NAP], i32x4 (&nxt)[NAP], int fnext) {
        const char* xp = frame_ptr(f);
        if constexpr (PF) {
            if (fnext < NF) {
                const char* np = frame_ptr(fnext);
#pragma unroll
                for (int s = 0; s < NAP; ++s) nxt[s] = *(const i32x4*)(np + s * 32);
            }
        }
        if constexpr (ONCE) {
#pragma unroll
            for (int s = 0; s < NAP; ++s) cur[s] = *(const i32x4*)(xp + s * 32);
        }
        // dy operand: the element for accumulator register r is dy[f, w = lane&31, c0 + acc_row(r, h)]; registers
        // 4g..4g+3 are the 4 consecutive channels c0 + 8g + 4h + (0..3)
        const char* dyp = (const char*)a.dy + (((size_t)f * V + jrow) * Cout) * ES;
        dyraw_t dyr[4][4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            if (nt < NTW && cbase + nt * 32 < Cout) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = cbase + nt * 32 + 8 * g + 4 * h;
                    if constexpr (RAG) {
                        // Cout is a multiple of 8, not of 32: the last tile's channels past Cout meet zero weight
                        // rows, but must not be read -- past the last row they lie outside the tensor
                        dyraw_t v = {};
                        if (c < Cout) v = *(const dyraw_t*)(dyp + (size_t)c * ES);
                        dyr[nt][g] = v;
                    } else {
                        dyr[nt][g] = *(const dyraw_t*)(dyp + (size_t)c * ES);
                    }
                }
            }
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            if (nt < NTW && cbase + nt * 32 < Cout) {
#pragma unroll
                for (int k = 0; k < GCN_KMAX; ++k) {
                    if (k < K) {
                        // z^T tile: rows = co (registers), cols = joint v (lane)
                        f32x16 accz;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 b4 = *(const f32x4*)(bl + k * NC + nt * 32 + 8 * g + 4 * h);
#pragma unroll
                            for (int j = 0; j < 4; ++j) accz[4 * g + j] = b4[j];
                        }
                        const char* wrow = wl + ((size_t)(k * NTW + nt) * 32 + l31) * RSW + h * 16;
                        if constexpr (NA == 0) {
                            const int nar = Cin / KSTEP;
                            for (int s = 0; s < nar; ++s) {
                                const i32x4 xfr = *(const i32x4*)(xp + s * 32);
                                const i32x4 wfr = *(const i32x4*)(wrow + s * 32);
                                mma_step<T>(accz, wfr, xfr);
                            }
                        } else {
#pragma unroll
                            for (int pass = 0; pass < NPASS; ++pass) {
                                if constexpr (NPASS > 1) {
#pragma unroll
                                    for (int s = 0; s < NAP; ++s)
                                        if (pass * 16 + s < NA) cur[s] = *(const i32x4*)(xp + (pass * 16 + s) * 32);
                                }
#pragma unroll
                                for (int s = 0; s < NAP; ++s) {
                                    if (pass * 16 + s < NA) {
                                        const i32x4 wfr = *(const i32x4*)(wrow + (pass * 16 + s) * 32);
                                        mma_step<T>(accz, wfr, cur[s]);
                                    }
                                }
                            }
                        }
                        // dA_k[v][w] += sum_co z^T[co][v] * dy[w][co]   (z^T as the A operand)
                        if constexpr (BF) {
#pragma unroll
                            for (int s2 = 0; s2 < 2; ++s2) {
                                i32x4 afr, bfr;
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    afr[j] = (int)pack_bf16x2(accz[8 * s2 + 2 * j], accz[8 * s2 + 2 * j + 1]);
                                    bfr[j] = dyr[nt][2 * s2 + (j >> 1)][j & 1];
                                }
                                accd[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8, afr), __builtin_bit_cast(s16x8, bfr), accd[k], 0, 0, 0);
                            }
                        } else {
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                accd[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(accz[r], dyr[nt][r >> 2][r & 3], accd[k], 0, 0, 0);
                        }
                    }
                }
            }
        }
    };

    {
        const int fstep = gridDim.x * NWV;
        int f = blockIdx.x * NWV + wave;
        i32x4 fa[NAP], fb[NAP];
        if constexpr (PF) {
            if (f < NF) {
                const char* xp = frame_ptr(f);
#pragma unroll
                for (int s = 0; s < NAP; ++s) fa[s] = *(const i32x4*)(xp + s * 32);
            }
        }
        while (f < NF) {
            process(f, fa, fb, f + fstep);
            f += fstep;
            if (f >= NF) break;
            if constexpr (PF) process(f, fb, fa, f + fstep);
            else process(f, fa, fb, f + fstep);
            f += fstep;
        }
    }
    // accd[k]: row = v (registers), col = w (lane).  Sum the waves through LDS, one k at a time.
    const size_t part = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    float* red = (float*)smem;                   // [NWV][16][64] floats (weights are dead now)
#pragma unroll
    for (int k = 0; k < GCN_KMAX; ++k) {
        if (k < K) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = accd[k][r];
            __syncthreads();
            for (int e = tid; e < 16 * 64; e += NTHR) {
                const int r = e >> 6, ln = e & 63;
                const int v = acc_row(r, ln >> 5), w = ln & 31;
                if (v < V && w < V) {
                    float t = 0.f;
#pragma unroll
                    for (int wv = 0; wv < NWV; ++wv) t += red[wv * 16 * 64 + e];
                    a.slab[((part * K + k) * V + v) * V + w] = t;
                }
            }
        }
    }
}

// The fixed-order sum of partials [nparts][n1 + n2] into out1[n1] and out2[n2] (out2 may be null): launched at once,
// or left on the caller's list.  `wave`: few elements and many parts (dA: 3*V*V sums of up to 512 partials), one wave
// per element.
static int gcn_sum(StgcnSumList* list, const float* slab, float* out1, size_t n1, float* out2, size_t n2, int nparts, bool wave, hipStream_t stream) {
    StgcnPendingSum s{};
    s.slab = slab; s.out1 = out1; s.out2 = out2; s.n1 = n1; s.n2 = n2; s.nparts = nparts;
    s.kind = wave ? STGCN_SUM_WAVE : STGCN_SUM_PLAIN;
    return stgcn_sum_or_defer(list, s, stream);
}

template <typename T, int NA, int NWV, bool RAG>
static int gcn_agrad_launch3(GcnAgradArgs& a, dim3 grid, size_t lds, hipStream_t stream) {
    auto kern = gcn_agrad_kernel<T, NA, NWV, RAG>;
    static std::atomic<uint64_t> lds_ok{0};
    if (stgcn_allow_big_lds((const void*)kern, lds_ok) != STGCN_OK) return STGCN_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, grid, dim3(NWV * 64), lds, stream, a);
    return hipGetLastError() == hipSuccess ? STGCN_OK : STGCN_ERR_LAUNCH;
}
template <typename T, int NA, int NWV>
static int gcn_agrad_launch2(GcnAgradArgs& a, dim3 grid, size_t lds, hipStream_t stream) {
    if (a.Cout % 32 != 0) return gcn_agrad_launch3<T, NA, NWV, true>(a, grid, lds, stream);
    return gcn_agrad_launch3<T, NA, NWV, false>(a, grid, lds, stream);
}
template <typename T, int NA>
static int gcn_agrad_launch(GcnAgradArgs& a, dim3 grid, size_t lds, int nwv, hipStream_t stream) {
    if constexpr (NA >= 8) {
        if (nwv == 8) return gcn_agrad_launch2<T, NA, 8>(a, grid, lds, stream);
    }
    return gcn_agrad_launch2<T, NA, 4>(a, grid, lds, stream);
}

template <typename T>
static int gcn_agrad_run(GcnAgradArgs& a, float* dA, size_t ws_bytes, StgcnSumList* list, hipStream_t stream) {
    constexpr int ES = ET<T>::ES, KSTEP = ET<T>::KSTEP;
    if (a.Cin % KSTEP != 0 || a.Cin <= 0 || a.Cin > 32 * KSTEP || a.Cout % 8 != 0 || a.K < 1 || a.K > GCN_KMAX || a.V < 1 || a.V > 32 || a.NF < 1) return STGCN_ERR_ARG;
    const size_t rsw = (size_t)a.Cin * ES + 16;
    int ntw = 4;
    while (ntw > 1 && ((size_t)a.K * ntw * 32 * rsw > 72 * 1024 || (ntw - 1) * 32 >= a.Cout)) --ntw;
    const int na = a.Cin / KSTEP;
    int nwv = 4;
    // Wide inputs: half a CU's LDS holds few column tiles, and every tile re-reads the frame from L2.  One 8-wave
    // workgroup with the whole CU's LDS (the same two waves per SIMD) takes twice the tiles per frame fetched.
    if (na >= 8 && (na & (na - 1)) == 0 && na <= 16) {
        int ntw8 = 4;                                // 4 or 2: equal column tiles for the usual power-of-two widths
        while (ntw8 > 1 && ((size_t)a.K * ntw8 * 32 * (rsw + 4) > 150 * 1024 || ntw8 * 32 > a.Cout)) ntw8 /= 2;
        if (ntw8 > ntw) { ntw = ntw8; nwv = 8; }
    }
    size_t lds = (size_t)a.K * ntw * 32 * rsw + (size_t)a.K * ntw * 32 * 4;
    if (lds < (size_t)nwv * 4096) lds = (size_t)nwv * 4096;   // the final cross-wave reduction uses 4 KiB per wave
    if (lds > 160 * 1024) return STGCN_ERR_ARG;
    a.ntw = ntw;
    const int gy = ceil_div_i(a.Cout, ntw * 32);
    const int resident = (lds <= 80 * 1024 ? 2 : 1) * stgcn_num_cus();
    int gx = max(1, resident / gy);
    if (gx > ceil_div_i(a.NF, nwv)) gx = ceil_div_i(a.NF, nwv);
    const size_t n = (size_t)a.K * a.V * a.V;
    if ((size_t)gx * gy * n * sizeof(float) > ws_bytes) return STGCN_ERR_WORKSPACE;
    const dim3 grid((unsigned)gx, (unsigned)gy);
    int rc;
    switch (a.Cin / KSTEP) {
        case 1: rc = gcn_agrad_launch<T, 1>(a, grid, lds, nwv, stream); break;
        case 2: rc = gcn_agrad_launch<T, 2>(a, grid, lds, nwv, stream); break;
        case 4: rc = gcn_agrad_launch<T, 4>(a, grid, lds, nwv, stream); break;
        case 8: rc = gcn_agrad_launch<T, 8>(a, grid, lds, nwv, stream); break;
        case 16: rc = gcn_agrad_launch<T, 16>(a, grid, lds, nwv, stream); break;
        case 32: rc = gcn_agrad_launch<T, 32>(a, grid, lds, nwv, stream); break;
        default: rc = gcn_agrad_launch<T, 0>(a, grid, lds, nwv, stream); break;
    }
    if (rc != STGCN_OK) return rc;
    return gcn_sum(list, a.slab, dA, n, nullptr, 0, gx * gy, true, stream);
}

static int gcn_agrad_any(int dtype, const void* x, const void* w_packed, const float* bias, const void* dy,
                         float* dA, int NF, int V, int Cin, int Cout, int K,
                         void* workspace, size_t workspace_bytes, StgcnSumList* list, hipStream_t stream) {
    if (!x || !w_packed || !dy || !dA || !workspace) return STGCN_ERR_ARG;
    GcnAgradArgs a;
    a.x = x; a.w = w_packed; a.bias = bias; a.dy = dy; a.slab = (float*)workspace;
    a.NF = NF; a.V = V; a.Cin = Cin; a.Cout = Cout; a.K = K; a.ntw = 1;
    if (dtype == STGCN_BF16) return gcn_agrad_run<bf16>(a, dA, workspace_bytes, list, stream);
    if (dtype == STGCN_F32) return gcn_agrad_run<float>(a, dA, workspace_bytes, list, stream);
    return STGCN_ERR_ARG;
}

extern "C" int stgcn_gcn_agrad(int dtype, const void* x, const void* w_packed, const float* bias, const void* dy,
                               float* dA, int NF, int V, int Cin, int Cout, int K,
                               void* workspace, size_t workspace_bytes, void* stream) {
    return gcn_agrad_any(dtype, x, w_packed, bias, dy, dA, NF, V, Cin, Cout, K, workspace, workspace_bytes, nullptr, (hipStream_t)stream);
}

// The same partial sums with the final sum left on the caller's list (see stgcn_conv_wgrad_deferred).
extern "C" int stgcn_gcn_agrad_deferred(int dtype, const void* x, const void* w_packed, const float* bias, const void* dy,
                                        float* dA, int NF, int V, int Cin, int Cout, int K,
                                        void* workspace, size_t workspace_bytes, void* sum_list, size_t* workspace_used, void* stream) {
    StgcnSumList* list = (StgcnSumList*)sum_list;
    if (!list || !workspace_used) return STGCN_ERR_ARG;
    const int first = list->n;
    const int rc = gcn_agrad_any(dtype, x, w_packed, bias, dy, dA, NF, V, Cin, Cout, K, workspace, workspace_bytes, list, (hipStream_t)stream);
    if (rc != STGCN_OK) { list->n = first; return rc; }
    *workspace_used = stgcn_sum_list_extent(list, first, workspace);
    return STGCN_OK;
}

// The same computation for callers that may not allocate (code under HIP graph capture, serving loops, harnesses
// that forbid hipMalloc) and therefore cannot bring a workspace: the partial-sum slab lives in a file-scope device
// array, one copy per GPU, held until the process ends.  Host code never passes the array's NAME to a kernel (on the
// host it is a stand-in, not a device address): its address on the current device is asked for once per device and
// cached; kernels receive that pointer.  Every call on a GPU shares the GPU's array, so calls that overlap on one
// GPU (two streams, two host threads) race; calls ordered on one stream do not.  Nothing needs clearing between
// calls: every partial the final sum reads is written in full by the workgroup that owns it.
constexpr size_t GCN_SCRATCH_BYTES = 4u << 20;
static __device__ __align__(256) unsigned char gcn_agrad_scratch[GCN_SCRATCH_BYTES];
static std::atomic<void*> g_gcn_scratch_ptr[STGCN_MAX_DEV];

static int gcn_scratch_on_current_device(void** out) {
    const int d = stgcn_cur_device();
    if (d < 0 || d >= STGCN_MAX_DEV) return STGCN_ERR_ARG;
    void* p = g_gcn_scratch_ptr[d].load(std::memory_order_acquire);
    if (!p) {
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(gcn_agrad_scratch)) != hipSuccess || !p) return STGCN_ERR_LAUNCH;
        g_gcn_scratch_ptr[d].store(p, std::memory_order_release);      // two first callers store the same address
    }
    *out = p;
    return STGCN_OK;
}

extern "C" int stgcn_gcn_agrad_noalloc(int dtype, const void* x, const void* w_packed, const float* bias, const void* dy,
                                       float* dA, int NF, int V, int Cin, int Cout, int K, void* stream) {
    if (!x || !w_packed || !dy || !dA) return STGCN_ERR_ARG;
    void* scratch = nullptr;
    int rc = gcn_scratch_on_current_device(&scratch);
    if (rc != STGCN_OK) return rc;
    GcnAgradArgs a;
    a.x = x; a.w = w_packed; a.bias = bias; a.dy = dy; a.slab = (float*)scratch;
    a.NF = NF; a.V = V; a.Cin = Cin; a.Cout = Cout; a.K = K; a.ntw = 1;
    if (dtype == STGCN_BF16) rc = gcn_agrad_run<bf16>(a, dA, GCN_SCRATCH_BYTES, nullptr, (hipStream_t)stream);
    else if (dtype == STGCN_F32) rc = gcn_agrad_run<float>(a, dA, GCN_SCRATCH_BYTES, nullptr, (hipStream_t)stream);
    else return STGCN_ERR_ARG;
    return rc == STGCN_ERR_WORKSPACE ? STGCN_ERR_SCRATCH : rc;       // partials of this shape exceed the static array
}

// ------------------------------------------------------------------------------------------
// gcn_wgrad
struct GcnWgradArgs {
    const void* x; const void* dy; const float* adj; float* slabW; float* slabB;
    int NF, V, Cin, Cout, K, cig, iters;
};

// LDS per wave: dy image [32][DRS] (32 channels of one co tile) and x image [32][XRS] (cig channels);
// rows >= V stay zero.  Then the adjacency A-operand table (natural contraction order).
template <typename T>
__global__ __launch_bounds__(256) void gcn_wgrad_kernel(GcnWgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ES = ET<T>::ES;
    constexpr bool BF = ET<T>::IS_BF16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int V = a.V, K = a.K, Cin = a.Cin, Cout = a.Cout, CIG = a.cig;
    const int DRS = 32 * ES + 16, XRS = CIG * ES + 16;
    const int img_bytes = 32 * DRS + 32 * XRS;
    char* dimg = smem + (size_t)wave * img_bytes;
    char* ximg = dimg + 32 * DRS;
    char* tab = smem + (size_t)4 * img_bytes;    // bf16: [K][2][64] x 16 B ; f32: [K][32(w)][32(v)] floats
    const int c0 = blockIdx.y * 32;              // co tile
    const int i0 = blockIdx.z * CIG;             // ci group
    const int CIT = (CIG + 31) / 32;             // 32-wide ci tiles in the group (1 or 2)

    for (int p = tid; p < 4 * img_bytes / 16; p += 256) *(i32x4*)(smem + (size_t)p * 16) = zero16();
    if constexpr (BF) {
        for (int e = tid; e < K * 2 * 64; e += 256) {
            const int ln = e & 63, s2 = (e >> 6) & 1, k = e >> 7;
            const int v = ln & 31, hh = ln >> 5;
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int w = 16 * s2 + 8 * hh + j;
                f[j] = (v < V && w < V) ? a.adj[((size_t)k * V + v) * V + w] : 0.f;
            }
            *(i32x4*)(tab + (size_t)e * 16) = pack16<bf16>(f);
        }
    } else {
        for (int e = tid; e < K * 32 * 32; e += 256) {
            const int v = e & 31, w = (e >> 5) & 31, k = e >> 10;
            ((float*)tab)[e] = (v < V && w < V) ? a.adj[((size_t)k * V + v) * V + w] : 0.f;
        }
    }
    __syncthreads();

    f32x16 accw[GCN_KMAX][2];
    float dbacc[GCN_KMAX];
#pragma unroll
    for (int k = 0; k < GCN_KMAX; ++k) {
        dbacc[k] = 0.f;
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r) accw[k][it][r] = 0.f;
    }
    const int q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = (lane >> 4) & 1;
    const int dppr = 32 * ES / 16, xppr = CIG * ES / 16;

    // The images are private to a wave, so no workgroup barrier is needed inside the frame loop; the next frame's
    // pieces are fetched into registers while this frame's MFMAs run.
    constexpr int DPN = ES;                      // dy pieces per lane: 32 rows x (32 channels * ES / 16) / 64
    constexpr int XPN = 2 * ES;                  // x pieces per lane for a 64-channel group
    i32x4 dreg[DPN], xreg[XPN];
    auto fetch = [&](int f) {
        const bool fok = f < a.NF;
#pragma unroll
        for (int i = 0; i < DPN; ++i) {
            const int p = lane + 64 * i;
            const int row = p / dppr, seg = p - row * dppr;
            dreg[i] = zero16();
            if (fok && row < V && c0 + seg * (16 / ES) < Cout)
                dreg[i] = *(const i32x4*)((const char*)a.dy + (((size_t)f * V + row) * Cout + c0) * ES + seg * 16);
        }
#pragma unroll
        for (int i = 0; i < XPN; ++i) {
            const int p = lane + 64 * i;
            const int row = p / xppr, seg = p - row * xppr;
            xreg[i] = zero16();
            if (fok && row < V) xreg[i] = *(const i32x4*)((const char*)a.x + (((size_t)f * V + row) * Cin + i0) * ES + seg * 16);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < DPN; ++i) {
            const int p = lane + 64 * i;
            const int row = p / dppr, seg = p - row * dppr;
            if (row < V) *(i32x4*)(dimg + row * DRS + seg * 16) = dreg[i];
        }
#pragma unroll
        for (int i = 0; i < XPN; ++i) {
            const int p = lane + 64 * i;
            const int row = p / xppr, seg = p - row * xppr;
            if (row < V) *(i32x4*)(ximg + row * XRS + seg * 16) = xreg[i];
        }
    };
    fetch(blockIdx.x * 4 + wave);
    for (int it0 = 0; it0 < a.iters; ++it0) {
        __builtin_amdgcn_wave_barrier();         // this wave's reads of the previous images are issued (LDS is in order per wave)
        commit();
        __builtin_amdgcn_wave_barrier();
        if (it0 + 1 < a.iters) fetch(((it0 + 1) * gridDim.x + blockIdx.x) * 4 + wave);
#pragma unroll
        for (int k = 0; k < GCN_KMAX; ++k) {
            if (k < K) {
                // dz = A_eff[k] (v x w) * dy (w x co): rows = v (registers), cols = co (lane)
                f32x16 dz;
#pragma unroll
                for (int r = 0; r < 16; ++r) dz[r] = 0.f;
                if constexpr (BF) {
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const i32x4 afr = *(const i32x4*)(tab + (((size_t)k * 2 + s2) * 64 + lane) * 16);
                        const int row0 = 16 * s2 + 8 * h + q4;
                        const int col = (16 * g1 + 4 * p4) * 2;
                        const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(dimg + row0 * DRS + col));
                        const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(dimg + (row0 + 4) * DRS + col));
                        const s16x8 bfr = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                        dz = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8, afr), bfr, dz, 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < 16; ++s) {
                        const int w = 2 * s + h;
                        const float av = ((const float*)tab)[((size_t)k * 32 + w) * 32 + l31];
                        const float bv = *(const float*)(dimg + w * DRS + l31 * 4);
                        dz = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, dz, 0, 0, 0);
                    }
                }
                float sdz = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) sdz += dz[r];
                dbacc[k] += sdz;
                // dW_k[co][ci] += sum_v dz[v][co] * x[v][ci]   (dz as the A operand, contraction in acc-row order)
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    if (it < CIT) {
                        if constexpr (BF) {
#pragma unroll
                            for (int s2 = 0; s2 < 2; ++s2) {
                                i32x4 afr;
#pragma unroll
                                for (int j = 0; j < 4; ++j) afr[j] = (int)pack_bf16x2(dz[8 * s2 + 2 * j], dz[8 * s2 + 2 * j + 1]);
                                const int row0 = 16 * s2 + 4 * h + q4;
                                const int col = (it * 32 + 16 * g1 + 4 * p4) * 2;
                                const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ximg + row0 * XRS + col));
                                const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ximg + (row0 + 8) * XRS + col));
                                const s16x8 bfr = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                                accw[k][it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8, afr), bfr, accw[k][it], 0, 0, 0);
                            }
                        } else {
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const float bv = *(const float*)(ximg + acc_row(r, h) * XRS + (it * 32 + l31) * 4);
                                accw[k][it] = __builtin_amdgcn_mfma_f32_32x32x2f32(dz[r], bv, accw[k][it], 0, 0, 0);
                            }
                        }
                    }
                }
            }
        }
    }

    // reduce the 4 waves through LDS (one 32x32 tile at a time), then write this workgroup's partial
    float* red = (float*)smem;                   // [4][16][64] floats = 16 KiB (images are free now)
#pragma unroll
    for (int k = 0; k < GCN_KMAX; ++k) {
        if (k < K) {
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                if (it < CIT) {
                    __syncthreads();
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = accw[k][it][r];
                    __syncthreads();
                    for (int e = tid; e < 16 * 64; e += 256) {
                        const float s = red[e] + red[16 * 64 + e] + red[2 * 16 * 64 + e] + red[3 * 16 * 64 + e];
                        const int r = e >> 6, ln = e & 63;
                        const int co = c0 + acc_row(r, ln >> 5);
                        const int ci = i0 + it * 32 + (ln & 31);
                        if (co < Cout && ci < Cin && (it * 32 + (ln & 31)) < CIG)
                            a.slabW[(((size_t)blockIdx.x * K + k) * Cout + co) * Cin + ci] = s;
                    }
                }
            }
            __syncthreads();
            const float sb = dbacc[k] + __shfl_xor(dbacc[k], 32);
            if (h == 0) red[wave * 32 + l31] = sb;
            __syncthreads();
            if (tid < 32 && blockIdx.z == 0 && c0 + tid < Cout)
                a.slabB[((size_t)blockIdx.x * K + k) * Cout + c0 + tid] = red[tid] + red[32 + tid] + red[64 + tid] + red[96 + tid];
        }
    }
}

// ------------------------------------------------------------------------------------------
// gcn_wgrad, bf16 wide form.  A workgroup owns a 64-channel slice of Cout and a contiguous range of frames; it
// stages batches of FB frames (dy slice and all of x, 32 padded rows per frame) in LDS once and every wave reads
// them from there.  A wave owns one (k, 32-wide co tile, group of `nci` 32-wide ci tiles): per frame it forms
// dz = A_eff[k] * dy (2 MFMAs), keeps it in registers as the A operand, and adds dz^T * x into its nci accumulator
// tiles (2 MFMAs each).  No cross-wave reduction: every (k, co, ci) element lives in exactly one wave.  The next
// batch is fetched into registers while the current one is consumed.
struct GcnWgradWideArgs {
    const void* x; const void* dy; const float* adj; float* slabW; float* slabB;
    int NF, V, Cin, Cout, K, cit, nci, FB, frames_per_wg;
    size_t part_stride;      // floats per workgroup partial: K*Cout*Cin weight sums, then K*Cout bias sums
};

constexpr int GW_MAXD = 2, GW_MAXX = 4;          // staged 16-byte pieces per thread per batch (dy, x)

template <int NCI>
__global__ __launch_bounds__(768) void gcn_wgrad_wide_kernel(GcnWgradWideArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NT = blockDim.x;
    const int l31 = lane & 31, h = lane >> 5;
    const int V = a.V, K = a.K, Cin = a.Cin, Cout = a.Cout, FB = a.FB;
    constexpr int nci = NCI;
    // row strides are 64 bytes past a multiple of 256, so the four consecutive rows a transposed read touches per
    // 32-lane half fall on four different quarters of the 64 banks (conflict-free)
    constexpr int DRS = 64 * 2 + 64;
    const int XRS = (Cin * 2 + 127) / 128 * 128 + 64;
    char* tab = smem;                                     // [K][2][64] x 16 B
    char* dimg = smem + (size_t)K * 2 * 64 * 16;          // [FB][32][DRS]
    char* ximg = dimg + (size_t)FB * 32 * DRS;            // [FB][32][XRS] (+32 B slack for partial ci tiles)
    const int c0 = blockIdx.y * 64;
    const int ngrp = a.cit / nci;
    const int cig = wave % ngrp, pair = wave / ngrp;
    const int cot = pair & 1, k = pair >> 1;

    {
        const int nz = (int)(((size_t)FB * 32 * (DRS + XRS) + 32) / 16);
        for (int p = tid; p < nz; p += NT) *(i32x4*)(dimg + (size_t)p * 16) = zero16();
        for (int e = tid; e < K * 2 * 64; e += NT) {
            const int ln = e & 63, s2 = (e >> 6) & 1, kk = e >> 7;
            const int v = ln & 31, hh = ln >> 5;
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int w = 16 * s2 + 8 * hh + j;
                f[j] = (v < V && w < V) ? a.adj[((size_t)kk * V + v) * V + w] : 0.f;
            }
            *(i32x4*)(tab + (size_t)e * 16) = pack16<bf16>(f);
        }
    }

    // this thread's staging pieces: piece p = tid + NT * i is segment p % ppr of batch row p / ppr; NT is a multiple
    // of ppr, so the segment is fixed per thread and the row advances by NT / ppr per piece.  Kept per piece: the
    // padded LDS row (frame * 32 + joint), or -1 past the batch.
    const int dppr = 8, xppr = Cin / 8;
    const int dseg = tid % dppr, dj0 = tid / dppr, djs = NT / dppr;
    const int xseg = tid % xppr, xj0 = tid / xppr, xjs = NT / xppr;
    const bool dseg_ok = c0 + dseg * 8 < Cout;
    int d_lr[GW_MAXD], x_lr[GW_MAXX];
#pragma unroll
    for (int i = 0; i < GW_MAXD; ++i) {
        const int j = dj0 + djs * i;
        const int fi = j / V, v = j - fi * V;
        d_lr[i] = (j < FB * V && dseg_ok) ? fi * 32 + v : -1;
    }
#pragma unroll
    for (int i = 0; i < GW_MAXX; ++i) {
        const int j = xj0 + xjs * i;
        const int fi = j / V, v = j - fi * V;
        x_lr[i] = j < FB * V ? fi * 32 + v : -1;
    }

    f32x16 accw[NCI];
#pragma unroll
    for (int it = 0; it < NCI; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r) accw[it][r] = 0.f;
    float dbacc = 0.f;
    const int q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = (lane >> 4) & 1;
    const int dcol = (cot * 32 + 16 * g1 + 4 * p4) * 2;
    const bool active = k < K;

    const int f_beg = blockIdx.x * a.frames_per_wg;
    const int f_end = min(a.NF, f_beg + a.frames_per_wg);
    i32x4 dreg[GW_MAXD], xreg[GW_MAXX];
    auto fetch = [&](int f0) {
        const int nrows = (min(f_end, f0 + FB) - f0) * V;           // valid batch rows
        const char* dyb = (const char*)a.dy + ((size_t)f0 * V * Cout + c0 + dseg * 8) * 2;
        const char* xb = (const char*)a.x + ((size_t)f0 * V * Cin + xseg * 8) * 2;
#pragma unroll
        for (int i = 0; i < GW_MAXD; ++i) {
            const int j = dj0 + djs * i;
            dreg[i] = zero16();
            if (d_lr[i] >= 0 && j < nrows) dreg[i] = *(const i32x4*)(dyb + (size_t)j * Cout * 2);
        }
#pragma unroll
        for (int i = 0; i < GW_MAXX; ++i) {
            const int j = xj0 + xjs * i;
            xreg[i] = zero16();
            if (x_lr[i] >= 0 && j < nrows) xreg[i] = *(const i32x4*)(xb + (size_t)j * Cin * 2);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < GW_MAXD; ++i)
            if (d_lr[i] >= 0) *(i32x4*)(dimg + d_lr[i] * DRS + dseg * 16) = dreg[i];
#pragma unroll
        for (int i = 0; i < GW_MAXX; ++i)
            if (x_lr[i] >= 0) *(i32x4*)(ximg + x_lr[i] * XRS + xseg * 16) = xreg[i];
    };

    if (f_beg < f_end) fetch(f_beg);
    for (int f0 = f_beg; f0 < f_end; f0 += FB) {
        __syncthreads();                                  // images of the previous batch fully consumed (and the zero fill)
        commit();
        __syncthreads();
        if (f0 + FB < f_end) fetch(f0 + FB);
        if (active) {
            const int nfr = min(FB, f_end - f0);
            for (int fi = 0; fi < nfr; ++fi) {
                const char* di = dimg + (size_t)fi * 32 * DRS;
                const char* xi = ximg + (size_t)fi * 32 * XRS;
                // dz = A_eff[k] (v x w) * dy (w x co): rows = v (registers), cols = co (lane)
                f32x16 dz;
#pragma unroll
                for (int r = 0; r < 16; ++r) dz[r] = 0.f;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const i32x4 afr = *(const i32x4*)(tab + (((size_t)k * 2 + s2) * 64 + lane) * 16);
                    const int row0 = 16 * s2 + 8 * h + q4;
                    const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(di + row0 * DRS + dcol));
                    const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(di + (row0 + 4) * DRS + dcol));
                    const s16x8 bfr = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                    dz = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8, afr), bfr, dz, 0, 0, 0);
                }
                float sdz = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) sdz += dz[r];
                dbacc += sdz;
                // dW_k[co][ci] += sum_v dz[v][co] * x[v][ci]   (dz as the A operand, contraction in acc-row order)
                i32x4 afr2[2];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int j = 0; j < 4; ++j) afr2[s2][j] = (int)pack_bf16x2(dz[8 * s2 + 2 * j], dz[8 * s2 + 2 * j + 1]);
#pragma unroll
                for (int it = 0; it < NCI; ++it) {
                    {
                        const int col = ((cig * nci + it) * 32 + 16 * g1 + 4 * p4) * 2;
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2) {
                            const int row0 = 16 * s2 + 4 * h + q4;
                            const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(xi + row0 * XRS + col));
                            const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(xi + (row0 + 8) * XRS + col));
                            const s16x8 bfr = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                            accw[it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8, afr2[s2]), bfr, accw[it], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }

    if (active) {
        const size_t part = blockIdx.x;
#pragma unroll
        for (int it = 0; it < NCI; ++it) {
            {
                const int ci = (cig * nci + it) * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = c0 + cot * 32 + acc_row(r, h);
                    if (co < Cout && ci < Cin) a.slabW[part * a.part_stride + ((size_t)k * Cout + co) * Cin + ci] = accw[it][r];
                }
            }
        }
        const float sb = dbacc + __shfl_xor(dbacc, 32);
        const int co = c0 + cot * 32 + l31;
        if (cig == 0 && h == 0 && co < Cout) a.slabB[part * a.part_stride + (size_t)k * Cout + co] = sb;
    }
}

// -1: shape outside the wide kernel's domain
static int gcn_wgrad_wide_try(const void* x, const void* dy, const float* adj, float* dW, float* db,
                              int NF, int V, int Cin, int Cout, int K, void* ws, size_t ws_bytes, StgcnSumList* list, hipStream_t stream) {
    const int cit = ceil_div_i(Cin, 32);
    if (!(cit == 1 || cit == 2 || cit == 4 || cit == 8) || Cin % 8 != 0 || Cout % 8 != 0 || K < 1 || K > GCN_KMAX || V < 1 || V > 32) return -1;
    GcnWgradWideArgs a;
    a.x = x; a.dy = dy; a.adj = adj; a.NF = NF; a.V = V; a.Cin = Cin; a.Cout = Cout; a.K = K;
    a.cit = cit; a.nci = cit >= 2 ? cit / 2 : 1;
    const int nw = K * 2 * (cit / a.nci);
    const int nt = nw * 64;
    const size_t per_frame = (size_t)32 * (192 + (Cin * 2 + 127) / 128 * 128 + 64);
    int fb = 8;
    while (fb > 1 && ((size_t)fb * per_frame > 120 * 1024 || ceil_div_i(fb * V * 8, nt) > GW_MAXD || ceil_div_i(fb * V * (Cin / 8), nt) > GW_MAXX)) --fb;
    if (ceil_div_i(fb * V * 8, nt) > GW_MAXD || ceil_div_i(fb * V * (Cin / 8), nt) > GW_MAXX) return -1;
    a.FB = fb;
    const size_t lds = (size_t)K * 2 * 64 * 16 + (size_t)fb * per_frame + 32;
    if (lds > 160 * 1024) return -1;
    const int gy = ceil_div_i(Cout, 64);
    // as many workgroups as stay resident: narrow inputs run few waves per workgroup and leave LDS for three
    const int per_cu = nt <= 384 && 3 * lds <= 160 * 1024 ? 3 : nt <= 512 && 2 * lds <= 160 * 1024 ? 2 : 1;
    int gx = max(1, per_cu * stgcn_num_cus() / gy);
    const size_t nW = (size_t)K * Cout * Cin, nB = (size_t)K * Cout;
    while (gx > 1 && (size_t)gx * (nW + nB) * sizeof(float) > ws_bytes) gx /= 2;
    if ((size_t)gx * (nW + nB) * sizeof(float) > ws_bytes) return STGCN_ERR_WORKSPACE;
    int fpw = ceil_div_i(NF, gx);
    fpw = ceil_div_i(fpw, fb) * fb;
    gx = ceil_div_i(NF, fpw);
    a.frames_per_wg = fpw;
    a.slabW = (float*)ws;
    a.part_stride = nW + nB;
    a.slabB = a.slabW + nW;                               // bias sums follow the weight sums inside every partial
    const void* fn = a.nci == 1 ? (const void*)gcn_wgrad_wide_kernel<1> : a.nci == 2 ? (const void*)gcn_wgrad_wide_kernel<2> : (const void*)gcn_wgrad_wide_kernel<4>;
    static std::atomic<uint64_t> lds_ok[3];
    const int si = a.nci == 1 ? 0 : a.nci == 2 ? 1 : 2;
    if (stgcn_allow_big_lds(fn, lds_ok[si]) != STGCN_OK) return STGCN_ERR_LAUNCH;
    const dim3 grid((unsigned)gx, (unsigned)gy), block((unsigned)nt);
    if (a.nci == 1) hipLaunchKernelGGL(gcn_wgrad_wide_kernel<1>, grid, block, lds, stream, a);
    else if (a.nci == 2) hipLaunchKernelGGL(gcn_wgrad_wide_kernel<2>, grid, block, lds, stream, a);
    else hipLaunchKernelGGL(gcn_wgrad_wide_kernel<4>, grid, block, lds, stream, a);
    if (hipGetLastError() != hipSuccess) return STGCN_ERR_LAUNCH;
    // one reduction over the combined partials, written to the two outputs
    return gcn_sum(list, a.slabW, dW, nW, db, nB, gx, false, stream);
}

template <typename T>
static int gcn_wgrad_run(GcnWgradArgs& a, float* dW, float* db, size_t ws_bytes, StgcnSumList* list, hipStream_t stream) {
    constexpr int ES = ET<T>::ES, KSTEP = ET<T>::KSTEP;
    if (a.Cin % KSTEP != 0 || a.Cin <= 0 || a.Cout % 8 != 0 || a.Cout <= 0 || a.K < 1 || a.K > GCN_KMAX || a.V < 1 || a.V > 32 || a.NF < 1) return STGCN_ERR_ARG;
    // input channels per workgroup: the widest group of whole k-steps, at most 64 channels, that divides Cin
    // (64 for the usual widths; 48 for 96 channels, 16 for 80)
    int cig = a.Cin >= 64 ? 64 : a.Cin;
    while (a.Cin % cig != 0) cig -= KSTEP;
    a.cig = cig;
    const int drs = 32 * ES + 16, xrs = cig * ES + 16;
    const size_t tabb = ET<T>::IS_BF16 ? (size_t)a.K * 2 * 64 * 16 : (size_t)a.K * 32 * 32 * 4;
    size_t lds = (size_t)4 * (32 * drs + 32 * xrs) + tabb;
    if (lds < 16 * 1024) lds = 16 * 1024;
    const int ncot = ceil_div_i(a.Cout, 32), ncig = a.Cin / cig;
    int gx = 3 * stgcn_num_cus() / (ncot * ncig);
    if (gx < 1) gx = 1;
    const int fgroups = ceil_div_i(a.NF, 4);
    if (gx > fgroups) gx = fgroups;
    const size_t nW = (size_t)a.K * a.Cout * a.Cin, nB = (size_t)a.K * a.Cout;
    while (gx > 1 && (size_t)gx * (nW + nB) * sizeof(float) > ws_bytes) gx /= 2;
    if ((size_t)gx * (nW + nB) * sizeof(float) > ws_bytes) return STGCN_ERR_WORKSPACE;
    a.iters = ceil_div_i(fgroups, gx);
    a.slabB = a.slabW + (size_t)gx * nW;
    auto kern = gcn_wgrad_kernel<T>;
    static std::atomic<uint64_t> lds_ok{0};
    if (stgcn_allow_big_lds((const void*)kern, lds_ok) != STGCN_OK) return STGCN_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)ncot, (unsigned)ncig), dim3(256), lds, stream, a);
    if (hipGetLastError() != hipSuccess) return STGCN_ERR_LAUNCH;
    int rc = gcn_sum(list, a.slabW, dW, nW, nullptr, 0, gx, false, stream);
    if (rc == STGCN_OK && db) rc = gcn_sum(list, a.slabB, db, nB, nullptr, 0, gx, false, stream);
    return rc;
}

static int gcn_wgrad_any(int dtype, const void* x, const void* dy, const float* adj, float* dW, float* db,
                         int NF, int V, int Cin, int Cout, int K,
                         void* workspace, size_t workspace_bytes, StgcnSumList* list, hipStream_t stream) {
    if (!x || !dy || !adj || !dW || !workspace) return STGCN_ERR_ARG;
    GcnWgradArgs a;
    a.x = x; a.dy = dy; a.adj = adj; a.slabW = (float*)workspace; a.slabB = nullptr;
    a.NF = NF; a.V = V; a.Cin = Cin; a.Cout = Cout; a.K = K; a.cig = 0; a.iters = 0;
    if (dtype == STGCN_BF16 && NF > 0) {
        const int rc = gcn_wgrad_wide_try(x, dy, adj, dW, db, NF, V, Cin, Cout, K, workspace, workspace_bytes, list, stream);
        if (rc >= 0) return rc;
    }
    if (dtype == STGCN_BF16) return gcn_wgrad_run<bf16>(a, dW, db, workspace_bytes, list, stream);
    if (dtype == STGCN_F32) return gcn_wgrad_run<float>(a, dW, db, workspace_bytes, list, stream);
    return STGCN_ERR_ARG;
}

// dW is [K][Cout][Cin] f32 (the torch conv weight [K*Cout, Cin, 1, 1]); db is [K][Cout] or null.
extern "C" int stgcn_gcn_wgrad(int dtype, const void* x, const void* dy, const float* adj, float* dW, float* db,
                               int NF, int V, int Cin, int Cout, int K,
                               void* workspace, size_t workspace_bytes, void* stream) {
    return gcn_wgrad_any(dtype, x, dy, adj, dW, db, NF, V, Cin, Cout, K, workspace, workspace_bytes, nullptr, (hipStream_t)stream);
}

// The same partial sums with the final sums left on the caller's list (see stgcn_conv_wgrad_deferred).
extern "C" int stgcn_gcn_wgrad_deferred(int dtype, const void* x, const void* dy, const float* adj, float* dW, float* db,
                                        int NF, int V, int Cin, int Cout, int K,
                                        void* workspace, size_t workspace_bytes, void* sum_list, size_t* workspace_used, void* stream) {
    StgcnSumList* list = (StgcnSumList*)sum_list;
    if (!list || !workspace_used) return STGCN_ERR_ARG;
    const int first = list->n;
    const int rc = gcn_wgrad_any(dtype, x, dy, adj, dW, db, NF, V, Cin, Cout, K, workspace, workspace_bytes, list, (hipStream_t)stream);
    if (rc != STGCN_OK) { list->n = first; return rc; }
    *workspace_used = stgcn_sum_list_extent(list, first, workspace);
    return STGCN_OK;
}
