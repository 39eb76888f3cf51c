// HBM-bound kernels of the st_gcn block: BatchNorm statistics finalisation, the fused
// BN2-apply + dropout + residual + ReLU output stage and its backward, BN1 backward apply,
// the input data_bn, the pooling / classifier head and the SGD update.
// All activation tensors are channels-last [rows][C]; a thread owns one 16-byte channel vector
// (fixed channels, so per-channel parameters live in registers and per-channel sums reduce in
// registers -> LDS -> one double atomic per channel per workgroup).
#include "stgcn_common.h"
#include <cstdlib>

// NT = threads per workgroup
template <typename T, int NT = 256> struct RowMap {
    int cvs, rpi, cv, rl; bool active;
    __device__ RowMap(int C) {
        cvs = C / ET<T>::VEC;
        rpi = NT / cvs; if (rpi < 1) rpi = 1;
        cv = threadIdx.x % cvs; rl = threadIdx.x / cvs;
        active = rl < rpi && (int)threadIdx.x < cvs * rpi;
    }
};

// sums[j] (per thread, NS floats for its VEC channels each) -> stats[s*C + c] += block total
template <typename T, int NS, typename OUT = double>
__device__ __forceinline__ void block_channel_reduce(float (*sums)[ET<T>::VEC], const RowMap<T>& m, int C, OUT* const* dst) {
    constexpr int VEC = ET<T>::VEC;
    __shared__ float red[256 * NS * VEC];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int j = 0; j < VEC; ++j) red[(threadIdx.x * NS + s) * VEC + j] = m.active ? sums[s][j] : 0.f;
    __syncthreads();
    // one atomic per (sum, channel) per workgroup; consecutive threads hit consecutive addresses
    for (int t = threadIdx.x; t < NS * C; t += 256) {
        const int s = t / C, c = t - s * C;
        if (!dst[s]) continue;
        const int cv = c / VEC, j = c - cv * VEC;
        float tot = 0.f;
        for (int r = 0; r < m.rpi; ++r) tot += red[((r * m.cvs + cv) * NS + s) * VEC + j];
        atomicAdd(&dst[s][c], (OUT)tot);
    }
}

// The same for workgroups of NT > 256 threads whose rows hold at most 64 channel vectors (cvs <= 64, a power of
// two): lanes of a wave that own the same channels are folded with shuffles first, then the NT/64 waves through
// LDS, one sum at a time.  Big workgroups keep the number of workgroups -- and with it the number of double
// atomics that pile up on the same few cache lines at the end of the kernel -- small.
template <typename T, int NS, int NT, typename OUT = double>
__device__ __forceinline__ void wide_channel_reduce(float (*sums)[ET<T>::VEC], const RowMap<T, NT>& m, int C, OUT* const* dst) {
    constexpr int VEC = ET<T>::VEC;
    constexpr int NW = NT / 64;
    __shared__ float red[NW * 64 * VEC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        float v[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            v[j] = m.active ? sums[s][j] : 0.f;
            for (int d = 32; d >= m.cvs; d >>= 1) v[j] += __shfl_xor(v[j], d);
        }
        __syncthreads();                                   // the previous sum's reads are done
        if (lane < m.cvs) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) red[(wave * 64 + lane) * VEC + j] = v[j];
        }
        __syncthreads();
        if (dst[s]) {
            for (int c = threadIdx.x; c < C; c += NT) {
                const int cv = c / VEC, j = c - cv * VEC;
                float tot = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) tot += red[(w * 64 + cv) * VEC + j];
                atomicAdd(&dst[s][c], (OUT)tot);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// BatchNorm parameter finalisation (tiny)
__global__ void bn_finalize_kernel(const double* stats, double count, const float* gamma, const float* beta,
                                   float* rmean, float* rvar, float momentum, float eps,
                                   float* mean, float* rstd, float* scale, float* shift, int C, int training) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    BnCoeff k;
    if (training) {
        k = bn_coeff_from_sums(stats[c], stats[C + c], count, g, b, eps);
        if (rmean) {
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * k.mean;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * k.unbiased_var;
        }
    } else {
        k.mean = rmean[c];
        k.rstd = 1.0f / sqrtf(rvar[c] + eps);
        k.scale = g * k.rstd;
        k.shift = fmaf(-k.mean, k.scale, b);
    }
    mean[c] = k.mean; rstd[c] = k.rstd;
    scale[c] = k.scale;
    shift[c] = k.shift;
}

extern "C" int stgcn_bn_finalize(const double* stats, double count, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, float momentum, float eps,
                                 float* mean, float* rstd, float* scale, float* shift, int C, int training, void* stream) {
    if (!mean || !rstd || !scale || !shift || C <= 0) return STGCN_ERR_ARG;
    if (training && (!stats || count <= 0)) return STGCN_ERR_ARG;
    if (!training && (!running_mean || !running_var)) return STGCN_ERR_ARG;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream,
                       stats, count, gamma, beta, running_mean, running_var, momentum, eps, mean, rstd, scale, shift, C, training);
    return hipGetLastError() == hipSuccess ? STGCN_OK : STGCN_ERR_LAUNCH;
}

// double sums -> float parameter gradients (dgamma = stats[1], dbeta = stats[0])
__global__ void stats_to_grads_kernel(const double* stats, float* dgamma, float* dbeta, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (dbeta) dbeta[c] = (float)stats[c];
    if (dgamma) dgamma[c] = (float)stats[C + c];
}
extern "C" int stgcn_stats_to_grads(const double* stats, float* dgamma, float* dbeta, int C, void* stream) {
    if (!stats || C <= 0) return STGCN_ERR_ARG;
    hipLaunchKernelGGL(stats_to_grads_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, stats, dgamma, dbeta, C);
    return hipGetLastError() == hipSuccess ? STGCN_OK : STGCN_ERR_LAUNCH;
}

// ------------------------------------------------------------------------------------------
// Block output stage:  out = relu( (s2*o + b2) * keep/(1-p) + res' ),  res' = sr*res + br | res | 0
// A BatchNorm whose statistics the output stage finishes itself (train mode): the sums its producer gathered, the
// parameters, where the coefficients go for the backward, the running statistics to advance.  stats == null: not used.
struct BnFromSums {
    const double* stats; const float* gamma; const float* beta; float* rmean; float* rvar;
    float* mean; float* rstd; float* scale; float* shift;
};
struct BlockOutArgs {
    const void* o; const void* res; void* out;
    const float* s2; const float* b2; const float* sr; const float* br;
    BnFromSums f2, fr; double count; float momentum, eps;
    uint16_t* mask;      // optional: two bytes per 16-byte piece: bit j = (element j > 0), bit 8+j = (element j kept by dropout)
    size_t rows; int C; uint32_t seed, thr; float keep_scale;
    const uint32_t* seed_step;   // optional device counter mixed into the seed (see stgcn_block_out_fwd)
};

constexpr int EW_TAB_MAXC = 512;     // widest row whose per-channel divisions go through an LDS table (one channel per thread)
constexpr int BOF_MAXC = 512;        // widest block whose BatchNorms the output stage finishes itself (coefficient table in LDS)

template <typename T, int NT = 256>
__global__ __launch_bounds__(NT) void block_out_fwd_kernel(BlockOutArgs a) {
    constexpr int VEC = ET<T>::VEC;
    __shared__ float coef[4][BOF_MAXC];      // scale2, shift2, scale_r, shift_r
    RowMap<T, NT> m(a.C);
    if (a.f2.stats) {
        // One channel per thread: the coefficients from the sums (double divisions: per thread for all its channels
        // they would cost more than the launch this saves on a short tensor), shared through LDS.  Workgroup 0 also
        // publishes them for the backward and advances the running statistics.
        const bool publish = blockIdx.x == 0;
        for (int c = threadIdx.x; c < a.C; c += NT) {
            const BnCoeff k = bn_coeff_from_sums(a.f2.stats[c], a.f2.stats[a.C + c], a.count, a.f2.gamma[c], a.f2.beta[c], a.eps);
            coef[0][c] = k.scale; coef[1][c] = k.shift;
            if (publish) {
                a.f2.mean[c] = k.mean; a.f2.rstd[c] = k.rstd; a.f2.scale[c] = k.scale; a.f2.shift[c] = k.shift;
                if (a.f2.rmean) {
                    a.f2.rmean[c] = (1.f - a.momentum) * a.f2.rmean[c] + a.momentum * k.mean;
                    a.f2.rvar[c] = (1.f - a.momentum) * a.f2.rvar[c] + a.momentum * k.unbiased_var;
                }
            }
            float qs = 1.f, qb = 0.f;
            if (a.fr.stats) {
                const BnCoeff q = bn_coeff_from_sums(a.fr.stats[c], a.fr.stats[a.C + c], a.count, a.fr.gamma[c], a.fr.beta[c], a.eps);
                qs = q.scale; qb = q.shift;
                if (publish) {
                    a.fr.mean[c] = q.mean; a.fr.rstd[c] = q.rstd; a.fr.scale[c] = q.scale; a.fr.shift[c] = q.shift;
                    if (a.fr.rmean) {
                        a.fr.rmean[c] = (1.f - a.momentum) * a.fr.rmean[c] + a.momentum * q.mean;
                        a.fr.rvar[c] = (1.f - a.momentum) * a.fr.rvar[c] + a.momentum * q.unbiased_var;
                    }
                }
            }
            coef[2][c] = qs; coef[3][c] = qb;
        }
        __syncthreads();
    }
    if (!m.active) return;
    if (a.thr && a.seed_step) a.seed += *a.seed_step * STGCN_SEED_STEP_MUL;
    float s2[VEC], b2[VEC], sr[VEC], br[VEC];
    if (a.f2.stats) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int c = m.cv * VEC + j;
            s2[j] = coef[0][c]; b2[j] = coef[1][c]; sr[j] = coef[2][c]; br[j] = coef[3][c];
        }
    } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int c = m.cv * VEC + j;
            s2[j] = a.s2[c]; b2[j] = a.b2[c];
            sr[j] = a.sr ? a.sr[c] : 1.f; br[j] = a.br ? a.br[c] : 0.f;
        }
    }
    for (size_t row = (size_t)blockIdx.x * m.rpi + m.rl; row < a.rows; row += (size_t)gridDim.x * m.rpi) {
        const size_t off = row * a.C + m.cv * VEC;
        float f[VEC], r[VEC];
        unpack16<T>(*(const i32x4*)((const T*)a.o + off), f);
        if (a.res) unpack16<T>(*(const i32x4*)((const T*)a.res + off), r);
        const uint32_t keep = a.thr ? drop_keep_bits<VEC>(off, a.seed, a.thr) : (1u << VEC) - 1u;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float q = fmaf(f[j], s2[j], b2[j]);
            if (a.thr) q = (keep >> j) & 1u ? q * a.keep_scale : 0.f;
            if (a.res) q += fmaf(r[j], sr[j], br[j]);
            f[j] = fmaxf(q, 0.f);
        }
        const i32x4 packed = pack16<T>(f);
        *(i32x4*)((T*)a.out + off) = packed;
        if (a.mask) {
            float fr[VEC];
            unpack16<T>(packed, fr);                      // the stored (rounded) values decide
            uint32_t bits = 0;
#pragma unroll
            for (int j = 0; j < VEC; ++j) bits |= (fr[j] > 0.f ? 1u : 0u) << j;
            a.mask[off / VEC] = (uint16_t)(bits | (keep << 8));
        }
    }
}

static inline unsigned ew_grid(size_t rows, int rows_per_block, size_t cap = 2048) {
    size_t g = (rows + rows_per_block - 1) / rows_per_block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}
template <typename T> static inline bool ew_ok(int C) {
    const int VEC = ET<T>::VEC;
    return C > 0 && C % VEC == 0 && C / VEC <= 256;
}
template <typename T> static inline int ew_rpi(int C) { int r = 256 / (C / ET<T>::VEC); return r < 1 ? 1 : r; }

// 16-bit dropout threshold: an element is kept when its 16 hash bits are >= round(p * 65536); 0 = no dropout
static inline uint32_t drop_thr(float p) {
    if (p <= 0.f) return 0u;
    double t = (double)p * 65536.0 + 0.5;
    if (t > 65535.0) t = 65535.0;
    if (t < 1.0) t = 1.0;
    return (uint32_t)t;
}

static int block_out_fwd_launch(int dtype, BlockOutArgs& a, float drop_p, hipStream_t stream) {
    const size_t rows = a.rows; const int C = a.C;
    a.thr = drop_thr(drop_p); a.keep_scale = 1.f / (1.f - drop_p);
    if (dtype == STGCN_BF16) {
        if (!ew_ok<bf16>(C)) return STGCN_ERR_ARG;
        const int cvs = C / ET<bf16>::VEC;
        if (cvs <= 64 && (cvs & (cvs - 1)) == 0)         // as bn_bwd_apply: 512 workgroups of 512 threads (1-2 % here)
            hipLaunchKernelGGL((block_out_fwd_kernel<bf16, 512>), dim3(ew_grid(rows, 512 / cvs, 512)), dim3(512), 0, stream, a);
        else
            hipLaunchKernelGGL((block_out_fwd_kernel<bf16, 256>), dim3(ew_grid(rows, ew_rpi<bf16>(C))), dim3(256), 0, stream, a);
    } else if (dtype == STGCN_F32) {
        if (!ew_ok<float>(C)) return STGCN_ERR_ARG;
        hipLaunchKernelGGL((block_out_fwd_kernel<float, 256>), dim3(ew_grid(rows, ew_rpi<float>(C))), dim3(256), 0, stream, a);
    } else return STGCN_ERR_ARG;
    return hipGetLastError() == hipSuccess ? STGCN_OK : STGCN_ERR_LAUNCH;
}

extern "C" int stgcn_block_out_fwd(int dtype, const void* o, const void* res, void* out, void* relu_mask,
                                   const float* scale2, const float* shift2, const float* res_scale, const float* res_shift,
                                   size_t rows, int C, float drop_p, uint32_t seed, const uint32_t* seed_step, void* stream) {
    if (!o || !out || !scale2 || !shift2 || drop_p < 0.f || drop_p >= 1.f) return STGCN_ERR_ARG;
    BlockOutArgs a{};
    a.seed_step = seed_step;
    a.o = o; a.res = res; a.out = out; a.mask = (uint16_t*)relu_mask; a.s2 = scale2; a.b2 = shift2; a.sr = res_scale; a.br = res_shift;
    a.rows = rows; a.C = C; a.seed = seed;
    return block_out_fwd_launch(dtype, a, drop_p, (hipStream_t)stream);
}

// The train-mode output stage with its BatchNorms finished in the same launch: bn2 (and, with bnr_stats, the residual
// branch's BatchNorm) arrive as the double sums [sum, sum of squares] their convolutions gathered over `rows` rows.
// Every workgroup derives scale and shift from the sums; workgroup 0 writes mean / rstd / scale / shift to bn*_out
// (float[4][C], for the backward) and advances the running statistics (either pair may be null).
extern "C" int stgcn_block_out_fwd_bn(int dtype, const void* o, const void* res, void* out, void* relu_mask,
                                      const double* bn2_stats, const float* gamma2, const float* beta2,
                                      float* running_mean2, float* running_var2, float* bn2_out,
                                      const double* bnr_stats, const float* gammar, const float* betar,
                                      float* running_meanr, float* running_varr, float* bnr_out,
                                      float momentum, float eps,
                                      size_t rows, int C, float drop_p, uint32_t seed, const uint32_t* seed_step, void* stream) {
    if (!o || !out || !bn2_stats || !gamma2 || !beta2 || !bn2_out || rows == 0 || drop_p < 0.f || drop_p >= 1.f) return STGCN_ERR_ARG;
    if (C > BOF_MAXC) return STGCN_ERR_ARG;               // wider blocks: stgcn_bn_finalize + stgcn_block_out_fwd
    if ((running_mean2 == nullptr) != (running_var2 == nullptr)) return STGCN_ERR_ARG;
    if (bnr_stats && (!res || !gammar || !betar || !bnr_out || (running_meanr == nullptr) != (running_varr == nullptr))) return STGCN_ERR_ARG;
    BlockOutArgs a{};
    a.seed_step = seed_step;
    a.o = o; a.res = res; a.out = out; a.mask = (uint16_t*)relu_mask;
    a.f2 = BnFromSums{bn2_stats, gamma2, beta2, running_mean2, running_var2, bn2_out, bn2_out + C, bn2_out + 2 * C, bn2_out + 3 * C};
    if (bnr_stats) a.fr = BnFromSums{bnr_stats, gammar, betar, running_meanr, running_varr, bnr_out, bnr_out + C, bnr_out + 2 * C, bnr_out + 3 * C};
    a.count = (double)rows; a.momentum = momentum; a.eps = eps;
    a.rows = rows; a.C = C; a.seed = seed;
    return block_out_fwd_launch(dtype, a, drop_p, (hipStream_t)stream);
}

// Backward of the output stage, pass 1: per-channel sums
//   g  = dout * (out > 0);  gd = g * keep/(1-p)
//   stats2 = [sum gd, sum gd*ohat]  (ohat = (o - mean2) * rstd2)
//   statsr = [sum g,  sum g*rhat]   (only for the conv residual: rhat = (r - meanr) * rstdr)
struct BlockOutBwdArgs {
    const void* dout; const float* dpool; size_t rows_per_sample; float dpool_scale;
    const void* out; const uint16_t* mask; const void* o; const void* r;
    const float* mean2; const float* rstd2; const float* gamma2;
    const float* meanr; const float* rstdr; const float* gammar;
    double* stats2; double* statsr;
    void* dO; void* dR; void* g;
    double* dbias_o; double* dbias_r;        // sums (double atomics of float32 partials: order-free), zeroed by the caller
    float* dgamma2; float* dbeta2; float* dgammar; float* dbetar;   // optional: BatchNorm parameter gradients from stats2 / statsr
    size_t rows; int C; uint32_t seed, thr; float keep_scale; double count; int training;
};

// The incoming gradient of the block output is a stored tensor and/or a per-(sample, channel) value that is
// constant over (t, v) -- the gradient of the global average pool that follows the last block.

// The raw operands of one 16-byte piece, fetched ahead of their use so that two rows' loads are in flight together.
constexpr int BOB_NR = 2, BOB_NA = 2;     // rows in flight per thread: reduce pass, apply pass
constexpr int BOB_NT = 1024, BOB_GRID = 256;   // reduce pass: threads per workgroup, workgroups
struct BlockOutBwdPiece { i32x4 d, o, r, y; uint32_t bits; };

template <typename T>
__device__ __forceinline__ void fetch_piece(const BlockOutBwdArgs& a, size_t off, bool want_r, BlockOutBwdPiece& p) {
    constexpr int VEC = ET<T>::VEC;
    if (a.dout) p.d = *(const i32x4*)((const T*)a.dout + off);
    if (a.mask) p.bits = a.mask[off / VEC];
    else p.y = *(const i32x4*)((const T*)a.out + off);
    p.o = *(const i32x4*)((const T*)a.o + off);
    if (want_r) p.r = *(const i32x4*)((const T*)a.r + off);
}

// incoming gradient of the piece: the stored tensor and/or the pooled per-(sample, channel) value
template <typename T>
__device__ __forceinline__ void piece_dout(const BlockOutBwdArgs& a, const BlockOutBwdPiece& p, size_t row, int cv, float* d) {
    constexpr int VEC = ET<T>::VEC;
    if (a.dout) unpack16<T>(p.d, d);
    else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) d[j] = 0.f;
    }
    if (a.dpool) {
        const float* q = a.dpool + (row / a.rows_per_sample) * a.C + cv * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) d[j] = fmaf(q[j], a.dpool_scale, d[j]);
    }
}

// bit j: element j of the piece passed the block's final ReLU; bit 8+j: it was kept by dropout.  From the mask the
// forward wrote, or from the stored output and the dropout hash.
template <typename T>
__device__ __forceinline__ uint32_t piece_mask_bits(const BlockOutBwdArgs& a, const BlockOutBwdPiece& p, size_t off) {
    constexpr int VEC = ET<T>::VEC;
    if (a.mask) return p.bits;
    float y[VEC];
    unpack16<T>(p.y, y);
    uint32_t bits = (a.thr ? drop_keep_bits<VEC>(off, a.seed, a.thr) : (1u << VEC) - 1u) << 8;
#pragma unroll
    for (int j = 0; j < VEC; ++j) bits |= (y[j] > 0.f ? 1u : 0u) << j;
    return bits;
}

template <typename T, bool HAS_R, int NROW, int NT>
__global__ __launch_bounds__(NT) void block_out_bwd_reduce_kernel(BlockOutBwdArgs a) {
    constexpr int VEC = ET<T>::VEC;
    RowMap<T, NT> m(a.C);
    float sums[4][VEC];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < VEC; ++j) sums[s][j] = 0.f;
    if (m.active) {
        // the loop gathers raw moments (sum gd, sum gd*o, sum g, sum g*r) and keeps no per-channel constants in
        // registers; sum gd*ohat = rstd * (sum gd*o - mean * sum gd) is formed once at the end
        constexpr bool want_r = HAS_R;
        auto consume = [&](size_t row, const BlockOutBwdPiece& p) {
            const size_t off = row * a.C + m.cv * VEC;
            float d[VEC], o[VEC], r[VEC];
            piece_dout<T>(a, p, row, m.cv, d);
            const uint32_t pos = piece_mask_bits<T>(a, p, off);
            unpack16<T>(p.o, o);
            if (want_r) unpack16<T>(p.r, r);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float g = (pos >> j) & 1u ? d[j] : 0.f;
                float gd = g;
                if (a.thr) gd = (pos >> (8 + j)) & 1u ? g * a.keep_scale : 0.f;
                sums[0][j] += gd;
                sums[1][j] = fmaf(gd, o[j], sums[1][j]);
                if (want_r) {
                    sums[2][j] += g;
                    sums[3][j] = fmaf(g, r[j], sums[3][j]);
                }
            }
        };
        const size_t rstep = (size_t)gridDim.x * m.rpi;
        for (size_t row = (size_t)blockIdx.x * m.rpi + m.rl; row < a.rows; row += NROW * rstep) {
            BlockOutBwdPiece p[NROW];
#pragma unroll
            for (int i = 0; i < NROW; ++i)
                if (row + i * rstep < a.rows) fetch_piece<T>(a, (row + i * rstep) * a.C + m.cv * VEC, want_r, p[i]);
#pragma unroll
            for (int i = 0; i < NROW; ++i)
                if (row + i * rstep < a.rows) consume(row + i * rstep, p[i]);
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int c = m.cv * VEC + j;
            sums[1][j] = a.rstd2[c] * (sums[1][j] - a.mean2[c] * sums[0][j]);
            if constexpr (want_r) sums[3][j] = a.rstdr[c] * (sums[3][j] - a.meanr[c] * sums[2][j]);
        }
    }
    double* dst[4] = {a.stats2, a.stats2 + a.C, HAS_R ? a.statsr : nullptr, HAS_R ? a.statsr + a.C : nullptr};
    if constexpr (NT == 256) block_channel_reduce<T, 4>(sums, m, a.C, dst);
    else wide_channel_reduce<T, HAS_R ? 4 : 2, NT>(sums, m, a.C, dst);
}

// pass 2: dO = BN2 backward of gd; dR = BNr backward of g (conv residual); g copy (identity residual)
template <typename T, bool HAS_R, int NROW, int NT>
__global__ __launch_bounds__(NT) void block_out_bwd_apply_kernel(BlockOutBwdArgs a) {
    constexpr int VEC = ET<T>::VEC;
    RowMap<T, NT> m(a.C);
    float sums[2][VEC];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < VEC; ++j) sums[s][j] = 0.f;
    // the means of the sums (double divisions) once per channel, through an LDS table: see bn_bwd_apply_kernel
    __shared__ float tab[4][EW_TAB_MAXC];
    const bool use_tab = a.training && a.C <= EW_TAB_MAXC;
    if (use_tab) {
        for (int c = threadIdx.x; c < a.C; c += NT) {
            tab[0][c] = (float)(a.stats2[c] / a.count);
            tab[1][c] = (float)(a.stats2[a.C + c] / a.count);
            if constexpr (HAS_R) {
                tab[2][c] = (float)(a.statsr[c] / a.count);
                tab[3][c] = (float)(a.statsr[a.C + c] / a.count);
            }
        }
        __syncthreads();
    }
    if (m.active) {
        // k*(gd - m1 - ohat*m2) with ohat = (o - mean)*rstd, folded to three constants per channel:
        //   k*gd + qa*o + qb,  qa = -k*m2*rstd,  qb = -k*m1 - qa*mean
        // every per-channel load first (from a valid address on every path), then the arithmetic: channel by channel
        // each channel's loads would wait behind the previous channel's divisions -- eight trips to memory in a row
        float k2[VEC], qa2[VEC], qb2[VEC], kr[VEC], qar[VEC], qbr[VEC];
        float rs2[VEC], mn2[VEC], rsr[VEC], mnr[VEC];
        double s2a[VEC], s2b[VEC], sra[VEC], srb[VEC];
        const bool divide = a.training && !use_tab;
        const double* st2 = divide ? a.stats2 : (const double*)a.mean2;      // not read otherwise
        const double* str = (HAS_R && divide) ? a.statsr : (const double*)a.mean2;
        const int cs = divide ? a.C : 0;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int c = m.cv * VEC + j;
            const int cd = divide ? c : 0;
            rs2[j] = a.rstd2[c]; k2[j] = a.gamma2[c]; mn2[j] = a.mean2[c];
            s2a[j] = st2[cd]; s2b[j] = st2[cs + cd];
            if constexpr (HAS_R) {
                rsr[j] = a.rstdr[c]; kr[j] = a.gammar[c]; mnr[j] = a.meanr[c];
                sra[j] = str[cd]; srb[j] = str[cs + cd];
            }
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int c = m.cv * VEC + j;
            k2[j] *= rs2[j];
            const float m1 = use_tab ? tab[0][c] : divide ? (float)(s2a[j] / a.count) : 0.f;
            const float m2 = use_tab ? tab[1][c] : divide ? (float)(s2b[j] / a.count) : 0.f;
            qa2[j] = -k2[j] * m2 * rs2[j];
            qb2[j] = -k2[j] * m1 - qa2[j] * mn2[j];
            if constexpr (HAS_R) {
                kr[j] *= rsr[j];
                const float n1 = use_tab ? tab[2][c] : divide ? (float)(sra[j] / a.count) : 0.f;
                const float n2 = use_tab ? tab[3][c] : divide ? (float)(srb[j] / a.count) : 0.f;
                qar[j] = -kr[j] * n2 * rsr[j];
                qbr[j] = -kr[j] * n1 - qar[j] * mnr[j];
            } else { kr[j] = qar[j] = qbr[j] = 0.f; }
        }
        constexpr bool want_r = HAS_R;
        auto consume = [&](size_t row, const BlockOutBwdPiece& p) {
            const size_t off = row * a.C + m.cv * VEC;
            float d[VEC], o[VEC], r[VEC], gg[VEC], dr[VEC];
            piece_dout<T>(a, p, row, m.cv, d);
            const uint32_t pos = piece_mask_bits<T>(a, p, off);
            unpack16<T>(p.o, o);
            if (want_r) unpack16<T>(p.r, r);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float g = (pos >> j) & 1u ? d[j] : 0.f;
                gg[j] = g;
                float gd = g;
                if (a.thr) gd = (pos >> (8 + j)) & 1u ? g * a.keep_scale : 0.f;
                o[j] = fmaf(k2[j], gd, fmaf(qa2[j], o[j], qb2[j]));
                sums[0][j] += o[j];
                if (want_r) {
                    dr[j] = fmaf(kr[j], g, fmaf(qar[j], r[j], qbr[j]));
                    sums[1][j] += dr[j];
                }
            }
            *(i32x4*)((T*)a.dO + off) = pack16<T>(o);
            if constexpr (want_r) *(i32x4*)((T*)a.dR + off) = pack16<T>(dr);
            if (a.g) *(i32x4*)((T*)a.g + off) = pack16<T>(gg);
        };
        const size_t rstep = (size_t)gridDim.x * m.rpi;
        for (size_t row = (size_t)blockIdx.x * m.rpi + m.rl; row < a.rows; row += NROW * rstep) {
            BlockOutBwdPiece p[NROW];
#pragma unroll
            for (int i = 0; i < NROW; ++i)
                if (row + i * rstep < a.rows) fetch_piece<T>(a, (row + i * rstep) * a.C + m.cv * VEC, want_r, p[i]);
#pragma unroll
            for (int i = 0; i < NROW; ++i)
                if (row + i * rstep < a.rows) consume(row + i * rstep, p[i]);
        }
    }
    double* dst[2] = {a.dbias_o, HAS_R ? a.dbias_r : nullptr};
    if constexpr (NT == 256) block_channel_reduce<T, 2>(sums, m, a.C, dst);
    else wide_channel_reduce<T, HAS_R ? 2 : 1, NT>(sums, m, a.C, dst);
    // the sums of the reduce pass are the BatchNorm parameter gradients: dbeta = sum g, dgamma = sum g*xhat
    if (blockIdx.x == 0) {
        for (int c = threadIdx.x; c < a.C; c += NT) {
            if (a.dbeta2) a.dbeta2[c] = (float)a.stats2[c];
            if (a.dgamma2) a.dgamma2[c] = (float)a.stats2[a.C + c];
            if (HAS_R && a.dbetar) a.dbetar[c] = (float)a.statsr[c];
            if (HAS_R && a.dgammar) a.dgammar[c] = (float)a.statsr[a.C + c];
        }
    }
}

// The reduce pass reads the conv residual whenever there is one; the apply pass only when it writes dR.
template <typename T>
static void block_out_bwd_launch(const BlockOutBwdArgs& a, int pass, hipStream_t st) {
    const unsigned gr = ew_grid(a.rows, ew_rpi<T>(a.C), 2048);   // every workgroup ends in C*NS double atomics
    if (pass == 0) {
        // the read-only pass is short enough for its closing atomics to show: with rows of <= 64 channel vectors it
        // runs as few, large workgroups
        const int cvs = a.C / ET<T>::VEC;
        if (cvs <= 64 && (cvs & (cvs - 1)) == 0) {
            const unsigned gw = ew_grid(a.rows, BOB_NT / cvs, BOB_GRID);
            if (a.r) hipLaunchKernelGGL((block_out_bwd_reduce_kernel<T, true, BOB_NR, BOB_NT>), dim3(gw), dim3(BOB_NT), 0, st, a);
            else hipLaunchKernelGGL((block_out_bwd_reduce_kernel<T, false, BOB_NR, BOB_NT>), dim3(gw), dim3(BOB_NT), 0, st, a);
        } else if (a.r) hipLaunchKernelGGL((block_out_bwd_reduce_kernel<T, true, BOB_NR, 256>), dim3(gr), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((block_out_bwd_reduce_kernel<T, false, BOB_NR, 256>), dim3(gr), dim3(256), 0, st, a);
    } else {
        const int cvs = a.C / ET<T>::VEC;
        if (a.dR && cvs <= 64 && (cvs & (cvs - 1)) == 0)      // 512 threads: the conv-residual form needs 168 registers
            hipLaunchKernelGGL((block_out_bwd_apply_kernel<T, true, BOB_NA, 512>), dim3(ew_grid(a.rows, 512 / cvs, 2 * BOB_GRID)), dim3(512), 0, st, a);
        else if (a.dR) hipLaunchKernelGGL((block_out_bwd_apply_kernel<T, true, BOB_NA, 256>), dim3(gr), dim3(256), 0, st, a);
        else if (cvs <= 64 && (cvs & (cvs - 1)) == 0)
            hipLaunchKernelGGL((block_out_bwd_apply_kernel<T, false, BOB_NA, BOB_NT>), dim3(ew_grid(a.rows, BOB_NT / cvs, BOB_GRID)), dim3(BOB_NT), 0, st, a);
        else hipLaunchKernelGGL((block_out_bwd_apply_kernel<T, false, BOB_NA, 256>), dim3(gr), dim3(256), 0, st, a);
    }
}

extern "C" int stgcn_block_out_bwd(int dtype, int pass, const void* dout, const float* dpool, size_t rows_per_sample, float dpool_scale,
                                   const void* out, const void* relu_mask, const void* o, const void* r,
                                   const float* mean2, const float* rstd2, const float* gamma2,
                                   const float* meanr, const float* rstdr, const float* gammar,
                                   double* stats2, double* statsr, void* dO, void* dR, void* g,
                                   double* dbias_o, double* dbias_r,
                                   float* dgamma2, float* dbeta2, float* dgammar, float* dbetar,
                                   size_t rows, int C, float drop_p, uint32_t seed, double count, int training, void* stream) {
    if ((!dout && !dpool) || (!out && !relu_mask) || !o || !mean2 || !rstd2 || !stats2) return STGCN_ERR_ARG;
    if (dpool && rows_per_sample == 0) return STGCN_ERR_ARG;
    if (r && (!meanr || !rstdr || !statsr)) return STGCN_ERR_ARG;
    BlockOutBwdArgs a;
    a.dout = dout; a.dpool = dpool; a.rows_per_sample = rows_per_sample; a.dpool_scale = dpool_scale; a.out = out; a.mask = (const uint16_t*)relu_mask; a.o = o; a.r = r; a.mean2 = mean2; a.rstd2 = rstd2; a.gamma2 = gamma2;
    a.meanr = meanr; a.rstdr = rstdr; a.gammar = gammar; a.stats2 = stats2; a.statsr = statsr;
    a.dO = dO; a.dR = dR; a.g = g; a.dbias_o = dbias_o; a.dbias_r = dbias_r;
    a.dgamma2 = dgamma2; a.dbeta2 = dbeta2; a.dgammar = dgammar; a.dbetar = dbetar;
    a.rows = rows; a.C = C; a.seed = seed; a.thr = drop_thr(drop_p); a.keep_scale = 1.f / (1.f - drop_p);
    a.count = count; a.training = training;
    if (pass == 1) {
        if (!dO || !gamma2 || !dbias_o || (dR && (!r || !gammar || !dbias_r))) return STGCN_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    if (dtype == STGCN_BF16) {
        if (!ew_ok<bf16>(C)) return STGCN_ERR_ARG;
        block_out_bwd_launch<bf16>(a, pass, st);
    } else if (dtype == STGCN_F32) {
        if (!ew_ok<float>(C)) return STGCN_ERR_ARG;
        block_out_bwd_launch<float>(a, pass, st);
    } else return STGCN_ERR_ARG;
    return hipGetLastError() == hipSuccess ? STGCN_OK : STGCN_ERR_LAUNCH;
}

// ------------------------------------------------------------------------------------------
// BN backward apply (BN1):  dy = gamma*rstd*(g - S1/n - yhat*S2/n)   [eval: g*gamma*rstd]
struct BnBwdArgs {
    const void* g; const void* y; void* dy; const float* gamma; const float* mean; const float* rstd;
    const double* stats; double count; size_t rows; int C; int training;
    float* dgamma; float* dbeta;             // optional: dbeta = stats[0..C), dgamma = stats[C..2C) as float32
};
template <typename T, int NT = 256>
__global__ __launch_bounds__(NT) void bn_bwd_apply_kernel(BnBwdArgs a) {
    constexpr int VEC = ET<T>::VEC;
    RowMap<T, NT> m(a.C);
    if (blockIdx.x == 0 && a.stats) {
        for (int c = threadIdx.x; c < a.C; c += NT) {
            if (a.dbeta) a.dbeta[c] = (float)a.stats[c];
            if (a.dgamma) a.dgamma[c] = (float)a.stats[a.C + c];
        }
    }
    // The two means per channel are double divisions: one channel per thread through an LDS table (a thread dividing
    // for all its channels spends 2-3 us on it, which shows on short tensors); wider rows than the table divide per thread.
    __shared__ float tab[2][EW_TAB_MAXC];
    const bool use_tab = a.training && a.C <= EW_TAB_MAXC;
    if (use_tab) {
        for (int c = threadIdx.x; c < a.C; c += NT) {
            tab[0][c] = (float)(a.stats[c] / a.count);
            tab[1][c] = (float)(a.stats[a.C + c] / a.count);
        }
        __syncthreads();
    }
    if (!m.active) return;
    // Per-channel constants: every load first (from a valid address on every path), then the arithmetic.  Written
    // channel by channel -- load, branch on `training`, divide -- the compiler keeps each channel's loads behind the
    // previous channel's wait: eight trips to memory in a row before the first row moves.
    float mu[VEC], rs[VEC], kk[VEC], m1[VEC], m2[VEC];
    double sa[VEC], sb[VEC];
    const bool divide = a.training && !use_tab;
    const double* st = divide ? a.stats : (const double*)a.mean;     // not read otherwise: any valid address
    const int cs = divide ? a.C : 0;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const int c = m.cv * VEC + j;
        mu[j] = a.mean[c]; rs[j] = a.rstd[c]; kk[j] = a.gamma[c];
        const int cd = divide ? c : 0;
        sa[j] = st[cd]; sb[j] = st[cs + cd];
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const int c = m.cv * VEC + j;
        kk[j] *= rs[j];
        m1[j] = use_tab ? tab[0][c] : divide ? (float)(sa[j] / a.count) : 0.f;
        m2[j] = use_tab ? tab[1][c] : divide ? (float)(sb[j] / a.count) : 0.f;
    }
    for (size_t row = (size_t)blockIdx.x * m.rpi + m.rl; row < a.rows; row += (size_t)gridDim.x * m.rpi) {
        const size_t off = row * a.C + m.cv * VEC;
        float g[VEC], y[VEC];
        unpack16<T>(*(const i32x4*)((const T*)a.g + off), g);
        unpack16<T>(*(const i32x4*)((const T*)a.y + off), y);
#pragma unroll
        for (int j = 0; j < VEC; ++j) g[j] = kk[j] * (g[j] - m1[j] - ((y[j] - mu[j]) * rs[j]) * m2[j]);
        *(i32x4*)((T*)a.dy + off) = pack16<T>(g);
    }
}
extern "C" int stgcn_bn_bwd_apply(int dtype, const void* g, const void* y, void* dy, const float* gamma,
                                  const float* mean, const float* rstd, const double* stats, double count,
                                  float* dgamma, float* dbeta, size_t rows, int C, int training, void* stream) {
    if (!g || !y || !dy || !gamma || !mean || !rstd || (training && !stats)) return STGCN_ERR_ARG;
    BnBwdArgs a;
    a.g = g; a.y = y; a.dy = dy; a.gamma = gamma; a.mean = mean; a.rstd = rstd; a.stats = stats; a.count = count;
    a.rows = rows; a.C = C; a.training = training; a.dgamma = dgamma; a.dbeta = dbeta;
    if (dtype == STGCN_BF16) {
        if (!ew_ok<bf16>(C)) return STGCN_ERR_ARG;
        // rows of <= 64 channel vectors: 512 workgroups of 512 threads (two per CU) walk the rows with a grid stride --
        // 10 % faster than up to 2048 workgroups of 256 (0.0745 vs 0.0826 ms at 960 000 rows x 64 channels)
        const int cvs = C / ET<bf16>::VEC;
        if (cvs <= 64 && (cvs & (cvs - 1)) == 0)
            hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16, 512>), dim3(ew_grid(rows, 512 / cvs, 512)), dim3(512), 0, (hipStream_t)stream, a);
        else
            hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16, 256>), dim3(ew_grid(rows, ew_rpi<bf16>(C))), dim3(256), 0, (hipStream_t)stream, a);
    } else if (dtype == STGCN_F32) {
        if (!ew_ok<float>(C)) return STGCN_ERR_ARG;
        hipLaunchKernelGGL((bn_bwd_apply_kernel<float, 256>), dim3(ew_grid(rows, ew_rpi<float>(C))), dim3(256), 0, (hipStream_t)stream, a);
    } else return STGCN_ERR_ARG;
    return hipGetLastError() == hipSuccess ? STGCN_OK : STGCN_ERR_LAUNCH;
}

// ------------------------------------------------------------------------------------------
// data_bn: BatchNorm1d over channel index v*C + c of x[N, C, T, V, M] (f32), statistics over
// (n, m, t); output is the channels-last [N*M][T][V][Cp] activation (Cp >= C, padding zero).
// FIXED: thread (r, j) = (tid / VM, tid % VM) keeps position j = v*M + m of the contiguous [t][v][m] slab and walks
// frames t0 + r, t0 + r + R, ...; the partial sums are folded over r and m in a fixed order, so a workgroup's sums do
// not depend on how its waves were scheduled (float atomics in LDS, the other form, add in arrival order: the
// statistics, and with them every activation of a float32 forward pass, then change in their last bits from run to
// run).  Needs V*M <= 256.
template <bool FIXED>
__global__ __launch_bounds__(256) void data_bn_stats_kernel(const float* x, double* stats, int N, int C, int T, int V, int M) {
    // one workgroup per (n, c, t-chunk)
    extern __shared__ float sh[];               // FIXED: [2][R][VM], else [2][V]
    const int nc = blockIdx.x;                  // n*C + c
    const int c = nc % C;
    const int VM = V * M;
    const int tchunk = (T + gridDim.y - 1) / gridDim.y;
    const int t0 = blockIdx.y * tchunk, t1 = min(T, t0 + tchunk);
    const float* base = x + (size_t)nc * T * VM;
    const int VC = V * C;
    if constexpr (FIXED) {
        const int R = 256 / VM;
        const int r = threadIdx.x / VM, j = threadIdx.x - r * VM;
        if (r < R) {
            float s = 0.f, s2 = 0.f;
            for (int t = t0 + r; t < t1; t += R) {
                const float val = base[(size_t)t * VM + j];
                s += val;
                s2 = fmaf(val, val, s2);
            }
            sh[r * VM + j] = s;
            sh[(R + r) * VM + j] = s2;
        }
        __syncthreads();
        for (int v = threadIdx.x; v < V; v += blockDim.x) {
            float s = 0.f, s2 = 0.f;
            for (int rr = 0; rr < R; ++rr)
                for (int m = 0; m < M; ++m) { s += sh[rr * VM + v * M + m]; s2 += sh[(R + rr) * VM + v * M + m]; }
            atomicAdd(&stats[v * C + c], (double)s);
            atomicAdd(&stats[VC + v * C + c], (double)s2);
        }
    } else {
        for (int i = threadIdx.x; i < 2 * V; i += blockDim.x) sh[i] = 0.f;
        __syncthreads();
        for (int i = t0 * VM + threadIdx.x; i < t1 * VM; i += blockDim.x) {
            const float val = base[i];
            const int v = (i % VM) / M;
            atomicAdd(&sh[v], val);
            atomicAdd(&sh[V + v], val * val);
        }
        __syncthreads();
        for (int v = threadIdx.x; v < V; v += blockDim.x) {
            atomicAdd(&stats[v * C + c], (double)sh[v]);
            atomicAdd(&stats[VC + v * C + c], (double)sh[V + v]);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void data_bn_apply_kernel(const float* x, T* out, const float* scale, const float* shift,
                                                           int N, int C, int T_, int V, int M, int Cp) {
    // thread per output (b = n*M+m, t, v)
    const size_t total = (size_t)N * M * T_ * V;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int v = (int)(i % V);
        size_t rest = i / V;
        const int t = (int)(rest % T_); rest /= T_;
        const int mm = (int)(rest % M);
        const int n = (int)(rest / M);
        // one 16-byte store per VEC channels (Cp is a multiple of VEC: whole k-steps)
        constexpr int VEC = ET<T>::VEC;
        for (int c0 = 0; c0 < Cp; c0 += VEC) {
            float f[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const int c = c0 + j;
                f[j] = 0.f;
                if (c < C) {
                    const float xv = x[((((size_t)n * C + c) * T_ + t) * V + v) * M + mm];
                    f[j] = fmaf(xv, scale[v * C + c], shift[v * C + c]);
                }
            }
            *(i32x4*)((T*)out + i * Cp + c0) = pack16<T>(f);
        }
    }
}

// FIXED (V*C <= 256): thread (r, ch) keeps channel ch = v*C + c and walks frames t0 + r, t0 + r + R, ...; fixed-order
// fold over r.  Otherwise float atomics in LDS (arrival order).
template <typename T, bool FIXED>
__global__ __launch_bounds__(256) void data_bn_bwd_kernel(const T* dx0, const float* x, const float* mean, const float* rstd,
                                                         double* stats, int N, int C, int T_, int V, int M, int Cp) {
    // one workgroup per (n*M+m, t-chunk): dbeta/dgamma per (v, c) in LDS, then double atomics
    extern __shared__ float sh[];               // FIXED: [2][R][VC], else [2][VC]
    const int VC = V * C;
    const int b = blockIdx.x;
    const int n = b / M, mm = b % M;
    const int tchunk = (T_ + gridDim.y - 1) / gridDim.y;
    const int t0 = blockIdx.y * tchunk, t1 = min(T_, t0 + tchunk);
    if constexpr (FIXED) {
        const int R = 256 / VC;
        const int r = threadIdx.x / VC, ch = threadIdx.x - r * VC;
        if (r < R) {
            const int v = ch / C, c = ch - v * C;
            const float mu = mean[ch], rs = rstd[ch];
            float s = 0.f, s2 = 0.f;
#pragma unroll 4
            for (int t = t0 + r; t < t1; t += R) {
                const float d = load_elem<T>(dx0, (((size_t)b * T_ + t) * V + v) * Cp + c);
                const float xv = x[((((size_t)n * C + c) * T_ + t) * V + v) * M + mm];
                s += d;
                s2 = fmaf(d, (xv - mu) * rs, s2);
            }
            sh[r * VC + ch] = s;
            sh[(R + r) * VC + ch] = s2;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * VC; i += blockDim.x) {
            const int which = i / VC, ch = i - which * VC;
            float s = 0.f;
            for (int rr = 0; rr < R; ++rr) s += sh[(which * R + rr) * VC + ch];
            atomicAdd(&stats[i], (double)s);
        }
    } else {
        for (int i = threadIdx.x; i < 2 * VC; i += blockDim.x) sh[i] = 0.f;
        __syncthreads();
        for (int i = t0 * VC + threadIdx.x; i < t1 * VC; i += blockDim.x) {
            const int c = i % C;
            const int v = (i / C) % V;
            const int t = i / VC;
            const float d = load_elem<T>(dx0, (((size_t)b * T_ + t) * V + v) * Cp + c);
            const float xv = x[((((size_t)n * C + c) * T_ + t) * V + v) * M + mm];
            const int ch = v * C + c;
            atomicAdd(&sh[ch], d);
            atomicAdd(&sh[VC + ch], d * ((xv - mean[ch]) * rstd[ch]));
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * VC; i += blockDim.x) atomicAdd(&stats[i], (double)sh[i]);
    }
}

// ------------------------------------------------------------------------------------------
// Input augmentation for a whole batch: temporal window (crop or zero-pad by a per-clip frame shift) followed by
// a per-frame rotation / scaling / translation of channels 0 and 1.  x, out: [N][C][T][V*M] float32.
__global__ __launch_bounds__(256) void augment_kernel(const float* in, float* out, const int* shift, const float* ca, const float* sa,
                                                      const float* tx, const float* ty, int N, int C, int Tin, int Tout, int VM) {
    const size_t total = (size_t)N * Tout * VM;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int vm = (int)(e % VM);
        const size_t nt = e / VM;
        const int t = (int)(nt % Tout), n = (int)(nt / Tout);
        const int src = t + (shift ? shift[n] : 0);
        const bool inside = src >= 0 && src < Tin;
        const float* ip = in + ((size_t)n * C * Tin + (inside ? src : 0)) * VM + vm;
        float* op = out + ((size_t)n * C * Tout + t) * VM + vm;
        float x = 0.f, y = 0.f;
        if (C > 0) x = inside ? ip[0] : 0.f;
        if (C > 1) y = inside ? ip[(size_t)Tin * VM] : 0.f;
        if (ca && C > 1) {
            const size_t f = (size_t)n * Tout + t;
            const float c = ca[f], s = sa[f];
            const float nx = c * x - s * y + tx[f];
            const float ny = s * x + c * y + ty[f];
            x = nx; y = ny;
        }
        if (C > 0) op[0] = x;
        if (C > 1) op[(size_t)Tout * VM] = y;
        for (int ch = 2; ch < C; ++ch) op[(size_t)ch * Tout * VM] = inside ? ip[(size_t)ch * Tin * VM] : 0.f;
    }
}
extern "C" int stgcn_augment(const float* in, float* out, const int* shift, const float* ca, const float* sa,
                             const float* tx, const float* ty, int N, int C, int Tin, int Tout, int VM, void* stream) {
    if (!in || !out || N < 1 || C < 1 || Tin < 1 || Tout < 1 || VM < 1) return STGCN_ERR_ARG;
    if (ca && (!sa || !tx || !ty)) return STGCN_ERR_ARG;
    const size_t total = (size_t)N * Tout * VM;
    size_t g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(augment_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, in, out, shift, ca, sa, tx, ty, N, C, Tin, Tout, VM);
    return hipGetLastError() == hipSuccess ? STGCN_OK : STGCN_ERR_LAUNCH;
}

extern "C" int stgcn_data_bn_stats(const float* x, double* stats, int N, int C, int T, int V, int M, void* stream) {
    if (!x || !stats || N < 1 || C < 1 || T < 1 || V < 1 || M < 1) return STGCN_ERR_ARG;
    // every workgroup ends in 2 V double atomics on the same V*C addresses: few, long workgroups (about two per CU)
    int gy = (2 * stgcn_num_cus() + N * C - 1) / (N * C);
    if (gy > (T + 31) / 32) gy = (T + 31) / 32;
    if (gy < 1) gy = 1;
    const dim3 grid((unsigned)(N * C), (unsigned)gy);
    if (V * M <= 256)
        hipLaunchKernelGGL(data_bn_stats_kernel<true>, grid, dim3(256), 2 * (size_t)(256 / (V * M)) * V * M * sizeof(float), (hipStream_t)stream, x, stats, N, C, T, V, M);
    else
        hipLaunchKernelGGL(data_bn_stats_kernel<false>, grid, dim3(256), 2 * (size_t)V * sizeof(float), (hipStream_t)stream, x, stats, N, C, T, V, M);
    return hipGetLastError() == hipSuccess ? STGCN_OK : STGCN_ERR_LAUNCH;
}
extern "C" int stgcn_data_bn_apply(int dtype, const float* x, void* out, const float* scale, const float* shift,
                                   int N, int C, int T, int V, int M, int Cp, void* stream) {
    if (!x || !out || !scale || !shift || Cp < C) return STGCN_ERR_ARG;
    if (Cp % (dtype == STGCN_BF16 ? 8 : 4) != 0) return STGCN_ERR_ARG;       // rows are written in 16-byte pieces
    const size_t total = (size_t)N * M * T * V;
    unsigned g = (unsigned)((total + 255) / 256); if (g > 4096) g = 4096;
    if (dtype == STGCN_BF16) hipLaunchKernelGGL(data_bn_apply_kernel<bf16>, dim3(g), dim3(256), 0, (hipStream_t)stream, x, (bf16*)out, scale, shift, N, C, T, V, M, Cp);
    else if (dtype == STGCN_F32) hipLaunchKernelGGL(data_bn_apply_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, x, (float*)out, scale, shift, N, C, T, V, M, Cp);
    else return STGCN_ERR_ARG;
    return hipGetLastError() == hipSuccess ? STGCN_OK : STGCN_ERR_LAUNCH;
}
extern "C" int stgcn_data_bn_bwd(int dtype, const void* dx0, const float* x, const float* mean, const float* rstd, double* stats,
                                 int N, int C, int T, int V, int M, int Cp, void* stream) {
    if (!dx0 || !x || !mean || !rstd || !stats || Cp < C) return STGCN_ERR_ARG;
    // as data_bn_stats: every workgroup ends in 2 V C double atomics on the same addresses -- about two workgroups per CU
    int gy = (2 * stgcn_num_cus() + N * M - 1) / (N * M);
    if (gy > (T + 31) / 32) gy = (T + 31) / 32;
    if (gy < 1) gy = 1;
    const bool fixed = V * C <= 256;
    const size_t lds = 2 * (size_t)(fixed ? 256 / (V * C) : 1) * V * C * sizeof(float);
    if (lds > 64 * 1024) return STGCN_ERR_ARG;
    dim3 grid((unsigned)(N * M), (unsigned)gy);
    if (dtype != STGCN_BF16 && dtype != STGCN_F32) return STGCN_ERR_ARG;
    if (dtype == STGCN_BF16) {
        if (fixed) hipLaunchKernelGGL((data_bn_bwd_kernel<bf16, true>), grid, dim3(256), lds, (hipStream_t)stream, (const bf16*)dx0, x, mean, rstd, stats, N, C, T, V, M, Cp);
        else hipLaunchKernelGGL((data_bn_bwd_kernel<bf16, false>), grid, dim3(256), lds, (hipStream_t)stream, (const bf16*)dx0, x, mean, rstd, stats, N, C, T, V, M, Cp);
    } else {
        if (fixed) hipLaunchKernelGGL((data_bn_bwd_kernel<float, true>), grid, dim3(256), lds, (hipStream_t)stream, (const float*)dx0, x, mean, rstd, stats, N, C, T, V, M, Cp);
        else hipLaunchKernelGGL((data_bn_bwd_kernel<float, false>), grid, dim3(256), lds, (hipStream_t)stream, (const float*)dx0, x, mean, rstd, stats, N, C, T, V, M, Cp);
    }
    return hipGetLastError() == hipSuccess ? STGCN_OK : STGCN_ERR_LAUNCH;
}

// ------------------------------------------------------------------------------------------
// Head: global average pool over (T, V), mean over persons, 1x1 classifier.
template <typename T, int NT = 256>
__global__ __launch_bounds__(NT) void pool_fwd_kernel(const T* feat, float* pooled, int R, int C, int Cw) {
    // workgroup (b, s): sample b, channels [s * Cw, (s + 1) * Cw) of feat [B][R][C].  One workgroup streams all rows of its channels (a fixed summation order), so
    // the loads in flight are what counts: NT = 1024 threads, 8 loads each
    constexpr int VEC = ET<T>::VEC;
    RowMap<T, NT> m(Cw);
    feat += blockIdx.y * Cw; pooled += blockIdx.y * Cw;
    float s[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) s[j] = 0.f;
    const int b = blockIdx.x;
    if (m.active) {
        // 8 independent loads in flight per thread: one workgroup streams a whole sample, so latency is what counts
        const T* base = feat + (size_t)b * R * C + m.cv * VEC;
        int row = m.rl;
        for (; row + 7 * m.rpi < R; row += 8 * m.rpi) {
            i32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *(const i32x4*)(base + (size_t)(row + u * m.rpi) * C);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float f[VEC];
                unpack16<T>(v[u], f);
#pragma unroll
                for (int j = 0; j < VEC; ++j) s[j] += f[j];
            }
        }
        for (; row < R; row += m.rpi) {
            float f[VEC];
            unpack16<T>(*(const i32x4*)(base + (size_t)row * C), f);
#pragma unroll
            for (int j = 0; j < VEC; ++j) s[j] += f[j];
        }
    }
    __shared__ float red[NT * VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) red[threadIdx.x * VEC + j] = m.active ? s[j] : 0.f;
    __syncthreads();
    if (m.active && m.rl == 0) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float t = 0.f;
            for (int r = 0; r < m.rpi; ++r) t += red[(r * m.cvs + m.cv) * VEC + j];
            pooled[(size_t)b * C + m.cv * VEC + j] = t / (float)R;
        }
    }
}
extern "C" int stgcn_pool_fwd(int dtype, const void* feat, float* pooled, int B, int R, int C, void* stream) {
    if (!feat || !pooled || B < 1 || R < 1) return STGCN_ERR_ARG;
    const int vec = dtype == STGCN_BF16 ? ET<bf16>::VEC : ET<float>::VEC;
    // channel groups per sample: 128 channels per workgroup (the 256-channel output of the last block: two workgroups per
    // sample, 256 for the 128 samples of a 64-clip step).  A function of C alone, so that a clip's sums are added in
    // the same order whatever the batch.
    const int S = (C % 128 == 0 && C / 128 <= 8 && 128 % vec == 0) ? C / 128 : 1;
    const dim3 grid((unsigned)B, (unsigned)S);
    if (dtype == STGCN_BF16) { if (!ew_ok<bf16>(C)) return STGCN_ERR_ARG; hipLaunchKernelGGL((pool_fwd_kernel<bf16, 1024>), grid, dim3(1024), 0, (hipStream_t)stream, (const bf16*)feat, pooled, R, C, C / S); }
    else if (dtype == STGCN_F32) { if (!ew_ok<float>(C)) return STGCN_ERR_ARG; hipLaunchKernelGGL((pool_fwd_kernel<float, 1024>), grid, dim3(1024), 0, (hipStream_t)stream, (const float*)feat, pooled, R, C, C / S); }
    else return STGCN_ERR_ARG;
    return hipGetLastError() == hipSuccess ? STGCN_OK : STGCN_ERR_LAUNCH;
}

// logits[n][j] = bias[j] + sum_c W[j][c] * mean_m pooled[n*M+m][c]
__global__ void head_fc_fwd_kernel(const float* pooled, const float* W, const float* bias, float* logits, int N, int M, int C, int ncls) {
    const int n = blockIdx.x;
    extern __shared__ float pm[];               // [C]
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f;
        for (int mm = 0; mm < M; ++mm) s += pooled[((size_t)n * M + mm) * C + c];
        pm[c] = s / (float)M;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < ncls; j += blockDim.x) {
        float s = bias ? bias[j] : 0.f;
        for (int c = 0; c < C; ++c) s = fmaf(W[(size_t)j * C + c], pm[c], s);
        logits[(size_t)n * ncls + j] = s;
    }
}
// dW[j][c] = sum_n dlog[n][j]*pm[n][c]; db[j] = sum_n dlog[n][j]; dpm[n][c] = sum_j dlog[n][j] W[j][c];
// dpooled[n*M + m][c] = dpm[n][c] / M (every person gets 1/M of its clip's gradient).  dpm / dpooled: either may be null.
__global__ void head_fc_bwd_kernel(const float* __restrict__ dlog, const float* __restrict__ pooled, const float* __restrict__ W,
                                   float* __restrict__ dW, float* __restrict__ db, float* __restrict__ dpm, float* __restrict__ dpooled,
                                   int N, int M, int C, int ncls) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nW = (size_t)ncls * C, nP = (size_t)N * C;
    if (tid < nW) {
        const int j = (int)(tid / C), c = (int)(tid % C);
        // the sum over n in its natural order (the result must not depend on the unrolling), the loads of 8 samples
        // issued together
        float s = 0.f;
        int n = 0;
        for (; n + 8 <= N; n += 8) {
            float pm[8], dl[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { pm[u] = 0.f; dl[u] = dlog[(size_t)(n + u) * ncls + j]; }
            // persons outside, samples inside: eight loads in flight per person (each sample's own sum keeps its order)
            for (int mm = 0; mm < M; ++mm) {
#pragma unroll
                for (int u = 0; u < 8; ++u) pm[u] += pooled[((size_t)(n + u) * M + mm) * C + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s = fmaf(dl[u], pm[u] / (float)M, s);
        }
        for (; n < N; ++n) {
            float pm = 0.f;
            for (int mm = 0; mm < M; ++mm) pm += pooled[((size_t)n * M + mm) * C + c];
            s = fmaf(dlog[(size_t)n * ncls + j], pm / (float)M, s);
        }
        dW[tid] = s;
    } else if (tid < nW + nP) {
        const size_t i = tid - nW;
        const int n = (int)(i / C), c = (int)(i % C);
        float s = 0.f;
        for (int j = 0; j < ncls; ++j) s = fmaf(dlog[(size_t)n * ncls + j], W[(size_t)j * C + c], s);
        if (dpm) dpm[i] = s;
        if (dpooled) {
            const float sp = s / (float)M;
            for (int mm = 0; mm < M; ++mm) dpooled[((size_t)n * M + mm) * C + c] = sp;
        }
    } else if (tid < nW + nP + ncls) {
        const int j = (int)(tid - nW - nP);
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += dlog[(size_t)n * ncls + j];
        db[j] = s;
    }
}
// dfeat[b][row][c] = dpm[b / M][c] * scale    (scale = 1 / (M * R))
template <typename T>
__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* dpm, T* dfeat, int R, int C, int M, float scale) {
    constexpr int VEC = ET<T>::VEC;
    RowMap<T> m(C);
    if (!m.active) return;
    const int b = blockIdx.x;
    float f[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) f[j] = dpm[(size_t)(b / M) * C + m.cv * VEC + j] * scale;
    const i32x4 v = pack16<T>(f);
    for (int row = blockIdx.y * m.rpi + m.rl; row < R; row += gridDim.y * m.rpi)
        *(i32x4*)(dfeat + ((size_t)b * R + row) * C + m.cv * VEC) = v;
}
extern "C" int stgcn_head_fc_fwd(const float* pooled, const float* W, const float* bias, float* logits, int N, int M, int C, int ncls, void* stream) {
    if (!pooled || !W || !logits || C > 8192) return STGCN_ERR_ARG;
    hipLaunchKernelGGL(head_fc_fwd_kernel, dim3(N), dim3(256), C * sizeof(float), (hipStream_t)stream, pooled, W, bias, logits, N, M, C, ncls);
    return hipGetLastError() == hipSuccess ? STGCN_OK : STGCN_ERR_LAUNCH;
}
extern "C" int stgcn_head_fc_bwd(const float* dlogits, const float* pooled, const float* W, float* dW, float* db, float* dpm, float* dpooled,
                                 int N, int M, int C, int ncls, void* stream) {
    if (!dlogits || !pooled || !W || !dW || !db || (!dpm && !dpooled) || M < 1) return STGCN_ERR_ARG;
    const size_t total = (size_t)ncls * C + (size_t)N * C + ncls;
    hipLaunchKernelGGL(head_fc_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dlogits, pooled, W, dW, db, dpm, dpooled, N, M, C, ncls);
    return hipGetLastError() == hipSuccess ? STGCN_OK : STGCN_ERR_LAUNCH;
}
extern "C" int stgcn_pool_bwd(int dtype, const float* dpm, void* dfeat, int B, int R, int C, int M, void* stream) {
    if (!dpm || !dfeat || B < 1 || R < 1 || M < 1) return STGCN_ERR_ARG;
    const float scale = 1.f / ((float)M * (float)R);
    dim3 grid((unsigned)B, 8);
    if (dtype == STGCN_BF16) { if (!ew_ok<bf16>(C)) return STGCN_ERR_ARG; hipLaunchKernelGGL(pool_bwd_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, dpm, (bf16*)dfeat, R, C, M, scale); }
    else if (dtype == STGCN_F32) { if (!ew_ok<float>(C)) return STGCN_ERR_ARG; hipLaunchKernelGGL(pool_bwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, dpm, (float*)dfeat, R, C, M, scale); }
    else return STGCN_ERR_ARG;
    return hipGetLastError() == hipSuccess ? STGCN_OK : STGCN_ERR_LAUNCH;
}

// ------------------------------------------------------------------------------------------
// out[i][e] = a[e] * in_i[e] for up to MM_MAX inputs in one launch (in_i null: zeros).  A * importance_i of every block
// (net/st_gcn.py:81-82 multiplies per block) and, backward, A * dA_i.
constexpr int MM_MAX = 16;
struct MulManyArgs { const float* a; const float* in[MM_MAX]; float* out; unsigned len; int n; };
__global__ __launch_bounds__(256) void mul_many_kernel(MulManyArgs m) {
    const unsigned e = blockIdx.x * 256 + threadIdx.x;
    if (e >= m.len) return;
    const float av = m.a[e];
    for (int i = 0; i < m.n; ++i) {
        const float* p = m.in[i];
        m.out[(size_t)i * m.len + e] = p ? av * p[e] : 0.f;
    }
}
extern "C" int stgcn_mul_many(const float* a, const float* const* inputs, int n, float* out, size_t len, void* stream) {
    if (!a || !inputs || !out || n < 1 || n > MM_MAX || len == 0 || len > 0x7fffffffu) return STGCN_ERR_ARG;
    MulManyArgs m{};
    m.a = a; m.out = out; m.len = (unsigned)len; m.n = n;
    for (int i = 0; i < n; ++i) m.in[i] = inputs[i];
    hipLaunchKernelGGL(mul_many_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, (hipStream_t)stream, m);
    return hipGetLastError() == hipSuccess ? STGCN_OK : STGCN_ERR_LAUNCH;
}

// ------------------------------------------------------------------------------------------
// Mean cross-entropy of logits [N][ncls] against labels [N] and its gradient (softmax - onehot) / N in one launch:
// one wave per sample (lanes over classes, fixed butterflies), samples dealt to the sixteen waves of ONE workgroup, the
// per-sample losses added in sample order -- the same bits every run.
__global__ __launch_bounds__(1024) void cross_entropy_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                                                            float* __restrict__ loss, float* __restrict__ dlogits, int N, int ncls) {
    extern __shared__ float per_sample[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv_n = 1.f / (float)N;
    for (int n = wave; n < N; n += 16) {
        const float* x = logits + (size_t)n * ncls;
        float mx = -INFINITY;
        for (int j = lane; j < ncls; j += 64) mx = fmaxf(mx, x[j]);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
        float se = 0.f;
        for (int j = lane; j < ncls; j += 64) se += expf(x[j] - mx);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) se += __shfl_xor(se, d);
        const float lse = logf(se);
        const int y = (int)labels[n];
        for (int j = lane; j < ncls; j += 64) {
            const float p = expf(x[j] - mx - lse);
            dlogits[(size_t)n * ncls + j] = (p - (j == y ? 1.f : 0.f)) * inv_n;
        }
        if (lane == 0) per_sample[n] = -(x[y] - mx - lse);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += per_sample[n];
        *loss = s * inv_n;
    }
}
extern "C" int stgcn_cross_entropy(const float* logits, const int64_t* labels, float* loss, float* dlogits, int N, int ncls, void* stream) {
    if (!logits || !labels || !loss || !dlogits || N < 1 || ncls < 1 || N > 8192) return STGCN_ERR_ARG;
    hipLaunchKernelGGL(cross_entropy_kernel, dim3(1), dim3(1024), (size_t)N * sizeof(float), (hipStream_t)stream,
                       logits, (const long long*)labels, loss, dlogits, N, ncls);
    return hipGetLastError() == hipSuccess ? STGCN_OK : STGCN_ERR_LAUNCH;
}

// ------------------------------------------------------------------------------------------
// SGD with momentum / Nesterov / weight decay over a flat f32 buffer (torch.optim.SGD semantics).
__global__ void sgd_kernel(float* p, const float* g, float* mom, size_t n, float lr, float momentum, float wd, int nesterov, int first, float gscale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float w = p[i];
        const float d = fmaf(wd, w, g[i] * gscale);
        const float b = first ? d : fmaf(momentum, mom[i], d);
        mom[i] = b;
        const float step = nesterov ? fmaf(momentum, b, d) : b;
        p[i] = w - lr * step;
    }
}
extern "C" int stgcn_sgd_step(float* params, const float* grads, float* momentum_buf, size_t n,
                              float lr, float momentum, float weight_decay, int nesterov, int first_step, float grad_scale, void* stream) {
    if (!params || !grads || !momentum_buf) return STGCN_ERR_ARG;
    if (n == 0) return STGCN_OK;
    size_t g = (n + 255) / 256; if (g > 2048) g = 2048;
    hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, params, grads, momentum_buf, n, lr, momentum, weight_decay, nesterov, first_step, grad_scale);
    return hipGetLastError() == hipSuccess ? STGCN_OK : STGCN_ERR_LAUNCH;
}

// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// Weight packing for a whole model in one launch.  One descriptor per convolution weight (float32, the
// reference's shapes): element (g, co, ci) is src[g*sg + co*sco + ci*sci] -- g is the adjacency partition of a
// graph-convolution weight [K*Cout, Cin, 1, 1] or the tap of a temporal weight [Cout, Cin, kt, 1].  Written, in the
// compute dtype, as  w[g][co][cip]  (the projection / forward operand) and  wt[g][cip][co]  (the input-gradient
// operand), cip < Cinp zero-padded past Cin.  Either destination may be null.
struct PackDesc {
    const float* src; void* w; void* wt;
    long long G, Cout, Cin, Cinp, sg, sco, sci;
};

template <typename T>
__global__ void pack_weights_kernel(const PackDesc* table) {
    const PackDesc d = table[blockIdx.y];
    const long long n = d.G * d.Cout * d.Cinp;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const long long cip = e % d.Cinp, co = (e / d.Cinp) % d.Cout, g = e / (d.Cinp * d.Cout);
        const float v = cip < d.Cin ? d.src[g * d.sg + co * d.sco + cip * d.sci] : 0.f;
        if (d.w) store_elem<T>((T*)d.w, (size_t)e, v);
        if (d.wt) store_elem<T>((T*)d.wt, (size_t)((g * d.Cinp + cip) * d.Cout + co), v);
    }
}

// table: ndesc descriptors of 10 x 8 bytes each, in device memory; max_elems = largest G*Cout*Cinp among them
extern "C" int stgcn_pack_weights(int dtype, const void* table, int ndesc, size_t max_elems, void* stream) {
    if (!table || ndesc < 1 || max_elems < 1) return STGCN_ERR_ARG;
    static_assert(sizeof(PackDesc) == 80, "descriptor layout is 10 x 8 bytes");
    unsigned gx = (unsigned)((max_elems + 255) / 256);
    if (gx > 1024) gx = 1024;
    const dim3 grid(gx, (unsigned)ndesc);
    if (dtype == STGCN_BF16) hipLaunchKernelGGL(pack_weights_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, (const PackDesc*)table);
    else if (dtype == STGCN_F32) hipLaunchKernelGGL(pack_weights_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const PackDesc*)table);
    else return STGCN_ERR_ARG;
    return hipGetLastError() == hipSuccess ? STGCN_OK : STGCN_ERR_LAUNCH;
}

extern "C" int stgcn_version(void) { return 1; }
extern "C" const char* stgcn_arch(void) { return "gfx950"; }
extern "C" int stgcn_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return -1; return n; }
extern "C" int stgcn_launch_state(int device, int* num_cus, int* lds_opt_ins) {
    if (device < 0 || device >= STGCN_MAX_DEV) return STGCN_ERR_ARG;
    if (num_cus) *num_cus = g_stgcn_cus[device].load(std::memory_order_relaxed);
    if (lds_opt_ins) *lds_opt_ins = g_stgcn_attr_sets[device].load(std::memory_order_relaxed);
    return STGCN_OK;
}

// Diagnostic: what ds_read_b64_tr_b16 hands each lane for a [32 rows][32 x 16-bit] LDS image whose
// element (row, col) holds the raw pattern row*256 + col, addressed exactly as the wgrad kernels do.
__global__ void probe_tr16_kernel(short* out) {
    __shared__ __attribute__((aligned(16))) short img[32 * 40];
    for (int i = threadIdx.x; i < 32 * 32; i += 64) img[(i >> 5) * 40 + (i & 31)] = (short)(((i >> 5) << 8) | (i & 31));
    __syncthreads();
    const int lane = threadIdx.x, h = lane >> 5;
    const int q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = (lane >> 4) & 1;
    const int row0 = 8 * h + q4, col = 16 * g1 + 4 * p4;
    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(&img[row0 * 40 + col]));
    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(&img[(row0 + 4) * 40 + col]));
#pragma unroll
    for (int j = 0; j < 4; ++j) { out[lane * 8 + j] = v0[j]; out[lane * 8 + 4 + j] = v1[j]; }
}
extern "C" int stgcn_probe_tr16(short* out, void* stream) {
    if (!out) return STGCN_ERR_ARG;
    hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
    return hipGetLastError() == hipSuccess ? STGCN_OK : STGCN_ERR_LAUNCH;
}
