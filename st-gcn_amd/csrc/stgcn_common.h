// Shared device helpers for the ST-GCN gfx950 kernels.
//
// Data layout used by every kernel in this library: activations are "channels-last",
// physical [B][T][V][C] with C contiguous (a torch tensor of logical shape [B,C,T,V] in
// torch.channels_last memory format).  Rows of the implicit GEMMs are the flattened
// (t, v) positions of one sample; the contraction index (channels) is contiguous, which
// is what the 32x32 MFMA operand fragments want: lane (r = lane&31, h = lane>>5) holds
// 16 contiguous bytes of row r at byte offset 32*s + 16*h of k-step s.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

#define STGCN_OK 0
#define STGCN_ERR_ARG 1
#define STGCN_ERR_LAUNCH 2
#define STGCN_ERR_WORKSPACE 3
#define STGCN_ERR_SCRATCH 4   // the problem needs more than the library's static scratch array holds
#define STGCN_ERR_LIST_FULL 5 // a sum list has no room for another pending sum: run it first

#define STGCN_F32 0
#define STGCN_BF16 1

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

struct bf16 { uint16_t v; };   // storage type tag for bf16 tensors

template <typename T> struct ET;
template <> struct ET<float> {
    static constexpr int ES = 4;        // bytes per element
    static constexpr int KSTEP = 8;     // elements per 32-byte k-step
    static constexpr int VEC = 4;       // elements per 16 bytes
    static constexpr bool IS_BF16 = false;
};
template <> struct ET<bf16> {
    static constexpr int ES = 2;
    static constexpr int KSTEP = 16;
    static constexpr int VEC = 8;
    static constexpr bool IS_BF16 = true;
};

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t bits16) { return __uint_as_float(bits16 << 16); }

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    f32x2 v = {lo, hi};
    bf16x2_t r = __builtin_convertvector(v, bf16x2_t);   // round-to-nearest-even, NaN stays NaN
    return __builtin_bit_cast(uint32_t, r);
}

// 16 bytes of T <-> floats (VEC = 4 for f32, 8 for bf16)
template <typename T> __device__ __forceinline__ void unpack16(const i32x4& raw, float* f) {
    if constexpr (ET<T>::IS_BF16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t u = (uint32_t)raw[j];
            f[2 * j] = __uint_as_float(u << 16);
            f[2 * j + 1] = __uint_as_float(u & 0xFFFF0000u);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = __int_as_float(raw[j]);
    }
}
template <typename T> __device__ __forceinline__ i32x4 pack16(const float* f) {
    i32x4 raw;
    if constexpr (ET<T>::IS_BF16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) raw[j] = (int)pack_bf16x2(f[2 * j], f[2 * j + 1]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) raw[j] = __float_as_int(f[j]);
    }
    return raw;
}

template <typename T> __device__ __forceinline__ float load_elem(const T* p, size_t i) {
    if constexpr (ET<T>::IS_BF16) return bf16_bits_to_f32(((const uint16_t*)p)[i]);
    else return ((const float*)p)[i];
}
template <typename T> __device__ __forceinline__ void store_elem(T* p, size_t i, float v) {
    if constexpr (ET<T>::IS_BF16) ((uint16_t*)p)[i] = (uint16_t)(pack_bf16x2(v, 0.f) & 0xFFFFu);
    else ((float*)p)[i] = v;
}

// One 32-byte k-step of a 32x32 tile product: acc += A_frag * B_frag.
//   bf16: one v_mfma_f32_32x32x16_bf16 (lane half h holds k = 8h..8h+7 of the step).
//   f32 : four v_mfma_f32_32x32x2_f32; lane half h holds k = 4h..4h+3 of the 8-wide step
//         and MFMA j contracts k = {j, 4+j}.  A and B use the same k permutation.
template <typename T> __device__ __forceinline__ void mma_step(f32x16& acc, const i32x4& a, const i32x4& b) {
    if constexpr (ET<T>::IS_BF16) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8, a), __builtin_bit_cast(s16x8, b), acc, 0, 0, 0);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__int_as_float(a[j]), __int_as_float(b[j]), acc, 0, 0, 0);
    }
}

// Row of a 32x32 MFMA accumulator held in register r of lane half h.
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Dropout keep decisions.  One 32-bit hash serves the two elements 2q and 2q+1 of the flat channels-last tensor:
// its low half decides the even one, its high half the odd one, each kept when its 16 bits are >= thr16
// (= round(p * 65536), so p is honoured to 1/65536).
// A launch whose seed comes with a device-side step counter uses seed + counter * STGCN_SEED_STEP_MUL (mod 2^32).
#define STGCN_SEED_STEP_MUL 0x85EBCA6Bu
__device__ __forceinline__ uint32_t drop_hash(uint64_t pair, uint32_t seed) {
    uint32_t x = (uint32_t)pair ^ seed;
    x += (uint32_t)(pair >> 32) * 0x9E3779B9u;
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
// keep bits of the NV (even) consecutive elements starting at the even flat offset off: bit j = element off+j kept
template <int NV> __device__ __forceinline__ uint32_t drop_keep_bits(uint64_t off, uint32_t seed, uint32_t thr16) {
    uint32_t bits = 0;
#pragma unroll
    for (int q = 0; q < NV / 2; ++q) {
        const uint32_t x = drop_hash((off >> 1) + q, seed);
        bits |= ((x & 0xFFFFu) >= thr16 ? 1u : 0u) << (2 * q);
        bits |= ((x >> 16) >= thr16 ? 1u : 0u) << (2 * q + 1);
    }
    return bits;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding global load and
// store of the wave (s_waitcnt vmcnt(0)), which serialises register prefetches and epilogue stores against the
// barrier; kernels whose waves exchange data through LDS alone use this instead.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ i32x4 zero16() { i32x4 z = {0, 0, 0, 0}; return z; }

static inline int ceil_div_i(int a, int b) { return (a + b - 1) / b; }

// ---- per-device launch state -------------------------------------------------------------------------------------
// A process may drive several GPUs (one host thread per device, or one thread switching devices).  Whatever a
// launcher learns or sets about "the device" is therefore kept per device ordinal, behind atomics so that the first
// calls of two host threads may race: both then do the same idempotent work.
constexpr int STGCN_MAX_DEV = 64;
inline std::atomic<int> g_stgcn_cus[STGCN_MAX_DEV];          // compute units of device d (0 = not asked yet)
inline std::atomic<int> g_stgcn_attr_sets[STGCN_MAX_DEV];    // diagnostics: large-LDS opt-ins done on device d

static inline int stgcn_cur_device() { int d = 0; if (hipGetDevice(&d) != hipSuccess) d = 0; return d; }

// Compute units of the current device (MI355X: 256); grids of the persistent kernels are sized from it.
static inline int stgcn_num_cus() {
    const int d = stgcn_cur_device();
    const bool slot = d >= 0 && d < STGCN_MAX_DEV;
    int c = slot ? g_stgcn_cus[d].load(std::memory_order_relaxed) : 0;
    if (c <= 0) {
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || c <= 0) c = 256;
        if (slot) g_stgcn_cus[d].store(c, std::memory_order_relaxed);
    }
    return c;
}

// Kernels that use more than 64 KiB of dynamic LDS must opt in with hipFuncSetAttribute, and the attribute belongs to
// the (function, device) pair.  `done` is the launcher's own bit set of devices already served (bit d = ordinal d).
static inline int stgcn_allow_big_lds(const void* fn, std::atomic<uint64_t>& done) {
    const int d = stgcn_cur_device();
    const uint64_t bit = (d >= 0 && d < STGCN_MAX_DEV) ? (1ull << d) : 0ull;
    if (bit && (done.load(std::memory_order_acquire) & bit)) return STGCN_OK;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return STGCN_ERR_LAUNCH;
    if (bit) { done.fetch_or(bit, std::memory_order_release); g_stgcn_attr_sets[d].fetch_add(1, std::memory_order_relaxed); }
    return STGCN_OK;
}

// Fixed-order reduction of a partial-sum slab [nparts][n] (f32) in one launch.  A workgroup of 256 threads is
// 64 consecutive elements x 4 part groups: thread (o = tid & 63, g = tid >> 6) adds parts g, g+4, g+8, ... of element
// blockIdx.x*64 + o (four loads in flight), the four group sums are added in the order 0..3 through LDS.  Every
// element is summed in the same order every run, so results are bitwise reproducible.  The total is returned to the
// threads with g == 0; `red` is 256 floats of LDS.  Launch with slab_sum_grid(n) workgroups of 256 threads.
__device__ __forceinline__ float slab_sum64(const float* slab, int nparts, size_t n, size_t i, float* red) {
    const int g = threadIdx.x >> 6;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (i < n) {
        int p = g;
        for (; p + 12 < nparts; p += 16) {
            a0 += slab[(size_t)p * n + i];
            a1 += slab[(size_t)(p + 4) * n + i];
            a2 += slab[(size_t)(p + 8) * n + i];
            a3 += slab[(size_t)(p + 12) * n + i];
        }
        for (; p < nparts; p += 4) a0 += slab[(size_t)p * n + i];
    }
    red[threadIdx.x] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    float tot = 0.f;
    if (g == 0) tot = (red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192]);
    return tot;
}
static inline unsigned slab_sum_grid(size_t n) { return (unsigned)((n + 63) / 64); }

// ---- BatchNorm coefficients from statistic sums ------------------------------------------------------------------
// One definition for stgcn_bn_finalize and for the kernels that fold it into their own launch: both must arrive at the
// same floats.
struct BnCoeff { float mean, rstd, scale, shift, unbiased_var; };
__device__ __forceinline__ BnCoeff bn_coeff_from_sums(double s1, double s2, double count, float gamma, float beta, float eps) {
    const double m = s1 / count;
    double v = s2 / count - m * m;
    if (v < 0) v = 0;
    BnCoeff k;
    k.mean = (float)m;
    k.unbiased_var = (float)(count > 1 ? v * (count / (count - 1)) : v);
    k.rstd = 1.0f / sqrtf((float)v + eps);
    k.scale = gamma * k.rstd;
    k.shift = fmaf(-k.mean, k.scale, beta);
    return k;
}

// The BatchNorm + ReLU ahead of a temporal convolution, finished by the convolution's own launch (train mode).  The
// convolution kernels read their prologue's scale / shift from plain arrays; with `stats` set those arrays are
// out + 2C and out + 3C of `out` (float[4][C]: mean, rstd, scale, shift) and EVERY workgroup first derives all C
// channels from the sums and writes them there -- the same values from every workgroup, so it does not matter whose
// store a line holds -- and reads back only what it has written itself, behind a workgroup barrier (a workgroup sees
// its own global stores after __syncthreads(); it never depends on another workgroup's).  The kernels that follow
// (weight gradient, data gradient, BatchNorm backward) read `out` across the kernel boundary.  Workgroup (0, 0, 0)
// also advances the running statistics.
struct BnPrologue {
    const double* stats; const float* gamma; const float* beta; float* rmean; float* rvar; float* out;
    double count; float momentum, eps; int C;
};
__device__ __forceinline__ void bn_prologue_preamble(const BnPrologue& f) {
    if (!f.stats) return;
    const bool first = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0;
    for (int c = threadIdx.x; c < f.C; c += blockDim.x) {
        const BnCoeff k = bn_coeff_from_sums(f.stats[c], f.stats[f.C + c], f.count, f.gamma[c], f.beta[c], f.eps);
        f.out[c] = k.mean; f.out[f.C + c] = k.rstd; f.out[2 * f.C + c] = k.scale; f.out[3 * f.C + c] = k.shift;
        if (first && f.rmean) {
            f.rmean[c] = (1.f - f.momentum) * f.rmean[c] + f.momentum * k.mean;
            f.rvar[c] = (1.f - f.momentum) * f.rvar[c] + f.momentum * k.unbiased_var;
        }
    }
    __syncthreads();
}

// ---- pending sums ------------------------------------------------------------------------------------------------
// The kernels that produce parameter gradients write per-workgroup partials [nparts][n1 + n2] (f32) and leave the
// fixed-order sum over the parts to a second launch.  A launcher describes that sum with a StgcnPendingSum and hands
// it to stgcn_sum_or_defer: with no list the sum is launched at once, with a list it is appended and the caller runs
// the whole list (sums of several launchers, one launch) with stgcn_sum_list_launch.  Either way the same device code
// adds the same parts in the same order.
enum { STGCN_SUM_PLAIN = 0,      // out1[i] (i < n1) and out2[i - n1] (out2 may be null) = sum over parts
       STGCN_SUM_WAVE = 1,       // few elements, many parts: one wave per element, lanes fold with a fixed butterfly
       STGCN_SUM_CONV_W = 2,     // element i = (tap, co, ci) of [ntaps][cout][cin] goes to out1[(co*cin + ci)*ntaps + tap]
       STGCN_SUM_CAST = 3 };     // no sum: `slab` holds n1 doubles (finished sums of double atomics), out1[i] = (float)slab[i]
struct StgcnPendingSum {
    const float* slab; float* out1; float* out2;
    unsigned long long n1, n2;
    int nparts, kind, ntaps, cout, cin, reserved;
};
constexpr int STGCN_MAX_PENDING = 8;
struct StgcnSumList { int n; int reserved; StgcnPendingSum d[STGCN_MAX_PENDING]; };   // host memory, caller-owned
static_assert(sizeof(StgcnSumList) <= 1024, "a sum list fits the size the public header promises");

int stgcn_sum_or_defer(StgcnSumList* list, const StgcnPendingSum& s, hipStream_t stream);
int stgcn_sum_list_launch(StgcnSumList* list, hipStream_t stream);
// bytes of [base, ...) that the slabs of the list's entries from index `first` on reach into
static inline size_t stgcn_sum_list_extent(const StgcnSumList* list, int first, const void* base) {
    size_t e = 0;
    if (!list) return 0;
    for (int i = first; i < list->n; ++i) {
        const StgcnPendingSum& d = list->d[i];
        if (d.kind == STGCN_SUM_CAST) continue;           // reads the caller's own buffer, not the workspace
        const size_t end = (size_t)((const char*)(d.slab + (size_t)d.nparts * (d.n1 + d.n2)) - (const char*)base);
        if (end > e) e = end;
    }
    return e;
}
