// Fixed-order sums of partial slabs: the second stage of every parameter-gradient kernel (temporal-convolution
// weights, graph-convolution weights and bias, the adjacency gradient).  One launch serves up to STGCN_MAX_PENDING
// sums; a workgroup finds its sum from the table of first workgroup indices.
#include "stgcn_common.h"

struct SumBatch {
    StgcnPendingSum d[STGCN_MAX_PENDING];
    unsigned first_wg[STGCN_MAX_PENDING + 1];
    int n;
};

__global__ __launch_bounds__(256) void sum_pending_kernel(SumBatch b) {
    __shared__ float red[256];
    int j = 0;
    while (j + 1 < b.n && blockIdx.x >= b.first_wg[j + 1]) ++j;
    const StgcnPendingSum d = b.d[j];
    const unsigned wg = blockIdx.x - b.first_wg[j];
    const size_t n = d.n1 + d.n2;
    if (d.kind == STGCN_SUM_WAVE) {
        // lane l adds parts l, l + 64, ...; the lanes are folded with a fixed butterfly
        const size_t i = (size_t)wg * 4 + (threadIdx.x >> 6);
        const int lane = threadIdx.x & 63;
        if (i >= n) return;
        float s = 0.f;
        for (int p = lane; p < d.nparts; p += 64) s += d.slab[(size_t)p * n + i];
#pragma unroll
        for (int w = 32; w >= 1; w >>= 1) s += __shfl_xor(s, w);
        if (lane == 0) {
            if (i < d.n1) d.out1[i] = s;
            else if (d.out2) d.out2[i - d.n1] = s;
        }
        return;
    }
    if (d.kind == STGCN_SUM_CAST) {
        const size_t i = (size_t)wg * 256 + threadIdx.x;
        if (i < n) d.out1[i] = (float)((const double*)d.slab)[i];
        return;
    }
    const size_t i = (size_t)wg * 64 + (threadIdx.x & 63);
    const float s = slab_sum64(d.slab, d.nparts, n, i, red);
    if (threadIdx.x < 64 && i < n) {
        if (d.kind == STGCN_SUM_CONV_W) {
            const int ci = (int)(i % d.cin);
            const size_t rest = i / d.cin;
            const int co = (int)(rest % d.cout);
            const int tap = (int)(rest / d.cout);
            d.out1[((size_t)co * d.cin + ci) * d.ntaps + tap] = s;
        } else if (i < d.n1) {
            d.out1[i] = s;
        } else if (d.out2) {
            d.out2[i - d.n1] = s;
        }
    }
}

static int launch_sums(const StgcnPendingSum* d, int n, hipStream_t stream) {
    if (n <= 0) return STGCN_OK;
    if (n > STGCN_MAX_PENDING) return STGCN_ERR_ARG;
    SumBatch b;
    unsigned wgs = 0;
    for (int j = 0; j < n; ++j) {
        const StgcnPendingSum& s = d[j];
        if (!s.slab || !s.out1 || s.nparts < 1 || s.n1 + s.n2 == 0) return STGCN_ERR_ARG;
        if (s.kind == STGCN_SUM_CONV_W && ((size_t)s.ntaps * s.cout * s.cin != s.n1 || s.n2 != 0)) return STGCN_ERR_ARG;
        b.d[j] = s;
        b.first_wg[j] = wgs;
        const size_t e = s.n1 + s.n2;
        wgs += s.kind == STGCN_SUM_WAVE ? (unsigned)((e + 3) / 4) : s.kind == STGCN_SUM_CAST ? (unsigned)((e + 255) / 256) : slab_sum_grid(e);
    }
    for (int j = n; j <= STGCN_MAX_PENDING; ++j) b.first_wg[j] = wgs;
    for (int j = n; j < STGCN_MAX_PENDING; ++j) b.d[j] = StgcnPendingSum{};
    b.n = n;
    hipLaunchKernelGGL(sum_pending_kernel, dim3(wgs), dim3(256), 0, stream, b);
    return hipGetLastError() == hipSuccess ? STGCN_OK : STGCN_ERR_LAUNCH;
}

int stgcn_sum_or_defer(StgcnSumList* list, const StgcnPendingSum& s, hipStream_t stream) {
    if (!list) return launch_sums(&s, 1, stream);
    if (list->n < 0 || list->n >= STGCN_MAX_PENDING) return STGCN_ERR_LIST_FULL;
    list->d[list->n++] = s;
    return STGCN_OK;
}

int stgcn_sum_list_launch(StgcnSumList* list, hipStream_t stream) {
    if (!list || list->n < 0 || list->n > STGCN_MAX_PENDING) return STGCN_ERR_ARG;
    const int rc = launch_sums(list->d, list->n, stream);
    if (rc == STGCN_OK) list->n = 0;
    return rc;
}

extern "C" int stgcn_sum_list_add_cast(void* sum_list, const double* src, float* dst, size_t n) {
    if (!sum_list || !src || !dst || n == 0) return STGCN_ERR_ARG;
    StgcnPendingSum s{};
    s.slab = (const float*)src; s.out1 = dst; s.n1 = n; s.nparts = 1; s.kind = STGCN_SUM_CAST;
    return stgcn_sum_or_defer((StgcnSumList*)sum_list, s, nullptr);
}

extern "C" int stgcn_sum_list_run(void* sum_list, void* stream) {
    return stgcn_sum_list_launch((StgcnSumList*)sum_list, (hipStream_t)stream);
}
