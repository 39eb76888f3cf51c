// conv_rows: convolution along T of a channels-last activation, as an implicit GEMM on MFMA.
//
//   out[b, to, v, co] = bias[co] + sum_tap sum_ci  pro(in[b, ti(to,tap), v, ci]) * w[tap][co][ci]
//     forward form    : ti = stride*to - pad + tap*dil
//     transposed form : ti = (to + pad - tap*dil) / stride, used only when divisible
//   zero where ti is outside [0, Tin).  pro(x) = relu(scale[ci]*x + shift[ci]) when a prologue
//   is given (the BatchNorm+ReLU that precedes the temporal conv), identity otherwise.
//
// One workgroup (4 waves) owns BM = 128*WM output rows (flattened (to, v) of one sample) and
// BN = 32*NT output channels.  Per 128-byte channel chunk it stages ONCE the window of input
// rows all taps need (with the prologue applied once per element) into LDS; the taps then read
// their A fragments from that window at shifted rows, so the 9-tap temporal conv reads each
// input row from L2 about (BM + 8V)/BM times instead of 9.  Per-tap weight tiles are
// double-buffered in LDS and prefetched through registers while the previous tap computes.
#include "stgcn_common.h"

struct ConvRowsArgs {
    const void* in; const void* w; void* out;
    const float* bias;
    const float* pro_scale; const float* pro_shift; BnPrologue pbn;
    const void* aux; const float* ep_scale; const float* ep_shift; const float* ep_mean; const float* ep_rstd;
    double* stats;
    int B, Tin, Tout, V, Cin, Cout;
    int ntaps, stride, pad, dil, transposed;
    int win_rows_max, tiles_per_sample;
};

#define EPI_PLAIN 0      // out = acc + bias ; stats = sum(out), sum(out^2)
#define EPI_RELU_BWD 1   // out = (ep_scale*aux+ep_shift > 0) ? acc : 0 ; stats = sum(out), sum(out*xhat(aux))
#define EPI_ADD_RELU 2   // out = max(acc + bias + aux, 0) (aux optional): the block's output stage when BatchNorm is folded away

constexpr int CR_RS = 144;   // LDS row stride: 128 data bytes + 16 pad (conflict-free ds_read_b128)

template <typename T, int WM, int NT, int EPI>
__global__ __launch_bounds__(256) void conv_rows_kernel(ConvRowsArgs a) {
    bn_prologue_preamble(a.pbn);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ES = ET<T>::ES;
    constexpr int KC = 128 / ES;           // channels per chunk
    constexpr int VEC = ET<T>::VEC;
    constexpr int BM = 128 * WM;
    constexpr int BN = 32 * NT;
    const int ZR = a.win_rows_max;         // index of the all-zero window row
    char* win = smem;
    char* wbuf = smem + (size_t)(a.win_rows_max + 1) * CR_RS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.x / a.tiles_per_sample;
    const int tile = blockIdx.x - b * a.tiles_per_sample;
    const int co0 = blockIdx.y * BN;
    const int V = a.V;
    const int Rout = a.Tout * V;
    const int r0 = tile * BM;
    const int r1 = min(r0 + BM, Rout);

    // input frames any row of this tile can touch
    const int tlo = r0 / V, thi = (r1 - 1) / V;
    int flo, fhi;
    if (!a.transposed) {
        flo = a.stride * tlo - a.pad;
        fhi = a.stride * thi - a.pad + (a.ntaps - 1) * a.dil;
    } else {
        const int nmin = tlo + a.pad - (a.ntaps - 1) * a.dil;
        flo = nmin <= 0 ? 0 : (nmin + a.stride - 1) / a.stride;
        fhi = (thi + a.pad) / a.stride;
    }
    flo = max(flo, 0);
    fhi = min(fhi, a.Tin - 1);
    const int wrows = max(0, (fhi - flo + 1) * V);

    if (tid < CR_RS / 16) *(i32x4*)(win + (size_t)ZR * CR_RS + tid * 16) = zero16();

    // per-lane output rows (one per row-subtile)
    int t_out[WM], vj[WM];
    bool rvalid[WM];
#pragma unroll
    for (int m = 0; m < WM; ++m) {
        const int r = r0 + (wave * WM + m) * 32 + l31;
        rvalid[m] = r < r1;
        const int rr = rvalid[m] ? r : r0;
        t_out[m] = rr / V;
        vj[m] = rr - t_out[m] * V;
    }

    f32x16 acc[WM][NT];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const int nchunks = a.Cin / KC;
    const int nsteps = nchunks * a.ntaps;
    const char* inb = (const char*)a.in + ((size_t)b * a.Tin + flo) * V * (size_t)a.Cin * ES;
    const size_t in_rs = (size_t)a.Cin * ES;

    // weight tile loader: NT pieces of 16 B per thread
    i32x4 wregs[NT];
    auto load_w = [&](int step) {
        const int chunk = step / a.ntaps, tap = step - chunk * a.ntaps;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int p = tid + 256 * i;
            const int n = p >> 3, seg = p & 7;
            const int co = co0 + n;
            if (co < a.Cout)
                wregs[i] = *(const i32x4*)((const char*)a.w + (((size_t)tap * a.Cout + co) * a.Cin + (size_t)chunk * KC) * ES + seg * 16);
            else
                wregs[i] = zero16();
        }
    };
    auto store_w = [&](int buf) {
        char* wb = wbuf + (size_t)buf * BN * CR_RS;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int p = tid + 256 * i;
            const int n = p >> 3, seg = p & 7;
            *(i32x4*)(wb + n * CR_RS + seg * 16) = wregs[i];
        }
    };

    load_w(0);
    for (int step = 0; step < nsteps; ++step) {
        const int chunk = step / a.ntaps, tap = step - chunk * a.ntaps;
        if (tap == 0) {
            if (step > 0) __syncthreads();     // every wave is done reading the previous window
            const int npieces = wrows * 8;
            for (int p0 = 0; p0 < npieces; p0 += 256 * 4) {
                i32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int p = p0 + u * 256 + tid;
                    if (p < npieces) v[u] = *(const i32x4*)(inb + (size_t)(p >> 3) * in_rs + (size_t)chunk * 128 + (p & 7) * 16);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int p = p0 + u * 256 + tid;
                    if (p < npieces) {
                        const int row = p >> 3, seg = p & 7;
                        if (a.pro_scale) {
                            float f[VEC];
                            unpack16<T>(v[u], f);
                            const int c0 = chunk * KC + seg * VEC;
#pragma unroll
                            for (int j = 0; j < VEC; ++j) f[j] = fmaxf(fmaf(f[j], a.pro_scale[c0 + j], a.pro_shift[c0 + j]), 0.f);
                            v[u] = pack16<T>(f);
                        }
                        *(i32x4*)(win + (size_t)row * CR_RS + seg * 16) = v[u];
                    }
                }
            }
        }
        store_w(step & 1);
        __syncthreads();
        if (step + 1 < nsteps) load_w(step + 1);   // in flight while this tap computes

        // A-fragment base of each row subtile for this tap
        int aoff[WM];
#pragma unroll
        for (int m = 0; m < WM; ++m) {
            int ti;
            bool ok = rvalid[m];
            if (!a.transposed) {
                ti = a.stride * t_out[m] - a.pad + tap * a.dil;
            } else {
                const int num = t_out[m] + a.pad - tap * a.dil;
                ti = num / a.stride;
                ok = ok && num >= 0 && (num - ti * a.stride) == 0;
            }
            ok = ok && ti >= 0 && ti < a.Tin;
            const int lrow = ok ? (ti - flo) * V + vj[m] : ZR;
            aoff[m] = lrow * CR_RS + h * 16;
        }
        const char* wb = wbuf + (size_t)(step & 1) * BN * CR_RS + l31 * CR_RS + h * 16;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            i32x4 af[WM];
#pragma unroll
            for (int m = 0; m < WM; ++m) af[m] = *(const i32x4*)(win + aoff[m] + s * 32);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const i32x4 bf = *(const i32x4*)(wb + n * 32 * CR_RS + s * 32);
#pragma unroll
                for (int m = 0; m < WM; ++m) mma_step<T>(acc[m][n], af[m], bf);
            }
        }
    }

    // epilogue: lane holds column co = co0 + 32n + l31 of rows acc_row(r, h)
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int co = co0 + n * 32 + l31;
        const bool cok = co < a.Cout;
        const float bias = (cok && a.bias) ? a.bias[co] : 0.f;
        float e_sc = 0.f, e_sh = 0.f, e_mu = 0.f, e_rs = 0.f;
        if constexpr (EPI == EPI_RELU_BWD) {
            if (cok) { e_sc = a.ep_scale[co]; e_sh = a.ep_shift[co]; e_mu = a.ep_mean[co]; e_rs = a.ep_rstd[co]; }
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int m = 0; m < WM; ++m) {
            const int rbase = r0 + (wave * WM + m) * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + acc_row(r, h);
                if (row < r1 && cok) {
                    const size_t o = ((size_t)b * Rout + row) * a.Cout + co;
                    float val = acc[m][n][r] + bias;
                    if constexpr (EPI == EPI_RELU_BWD) {
                        const float y = load_elem<T>((const T*)a.aux, o);
                        val = (fmaf(y, e_sc, e_sh) > 0.f) ? val : 0.f;
                        s1 += val;
                        s2 += val * ((y - e_mu) * e_rs);
                    } else if constexpr (EPI == EPI_ADD_RELU) {
                        if (a.aux) val += load_elem<T>((const T*)a.aux, o);
                        val = fmaxf(val, 0.f);
                    } else {
                        s1 += val;
                        s2 += val * val;
                    }
                    store_elem<T>((T*)a.out, o, val);
                }
            }
        }
        if (a.stats) {
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (h == 0 && cok) {
                atomicAdd(&a.stats[co], (double)s1);
                atomicAdd(&a.stats[a.Cout + co], (double)s2);
            }
        }
    }
}

static int conv_rows_win_rows_max(int BM, int V, int Tin, int ntaps, int stride, int dil, int transposed) {
    const int nf = (BM - 2) / V + 2;                       // output frames one tile can touch
    int fr;
    if (!transposed) fr = stride * (nf - 1) + (ntaps - 1) * dil + 1;
    else fr = (nf - 1 + (ntaps - 1) * dil) / stride + 2;
    if (fr > Tin) fr = Tin;
    return fr * V;
}

template <typename T, int WM, int NT, int EPI>
static int launch_conv_rows(ConvRowsArgs& a, hipStream_t stream) {
    constexpr int BM = 128 * WM, BN = 32 * NT;
    a.win_rows_max = conv_rows_win_rows_max(BM, a.V, a.Tin, a.ntaps, a.stride, a.dil, a.transposed);
    const size_t lds = (size_t)(a.win_rows_max + 1) * CR_RS + 2 * (size_t)BN * CR_RS;
    if (lds > 160 * 1024) return STGCN_ERR_ARG;
    a.tiles_per_sample = ceil_div_i(a.Tout * a.V, BM);
    dim3 grid((unsigned)(a.B * a.tiles_per_sample), (unsigned)ceil_div_i(a.Cout, BN));
    auto kern = conv_rows_kernel<T, WM, NT, EPI>;
    static std::atomic<uint64_t> lds_ok{0};
    if (stgcn_allow_big_lds((const void*)kern, lds_ok) != STGCN_OK) return STGCN_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a);
    return hipGetLastError() == hipSuccess ? STGCN_OK : STGCN_ERR_LAUNCH;
}

template <typename T>
static int conv_rows_dispatch(ConvRowsArgs& a, int epi, hipStream_t stream) {
    constexpr int KC = 128 / ET<T>::ES;
    if (a.Cin % KC != 0 || a.Cin <= 0 || a.Cout <= 0 || a.B <= 0 || a.Tin <= 0 || a.Tout <= 0 || a.V <= 0 || a.V > 64) return STGCN_ERR_ARG;
    if (a.ntaps < 1 || a.stride < 1 || a.dil < 1 || a.pad < 0) return STGCN_ERR_ARG;
    if (epi == EPI_RELU_BWD) {
        if (!a.aux || !a.ep_scale || !a.ep_shift || !a.ep_mean || !a.ep_rstd) return STGCN_ERR_ARG;
        return launch_conv_rows<T, 1, 2, EPI_RELU_BWD>(a, stream);
    }
    if (epi == EPI_ADD_RELU) return launch_conv_rows<T, 1, 2, EPI_ADD_RELU>(a, stream);
    return launch_conv_rows<T, 1, 2, EPI_PLAIN>(a, stream);
}

// fast path for the stride-1 temporal convolutions in bf16 (conv_t9.hip); -1 = shape not covered
int conv_t9_try(const void* in, const void* w, void* out, const float* bias, const float* pro_scale, const float* pro_shift,
                int epilogue, const void* aux, const float* ep_scale, const float* ep_shift, const float* ep_mean, const float* ep_rstd,
                double* stats, int B, int T, int V, int Cin, int Cout, int ntaps, int transposed, hipStream_t stream, const BnPrologue* pbn);

// tiled path for the remaining shapes with Cin a whole number of 128-byte chunks (conv_big.hip); -1 = shape not covered
int conv_big_try(int dtype, const void* in, const void* w, void* out, const float* bias, const float* pro_scale, const float* pro_shift,
                 int epilogue, const void* aux, const float* ep_scale, const float* ep_shift, const float* ep_mean, const float* ep_rstd,
                 double* stats, int B, int Tin, int Tout, int V, int Cin, int Cout, int ntaps, int stride, int pad, int dil,
                 int transposed, hipStream_t stream, const BnPrologue* pbn);

// whole-width tiles for 128 / 256 output channels at an input stride of one frame, bf16 (conv_wide.hip); -1 = shape not covered
int conv_wide_try(const void* in, const void* w, void* out, const float* bias, const float* pro_scale, const float* pro_shift,
                  int epilogue, const void* aux, const float* ep_scale, const float* ep_shift, const float* ep_mean, const float* ep_rstd,
                  double* stats, int B, int Tin, int Tout, int V, int Cin, int Cout, int ntaps, int stride, int pad, int dil,
                  int transposed, hipStream_t stream, const BnPrologue* pbn);

// streaming path for the strided 1-tap convolution in bf16 (conv1s.hip); -1 = shape not covered
int conv1s_try(const void* in, const void* w, void* out, const float* bias, double* stats,
               int B, int Tin, int Tout, int V, int Cin, int Cout, int stride, int transposed, hipStream_t stream);

// C-ABI ------------------------------------------------------------------------------------
static int conv_rows_any(int dtype, const void* in, const void* w_packed, void* out,
                         const float* bias, const float* pro_scale, const float* pro_shift,
                         int epilogue, const void* aux, const float* ep_scale, const float* ep_shift,
                         const float* ep_mean, const float* ep_rstd, double* stats,
                         int B, int Tin, int Tout, int V, int Cin, int Cout,
                         int ntaps, int stride, int pad, int dil, int transposed, void* stream, const BnPrologue* pbn) {
    ConvRowsArgs a;
    a.pbn = pbn ? *pbn : BnPrologue{};
    a.in = in; a.w = w_packed; a.out = out; a.bias = bias; a.pro_scale = pro_scale; a.pro_shift = pro_shift;
    a.aux = aux; a.ep_scale = ep_scale; a.ep_shift = ep_shift; a.ep_mean = ep_mean; a.ep_rstd = ep_rstd; a.stats = stats;
    a.B = B; a.Tin = Tin; a.Tout = Tout; a.V = V; a.Cin = Cin; a.Cout = Cout;
    a.ntaps = ntaps; a.stride = stride; a.pad = pad; a.dil = dil; a.transposed = transposed;
    a.win_rows_max = 0; a.tiles_per_sample = 0;
    if (!in || !w_packed || !out) return STGCN_ERR_ARG;
    if ((pro_scale == nullptr) != (pro_shift == nullptr)) return STGCN_ERR_ARG;
    if (dtype == STGCN_BF16 && stride == 1 && dil == 1 && ntaps >= 1 && pad == (ntaps - 1) / 2 && Tin == Tout && B > 0 && Tin > 0 &&
        (epilogue == EPI_PLAIN || epilogue == EPI_ADD_RELU || (aux && ep_scale && ep_shift && ep_mean && ep_rstd))) {
        const int rc = conv_t9_try(in, w_packed, out, bias, pro_scale, pro_shift, epilogue, aux, ep_scale, ep_shift, ep_mean, ep_rstd,
                                   stats, B, Tin, V, Cin, Cout, ntaps, transposed, (hipStream_t)stream, pbn);
        if (rc >= 0) return rc;
    }
    if (dtype == STGCN_BF16 && ntaps == 1 && stride == 2 && pad == 0 && dil == 1 && epilogue == EPI_PLAIN && !pro_scale) {
        const int rc = conv1s_try(in, w_packed, out, bias, stats, B, Tin, Tout, V, Cin, Cout, stride, transposed, (hipStream_t)stream);
        if (rc >= 0) return rc;
    }
    if ((dtype == STGCN_BF16 || dtype == STGCN_F32) && B > 0 && Tin > 0 && Tout > 0 &&
        (epilogue == EPI_PLAIN || epilogue == EPI_ADD_RELU || (epilogue == EPI_RELU_BWD && aux && ep_scale && ep_shift && ep_mean && ep_rstd))) {
        if (dtype == STGCN_BF16) {
            const int rc = conv_wide_try(in, w_packed, out, bias, pro_scale, pro_shift, epilogue, aux, ep_scale, ep_shift, ep_mean, ep_rstd,
                                         stats, B, Tin, Tout, V, Cin, Cout, ntaps, stride, pad, dil, transposed, (hipStream_t)stream, pbn);
            if (rc >= 0) return rc;
        }
        const int rc = conv_big_try(dtype, in, w_packed, out, bias, pro_scale, pro_shift, epilogue, aux, ep_scale, ep_shift, ep_mean, ep_rstd,
                                    stats, B, Tin, Tout, V, Cin, Cout, ntaps, stride, pad, dil, transposed, (hipStream_t)stream, pbn);
        if (rc >= 0) return rc;
    }
    if (dtype == STGCN_BF16) return conv_rows_dispatch<bf16>(a, epilogue, (hipStream_t)stream);
    if (dtype == STGCN_F32) return conv_rows_dispatch<float>(a, epilogue, (hipStream_t)stream);
    return STGCN_ERR_ARG;
}

extern "C" int stgcn_conv_rows(int dtype, const void* in, const void* w_packed, void* out,
                               const float* bias, const float* pro_scale, const float* pro_shift,
                               int epilogue, const void* aux, const float* ep_scale, const float* ep_shift,
                               const float* ep_mean, const float* ep_rstd, double* stats,
                               int B, int Tin, int Tout, int V, int Cin, int Cout,
                               int ntaps, int stride, int pad, int dil, int transposed, void* stream) {
    return conv_rows_any(dtype, in, w_packed, out, bias, pro_scale, pro_shift, epilogue, aux, ep_scale, ep_shift, ep_mean, ep_rstd, stats,
                         B, Tin, Tout, V, Cin, Cout, ntaps, stride, pad, dil, transposed, stream, nullptr);
}

// The forward form with the BatchNorm + ReLU ahead of it finished inside the launch (train mode): the prologue's
// statistics arrive as the double sums [sum[Cin], sum of squares[Cin]] over pro_count rows that the producer of `in`
// gathered; pro_bn_out (float[4][Cin]) receives mean, rstd, scale, shift (the later kernels of the step read them), the
// running statistics (either pair may be null) advance once.  See BnPrologue.
extern "C" int stgcn_conv_rows_bn(int dtype, const void* in, const void* w_packed, void* out, const float* bias,
                                  const double* pro_stats, const float* pro_gamma, const float* pro_beta,
                                  float* pro_running_mean, float* pro_running_var, float* pro_bn_out,
                                  double pro_count, float momentum, float eps, double* stats,
                                  int B, int Tin, int Tout, int V, int Cin, int Cout,
                                  int ntaps, int stride, int pad, int dil, void* stream) {
    if (!pro_stats || !pro_gamma || !pro_beta || !pro_bn_out || pro_count <= 0 || Cin <= 0) return STGCN_ERR_ARG;
    if ((pro_running_mean == nullptr) != (pro_running_var == nullptr)) return STGCN_ERR_ARG;
    BnPrologue f{pro_stats, pro_gamma, pro_beta, pro_running_mean, pro_running_var, pro_bn_out, pro_count, momentum, eps, Cin};
    return conv_rows_any(dtype, in, w_packed, out, bias, pro_bn_out + 2 * (size_t)Cin, pro_bn_out + 3 * (size_t)Cin, EPI_PLAIN,
                         nullptr, nullptr, nullptr, nullptr, nullptr, stats,
                         B, Tin, Tout, V, Cin, Cout, ntaps, stride, pad, dil, 0, stream, &f);
}
