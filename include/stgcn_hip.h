/* C ABI of libstgcn_hip.so: hand-written gfx950 (MI355X) kernels for the ST-GCN hot path.
 *
 * The reference (hensyou/st-gcn) has no FFI: its hot path is reached as a Python module
 * (net.st_gcn.Model / net.utils.tgcn.ConvTemporalGraphical) whose arithmetic is delegated to
 * stock PyTorch operators.  Each entry point below replaces the operator call(s) cited with it.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller (the two host-side
 *     arguments of the deferred-sum calls, `sum_list` and `workspace_used`, are named as such where they appear);
 *   - activations are channels-last: physical [B][T][V][C], C contiguous (a torch tensor of
 *     logical shape [B,C,T,V] in torch.channels_last memory format); dtype STGCN_F32 or
 *     STGCN_BF16; per-channel parameters and all gradients of parameters are float32;
 *   - `stream` is a hipStream_t; calls only enqueue work: no allocation, no synchronisation;
 *   - return value: 0 ok, 1 invalid argument, 2 launch failed, 3 workspace too small, 4 problem exceeds the
 *     library's static scratch array (stgcn_gcn_agrad_noalloc only), 5 sum list full (the _deferred calls only);
 *   - `stats` buffers are double[2*C]; kernels ADD into them, the caller zeroes them first;
 *   - channel counts of MFMA operands must be multiples of one 32-byte k-step (16 bf16 / 8 f32);
 *     conv_rows / conv_wgrad input channels multiples of one 128-byte chunk (64 bf16 / 32 f32).
 */
#ifndef STGCN_HIP_H
#define STGCN_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STGCN_F32 0
#define STGCN_BF16 1
#define STGCN_EPI_PLAIN 0    /* out = acc + bias;                   stats += [sum out, sum out^2]        */
#define STGCN_EPI_RELU_BWD 1 /* out = (ep_scale*aux+ep_shift>0)?acc:0; stats += [sum out, sum out*xhat(aux)] */
#define STGCN_EPI_ADD_RELU 2 /* out = max(acc + bias + aux, 0), aux optional, no statistics: the whole output stage of
                              * an st_gcn block in eval mode (net/st_gcn.py:169, 193-197) once BatchNorm is folded
                              * into the conv weights and bias; aux = the residual branch */

/* Convolution along T as an implicit GEMM (MFMA):
 *   out[b,to,v,co] = bias[co] + sum_tap sum_ci pro(in[b, ti, v, ci]) * w[tap][co][ci]
 *   forward form ti = stride*to - pad + tap*dil; transposed form ti = (to + pad - tap*dil)/stride when divisible;
 *   pro(x) = relu(pro_scale*x + pro_shift) when given (BatchNorm + ReLU ahead of the conv).
 * Replaces: nn.Conv2d(C,C,(9,1),(s,1),(4,0)) forward of st_gcn.tcn (net/st_gcn.py:162-168) fused with the
 * preceding BatchNorm2d+ReLU (st_gcn.py:160-161) and the statistics pass of the following BatchNorm2d (st_gcn.py:169);
 * the residual nn.Conv2d(Cin,Cout,1,(s,1)) (st_gcn.py:181-185); Model.extract_feature's per-position fcn
 * (st_gcn.py:113); and, in the transposed form, their input gradients (autograd of the same lines), with
 * STGCN_EPI_RELU_BWD fusing the ReLU mask and the reduction pass of the BatchNorm backward. */
int stgcn_conv_rows(int dtype, const void* in, const void* w_packed, void* out,
                    const float* bias, const float* pro_scale, const float* pro_shift,
                    int epilogue, const void* aux, const float* ep_scale, const float* ep_shift,
                    const float* ep_mean, const float* ep_rstd, double* stats,
                    int B, int Tin, int Tout, int V, int Cin, int Cout,
                    int ntaps, int stride, int pad, int dil, int transposed, void* stream);

/* The forward form of stgcn_conv_rows with the BatchNorm + ReLU ahead of it finished inside the launch (train mode):
 * instead of pro_scale / pro_shift the call takes the double sums [sum[Cin], sum of squares[Cin]] over pro_count rows
 * that the producer of `in` gathered, and the BatchNorm's parameters.  Every workgroup derives the coefficients of all
 * Cin channels with the arithmetic of stgcn_bn_finalize (same floats) and writes them to pro_bn_out (float[4][Cin]:
 * mean, rstd, scale, shift -- the same values from every workgroup; the weight gradient, the data gradient and
 * stgcn_bn_bwd_apply of the step read them from there); the running statistics (either pair may be null) advance once.
 * One launch instead of two.
 * Replaces: the same operators as stgcn_conv_rows plus the statistics half of nn.BatchNorm2d (net/st_gcn.py:160). */
int stgcn_conv_rows_bn(int dtype, const void* in, const void* w_packed, void* out, const float* bias,
                       const double* pro_stats, const float* pro_gamma, const float* pro_beta,
                       float* pro_running_mean, float* pro_running_var, float* pro_bn_out,
                       double pro_count, float momentum, float eps, double* stats,
                       int B, int Tin, int Tout, int V, int Cin, int Cout,
                       int ntaps, int stride, int pad, int dil, void* stream);

/* Weight gradient of the forward form above: dW[co][ci][tap] (float32, the torch conv weight layout).
 * Replaces: autograd weight gradient of the convolutions at net/st_gcn.py:162-168 and 181-185.
 * workspace: partial-sum slab, at least ntaps*Cout*Cin*4 bytes (more lets more workgroups run). */
int stgcn_conv_wgrad(int dtype, const void* in, const void* dout, float* dW,
                     const float* pro_scale, const float* pro_shift,
                     int B, int Tin, int Tout, int V, int Cin, int Cout,
                     int ntaps, int stride, int pad, int dil,
                     void* workspace, size_t workspace_bytes, void* stream);

/* Fused graph convolution over NF = B*T frames:
 *   out[f,a,n] = sum_k sum_b adj[k][a][b] * (sum_c in[f,b,c]*w[k][n][c] + bias[k][n])  (+ add[f,a,n])
 * add_mask (optional, with add): 16 bits per 16-byte piece of add, as written by stgcn_block_out_fwd; elements whose
 * bit j is clear count as zero -- the gradient of an identity residual is then the block's incoming gradient itself,
 * masked on the fly, and is never written out.
 * adj_transposed != 0 reads adj[k][b][a] in place of adj[k][a][b], so both directions take A*importance as stored:
 * Forward:  adj = A*importance, adj_transposed = 1, w[k][co][ci] = conv.weight[k*Cout+co][ci].
 * Backward: adj = A*importance, adj_transposed = 0, w[k][ci][co] = conv.weight[k*Cout+co][ci], no bias.
 * Replaces: ConvTemporalGraphical.forward -- nn.Conv2d 1x1 (net/utils/tgcn.py:60), view (tgcn.py:62-63) and
 * torch.einsum('nkctv,kvw->nctw') (tgcn.py:64) -- and its input gradient; the K*C intermediate stays in registers. */
int stgcn_gcn_fused(int dtype, const void* in, const void* w_packed, const float* adj, int adj_transposed, const float* bias,
                    void* out, const void* add, const void* add_mask, double* stats,
                    int NF, int V, int Cin, int Nout, int K, void* stream);

/* dA[k][v][w] = sum_{f,co} (sum_ci w[k][co][ci] x[f,v,ci] + bias[k][co]) * dy[f,w,co]   (K <= 3 per call)
 * Replaces: autograd gradient of the einsum w.r.t. A (tgcn.py:64), i.e. of A*importance (net/st_gcn.py:82). */
int stgcn_gcn_agrad(int dtype, const void* x, const void* w_packed, const float* bias, const void* dy,
                    float* dA, int NF, int V, int Cin, int Cout, int K,
                    void* workspace, size_t workspace_bytes, void* stream);

/* stgcn_gcn_agrad without a workspace argument and without any allocation.  No counterpart in the reference: it is
 * for callers that cannot allocate device memory -- code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  The partial sums go to a static device array inside the library (4 MiB per GPU, held until the
 * process ends) instead of a caller-provided workspace; same kernels, same result bit for bit.  Limit: (resident
 * workgroups, at most 2 per compute unit) * K * V * V * 4 bytes <= 4 MiB -- every ST-GCN graph (V <= 25, K <= 3) on a
 * 256-CU part; larger problems return status 4.  All calls on one GPU share that array: calls that overlap on one GPU
 * race, calls ordered on one stream do not.  The first call on each device looks the array's address up and must not
 * run under stream capture. */
int stgcn_gcn_agrad_noalloc(int dtype, const void* x, const void* w_packed, const float* bias, const void* dy,
                            float* dA, int NF, int V, int Cin, int Cout, int K, void* stream);

/* dW[k][co][ci] = sum_{f,v} (sum_w adj[k][v][w] dy[f,w,co]) x[f,v,ci];  db[k][co] = sum of the same bracket.
 * adj = (A*importance)[k][v][w].  K <= 3 per call.
 * Replaces: autograd gradients of conv.weight / conv.bias of ConvTemporalGraphical (tgcn.py:48-55, 60). */
int stgcn_gcn_wgrad(int dtype, const void* x, const void* dy, const float* adj, float* dW, float* db,
                    int NF, int V, int Cin, int Cout, int K,
                    void* workspace, size_t workspace_bytes, void* stream);

/* Deferred sums.  The three parameter-gradient entry points above run two stages: kernels that write per-workgroup
 * partial sums into the workspace, and one small launch that adds the partials in a fixed order.  The _deferred forms
 * run the first stage only and append the second to a SUM LIST; stgcn_sum_list_run then adds the partials of every
 * entry on the list in ONE launch (a block's temporal-convolution, graph-convolution and adjacency gradients: one
 * launch instead of three or four).  Results are bit-identical to the undeferred calls given the same workspace_bytes.
 *   sum_list        caller-owned HOST memory of STGCN_SUM_LIST_BYTES bytes, zero-filled before its first use; holds at
 *                   most 8 sums (a call adds one or two).  A full list makes the call return status 5, nothing launched.
 *   workspace       as above, but it must not overlap the part of any workspace that an entry still on the list uses:
 *                   *workspace_used receives the bytes, from `workspace` on, that have to stay untouched until
 *                   stgcn_sum_list_run has been ISSUED on the same stream (stream order does the rest).
 * Replaces: the same autograd gradients as stgcn_conv_wgrad / stgcn_gcn_agrad / stgcn_gcn_wgrad. */
#define STGCN_SUM_LIST_BYTES 1024
int stgcn_conv_wgrad_deferred(int dtype, const void* in, const void* dout, float* dW,
                              const float* pro_scale, const float* pro_shift,
                              int B, int Tin, int Tout, int V, int Cin, int Cout,
                              int ntaps, int stride, int pad, int dil,
                              void* workspace, size_t workspace_bytes, void* sum_list, size_t* workspace_used, void* stream);
int stgcn_gcn_agrad_deferred(int dtype, const void* x, const void* w_packed, const float* bias, const void* dy,
                             float* dA, int NF, int V, int Cin, int Cout, int K,
                             void* workspace, size_t workspace_bytes, void* sum_list, size_t* workspace_used, void* stream);
int stgcn_gcn_wgrad_deferred(int dtype, const void* x, const void* dy, const float* adj, float* dW, float* db,
                             int NF, int V, int Cin, int Cout, int K,
                             void* workspace, size_t workspace_bytes, void* sum_list, size_t* workspace_used, void* stream);
/* Appends dst[i] = (float)src[i], i < n, to the list: finished double-precision sums (the bias gradients that
 * stgcn_block_out_bwd accumulates) become float32 gradients in the list's launch instead of one of their own.  Host
 * work only; `src` must hold its final values by the time stgcn_sum_list_run is issued.  Status 5 when the list is full. */
int stgcn_sum_list_add_cast(void* sum_list, const double* src, float* dst, size_t n);
/* Launches the sums waiting on the list (none: no launch) and empties it. */
int stgcn_sum_list_run(void* sum_list, void* stream);

/* BatchNorm bookkeeping: batch statistics -> mean / rstd / scale=gamma*rstd / shift=beta-mean*scale, running
 * statistics update (momentum, unbiased variance); eval mode reads the running statistics instead.
 * Replaces: the statistics half of nn.BatchNorm2d / nn.BatchNorm1d (net/st_gcn.py:42,160,169,186). */
int stgcn_bn_finalize(const double* stats, double count, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, float momentum, float eps,
                      float* mean, float* rstd, float* scale, float* shift, int C, int training, void* stream);

/* dbeta = stats[0..C), dgamma = stats[C..2C) as float32. */
int stgcn_stats_to_grads(const double* stats, float* dgamma, float* dbeta, int C, void* stream);

/* out = relu((scale2*o + shift2) * keep/(1-p) + res'), res' = res_scale*res + res_shift | res | 0.
 * relu_mask (optional, 2 bytes per 16-byte piece of out: rows*C/4 bytes for bf16, rows*C/2 for f32): bit j set when
 * element j of the piece is positive, bit 8+j when dropout kept it -- the backward then reads 1/8 of the bytes of
 * out and evaluates no dropout hash.
 * seed_step (optional, device memory): a step counter the caller advances on the device; the launch then draws with
 * seed + (*seed_step) * 0x85EBCA6B (mod 2^32; only the low 32 bits of the counter are read, so it may be an int64).
 * A step replayed from a captured HIP graph thus draws a fresh mask every replay although `seed` is frozen in it.
 * Replaces: BatchNorm2d apply + Dropout + residual add + ReLU (net/st_gcn.py:169-170, 193-197). */
int stgcn_block_out_fwd(int dtype, const void* o, const void* res, void* out, void* relu_mask,
                        const float* scale2, const float* shift2, const float* res_scale, const float* res_shift,
                        size_t rows, int C, float drop_p, uint32_t seed, const uint32_t* seed_step, void* stream);

/* stgcn_block_out_fwd for train mode with the two BatchNorms finished inside the launch: bn2 (of o) and, when
 * bnr_stats is given, the residual branch's BatchNorm (of res) arrive as the double sums [sum[C], sum of squares[C]] that
 * the producing convolutions gathered over `rows` rows.  Every workgroup derives scale / shift from the sums with the
 * arithmetic of stgcn_bn_finalize (same floats); workgroup 0 also writes mean, rstd, scale, shift to bn2_out / bnr_out
 * (float[4][C], read by stgcn_block_out_bwd) and advances the running statistics (momentum, unbiased variance; either
 * pair may be null).  One launch instead of three.  C <= 512 (the coefficient table lives in LDS); wider: status 1.
 * Replaces: the same operators as stgcn_block_out_fwd plus the statistics half of the two nn.BatchNorm2d
 * (net/st_gcn.py:169, 186). */
int stgcn_block_out_fwd_bn(int dtype, const void* o, const void* res, void* out, void* relu_mask,
                           const double* bn2_stats, const float* gamma2, const float* beta2,
                           float* running_mean2, float* running_var2, float* bn2_out,
                           const double* bnr_stats, const float* gammar, const float* betar,
                           float* running_meanr, float* running_varr, float* bnr_out,
                           float momentum, float eps,
                           size_t rows, int C, float drop_p, uint32_t seed, const uint32_t* seed_step, void* stream);

/* Backward of the stage above.  pass 0: per-channel sums (stats2, statsr).  pass 1: dO (BatchNorm backward of the
 * main branch), dR (same for a conv residual), g (= masked incoming gradient, for an identity residual), the
 * bias gradients of the two convolutions (dbias_*: double[C] sums accumulated with atomics, zeroed by the caller) and,
 * when asked for, the BatchNorm parameter gradients read off the sums of pass 0 (dbeta = sum, dgamma = sum*xhat).  The incoming gradient is dout and/or
 * dpool (float32 [B][C], constant over (t,v): every row of sample b receives dpool[b][c] * dpool_scale -- the gradient of the
 * global average pool when dpool is the gradient of the pooled mean and dpool_scale = 1 / rows_per_sample).  The ReLU and dropout decisions come from relu_mask when
 * given (then out may be null), else from out and the dropout hash.
 * Replaces: autograd of net/st_gcn.py:169-170, 193-197 (and of F.avg_pool2d, st_gcn.py:85, for dpool). */
int stgcn_block_out_bwd(int dtype, int pass, const void* dout, const float* dpool, size_t rows_per_sample, float dpool_scale,
                        const void* out, const void* relu_mask, const void* o, const void* r,
                        const float* mean2, const float* rstd2, const float* gamma2,
                        const float* meanr, const float* rstdr, const float* gammar,
                        double* stats2, double* statsr, void* dO, void* dR, void* g,
                        double* dbias_o, double* dbias_r,
                        float* dgamma2, float* dbeta2, float* dgammar, float* dbetar,
                        size_t rows, int C, float drop_p, uint32_t seed, double count, int training, void* stream);

/* dy = gamma*rstd*(g - S1/n - yhat*S2/n)  (eval: g*gamma*rstd); optionally also dbeta = S1, dgamma = S2 as float32.
 * Replaces: autograd of BatchNorm2d (st_gcn.py:160). */
int stgcn_bn_bwd_apply(int dtype, const void* g, const void* y, void* dy, const float* gamma,
                       const float* mean, const float* rstd, const double* stats, double count,
                       float* dgamma, float* dbeta, size_t rows, int C, int training, void* stream);

/* Batch input augmentation, x and out [N][C][T][V*M] float32: out frame t of clip n shows input frame t + shift[n]
 * (zeros outside [0, Tin)); then, when ca is given, channels 0/1 of every frame f = n*Tout + t become
 * x' = ca[f]*x - sa[f]*y + tx[f], y' = sa[f]*x + ca[f]*y + ty[f]  (ca = s*cos a, sa = s*sin a).
 * Replaces: feeder/tools.py random_choose / auto_pading (tools.py:31-54) and random_move's transformation
 * (tools.py:96-107) applied per clip on the host; the random draws stay on the host (feeder.tools.choose_shift,
 * move_params). */
int stgcn_augment(const float* in, float* out, const int* shift, const float* ca, const float* sa,
                  const float* tx, const float* ty, int N, int C, int Tin, int Tout, int VM, void* stream);

/* data_bn: BatchNorm1d over channel v*C+c of x[N,C,T,V,M] (float32) and the two permutes around it.
 * Replaces: net/st_gcn.py:72-78 (output is the channels-last [N*M][T][V][Cp] activation, Cp >= C zero padded, Cp a
 * multiple of one 16-byte piece: 8 bf16 / 4 f32 channels). */
int stgcn_data_bn_stats(const float* x, double* stats, int N, int C, int T, int V, int M, void* stream);
int stgcn_data_bn_apply(int dtype, const float* x, void* out, const float* scale, const float* shift,
                        int N, int C, int T, int V, int M, int Cp, void* stream);
int stgcn_data_bn_bwd(int dtype, const void* dx0, const float* x, const float* mean, const float* rstd, double* stats,
                      int N, int C, int T, int V, int M, int Cp, void* stream);

/* Head.  Replaces: F.avg_pool2d over (T,V) (st_gcn.py:85), mean over M (st_gcn.py:86), fcn 1x1 conv (st_gcn.py:89-90). */
int stgcn_pool_fwd(int dtype, const void* feat, float* pooled, int B, int R, int C, void* stream);
int stgcn_pool_bwd(int dtype, const float* dpm, void* dfeat, int B, int R, int C, int M, void* stream);
int stgcn_head_fc_fwd(const float* pooled, const float* W, const float* bias, float* logits,
                      int N, int M, int C, int ncls, void* stream);
/* dpm [N][C]: gradient of the per-clip mean over persons; dpooled [N*M][C] = dpm / M per person (either may be null). */
int stgcn_head_fc_bwd(const float* dlogits, const float* pooled, const float* W, float* dW, float* db, float* dpm, float* dpooled,
                      int N, int M, int C, int ncls, void* stream);

/* out[i][e] = a[e] * inputs[i][e], e < len, for n <= 16 inputs in one launch (a null input gives zeros).  `inputs` is a
 * HOST array of n device pointers (copied into the launch).  Forward: a = A, inputs = the blocks' edge_importance
 * parameters; backward: inputs = the blocks' adjacency gradients.
 * Replaces: self.A * importance of every block (net/st_gcn.py:81-82) and its autograd. */
int stgcn_mul_many(const float* a, const float* const* inputs, int n, float* out, size_t len, void* stream);

/* loss[0] = mean over n of -log softmax(logits[n])[labels[n]] and dlogits = (softmax - onehot) / N, one launch.
 * labels: int64, every one in [0, ncls) (no ignore_index); N <= 8192.
 * Replaces: nn.CrossEntropyLoss() of the training step (processor/recognition.py:44, 91) and its autograd. */
int stgcn_cross_entropy(const float* logits, const int64_t* labels, float* loss, float* dlogits, int N, int ncls, void* stream);

/* Re-packs every convolution weight of a model in one launch.  table: ndesc descriptors in device memory, each
 * 10 x 8 bytes { const float* src; void* w; void* wt; int64 G, Cout, Cin, Cinp, sg, sco, sci }: element (g, co, ci) =
 * src[g*sg + co*sco + ci*sci] is written in the compute dtype to w[g][co][cip] and wt[g][cip][co] (cip < Cinp,
 * zero past Cin; either destination may be null).  max_elems = the largest G*Cout*Cinp.
 * Replaces: the weight reads inside nn.Conv2d at net/utils/tgcn.py:60 and net/st_gcn.py:162-168, 181-185 (the
 * kernels above want the weights tap- / partition-major in the activation dtype). */
int stgcn_pack_weights(int dtype, const void* table, int ndesc, size_t max_elems, void* stream);

/* SGD with momentum / Nesterov / weight decay over flat float32 buffers; g is multiplied by grad_scale first.
 * Replaces: optim.SGD(...).step() as configured at processor/recognition.py:46-53. */
int stgcn_sgd_step(float* params, const float* grads, float* momentum_buf, size_t n,
                   float lr, float momentum, float weight_decay, int nesterov, int first_step, float grad_scale,
                   void* stream);

/* Diagnostics. */
int stgcn_probe_tr16(short* out, void* stream);
int stgcn_version(void);
const char* stgcn_arch(void);
int stgcn_device_count(void);
/* What the launchers have learnt / set for device ordinal `device` so far (both kept per device: a process may drive
 * several GPUs, as the reference's nn.DataParallel does, processor/io.py:86-87): its compute-unit count (0 = not
 * asked yet) and how many kernels have opted in to more than 64 KiB of LDS there. */
int stgcn_launch_state(int device, int* num_cus, int* lds_opt_ins);

#ifdef __cplusplus
}
#endif
#endif
