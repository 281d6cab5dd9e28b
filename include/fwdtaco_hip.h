/* fwdtaco_hip.h -- C ABI of libfwdtaco_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for the ForwardTacotron mel-generation hot path of ziyaad30/ForwardTacotron.
 * The reference has no FFI layer of its own (its "kernels" are stock torch.nn modules), so each entry
 * point below names the reference op it replaces (file:line relative to the reference root).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory unless stated; all float tensors are
 *     contiguous fp32, activations are CHANNELS-LAST  [B, T, C]  (row = one (b,t) position, `ld*` = row
 *     stride in floats) -- the reference's [B,C,T] transposes (forward_tacotron.py:32,36,134,155,...) vanish;
 *   - `stream` is a hipStream_t passed as void*; calls only enqueue work (no sync, no allocation) unless
 *     stated, so they are capturable in a hipGraph -- except the recurrences (ft_gru_*, ft_lstm_*): their persistent
 *     form keeps host-side admission bookkeeping (event records / queries per launch), capture those with
 *     ft_rnn_set_persistent(0) or leave them outside the graph (tests/test_gpu_primitives.py captures a token-side chain);
 *   - return 0 on success; non-zero = error, message via ft_last_error() (thread-local);
 *   - workspaces are caller-allocated device memory, sized by the matching *_workspace() query (bytes).
 */
#ifndef FWDTACO_HIP_H
#define FWDTACO_HIP_H

#include <stddef.h>
#include <stdint.h>

#define FWDTACO_ABI_VERSION 4

#ifdef __cplusplus
extern "C" {
#endif

const char* ft_last_error(void);
int ft_abi_version(void);
/* host query: CU count and whether device 0 is gfx950 */
int ft_device_info(int* cu_count, int* is_gfx950);

/* Matmul precision of every GEMM-shaped entry point below (process-wide; returns the previous setting):
 * 0 (default) = fp32-exact (f32 MFMA, or exact three-way bf16 splits with fp32 accumulation);
 * 1 = bf16: operands rounded to nearest bf16 on load, ONE bf16 MFMA per product, fp32 accumulation, fp32 outputs --
 *     BASELINE configs[2] (FastPitch "bf16"); applies to the NT-form aligned launches (Linear / Conv1d forward, their
 *     data gradients, every weight gradient, attention Q K^T); the rest (NN-form attention products, odd shapes) and all
 *     non-GEMM kernels (LayerNorm, softmax statistics, losses, optimizer, recurrences) stay fp32. */
int ft_set_gemm_precision(int bf16);
/* weight-gradient launches that took the software-pipelined 128x128 kernel since the library loaded (tests use it to
 * prove which kernel they exercised; the choice never changes results: both kernels give the same bits) */
int ft_gemm_tn_pipelined_launches(void);

/* ---- nn.Linear (models/forward_tacotron.py:25,100,108 ; common_layers.py:31-32,83) ------------------ */
/* Row layouts: the rows of an activation matrix are the (b,t) positions in batch-major order (row = b*T+t,
 * i.e. a contiguous [B,T,C] tensor) unless a `*_tm_B` argument is > 0, in which case that operand is stored
 * TIME-major ([T,B,C], row = t*B+b, B = the argument; rows % B == 0).  The recurrences keep their buffers
 * time-major (each timestep's slab contiguous); the GEMMs read / write either order for free.
 * y[rows,out_f] (+)= x[rows,in_f] * w[out_f,in_f]^T + bias ; optional relu */
int ft_linear_fwd(const float* x, long ldx, const float* w, const float* bias, float* y, long ldy, int rows,
                  int in_f, int out_f, int relu, int accumulate, int x_tm_B, int y_tm_B, void* stream);
/* several Linear layers sharing one input, written side by side into y (highway W1|W2, RNN fwd|rev W_ih):
 * y[:, col_offset[i] : col_offset[i]+out_f[i]] = x * w[i]^T + bias[i]   (host arrays of device pointers) */
int ft_linear_multi_fwd(const float* x, long ldx, int ntasks, const float* const* w, const float* const* bias,
                        float* y, long ldy, const int* col_offset, const int* out_f, int rows, int in_f, int relu,
                        int x_tm_B, int y_tm_B, void* stream);
/* The same product, computed by the kernel that a launch over `as_rows` rows of the same layers would take
 * (the launcher picks 128x128 bf16-split or 64x64 f32 tiles by size, and the two round differently): the LengthRegulator
 * repeats rows, so a projection of its RESULT can be formed on its input with the bits the frame-level launch would give
 * every row (ops.LRBiLSTMFn). */
int ft_linear_multi_fwd_as(const float* x, long ldx, int ntasks, const float* const* w, const float* const* bias,
                           float* y, long ldy, const int* col_offset, const int* out_f, int rows, int in_f, long as_rows,
                           void* stream);
/* dx[rows,in_f] (+)= dy[rows,out_f] * w[out_f,in_f] ; w_transposed = 1: `w` points at w^T [in_f,out_f] instead
 * (both operands then have the contraction index contiguous -- the form the bf16-split MFMA kernel takes) */
int ft_linear_bwd_data(const float* dy, long lddy, const float* w, float* dx, long lddx, int rows, int in_f,
                       int out_f, int accumulate, int dy_tm_B, int dx_tm_B, int w_transposed, void* stream);
/* dw[out_f,in_f] (+)= dy^T * shift(x): rows = B*T logical positions; x_shift != 0 reads x row (b, t+x_shift),
 * zero outside [0,T) (recurrent-weight gradients: h_{t-1} / h_{t+1}); dy_time_major / x_time_major = 1 when
 * that operand is stored [T,B,*].  Deterministic split + ordered reduce. */
size_t ft_linear_bwd_weight_workspace(int rows, int in_f, int out_f);
int ft_linear_bwd_weight(const float* dy, long lddy, const float* x, long ldx, float* dw, int rows, int in_f,
                         int out_f, int B, int T, int x_shift, int accumulate, int dy_time_major, int x_time_major,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ---- nn.Conv1d(stride 1, padding k//2, bias=False) of BatchNormConv (common_layers.py:50,55-56) ------ */
/* weights are consumed TAP-MAJOR: wp[k][Cout][Cin] (ft_conv_pack_weight from torch's [Cout][Cin][k]).
 * y[b,t',co] = relu?( sum_j sum_ci x[b, t'+j-k/2, ci] * w[co,ci,j] ) (* scale[co] + shift[co] if scale) for
 * t' in [0,Tout); Tout = T (odd k, or even k sliced as the CBHG does, common_layers.py:99) or T+1 (even k);
 * accumulate=1 adds the result onto y (residual connection, common_layers.py:114). */
int ft_conv_pack_weight(const float* w, float* wp, int Cout, int Cin, int k, void* stream);
/* transposed tap-major pack wpt[k][Cin][Cout] for the data gradients (`wp_transposed` = 1 below): the contraction
 * index Cout becomes contiguous, which lets the gradient GEMM run in the NT form (see `w_transposed`) */
int ft_conv_pack_weight_t(const float* w, float* wpt, int Cout, int Cin, int k, void* stream);
/* All of a model's operand re-layouts in ONE launch (weights change once per optimizer step, so a training step
 * refreshes every pack / transpose up front instead of one small launch per layer).  Entry e re-lays
 * src [d0][d1][k] (a Conv1d weight [Cout][Cin][k]; a Linear / RNN weight [out_f][in_f] has k = 1) as
 * dst [k][d0][d1] (ft_conv_pack_weight) and / or dst_t [k][d1][d0] (ft_conv_pack_weight_t; the transpose of a
 * matrix); either may be NULL.  tile_begin = sum over earlier entries of k*ceil(d0/32)*ceil(d1/32), ascending;
 * `descs` is a DEVICE array of n entries, total_tiles the sum over all entries. */
typedef struct FtPackDesc {
  const float* src;
  float* dst;
  float* dst_t;
  long tile_begin;
  int d0, d1, k, reserved;
} FtPackDesc;
int ft_pack_weights(const FtPackDesc* descs, int n, long total_tiles, void* stream);
int ft_conv1d_fwd(const float* x, long ldx, const float* wp, const float* scale, const float* shift, float* y,
                  long ldy, int B, int T, int Cin, int Cout, int k, int Tout, int relu, int accumulate,
                  void* stream);
/* CBHG conv1d_bank (common_layers.py:72-76,97-102): members k=1..K (K<=16) in ONE launch; member i writes
 * ybank[:, :, i*C:(i+1)*C] of ybank[B,Tout,K*C]; wp_all = packed member weights back to back. */
int ft_conv_bank_fwd(const float* x, long ldx, const float* wp_all, const float* scale, const float* shift,
                     float* ybank, int B, int T, int Cin, int C, int K, int Tout, int relu, void* stream);
/* dx[b,t,ci] (+)= sum_j sum_co dy[b, t-j+k/2, co] * w[co,ci,j]; dy is [B,Tbuf,*] of which rows < Tvalid count */
int ft_conv1d_bwd_data(const float* dy, long lddy, const float* wp, float* dx, long lddx, int B, int T, int Cin,
                       int Cout, int k, int Tbuf, int Tvalid, int accumulate, int wp_transposed, void* stream);
/* the same through a ReLU's derivative: dx[r,c] = (that product) if y[r*lddx + c] > 0 else 0, y = the OUTPUT of the conv + ReLU
 * whose result this convolution consumed (FFTBlock: conv2's data gradient is conv1's pre-ReLU gradient,
 * common_layers.py:178-180) -- the mask is the GEMM's epilogue, not an element-wise pass.  T = Tbuf = Tvalid. */
int ft_conv1d_bwd_data_relu(const float* dy, long lddy, const float* wp, const float* y, float* dx, long lddx, int B,
                            int T, int Cin, int Cout, int k, int wp_transposed, void* stream);
/* dx (+)= sum_i dy_i[rows,out_f] * w_i[out_f,in_f]: the data gradients of several nn.Linear that read the same input
 * (HighwayNetwork W1/W2, common_layers.py:35-40; the two directions' W_ih of nn.GRU / nn.LSTM) in ONE launch,
 * accumulated in registers.  dy / w: host arrays of ntasks device pointers (ntasks <= 16). */
int ft_linear_bwd_data_multi(int ntasks, const float* const* dy, long lddy, const float* const* w, float* dx, long lddx,
                             int rows, int in_f, int out_f, int accumulate, int dy_tm_B, int dx_tm_B, int w_transposed,
                             void* stream);
/* The same product with scratch for split-K (ft_linear_bwd_data_multi_workspace bytes; 0 = it would not be split): few
 * output tiles -- token-side rows -- and a long contraction leave most of the chip idle in one pass over K, so the
 * contraction is cut into up to 16 ranges whose partial tiles a second launch adds in range order.  Same products, another
 * summation order than the unsplit launch. */
size_t ft_linear_bwd_data_multi_workspace(int ntasks, int rows, int in_f, int out_f, void* stream);
int ft_linear_bwd_data_multi_ws(int ntasks, const float* const* dy, long lddy, const float* const* w, float* dx, long lddx,
                                int rows, int in_f, int out_f, int accumulate, int dy_tm_B, int dx_tm_B, int w_transposed,
                                void* workspace, size_t workspace_bytes, void* stream);
/* data gradient of the whole conv bank (backward of common_layers.py:97-102) in ONE GEMM launch; dy = [B,Tbuf,K*C]
 * gradient of the bank buffer (Tbuf = T or T+1).  Long sequences: the K members' products are accumulated in registers
 * straight into dx[B,T,Cin].  Short ones (too few output tiles to fill the chip, e.g. the prenet's B*T = 4096 rows):
 * every member writes its own partial into `workspace` (size from the query, may be 0) and an ordered sum follows. */
size_t ft_conv_bank_bwd_data_workspace(int B, int T, int Cin, int K);
int ft_conv_bank_bwd_data(const float* dy, long lddy, const float* wp_all, float* dx, long lddx, int B, int T, int Cin,
                          int C, int K, int Tbuf, int wp_transposed, void* workspace, size_t workspace_bytes,
                          void* stream);
/* weight gradients of ALL K members of a conv bank in one GEMM launch + one ordered reduction: dy [B,Tbuf,K*C] =
 * gradient of the bank buffer, x [B,T,Cin] = the bank's input, dw = HOST array of K device pointers, dw[i] =
 * torch-layout [C][Cin][i+1] gradient of member i (kernel size i+1).  C % 128 == 0. */
size_t ft_conv_bank_bwd_weight_workspace(int B, int T, int Cin, int C, int K, int Tbuf);
int ft_conv_bank_bwd_weight(const float* dy, long lddy, const float* x, long ldx, float* const* dw, int B, int T,
                            int Cin, int C, int K, int Tbuf, void* workspace, size_t workspace_bytes, void* stream);
/* dw[co,ci,j] = sum_{b,t'<Tvalid} dy[b,t',co] * x[b,t'+j-k/2,ci]   (torch layout [Cout][Cin][k]) */
size_t ft_conv1d_bwd_weight_workspace(int B, int T, int Cin, int Cout, int k, int Tvalid);
int ft_conv1d_bwd_weight(const float* dy, long lddy, const float* x, long ldx, float* dw, int B, int T, int Cin,
                         int Cout, int k, int Tbuf, int Tvalid, void* workspace, size_t workspace_bytes,
                         void* stream);

/* ---- FastPitch building blocks (models/fast_pitch.py, common_layers.py:127-223) -------------------------- */
/* nn.Conv1d WITH bias, odd k, padding k//2 (FFTBlock conv1 k=9 + ReLU, conv2 k=1; common_layers.py:161-164,178-180) */
int ft_conv1d_bias_fwd(const float* x, long ldx, const float* wp, const float* bias, float* y, long ldy, int B, int T,
                       int Cin, int Cout, int k, int relu, void* stream);
/* strided-batch GEMMs for nn.MultiheadAttention (common_layers.py:158,172-174): nb0*nb1 independent instances,
 * instance z -> (z0,z1) = (z/nb1, z%nb1), operand X_z = X + z0*sX0 + z1*sX1 (floats).
 *   nt: C_z[M,N] = A_z[M,K] * B_z[N,K]^T   (scores = Q K^T ; dP = dCtx V^T)
 *   nn: C_z[M,N] = A_z[M,K] * B_z[K,N]     (ctx = P V ; dQ = dS K)
 *   tn: C_z[M,N] = A_z[R,M]^T * B_z[R,N]   (dV = P^T dCtx ; dK = dS^T Q), deterministic split + reduce
 * Row padding: attention's [T,T] matrices have a row length (841 frames) that is not a multiple of 4.  The caller keeps
 * them with a row stride rounded up to 4 and says so -- nn: a_rows_padded = 1 (every row of A readable up to the next
 * multiple of 4 of K; the values there are ignored), tn: rows_padded = 1 (rows of A / B readable up to the next multiple
 * of 4 of M / N; those columns only reach masked outputs) -- which keeps these launches on the 16-B-load paths. */
int ft_bgemm_nt(const float* A, long lda, long sA0, long sA1, const float* Bm, long ldb, long sB0, long sB1, float* C,
                long ldc, long sC0, long sC1, int M, int N, int K, int nb0, int nb1, void* stream);
int ft_bgemm_nn(const float* A, long lda, long sA0, long sA1, const float* Bm, long ldb, long sB0, long sB1, float* C,
                long ldc, long sC0, long sC1, int M, int N, int K, int nb0, int nb1, int a_rows_padded, void* stream);
size_t ft_bgemm_tn_workspace(int M, int N, int R, int nb0, int nb1);
int ft_bgemm_tn(const float* A, long lda, long sA0, long sA1, const float* Bm, long ldb, long sB0, long sB1, float* C,
                long ldc, long sC0, long sC1, int M, int N, int R, int nb0, int nb1, int rows_padded, void* workspace,
                size_t workspace_bytes, void* stream);
/* scores[B,nh,Tq,Tk] (row stride ld >= Tk floats; columns Tk..ld-1 are written as zeros) <- softmax(scale*scores +
 * mask) in place; key_pad[B,Tk] bytes (non-zero = padded key) or NULL.  dropout_p > 0 also writes dropped =
 * F.dropout(P, p) (nn.MultiheadAttention's attention dropout; same layout) with ft_dropout's counter-based mask over the
 * flat LOGICAL [B,nh,Tq,Tk] index, in the same pass. */
int ft_softmax_fwd(float* scores, const unsigned char* key_pad, int B, int nh, int Tq, int Tk, long ld, float scale,
                   float* dropped, float dropout_p, uint64_t dropout_seed, void* stream);
/* dprobs <- scale * P * (dprobs' - rowsum(dprobs'*P)) in place (pad columns -> 0); dropout_p > 0: dprobs arrives as the
 * gradient of the DROPPED probabilities and dprobs' = mask * dprobs / (1-p) is formed here (same seed as the forward) */
int ft_softmax_bwd(const float* probs, float* dprobs, int B, int nh, int Tq, int Tk, long ld, float scale, float dropout_p,
                   uint64_t dropout_seed, void* stream);
/* nn.LayerNorm(D) over the last dim with an optional fused residual add: s = x (+res) (stored to sum_out if not
 * NULL), y = LN(s); per-row mean / rstd saved.  bwd: dx (gradient wrt s) and dy_xhat = dy*xhat whose column
 * sums are dgamma (dbeta = column sums of dy), via ft_colsum. */
/* res_dropout_p > 0: the residual branch is F.dropout(res, p) computed on the fly with ft_dropout's counter-based mask
 * (seed, flat element index) -- FFTBlock's norm(src + dropout(src2)) (common_layers.py:175-176,181-183) in one pass;
 * the backward then also writes dres = d/d(res) = mask * dx / (1-p) (dres may be NULL). */
int ft_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* sum_out,
                     float* y, float* mean, float* rstd, long rows, int D, float eps, float res_dropout_p,
                     uint64_t res_dropout_seed, void* stream);
int ft_layernorm_bwd(const float* dy, const float* s, const float* gamma, const float* mean, const float* rstd,
                     float* dx, float* dy_xhat, float* dres, long rows, int D, float res_dropout_p,
                     uint64_t res_dropout_seed, void* stream);
/* PositionalEncoding (common_layers.py:127-145): out = x + scale[0]*pe[t,:]; dscale = sum(dout*pe) */
int ft_posenc_fwd(const float* x, const float* pe, const float* scale, float* out, int B, int T, int D,
                  void* stream);
size_t ft_posenc_workspace(void);
int ft_posenc_bwd_scale(const float* dout, const float* pe, float* dscale, int B, int T, int D, void* workspace,
                        size_t workspace_bytes, void* stream);
int ft_relu_bwd(const float* dy, const float* y, float* dx, long n, void* stream);

/* ---- LengthRegulator (common_layers.py:12-24) ------------------------------------------------------- */
/* scan: clamps dur[dur<0]=0 IN PLACE (as the reference does), r=(long)(dur+0.5); cum[b][0..Tx] exclusive
 * frame offsets (int32, [B,Tx+1]); total[b] = cum[b][Tx].  Caller reads max(total) to size the output. */
int ft_lr_scan(float* dur, int B, int Tx, int* cum, int* total, void* stream);
/* y[b,t,:] = x[b,tok(t),:] for t < total[b], else 0; optional src_idx[B,Tm] (token or -1). Bit-exact copy. */
int ft_lr_expand(const float* x, const int* cum, float* y, int* src_idx, int B, int Tx, int Tm, int C,
                 void* stream);
/* dx[b,j,:] = sum of dy rows of token j (fixed order) */
int ft_lr_bwd(const float* dy, const int* cum, float* dx, int B, int Tx, int Tm, int C, void* stream);
/* The same pair with the FRAME side time-major ([Tm,B,C]) -- the layout of the recurrences' buffers.  They carry the
 * decoder LSTM's input projection through the LengthRegulator: the projection is a row-wise linear map and the
 * regulator only repeats rows, so x_tok * W_ih^T + b_ih is formed once per TOKEN (4,096 rows at the benchmark shape
 * instead of 26,912 frames; same kernel, same k order: every row has the bits the frame-level GEMM gives it) and
 * ft_lr_expand_tm writes it out per frame; pad_row (C floats, optional) is what frames beyond an item's length hold
 * -- the projection of the regulator's zero rows is the bias.  Backward: ft_lr_bwd_tm sums d(pre-activations) over each
 * token's frames, and the input / weight gradients of W_ih are token-level GEMMs (forward_tacotron.py:145-152). */
int ft_lr_expand_tm(const float* x, const int* cum, const float* pad_row, float* y, int B, int Tx, int Tm, int C,
                    void* stream);
/* dtail (optional, [B,C]): per item, the sum of the frames beyond its last token (t >= total[b]) -- they reach no token,
 * but the bias gradients are column sums over ALL frames (an unpacked LSTM runs over those frames too) */
int ft_lr_bwd_tm(const float* dy, const int* cum, float* dx, float* dtail, int B, int Tx, int Tm, int C, void* stream);

/* ---- nn.BatchNorm1d of BatchNormConv (common_layers.py:51,57) on channels-last y[B,Tbuf,C] ----------- */
/* group > 0 = CBHG bank buffer [B,T+1,K*group]: channel c belongs to kernel size k=c/group+1 and has
 * Tbuf-1 valid rows for odd k, Tbuf for even k (common_layers.py:97-99); group = 0: all Tbuf rows valid.
 * train fwd: batch statistics (biased var) over valid rows -> save_mean/save_rstd; running stats updated
 * in place (momentum, unbiased var), *num_batches_tracked += 1; out[B,Tout,C] = bn(y)[:, :Tout] (+residual). */
size_t ft_bn_workspace(int B, int Tbuf, int C);
/* BatchNormConv with the statistics pass fused into the convolution (north_star: "BatchNormConv fused"): the conv's
 * GEMM epilogue reduces (sum, sum of squares) of its 128-row output tile per channel -- doubles, fixed order -- into
 * `partial` ([nchunks][C][2], ft_conv_stats_workspace bytes) while the tile is still in the MFMA accumulators, so the
 * activation is not read back for the statistics; launches that do not take the 128x128 kernel (token-side shapes) run
 * the stand-alone column pass instead -- either way *nchunks (HOST int) tells ft_bn_train_from_partials how many partial
 * rows to finalize (ordered sum; mean / biased var / running stats as ft_bn_train_fwd) before it normalises.
 * conv1d: y contiguous [B,Tout,Cout].  conv_bank: training-mode bank buffer [B,T+1,K*C]; odd-k members count T rows.
 * ft_conv1d_fwd_stats_workspace >= ft_conv_stats_workspace: it adds room for per-tap output slabs where a token-side
 * convolution (too few 128x128 tiles to fill the chip) runs its k taps as independent tasks followed by one ordered sum;
 * a caller that passes only ft_conv_stats_workspace bytes simply gets the single-task form. */
size_t ft_conv_stats_workspace(int B, int Tbuf, int C);
size_t ft_conv1d_fwd_stats_workspace(int B, int Tout, int Cout, int k);
int ft_conv1d_fwd_stats(const float* x, long ldx, const float* wp, float* y, long ldy, int B, int T, int Cin, int Cout,
                        int k, int Tout, int relu, double* partial, size_t partial_bytes, int* nchunks, void* stream);
int ft_conv_bank_fwd_stats(const float* x, long ldx, const float* wp_all, float* ybank, int B, int T, int Cin, int C,
                           int K, int relu, double* partial, size_t partial_bytes, int* nchunks, void* stream);
int ft_bn_train_from_partials(const double* partial, int nchunks, const float* y, const float* gamma, const float* beta,
                              const float* residual, float* out, float* running_mean, float* running_var,
                              long* num_batches_tracked, float* save_mean, float* save_rstd, int B, int Tbuf, int Tout,
                              int C, int group, float momentum, float eps, void* stream);
/* CBHG conv bank, training mode (common_layers.py:100-105): BatchNorm apply fused with MaxPool1d(2,1,1)[:Tout].
 * ft_bn_pool_from_partials: finalize the partials as ft_bn_train_from_partials, then out[B,Tout,C][t] =
 *   max(z[t-1], z[t]) with z = bn(y) recomputed on the fly -- z is never written.
 * ft_bn_pool_bwd: dout[B,Tout,C] is the gradient of the POOLED output; the BatchNorm output's gradient (torch's
 *   max_pool rule: a window's gradient goes to its first maximal element) is recomputed from y in both passes
 *   (statistics and apply), so neither z nor dz ever exists in memory.  dy / dgamma / dbeta as ft_bn_bwd.
 * Both need C % 4 == 0, group % 4 == 0 and 16-byte aligned buffers (FT_ERR otherwise: callers fall back to
 * ft_bn_train_from_partials + ft_maxpool2_fwd / ft_maxpool2_bwd + ft_bn_bwd). */
int ft_bn_pool_from_partials(const double* partial, int nchunks, const float* y, const float* gamma, const float* beta,
                             float* out, float* running_mean, float* running_var, long* num_batches_tracked,
                             float* save_mean, float* save_rstd, int B, int Tbuf, int Tout, int C, int group,
                             float momentum, float eps, void* stream);
int ft_bn_pool_bwd(const float* dout, const float* y, const float* gamma, const float* beta, const float* save_mean,
                   const float* save_rstd, float* dy, float* dgamma, float* dbeta, int B, int Tbuf, int Tout, int C,
                   int group, int relu, void* workspace, size_t workspace_bytes, void* stream);
int ft_bn_train_fwd(const float* y, const float* gamma, const float* beta, const float* residual, float* out,
                    float* running_mean, float* running_var, long* num_batches_tracked, float* save_mean,
                    float* save_rstd, int B, int Tbuf, int Tout, int C, int group, float momentum, float eps,
                    void* workspace, size_t workspace_bytes, void* stream);
/* dy[B,Tbuf,C] (0 on invalid rows), dgamma, dbeta from dout[B,Tout,C]; relu=1 also applies the ReLU mask
 * (y>0) of the conv->ReLU->BN order (common_layers.py:55-57) */
int ft_bn_bwd(const float* dout, const float* y, const float* gamma, const float* save_mean, const float* save_rstd,
              float* dy, float* dgamma, float* dbeta, int B, int Tbuf, int Tout, int C, int group, int relu,
              void* workspace, size_t workspace_bytes, void* stream);
/* eval mode: scale = gamma/sqrt(running_var+eps), shift = beta - running_mean*scale (fed to the conv epilogue) */
int ft_bn_fold_eval(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                    float eps, float* scale, float* shift, int C, void* stream);
/* out[c] (+)= scale * sum_rows x[row*ldx + c]   (bias gradients; ordered, reproducible) */
size_t ft_colsum_workspace(int rows, int C);
int ft_colsum(const float* x, long ldx, float* out, int rows, int C, float scale, int accumulate, void* workspace,
              size_t workspace_bytes, void* stream);
/* column sums of two matrices of one shape (x0 -> out0, x1 -> out1) in one partial + one finalize launch: LayerNorm's
 * dgamma = colsum(dy * xhat), dbeta = colsum(dy) */
int ft_colsum2(const float* x0, const float* x1, long ldx, float* out0, float* out1, int rows, int C, void* workspace,
               size_t workspace_bytes, void* stream);
/* n <= 16 column sums over matrices with the same row count in ONE partial + ONE finalize launch: out[i][c] = sum_r
 * x[i][r * ld[i] + c].  Every sum keeps the chunking and summation order of its own ft_colsum call (bit-identical). */
size_t ft_colsum_batch_workspace(int n, const int* C, int rows);
int ft_colsum_batch(int n, const float* const* x, const long* ld, float* const* out, const int* C, int rows,
                    void* workspace, size_t workspace_bytes, void* stream);

/* ---- F.dropout (forward_tacotron.py:35 ; common_layers.py:106,110) and scalar scale (x/alpha, :39) ----- */
/* out = keep ? x/(1-p) : 0 with keep(i) = hash(seed,i) >= p; calling it on the gradient with the same seed
 * is the backward (no mask tensor).  torch's RNG stream cannot be matched: parity runs use p = 0. */
int ft_dropout(const float* x, float* out, long n, float p, uint64_t seed, void* stream);
int ft_scale(const float* x, float* out, long n, float s, void* stream);
/* n <= 64 independent device-to-device copies in ONE launch: dst[i][0:len[i]] = src[i][0:len[i]]  (src / dst / len are
 * HOST arrays).  Used to hand fused-kernel results (a conv bank's K dgamma / dbeta slices, an RNN's four bias
 * gradients) to their slots in the flat gradient buffer without one copy launch per parameter. */
int ft_copy_segments(const float* const* src, float* const* dst, const long* len, int n, void* stream);

/* ---- nn.Embedding (forward_tacotron.py:18,31,73,133) -------------------------------------------------- */
/* out[row,:] = w[idx[row],:] ; *err_flag set to 1 on an out-of-range index (row zero-filled) */
int ft_embedding_fwd(const long* idx, const float* w, float* out, long rows, int C, int V, int* err_flag,
                     void* stream);
/* backward: onehot[rows,V] = (idx == v); dW[V,C] = onehot^T * dout via ft_linear_bwd_weight (ordered, reproducible) */
int ft_onehot(const long* idx, float* out, long rows, int V, void* stream);

/* ---- HighwayNetwork gate (common_layers.py:35-40); x12 = [W1 x + b1 | W2 x + b2] from ft_linear_multi_fwd */
int ft_highway_gate_fwd(const float* x12, const float* x, float* out, long rows, int C, void* stream);
/* d12 = gradient wrt x12, dx = direct-path gradient dout*(1-g) (caller adds d12*[W1;W2]) */
int ft_highway_gate_bwd(const float* dout, const float* x12, const float* x, float* d12, float* dx, long rows, int C,
                        void* stream);

/* The same layer with the gate INSIDE the GEMMs (C % 32 == 0; the two calls above stay for other widths).
 * ft_highway_pack: w12i [2C, C] = the 32-row interleave of W1 and W2 (row n: unit (n/64)*32 + n%32 of W1 if n%64 < 32, else
 *   of W2), so that one accumulator lane pair holds W1 x and W2 x of the same unit.
 * ft_highway_fwd: out [rows, C] = g * relu(y1) + (1 - g) * x with y1 | y2 = x W1^T + b1 | x W2^T + b2, g = sigmoid(y2), from
 *   the epilogue of ONE product x * w12i^T; x12 (may be NULL) receives y1 | y2 as [rows, 2C] for the backward.
 * ft_highway_bwd_data: dx [rows, C] += d12[:, :C] W1 + d12[:, C:] W2 (dx holds the direct-path term; w1 / w2 are W^T
 *   [C, C] if w_transposed).  With below_* set, dx is d(out) of the highway layer BELOW and the epilogue turns it into that
 *   layer's ft_highway_gate_bwd outputs at once: below_d12 [rows, 2C] and dx = direct-path term of the layer below
 *   (below_x12 / below_x = that layer's saved pre-activations and input). */
int ft_highway_pack(const float* w1, const float* w2, float* w12i, int C, void* stream);
int ft_highway_fwd(const float* x, const float* w12i, const float* b1, const float* b2, float* out, float* x12, int rows,
                   int C, void* stream);
int ft_highway_bwd_data(const float* d12, const float* w1, const float* w2, int w_transposed, float* dx, int rows, int C,
                        const float* below_x12, const float* below_x, float* below_d12, void* stream);

/* ---- MaxPool1d(kernel 2, stride 1, padding 1)[:T] (common_layers.py:78,105): out[t]=max(x[t-1],x[t]) ---- */
int ft_maxpool2_fwd(const float* x, float* out, int B, int T, int C, void* stream);
int ft_maxpool2_bwd(const float* dout, const float* x, float* dx, int B, int T, int C, void* stream);

/* ---- pitch/energy conditioning (forward_tacotron.py:111-112,137-143): Conv1d(1->C,k3,p1)+bias, scaled add */
int ft_cond_add_fwd(const float* x, const float* pitch, const float* energy, const float* w_pitch,
                    const float* b_pitch, const float* w_energy, const float* b_energy, float pitch_strength,
                    float energy_strength, float* out, int B, int T, int C, int x_time_major, void* stream);
/* taps[row][8] = [p[t-1],p[t],p[t+1],1,e[t-1],e[t],e[t+1],1]; weight/bias grads = dy^T taps (ft_linear_bwd_weight) */
int ft_cond_taps(const float* pitch, const float* energy, float* taps, int B, int T, void* stream);

/* ---- multispeaker plumbing (models/multi_forward_tacotron.py:39-42,83-85,208-210) ---------------------- */
/* out[b,t,:] = [ a[b,t,:Ca] | b2[b,t,:Cb] | semb[b,:S] ]  (speaker embedding broadcast over t); a may be a
 * time-major [T,B,Ca] recurrence output; b2 / semb optional (Cb = 0 / S = 0) */
int ft_concat_cols(const float* a, int Ca, const float* b2, int Cb, const float* semb, int S, float* out, int B,
                   int T, int a_time_major, void* stream);
/* dst[b,t,:C] = src[(b,t)*ld + col0 + :C] ; dst optionally time-major (backward of the concat) */
int ft_slice_cols(const float* src, long ld, int col0, int C, float* dst, int B, int T, int dst_time_major,
                  void* stream);
/* nn.CrossEntropyLoss(ignore_index) on logits[rows,K] / int64 target[rows] (trainer/multi_forward_trainer.py:34,88) */
size_t ft_cross_entropy_workspace(void);
int ft_cross_entropy_fwd(const float* logits, const long* target, long rows, int K, long ignore_index, float* loss,
                         float* inv_count, void* workspace, size_t workspace_bytes, void* stream);
int ft_cross_entropy_bwd(const float* logits, const long* target, const float* inv_count, const float* grad_out,
                         float* dlogits, long rows, int K, long ignore_index, void* stream);

/* ---- output layout + ForwardTacotron._pad (forward_tacotron.py:155,159,161-162,236-239) ---------------- */
/* out[b,c,t] = t < T ? x[b,t,c] : pad, t < Tout   ([B,T,C] -> [B,C,Tout]) ; bwd is the masked transpose back */
int ft_transpose_pad_fwd(const float* x, float* out, int B, int T, int C, int Tout, float pad, void* stream);
int ft_transpose_pad_bwd(const float* dout, float* dx, int B, int T, int C, int Tout, void* stream);

/* ---- MaskedL1 (trainer/common.py:69-92) on [B,C,T] with int64 lens ----------------------------------- */
size_t ft_masked_l1_workspace(void);
int ft_masked_l1_fwd(const float* x, const float* target, const long* lens, float* loss, float* inv_denom, int B,
                     int C, int T, void* workspace, size_t workspace_bytes, void* stream);
/* dx = sign(x-target)*mask*inv_denom * factor * (grad_out ? grad_out[0] : 1) */
int ft_masked_l1_bwd(const float* x, const float* target, const long* lens, const float* inv_denom,
                     const float* grad_out, float factor, float* dx, int B, int C, int T, void* stream);

/* ---- nn.GRU(bidirectional, batch_first), h0 = 0 (common_layers.py:89,123 ; forward_tacotron.py:24,37) - */
/* ALL recurrence buffers are TIME-major: xp[T,B,2*3H] = x W_ih^T + b_ih (dir 0 | dir 1); out[T,B,2H];
 * gates[T,B,2,4H] = (r,z,n,W_hn h+b_hn) or NULL */
/* `workspace` (ft_rnn_workspace(gates,B,H) bytes, may be NULL) enables the PERSISTENT form: one launch runs all T
 * steps with W_hh resident in registers and h exchanged between workgroups inside the kernel (bounded spins).
 * It is used when H % 16 == 0, the grid is co-resident per the occupancy query and FT_RNN_PERSISTENT != 0;
 * otherwise one kernel per timestep is launched.  A launch is admitted only while the persistent grids still in flight
 * on OTHER streams of the device plus its own fit every XCD in the worst case, because every workgroup spins until its
 * whole grid is resident.  Grids whose (direction, batch group) groups each fit one XCD slot are laid out that way and
 * hand h / d(gates) over through that XCD's L2 (placement verified in the kernel, agent-scope protocol otherwise).
 * Faults: a workgroup whose bounded poll runs out sets the device's STICKY fault word and leaves (the grid drains);
 * no launch clears that word.  ft_clip_grad_norm / ft_adam_step read it on the device and skip the parameter update;
 * ft_rnn_status(clear) synchronises the device, returns an error if the word is set and (clear != 0) resets it. */
size_t ft_rnn_workspace(int gates, int B, int H);
int ft_rnn_status(int clear);
/* runtime override of FT_RNN_PERSISTENT (1 = allow the persistent form); returns the previous setting */
int ft_rnn_set_persistent(int enabled);
/* bound of the arrival polls (0 restores the default 2^18; -1 = fault injection for tests: every poll of every
 * later launch fails at once, deterministically); returns the previous bound (0 while injecting) */
int ft_rnn_set_max_spins(int max_spins);
/* Tell the admission bookkeeping that `waiting_stream` has just been made to wait for everything enqueued so far on
 * `joined_stream` (hipStreamWaitEvent / torch's wait_stream): the joined stream's earlier persistent launches then
 * precede whatever the waiting stream launches next and no longer count against it.  Optional -- without it the
 * bookkeeping is merely conservative. */
int ft_rnn_note_join(void* waiting_stream, void* joined_stream);
/* A HIP stream restricted to the first `cus_per_xcd` CUs of every XCD (hipExtStreamCreateWithCUMask).  New work, no
 * reference counterpart (the reference is single-stream, trainer/forward_trainer.py:69-99): trainer.TrainStep puts its
 * weight-gradient side stream on one, so that the one-wave weight-gradient GEMMs leave a few CUs per XCD to the small
 * dependent kernels of the step's critical stream.  ft_stream_destroy releases it. */
int ft_stream_create_cu_limited(int cus_per_xcd, void** stream);
int ft_stream_destroy(void* stream);
/* (direction, batch group) groups of persistent launches that ran the XCD-local hand-off / the agent-scope one since
 * the library loaded (synchronises the device) */
int ft_rnn_mode_counts(long* xcd_local_groups, long* agent_scope_groups);
/* launches that ran in the persistent form / that did not fit the chip even alone (and ran per-step) since the library
 * loaded; ft_rnn_waited_launches: persistent launches whose stream was first made to wait for other streams' */
int ft_rnn_counters(long* persistent_launches, long* refused_launches);
/* Share (percent) of ONE XCD's CUs the persistent recurrence of this shape holds while it runs (gates 3 = GRU, 4 = LSTM;
 * backward != 0: the BPTT kernel); -1: it would not run persistent.  Two persistent launches on different streams are
 * co-resident while their shares add up to at most ft_rnn_admit_budget_pct(); a launch at 100 fills whole XCDs, and no
 * other kernel is dispatched anywhere while it is resident (the dispatcher deals workgroups to the XCDs round-robin and
 * waits at the first full one). */
int ft_rnn_xcd_fill_pct(int gates, int backward, int B, int T, int H);
int ft_rnn_admit_budget_pct(void);
int ft_rnn_waited_launches(void);
int ft_gru_fwd(const float* xp, const float* whh_f, const float* whh_r, const float* bhh_f, const float* bhh_r,
               float* out, float* gates, int B, int T, int H, void* workspace, size_t workspace_bytes,
               void* stream);
/* BPTT: dout[T,B,2H]; whhT = W_hh^T [H,3H]; dxp / dhp [T,B,2*3H] = d(pre-activations) wrt input / hidden
 * projections; carry [B,2,H] scratch.  Weight grads follow from ft_linear_bwd_weight on dxp / dhp. */
int ft_gru_bwd(const float* dout, const float* out, const float* gates, const float* whhT_f, const float* whhT_r,
               float* dxp, float* dhp, float* carry, int B, int T, int H, void* workspace, size_t workspace_bytes,
               void* stream);

/* ---- pack_padded_sequence -> nn.LSTM(bidirectional) -> pad_packed_sequence (forward_tacotron.py:96-99,147-152)
 * lens (int64 [B], device) or NULL = run over the padded length (generate path, :224).  Item b is processed
 * over exactly lens[b] frames in both directions; out_raw / cstate [T,B,2H] (time-major, like xp / gates /
 * dgates) are ZERO at t >= lens[b]; ft_fill_padded converts to batch-major [B,T,2H] and writes the
 * padding_value the reference's unpack inserts (lens NULL: plain layout change). xp includes b_ih; b_hh added here. */
int ft_lstm_fwd(const float* xp, const float* whh_f, const float* whh_r, const float* bhh_f, const float* bhh_r,
                const long* lens, float* out_raw, float* cstate, float* gates, int B, int T, int H,
                void* workspace, size_t workspace_bytes, void* stream);
int ft_lstm_bwd(const float* dout, const float* out_raw, const float* cstate, const float* gates,
                const float* whhT_f, const float* whhT_r, const long* lens, float* dgates, float* carry, int B, int T,
                int H, void* workspace, size_t workspace_bytes, void* stream);
/* ---- a recurrent LAYER's forward with the input projection overlapped with the recurrence (no reference counterpart:
 * nn.LSTM / nn.GRU, forward_tacotron.py:96-99,147-152 / common_layers.py:89,123, run the projection in front).
 * x [B,T,in_f] batch-major; xp [T,B,2*G*H] (scratch for the projection, G = 4 | 3); the other arguments as in
 * ft_lstm_fwd / ft_gru_fwd.  x * W_ih^T + b_ih is formed in `nchunks` time chunks: chunk 0 of both directions on
 * `stream`, the others on `side_stream`, each followed by a one-thread kernel that raises gate[direction]; the
 * persistent recurrence is launched right behind chunk 0 and polls (bounded) the gate word before it reads a row of a
 * chunk it has not seen complete.  gate: >= 2 device words owned by the call.  rev_lead: chunks of the reverse direction
 * issued first (a packed item of length L starts at t = L - 1).  nchunks < 2 / side_stream NULL or == stream: the
 * projection runs whole, in front.  If the persistent form is refused, `stream` first waits for all chunks.  Results are
 * bit-identical to ft_linear_multi_fwd + ft_lstm_fwd / ft_gru_fwd. */
int ft_lstm_layer_fwd(const float* x, int in_f, const float* wih_f, const float* wih_r, const float* bih_f,
                      const float* bih_r, float* xp, const float* whh_f, const float* whh_r, const float* bhh_f,
                      const float* bhh_r, const long* lens, float* out_raw, float* cstate, float* gates, int B, int T,
                      int H, void* workspace, size_t workspace_bytes, unsigned* gate, int nchunks, int rev_lead,
                      void* stream, void* side_stream);
int ft_gru_layer_fwd(const float* x, int in_f, const float* wih_f, const float* wih_r, const float* bih_f,
                     const float* bih_r, float* xp, const float* whh_f, const float* whh_r, const float* bhh_f,
                     const float* bhh_r, float* out, float* gates, int B, int T, int H, void* workspace,
                     size_t workspace_bytes, unsigned* gate, int nchunks, void* stream, void* side_stream);
int ft_fill_padded(const float* raw, const long* lens, float* out, int B, int T, int C, float pad, void* stream);
int ft_mask_rows(const float* src, const long* lens, float* dst, int B, int T, int C, void* stream);
/* [B,T,C] -> [T,B,C] (dst_time_major = 1) or back (0) */
int ft_bt_transpose(const float* src, float* dst, int B, int T, int C, int dst_time_major, void* stream);

/* ---- clip_grad_norm_ + torch.optim.Adam (trainer/forward_trainer.py:95-99 ; train_forward.py:76) ------ */
/* over FLAT fp32 buffers (all parameters back to back, 16-B aligned).  coef_and_norm holds FOUR floats:
 * [0] = pre_scale * min(1, max_norm/(norm+1e-6)), [1] = norm = pre_scale*||grads||_2 (pre_scale = 1/world_size when
 * the buffer holds an all-reduced SUM), [2] = 1 if the device's recurrence-fault word is set, 2 if only fault_lane says
 * so (then [0] = 0, [1] = NaN and ft_adam_step leaves params / moments untouched), [3] = 0; max_norm <= 0 disables
 * clipping.  Stays on device: no host sync.
 * Data parallelism (new work: the reference is single device, train_forward.py:70): a recurrence fault must be GLOBAL --
 * the faulting rank's garbage gradient is summed into every rank's buckets.  ft_fault_lane_set writes this device's fault
 * word as 1.0 / 0.0 into lane[0] (lane = 4 floats, 16-B aligned; [1..3] = 0); the caller SUM-all-reduces the lane with the
 * gradient and hands it to ft_clip_grad_norm as fault_lane (NULL: single device): a non-zero lane[0] skips the update on
 * every rank alike. */
size_t ft_grad_norm_workspace(void);
int ft_clip_grad_norm(const float* grads, long n, float max_norm, float pre_scale, const float* fault_lane,
                      float* coef_and_norm, void* workspace, size_t workspace_bytes, void* stream);
int ft_fault_lane_set(float* lane, void* stream);
/* dst[0..nwords) <- snapshot (4-byte words) iff coef[2] != 0 (ft_clip_grad_norm's record): puts the buffers a forward
 * pass updates in place (BatchNorm running_mean / running_var / num_batches_tracked, forward_tacotron.py:101,126-127
 * `step`) back to their values from before a faulted step; a no-op launch otherwise */
int ft_guarded_restore(void* dst, const void* snapshot, long nwords, const float* coef, void* stream);
/* Adam, torch defaults semantics (no weight decay / amsgrad): g = grads*coef[0]; step counts from 1; coef (may be NULL)
 * is ft_clip_grad_norm's 4-float record: a set [2] skips the update */
int ft_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
                 float beta2, float eps, long step, const float* coef, void* stream);

/* ---- fused self-attention, bf16 matmul mode (nn.MultiheadAttention inside FFTBlock, common_layers.py:172-174) ---- */
/* qkv [B,T,3d] = the in-projection's output rows (q | k | v, d = nheads*hd, head h at columns h*hd of each third), fp32;
 * key_pad [B,T] bytes (non-zero = padded key) or NULL.  ft_attn_fwd: att [B,T,d] = concat_h softmax(scale q_h k_h^T +
 * mask) v_h with attention dropout p_drop (the counter-based mask of ft_softmax_fwd: element index = flat index into
 * [B,nheads,T,T], same seed -> same mask), and lse2 [B,nheads,T] = log2-domain log-sum-exp per query row for the
 * backward.  One flash-style launch: the [B,nheads,T,T] scores never reach memory.  Operands are rounded to bf16 while
 * staged (bf16 MFMA, fp32 statistics / accumulation / outputs): this IS the bf16 mode, there is no fp32-exact variant (the
 * fp32 mode keeps ft_bgemm_* + ft_softmax_*).  hd = 64 or 128.
 * ft_attn_bwd: dqkv [B,T,3d] (every element written) from datt = d(att); recomputes the probabilities from qkv and lse2
 * (three launches: row sums of datt*att, dQ, dK+dV; no atomics, bitwise reproducible).  workspace: ft_attn_workspace. */
size_t ft_attn_workspace(int B, int T, int nheads);
int ft_attn_fwd(const float* qkv, const unsigned char* key_pad, float* att, float* lse2, int B, int T, int nheads, int hd,
                float scale, float p_drop, uint64_t seed, void* stream);
int ft_attn_bwd(const float* qkv, const float* att, const float* datt, const unsigned char* key_pad, const float* lse2,
                float* dqkv, int B, int T, int nheads, int hd, float scale, float p_drop, uint64_t seed, void* workspace,
                size_t workspace_bytes, void* stream);

/* ---- a whole FFTBlock per call (common_layers.py:148-185), bf16 matmul mode with the fused attention ------------ */
/* The FastPitch step is ~900 launches for ~13 ms of GPU work: issued one entry point at a time from Python it is bound by
 * the host (15 us per launch).  These two calls issue every launch of n consecutive FFTBlocks -- forward: in-projection,
 * ft_attn_fwd, out-projection, add + LayerNorm, conv1 + ReLU, conv2, add + LayerNorm; backward: the reverse, with the
 * two residual joins as accumulate epilogues, then the blocks' weight / bias / LayerNorm gradients on wgrad_stream behind
 * one event -- from C.  All buffers are the caller's; rows = B*T, row-major.  Same kernels, same order, same results as
 * the single entry points above. */
typedef struct FtFFTBlock {
  int B, T, d, nheads, dfft, k1, k2;
  float p_drop, eps1, eps2;
  uint64_t seed_attn, seed_ln1, seed_ln2;
  const unsigned char* key_pad;                  /* [B,T] or NULL */
  /* parameters: torch layouts, the convolutions as tap-major packs [k][Cout][Cin] */
  const float *in_w, *in_b, *out_w, *out_b, *c1_wp, *c1_b, *c2_wp, *c2_b, *n1_g, *n1_b, *n2_g, *n2_b;
  /* backward only: W^T of the two projections and the transposed packs [k][Cin][Cout] (data gradients in the NT form) */
  const float *in_wT, *out_wT, *c1_wpt, *c2_wpt;
  /* activations: x [rows,d] in; the forward writes the rest, the backward reads them */
  const float* x;
  float *qkv, *att, *lse2, *sa, *s1, *mean1, *rstd1, *y1, *h1, *h2, *s2, *mean2, *rstd2, *y2;
} FtFFTBlock;
typedef struct FtFFTBlockGrads {
  const float* dy2;                              /* d(y2) [rows,d] */
  float* dx;                                     /* out: d(x) [rows,d] */
  /* scratch, kept by the caller until wgrad_stream has been joined: [rows,d] each except d_h1 / g_h1 [rows,dfft] and
   * dqkv [rows,3d] */
  float *t2, *d_y1, *d_h2, *d_h1, *g_h1, *t1, *d_h, *d_sa, *datt, *dqkv;
  /* parameter gradients, overwritten (torch layouts) */
  float *g_in_w, *g_in_b, *g_out_w, *g_out_b, *g_c1_w, *g_c1_b, *g_c2_w, *g_c2_b, *g_n1_g, *g_n1_b, *g_n2_g, *g_n2_b;
} FtFFTBlockGrads;
int ft_fft_blocks_fwd(const FtFFTBlock* blocks, int n, void* stream);
/* workspace: >= ft_attn_workspace(B,T,nheads) bytes (main stream); wgrad_workspace: >= ft_fft_block_wgrad_workspace(...)
 * bytes, used on wgrad_stream only (the four weight-gradient GEMMs of a block); sums_workspace: >=
 * ft_fft_block_sums_workspace(...) bytes, used on sums_stream only (its bias / LayerNorm column sums).  wgrad_stream /
 * sums_stream NULL = stream / wgrad_stream; sums_workspace NULL = the sums share wgrad_stream and its workspace. */
size_t ft_fft_block_wgrad_workspace(int B, int T, int d, int dfft, int k1, int k2);
size_t ft_fft_block_sums_workspace(int B, int T, int d, int dfft);
int ft_fft_blocks_bwd(const FtFFTBlock* blocks, const FtFFTBlockGrads* grads, int n, void* workspace,
                      size_t workspace_bytes, void* wgrad_workspace, size_t wgrad_workspace_bytes, void* sums_workspace,
                      size_t sums_workspace_bytes, void* stream, void* wgrad_stream, void* sums_stream);

/* ---- mel inversion + Griffin-Lim (utils/dsp.py:80-94 DSP.griffinlim ; gen_forward.py:109-116) ------------ */
/* The DFTs are GEMMs on ft_linear_fwd (frames read in place out of the zero-padded signal with ldx = hop); these are
 * the element-wise / gather pieces.  Complex spectra are split [N][2*Fp] = Re | Im, Fp = F rounded up to 4.
 * exp_transpose: log-mel [C,T] -> exp -> [T,C].  nnls_step: x = max(0, x - inv_l*g).  sub: out = a - b.
 * gl_init: proj = S * exp(2 pi i u) (u in [0,1), drawn by the host).  gl_phase: c = rebuilt - alpha*tprev (if has_prev);
 * proj = S * c / (|c| + FLT_MIN); tprev = rebuilt.  overlap_add: ypad [n_fft + hop*(N-1)] = sum of the (already windowed)
 * frames [N,n_fft] at offsets n*hop, times inv_wss (1 / summed squared window, 0 where that is below FLT_MIN), zero in
 * the n_fft/2 margins -- the signal sits at ypad + n_fft/2 and the padded buffer is the next STFT's operand. */
int ft_exp_transpose(const float* mel_log, float* out, int C, int T, void* stream);
int ft_nnls_step(float* x, const float* g, float inv_l, long n, void* stream);
int ft_sub(const float* a, const float* b, float* out, long n, void* stream);
int ft_gl_init(const float* u, const float* S, float* proj, int N, int Fp, void* stream);
int ft_gl_phase(const float* rebuilt, float* tprev, const float* S, float* proj, int N, int Fp, float alpha,
                int has_prev, void* stream);
int ft_overlap_add(const float* frames, const float* inv_wss, float* ypad, int N, int n_fft, int hop, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FWDTACO_HIP_H */
