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
 *     stated, so they are capturable in a hipGraph;
 *   - return 0 on success; non-zero = error, message via ft_last_error() (thread-local);
 *   - workspaces are caller-allocated device memory, sized by the matching *_workspace() query (bytes).
 */
#ifndef FWDTACO_HIP_H
#define FWDTACO_HIP_H

#include <stddef.h>
#include <stdint.h>

#define FWDTACO_ABI_VERSION 1

#ifdef __cplusplus
extern "C" {
#endif

const char* ft_last_error(void);
int ft_abi_version(void);
/* host query: CU count and whether device 0 is gfx950 */
int ft_device_info(int* cu_count, int* is_gfx950);

/* ---- nn.Linear (models/forward_tacotron.py:25,100,108 ; common_layers.py:31-32,83) ------------------ */
/* y[rows,out_f] (+)= x[rows,in_f] * w[out_f,in_f]^T + bias ; optional relu */
int ft_linear_fwd(const float* x, long ldx, const float* w, const float* bias, float* y, long ldy, int rows,
                  int in_f, int out_f, int relu, int accumulate, void* stream);
/* several Linear layers sharing one input, written side by side into y (highway W1|W2, RNN fwd|rev W_ih):
 * y[:, col_offset[i] : col_offset[i]+out_f[i]] = x * w[i]^T + bias[i]   (host arrays of device pointers) */
int ft_linear_multi_fwd(const float* x, long ldx, int ntasks, const float* const* w, const float* const* bias,
                        float* y, long ldy, const int* col_offset, const int* out_f, int rows, int in_f, int relu,
                        void* stream);
/* dx[rows,in_f] (+)= dy[rows,out_f] * w[out_f,in_f] */
int ft_linear_bwd_data(const float* dy, long lddy, const float* w, float* dx, long lddx, int rows, int in_f,
                       int out_f, int accumulate, void* stream);
/* dw[out_f,in_f] (+)= dy^T * shift(x): rows = B*T logical positions; x_shift != 0 reads x row (b, t+x_shift),
 * zero outside [0,T) (recurrent-weight gradients: h_{t-1} / h_{t+1}).  Deterministic split + ordered reduce. */
size_t ft_linear_bwd_weight_workspace(int rows, int in_f, int out_f);
int ft_linear_bwd_weight(const float* dy, long lddy, const float* x, long ldx, float* dw, int rows, int in_f,
                         int out_f, int B, int T, int x_shift, int accumulate, void* workspace,
                         size_t workspace_bytes, void* stream);

/* ---- nn.Conv1d(stride 1, padding k//2, bias=False) of BatchNormConv (common_layers.py:50,55-56) ------ */
/* weights are consumed TAP-MAJOR: wp[k][Cout][Cin] (ft_conv_pack_weight from torch's [Cout][Cin][k]).
 * y[b,t',co] = relu?( sum_j sum_ci x[b, t'+j-k/2, ci] * w[co,ci,j] ) (* scale[co] + shift[co] if scale) for
 * t' in [0,Tout); Tout = T (odd k, or even k sliced as the CBHG does, common_layers.py:99) or T+1 (even k). */
int ft_conv_pack_weight(const float* w, float* wp, int Cout, int Cin, int k, void* stream);
int ft_conv1d_fwd(const float* x, long ldx, const float* wp, const float* scale, const float* shift, float* y,
                  long ldy, int B, int T, int Cin, int Cout, int k, int Tout, int relu, void* stream);
/* CBHG conv1d_bank (common_layers.py:72-76,97-102): members k=1..K (K<=16) in ONE launch; member i writes
 * ybank[:, :, i*C:(i+1)*C] of ybank[B,Tout,K*C]; wp_all = packed member weights back to back. */
int ft_conv_bank_fwd(const float* x, long ldx, const float* wp_all, const float* scale, const float* shift,
                     float* ybank, int B, int T, int Cin, int C, int K, int Tout, int relu, void* stream);
/* dx[b,t,ci] (+)= sum_j sum_co dy[b, t-j+k/2, co] * w[co,ci,j]; dy is [B,Tbuf,*] of which rows < Tvalid count */
int ft_conv1d_bwd_data(const float* dy, long lddy, const float* wp, float* dx, long lddx, int B, int T, int Cin,
                       int Cout, int k, int Tbuf, int Tvalid, int accumulate, void* stream);
/* dw[co,ci,j] = sum_{b,t'<Tvalid} dy[b,t',co] * x[b,t'+j-k/2,ci]   (torch layout [Cout][Cin][k]) */
size_t ft_conv1d_bwd_weight_workspace(int B, int T, int Cin, int Cout, int k, int Tvalid);
int ft_conv1d_bwd_weight(const float* dy, long lddy, const float* x, long ldx, float* dw, int B, int T, int Cin,
                         int Cout, int k, int Tbuf, int Tvalid, void* workspace, size_t workspace_bytes,
                         void* stream);

/* ---- LengthRegulator (common_layers.py:12-24) ------------------------------------------------------- */
/* scan: clamps dur[dur<0]=0 IN PLACE (as the reference does), r=(long)(dur+0.5); cum[b][0..Tx] exclusive
 * frame offsets (int32, [B,Tx+1]); total[b] = cum[b][Tx].  Caller reads max(total) to size the output. */
int ft_lr_scan(float* dur, int B, int Tx, int* cum, int* total, void* stream);
/* y[b,t,:] = x[b,tok(t),:] for t < total[b], else 0; optional src_idx[B,Tm] (token or -1). Bit-exact copy. */
int ft_lr_expand(const float* x, const int* cum, float* y, int* src_idx, int B, int Tx, int Tm, int C,
                 void* stream);
/* dx[b,j,:] = sum of dy rows of token j (fixed order) */
int ft_lr_bwd(const float* dy, const int* cum, float* dx, int B, int Tx, int Tm, int C, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FWDTACO_HIP_H */
