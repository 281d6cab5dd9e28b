// FastPitch building blocks (models/common_layers.py:127-223, models/fast_pitch.py): masked softmax of the
// attention scores, LayerNorm (+ fused residual add), PositionalEncoding, ReLU gradient mask.
// All row-wise, one wave64 per row, HBM-bound.  The matmuls of the attention (QK^T, PV and their
// gradients) are strided-batch launches of the f32-MFMA GEMM kernels (ft_bgemm_* in ft_capi_core.hip).
#include "ft_common.h"

namespace {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum_all(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// P[row, :] = softmax(scale * S[row, :] + mask) in place; row = ((b*nh + h)*Tq + q); key k masked when
// key_pad[b*Tk + k] != 0 (nn.MultiheadAttention key_padding_mask, common_layers.py:172-174)
__global__ __launch_bounds__(256) void ft_softmax_fwd_kernel(float* __restrict__ S, const unsigned char* __restrict__ key_pad,
                                                             long rows, int rows_per_b, int Tk, long ld, float scale,
                                                             float* __restrict__ dropped, float p, uint64_t seed) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  float* s = S + row * ld;
  const unsigned char* kp = key_pad ? key_pad + (row / rows_per_b) * Tk : nullptr;
  float mx = -INFINITY;
  for (int k = lane; k < Tk; k += 64) {
    float v = (kp && kp[k]) ? -INFINITY : s[k] * scale;
    mx = fmaxf(mx, v);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int k = lane; k < Tk; k += 64) {
    float v = (kp && kp[k]) ? 0.f : expf(s[k] * scale - mx);
    s[k] = v;
    sum += v;
  }
  sum = wave_sum_all(sum);
  const float inv = 1.0f / sum;
  if (dropped) {      // attention dropout (nn.MultiheadAttention(dropout=p)) in the same pass: mask index = flat index
    const float ks = 1.0f / (1.0f - p);
    float* dr = dropped + row * ld;
    for (int k = lane; k < Tk; k += 64) {
      const float v = s[k] * inv;
      s[k] = v;
      dr[k] = ft_dropout_keep(seed, row * Tk + k, p) ? v * ks : 0.f;
    }
    for (int k = Tk + lane; k < ld; k += 64) dr[k] = 0.f;
  } else {
    for (int k = lane; k < Tk; k += 64) s[k] *= inv;
  }
  for (int k = Tk + lane; k < ld; k += 64) s[k] = 0.f;       // row padding reads as zeros
}

// dS = scale * P * (dP' - sum_k dP'*P)   (in place on dP); p > 0: dP' = dropout-mask(dP) / (1-p), i.e. dP arrives as the
// gradient of the DROPPED probabilities and the mask is re-derived here
__global__ __launch_bounds__(256) void ft_softmax_bwd_kernel(const float* __restrict__ P, float* __restrict__ dP, long rows,
                                                             int Tk, long ld, float scale, float p, uint64_t seed) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const float* pr = P + row * ld;
  float* d = dP + row * ld;
  const float ks = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  float dot = 0.f;
  for (int k = lane; k < Tk; k += 64) {
    float g = d[k];
    if (p > 0.f) {
      g = ft_dropout_keep(seed, row * Tk + k, p) ? g * ks : 0.f;
      d[k] = g;
    }
    dot += g * pr[k];
  }
  dot = wave_sum_all(dot);
  for (int k = lane; k < Tk; k += 64) d[k] = scale * pr[k] * (d[k] - dot);
  for (int k = Tk + lane; k < ld; k += 64) d[k] = 0.f;
}

// s = x (+ dropout_p(res)) ; y = (s - mean)/sqrt(var + eps) * gamma + beta ; per-row mean / rstd saved.
// p > 0: the residual branch goes through F.dropout on the fly (mask index = flat index of the element, the same
// counter-based mask ft_dropout would apply to `res`), FFTBlock's  norm(src + dropout(src2))  in one pass.
__global__ __launch_bounds__(256) void ft_layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ sum_out,
                                                               float* __restrict__ y, float* __restrict__ mean,
                                                               float* __restrict__ rstd, long rows, int D, float eps,
                                                               float p, uint64_t seed) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const float* xr = x + row * D;
  const float* rr = res ? res + row * D : nullptr;
  const float keep_scale = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  auto elem = [&](int c) {
    float v = xr[c];
    if (rr) {
      float r = rr[c];
      if (p > 0.f) r = ft_dropout_keep(seed, row * D + c, p) ? r * keep_scale : 0.f;
      v += r;
    }
    return v;
  };
  float s1 = 0.f;
  for (int c = lane; c < D; c += 64) {
    float v = elem(c);
    if (sum_out) sum_out[row * D + c] = v;
    s1 += v;
  }
  const float mu = wave_sum_all(s1) / (float)D;
  float s2 = 0.f;
  for (int c = lane; c < D; c += 64) {
    float v = elem(c) - mu;
    s2 += v * v;
  }
  const float rs = 1.0f / sqrtf(wave_sum_all(s2) / (float)D + eps);
  for (int c = lane; c < D; c += 64) {
    float v = elem(c);
    y[row * D + c] = (v - mu) * rs * gamma[c] + beta[c];
  }
  if (lane == 0) {
    mean[row] = mu;
    rstd[row] = rs;
  }
}

// dx = rstd * (g - mean(g) - xhat*mean(g*xhat)), g = dy*gamma ; also t_xhat = dy*xhat (for dgamma via column sums);
// dres (optional) = the gradient of the dropped-out residual branch = mask * dx / (1-p)
__global__ __launch_bounds__(256) void ft_layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ s,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               float* __restrict__ dx, float* __restrict__ dy_xhat,
                                                               float* __restrict__ dres, long rows, int D, float p,
                                                               uint64_t seed) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const float mu = mean[row], rs = rstd[row];
  const float keep_scale = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  float a = 0.f, b = 0.f;
  for (int c = lane; c < D; c += 64) {
    float xh = (s[row * D + c] - mu) * rs;
    float g = dy[row * D + c] * gamma[c];
    a += g;
    b += g * xh;
  }
  a = wave_sum_all(a) / (float)D;
  b = wave_sum_all(b) / (float)D;
  for (int c = lane; c < D; c += 64) {
    float xh = (s[row * D + c] - mu) * rs;
    float d = dy[row * D + c];
    const float g = rs * (d * gamma[c] - a - xh * b);
    dx[row * D + c] = g;
    dy_xhat[row * D + c] = d * xh;
    if (dres) dres[row * D + c] = (p > 0.f && !ft_dropout_keep(seed, row * D + c, p)) ? 0.f : g * keep_scale;
  }
}

// out[b,t,c] = x[b,t,c] + scale[0] * pe[t*D + c]     (PositionalEncoding.forward, common_layers.py:143-145)
__global__ void ft_posenc_fwd_kernel(const float* __restrict__ x, const float* __restrict__ pe,
                                     const float* __restrict__ scale, float* __restrict__ out, int B, int T, int D) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)B * T * D;
  if (i >= total) return;
  int c = (int)(i % D);
  int t = (int)((i / D) % T);
  out[i] = x[i] + scale[0] * pe[(long)t * D + c];
}
// partial[block] = sum dout * pe  -> dscale
__global__ __launch_bounds__(256) void ft_posenc_dscale_partial_kernel(const float* __restrict__ dout,
                                                                       const float* __restrict__ pe, int B, int T, int D,
                                                                       double* __restrict__ partial) {
  __shared__ double red[4];
  long total = (long)B * T * D;
  double acc = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    int c = (int)(i % D);
    int t = (int)((i / D) % T);
    acc += (double)dout[i] * (double)pe[(long)t * D + c];
  }
  acc = ft_wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void ft_sum_partials_kernel(const double* __restrict__ partial, int n, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += partial[i];
  out[0] = (float)s;
}

__global__ void ft_relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx,
                                   long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}

}  // namespace

extern "C" {

int ft_softmax_fwd(float* scores, const unsigned char* key_pad, int B, int nh, int Tq, int Tk, long ld, float scale,
                   float* dropped, float dropout_p, uint64_t dropout_seed, void* stream) {
  FT_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "softmax_fwd: dropout p must be in [0,1)");
  FT_REQUIRE(ld >= Tk, "softmax_fwd: row stride smaller than the row");
  long rows = (long)B * nh * Tq;
  if (rows <= 0 || Tk <= 0) return FT_OK;
  hipLaunchKernelGGL(ft_softmax_fwd_kernel, dim3(ft_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, scores, key_pad,
                     rows, nh * Tq, Tk, ld, scale, dropout_p > 0.f ? dropped : nullptr, dropout_p, dropout_seed);
  return ft_check_launch("softmax_fwd");
}

int ft_softmax_bwd(const float* probs, float* dprobs, int B, int nh, int Tq, int Tk, long ld, float scale, float dropout_p,
                   uint64_t dropout_seed, void* stream) {
  FT_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "softmax_bwd: dropout p must be in [0,1)");
  FT_REQUIRE(ld >= Tk, "softmax_bwd: row stride smaller than the row");
  long rows = (long)B * nh * Tq;
  if (rows <= 0 || Tk <= 0) return FT_OK;
  hipLaunchKernelGGL(ft_softmax_bwd_kernel, dim3(ft_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, probs, dprobs,
                     rows, Tk, ld, scale, dropout_p, dropout_seed);
  return ft_check_launch("softmax_bwd");
}

int ft_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* sum_out,
                     float* y, float* mean, float* rstd, long rows, int D, float eps, float res_dropout_p,
                     uint64_t res_dropout_seed, void* stream) {
  FT_REQUIRE(res_dropout_p >= 0.f && res_dropout_p < 1.f, "layernorm_fwd: dropout p must be in [0,1)");
  if (rows <= 0 || D <= 0) return FT_OK;
  hipLaunchKernelGGL(ft_layernorm_fwd_kernel, dim3(ft_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, res, gamma,
                     beta, sum_out, y, mean, rstd, rows, D, eps, res_dropout_p, res_dropout_seed);
  return ft_check_launch("layernorm_fwd");
}

int ft_layernorm_bwd(const float* dy, const float* s, const float* gamma, const float* mean, const float* rstd,
                     float* dx, float* dy_xhat, float* dres, long rows, int D, float res_dropout_p,
                     uint64_t res_dropout_seed, void* stream) {
  FT_REQUIRE(res_dropout_p >= 0.f && res_dropout_p < 1.f, "layernorm_bwd: dropout p must be in [0,1)");
  if (rows <= 0 || D <= 0) return FT_OK;
  hipLaunchKernelGGL(ft_layernorm_bwd_kernel, dim3(ft_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, dy, s, gamma,
                     mean, rstd, dx, dy_xhat, dres, rows, D, res_dropout_p, res_dropout_seed);
  return ft_check_launch("layernorm_bwd");
}

int ft_posenc_fwd(const float* x, const float* pe, const float* scale, float* out, int B, int T, int D,
                  void* stream) {
  long total = (long)B * T * D;
  if (total <= 0) return FT_OK;
  hipLaunchKernelGGL(ft_posenc_fwd_kernel, dim3(ft_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, pe, scale,
                     out, B, T, D);
  return ft_check_launch("posenc_fwd");
}

size_t ft_posenc_workspace(void) { return 512 * sizeof(double); }

int ft_posenc_bwd_scale(const float* dout, const float* pe, float* dscale, int B, int T, int D, void* workspace,
                        size_t workspace_bytes, void* stream) {
  FT_REQUIRE(workspace && workspace_bytes >= ft_posenc_workspace(), "posenc_bwd_scale: workspace too small");
  long total = (long)B * T * D;
  int nb = ft_cdiv(total, 256 * 16);
  if (nb > 512) nb = 512;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(ft_posenc_dscale_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, dout, pe, B, T, D,
                     (double*)workspace);
  hipLaunchKernelGGL(ft_sum_partials_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double*)workspace, nb,
                     dscale);
  return ft_check_launch("posenc_bwd_scale");
}

int ft_relu_bwd(const float* dy, const float* y, float* dx, long n, void* stream) {
  if (n <= 0) return FT_OK;
  hipLaunchKernelGGL(ft_relu_bwd_kernel, dim3(ft_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n);
  return ft_check_launch("relu_bwd");
}

}  // extern "C"
