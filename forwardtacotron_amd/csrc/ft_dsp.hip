// Mel inversion + Griffin-Lim on the GPU (reference: utils/dsp.py:80-94 DSP.griffinlim, called from gen_forward.py:109-116).
// The transforms themselves are GEMMs: a 1024-point real DFT of N windowed frames is frames[N,1024] x basis^T on the
// MFMA GEMM kernels (ft_linear_fwd; the frames are read IN PLACE out of the zero-padded signal with a row stride of
// hop samples), the inverse is proj[N,2F] x inverse-basis^T.  What is left for this file is element-wise / gather work,
// all HBM-bound:
//   ft_exp_transpose : log-mel [C,T] -> exp -> [T,C]                     (DSP.denormalize + layout for the GEMMs)
//   ft_nnls_step     : X = max(0, X - G / L)                             (projected-gradient step of the mel inversion)
//   ft_sub           : out = a - b
//   ft_gl_init       : proj = S * exp(2 pi i u)                          (random initial phases, u drawn by the host)
//   ft_gl_phase      : c = R - alpha * R_prev ; proj = S * c / (|c| + tiny) ; R_prev = R     (fast Griffin-Lim update)
//   ft_overlap_add   : y_pad[t] = sum_n frames[n][t - n*hop] * inv_wss[t]  inside [n_fft/2, n_fft/2 + L), 0 in the
//                      margins (gather form: every sample sums its <= n_fft/hop frames in a fixed order)
// Complex spectra are stored split: [N][2*Fp] = Re (Fp columns) | Im (Fp columns), Fp = F rounded up to 4.
#include <math.h>

#include "ft_common.h"

namespace {

__global__ __launch_bounds__(256) void ft_exp_transpose_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                               int C, int T) {
  __shared__ float tile[32][33];
  const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int c = c0 + i, t = t0 + threadIdx.x;
    tile[i][threadIdx.x] = (c < C && t < T) ? expf(in[(long)c * T + t]) : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int t = t0 + i, c = c0 + threadIdx.x;
    if (t < T && c < C) out[(long)t * C + c] = tile[threadIdx.x][i];
  }
}

__global__ __launch_bounds__(256) void ft_nnls_step_kernel(float* __restrict__ x, const float* __restrict__ g,
                                                           float inv_l, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] = fmaxf(x[i] - inv_l * g[i], 0.f);
}

__global__ __launch_bounds__(256) void ft_sub_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     float* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = a[i] - b[i];
}

__global__ __launch_bounds__(256) void ft_gl_init_kernel(const float* __restrict__ u, const float* __restrict__ S,
                                                         float* __restrict__ proj, int N, int Fp) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)N * Fp) return;
  const long n = i / Fp;
  const int m = (int)(i - n * Fp);
  float sn, cs;
  sincosf(6.283185307179586f * u[i], &sn, &cs);
  const float s = S[i];
  proj[n * 2 * Fp + m] = s * cs;
  proj[n * 2 * Fp + Fp + m] = s * sn;
}

__global__ __launch_bounds__(256) void ft_gl_phase_kernel(const float* __restrict__ rebuilt, float* __restrict__ tprev,
                                                          const float* __restrict__ S, float* __restrict__ proj, int N,
                                                          int Fp, float alpha, int has_prev) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)N * Fp) return;
  const long n = i / Fp;
  const int m = (int)(i - n * Fp);
  const long ire = n * 2 * Fp + m, iim = ire + Fp;
  const float re = rebuilt[ire], im = rebuilt[iim];
  float cr = re, cim = im;
  if (has_prev) {
    cr -= alpha * tprev[ire];
    cim -= alpha * tprev[iim];
  }
  const float inv = 1.0f / (sqrtf(cr * cr + cim * cim) + 1.17549435e-38f);
  const float s = S[i];
  proj[ire] = s * cr * inv;
  proj[iim] = s * cim * inv;
  tprev[ire] = re;
  tprev[iim] = im;
}

__global__ __launch_bounds__(256) void ft_overlap_add_kernel(const float* __restrict__ frames,
                                                             const float* __restrict__ inv_wss, float* __restrict__ ypad,
                                                             int N, int n_fft, int hop) {
  const long total = (long)n_fft + (long)hop * (N - 1);
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const long pad = n_fft / 2;
  float acc = 0.f;
  if (t >= pad && t < total - pad) {
    // frames n with 0 <= t - n*hop < n_fft, ascending n (fixed order)
    long n_lo = (t - n_fft + hop) / hop;            // ceil((t - n_fft + 1) / hop) for t - n_fft + 1 > 0
    if (t - n_fft + 1 <= 0) n_lo = 0;
    long n_hi = t / hop;
    if (n_hi > N - 1) n_hi = N - 1;
    for (long n = n_lo; n <= n_hi; ++n) acc += frames[n * n_fft + (t - n * hop)];
    acc *= inv_wss[t];
  }
  ypad[t] = acc;
}

}  // namespace

extern "C" {

int ft_exp_transpose(const float* mel_log, float* out, int C, int T, void* stream) {
  FT_REQUIRE(C >= 0 && T >= 0, "exp_transpose: bad dims");
  if (C == 0 || T == 0) return FT_OK;
  hipLaunchKernelGGL(ft_exp_transpose_kernel, dim3(ft_cdiv(T, 32), ft_cdiv(C, 32)), dim3(32, 8), 0, (hipStream_t)stream,
                     mel_log, out, C, T);
  return ft_check_launch("exp_transpose");
}

int ft_nnls_step(float* x, const float* g, float inv_l, long n, void* stream) {
  if (n <= 0) return FT_OK;
  hipLaunchKernelGGL(ft_nnls_step_kernel, dim3(ft_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, g, inv_l, n);
  return ft_check_launch("nnls_step");
}

int ft_sub(const float* a, const float* b, float* out, long n, void* stream) {
  if (n <= 0) return FT_OK;
  hipLaunchKernelGGL(ft_sub_kernel, dim3(ft_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
  return ft_check_launch("sub");
}

int ft_gl_init(const float* u, const float* S, float* proj, int N, int Fp, void* stream) {
  FT_REQUIRE(N >= 0 && Fp >= 0, "gl_init: bad dims");
  if ((long)N * Fp == 0) return FT_OK;
  hipLaunchKernelGGL(ft_gl_init_kernel, dim3(ft_cdiv((long)N * Fp, 256)), dim3(256), 0, (hipStream_t)stream, u, S, proj,
                     N, Fp);
  return ft_check_launch("gl_init");
}

int ft_gl_phase(const float* rebuilt, float* tprev, const float* S, float* proj, int N, int Fp, float alpha,
                int has_prev, void* stream) {
  FT_REQUIRE(N >= 0 && Fp >= 0, "gl_phase: bad dims");
  if ((long)N * Fp == 0) return FT_OK;
  hipLaunchKernelGGL(ft_gl_phase_kernel, dim3(ft_cdiv((long)N * Fp, 256)), dim3(256), 0, (hipStream_t)stream, rebuilt,
                     tprev, S, proj, N, Fp, alpha, has_prev);
  return ft_check_launch("gl_phase");
}

int ft_overlap_add(const float* frames, const float* inv_wss, float* ypad, int N, int n_fft, int hop, void* stream) {
  FT_REQUIRE(N >= 1 && n_fft >= 2 && hop >= 1 && hop <= n_fft, "overlap_add: bad dims");
  const long total = (long)n_fft + (long)hop * (N - 1);
  hipLaunchKernelGGL(ft_overlap_add_kernel, dim3(ft_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, frames,
                     inv_wss, ypad, N, n_fft, hop);
  return ft_check_launch("overlap_add");
}

}  // extern "C"
