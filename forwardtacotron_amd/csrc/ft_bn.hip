// BatchNorm1d over channels-last activations (models/common_layers.py:51,57) + column reductions.
//
// Training statistics follow torch.nn.BatchNorm1d exactly: mean / biased variance over ALL B*T positions
// (padding included), running stats updated with momentum and the UNBIASED variance, counter += 1.
// CBHG quirk (common_layers.py:97-99): an even-k bank conv yields T+1 outputs and its BatchNorm sees all
// of them before the caller slices [:T].  The bank buffer is therefore [B, T+1, K*C] and the number of
// valid rows depends on the channel group: tvalid(c) = Tbuf-1 for odd k, Tbuf for even k (group > 0).
//
// All reductions are two-stage (per-block double partials -> ordered finalize), so results are bitwise
// reproducible run to run.  These kernels are HBM-bound: one pass over the activation per stage.
#include <string.h>

#include "ft_common.h"

namespace {

__device__ __forceinline__ int tvalid_of(int c, int Tbuf, int group) {
  return (group > 0 && (((c / group) + 1) & 1)) ? Tbuf - 1 : Tbuf;
}

// ---- BatchNorm apply fused with MaxPool1d(2,1,1)[:T] (CBHG bank: common_layers.py:100-105) -----------------------------
// z[t] = (y[t] - mean) * rstd * gamma + beta is never stored: the forward writes out[t] = max(z[t-1], z[t]) straight
// from y, and the backward recomputes the three z rows a pooling gradient needs with the SAME expression (so the
// argmax decisions are bit-identical to the forward's):
//   dz[t] = dout[t] * [t == 0 or z[t] > z[t-1]]  +  dout[t+1] * [t+1 < T and not z[t+1] > z[t]]
// (torch sends a window's gradient to its FIRST maximal element: after the ReLU a bank row is full of equal zeros).
struct BnAff4 { float4 mu, rs, ga, be; };
__device__ __forceinline__ float4 bn_z4(const float4& v, const BnAff4& a) {
  float4 o;
  o.x = (v.x - a.mu.x) * a.rs.x * a.ga.x + a.be.x;
  o.y = (v.y - a.mu.y) * a.rs.y * a.ga.y + a.be.y;
  o.z = (v.z - a.mu.z) * a.rs.z * a.ga.z + a.be.z;
  o.w = (v.w - a.mu.w) * a.rs.w * a.ga.w + a.be.w;
  return o;
}
// dz of row (b, t), t < Tout, 4 channels; yrow = &y[(b*Tbuf + t)*C + c], drow = &dout[(b*Tout + t)*C + c]; vt = y[t]
__device__ __forceinline__ float4 pool_grad4(const float* __restrict__ yrow, const float* __restrict__ drow,
                                             const float4& vt, int t, int Tout, int C, const BnAff4& a) {
  const float4 z = bn_z4(vt, a);
  const float4 d0 = *reinterpret_cast<const float4*>(drow);
  float4 g;
  if (t == 0) {
    g = d0;
  } else {
    const float4 p = bn_z4(*reinterpret_cast<const float4*>(yrow - C), a);
    g.x = z.x > p.x ? d0.x : 0.f;
    g.y = z.y > p.y ? d0.y : 0.f;
    g.z = z.z > p.z ? d0.z : 0.f;
    g.w = z.w > p.w ? d0.w : 0.f;
  }
  if (t + 1 < Tout) {
    const float4 n = bn_z4(*reinterpret_cast<const float4*>(yrow + C), a);
    const float4 d1 = *reinterpret_cast<const float4*>(drow + C);
    g.x += !(n.x > z.x) ? d1.x : 0.f;
    g.y += !(n.y > z.y) ? d1.y : 0.f;
    g.z += !(n.z > z.z) ? d1.z : 0.f;
    g.w += !(n.w > z.w) ? d1.w : 0.f;
  }
  return g;
}

// mode 0: (sum y, sum y^2) over rows t < tvalid(c) of y[B,Tbuf,C]
// mode 1: (sum dout, sum dout*xhat), dout[B,Tout,C] (zero for t >= Tout), xhat from y, mean, rstd
// mode 2: (sum x, 0) over all rows of x[rows, ldx] (bias gradients); Tbuf = rows, B = 1
template <int MODE>
__global__ __launch_bounds__(256) void ft_col_partial_kernel(const float* __restrict__ y, long ldy,
                                                             const float* __restrict__ dout,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, int B, int Tbuf, int Tout,
                                                             int C, int group, int rows_per_chunk,
                                                             double* __restrict__ partial) {
  __shared__ double red[2][4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const long rows = (long)B * Tbuf;
  long r0 = (long)blockIdx.y * rows_per_chunk;
  long r1 = r0 + rows_per_chunk;
  if (r1 > rows) r1 = rows;
  double s0 = 0.0, s1 = 0.0;
  if (c < C) {
    const int tv = tvalid_of(c, Tbuf, group);
    float mu = 0.f, rs = 0.f;
    if (MODE == 1) {
      mu = mean[c];
      rs = rstd[c];
    }
    long r = r0 + rl;
    int b = (int)(r / Tbuf);
    int t = (int)(r - (long)b * Tbuf);
    for (; r < r1; r += 4) {
      if (t < tv) {
        float v = y[r * ldy + c];
        if (MODE == 0) {
          s0 += (double)v;
          s1 += (double)v * (double)v;
        } else if (MODE == 1) {
          if (t < Tout) {
            float g = dout[((long)b * Tout + t) * C + c];
            s0 += (double)g;
            s1 += (double)g * (double)((v - mu) * rs);
          }
        } else {
          s0 += (double)v;
        }
      }
      t += 4;
      while (t >= Tbuf) {
        t -= Tbuf;
        ++b;
      }
    }
  }
  red[0][rl][cl] = s0;
  red[1][rl][cl] = s1;
  __syncthreads();
  if (rl == 0 && c < C) {
    double a0 = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
    double a1 = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
    partial[((long)blockIdx.y * C + c) * 2 + 0] = a0;
    partial[((long)blockIdx.y * C + c) * 2 + 1] = a1;
  }
}

// mode 2 with 16-B lanes (C % 4 == 0, 16-B aligned rows): 16 row lanes x 16 column quads per block, four rows in
// flight per lane; the 16 row-lane sums are combined in fixed order
__global__ __launch_bounds__(256) void ft_colsum4_partial_kernel(const float* __restrict__ x, long ldx, long rows, int C,
                                                                 int rows_per_chunk, double* __restrict__ partial,
                                                                 const float* __restrict__ x1) {
  if (x1) {      // second matrix of the same shape -> slot 1 (block-uniform: blockIdx.z selects the matrix)
    if (blockIdx.z == 1) x = x1;
  }
  const int slot = x1 ? (int)blockIdx.z : 0;
  __shared__ double red[16][65];
  const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + cq * 4;
  const long r0 = (long)blockIdx.y * rows_per_chunk;
  long r1 = r0 + rows_per_chunk;
  if (r1 > rows) r1 = rows;
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  if (c < C) {
    for (long r = r0 + rl; r < r1; r += 64) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long rr = r + 16 * u;
        v[u] = rr < r1 ? *reinterpret_cast<const float4*>(x + rr * ldx + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s[0] += (double)v[u].x;
        s[1] += (double)v[u].y;
        s[2] += (double)v[u].z;
        s[3] += (double)v[u].w;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[rl][cq * 4 + e] = s[e];
  __syncthreads();
  if (threadIdx.x < 64) {
    const int cc = blockIdx.x * 64 + threadIdx.x;
    if (cc < C) {
      double a = 0.0;
#pragma unroll
      for (int i = 0; i < 16; ++i) a += red[i][threadIdx.x];
      partial[((long)blockIdx.y * C + cc) * 2 + slot] = a;
      if (!x1) partial[((long)blockIdx.y * C + cc) * 2 + 1] = 0.0;
    }
  }
}

// modes 0 / 1 with 16-B lanes (C % 4 == 0, group % 4 == 0, 16-B aligned buffers): same block shape as the kernel above
template <int MODE>
__global__ __launch_bounds__(256) void ft_col_partial4_kernel(const float* __restrict__ y, const float* __restrict__ dout,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, int B, int Tbuf, int Tout,
                                                              int C, int group, int rows_per_chunk,
                                                              double* __restrict__ partial,
                                                              const float* __restrict__ gamma = nullptr,
                                                              const float* __restrict__ beta = nullptr) {
  // MODE 2 = MODE 1 with dout being the gradient of the POOLED output: dz is recomputed per element (pool_grad4)
  __shared__ double red[2][16][65];
  const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + cq * 4;
  const long rows = (long)B * Tbuf;
  const long r0 = (long)blockIdx.y * rows_per_chunk;
  long r1 = r0 + rows_per_chunk;
  if (r1 > rows) r1 = rows;
  double s0[4] = {0.0, 0.0, 0.0, 0.0}, s1[4] = {0.0, 0.0, 0.0, 0.0};
  if (c < C) {
    const int tv = tvalid_of(c, Tbuf, group);
    float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), rs = mu;
    BnAff4 aff;
    if (MODE >= 1) {
      mu = *reinterpret_cast<const float4*>(mean + c);
      rs = *reinterpret_cast<const float4*>(rstd + c);
    }
    if (MODE == 2) {
      aff.mu = mu;
      aff.rs = rs;
      aff.ga = *reinterpret_cast<const float4*>(gamma + c);
      aff.be = *reinterpret_cast<const float4*>(beta + c);
    }
    int bb[4], tt[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long r = r0 + rl + 16 * u;
      bb[u] = (int)(r / Tbuf);
      tt[u] = (int)(r - (long)bb[u] * Tbuf);
    }
    for (long r = r0 + rl; r < r1; r += 64) {
      float4 v[4], g[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long rr = r + 16 * u;
        ok[u] = rr < r1 && tt[u] < tv && (MODE == 0 || tt[u] < Tout);
        v[u] = ok[u] ? *reinterpret_cast<const float4*>(y + rr * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE == 1)
          g[u] = ok[u] ? *reinterpret_cast<const float4*>(dout + ((long)bb[u] * Tout + tt[u]) * C + c)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE == 2)
          g[u] = ok[u] ? pool_grad4(y + rr * C + c, dout + ((long)bb[u] * Tout + tt[u]) * C + c, v[u], tt[u], Tout, C, aff)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (MODE == 0) {
          s0[0] += (double)v[u].x; s1[0] += (double)v[u].x * (double)v[u].x;
          s0[1] += (double)v[u].y; s1[1] += (double)v[u].y * (double)v[u].y;
          s0[2] += (double)v[u].z; s1[2] += (double)v[u].z * (double)v[u].z;
          s0[3] += (double)v[u].w; s1[3] += (double)v[u].w * (double)v[u].w;
        } else if (ok[u]) {
          s0[0] += (double)g[u].x; s1[0] += (double)g[u].x * (double)((v[u].x - mu.x) * rs.x);
          s0[1] += (double)g[u].y; s1[1] += (double)g[u].y * (double)((v[u].y - mu.y) * rs.y);
          s0[2] += (double)g[u].z; s1[2] += (double)g[u].z * (double)((v[u].z - mu.z) * rs.z);
          s0[3] += (double)g[u].w; s1[3] += (double)g[u].w * (double)((v[u].w - mu.w) * rs.w);
        }
        tt[u] += 64;
        while (tt[u] >= Tbuf) {
          tt[u] -= Tbuf;
          ++bb[u];
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[0][rl][cq * 4 + e] = s0[e];
    red[1][rl][cq * 4 + e] = s1[e];
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int cc = blockIdx.x * 64 + threadIdx.x;
    if (cc < C) {
      double a0 = 0.0, a1 = 0.0;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        a0 += red[0][i][threadIdx.x];
        a1 += red[1][i][threadIdx.x];
      }
      partial[((long)blockIdx.y * C + cc) * 2 + 0] = a0;
      partial[((long)blockIdx.y * C + cc) * 2 + 1] = a1;
    }
  }
}

// ordered two-level sum of the chunk partials of column c: 8 lanes take contiguous chunk ranges, lane 0 of the group
// adds the 8 range sums in range order (deterministic; 8x shorter dependent chain than one serial loop)
constexpr int FIN_PARTS = 8;
__device__ __forceinline__ void sum_partials(const double* __restrict__ partial, int nchunks, int C, int c, int part,
                                             double (*sh)[FIN_PARTS][2], int slot, double& s0, double& s1) {
  double a0 = 0.0, a1 = 0.0;
  if (c < C) {
    const int per = (nchunks + FIN_PARTS - 1) / FIN_PARTS;
    const int i0 = part * per, i1 = min(nchunks, i0 + per);
    for (int i = i0; i < i1; ++i) {
      a0 += partial[((long)i * C + c) * 2 + 0];
      a1 += partial[((long)i * C + c) * 2 + 1];
    }
  }
  sh[slot][part][0] = a0;
  sh[slot][part][1] = a1;
  __syncthreads();
  s0 = 0.0;
  s1 = 0.0;
  if (part == 0) {
#pragma unroll
    for (int p = 0; p < FIN_PARTS; ++p) {
      s0 += sh[slot][p][0];
      s1 += sh[slot][p][1];
    }
  }
}

// block = 32 columns x 8 parts
__global__ __launch_bounds__(256) void ft_bn_finalize_kernel(const double* __restrict__ partial, int nchunks, int B,
                                                             int Tbuf, int C, int group, float momentum, float eps,
                                                             float* __restrict__ running_mean,
                                                             float* __restrict__ running_var, long* __restrict__ nbt,
                                                             float* __restrict__ save_mean,
                                                             float* __restrict__ save_rstd) {
  __shared__ double sh[32][FIN_PARTS][2];
  const int slot = threadIdx.x >> 3, part = threadIdx.x & 7;
  const int c = blockIdx.x * 32 + slot;
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
  double s0, s1;
  sum_partials(partial, nchunks, C, c, part, sh, slot, s0, s1);
  if (part != 0 || c >= C) return;
  double n = (double)B * (double)tvalid_of(c, Tbuf, group);
  double mu = s0 / n;
  double var = s1 / n - mu * mu;
  if (var < 0.0) var = 0.0;
  save_mean[c] = (float)mu;
  save_rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    double unb = n > 1.0 ? var * n / (n - 1.0) : var;
    running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mu);
    running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unb);
  }
}

// finalize of modes 1 / 2: out0[c] = sum0, out1[c] = sum1 (floats)
__global__ __launch_bounds__(256) void ft_col_finalize_kernel(const double* __restrict__ partial, int nchunks, int C,
                                                              float* __restrict__ out0, float* __restrict__ out1,
                                                              float scale0, int accumulate) {
  __shared__ double sh[32][FIN_PARTS][2];
  const int slot = threadIdx.x >> 3, part = threadIdx.x & 7;
  const int c = blockIdx.x * 32 + slot;
  double s0, s1;
  sum_partials(partial, nchunks, C, c, part, sh, slot, s0, s1);
  if (part != 0 || c >= C) return;
  if (out0) out0[c] = (accumulate ? out0[c] : 0.f) + (float)(s0 * scale0);
  if (out1) out1[c] = (accumulate ? out1[c] : 0.f) + (float)s1;
}

// out[b,t,c] = (y[b,t,c]-mean)*rstd*gamma + beta (+ residual[b,t,c])   for t < Tout
__global__ __launch_bounds__(256) void ft_bn_apply_kernel(const float* __restrict__ y, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          const float* __restrict__ residual, float* __restrict__ out,
                                                          int B, int Tbuf, int Tout, int C) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)B * Tout * C;
  if (idx >= total) return;
  int c = (int)(idx % C);
  long row = idx / C;
  int b = (int)(row / Tout), t = (int)(row - (long)b * Tout);
  float v = y[((long)b * Tbuf + t) * C + c];
  float o = (v - mean[c]) * rstd[c] * gamma[c] + beta[c];
  if (residual) o += residual[idx];
  out[idx] = o;
}

// dy[b,t,c] = gamma*rstd*(dout - dbeta/n - xhat*dgamma/n) [* (y>0)]  for t < tvalid(c), else 0
__global__ __launch_bounds__(256) void ft_bn_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ y,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ rstd,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ dgamma,
                                                              const float* __restrict__ dbeta, float* __restrict__ dy,
                                                              int B, int Tbuf, int Tout, int C, int group, int relu) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)B * Tbuf * C;
  if (idx >= total) return;
  int c = (int)(idx % C);
  long row = idx / C;
  int b = (int)(row / Tbuf), t = (int)(row - (long)b * Tbuf);
  int tv = tvalid_of(c, Tbuf, group);
  float r = 0.f;
  if (t < tv) {
    float v = y[idx];
    float g = t < Tout ? dout[((long)b * Tout + t) * C + c] : 0.f;
    float inv_n = 1.0f / ((float)B * (float)tv);
    float xh = (v - mean[c]) * rstd[c];
    r = gamma[c] * rstd[c] * (g - dbeta[c] * inv_n - xh * dgamma[c] * inv_n);
    if (relu && !(v > 0.f)) r = 0.f;
  }
  dy[idx] = r;
}

// 16-B-lane forms of the two kernels above (C % 4 == 0, group % 4 == 0, 16-B aligned buffers): one thread = 4 channels
__global__ __launch_bounds__(256) void ft_bn_apply4_kernel(const float4* __restrict__ y, const float4* __restrict__ mean,
                                                           const float4* __restrict__ rstd,
                                                           const float4* __restrict__ gamma,
                                                           const float4* __restrict__ beta,
                                                           const float4* __restrict__ residual, float4* __restrict__ out,
                                                           long total4, int Tbuf, int Tout, int C4) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total4) return;
  const int c = (int)(idx % C4);
  const long row = idx / C4;
  const int b = (int)(row / Tout), t = (int)(row - (long)b * Tout);
  const float4 v = y[((long)b * Tbuf + t) * C4 + c];
  const float4 mu = mean[c], rs = rstd[c], ga = gamma[c], be = beta[c];
  float4 o;
  o.x = (v.x - mu.x) * rs.x * ga.x + be.x;
  o.y = (v.y - mu.y) * rs.y * ga.y + be.y;
  o.z = (v.z - mu.z) * rs.z * ga.z + be.z;
  o.w = (v.w - mu.w) * rs.w * ga.w + be.w;
  if (residual) {
    const float4 r = residual[idx];
    o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
  }
  out[idx] = o;
}

// out[b,t] = max(z[b,t-1], z[b,t]) (z[-1] = -inf), z never stored
__global__ __launch_bounds__(256) void ft_bn_apply_pool4_kernel(const float* __restrict__ y, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float4* __restrict__ out,
                                                                long total4, int Tbuf, int Tout, int C) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total4) return;
  const int C4 = C / 4;
  const int c = (int)(idx % C4) * 4;
  const long row = idx / C4;
  const int b = (int)(row / Tout), t = (int)(row - (long)b * Tout);
  BnAff4 a;
  a.mu = *reinterpret_cast<const float4*>(mean + c);
  a.rs = *reinterpret_cast<const float4*>(rstd + c);
  a.ga = *reinterpret_cast<const float4*>(gamma + c);
  a.be = *reinterpret_cast<const float4*>(beta + c);
  const float* yrow = y + ((long)b * Tbuf + t) * C + c;
  float4 z = bn_z4(*reinterpret_cast<const float4*>(yrow), a);
  if (t > 0) {
    const float4 p = bn_z4(*reinterpret_cast<const float4*>(yrow - C), a);
    z.x = z.x > p.x ? z.x : p.x;
    z.y = z.y > p.y ? z.y : p.y;
    z.z = z.z > p.z ? z.z : p.z;
    z.w = z.w > p.w ? z.w : p.w;
  }
  out[idx] = z;
}

template <bool POOL>
__global__ __launch_bounds__(256) void ft_bn_bwd_apply4_kernel(const float4* __restrict__ dout, const float4* __restrict__ y,
                                                               const float4* __restrict__ mean,
                                                               const float4* __restrict__ rstd,
                                                               const float4* __restrict__ gamma,
                                                               const float4* __restrict__ dgamma,
                                                               const float4* __restrict__ dbeta, float4* __restrict__ dy,
                                                               long total4, int B, int Tbuf, int Tout, int C4, int group,
                                                               int relu, const float4* __restrict__ beta = nullptr) {
  // POOL: dout is the gradient of the pooled output, the BatchNorm output's gradient is recomputed (pool_grad4)
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total4) return;
  const int c = (int)(idx % C4);
  const long row = idx / C4;
  const int b = (int)(row / Tbuf), t = (int)(row - (long)b * Tbuf);
  const int tv = tvalid_of(4 * c, Tbuf, group);
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (t < tv) {
    const float4 v = y[idx];
    const float4 mu = mean[c], rs = rstd[c], ga = gamma[c], dg = dgamma[c], db = dbeta[c];
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < Tout) {
      if constexpr (POOL) {
        BnAff4 a;
        a.mu = mu; a.rs = rs; a.ga = ga; a.be = beta[c];
        g = pool_grad4(reinterpret_cast<const float*>(y + idx), reinterpret_cast<const float*>(dout + ((long)b * Tout + t) * C4 + c),
                       v, t, Tout, 4 * C4, a);
      } else {
        g = dout[((long)b * Tout + t) * C4 + c];
      }
    }
    const float inv_n = 1.0f / ((float)B * (float)tv);
#define FT_BN_BWD1(f)                                                                \
  {                                                                                  \
    const float xh = (v.f - mu.f) * rs.f;                                            \
    r.f = ga.f * rs.f * (g.f - db.f * inv_n - xh * dg.f * inv_n);                    \
    if (relu && !(v.f > 0.f)) r.f = 0.f;                                             \
  }
    FT_BN_BWD1(x) FT_BN_BWD1(y) FT_BN_BWD1(z) FT_BN_BWD1(w)
#undef FT_BN_BWD1
  }
  dy[idx] = r;
}

// eval-mode fold: scale = gamma/sqrt(rv+eps), shift = beta - rm*scale
__global__ void ft_bn_fold_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                  float* scale, float* shift, int C) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = gamma[c] / sqrtf(rv[c] + eps);
  scale[c] = s;
  shift[c] = beta[c] - rm[c] * s;
}

template <typename... P>
bool all16(P... ptrs) {
  return (((uintptr_t)(const void*)ptrs % 16 == 0) && ...);
}

// ---- several column sums over matrices with the SAME row count in one partial + one finalize launch -----------------
// (a FastPitch step computed 255 bias / LayerNorm gradients as 255 pairs of 5-8 us launches: 1.7 ms of kernel time).
// Every task keeps the chunking and the summation order of its stand-alone ft_colsum call: bit-identical results.
constexpr int CSB_MAX = 16;
struct ColsumBatch {
  const float* x[CSB_MAX];
  float* out[CSB_MAX];
  long ld[CSB_MAX], ws_off[CSB_MAX];       // ws_off in doubles
  int C[CSB_MAX], nchunks[CSB_MAX], rpc[CSB_MAX];
  int tile64[CSB_MAX + 1], tile32[CSB_MAX + 1];   // prefix sums of the tasks' 64- / 32-column tiles
  int n;
  long rows;
};

__global__ __launch_bounds__(256) void ft_colsum_batch_partial_kernel(ColsumBatch b, double* __restrict__ ws) {
  int t = 0;
  while (t + 1 < b.n && (int)blockIdx.x >= b.tile64[t + 1]) ++t;
  if ((int)blockIdx.y >= b.nchunks[t]) return;
  const float* __restrict__ x = b.x[t];
  const long ldx = b.ld[t];
  const int C = b.C[t];
  double* __restrict__ partial = ws + b.ws_off[t];
  __shared__ double red[16][65];
  const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int tile = blockIdx.x - b.tile64[t];
  const int c = tile * 64 + cq * 4;
  const long r0 = (long)blockIdx.y * b.rpc[t];
  long r1 = r0 + b.rpc[t];
  if (r1 > b.rows) r1 = b.rows;
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  if (c < C) {
    for (long r = r0 + rl; r < r1; r += 64) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long rr = r + 16 * u;
        v[u] = rr < r1 ? *reinterpret_cast<const float4*>(x + rr * ldx + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s[0] += (double)v[u].x;
        s[1] += (double)v[u].y;
        s[2] += (double)v[u].z;
        s[3] += (double)v[u].w;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[rl][cq * 4 + e] = s[e];
  __syncthreads();
  if (threadIdx.x < 64) {
    const int cc = tile * 64 + threadIdx.x;
    if (cc < C) {
      double a = 0.0;
#pragma unroll
      for (int i = 0; i < 16; ++i) a += red[i][threadIdx.x];
      partial[((long)blockIdx.y * C + cc) * 2] = a;
      partial[((long)blockIdx.y * C + cc) * 2 + 1] = 0.0;
    }
  }
}

__global__ __launch_bounds__(256) void ft_colsum_batch_finalize_kernel(ColsumBatch b, const double* __restrict__ ws) {
  int t = 0;
  while (t + 1 < b.n && (int)blockIdx.x >= b.tile32[t + 1]) ++t;
  __shared__ double sh[32][FIN_PARTS][2];
  const int slot = threadIdx.x >> 3, part = threadIdx.x & 7;
  const int c = (blockIdx.x - b.tile32[t]) * 32 + slot;
  double s0, s1;
  sum_partials(ws + b.ws_off[t], b.nchunks[t], b.C[t], c, part, sh, slot, s0, s1);
  if (part != 0 || c >= b.C[t]) return;
  b.out[t][c] = (float)s0;
}

struct ChunkPlan {
  int nchunks, rows_per_chunk;
};
ChunkPlan plan_chunks(long rows, int C) {
  int cb = ft_cdiv(C, 64);
  // enough blocks to fill the chip, few enough chunks that the per-channel ordered finalize stays short
  long want = 1024 / (cb > 0 ? cb : 1);
  if (want < 16) want = 16;
  if (want > 64) want = 64;
  long maxc = (rows + 63) / 64;
  if (maxc < 1) maxc = 1;
  long n = want < maxc ? want : maxc;
  if (n > 65535) n = 65535;
  long rpc = (rows + n - 1) / n;
  rpc = ((rpc + 3) / 4) * 4;
  if (rpc < 4) rpc = 4;
  n = (rows + rpc - 1) / rpc;
  if (n < 1) n = 1;
  ChunkPlan p = {(int)n, (int)rpc};
  return p;
}

}  // namespace

// (sum, sum of squares) partials of y[B,Tbuf,C] by the stand-alone pass; returns the number of chunks written
int ft_bn_stat_partials(const float* y, int B, int Tbuf, int C, int group, double* partial, hipStream_t s) {
  ChunkPlan p = plan_chunks((long)B * Tbuf, C);
  if (C % 4 == 0 && (group == 0 || group % 4 == 0) && all16(y))
    hipLaunchKernelGGL(ft_col_partial4_kernel<0>, dim3(ft_cdiv(C, 64), p.nchunks), dim3(256), 0, s, y, nullptr, nullptr,
                       nullptr, B, Tbuf, Tbuf, C, group, p.rows_per_chunk, partial);
  else
    hipLaunchKernelGGL(ft_col_partial_kernel<0>, dim3(ft_cdiv(C, 64), p.nchunks), dim3(256), 0, s, y, (long)C, nullptr,
                       nullptr, nullptr, B, Tbuf, Tbuf, C, group, p.rows_per_chunk, partial);
  return p.nchunks;
}

extern "C" {

size_t ft_bn_workspace(int B, int Tbuf, int C);

size_t ft_conv_stats_workspace(int B, int Tbuf, int C) {
  const size_t tiles = (size_t)ft_cdiv((long)B * Tbuf, 128) * C * 2 * sizeof(double);      // one partial per 128-row GEMM tile
  const size_t chunks = ft_bn_workspace(B, Tbuf, C);
  return tiles > chunks ? tiles : chunks;
}

int ft_bn_train_from_partials(const double* partial, int nchunks, const float* y, const float* gamma, const float* beta,
                              const float* residual, float* out, float* running_mean, float* running_var,
                              long* num_batches_tracked, float* save_mean, float* save_rstd, int B, int Tbuf, int Tout,
                              int C, int group, float momentum, float eps, void* stream) {
  FT_REQUIRE(B > 0 && Tbuf > 0 && C > 0 && Tout <= Tbuf && nchunks >= 1, "bn_train_from_partials: bad dims");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(ft_bn_finalize_kernel, dim3(ft_cdiv(C, 32)), dim3(256), 0, s, partial, nchunks, B, Tbuf, C, group,
                     momentum, eps, running_mean, running_var, num_batches_tracked, save_mean, save_rstd);
  if (out && Tout > 0) {
    long total = (long)B * Tout * C;
    if (C % 4 == 0 && all16(y, save_mean, save_rstd, gamma, beta, residual, out))
      hipLaunchKernelGGL(ft_bn_apply4_kernel, dim3(ft_cdiv(total / 4, 256)), dim3(256), 0, s, (const float4*)y,
                         (const float4*)save_mean, (const float4*)save_rstd, (const float4*)gamma, (const float4*)beta,
                         (const float4*)residual, (float4*)out, total / 4, Tbuf, Tout, C / 4);
    else
      hipLaunchKernelGGL(ft_bn_apply_kernel, dim3(ft_cdiv(total, 256)), dim3(256), 0, s, y, save_mean, save_rstd, gamma,
                         beta, residual, out, B, Tbuf, Tout, C);
  }
  return ft_check_launch("bn_train_from_partials");
}

size_t ft_bn_workspace(int B, int Tbuf, int C) {
  ChunkPlan p = plan_chunks((long)B * Tbuf, C);
  return (size_t)p.nchunks * C * 2 * sizeof(double);
}

int ft_bn_train_fwd(const float* y, const float* gamma, const float* beta, const float* residual, float* out,
                    float* running_mean, float* running_var, long* num_batches_tracked, float* save_mean,
                    float* save_rstd, int B, int Tbuf, int Tout, int C, int group, float momentum, float eps,
                    void* workspace, size_t workspace_bytes, void* stream) {
  FT_REQUIRE(B > 0 && Tbuf > 0 && C > 0 && Tout <= Tbuf, "bn_train_fwd: bad dims");
  FT_REQUIRE(workspace && workspace_bytes >= ft_bn_workspace(B, Tbuf, C), "bn_train_fwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  ChunkPlan p = plan_chunks((long)B * Tbuf, C);
  if (C % 4 == 0 && (group == 0 || group % 4 == 0) && all16(y))
    hipLaunchKernelGGL(ft_col_partial4_kernel<0>, dim3(ft_cdiv(C, 64), p.nchunks), dim3(256), 0, s, y, nullptr, nullptr,
                       nullptr, B, Tbuf, Tout, C, group, p.rows_per_chunk, (double*)workspace);
  else
    hipLaunchKernelGGL(ft_col_partial_kernel<0>, dim3(ft_cdiv(C, 64), p.nchunks), dim3(256), 0, s, y, (long)C, nullptr,
                       nullptr, nullptr, B, Tbuf, Tout, C, group, p.rows_per_chunk, (double*)workspace);
  hipLaunchKernelGGL(ft_bn_finalize_kernel, dim3(ft_cdiv(C, 32)), dim3(256), 0, s, (const double*)workspace,
                     p.nchunks, B, Tbuf, C, group, momentum, eps, running_mean, running_var, num_batches_tracked,
                     save_mean, save_rstd);
  if (out && Tout > 0) {
    long total = (long)B * Tout * C;
    if (C % 4 == 0 && all16(y, save_mean, save_rstd, gamma, beta, residual, out))
      hipLaunchKernelGGL(ft_bn_apply4_kernel, dim3(ft_cdiv(total / 4, 256)), dim3(256), 0, s, (const float4*)y,
                         (const float4*)save_mean, (const float4*)save_rstd, (const float4*)gamma, (const float4*)beta,
                         (const float4*)residual, (float4*)out, total / 4, Tbuf, Tout, C / 4);
    else
      hipLaunchKernelGGL(ft_bn_apply_kernel, dim3(ft_cdiv(total, 256)), dim3(256), 0, s, y, save_mean, save_rstd, gamma,
                         beta, residual, out, B, Tbuf, Tout, C);
  }
  return ft_check_launch("bn_train_fwd");
}

int ft_bn_bwd(const float* dout, const float* y, const float* gamma, const float* save_mean, const float* save_rstd,
              float* dy, float* dgamma, float* dbeta, int B, int Tbuf, int Tout, int C, int group, int relu,
              void* workspace, size_t workspace_bytes, void* stream) {
  FT_REQUIRE(B > 0 && Tbuf > 0 && C > 0 && Tout <= Tbuf, "bn_bwd: bad dims");
  FT_REQUIRE(workspace && workspace_bytes >= ft_bn_workspace(B, Tbuf, C), "bn_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  ChunkPlan p = plan_chunks((long)B * Tbuf, C);
  if (C % 4 == 0 && (group == 0 || group % 4 == 0) && all16(y, dout, save_mean, save_rstd))
    hipLaunchKernelGGL(ft_col_partial4_kernel<1>, dim3(ft_cdiv(C, 64), p.nchunks), dim3(256), 0, s, y, dout, save_mean,
                       save_rstd, B, Tbuf, Tout, C, group, p.rows_per_chunk, (double*)workspace);
  else
    hipLaunchKernelGGL(ft_col_partial_kernel<1>, dim3(ft_cdiv(C, 64), p.nchunks), dim3(256), 0, s, y, (long)C, dout,
                       save_mean, save_rstd, B, Tbuf, Tout, C, group, p.rows_per_chunk, (double*)workspace);
  hipLaunchKernelGGL(ft_col_finalize_kernel, dim3(ft_cdiv(C, 32)), dim3(256), 0, s, (const double*)workspace,
                     p.nchunks, C, dbeta, dgamma, 1.0f, 0);
  long total = (long)B * Tbuf * C;
  if (C % 4 == 0 && (group == 0 || group % 4 == 0) && all16(dout, y, save_mean, save_rstd, gamma, dgamma, dbeta, dy))
    hipLaunchKernelGGL(ft_bn_bwd_apply4_kernel<false>, dim3(ft_cdiv(total / 4, 256)), dim3(256), 0, s, (const float4*)dout,
                       (const float4*)y, (const float4*)save_mean, (const float4*)save_rstd, (const float4*)gamma,
                       (const float4*)dgamma, (const float4*)dbeta, (float4*)dy, total / 4, B, Tbuf, Tout, C / 4, group,
                       relu, nullptr);
  else
    hipLaunchKernelGGL(ft_bn_bwd_apply_kernel, dim3(ft_cdiv(total, 256)), dim3(256), 0, s, dout, y, save_mean, save_rstd,
                       gamma, dgamma, dbeta, dy, B, Tbuf, Tout, C, group, relu);
  return ft_check_launch("bn_bwd");
}

// CBHG bank (training): finalize the statistics partials, then out[B,Tout,C] = MaxPool1d(2,1,1)(BatchNorm(y))[:Tout] in
// one pass over y -- the normalised tensor is never written.  16-B path only (C % 4 == 0, group % 4 == 0, aligned).
int ft_bn_pool_from_partials(const double* partial, int nchunks, const float* y, const float* gamma, const float* beta,
                             float* out, float* running_mean, float* running_var, long* num_batches_tracked,
                             float* save_mean, float* save_rstd, int B, int Tbuf, int Tout, int C, int group,
                             float momentum, float eps, void* stream) {
  FT_REQUIRE(B > 0 && Tbuf > 0 && C > 0 && Tout > 0 && Tout <= Tbuf && nchunks >= 1, "bn_pool_from_partials: bad dims");
  FT_REQUIRE(C % 4 == 0 && (group == 0 || group % 4 == 0) && all16(y, save_mean, save_rstd, gamma, beta, out),
             "bn_pool_from_partials: needs C %% 4 == 0 and 16-byte aligned buffers");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(ft_bn_finalize_kernel, dim3(ft_cdiv(C, 32)), dim3(256), 0, s, partial, nchunks, B, Tbuf, C, group,
                     momentum, eps, running_mean, running_var, num_batches_tracked, save_mean, save_rstd);
  const long total = (long)B * Tout * C;
  hipLaunchKernelGGL(ft_bn_apply_pool4_kernel, dim3(ft_cdiv(total / 4, 256)), dim3(256), 0, s, y, save_mean, save_rstd,
                     gamma, beta, (float4*)out, total / 4, Tbuf, Tout, C);
  return ft_check_launch("bn_pool_from_partials");
}

// backward of the pair above: dout [B,Tout,C] = gradient of the POOLED output; dy [B,Tbuf,C], dgamma, dbeta as ft_bn_bwd
int ft_bn_pool_bwd(const float* dout, const float* y, const float* gamma, const float* beta, const float* save_mean,
                   const float* save_rstd, float* dy, float* dgamma, float* dbeta, int B, int Tbuf, int Tout, int C,
                   int group, int relu, void* workspace, size_t workspace_bytes, void* stream) {
  FT_REQUIRE(B > 0 && Tbuf > 0 && C > 0 && Tout > 0 && Tout <= Tbuf, "bn_pool_bwd: bad dims");
  FT_REQUIRE(workspace && workspace_bytes >= ft_bn_workspace(B, Tbuf, C), "bn_pool_bwd: workspace too small");
  FT_REQUIRE(C % 4 == 0 && (group == 0 || group % 4 == 0) &&
                 all16(dout, y, save_mean, save_rstd, gamma, beta, dgamma, dbeta, dy),
             "bn_pool_bwd: needs C %% 4 == 0 and 16-byte aligned buffers");
  hipStream_t s = (hipStream_t)stream;
  ChunkPlan p = plan_chunks((long)B * Tbuf, C);
  hipLaunchKernelGGL(ft_col_partial4_kernel<2>, dim3(ft_cdiv(C, 64), p.nchunks), dim3(256), 0, s, y, dout, save_mean,
                     save_rstd, B, Tbuf, Tout, C, group, p.rows_per_chunk, (double*)workspace, gamma, beta);
  hipLaunchKernelGGL(ft_col_finalize_kernel, dim3(ft_cdiv(C, 32)), dim3(256), 0, s, (const double*)workspace,
                     p.nchunks, C, dbeta, dgamma, 1.0f, 0);
  const long total = (long)B * Tbuf * C;
  hipLaunchKernelGGL(ft_bn_bwd_apply4_kernel<true>, dim3(ft_cdiv(total / 4, 256)), dim3(256), 0, s, (const float4*)dout,
                     (const float4*)y, (const float4*)save_mean, (const float4*)save_rstd, (const float4*)gamma,
                     (const float4*)dgamma, (const float4*)dbeta, (float4*)dy, total / 4, B, Tbuf, Tout, C / 4, group, relu,
                     (const float4*)beta);
  return ft_check_launch("bn_pool_bwd");
}

int ft_bn_fold_eval(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                    float eps, float* scale, float* shift, int C, void* stream) {
  if (C <= 0) return FT_OK;
  hipLaunchKernelGGL(ft_bn_fold_kernel, dim3(ft_cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, gamma, beta,
                     running_mean, running_var, eps, scale, shift, C);
  return ft_check_launch("bn_fold_eval");
}

size_t ft_colsum_workspace(int rows, int C) { return ft_bn_workspace(1, rows > 0 ? rows : 1, C > 0 ? C : 1); }

int ft_colsum(const float* x, long ldx, float* out, int rows, int C, float scale, int accumulate, void* workspace,
              size_t workspace_bytes, void* stream) {
  FT_REQUIRE(rows >= 0 && C >= 0, "colsum: bad dims");
  if (C == 0) return FT_OK;
  hipStream_t s = (hipStream_t)stream;
  if (rows == 0) {
    if (!accumulate) (void)hipMemsetAsync(out, 0, sizeof(float) * C, s);
    return FT_OK;
  }
  FT_REQUIRE(workspace && workspace_bytes >= ft_colsum_workspace(rows, C), "colsum: workspace too small");
  ChunkPlan p = plan_chunks(rows, C);
  if (C % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)x) % 16 == 0)
    hipLaunchKernelGGL(ft_colsum4_partial_kernel, dim3(ft_cdiv(C, 64), p.nchunks), dim3(256), 0, s, x, ldx, (long)rows, C,
                       p.rows_per_chunk, (double*)workspace, (const float*)nullptr);
  else
    hipLaunchKernelGGL(ft_col_partial_kernel<2>, dim3(ft_cdiv(C, 64), p.nchunks), dim3(256), 0, s, x, ldx, nullptr,
                       nullptr, nullptr, 1, rows, rows, C, 0, p.rows_per_chunk, (double*)workspace);
  hipLaunchKernelGGL(ft_col_finalize_kernel, dim3(ft_cdiv(C, 32)), dim3(256), 0, s, (const double*)workspace,
                     p.nchunks, C, out, nullptr, scale, accumulate);
  return ft_check_launch("colsum");
}

// column sums of TWO matrices of one shape in one partial + one finalize launch (LayerNorm's dgamma / dbeta)
int ft_colsum2(const float* x0, const float* x1, long ldx, float* out0, float* out1, int rows, int C, void* workspace,
               size_t workspace_bytes, void* stream) {
  FT_REQUIRE(rows >= 0 && C >= 0, "colsum2: bad dims");
  if (C == 0) return FT_OK;
  const bool vec = C % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)x0) % 16 == 0 && ((uintptr_t)x1) % 16 == 0;
  if (rows == 0 || !vec) {     // rare shapes: two plain calls
    int rc = ft_colsum(x0, ldx, out0, rows, C, 1.0f, 0, workspace, workspace_bytes, stream);
    return rc ? rc : ft_colsum(x1, ldx, out1, rows, C, 1.0f, 0, workspace, workspace_bytes, stream);
  }
  hipStream_t s = (hipStream_t)stream;
  FT_REQUIRE(workspace && workspace_bytes >= ft_colsum_workspace(rows, C), "colsum2: workspace too small");
  ChunkPlan p = plan_chunks(rows, C);
  hipLaunchKernelGGL(ft_colsum4_partial_kernel, dim3(ft_cdiv(C, 64), p.nchunks, 2), dim3(256), 0, s, x0, ldx, (long)rows,
                     C, p.rows_per_chunk, (double*)workspace, x1);
  hipLaunchKernelGGL(ft_col_finalize_kernel, dim3(ft_cdiv(C, 32)), dim3(256), 0, s, (const double*)workspace,
                     p.nchunks, C, out0, out1, 1.0f, 0);
  return ft_check_launch("colsum2");
}

size_t ft_colsum_batch_workspace(int n, const int* C, int rows) {
  size_t total = 0;
  for (int i = 0; i < n; ++i) total += ft_colsum_workspace(rows, C[i]);
  return total;
}

// out[i][c] = sum over rows of x[i][r * ld[i] + c], i < n <= 16, all with `rows` rows: one partial and one finalize launch
// for the lot, each sum bit-identical to its own ft_colsum call
int ft_colsum_batch(int n, const float* const* x, const long* ld, float* const* out, const int* C, int rows,
                    void* workspace, size_t workspace_bytes, void* stream) {
  FT_REQUIRE(n >= 0 && n <= CSB_MAX && rows >= 0, "colsum_batch: n = %d (max %d), rows = %d", n, CSB_MAX, rows);
  if (n == 0) return FT_OK;
  bool vec = rows > 0;
  for (int i = 0; i < n; ++i)
    vec = vec && C[i] > 0 && C[i] % 4 == 0 && ld[i] % 4 == 0 && ((uintptr_t)x[i]) % 16 == 0;
  if (!vec) {                   // rare shapes: one call each
    for (int i = 0; i < n; ++i) {
      const int rc = ft_colsum(x[i], ld[i], out[i], rows, C[i], 1.0f, 0, workspace, workspace_bytes, stream);
      if (rc) return rc;
    }
    return FT_OK;
  }
  FT_REQUIRE(workspace && workspace_bytes >= ft_colsum_batch_workspace(n, C, rows), "colsum_batch: workspace too small");
  ColsumBatch b;
  memset(&b, 0, sizeof(b));
  b.n = n;
  b.rows = rows;
  long off = 0;
  int maxchunks = 1;
  for (int i = 0; i < n; ++i) {
    const ChunkPlan p = plan_chunks(rows, C[i]);
    b.x[i] = x[i]; b.out[i] = out[i]; b.ld[i] = ld[i]; b.C[i] = C[i];
    b.nchunks[i] = p.nchunks; b.rpc[i] = p.rows_per_chunk;
    b.ws_off[i] = off;
    off += (long)(ft_colsum_workspace(rows, C[i]) / sizeof(double));
    b.tile64[i + 1] = b.tile64[i] + ft_cdiv(C[i], 64);
    b.tile32[i + 1] = b.tile32[i] + ft_cdiv(C[i], 32);
    if (p.nchunks > maxchunks) maxchunks = p.nchunks;
  }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(ft_colsum_batch_partial_kernel, dim3(b.tile64[n], maxchunks), dim3(256), 0, s, b, (double*)workspace);
  hipLaunchKernelGGL(ft_colsum_batch_finalize_kernel, dim3(b.tile32[n]), dim3(256), 0, s, b, (const double*)workspace);
  return ft_check_launch("colsum_batch");
}

}  // extern "C"
