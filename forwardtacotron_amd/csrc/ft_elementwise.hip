// Element-wise / data-movement kernels (HBM-bound): conv weight packing, embedding, highway gates,
// maxpool, conditioning projections, layout changes, masked L1.  See include/fwdtaco_hip.h.
#include "ft_common.h"
#include "../../include/fwdtaco_hip.h"

namespace {

// w[Cout][Cin][k] -> wp[k][Cout][Cin]
__global__ void ft_pack_conv_w_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int k) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)Cout * Cin * k;
  if (idx >= total) return;
  int j = (int)(idx / ((long)Cout * Cin));
  long rem = idx - (long)j * Cout * Cin;     // co*Cin + ci
  wp[idx] = w[rem * k + j];
}

// tap-major TRANSPOSED pack for the data gradient: wpt[j][ci][co] = w[co][ci][j]  (K-contiguous in Cout, the
// contraction index of dx = sum_j shift_j(dy) * W_j)
__global__ void ft_pack_conv_wt_kernel(const float* __restrict__ w, float* __restrict__ wpt, int Cout, int Cin, int k) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)Cout * Cin * k;
  if (idx >= total) return;
  int j = (int)(idx / ((long)Cout * Cin));
  long rem = idx - (long)j * Cout * Cin;     // ci*Cout + co
  int ci = (int)(rem / Cout), co = (int)(rem - (long)ci * Cout);
  wpt[idx] = w[((long)co * Cin + ci) * k + j];
}

// every pack / transpose of a model in one launch: one workgroup = one 32x32 (d0 x d1) tile of one tap of one entry
// (entry found by bisecting tile_begin); dst straight from registers, dst_t through a padded LDS tile
__global__ __launch_bounds__(256) void ft_pack_weights_kernel(const FtPackDesc* __restrict__ descs, int n) {
  __shared__ float tile[32][33];
  const long bid = blockIdx.x;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (descs[mid].tile_begin <= bid) lo = mid; else hi = mid - 1;
  }
  const FtPackDesc d = descs[lo];
  long t = bid - d.tile_begin;
  const int n1 = (d.d1 + 31) >> 5, n0 = (d.d0 + 31) >> 5;
  const int j1 = (int)(t % n1);
  t /= n1;
  const int j0 = (int)(t % n0), a = (int)(t / n0);
  if (a >= d.k) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c1 = j1 * 32 + tx;
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int c0 = j0 * 32 + ty + 8 * r;
    v[r] = (c0 < d.d0 && c1 < d.d1) ? d.src[((long)c0 * d.d1 + c1) * d.k + a] : 0.f;
  }
  if (d.dst) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int c0 = j0 * 32 + ty + 8 * r;
      if (c0 < d.d0 && c1 < d.d1) d.dst[((long)a * d.d0 + c0) * d.d1 + c1] = v[r];
    }
  }
  if (d.dst_t) {     // block-uniform
#pragma unroll
    for (int r = 0; r < 4; ++r) tile[ty + 8 * r][tx] = v[r];
    __syncthreads();
    const int o0 = j0 * 32 + tx;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int o1 = j1 * 32 + ty + 8 * r;
      if (o1 < d.d1 && o0 < d.d0) d.dst_t[((long)a * d.d1 + o1) * d.d0 + o0] = tile[tx][ty + 8 * r];
    }
  }
}

struct FtSegs {
  const float* src[64];
  float* dst[64];
  long len[64];
};
__global__ __launch_bounds__(256) void ft_copy_segments_kernel(FtSegs sg) {
  const float* s = sg.src[blockIdx.x];
  float* d = sg.dst[blockIdx.x];
  const long n = sg.len[blockIdx.x];
  for (long i = threadIdx.x; i < n; i += 256) d[i] = s[i];
}

// ---- dropout (F.dropout, forward_tacotron.py:35 ; common_layers.py:106,110) -------------------------
// Counter-based mask: keep(i) = hash(seed, i) >= p ; the backward re-derives the same mask from the seed,
// so no mask tensor is stored.  out = keep ? x/(1-p) : 0
__global__ void ft_dropout_kernel(const float* __restrict__ x, float* __restrict__ out, long n, float p,
                                  uint64_t seed) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = ft_dropout_keep(seed, i, p) ? x[i] / (1.0f - p) : 0.f;
}
__global__ void ft_scale_kernel(const float* __restrict__ x, float* __restrict__ out, long n, float s) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = x[i] * s;
}

// ---- embedding ------------------------------------------------------------------------------------
__global__ void ft_embedding_fwd_kernel(const long* __restrict__ idx, const float* __restrict__ w,
                                        float* __restrict__ out, long rows, int C, int V, int* __restrict__ err) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * C) return;
  long r = i / C;
  int c = (int)(i - r * C);
  long v = idx[r];
  if (v < 0 || v >= V) {
    if (c == 0) atomicExch(err, 1);
    out[i] = 0.f;
    return;
  }
  out[i] = w[v * C + c];
}

// onehot[row][v] = (idx[row] == v): the embedding weight gradient is then onehot^T * dout, one TN MFMA GEMM
// (ordered split reduction -> reproducible), shared by every embedding table fed by the same ids
__global__ void ft_onehot_kernel(const long* __restrict__ idx, float* __restrict__ out, long rows, int V) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * V) return;
  long r = i / V;
  int v = (int)(i - r * V);
  out[i] = idx[r] == v ? 1.f : 0.f;
}

// ---- highway (common_layers.py:35-40): x12 = [W1 x + b1 | W2 x + b2] -------------------------------
__global__ void ft_highway_fwd_kernel(const float* __restrict__ x12, const float* __restrict__ x,
                                      float* __restrict__ out, long rows, int C) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * C) return;
  long r = i / C;
  int c = (int)(i - r * C);
  float x1 = x12[r * 2 * C + c], x2 = x12[r * 2 * C + C + c];
  float g = ft_sigmoid(x2);
  out[i] = g * fmaxf(x1, 0.f) + (1.f - g) * x[i];
}

// d12 = [dout*g*(x1>0) | dout*(relu(x1)-x)*g*(1-g)] ; dx_direct = dout*(1-g)
__global__ void ft_highway_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ x12,
                                      const float* __restrict__ x, float* __restrict__ d12,
                                      float* __restrict__ dx, long rows, int C) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * C) return;
  long r = i / C;
  int c = (int)(i - r * C);
  float x1 = x12[r * 2 * C + c], x2 = x12[r * 2 * C + C + c];
  float g = ft_sigmoid(x2);
  float d = dout[i];
  float rx1 = fmaxf(x1, 0.f);
  d12[r * 2 * C + c] = x1 > 0.f ? d * g : 0.f;
  d12[r * 2 * C + C + c] = d * (rx1 - x[i]) * g * (1.f - g);
  dx[i] = d * (1.f - g);
}

// W1 | W2 ([C, C] each) -> their 32-row interleave [2C, C]: row n = row (n / 64) * 32 + n % 32 of W1 (n % 64 < 32) or W2
// -- the B operand of the fused highway forward (ft_gemm.h: FtGemmBatch.hw_mode 1)
__global__ void ft_highway_pack_kernel(const float* __restrict__ w1, const float* __restrict__ w2,
                                       float* __restrict__ out, int C) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2L * C * C) return;
  const int n = (int)(i / C), k = (int)(i - (long)n * C);
  const int unit = (n >> 6) * 32 + (n & 31);
  out[i] = ((n & 63) >> 5 ? w2 : w1)[(long)unit * C + k];
}

// ---- MaxPool1d(2,1,1)[:T] over channels-last: out[t] = max(x[t-1], x[t]) (first max wins ties) -----
__global__ void ft_maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int T, int C) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)B * T * C;
  if (i >= total) return;
  long row = i / C;
  int t = (int)(row % T);
  float v = x[i];
  if (t > 0) {
    float p = x[i - C];
    v = v > p ? v : p;
  }
  out[i] = v;
}
// 16-B lanes (C % 4 == 0, 16-B aligned): one thread = 4 channels of one (b, t) row
__global__ __launch_bounds__(256) void ft_maxpool_fwd4_kernel(const float4* __restrict__ x, float4* __restrict__ out,
                                                              long total4, int T, int C4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const long row = i / C4;
  const int t = (int)(row % T);
  float4 v = x[i];
  if (t > 0) {
    const float4 p = x[i - C4];
    v.x = v.x > p.x ? v.x : p.x;
    v.y = v.y > p.y ? v.y : p.y;
    v.z = v.z > p.z ? v.z : p.z;
    v.w = v.w > p.w ? v.w : p.w;
  }
  out[i] = v;
}
// torch's max_pool backward sends the gradient to the FIRST maximal element of the window (t-1 on ties)
__global__ void ft_maxpool_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ x,
                                      float* __restrict__ dx, int B, int T, int C) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)B * T * C;
  if (i >= total) return;
  long row = i / C;
  int t = (int)(row % T);
  float v = x[i];
  float g = 0.f;
  if (t == 0 || v > x[i - C]) g += dout[i];            // window t picks x[t]
  if (t + 1 < T && !(x[i + C] > v)) g += dout[i + C];  // window t+1 picks x[t] (its first element)
  dx[i] = g;
}
__global__ __launch_bounds__(256) void ft_maxpool_bwd4_kernel(const float4* __restrict__ dout,
                                                              const float4* __restrict__ x, float4* __restrict__ dx,
                                                              long total4, int T, int C4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const long row = i / C4;
  const int t = (int)(row % T);
  const float4 v = x[i];
  const float4 d0 = dout[i];
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  if (t == 0) {
    g = d0;
  } else {
    const float4 p = x[i - C4];
    g.x = v.x > p.x ? d0.x : 0.f;
    g.y = v.y > p.y ? d0.y : 0.f;
    g.z = v.z > p.z ? d0.z : 0.f;
    g.w = v.w > p.w ? d0.w : 0.f;
  }
  if (t + 1 < T) {
    const float4 n = x[i + C4];
    const float4 d1 = dout[i + C4];
    g.x += !(n.x > v.x) ? d1.x : 0.f;
    g.y += !(n.y > v.y) ? d1.y : 0.f;
    g.z += !(n.z > v.z) ? d1.z : 0.f;
    g.w += !(n.w > v.w) ? d1.w : 0.f;
  }
  dx[i] = g;
}

// ---- pitch / energy conditioning (forward_tacotron.py:137-143) -------------------------------------
// out[b,t,c] = x[b,t,c] + sp*(bp[c] + sum_j wp[c][j]*pitch[b,t+j-1]) + se*(be[c] + sum_j we[c][j]*energy[b,t+j-1])
__global__ void ft_cond_add_kernel(const float* __restrict__ x, const float* __restrict__ pitch,
                                   const float* __restrict__ energy, const float* __restrict__ wp,
                                   const float* __restrict__ bp, const float* __restrict__ we,
                                   const float* __restrict__ be, float sp, float se, float* __restrict__ out, int B,
                                   int T, int C, int x_time_major) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)B * T * C;
  if (i >= total) return;
  int c = (int)(i % C);
  long row = i / C;
  int t = (int)(row % T);
  const float xv = x_time_major ? x[((long)t * B + (row / T)) * C + c] : x[i];
  const float* p = pitch + row;
  const float* e = energy + row;
  float pm = t > 0 ? p[-1] : 0.f, pc = p[0], pn = t + 1 < T ? p[1] : 0.f;
  float em = t > 0 ? e[-1] : 0.f, ec = e[0], en = t + 1 < T ? e[1] : 0.f;
  float a = bp[c] + wp[c * 3 + 0] * pm + wp[c * 3 + 1] * pc + wp[c * 3 + 2] * pn;
  float b = be[c] + we[c * 3 + 0] * em + we[c * 3 + 1] * ec + we[c * 3 + 2] * en;
  out[i] = xv + sp * a + se * b;
}
// P[row][0..7] = [p[t-1], p[t], p[t+1], 1, e[t-1], e[t], e[t+1], 1]  (weight grads = dy^T P via the TN GEMM)
__global__ void ft_cond_taps_kernel(const float* __restrict__ pitch, const float* __restrict__ energy,
                                    float* __restrict__ P, int B, int T) {
  long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= (long)B * T) return;
  int t = (int)(row % T);
  const float* p = pitch + row;
  const float* e = energy + row;
  float* o = P + row * 8;
  o[0] = t > 0 ? p[-1] : 0.f; o[1] = p[0]; o[2] = t + 1 < T ? p[1] : 0.f; o[3] = 1.f;
  o[4] = t > 0 ? e[-1] : 0.f; o[5] = e[0]; o[6] = t + 1 < T ? e[1] : 0.f; o[7] = 1.f;
}

// ---- feature concat with a broadcast speaker embedding (multi_forward_tacotron.py:39-42,83-85,208-210) ----
// out[b,t,:] = [ a[b,t,:Ca] | b2[b,t,:Cb] | semb[b,:S] ]   (a may be time-major [T,B,Ca])
__global__ void ft_concat_cols_kernel(const float* __restrict__ a, int Ca, const float* __restrict__ b2, int Cb,
                                      const float* __restrict__ semb, int S, float* __restrict__ out, int B, int T,
                                      int a_time_major) {
  const int C = Ca + Cb + S;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)B * T * C;
  if (i >= total) return;
  long row = i / C;
  int c = (int)(i - row * C);
  int b = (int)(row / T), t = (int)(row - (long)b * T);
  float v;
  if (c < Ca) v = a[(a_time_major ? ((long)t * B + b) : row) * Ca + c];
  else if (c < Ca + Cb) v = b2[row * Cb + (c - Ca)];
  else v = semb[(long)b * S + (c - Ca - Cb)];
  out[i] = v;
}
// dst[b,t,:C] = src[(b,t)*ld + col0 + :C]   (dst optionally time-major [T,B,C])
__global__ void ft_slice_cols_kernel(const float* __restrict__ src, long ld, int col0, int C,
                                     float* __restrict__ dst, int B, int T, int dst_time_major) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)B * T * C;
  if (i >= total) return;
  long row = i / C;
  int c = (int)(i - row * C);
  long srow = row;
  if (dst_time_major) {
    int t = (int)(row / B), b = (int)(row - (long)t * B);
    srow = (long)b * T + t;
  }
  dst[i] = src[srow * ld + col0 + c];
}

// ---- CrossEntropyLoss(ignore_index) over logits [rows, K] (multi_forward_trainer.py:34,88) ----------------
// loss = mean over rows with target != ignore of (logsumexp(logits) - logits[target])
__global__ __launch_bounds__(256) void ft_ce_partial_kernel(const float* __restrict__ logits,
                                                            const long* __restrict__ target, long rows, int K,
                                                            long ignore, double* __restrict__ partial) {
  __shared__ double red[2][4];
  double s = 0.0, n = 0.0;
  for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long)gridDim.x * 256) {
    long tg = target[r];
    if (tg == ignore || tg < 0 || tg >= K) continue;
    const float* l = logits + r * K;
    float mx = l[0];
    for (int k = 1; k < K; ++k) mx = fmaxf(mx, l[k]);
    float se = 0.f;
    for (int k = 0; k < K; ++k) se += expf(l[k] - mx);
    s += (double)(logf(se) + mx - l[tg]);
    n += 1.0;
  }
  s = ft_wave_sum_d(s);
  n = ft_wave_sum_d(n);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s;
    red[1][threadIdx.x >> 6] = n;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    partial[2 * blockIdx.x + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}
__global__ void ft_ce_finalize_kernel(const double* __restrict__ partial, int nblocks, float* __restrict__ loss,
                                      float* __restrict__ inv_count) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s = 0.0, n = 0.0;
  for (int i = 0; i < nblocks; ++i) {
    s += partial[2 * i];
    n += partial[2 * i + 1];
  }
  *loss = (float)(s / n);              // n == 0 -> NaN, like torch
  *inv_count = (float)(1.0 / n);
}
// dlogits[r,k] = (softmax_k - [k == target]) * inv_count * g   (0 for ignored rows)
__global__ void ft_ce_bwd_kernel(const float* __restrict__ logits, const long* __restrict__ target,
                                 const float* __restrict__ inv_count, const float* __restrict__ gout,
                                 float* __restrict__ dlogits, long rows, int K, long ignore) {
  long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  long tg = target[r];
  const float* l = logits + r * K;
  float* d = dlogits + r * K;
  if (tg == ignore || tg < 0 || tg >= K) {
    for (int k = 0; k < K; ++k) d[k] = 0.f;
    return;
  }
  float mx = l[0];
  for (int k = 1; k < K; ++k) mx = fmaxf(mx, l[k]);
  float se = 0.f;
  for (int k = 0; k < K; ++k) se += expf(l[k] - mx);
  const float sc = inv_count[0] * (gout ? gout[0] : 1.f);
  for (int k = 0; k < K; ++k) d[k] = (expf(l[k] - mx) / se - (k == tg ? 1.f : 0.f)) * sc;
}

// ---- [B,T,C] <-> [B,C,Tout] with padding (forward_tacotron.py:155,159,236-239) ---------------------
// out[b,c,t] = t < T ? x[b,t,c] : pad   for t < Tout
__global__ __launch_bounds__(256) void ft_transpose_pad_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                               int T, int C, int Tout, float pad) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    int t = t0 + i, c = c0 + tx;
    tile[i][tx] = (t < T && c < C) ? x[((long)b * T + t) * C + c] : pad;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    int c = c0 + i, t = t0 + tx;
    if (c < C && t < Tout) out[((long)b * C + c) * Tout + t] = tile[tx][i];
  }
}
// dx[b,t,c] = t < Tout ? dout[b,c,t] : 0   for t < T
__global__ __launch_bounds__(256) void ft_transpose_pad_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx,
                                                                   int T, int C, int Tout) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    int c = c0 + i, t = t0 + tx;
    tile[i][tx] = (c < C && t < Tout) ? dout[((long)b * C + c) * Tout + t] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    int t = t0 + i, c = c0 + tx;
    if (t < T && c < C) dx[((long)b * T + t) * C + c] = tile[tx][i];
  }
}

// ---- MaskedL1 (trainer/common.py:69-92) on [B,C,T] --------------------------------------------------
// partial[block] = sum over its elements of |x - target| * (t < len[b])
__global__ __launch_bounds__(256) void ft_masked_l1_partial_kernel(const float* __restrict__ x,
                                                                   const float* __restrict__ target,
                                                                   const long* __restrict__ lens, int B, int C, int T,
                                                                   double* __restrict__ partial) {
  __shared__ double red[4];
  long total = (long)B * C * T;
  double acc = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    int t = (int)(i % T);
    int b = (int)(i / ((long)C * T));
    if (t < lens[b]) acc += (double)fabsf(x[i] - target[i]);
  }
  acc = ft_wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// loss = sum(partial) / (C * sum_b min(len_b, T)) ; also stores 1/denominator for the backward
__global__ __launch_bounds__(256) void ft_masked_l1_finalize_kernel(const double* __restrict__ partial, int nblocks,
                                                                    const long* __restrict__ lens, int B, int C,
                                                                    int T, float* __restrict__ loss,
                                                                    float* __restrict__ inv_denom) {
  __shared__ double red[2][4];
  double s = 0.0, n = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) s += partial[i];
  for (int b = threadIdx.x; b < B; b += 256) {
    long l = lens[b];
    if (l < 0) l = 0;
    if (l > T) l = T;
    n += (double)l;
  }
  s = ft_wave_sum_d(s);
  n = ft_wave_sum_d(n);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s;
    red[1][threadIdx.x >> 6] = n;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  s = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  n = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  n *= (double)C;
  *loss = (float)(s / n);
  *inv_denom = (float)(1.0 / n);
}
// dx = sign(x - target) * mask * inv_denom * gscale[0]*factor
__global__ void ft_masked_l1_bwd_kernel(const float* __restrict__ x, const float* __restrict__ target,
                                        const long* __restrict__ lens, const float* __restrict__ inv_denom,
                                        const float* __restrict__ gout, float factor, float* __restrict__ dx, int B,
                                        int C, int T) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)B * C * T;
  if (i >= total) return;
  int t = (int)(i % T);
  int b = (int)(i / ((long)C * T));
  float g = 0.f;
  if (t < lens[b]) {
    float d = x[i] - target[i];
    float s = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    g = s * inv_denom[0] * (gout ? gout[0] : 1.f) * factor;
  }
  dx[i] = g;
}

}  // namespace

extern "C" {

int ft_conv_pack_weight(const float* w, float* wp, int Cout, int Cin, int k, void* stream) {
  FT_REQUIRE(Cout >= 0 && Cin >= 0 && k >= 1, "conv_pack_weight: bad dims");
  long total = (long)Cout * Cin * k;
  if (total == 0) return FT_OK;
  hipLaunchKernelGGL(ft_pack_conv_w_kernel, dim3(ft_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, wp, Cout,
                     Cin, k);
  return ft_check_launch("conv_pack_weight");
}

int ft_conv_pack_weight_t(const float* w, float* wpt, int Cout, int Cin, int k, void* stream) {
  FT_REQUIRE(Cout >= 0 && Cin >= 0 && k >= 1, "conv_pack_weight_t: bad dims");
  long total = (long)Cout * Cin * k;
  if (total == 0) return FT_OK;
  hipLaunchKernelGGL(ft_pack_conv_wt_kernel, dim3(ft_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, wpt,
                     Cout, Cin, k);
  return ft_check_launch("conv_pack_weight_t");
}

int ft_pack_weights(const FtPackDesc* descs, int n, long total_tiles, void* stream) {
  FT_REQUIRE(n >= 0 && total_tiles >= 0 && total_tiles < (1L << 31), "pack_weights: bad sizes");
  if (n == 0 || total_tiles == 0) return FT_OK;
  FT_REQUIRE(descs != nullptr, "pack_weights: null descriptor array");
  hipLaunchKernelGGL(ft_pack_weights_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, descs, n);
  return ft_check_launch("pack_weights");
}

int ft_copy_segments(const float* const* src, float* const* dst, const long* len, int n, void* stream) {
  FT_REQUIRE(n >= 0 && n <= 64, "copy_segments: n=%d (max 64)", n);
  if (n == 0) return FT_OK;
  FtSegs sg;
  for (int i = 0; i < n; ++i) {
    FT_REQUIRE(len[i] >= 0 && (len[i] == 0 || (src[i] && dst[i])), "copy_segments: bad segment %d", i);
    sg.src[i] = src[i];
    sg.dst[i] = dst[i];
    sg.len[i] = len[i];
  }
  hipLaunchKernelGGL(ft_copy_segments_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, sg);
  return ft_check_launch("copy_segments");
}

int ft_dropout(const float* x, float* out, long n, float p, uint64_t seed, void* stream) {
  FT_REQUIRE(p >= 0.f && p < 1.f, "dropout: p must be in [0,1)");
  if (n <= 0) return FT_OK;
  hipLaunchKernelGGL(ft_dropout_kernel, dim3(ft_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, out, n, p, seed);
  return ft_check_launch("dropout");
}

int ft_scale(const float* x, float* out, long n, float s, void* stream) {
  if (n <= 0) return FT_OK;
  hipLaunchKernelGGL(ft_scale_kernel, dim3(ft_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, out, n, s);
  return ft_check_launch("scale");
}

int ft_embedding_fwd(const long* idx, const float* w, float* out, long rows, int C, int V, int* err_flag,
                     void* stream) {
  FT_REQUIRE(rows >= 0 && C >= 0 && V > 0, "embedding_fwd: bad dims");
  if (rows * C == 0) return FT_OK;
  hipLaunchKernelGGL(ft_embedding_fwd_kernel, dim3(ft_cdiv(rows * C, 256)), dim3(256), 0, (hipStream_t)stream, idx, w,
                     out, rows, C, V, err_flag);
  return ft_check_launch("embedding_fwd");
}

int ft_onehot(const long* idx, float* out, long rows, int V, void* stream) {
  FT_REQUIRE(rows >= 0 && V > 0, "onehot: bad dims");
  if (rows == 0) return FT_OK;
  hipLaunchKernelGGL(ft_onehot_kernel, dim3(ft_cdiv(rows * V, 256)), dim3(256), 0, (hipStream_t)stream, idx, out, rows,
                     V);
  return ft_check_launch("onehot");
}

int ft_highway_gate_fwd(const float* x12, const float* x, float* out, long rows, int C, void* stream) {
  if (rows * C <= 0) return FT_OK;
  hipLaunchKernelGGL(ft_highway_fwd_kernel, dim3(ft_cdiv(rows * C, 256)), dim3(256), 0, (hipStream_t)stream, x12, x,
                     out, rows, C);
  return ft_check_launch("highway_gate_fwd");
}

int ft_highway_gate_bwd(const float* dout, const float* x12, const float* x, float* d12, float* dx, long rows, int C,
                        void* stream) {
  if (rows * C <= 0) return FT_OK;
  hipLaunchKernelGGL(ft_highway_bwd_kernel, dim3(ft_cdiv(rows * C, 256)), dim3(256), 0, (hipStream_t)stream, dout, x12,
                     x, d12, dx, rows, C);
  return ft_check_launch("highway_gate_bwd");
}

int ft_highway_pack(const float* w1, const float* w2, float* w12i, int C, void* stream) {
  FT_REQUIRE(C > 0 && C % 32 == 0, "highway_pack: the fused highway needs C %% 32 == 0 (C = %d)", C);
  hipLaunchKernelGGL(ft_highway_pack_kernel, dim3(ft_cdiv(2L * C * C, 256)), dim3(256), 0, (hipStream_t)stream, w1, w2,
                     w12i, C);
  return ft_check_launch("highway_pack");
}

int ft_maxpool2_fwd(const float* x, float* out, int B, int T, int C, void* stream) {
  long total = (long)B * T * C;
  if (total <= 0) return FT_OK;
  if (C % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0)
    hipLaunchKernelGGL(ft_maxpool_fwd4_kernel, dim3(ft_cdiv(total / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)x, (float4*)out, total / 4, T, C / 4);
  else
    hipLaunchKernelGGL(ft_maxpool_fwd_kernel, dim3(ft_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, out, B,
                       T, C);
  return ft_check_launch("maxpool2_fwd");
}

int ft_maxpool2_bwd(const float* dout, const float* x, float* dx, int B, int T, int C, void* stream) {
  long total = (long)B * T * C;
  if (total <= 0) return FT_OK;
  if (C % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dout % 16) == 0 && ((uintptr_t)dx % 16) == 0)
    hipLaunchKernelGGL(ft_maxpool_bwd4_kernel, dim3(ft_cdiv(total / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)dout, (const float4*)x, (float4*)dx, total / 4, T, C / 4);
  else
    hipLaunchKernelGGL(ft_maxpool_bwd_kernel, dim3(ft_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, dout, x,
                       dx, B, T, C);
  return ft_check_launch("maxpool2_bwd");
}

int ft_cond_add_fwd(const float* x, const float* pitch, const float* energy, const float* w_pitch,
                    const float* b_pitch, const float* w_energy, const float* b_energy, float pitch_strength,
                    float energy_strength, float* out, int B, int T, int C, int x_time_major, void* stream) {
  long total = (long)B * T * C;
  if (total <= 0) return FT_OK;
  hipLaunchKernelGGL(ft_cond_add_kernel, dim3(ft_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, pitch,
                     energy, w_pitch, b_pitch, w_energy, b_energy, pitch_strength, energy_strength, out, B, T, C,
                     x_time_major);
  return ft_check_launch("cond_add_fwd");
}

int ft_cond_taps(const float* pitch, const float* energy, float* taps, int B, int T, void* stream) {
  long rows = (long)B * T;
  if (rows <= 0) return FT_OK;
  hipLaunchKernelGGL(ft_cond_taps_kernel, dim3(ft_cdiv(rows, 256)), dim3(256), 0, (hipStream_t)stream, pitch, energy,
                     taps, B, T);
  return ft_check_launch("cond_taps");
}

int ft_transpose_pad_fwd(const float* x, float* out, int B, int T, int C, int Tout, float pad, void* stream) {
  if (B <= 0 || C <= 0 || Tout <= 0) return FT_OK;
  FT_REQUIRE(B <= 65535 && ft_cdiv(C, 32) <= 65535, "transpose_pad_fwd: grid too large");
  hipLaunchKernelGGL(ft_transpose_pad_kernel, dim3(ft_cdiv(Tout, 32), ft_cdiv(C, 32), B), dim3(256), 0,
                     (hipStream_t)stream, x, out, T, C, Tout, pad);
  return ft_check_launch("transpose_pad_fwd");
}

int ft_transpose_pad_bwd(const float* dout, float* dx, int B, int T, int C, int Tout, void* stream) {
  if (B <= 0 || C <= 0 || T <= 0) return FT_OK;
  FT_REQUIRE(B <= 65535 && ft_cdiv(C, 32) <= 65535, "transpose_pad_bwd: grid too large");
  hipLaunchKernelGGL(ft_transpose_pad_bwd_kernel, dim3(ft_cdiv(T, 32), ft_cdiv(C, 32), B), dim3(256), 0,
                     (hipStream_t)stream, dout, dx, T, C, Tout);
  return ft_check_launch("transpose_pad_bwd");
}

int ft_concat_cols(const float* a, int Ca, const float* b2, int Cb, const float* semb, int S, float* out, int B,
                   int T, int a_time_major, void* stream) {
  FT_REQUIRE(Ca >= 0 && Cb >= 0 && S >= 0 && (Cb == 0 || b2) && (S == 0 || semb), "concat_cols: bad arguments");
  long total = (long)B * T * (Ca + Cb + S);
  if (total <= 0) return FT_OK;
  hipLaunchKernelGGL(ft_concat_cols_kernel, dim3(ft_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, a, Ca, b2,
                     Cb, semb, S, out, B, T, a_time_major);
  return ft_check_launch("concat_cols");
}

int ft_slice_cols(const float* src, long ld, int col0, int C, float* dst, int B, int T, int dst_time_major,
                  void* stream) {
  long total = (long)B * T * C;
  if (total <= 0) return FT_OK;
  hipLaunchKernelGGL(ft_slice_cols_kernel, dim3(ft_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, src, ld, col0,
                     C, dst, B, T, dst_time_major);
  return ft_check_launch("slice_cols");
}

size_t ft_cross_entropy_workspace(void) { return 2 * 256 * sizeof(double); }

int ft_cross_entropy_fwd(const float* logits, const long* target, long rows, int K, long ignore_index, float* loss,
                         float* inv_count, void* workspace, size_t workspace_bytes, void* stream) {
  FT_REQUIRE(rows > 0 && K > 0, "cross_entropy_fwd: bad dims");
  FT_REQUIRE(workspace && workspace_bytes >= ft_cross_entropy_workspace(), "cross_entropy_fwd: workspace too small");
  int nb = ft_cdiv(rows, 256);
  if (nb > 256) nb = 256;
  hipLaunchKernelGGL(ft_ce_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, logits, target, rows, K,
                     ignore_index, (double*)workspace);
  hipLaunchKernelGGL(ft_ce_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double*)workspace, nb,
                     loss, inv_count);
  return ft_check_launch("cross_entropy_fwd");
}

int ft_cross_entropy_bwd(const float* logits, const long* target, const float* inv_count, const float* grad_out,
                         float* dlogits, long rows, int K, long ignore_index, void* stream) {
  if (rows <= 0) return FT_OK;
  hipLaunchKernelGGL(ft_ce_bwd_kernel, dim3(ft_cdiv(rows, 256)), dim3(256), 0, (hipStream_t)stream, logits, target,
                     inv_count, grad_out, dlogits, rows, K, ignore_index);
  return ft_check_launch("cross_entropy_bwd");
}

size_t ft_masked_l1_workspace(void) { return 1024 * sizeof(double); }

int ft_masked_l1_fwd(const float* x, const float* target, const long* lens, float* loss, float* inv_denom, int B,
                     int C, int T, void* workspace, size_t workspace_bytes, void* stream) {
  FT_REQUIRE(B > 0 && C > 0 && T > 0, "masked_l1_fwd: bad dims");
  FT_REQUIRE(workspace && workspace_bytes >= ft_masked_l1_workspace(), "masked_l1_fwd: workspace too small");
  long total = (long)B * C * T;
  int nb = ft_cdiv(total, 256 * 8);
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(ft_masked_l1_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, target, lens, B, C, T,
                     (double*)workspace);
  hipLaunchKernelGGL(ft_masked_l1_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream,
                     (const double*)workspace, nb, lens, B, C, T, loss, inv_denom);
  return ft_check_launch("masked_l1_fwd");
}

int ft_masked_l1_bwd(const float* x, const float* target, const long* lens, const float* inv_denom,
                     const float* grad_out, float factor, float* dx, int B, int C, int T, void* stream) {
  long total = (long)B * C * T;
  if (total <= 0) return FT_OK;
  hipLaunchKernelGGL(ft_masked_l1_bwd_kernel, dim3(ft_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, target,
                     lens, inv_denom, grad_out, factor, dx, B, C, T);
  return ft_check_launch("masked_l1_bwd");
}

}  // extern "C"
