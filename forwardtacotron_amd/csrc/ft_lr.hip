// LengthRegulator (models/common_layers.py:12-24) for gfx950: integer prefix-sum + coalesced row copy.
//   scan   : dur[dur<0]=0 (in place, like the reference); r=(long)(dur+0.5f); wave64 inclusive scan ->
//            cum[b][j] = exclusive frame offset of token j, total[b] = frames of item b
//   expand : y[b][t][:] = x[b][tok(t)][:] for t < total[b], 0 above (pad_sequence padding_value 0.)
//            tok(t) by binary search in cum (L1/L2 resident), rows moved as 16-B lanes (1 KiB / wave op)
//   bwd    : dx[b][j][:] = sum_{t in [cum_j, cum_j + r_j)} dy[b][t][:]   (fixed order, reproducible)
// Output rows are exact copies -> bit-exact against the reference.
#include "ft_common.h"

namespace {

__global__ __launch_bounds__(64) void ft_lr_scan_kernel(float* dur, int Tx, int* cum, int* total) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float* d = dur + (long)b * Tx;
  int* c = cum + (long)b * (Tx + 1);
  int carry = 0;
  for (int j0 = 0; j0 < Tx; j0 += 64) {
    int j = j0 + lane;
    int r = 0;
    if (j < Tx) {
      float v = d[j];
      if (v < 0.f) {
        v = 0.f;
        d[j] = 0.f;
      }
      r = (int)(long)(v + 0.5f);      // (dur + 0.5).long(): truncation toward zero
    }
    int s = r;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int u = __shfl_up(s, o, 64);
      if (lane >= o) s += u;
    }
    if (j < Tx) c[j] = carry + s - r;
    carry += __shfl(s, 63, 64);
  }
  if (lane == 0) {
    c[Tx] = carry;
    total[b] = carry;
  }
}

// one block = FR frames of one item; each wave copies FR/4 rows
// output row of (b, t) = b * y_bs + t * y_ts (batch-major: Tm, 1; time-major: 1, B); pad_row (optional, C floats) is
// what rows beyond an item's frames hold instead of zeros
template <int FR>
__global__ __launch_bounds__(256) void ft_lr_expand_kernel(const float* __restrict__ x, const int* __restrict__ cum,
                                                           float* __restrict__ y, int* __restrict__ src_idx,
                                                           int Tx, int Tm, int C, int vec, long y_bs, long y_ts,
                                                           const float* __restrict__ pad_row) {
  __shared__ int src[FR];
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * FR;
  const int* c = cum + (long)b * (Tx + 1);
  if (threadIdx.x < FR) {
    int t = t0 + threadIdx.x;
    int tok = -1;
    if (t < Tm && t < c[Tx]) {
      int lo = 0, hi = Tx - 1;          // largest j with cum[j] <= t  (tokens with r=0 are skipped: cum[j+1]==cum[j])
      while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (c[mid] <= t) lo = mid; else hi = mid - 1;
      }
      tok = lo;
    }
    src[threadIdx.x] = tok;
    if (t < Tm && src_idx) src_idx[(long)b * Tm + t] = tok;
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int f = wave; f < FR; f += 4) {
    int t = t0 + f;
    if (t >= Tm) break;
    int tok = src[f];
    float* yr = y + ((long)b * y_bs + (long)t * y_ts) * C;
    if (vec) {
      const float4* xr = reinterpret_cast<const float4*>(tok < 0 ? pad_row : x + ((long)b * Tx + tok) * C);
      float4* y4 = reinterpret_cast<float4*>(yr);
      for (int i = lane; i < C / 4; i += 64) y4[i] = xr ? xr[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      const float* xr = tok < 0 ? pad_row : x + ((long)b * Tx + tok) * C;
      for (int i = lane; i < C; i += 64) yr[i] = xr ? xr[i] : 0.f;
    }
  }
}

// one wave per (token row, 256-column chunk): a lane adds the frames' float4 of its 4 columns, in frame order.  dy row of
// (b, t) = b * dy_bs + t * dy_ts (as in the expand kernel)
// dtail (optional, [B,C]): rows B*Tx .. B*Tx+B-1 of the grid add up the frames BEYOND item b's tokens (t >= cum[b][Tx]) --
// they belong to no token, but a column sum over all frames (a bias gradient) needs them
__global__ __launch_bounds__(256) void ft_lr_bwd_cols_kernel(const float* __restrict__ dy, const int* __restrict__ cum,
                                                             float* __restrict__ dx, float* __restrict__ dtail, int B,
                                                             int Tx, int Tm, int C, long dy_bs, long dy_ts) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + wave;
  const int col = (blockIdx.y * 64 + lane) * 4;
  const long ntok = (long)B * Tx;
  if (row >= ntok + (dtail ? B : 0) || col >= C) return;
  const bool tail = row >= ntok;
  const int b = tail ? (int)(row - ntok) : (int)(row / Tx), j = tail ? Tx : (int)(row - (long)b * Tx);
  const int* c = cum + (long)b * (Tx + 1);
  int f0 = c[j], f1 = tail ? Tm : c[j + 1];
  if (f1 > Tm) f1 = Tm;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* src = dy + ((long)b * dy_bs + (long)f0 * dy_ts) * C + col;
  const long step = dy_ts * C;
  // four frames in flight per lane (the rows of one token are B * C floats apart in the time-major layout: every load is
  // an HBM access of its own), added in frame order
  for (int t = f0; t < f1; t += 4, src += 4 * step) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 g0 = *reinterpret_cast<const float4*>(src);
    const float4 g1 = t + 1 < f1 ? *reinterpret_cast<const float4*>(src + step) : z;
    const float4 g2 = t + 2 < f1 ? *reinterpret_cast<const float4*>(src + 2 * step) : z;
    const float4 g3 = t + 3 < f1 ? *reinterpret_cast<const float4*>(src + 3 * step) : z;
    a.x = (((a.x + g0.x) + g1.x) + g2.x) + g3.x;
    a.y = (((a.y + g0.y) + g1.y) + g2.y) + g3.y;
    a.z = (((a.z + g0.z) + g1.z) + g2.z) + g3.z;
    a.w = (((a.w + g0.w) + g1.w) + g2.w) + g3.w;
  }
  *reinterpret_cast<float4*>((tail ? dtail + (long)b * C : dx + row * C) + col) = a;
}

// one wave per token row
__global__ __launch_bounds__(256) void ft_lr_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ cum,
                                                        float* __restrict__ dx, int B, int Tx, int Tm, int C, int vec) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  long row = (long)blockIdx.x * 4 + wave;
  if (row >= (long)B * Tx) return;
  int b = (int)(row / Tx), j = (int)(row - (long)b * Tx);
  const int* c = cum + (long)b * (Tx + 1);
  int f0 = c[j], f1 = c[j + 1];
  if (f1 > Tm) f1 = Tm;
  float* dxr = dx + row * C;
  if (vec) {
    for (int i = lane; i < C / 4; i += 64) {
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int t = f0; t < f1; ++t) {
        float4 g = reinterpret_cast<const float4*>(dy + ((long)b * Tm + t) * C)[i];
        a.x += g.x; a.y += g.y; a.z += g.z; a.w += g.w;
      }
      reinterpret_cast<float4*>(dxr)[i] = a;
    }
  } else {
    for (int i = lane; i < C; i += 64) {
      float a = 0.f;
      for (int t = f0; t < f1; ++t) a += dy[((long)b * Tm + t) * C + i];
      dxr[i] = a;
    }
  }
}

}  // namespace

int ft_lr_scan_impl(float* dur, int B, int Tx, int* cum, int* total, hipStream_t stream) {
  FT_REQUIRE(B >= 0 && Tx >= 0, "lr_scan: bad dims");
  if (B == 0) return FT_OK;
  hipLaunchKernelGGL(ft_lr_scan_kernel, dim3(B), dim3(64), 0, stream, dur, Tx, cum, total);
  return ft_check_launch("lr_scan");
}

int ft_lr_expand_impl(const float* x, const int* cum, float* y, int* src_idx, int B, int Tx, int Tm, int C,
                      hipStream_t stream, int y_time_major, const float* pad_row) {
  FT_REQUIRE(B >= 0 && Tx >= 0 && Tm >= 0 && C >= 0, "lr_expand: bad dims");
  if (B == 0 || Tm == 0 || C == 0) return FT_OK;
  FT_REQUIRE(Tx > 0, "lr_expand: Tm > 0 requires Tx > 0");
  int vec = (C % 4 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0) && ((uintptr_t)pad_row % 16 == 0);
  constexpr int FR = 16;
  hipLaunchKernelGGL(ft_lr_expand_kernel<FR>, dim3(ft_cdiv(Tm, FR), B), dim3(256), 0, stream, x, cum, y, src_idx,
                     Tx, Tm, C, vec, y_time_major ? 1L : (long)Tm, y_time_major ? (long)B : 1L, pad_row);
  return ft_check_launch("lr_expand");
}

int ft_lr_bwd_impl(const float* dy, const int* cum, float* dx, int B, int Tx, int Tm, int C, hipStream_t stream,
                   int dy_time_major, float* dtail) {
  FT_REQUIRE(B >= 0 && Tx >= 0 && Tm >= 0 && C >= 0, "lr_bwd: bad dims");
  if (B == 0 || Tx == 0 || C == 0) return FT_OK;
  int vec = (C % 4 == 0) && ((uintptr_t)dy % 16 == 0) && ((uintptr_t)dx % 16 == 0);
  if (dy_time_major || dtail || (vec && C >= 1024)) {      // wide rows: one wave per 256 columns instead of per row
    FT_REQUIRE(vec && (uintptr_t)dtail % 16 == 0, "lr_bwd: this form needs C %% 4 == 0 and 16-byte aligned buffers");
    hipLaunchKernelGGL(ft_lr_bwd_cols_kernel, dim3(ft_cdiv((long)B * Tx + (dtail ? B : 0), 4), ft_cdiv(C, 256)), dim3(256),
                       0, stream, dy, cum, dx, dtail, B, Tx, Tm, C, dy_time_major ? 1L : (long)Tm,
                       dy_time_major ? (long)B : 1L);
    return ft_check_launch("lr_bwd");
  }
  hipLaunchKernelGGL(ft_lr_bwd_kernel, dim3(ft_cdiv((long)B * Tx, 4)), dim3(256), 0, stream, dy, cum, dx, B, Tx, Tm,
                     C, vec);
  return ft_check_launch("lr_bwd");
}
