// Bidirectional GRU / (packed) LSTM recurrences for gfx950 -- nn.GRU / nn.LSTM semantics
// (models/common_layers.py:89,123 ; models/forward_tacotron.py:24,96-99,147-152).
//
// The input projections x*W_ih^T (+b_ih) are hoisted into one big MFMA GEMM (ft_linear_multi_fwd); what is
// left per timestep is h[B,H] * W_hh^T[H,G*H] plus the cell math.  Version 1 of the recurrence issues ONE
// launch per timestep covering both directions: a workgroup owns 32 batch rows x U=8 hidden units (all G
// gates of those units -> 32 MFMA columns), the K=H contraction is split across the waves of the block
// (each wave a strided set of 8-wide k-octets, operands loaded straight from L2 as 16-B lanes using the
// "any K order, as long as A and B agree" freedom of the MFMA), partial 32x32 tiles are reduced through
// LDS and 256 threads finish the cell update.  Kernel boundaries provide the step-to-step dependency.
//
// Time indexing (L_b = lens[b] for the packed LSTM, T otherwise); s = launch index:
//   forward pass : t = s (dir 0) | L_b-1-s (dir 1) ;  h_prev at t-1 | t+1           ; active iff s < L_b
//   backward pass: t = L_b-1-s (dir 0) | s (dir 1) ;  "next" (already done) t+1|t-1 ; h_prev at t-1 | t+1
#include "ft_common.h"

namespace {

constexpr int U = 8;        // hidden units per block in the forward step
constexpr int RLD = 40;     // LDS row stride of the 32x32 partial tiles (conflict-free 8-wide reads)

__device__ __forceinline__ float4 ld4g(const float* p, int remaining, bool vec) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (remaining >= 4 && vec) {
    v = *reinterpret_cast<const float4*>(p);
  } else {
    if (remaining > 0) v.x = p[0];
    if (remaining > 1) v.y = p[1];
    if (remaining > 2) v.z = p[2];
    if (remaining > 3) v.w = p[3];
  }
  return v;
}

__device__ __forceinline__ void mfma4(const float4& a, const float4& b, f32x16& acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
}

// store this wave's 32x32 partial (lane: column l31, rows (e&3)+8*(e>>2)+4*half)
__device__ __forceinline__ void store_partial(float* red, int wave, int lane, const f32x16& acc) {
  const int half = lane >> 5, l31 = lane & 31;
  float* r = red + wave * 32 * RLD;
#pragma unroll
  for (int e = 0; e < 16; ++e) r[((e & 3) + 8 * (e >> 2) + 4 * half) * RLD + l31] = acc[e];
}

struct RnnFwdArgs {
  const float* xp;        // [B,T,ND*G*H]   x W_ih^T + b_ih
  const float* whh[2];    // [G*H,H]
  const float* bhh[2];    // [G*H]
  float* out;             // [B,T,ND*H]  raw hidden states (zero where inactive)
  float* cst;             // LSTM: [B,T,ND*H] cell states
  float* gates;           // optional [B,T,ND,4*H] saved activations (training)
  const long* lens;       // optional [B]
  int B, T, H, ND, s, vec;
};

template <int G, int NW>
__global__ __launch_bounds__(NW * 64) void ft_rnn_fwd_step_kernel(RnnFwdArgs a) {
  __shared__ float red[NW * 32 * RLD];
  const int d = blockIdx.z;
  const int u0 = blockIdx.x * U;
  const int b0 = blockIdx.y * 32;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int half = lane >> 5, l31 = lane & 31;
  const int H = a.H, T = a.T, s = a.s;
  const long ldo = (long)a.ND * H;

  // ---- A row (batch item) of this lane
  const int bA = b0 + l31;
  int LA = T;
  if (bA < a.B && a.lens) {
    long l = a.lens[bA];
    LA = l < 0 ? 0 : (l > T ? T : (int)l);
  }
  const bool actA = bA < a.B && s < LA && s > 0;
  const int tprevA = d == 0 ? s - 1 : LA - s;          // (t-1) | (t+1) with t = LA-1-s
  const float* arow = a.out + ((long)bA * T + tprevA) * ldo + (long)d * H;
  // ---- B column (gate row) of this lane
  const int gj = l31 / U, ul = l31 - gj * U;
  const bool okB = gj < G && (u0 + ul) < H;
  const float* brow = a.whh[d] + ((long)gj * H + u0 + ul) * H;

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int noct = (H + 7) / 8;
  if (s > 0) {

    for (int c = wave; c < noct; c += NW) {
      const int k = 8 * c + 4 * half;
      float4 av = ld4g(arow + k, actA ? H - k : 0, a.vec);
      float4 bv = ld4g(brow + k, okB ? H - k : 0, a.vec);
      mfma4(av, bv, acc);
    }
  }
  store_partial(red, wave, lane, acc);
  __syncthreads();

  if (tid < 32 * U) {
    const int i = tid / U, uu = tid - i * U;
    const int b = b0 + i, u = u0 + uu;
    if (b >= a.B || u >= H) return;
    int L = T;
    if (a.lens) {
      long l = a.lens[b];
      L = l < 0 ? 0 : (l > T ? T : (int)l);
    }
    if (s >= L) return;
    const int t = d == 0 ? s : L - 1 - s;
    const int tprev = d == 0 ? t - 1 : t + 1;
    float hp[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += red[w * 32 * RLD + i * RLD + g * U + uu];
      hp[g] = v + a.bhh[d][g * H + u];
    }
    const float* xr = a.xp + ((long)b * T + t) * ((long)a.ND * G * H) + (long)d * G * H + u;
    const long o = ((long)b * T + t) * ldo + (long)d * H + u;
    const long op = ((long)b * T + tprev) * ldo + (long)d * H + u;
    float* gs = a.gates ? a.gates + (((long)b * T + t) * a.ND + d) * 4 * H + u : nullptr;
    if (G == 3) {
      float r = ft_sigmoid(xr[0] + hp[0]);
      float z = ft_sigmoid(xr[H] + hp[1]);
      float n = ft_tanh(xr[2 * H] + r * hp[2]);
      float hprev = s > 0 ? a.out[op] : 0.f;
      a.out[o] = (1.f - z) * n + z * hprev;
      if (gs) {
        gs[0] = r; gs[H] = z; gs[2 * H] = n; gs[3 * H] = hp[2];
      }
    } else {
      float ig = ft_sigmoid(xr[0] + hp[0]);
      float fg = ft_sigmoid(xr[H] + hp[1]);
      float gg = ft_tanh(xr[2 * H] + hp[2]);
      float og = ft_sigmoid(xr[3 * H] + hp[G - 1]);
      float cprev = s > 0 ? a.cst[op] : 0.f;
      float c = fg * cprev + ig * gg;
      a.cst[o] = c;
      a.out[o] = og * ft_tanh(c);
      if (gs) {
        gs[0] = ig; gs[H] = fg; gs[2 * H] = gg; gs[3 * H] = og;
      }
    }
  }
}

struct RnnBwdArgs {
  const float* dout;      // [B,T,ND*H]
  const float* out;       // [B,T,ND*H] raw hidden states
  const float* cst;       // LSTM cell states
  const float* gates;     // [B,T,ND,4H]
  const float* whhT[2];   // [H, G*H]  (W_hh transposed)
  float* dxp;             // [B,T,ND*G*H]  d(pre-activation) wrt the input projection
  float* dhp;             // GRU only: [B,T,ND*G*H] d wrt the hidden projection (n gate scaled by r); LSTM: == dxp
  float* carry;           // [B,ND,H]  GRU: dh*z ; LSTM: dc*f
  const long* lens;
  int B, T, H, ND, s, vec;
};

// block = 32 batch rows x 32 hidden units ; K = G*H split over NW waves
template <int G, int NW>
__global__ __launch_bounds__(NW * 64) void ft_rnn_bwd_step_kernel(RnnBwdArgs a) {
  __shared__ float red[NW * 32 * RLD];
  const int d = blockIdx.z;
  const int u0 = blockIdx.x * 32;
  const int b0 = blockIdx.y * 32;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int half = lane >> 5, l31 = lane & 31;
  const int H = a.H, T = a.T, s = a.s, K = G * H;
  const long ldg = (long)a.ND * K;

  const int bA = b0 + l31;
  int LA = T;
  if (bA < a.B && a.lens) {
    long l = a.lens[bA];
    LA = l < 0 ? 0 : (l > T ? T : (int)l);
  }
  const bool actA = bA < a.B && s < LA && s > 0;
  const int tnextA = d == 0 ? LA - s : s - 1;          // (t+1) with t=LA-1-s | (t-1) with t=s
  const float* arow = a.dhp + ((long)bA * T + tnextA) * ldg + (long)d * K;
  const bool okB = (u0 + l31) < H;
  const float* brow = a.whhT[d] + (long)(u0 + l31) * K;

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int noct = (K + 7) / 8;
  if (s > 0) {

    for (int c = wave; c < noct; c += NW) {
      const int k = 8 * c + 4 * half;
      float4 av = ld4g(arow + k, actA ? K - k : 0, a.vec);
      float4 bv = ld4g(brow + k, okB ? K - k : 0, a.vec);
      mfma4(av, bv, acc);
    }
  }
  store_partial(red, wave, lane, acc);
  __syncthreads();

  for (int p = tid; p < 32 * 32; p += NW * 64) {
    const int i = p >> 5, j = p & 31;
    const int b = b0 + i, u = u0 + j;
    if (b >= a.B || u >= H) continue;
    int L = T;
    if (a.lens) {
      long l = a.lens[b];
      L = l < 0 ? 0 : (l > T ? T : (int)l);
    }
    if (s >= L) continue;
    const int t = d == 0 ? L - 1 - s : s;
    const int tprev = d == 0 ? t - 1 : t + 1;
    const bool has_prev = tprev >= 0 && tprev < L;
    float rec = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) rec += red[w * 32 * RLD + i * RLD + j];
    const long ldo = (long)a.ND * H;
    const long o = ((long)b * T + t) * ldo + (long)d * H + u;
    const long op = ((long)b * T + tprev) * ldo + (long)d * H + u;
    const float* gs = a.gates + (((long)b * T + t) * a.ND + d) * 4 * H + u;
    float* cr = a.carry + ((long)b * a.ND + d) * H + u;
    const float cin = s > 0 ? *cr : 0.f;
    float* dx = a.dxp + ((long)b * T + t) * ldg + (long)d * K + u;
    if (G == 3) {
      const float dh = a.dout[o] + rec + cin;
      const float r = gs[0], z = gs[H], n = gs[2 * H], hn = gs[3 * H];
      const float hprev = has_prev ? a.out[op] : 0.f;
      const float dz = dh * (hprev - n) * z * (1.f - z);
      const float dn = dh * (1.f - z) * (1.f - n * n);
      const float dr = dn * hn * r * (1.f - r);
      dx[0] = dr; dx[H] = dz; dx[2 * H] = dn;
      float* dhh = a.dhp + ((long)b * T + t) * ldg + (long)d * K + u;
      dhh[0] = dr; dhh[H] = dz; dhh[2 * H] = dn * r;
      *cr = dh * z;
    } else {
      const float dh = a.dout[o] + rec;
      const float ig = gs[0], fg = gs[H], gg = gs[2 * H], og = gs[3 * H];
      const float c = a.cst[o];
      const float cprev = has_prev ? a.cst[op] : 0.f;
      const float tc = ft_tanh(c);
      const float dc = dh * og * (1.f - tc * tc) + cin;
      dx[0] = dc * gg * ig * (1.f - ig);
      dx[H] = dc * cprev * fg * (1.f - fg);
      dx[2 * H] = dc * ig * (1.f - gg * gg);
      dx[3 * H] = dh * tc * og * (1.f - og);
      *cr = dc * fg;
    }
  }
}

// out[b,t,:] = t < len[b] ? raw[b,t,:] : pad      (pad_packed_sequence padding_value, forward_tacotron.py:152)
__global__ void ft_fill_padded_kernel(const float* __restrict__ raw, const long* __restrict__ lens,
                                      float* __restrict__ out, int B, int T, int C, float pad) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)B * T * C;
  if (i >= total) return;
  long row = i / C;
  int b = (int)(row / T), t = (int)(row - (long)b * T);
  out[i] = t < lens[b] ? raw[i] : pad;
}
// dst = t < len[b] ? src : 0
__global__ void ft_mask_rows_kernel(const float* __restrict__ src, const long* __restrict__ lens,
                                    float* __restrict__ dst, int B, int T, int C) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)B * T * C;
  if (i >= total) return;
  long row = i / C;
  int b = (int)(row / T), t = (int)(row - (long)b * T);
  dst[i] = t < lens[b] ? src[i] : 0.f;
}

template <int G>
int rnn_fwd(const float* xp, const float* whh_f, const float* whh_r, const float* bhh_f, const float* bhh_r,
            float* out, float* cst, float* gates, const long* lens, int B, int T, int H, hipStream_t stream) {
  RnnFwdArgs a;
  a.xp = xp; a.whh[0] = whh_f; a.whh[1] = whh_r; a.bhh[0] = bhh_f; a.bhh[1] = bhh_r;
  a.out = out; a.cst = cst; a.gates = gates; a.lens = lens;
  a.B = B; a.T = T; a.H = H; a.ND = 2;
  a.vec = (H % 4 == 0) && ((uintptr_t)out % 16 == 0) && ((uintptr_t)whh_f % 16 == 0) && ((uintptr_t)whh_r % 16 == 0);
  dim3 grid(ft_cdiv(H, U), ft_cdiv(B, 32), 2);
  const bool wide = H > 256;
  for (int s = 0; s < T; ++s) {
    a.s = s;
    if (wide)
      hipLaunchKernelGGL((ft_rnn_fwd_step_kernel<G, 8>), grid, dim3(512), 0, stream, a);
    else
      hipLaunchKernelGGL((ft_rnn_fwd_step_kernel<G, 4>), grid, dim3(256), 0, stream, a);
  }
  return ft_check_launch("rnn_fwd");
}

template <int G>
int rnn_bwd(const float* dout, const float* out, const float* cst, const float* gates, const float* whhT_f,
            const float* whhT_r, float* dxp, float* dhp, float* carry, const long* lens, int B, int T, int H,
            hipStream_t stream) {
  RnnBwdArgs a;
  a.dout = dout; a.out = out; a.cst = cst; a.gates = gates; a.whhT[0] = whhT_f; a.whhT[1] = whhT_r;
  a.dxp = dxp; a.dhp = dhp; a.carry = carry; a.lens = lens;
  a.B = B; a.T = T; a.H = H; a.ND = 2;
  a.vec = (H % 4 == 0) && ((uintptr_t)dhp % 16 == 0) && ((uintptr_t)whhT_f % 16 == 0) &&
          ((uintptr_t)whhT_r % 16 == 0);
  dim3 grid(ft_cdiv(H, 32), ft_cdiv(B, 32), 2);
  const bool wide = (long)G * H > 1024;
  for (int s = 0; s < T; ++s) {
    a.s = s;
    if (wide)
      hipLaunchKernelGGL((ft_rnn_bwd_step_kernel<G, 16>), grid, dim3(1024), 0, stream, a);
    else
      hipLaunchKernelGGL((ft_rnn_bwd_step_kernel<G, 8>), grid, dim3(512), 0, stream, a);
  }
  return ft_check_launch("rnn_bwd");
}

}  // namespace

extern "C" {

int ft_gru_fwd(const float* xp, const float* whh_f, const float* whh_r, const float* bhh_f, const float* bhh_r,
               float* out, float* gates, int B, int T, int H, void* stream) {
  FT_REQUIRE(B > 0 && T >= 0 && H > 0, "gru_fwd: bad dims");
  return rnn_fwd<3>(xp, whh_f, whh_r, bhh_f, bhh_r, out, nullptr, gates, nullptr, B, T, H, (hipStream_t)stream);
}

int ft_gru_bwd(const float* dout, const float* out, const float* gates, const float* whhT_f, const float* whhT_r,
               float* dxp, float* dhp, float* carry, int B, int T, int H, void* stream) {
  FT_REQUIRE(B > 0 && T >= 0 && H > 0, "gru_bwd: bad dims");
  return rnn_bwd<3>(dout, out, nullptr, gates, whhT_f, whhT_r, dxp, dhp, carry, nullptr, B, T, H,
                    (hipStream_t)stream);
}

int ft_lstm_fwd(const float* xp, const float* whh_f, const float* whh_r, const float* bhh_f, const float* bhh_r,
                const long* lens, float* out_raw, float* cstate, float* gates, int B, int T, int H, void* stream) {
  FT_REQUIRE(B > 0 && T >= 0 && H > 0, "lstm_fwd: bad dims");
  hipStream_t s = (hipStream_t)stream;
  if (lens) {   // inactive positions must read as zeros
    (void)hipMemsetAsync(out_raw, 0, sizeof(float) * (size_t)B * T * 2 * H, s);
    (void)hipMemsetAsync(cstate, 0, sizeof(float) * (size_t)B * T * 2 * H, s);
  }
  return rnn_fwd<4>(xp, whh_f, whh_r, bhh_f, bhh_r, out_raw, cstate, gates, lens, B, T, H, s);
}

int ft_lstm_bwd(const float* dout, const float* out_raw, const float* cstate, const float* gates,
                const float* whhT_f, const float* whhT_r, const long* lens, float* dgates, float* carry, int B, int T,
                int H, void* stream) {
  FT_REQUIRE(B > 0 && T >= 0 && H > 0, "lstm_bwd: bad dims");
  hipStream_t s = (hipStream_t)stream;
  if (lens) (void)hipMemsetAsync(dgates, 0, sizeof(float) * (size_t)B * T * 2 * 4 * H, s);
  return rnn_bwd<4>(dout, out_raw, cstate, gates, whhT_f, whhT_r, dgates, dgates, carry, lens, B, T, H, s);
}

int ft_fill_padded(const float* raw, const long* lens, float* out, int B, int T, int C, float pad, void* stream) {
  long total = (long)B * T * C;
  if (total <= 0) return FT_OK;
  hipLaunchKernelGGL(ft_fill_padded_kernel, dim3(ft_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, raw, lens,
                     out, B, T, C, pad);
  return ft_check_launch("fill_padded");
}

int ft_mask_rows(const float* src, const long* lens, float* dst, int B, int T, int C, void* stream) {
  long total = (long)B * T * C;
  if (total <= 0) return FT_OK;
  hipLaunchKernelGGL(ft_mask_rows_kernel, dim3(ft_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, src, lens, dst,
                     B, T, C);
  return ft_check_launch("mask_rows");
}

}  // extern "C"
