// Bidirectional GRU / (packed) LSTM recurrences for gfx950 -- nn.GRU / nn.LSTM semantics
// (models/common_layers.py:89,123 ; models/forward_tacotron.py:24,96-99,147-152).
//
// The input projections x*W_ih^T (+b_ih) are hoisted into one big MFMA GEMM (ft_linear_multi_fwd); what is
// left per timestep is h[B,H] * W_hh^T[H,G*H] plus the cell math.
//
// This file: the C entry points and the PER-STEP form (one launch per timestep covering both directions; the kernel
// boundary is the step-to-step dependency).  It is the fallback -- hidden sizes that are not a multiple of 16, a grid
// that would not be co-resident, FT_RNN_PERSISTENT=0; the normal path is the one-launch persistent form in
// ft_rnn_persist.hip (ft_rnn_{fwd,bwd}_persistent), tried first by rnn_fwd / rnn_bwd below.
//
// Per step the chip has to move W_hh (4 MB per direction for the 512-wide LSTM) plus the recurrent operand
// through the per-CU L1 path, so the decomposition minimises BYTES PER CU rather than MFMA count:
//   forward : workgroup = MT*16 batch rows x 16 gate rows (UB units, all G gates)  -> 256 workgroups
//             for the LSTM, 96 KB of operands each; K = H split over the NW waves of the workgroup
//   backward: workgroup = 16 batch rows x 16 hidden units, K = G*H split over NW waves (the 2048-wide
//             d(gates) row block is the expensive operand, so the batch is split, not the K range)
// Each wave prefetches its whole K slice into registers as 16-B lanes (f32 16x16x4 MFMA: lane (i,q) holds
// row i, k = 16c+4q..+3 -- any K order is legal as long as A and B agree), then runs the MFMA chain;
// per-wave partial 16x16 tiles are reduced through LDS and the same workgroup finishes the cell update.
// The cell-update operands (x projection, biases, previous state, saved gates) are requested BEFORE the
// matmul so their HBM latency hides under it.
//
// Time indexing (L_b = lens[b] for the packed LSTM, T otherwise); s = launch index:
//   forward pass : t = s (dir 0) | L_b-1-s (dir 1) ;  h_prev at t-1 | t+1           ; active iff s < L_b
//   backward pass: t = L_b-1-s (dir 0) | s (dir 1) ;  "next" (already done) t+1|t-1 ; h_prev at t-1 | t+1
#include <stdlib.h>

#include <string.h>

#include "ft_gemm.h"
#include "ft_rnn.h"

namespace {

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

// acc[m] (+)= A_m[16 x K] * B[16 x K]^T over this wave's K groups [g0, g1); arow[m]/brow are the lane's
// row pointers (row = lane&15), K contiguous.  Whole chunks of GCH groups are loaded before the MFMA chain.
// FAST (K % 16 == 0, 16-B aligned rows): every load is an unconditional 16-B lane load -- callers clamp the
// row pointer of an invalid lane to a valid row instead of predicating (row i of the product depends only on
// A row i, column j only on B row j, and the invalid ones are never consumed), so the loop is branch-free
// apart from the wave-uniform group-count test.  The generic path predicates every element (odd sizes).
template <int MT, bool FAST>
__device__ __forceinline__ void wave_matmul(const float* const (&arow)[MT], const bool (&aok)[MT], const float* brow,
                                            bool bok, int K, int g0, int g1, int q, bool vec, f32x4 (&acc)[MT]) {
  for (int gb = g0; gb < g1; gb += GCH) {
    float4 av[MT][GCH], bv[GCH];
#pragma unroll
    for (int c = 0; c < GCH; ++c) {
      const int k = 16 * (gb + c) + 4 * q;
      if (FAST) {
        if (gb + c < g1) {       // wave-uniform
          bv[c] = *reinterpret_cast<const float4*>(brow + k);
#pragma unroll
          for (int m = 0; m < MT; ++m) av[m][c] = *reinterpret_cast<const float4*>(arow[m] + k);
        }
      } else {
        const bool in = (gb + c) < g1;
        bv[c] = ld4g(brow + k, (in && bok) ? K - k : 0, vec);
#pragma unroll
        for (int m = 0; m < MT; ++m) av[m][c] = ld4g(arow[m] + k, (in && aok[m]) ? K - k : 0, vec);
      }
    }
#pragma unroll
    for (int c = 0; c < GCH; ++c) {
      if (gb + c < g1) {        // wave-uniform: skip the MFMAs of groups beyond this wave's slice
#pragma unroll
        for (int m = 0; m < MT; ++m) mfma4(av[m][c], bv[c], acc[m]);
      }
    }
  }
}

// workgroup: MT*16 batch rows x (UB = 16/G units, all G gates)
template <int G, int NW, int MT, bool FAST>
__global__ __launch_bounds__(NW * 64) void ft_rnn_fwd_step_kernel(RnnFwdArgs a) {
  constexpr int UB = 16 / G;
  __shared__ float red[NW * MT * 16 * RLD];
  const int d = blockIdx.z;
  const int u0 = blockIdx.x * UB;
  const int b0 = blockIdx.y * (MT * 16);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int q = lane >> 4, l15 = lane & 15;
  const int H = a.H, T = a.T, s = a.s;
  const long ldo = (long)a.ND * H;

  // ---- cell-update operands of this thread (requested now, consumed after the matmul)
  const int ci = tid / UB, cu = tid - ci * UB;
  const int cb = b0 + ci, cun = u0 + cu;
  bool cact = tid < MT * 16 * UB && cb < a.B && cun < H;
  int ct = 0;
  float xg[G], bg[G], prev = 0.f;
#pragma unroll
  for (int g = 0; g < G; ++g) xg[g] = bg[g] = 0.f;
  if (cact) {
    const int L = clamp_len(a.lens, cb, T);
    cact = s < L;
    if (cact) {
      ct = d == 0 ? s : L - 1 - s;
      const int tprev = d == 0 ? ct - 1 : ct + 1;
      const float* xr = a.xp + ((long)ct * a.B + cb) * ((long)a.ND * G * H) + (long)d * G * H + cun;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        xg[g] = xr[(long)g * H];
        bg[g] = a.bhh[d][g * H + cun];
      }
      if (s > 0) {
        const long op = ((long)tprev * a.B + cb) * ldo + (long)d * H + cun;
        prev = G == 3 ? a.out[op] : a.cst[op];
      }
    }
  }

  f32x4 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[m][e] = 0.f;

  if (s > 0) {
    const float* arow[MT];
    bool aok[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int bA = b0 + m * 16 + l15;
      const int LA = bA < a.B ? clamp_len(a.lens, bA, T) : 0;
      aok[m] = bA < a.B && s < LA;
      const int tprev = d == 0 ? s - 1 : LA - s;        // (t-1) | (t+1) with t = LA-1-s
      const long row = aok[m] ? (long)tprev * a.B + bA : 0;   // invalid lanes read row 0 (never consumed)
      arow[m] = a.out + row * ldo + (long)d * H;
    }
    const int gj = l15 / UB, ul = l15 - gj * UB;
    const bool bok = gj < G && (u0 + ul) < H;
    const float* brow = a.whh[d] + (bok ? ((long)gj * H + u0 + ul) * H : 0);
    const int ngroups = (H + 15) / 16;
    const int gpw = (ngroups + NW - 1) / NW;
    const int g0 = wave * gpw, g1 = min(ngroups, g0 + gpw);
    wave_matmul<MT, FAST>(arow, aok, brow, bok, H, g0, g1, q, a.vec, acc);
  }
  store_partials<MT>(red, wave, lane, acc);
  __syncthreads();

  if (cact) {
    float hp[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += red[w * (MT * 16 * RLD) + ci * RLD + g * UB + cu];
      hp[g] = v + bg[g];
    }
    const long o = ((long)ct * a.B + cb) * ldo + (long)d * H + cun;
    float* gs = a.gates ? a.gates + (((long)ct * a.B + cb) * a.ND + d) * 4 * H + cun : nullptr;
    if (G == 3) {
      float r = ft_sigmoid(xg[0] + hp[0]);
      float z = ft_sigmoid(xg[1] + hp[1]);
      float n = ft_tanh(xg[2] + r * hp[2]);
      a.out[o] = (1.f - z) * n + z * prev;
      if (gs) {
        gs[0] = r; gs[H] = z; gs[2 * H] = n; gs[3 * H] = hp[2];
      }
    } else {
      float ig = ft_sigmoid(xg[0] + hp[0]);
      float fg = ft_sigmoid(xg[1] + hp[1]);
      float gg = ft_tanh(xg[2] + hp[2]);
      float og = ft_sigmoid(xg[G - 1] + hp[G - 1]);
      float c = fg * prev + ig * gg;
      a.cst[o] = c;
      a.out[o] = og * ft_tanh(c);
      if (gs) {
        gs[0] = ig; gs[H] = fg; gs[2 * H] = gg; gs[3 * H] = og;
      }
    }
  }
}

// workgroup: 16 batch rows x 16 hidden units ; K = G*H split over NW waves
template <int G, int NW, bool FAST>
__global__ __launch_bounds__(NW * 64) void ft_rnn_bwd_step_kernel(RnnBwdArgs a) {
  __shared__ float red[NW * 16 * RLD];
  const int d = blockIdx.z;
  const int u0 = blockIdx.x * 16;
  const int b0 = blockIdx.y * 16;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int q = lane >> 4, l15 = lane & 15;
  const int H = a.H, T = a.T, s = a.s, K = G * H;
  const long ldg = (long)a.ND * K;
  const long ldo = (long)a.ND * H;

  // ---- cell-gradient operands of this thread (requested now, consumed after the matmul)
  const int ci = tid >> 4, cj = tid & 15;
  const int cb = b0 + ci, cun = u0 + cj;
  bool cact = tid < 256 && cb < a.B && cun < H;
  int ct = 0;
  float gv[4] = {0.f, 0.f, 0.f, 0.f}, dov = 0.f, cin = 0.f, cc = 0.f, prev = 0.f;
  if (cact) {
    const int L = clamp_len(a.lens, cb, T);
    cact = s < L;
    if (cact) {
      ct = d == 0 ? L - 1 - s : s;
      const int tprev = d == 0 ? ct - 1 : ct + 1;
      const bool has_prev = tprev >= 0 && tprev < L;
      const long o = ((long)ct * a.B + cb) * ldo + (long)d * H + cun;
      const long op = ((long)tprev * a.B + cb) * ldo + (long)d * H + cun;
      const float* gs = a.gates + (((long)ct * a.B + cb) * a.ND + d) * 4 * H + cun;
#pragma unroll
      for (int g = 0; g < 4; ++g) gv[g] = gs[(long)g * H];
      dov = a.dout[o];
      if (s > 0) cin = a.carry[((long)cb * a.ND + d) * H + cun];
      if (G == 3) {
        prev = has_prev ? a.out[op] : 0.f;
      } else {
        cc = a.cst[o];
        prev = has_prev ? a.cst[op] : 0.f;
      }
    }
  }

  f32x4 acc[1];
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[0][e] = 0.f;
  if (s > 0) {
    const int bA = b0 + l15;
    const int LA = bA < a.B ? clamp_len(a.lens, bA, T) : 0;
    const bool aok[1] = {bA < a.B && s < LA};
    const int tnext = d == 0 ? LA - s : s - 1;           // (t+1) with t=LA-1-s | (t-1) with t=s
    const long row = aok[0] ? (long)tnext * a.B + bA : 0;
    const float* arow[1] = {a.dhp + row * ldg + (long)d * K};
    const bool bok = (u0 + l15) < H;
    const float* brow = a.whhT[d] + (bok ? (long)(u0 + l15) * K : 0);
    const int ngroups = (K + 15) / 16;
    const int gpw = (ngroups + NW - 1) / NW;
    const int g0 = wave * gpw, g1 = min(ngroups, g0 + gpw);
    wave_matmul<1, FAST>(arow, aok, brow, bok, K, g0, g1, q, a.vec, acc);
  }
  store_partials<1>(red, wave, lane, acc);
  __syncthreads();

  if (cact) {
    float rec = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) rec += red[w * (16 * RLD) + ci * RLD + cj];
    float* cr = a.carry + ((long)cb * a.ND + d) * H + cun;
    float* dx = a.dxp + ((long)ct * a.B + cb) * ldg + (long)d * K + cun;
    if (G == 3) {
      const float dh = dov + rec + cin;
      const float r = gv[0], z = gv[1], n = gv[2], hn = gv[3];
      const float dz = dh * (prev - n) * z * (1.f - z);
      const float dn = dh * (1.f - z) * (1.f - n * n);
      const float dr = dn * hn * r * (1.f - r);
      dx[0] = dr; dx[H] = dz; dx[2 * H] = dn;
      float* dhh = a.dhp + ((long)ct * a.B + cb) * ldg + (long)d * K + cun;
      dhh[0] = dr; dhh[H] = dz; dhh[2 * H] = dn * r;
      *cr = dh * z;
    } else {
      const float dh = dov + rec;
      const float ig = gv[0], fg = gv[1], gg = gv[2], og = gv[3];
      const float tc = ft_tanh(cc);
      const float dc = dh * og * (1.f - tc * tc) + cin;
      dx[0] = dc * gg * ig * (1.f - ig);
      dx[H] = dc * prev * fg * (1.f - fg);
      dx[2 * H] = dc * ig * (1.f - gg * gg);
      dx[3 * H] = dh * tc * og * (1.f - og);
      *cr = dc * fg;
    }
  }
}

// out[b,t,:] = t < len[b] ? raw[b,t,:] : pad      (pad_packed_sequence padding_value, forward_tacotron.py:152)
// raw is TIME-major [T,B,C] (the recurrence's layout), out is batch-major [B,T,C]
__global__ void ft_fill_padded_kernel(const float* __restrict__ raw, const long* __restrict__ lens,
                                      float* __restrict__ out, int B, int T, int C, float pad) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)B * T * C;
  if (i >= total) return;
  long row = i / C;
  int c = (int)(i - row * C);
  int b = (int)(row / T), t = (int)(row - (long)b * T);
  out[i] = (!lens || t < lens[b]) ? raw[((long)t * B + b) * C + c] : pad;
}
// [B,T,C] <-> [T,B,C] row permutation (dst_time_major: dst is [T,B,C])
__global__ void ft_bt_transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int T, int C,
                                       int dst_time_major) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)B * T * C;
  if (i >= total) return;
  long row = i / C;
  int c = (int)(i - row * C);
  if (dst_time_major) {
    int t = (int)(row / B), b = (int)(row - (long)t * B);
    dst[i] = src[((long)b * T + t) * C + c];
  } else {
    int b = (int)(row / T), t = (int)(row - (long)b * T);
    dst[i] = src[((long)t * B + b) * C + c];
  }
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

template <int G, int NW, int MT>
void launch_fwd(bool fast, dim3 grid, hipStream_t stream, const RnnFwdArgs& a) {
  if (fast)
    hipLaunchKernelGGL((ft_rnn_fwd_step_kernel<G, NW, MT, true>), grid, dim3(NW * 64), 0, stream, a);
  else
    hipLaunchKernelGGL((ft_rnn_fwd_step_kernel<G, NW, MT, false>), grid, dim3(NW * 64), 0, stream, a);
}

template <int G>
int rnn_fwd(const float* xp, const float* whh_f, const float* whh_r, const float* bhh_f, const float* bhh_r,
            float* out, float* cst, float* gates, const long* lens, int B, int T, int H, void* ws,
            size_t ws_bytes, hipStream_t stream, const unsigned* gate = nullptr, int gate_cs = 0,
            hipEvent_t xp_complete = nullptr) {
  RnnFwdArgs a;
  a.xp = xp; a.whh[0] = whh_f; a.whh[1] = whh_r; a.bhh[0] = bhh_f; a.bhh[1] = bhh_r;
  a.out = out; a.cst = cst; a.gates = gates; a.lens = lens;
  a.B = B; a.T = T; a.H = H; a.ND = 2; a.Bld = B;
  a.gate = gate; a.gate_cs = gate_cs;
  a.vec = (H % 4 == 0) && ((uintptr_t)out % 16 == 0) && ((uintptr_t)whh_f % 16 == 0) && ((uintptr_t)whh_r % 16 == 0);
  const bool fast = a.vec && (H % 16 == 0);
  {
    const int rc = ft_rnn_fwd_persistent(G, a, ws, ws_bytes, stream);      // writes the zeros of finished items itself
    if (rc != -1) return rc;
  }
  // only the single persistent launch reads xp chunk by chunk: every other form waits for all of it
  if (gate) {
    if (xp_complete) (void)hipStreamWaitEvent(stream, xp_complete, 0);
    a.gate = nullptr;
    a.gate_cs = 0;
  }
  // A batch too large for ONE persistent grid (all workgroups must be co-resident: the 512-wide LSTM fills the chip at 64
  // rows) runs as several persistent launches over 64-row slices of the batch, one after the other -- batch rows are
  // independent -- instead of T per-step launches: the long-form inference batch (128 items x 6093 frames) spent 137 of
  // its 255 ms in 6093 per-step LSTM launches of 22 us each.
  constexpr int SLICE = 64;
  if (B > SLICE && a.vec && H % 16 == 0) {
    auto slice = [&](int bo) {
      RnnFwdArgs c = a;
      const long ldo = (long)a.ND * H;
      c.xp = a.xp + (long)bo * a.ND * G * H;
      c.out = a.out + bo * ldo;
      c.cst = a.cst ? a.cst + bo * ldo : nullptr;
      c.gates = a.gates ? a.gates + (long)bo * a.ND * 4 * H : nullptr;
      c.lens = a.lens ? a.lens + bo : nullptr;
      c.B = B - bo < SLICE ? B - bo : SLICE;
      return c;
    };
    int rc = ft_rnn_fwd_persistent(G, slice(0), ws, ws_bytes, stream);
    if (rc != -1) {
      for (int bo = SLICE; bo < B && rc == FT_OK; bo += SLICE) {
        rc = ft_rnn_fwd_persistent(G, slice(bo), ws, ws_bytes, stream);
        FT_REQUIRE(rc != -1, "rnn_fwd: a later batch slice was refused the persistent form the first one got");
      }
      return rc;
    }
  }
  if (lens) {   // inactive positions must read as zeros
    (void)hipMemsetAsync(out, 0, sizeof(float) * (size_t)B * T * 2 * H, stream);
    if (cst) (void)hipMemsetAsync(cst, 0, sizeof(float) * (size_t)B * T * 2 * H, stream);
  }
  constexpr int UB = 16 / G;
  const bool two = B > 16;                        // 32 batch rows per workgroup when there are that many
  dim3 grid(ft_cdiv(H, UB), ft_cdiv(B, two ? 32 : 16), 2);
  const int ngroups = ft_cdiv(H, 16);
  for (int s = 0; s < T; ++s) {
    a.s = s;
    if (ngroups > 16) {
      if (two) launch_fwd<G, 8, 2>(fast, grid, stream, a);
      else launch_fwd<G, 8, 1>(fast, grid, stream, a);
    } else {
      if (two) launch_fwd<G, 4, 2>(fast, grid, stream, a);
      else launch_fwd<G, 4, 1>(fast, grid, stream, a);
    }
  }
  return ft_check_launch("rnn_fwd");
}

template <int G, int NW>
void launch_bwd(bool fast, dim3 grid, hipStream_t stream, const RnnBwdArgs& a) {
  if (fast)
    hipLaunchKernelGGL((ft_rnn_bwd_step_kernel<G, NW, true>), grid, dim3(NW * 64), 0, stream, a);
  else
    hipLaunchKernelGGL((ft_rnn_bwd_step_kernel<G, NW, false>), grid, dim3(NW * 64), 0, stream, a);
}

template <int G>
int rnn_bwd(const float* dout, const float* out, const float* cst, const float* gates, const float* whhT_f,
            const float* whhT_r, float* dxp, float* dhp, float* carry, const long* lens, int B, int T, int H,
            void* ws, size_t ws_bytes, hipStream_t stream) {
  RnnBwdArgs a;
  a.dout = dout; a.out = out; a.cst = cst; a.gates = gates; a.whhT[0] = whhT_f; a.whhT[1] = whhT_r;
  a.dxp = dxp; a.dhp = dhp; a.carry = carry; a.lens = lens;
  a.B = B; a.T = T; a.H = H; a.ND = 2;
  a.vec = (H % 4 == 0) && ((uintptr_t)dhp % 16 == 0) && ((uintptr_t)whhT_f % 16 == 0) &&
          ((uintptr_t)whhT_r % 16 == 0);
  const bool fast = a.vec && (((long)G * H) % 16 == 0);
  {
    const int rc = ft_rnn_bwd_persistent(G, a, ws, ws_bytes, stream);      // writes the zeros of finished items itself
    if (rc != -1) return rc;
  }
  if (lens) {
    (void)hipMemsetAsync(dxp, 0, sizeof(float) * (size_t)B * T * 2 * G * H, stream);
    if (dhp != dxp) (void)hipMemsetAsync(dhp, 0, sizeof(float) * (size_t)B * T * 2 * G * H, stream);
  }
  dim3 grid(ft_cdiv(H, 16), ft_cdiv(B, 16), 2);
  const int ngroups = ft_cdiv((long)G * H, 16);
  for (int s = 0; s < T; ++s) {
    a.s = s;
    if (ngroups > 64) launch_bwd<G, 16>(fast, grid, stream, a);
    else if (ngroups > 16) launch_bwd<G, 8>(fast, grid, stream, a);
    else launch_bwd<G, 4>(fast, grid, stream, a);
  }
  return ft_check_launch("rnn_bwd");
}

#define FT_HIP_OK(call)                                   \
  do {                                                    \
    if ((call) != hipSuccess) {                           \
      ft_set_error("rnn layer: %s failed", #call);        \
      return FT_ERR_HIP;                                  \
    }                                                     \
  } while (0)

__global__ void ft_gate_set_kernel(unsigned* word, unsigned value) {
  __hip_atomic_store(word, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

hipEvent_t layer_event(int which) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  static hipEvent_t ev[16][2] = {};
  if (dev < 0 || dev >= 16) return nullptr;
  hipEvent_t& e = ev[dev][which];
  if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) e = nullptr;
  return e;
}

// A recurrent LAYER's forward with the input projection overlapped with the recurrence: x * W_ih^T + b_ih is formed in
// time chunks -- chunk 0 of both directions on `stream`, the others on `side` -- and the persistent recurrence, launched
// right behind chunk 0, takes each chunk when its gate word says it is complete (RnnFwdArgs.gate).  The recurrence
// leaves most of the chip idle (it is bound by the cross-CU hand-off, not by work), which is where the rest of the
// projection GEMM now runs instead of in front of it.  x is batch-major [B,T,In]; xp time-major [T,B,2*G*H].
// rev_lead: chunks of direction 1 issued before the alternation starts (a packed item of length L_b starts at
// t = L_b - 1, i.e. inside chunk (T - L_b) / cs of that direction, so its group waits for that many chunks at once).
template <int G>
int rnn_layer_fwd(const float* x, int In, const float* const* wih, const float* const* bih, float* xp,
                  const float* whh_f, const float* whh_r, const float* bhh_f, const float* bhh_r, const long* lens,
                  float* out, float* cst, float* gates, int B, int T, int H, void* ws, size_t ws_bytes, unsigned* gate,
                  int nchunks, int rev_lead, hipStream_t stream, hipStream_t side) {
  const long ldy = 2L * G * H;
  auto project = [&](int d, int t0, int n, hipStream_t st) {
    FtGemmBatch b;
    memset(&b, 0, sizeof(b));
    FtGemmTask& t = b.t[0];
    t.A = x + (long)t0 * In; t.B = wih[d]; t.C = xp + (long)t0 * B * ldy + (long)d * G * H; t.bias = bih ? bih[d] : nullptr;
    t.lda = In; t.ldb = In; t.ldc = ldy;
    t.M = B * n; t.N = G * H; t.K = In; t.taps = 1;
    t.amap = FtRowMap{n, T, 1, n, 0, 0};            // rows (b, t0 + t) of the batch-major input
    t.cmap = FtRowMap{n, 1, B, n, 0, 0};            // time-major output
    // the same kernel as ft_linear_multi_fwd's two-task launch over all B * T rows takes: bit-identical results
    b.force_tile = ft_rows_tile_is_big(2L * ft_cdiv((long)B * T, 128) * ft_cdiv(G * H, 128), B * T, G * H) ? 2 : 1;
    return ft_launch_gemm_rows(&b, 1, false, st);
  };
  int cs = nchunks > 1 ? (T + nchunks - 1) / nchunks : T;
  if (cs < 16) cs = 16;
  const int nc = T > 0 ? (T + cs - 1) / cs : 1;
  hipEvent_t ev0 = layer_event(0), ev1 = layer_event(1);
  // a recurrence that fills whole XCDs (the 512-wide LSTM) stops the dispatch of every other kernel while it is resident:
  // chunks launched beside it would only run after it -- and it would wait for them (ft_rnn_fwd_xcd_fill)
  const double fill = (nc >= 2 && side && gate) ? ft_rnn_fwd_xcd_fill(G, B, T, H, ws, ws_bytes) : -1.0;
  if (nc < 2 || !side || side == stream || !gate || !ev0 || !ev1 || fill < 0.0 || fill > 0.75) {
    int rc = project(0, 0, T, stream);
    if (rc == FT_OK) rc = project(1, 0, T, stream);
    if (rc != FT_OK) return rc;
    return rnn_fwd<G>(xp, whh_f, whh_r, bhh_f, bhh_r, out, cst, gates, lens, B, T, H, ws, ws_bytes, stream);
  }
  auto chunk = [&](int d, int j, hipStream_t st) {
    const int lo = d == 0 ? j * cs : (T - (j + 1) * cs > 0 ? T - (j + 1) * cs : 0);
    const int hi = d == 0 ? (lo + cs < T ? lo + cs : T) : T - j * cs;
    return project(d, lo, hi - lo, st);
  };
  (void)hipMemsetAsync(gate, 0, 2 * sizeof(unsigned), stream);
  int rc = chunk(0, 0, stream);
  if (rc == FT_OK) rc = chunk(1, 0, stream);
  if (rc != FT_OK) return rc;
  FT_HIP_OK(hipEventRecord(ev0, stream));
  FT_HIP_OK(hipStreamWaitEvent(side, ev0, 0));
  int next[2] = {1, 1};
  auto issue = [&](int d) {
    if (next[d] >= nc || rc != FT_OK) return;
    rc = chunk(d, next[d], side);
    hipLaunchKernelGGL(ft_gate_set_kernel, dim3(1), dim3(1), 0, side, gate + d, (unsigned)next[d]);
    ++next[d];
  };
  for (int i = 0; i < rev_lead; ++i) issue(1);
  while ((next[0] < nc || next[1] < nc) && rc == FT_OK) {
    issue(0);
    issue(1);
  }
  FT_HIP_OK(hipEventRecord(ev1, side));
  if (rc != FT_OK) {
    (void)hipStreamWaitEvent(stream, ev1, 0);
    return rc;
  }
  rc = rnn_fwd<G>(xp, whh_f, whh_r, bhh_f, bhh_r, out, cst, gates, lens, B, T, H, ws, ws_bytes, stream, gate, cs, ev1);
  (void)hipStreamWaitEvent(stream, ev1, 0);         // (already passed: the recurrence consumed every chunk)
  return rc;
}

}  // namespace

extern "C" {

int ft_lstm_layer_fwd(const float* x, int in_f, const float* wih_f, const float* wih_r, const float* bih_f,
                      const float* bih_r, float* xp, const float* whh_f, const float* whh_r, const float* bhh_f,
                      const float* bhh_r, const long* lens, float* out_raw, float* cstate, float* gates, int B, int T,
                      int H, void* workspace, size_t workspace_bytes, unsigned* gate, int nchunks, int rev_lead,
                      void* stream, void* side_stream) {
  FT_REQUIRE(B > 0 && T >= 0 && H > 0 && in_f > 0, "lstm_layer_fwd: bad dims");
  const float* wih[2] = {wih_f, wih_r};
  const float* bih[2] = {bih_f, bih_r};
  return rnn_layer_fwd<4>(x, in_f, wih, bih, xp, whh_f, whh_r, bhh_f, bhh_r, lens, out_raw, cstate, gates, B, T, H,
                          workspace, workspace_bytes, gate, nchunks, rev_lead, (hipStream_t)stream,
                          (hipStream_t)side_stream);
}

int ft_gru_layer_fwd(const float* x, int in_f, const float* wih_f, const float* wih_r, const float* bih_f,
                     const float* bih_r, float* xp, const float* whh_f, const float* whh_r, const float* bhh_f,
                     const float* bhh_r, float* out, float* gates, int B, int T, int H, void* workspace,
                     size_t workspace_bytes, unsigned* gate, int nchunks, void* stream, void* side_stream) {
  FT_REQUIRE(B > 0 && T >= 0 && H > 0 && in_f > 0, "gru_layer_fwd: bad dims");
  const float* wih[2] = {wih_f, wih_r};
  const float* bih[2] = {bih_f, bih_r};
  return rnn_layer_fwd<3>(x, in_f, wih, bih, xp, whh_f, whh_r, bhh_f, bhh_r, nullptr, out, nullptr, gates, B, T, H,
                          workspace, workspace_bytes, gate, nchunks, 0, (hipStream_t)stream, (hipStream_t)side_stream);
}

int ft_gru_fwd(const float* xp, const float* whh_f, const float* whh_r, const float* bhh_f, const float* bhh_r,
               float* out, float* gates, int B, int T, int H, void* workspace, size_t workspace_bytes,
               void* stream) {
  FT_REQUIRE(B > 0 && T >= 0 && H > 0, "gru_fwd: bad dims");
  return rnn_fwd<3>(xp, whh_f, whh_r, bhh_f, bhh_r, out, nullptr, gates, nullptr, B, T, H, workspace,
                    workspace_bytes, (hipStream_t)stream);
}

int ft_gru_bwd(const float* dout, const float* out, const float* gates, const float* whhT_f, const float* whhT_r,
               float* dxp, float* dhp, float* carry, int B, int T, int H, void* workspace, size_t workspace_bytes,
               void* stream) {
  FT_REQUIRE(B > 0 && T >= 0 && H > 0, "gru_bwd: bad dims");
  return rnn_bwd<3>(dout, out, nullptr, gates, whhT_f, whhT_r, dxp, dhp, carry, nullptr, B, T, H, workspace,
                    workspace_bytes, (hipStream_t)stream);
}

int ft_lstm_fwd(const float* xp, const float* whh_f, const float* whh_r, const float* bhh_f, const float* bhh_r,
                const long* lens, float* out_raw, float* cstate, float* gates, int B, int T, int H,
                void* workspace, size_t workspace_bytes, void* stream) {
  FT_REQUIRE(B > 0 && T >= 0 && H > 0, "lstm_fwd: bad dims");
  hipStream_t s = (hipStream_t)stream;
  return rnn_fwd<4>(xp, whh_f, whh_r, bhh_f, bhh_r, out_raw, cstate, gates, lens, B, T, H, workspace,
                    workspace_bytes, s);
}

int ft_lstm_bwd(const float* dout, const float* out_raw, const float* cstate, const float* gates,
                const float* whhT_f, const float* whhT_r, const long* lens, float* dgates, float* carry, int B, int T,
                int H, void* workspace, size_t workspace_bytes, void* stream) {
  FT_REQUIRE(B > 0 && T >= 0 && H > 0, "lstm_bwd: bad dims");
  hipStream_t s = (hipStream_t)stream;
  return rnn_bwd<4>(dout, out_raw, cstate, gates, whhT_f, whhT_r, dgates, dgates, carry, lens, B, T, H, workspace,
                    workspace_bytes, s);
}

int ft_fill_padded(const float* raw, const long* lens, float* out, int B, int T, int C, float pad, void* stream) {
  long total = (long)B * T * C;
  if (total <= 0) return FT_OK;
  hipLaunchKernelGGL(ft_fill_padded_kernel, dim3(ft_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, raw, lens,
                     out, B, T, C, pad);
  return ft_check_launch("fill_padded");
}

int ft_bt_transpose(const float* src, float* dst, int B, int T, int C, int dst_time_major, void* stream) {
  long total = (long)B * T * C;
  if (total <= 0) return FT_OK;
  hipLaunchKernelGGL(ft_bt_transpose_kernel, dim3(ft_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, B,
                     T, C, dst_time_major);
  return ft_check_launch("bt_transpose");
}

int ft_mask_rows(const float* src, const long* lens, float* dst, int B, int T, int C, void* stream) {
  long total = (long)B * T * C;
  if (total <= 0) return FT_OK;
  hipLaunchKernelGGL(ft_mask_rows_kernel, dim3(ft_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, src, lens, dst,
                     B, T, C);
  return ft_check_launch("mask_rows");
}

}  // extern "C"
