// Bidirectional GRU / (packed) LSTM recurrences for gfx950 -- nn.GRU / nn.LSTM semantics
// (models/common_layers.py:89,123 ; models/forward_tacotron.py:24,96-99,147-152).
//
// The input projections x*W_ih^T (+b_ih) are hoisted into one big MFMA GEMM (ft_linear_multi_fwd); what is
// left per timestep is h[B,H] * W_hh^T[H,G*H] plus the cell math.  The recurrence issues ONE launch per
// timestep covering both directions; the kernel boundary is the step-to-step dependency.
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

#include "ft_common.h"

namespace {

constexpr int RLD = 20;     // LDS row stride of the 16x16 partial tiles
constexpr int GCH = 8;      // K groups (16 k each) prefetched per chunk

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

__device__ __forceinline__ void mfma4(const float4& a, const float4& b, f32x4& acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
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

// partial tile store: lane holds column lane&15, rows (lane>>4)*4 + e
template <int MT>
__device__ __forceinline__ void store_partials(float* red, int wave, int lane, const f32x4 (&acc)[MT]) {
  float* r = red + wave * (MT * 16 * RLD);
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int e = 0; e < 4; ++e) r[(m * 16 + (lane >> 4) * 4 + e) * RLD + (lane & 15)] = acc[m][e];
}

__device__ __forceinline__ int clamp_len(const long* lens, int b, int T) {
  if (!lens) return T;
  long l = lens[b];
  return l < 0 ? 0 : (l > T ? T : (int)l);
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

// =================================================================================================
// PERSISTENT recurrences: one launch runs all T steps.
//
// Why: the XCD L2s do not keep W_hh across kernel boundaries, so the per-step launches above re-stream the
// recurrent weights from Infinity Cache every step (measured 14.8 MB/step, profiles/r01_pmc_rnn_step.txt).
// Here every wave keeps its W_hh fragments in REGISTERS for the whole sequence and only h (forward) or
// d(gates) (backward) travels between workgroups, through a small exchange buffer:
//   exchange layout  xb[parity][dir][batch group][k/4][MB][4]   (k = contraction index; a workgroup owns whole
//                    [MB][4] blocks = full 128-B lines, written by single wave-wide store instructions)
//   producer: write-through (sc1) stores -> every storing wave s_waitcnt vmcnt(0) -> workgroup barrier -> ONE
//             lane adds 1 to its arrival-counter shard (agent scope)
//   consumer: 8 lanes of wave 0 poll the 8 shards of (dir, batch group) with sc1 loads until each holds
//             step * (#producers of the shard); workgroup barrier; sc1 16-B loads of the operand rows
//             (bypass the CU's L1; the lines were dropped from every L2 by the write-through stores)
//   (cdna_hip_programming.md Guideline 16 R1 / MI355X_MICROARCH.md "Valid forms", sharded-counter row.)
// Step s reads parity (s-1)&1 and writes parity s&1; a workgroup can only be one step ahead of the slowest
// producer it depends on, so two parities suffice.  Every spin is bounded: on timeout the workgroup raises
// *err and leaves; all others then time out at the same step, so the grid always drains.
// All workgroups must be co-resident: the host checks the grid against the occupancy query and falls back
// to the per-step kernels otherwise (or when FT_RNN_PERSISTENT=0).
constexpr int NSH = 8;                 // arrival-counter shards per (direction, batch group)
constexpr int CSTRIDE = 32;            // one counter per 128-B line
constexpr unsigned MAX_SPINS = 1u << 18;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 ld_sc1_b128(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16);     // aux 16 = sc1
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// wave 0 waits until every shard k holds >= step * n_k arrivals; returns false on timeout (wave-uniform)
__device__ __forceinline__ bool wait_arrivals(const unsigned* cnt, unsigned step, int nprod, int lane) {
  bool ok = true;
  if (lane < NSH) {
    const unsigned nk = lane < nprod ? (unsigned)((nprod - lane + NSH - 1) / NSH) : 0u;
    const unsigned target = step * nk;
    unsigned spins = 0;
    while (__hip_atomic_load(cnt + lane * CSTRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > MAX_SPINS) {
        ok = false;
        break;
      }
    }
  }
  return __all(ok);
}

// workgroup: MT*16 batch rows x 4 hidden units (all G gates) ; grid (H/4, batch groups, 2 directions)
template <int G, int NW, int MT>
__global__ __launch_bounds__(NW * 64) void ft_rnn_fwd_persist_kernel(RnnFwdArgs a, float* xb, unsigned* cnt,
                                                                     unsigned* err, unsigned xb_bytes) {
  constexpr int UB = 4, MB = MT * 16;
  __shared__ float red[NW * MT * 16 * RLD];
  __shared__ int s_ok;
  const int d = blockIdx.z, chunk = blockIdx.x, bgp = blockIdx.y, nbg = gridDim.y;
  const int u0 = chunk * UB, b0 = bgp * MB;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int q = lane >> 4, l15 = lane & 15;
  const int H = a.H, T = a.T, nq = H / 4;
  const long ldo = (long)a.ND * H;
  const long par_floats = (long)2 * nbg * nq * MB * 4;                    // one parity
  const long base_floats = ((long)d * nbg + bgp) * nq * MB * 4;           // this (dir, batch group)
  unsigned* mycnt = cnt + ((long)d * nbg + bgp) * NSH * CSTRIDE;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(xb, 0, (int)xb_bytes, 0x00020000);

  // ---- resident W_hh fragments of this wave: column l15 = (gate l15/4, unit l15%4), K groups [g0,g1)
  const int gj = l15 / UB, ul = l15 - gj * UB;
  const float* brow = a.whh[d] + (gj < G ? ((long)gj * H + u0 + ul) * H : 0);
  const int ngroups = H / 16;
  const int gpw = (ngroups + NW - 1) / NW;
  const int g0 = wave * gpw, g1 = min(ngroups, g0 + gpw);
  float4 bv[GCH];
#pragma unroll
  for (int c = 0; c < GCH; ++c)
    bv[c] = (g0 + c < g1) ? *reinterpret_cast<const float4*>(brow + 16 * (g0 + c) + 4 * q) : make_float4(0, 0, 0, 0);

  // ---- cell thread: batch row ci, unit cu of this chunk
  const int ci = tid / UB, cu = tid - ci * UB;
  const int cb = b0 + ci, cun = u0 + cu;
  const bool cthr = tid < MB * UB && cb < a.B;
  const int L = cthr ? clamp_len(a.lens, cb, T) : 0;
  float bg[G];
#pragma unroll
  for (int g = 0; g < G; ++g) bg[g] = a.bhh[d][g * H + cun];
  float hprev = 0.f, cprev = 0.f;

  for (int s = 0; s < T; ++s) {
    const bool cact = cthr && s < L;
    const int ct = d == 0 ? s : L - 1 - s;
    float xg[G];
#pragma unroll
    for (int g = 0; g < G; ++g) xg[g] = 0.f;
    if (cact) {
      const float* xr = a.xp + ((long)ct * a.B + cb) * ((long)a.ND * G * H) + (long)d * G * H + cun;
#pragma unroll
      for (int g = 0; g < G; ++g) xg[g] = xr[(long)g * H];
    }
    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[m][e] = 0.f;

    if (s > 0) {
      if (wave == 0) {
        const bool ok = wait_arrivals(mycnt, (unsigned)s, nq, lane);
        if (lane == 0) s_ok = ok;
      }
      __syncthreads();
      if (!s_ok) {
        if (tid == 0) atomicExch(err, 1u);
        return;
      }
      const long rbase = (long)((s - 1) & 1) * par_floats + base_floats;
      float4 av[MT][GCH];
#pragma unroll
      for (int c = 0; c < GCH; ++c) {
        if (g0 + c < g1) {
          const long quad = 4 * (g0 + c) + q;
#pragma unroll
          for (int m = 0; m < MT; ++m)
            av[m][c] = ld_sc1_b128(rs, (unsigned)((rbase + (quad * MB + m * 16 + l15) * 4) * 4));
        }
      }
#pragma unroll
      for (int c = 0; c < GCH; ++c) {
        if (g0 + c < g1) {
#pragma unroll
          for (int m = 0; m < MT; ++m) mfma4(av[m][c], bv[c], acc[m]);
        }
      }
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
      float hnew;
      if (G == 3) {
        float r = ft_sigmoid(xg[0] + hp[0]);
        float z = ft_sigmoid(xg[1] + hp[1]);
        float n = ft_tanh(xg[2] + r * hp[2]);
        hnew = (1.f - z) * n + z * hprev;
        if (gs) {
          gs[0] = r; gs[H] = z; gs[2 * H] = n; gs[3 * H] = hp[2];
        }
      } else {
        float ig = ft_sigmoid(xg[0] + hp[0]);
        float fg = ft_sigmoid(xg[1] + hp[1]);
        float gg = ft_tanh(xg[2] + hp[2]);
        float og = ft_sigmoid(xg[G - 1] + hp[G - 1]);
        float c = fg * cprev + ig * gg;
        cprev = c;
        a.cst[o] = c;
        hnew = og * ft_tanh(c);
        if (gs) {
          gs[0] = ig; gs[H] = fg; gs[2 * H] = gg; gs[3 * H] = og;
        }
      }
      hprev = hnew;
      a.out[o] = hnew;
      // exchange block [chunk][MB][4] of parity s&1 (write-through)
      float* xw = xb + (long)(s & 1) * par_floats + base_floats + ((long)chunk * MB + ci) * 4 + cu;
      __hip_atomic_store(xw, hnew, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0)
      __hip_atomic_fetch_add(mycnt + (chunk % NSH) * CSTRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// workgroup: 16 batch rows x 16 hidden units, K = G*H over NW waves ; grid (H/16, B/16, 2)
template <int G, int NW>
__global__ __launch_bounds__(NW * 64) void ft_rnn_bwd_persist_kernel(RnnBwdArgs a, float* xb, unsigned* cnt,
                                                                     unsigned* err, unsigned xb_bytes) {
  constexpr int MB = 16;
  __shared__ float red[NW * 16 * RLD];
  __shared__ int s_ok;
  const int d = blockIdx.z, chunk = blockIdx.x, bgp = blockIdx.y, nbg = gridDim.y, nchunks = gridDim.x;
  const int u0 = chunk * 16, b0 = bgp * MB;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int q = lane >> 4, l15 = lane & 15;
  const int H = a.H, T = a.T, K = G * H, nq = K / 4;
  const long ldg = (long)a.ND * K;
  const long ldo = (long)a.ND * H;
  const long par_floats = (long)2 * nbg * nq * MB * 4;
  const long base_floats = ((long)d * nbg + bgp) * nq * MB * 4;
  unsigned* mycnt = cnt + ((long)d * nbg + bgp) * NSH * CSTRIDE;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(xb, 0, (int)xb_bytes, 0x00020000);

  // ---- resident W_hh^T fragments: column l15 = unit u0+l15 ; K groups [g0,g1)
  const bool bok = (u0 + l15) < H;
  const float* brow = a.whhT[d] + (bok ? (long)(u0 + l15) * K : 0);
  const int ngroups = K / 16;
  const int gpw = (ngroups + NW - 1) / NW;
  const int g0 = wave * gpw, g1 = min(ngroups, g0 + gpw);
  float4 bv[GCH];
#pragma unroll
  for (int c = 0; c < GCH; ++c)
    bv[c] = (g0 + c < g1) ? *reinterpret_cast<const float4*>(brow + 16 * (g0 + c) + 4 * q) : make_float4(0, 0, 0, 0);

  // ---- cell thread (first 256 threads): wave j4 owns units 4*j4..4*j4+3 -> whole [16][4] exchange blocks
  const int j4 = tid >> 6, ci = (tid >> 2) & 15, jj = tid & 3;
  const int cj = 4 * j4 + jj;                       // unit within the chunk
  const int cb = b0 + ci, cun = u0 + cj;
  const bool cthr = tid < 256 && cb < a.B && cun < H;
  const int L = cthr ? clamp_len(a.lens, cb, T) : 0;
  float carry = 0.f;

  for (int s = 0; s < T; ++s) {
    const bool cact = cthr && s < L;
    const int ct = d == 0 ? L - 1 - s : s;
    float gv[4] = {0.f, 0.f, 0.f, 0.f}, dov = 0.f, cc = 0.f, prev = 0.f;
    if (cact) {
      const int tprev = d == 0 ? ct - 1 : ct + 1;
      const bool has_prev = tprev >= 0 && tprev < L;
      const long o = ((long)ct * a.B + cb) * ldo + (long)d * H + cun;
      const long op = ((long)tprev * a.B + cb) * ldo + (long)d * H + cun;
      const float* gs = a.gates + (((long)ct * a.B + cb) * a.ND + d) * 4 * H + cun;
#pragma unroll
      for (int g = 0; g < 4; ++g) gv[g] = gs[(long)g * H];
      dov = a.dout[o];
      if (G == 3) {
        prev = has_prev ? a.out[op] : 0.f;
      } else {
        cc = a.cst[o];
        prev = has_prev ? a.cst[op] : 0.f;
      }
    }
    f32x4 acc[1];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[0][e] = 0.f;
    if (s > 0) {
      if (wave == 0) {
        const bool ok = wait_arrivals(mycnt, (unsigned)s, nchunks, lane);
        if (lane == 0) s_ok = ok;
      }
      __syncthreads();
      if (!s_ok) {
        if (tid == 0) atomicExch(err, 1u);
        return;
      }
      const long rbase = (long)((s - 1) & 1) * par_floats + base_floats;
      float4 av[GCH];
#pragma unroll
      for (int c = 0; c < GCH; ++c)
        if (g0 + c < g1) {
          const long quad = 4 * (g0 + c) + q;
          av[c] = ld_sc1_b128(rs, (unsigned)((rbase + (quad * MB + l15) * 4) * 4));
        }
#pragma unroll
      for (int c = 0; c < GCH; ++c)
        if (g0 + c < g1) mfma4(av[c], bv[c], acc[0]);
    }
    store_partials<1>(red, wave, lane, acc);
    __syncthreads();

    if (cact) {
      float rec = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) rec += red[w * (16 * RLD) + ci * RLD + cj];
      float* dx = a.dxp + ((long)ct * a.B + cb) * ldg + (long)d * K + cun;
      float dgx[G];
      if (G == 3) {
        const float dh = dov + rec + carry;
        const float r = gv[0], z = gv[1], n = gv[2], hn = gv[3];
        const float dz = dh * (prev - n) * z * (1.f - z);
        const float dn = dh * (1.f - z) * (1.f - n * n);
        const float dr = dn * hn * r * (1.f - r);
        dx[0] = dr; dx[H] = dz; dx[2 * H] = dn;
        float* dhh = a.dhp + ((long)ct * a.B + cb) * ldg + (long)d * K + cun;
        dhh[0] = dr; dhh[H] = dz; dhh[2 * H] = dn * r;
        dgx[0] = dr; dgx[1] = dz; dgx[2] = dn * r;
        carry = dh * z;
      } else {
        const float dh = dov + rec;
        const float ig = gv[0], fg = gv[1], gg = gv[2], og = gv[3];
        const float tc = ft_tanh(cc);
        const float dc = dh * og * (1.f - tc * tc) + carry;
        dgx[0] = dc * gg * ig * (1.f - ig);
        dgx[1] = dc * prev * fg * (1.f - fg);
        dgx[2] = dc * ig * (1.f - gg * gg);
        dgx[G - 1] = dh * tc * og * (1.f - og);
#pragma unroll
        for (int g = 0; g < G; ++g) dx[(long)g * H] = dgx[g];
        carry = dc * fg;
      }
      // exchange: k = g*H + cun -> block k/4 = (g*H + u0)/4 + j4, row ci, slot jj
      float* xw = xb + (long)(s & 1) * par_floats + base_floats;
#pragma unroll
      for (int g = 0; g < G; ++g)
        __hip_atomic_store(xw + (((long)(g * H + u0) / 4 + j4) * MB + ci) * 4 + jj, dgx[g], __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    } else if (cthr) {
      // inactive item: its exchange rows must read as zero for the workgroups that still multiply them
      float* xw = xb + (long)(s & 1) * par_floats + base_floats;
#pragma unroll
      for (int g = 0; g < G; ++g)
        __hip_atomic_store(xw + (((long)(g * H + u0) / 4 + j4) * MB + ci) * 4 + jj, 0.f, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0)
      __hip_atomic_fetch_add(mycnt + (chunk % NSH) * CSTRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

static int g_persistent = -1;      // -1: take FT_RNN_PERSISTENT from the environment
static bool persistent_enabled() {
  if (g_persistent < 0) {
    const char* e = getenv("FT_RNN_PERSISTENT");
    g_persistent = (e && e[0] == '0') ? 0 : 1;
  }
  return g_persistent == 1;
}

static int device_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
    if (cus <= 0) cus = 1;
  }
  return cus;
}

struct PersistWs {
  float* xb;
  unsigned* cnt;
  unsigned* err;
  size_t xb_bytes, total_bytes;
};
// workspace = [err + counters | exchange buffer]; both zeroed per call
static PersistWs carve_ws(void* ws, int nbg, int K, int MB) {
  PersistWs p;
  size_t cnt_bytes = (size_t)(1 + 2 * nbg * NSH) * CSTRIDE * sizeof(unsigned);
  p.err = (unsigned*)ws;
  p.cnt = p.err + CSTRIDE;
  p.xb_bytes = (size_t)2 * 2 * nbg * (K / 4) * MB * 4 * sizeof(float);
  p.xb = (float*)((char*)ws + cnt_bytes);
  p.total_bytes = cnt_bytes + p.xb_bytes;
  return p;
}

template <typename KernelT>
static bool grid_fits(KernelT kernel, int block, long nblocks) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, 0) != hipSuccess) return false;
  if (per_cu > 2) per_cu = 2;      // stay well inside what the dispatcher really admits
  return per_cu >= 1 && nblocks <= (long)per_cu * device_cus();
}

// returns FT_OK if launched, -1 if the persistent form does not apply (caller uses the per-step kernels)
template <int G>
int rnn_fwd_persistent(RnnFwdArgs a, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (!persistent_enabled() || !ws || a.T < 2) return -1;
  const int H = a.H, B = a.B;
  if (!a.vec || H % 16 != 0) return -1;
  const int ngroups = H / 16;
  const bool two = B > 16;
  const int MB = two ? 32 : 16;
  const int nbg = ft_cdiv(B, MB);
  const int NW = ngroups > 16 ? 8 : 4;
  if (ft_cdiv(ngroups, NW) > GCH) return -1;
  PersistWs p = carve_ws(ws, nbg, H, MB);
  if (ws_bytes < p.total_bytes || p.xb_bytes >= (1ull << 31)) return -1;
  dim3 grid(H / 4, nbg, 2);
  const long nblocks = (long)grid.x * grid.y * grid.z;
  bool fits;
  if (NW == 8) fits = two ? grid_fits(ft_rnn_fwd_persist_kernel<G, 8, 2>, 512, nblocks)
                          : grid_fits(ft_rnn_fwd_persist_kernel<G, 8, 1>, 512, nblocks);
  else fits = two ? grid_fits(ft_rnn_fwd_persist_kernel<G, 4, 2>, 256, nblocks)
                  : grid_fits(ft_rnn_fwd_persist_kernel<G, 4, 1>, 256, nblocks);
  if (!fits) return -1;
  (void)hipMemsetAsync(ws, 0, p.total_bytes, stream);
  a.s = 0;
  if (NW == 8) {
    if (two) hipLaunchKernelGGL((ft_rnn_fwd_persist_kernel<G, 8, 2>), grid, dim3(512), 0, stream, a, p.xb, p.cnt, p.err, (unsigned)p.xb_bytes);
    else hipLaunchKernelGGL((ft_rnn_fwd_persist_kernel<G, 8, 1>), grid, dim3(512), 0, stream, a, p.xb, p.cnt, p.err, (unsigned)p.xb_bytes);
  } else {
    if (two) hipLaunchKernelGGL((ft_rnn_fwd_persist_kernel<G, 4, 2>), grid, dim3(256), 0, stream, a, p.xb, p.cnt, p.err, (unsigned)p.xb_bytes);
    else hipLaunchKernelGGL((ft_rnn_fwd_persist_kernel<G, 4, 1>), grid, dim3(256), 0, stream, a, p.xb, p.cnt, p.err, (unsigned)p.xb_bytes);
  }
  return ft_check_launch("rnn_fwd_persistent");
}

template <int G>
int rnn_bwd_persistent(RnnBwdArgs a, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (!persistent_enabled() || !ws || a.T < 2) return -1;
  const int H = a.H, B = a.B, K = G * H;
  if (!a.vec || H % 16 != 0) return -1;
  const int ngroups = K / 16;
  const int nbg = ft_cdiv(B, 16);
  const int NW = ngroups > 64 ? 16 : (ngroups > 16 ? 8 : 4);
  if (ft_cdiv(ngroups, NW) > GCH) return -1;
  PersistWs p = carve_ws(ws, nbg, K, 16);
  if (ws_bytes < p.total_bytes || p.xb_bytes >= (1ull << 31)) return -1;
  dim3 grid(H / 16, nbg, 2);
  const long nblocks = (long)grid.x * grid.y * grid.z;
  bool fits = NW == 16 ? grid_fits(ft_rnn_bwd_persist_kernel<G, 16>, 1024, nblocks)
              : NW == 8 ? grid_fits(ft_rnn_bwd_persist_kernel<G, 8>, 512, nblocks)
                        : grid_fits(ft_rnn_bwd_persist_kernel<G, 4>, 256, nblocks);
  if (!fits) return -1;
  (void)hipMemsetAsync(ws, 0, p.total_bytes, stream);
  a.s = 0;
  if (NW == 16) hipLaunchKernelGGL((ft_rnn_bwd_persist_kernel<G, 16>), grid, dim3(1024), 0, stream, a, p.xb, p.cnt, p.err, (unsigned)p.xb_bytes);
  else if (NW == 8) hipLaunchKernelGGL((ft_rnn_bwd_persist_kernel<G, 8>), grid, dim3(512), 0, stream, a, p.xb, p.cnt, p.err, (unsigned)p.xb_bytes);
  else hipLaunchKernelGGL((ft_rnn_bwd_persist_kernel<G, 4>), grid, dim3(256), 0, stream, a, p.xb, p.cnt, p.err, (unsigned)p.xb_bytes);
  return ft_check_launch("rnn_bwd_persistent");
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
            size_t ws_bytes, hipStream_t stream) {
  RnnFwdArgs a;
  a.xp = xp; a.whh[0] = whh_f; a.whh[1] = whh_r; a.bhh[0] = bhh_f; a.bhh[1] = bhh_r;
  a.out = out; a.cst = cst; a.gates = gates; a.lens = lens;
  a.B = B; a.T = T; a.H = H; a.ND = 2;
  a.vec = (H % 4 == 0) && ((uintptr_t)out % 16 == 0) && ((uintptr_t)whh_f % 16 == 0) && ((uintptr_t)whh_r % 16 == 0);
  const bool fast = a.vec && (H % 16 == 0);
  {
    const int rc = rnn_fwd_persistent<G>(a, ws, ws_bytes, stream);
    if (rc != -1) return rc;
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
    const int rc = rnn_bwd_persistent<G>(a, ws, ws_bytes, stream);
    if (rc != -1) return rc;
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

}  // namespace

extern "C" {

size_t ft_rnn_workspace(int gates, int B, int H) {
  if (gates < 3 || gates > 4 || B <= 0 || H <= 0 || H % 16 != 0) return 0;
  const int MBf = B > 16 ? 32 : 16;
  PersistWs f = carve_ws(nullptr, ft_cdiv(B, MBf), H, MBf);
  PersistWs b = carve_ws(nullptr, ft_cdiv(B, 16), gates * H, 16);
  return f.total_bytes > b.total_bytes ? f.total_bytes : b.total_bytes;
}

int ft_rnn_set_persistent(int enabled) {
  int old = persistent_enabled() ? 1 : 0;
  g_persistent = enabled ? 1 : 0;
  return old;
}

int ft_rnn_status(const void* workspace, void* stream) {
  if (!workspace) return 0;
  unsigned flag = 0;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess ||
      hipMemcpy(&flag, workspace, sizeof(flag), hipMemcpyDeviceToHost) != hipSuccess) {
    ft_set_error("rnn_status: HIP error while reading the status word");
    return FT_ERR_HIP;
  }
  if (flag != 0) {
    ft_set_error("persistent recurrence timed out waiting for another workgroup (grid not co-resident?); "
                 "set FT_RNN_PERSISTENT=0 to use the per-step kernels");
    return FT_ERR_HIP;
  }
  return FT_OK;
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
  if (lens) {   // inactive positions must read as zeros
    (void)hipMemsetAsync(out_raw, 0, sizeof(float) * (size_t)B * T * 2 * H, s);
    (void)hipMemsetAsync(cstate, 0, sizeof(float) * (size_t)B * T * 2 * H, s);
  }
  return rnn_fwd<4>(xp, whh_f, whh_r, bhh_f, bhh_r, out_raw, cstate, gates, lens, B, T, H, workspace,
                    workspace_bytes, s);
}

int ft_lstm_bwd(const float* dout, const float* out_raw, const float* cstate, const float* gates,
                const float* whhT_f, const float* whhT_r, const long* lens, float* dgates, float* carry, int B, int T,
                int H, void* workspace, size_t workspace_bytes, void* stream) {
  FT_REQUIRE(B > 0 && T >= 0 && H > 0, "lstm_bwd: bad dims");
  hipStream_t s = (hipStream_t)stream;
  if (lens) (void)hipMemsetAsync(dgates, 0, sizeof(float) * (size_t)B * T * 2 * 4 * H, s);
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
