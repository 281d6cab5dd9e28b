// Shared pieces of the recurrence kernels (ft_rnn.hip: per-step launches + C entry points;
// ft_rnn_persist.hip: one-launch persistent forms).
#pragma once
#include "ft_common.h"

constexpr int RLD = 20;     // LDS row stride of the 16x16 partial tiles
constexpr int GCH = 8;      // K groups (16 k each) a wave keeps in registers

struct RnnFwdArgs {
  const float* xp;        // [T,B,ND*G*H]   x W_ih^T + b_ih   (time-major)
  const float* whh[2];    // [G*H,H]
  const float* bhh[2];    // [G*H]
  float* out;             // [T,B,ND*H]  raw hidden states (zero where inactive)
  float* cst;             // LSTM: [T,B,ND*H] cell states
  float* gates;           // optional [T,B,ND,4*H] saved activations (training)
  const long* lens;       // optional [B]
  int B, T, H, ND, s, vec;
  // rows per time step of the buffers (the persistent forward can be launched on a SLICE of the batch: B rows starting
  // at the pointers above, inside buffers whose time steps are Bld rows apart); the per-step kernels take Bld == B
  int Bld;
  // input gate (persistent forward only; null = off): the x projection arrives in TIME CHUNKS of gate_cs steps while the
  // recurrence already runs -- direction 0 in ascending t, direction 1 in descending t (chunk j of direction 1 covers
  // t in [T - (j+1)*gate_cs, T - j*gate_cs)).  gate[d] = index of the last chunk of direction d whose rows are complete
  // in memory (chunk 0 is complete at launch; the word is raised by ft_gate_set_kernel launches behind the chunks'
  // GEMMs on another stream).  A cell thread polls it (bounded, like every spin here) before it requests a row of a
  // chunk it has not seen complete, and reads xp past L1.
  const unsigned* gate;
  int gate_cs;
};

struct RnnBwdArgs {
  const float* dout;      // [T,B,ND*H]
  const float* out;       // [T,B,ND*H] raw hidden states
  const float* cst;       // LSTM cell states
  const float* gates;     // [T,B,ND,4H]
  const float* whhT[2];   // [H, G*H]  (W_hh transposed)
  float* dxp;             // [T,B,ND*G*H]  d(pre-activation) wrt the input projection
  float* dhp;             // GRU only: [T,B,ND*G*H] d wrt the hidden projection (n gate scaled by r); LSTM: == dxp
  float* carry;           // [B,ND,H]  GRU: dh*z ; LSTM: dc*f  (per-step kernels only)
  const long* lens;
  int B, T, H, ND, s, vec;
};

__device__ __forceinline__ void mfma4(const float4& a, const float4& b, f32x4& acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
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

// One-launch persistent recurrences (ft_rnn_persist.hip).  Return FT_OK if launched, -1 if the persistent form does
// not apply (the caller then issues the per-step kernels).
int ft_rnn_fwd_persistent(int G, RnnFwdArgs a, void* ws, size_t ws_bytes, hipStream_t stream);
// share of one XCD's CUs the persistent forward of this shape would hold (-1: it would not run persistent)
double ft_rnn_fwd_xcd_fill(int G, int B, int T, int H, void* ws, size_t ws_bytes);
double ft_rnn_bwd_xcd_fill(int G, int B, int T, int H, void* ws, size_t ws_bytes);
int ft_rnn_bwd_persistent(int G, RnnBwdArgs a, void* ws, size_t ws_bytes, hipStream_t stream);
