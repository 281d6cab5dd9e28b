// REJECTED EXPERIMENT (kept for the record, not built): measured 6.2 / 7.4 us per step (GRU-128 fwd / bwd, B=32) against
// 2.4 / 2.8 for the multi-workgroup persistent kernels.  The matrix work fits one CU, but the cell math does not: 16 rows x
// 128 units x ~150 VALU instructions (accurate expf / tanhf) is ~2 us on the four SIMDs of ONE CU, where the persistent
// form spreads it over 16 CUs; add the un-hidden HBM latency of the per-step operands and the hand-off it saves is dwarfed.
// SOLO recurrences: small bidirectional GRU / LSTM layers (G*H <= 384, H <= 128 -- the postnet CBHG's GRU-128 over 841
// frames, the predictors' GRU-64) run with ONE workgroup per (direction, 16 batch rows) and no global hand-off at all.
//
// Why: the persistent kernels of ft_rnn_persist.hip spread a layer over many CUs and pay ~2.4 us per step for the
// cross-workgroup exchange (store -> vmcnt(0) -> arrival counter -> poll -> load) however small H is; GRU-128 ran at
// 2.7 (forward) / 3.2 us (backward) per step, i.e. 5 ms of the 27 ms train step for 0.3 % of its FLOPs.  At H = 128 the
// whole W_hh (3 x 128 x 128) fits the registers of one workgroup as pre-split bf16 (hi, mid, lo) MFMA fragments
// (288 of the 512 unified registers a lane has at ONE wave per SIMD: 4 waves per workgroup), h / d(gates) only ever travel through LDS, and a step costs two workgroup barriers:
//   phase 1 (all 4 waves)   A fragments (already split, from LDS) x resident W fragments -> six v_mfma_f32_16x16x32_bf16
//                           per 32-k block (exact 3-way split, fp32 accumulate: same arithmetic as the persistent B3
//                           kernels) -> fp32 result tile to LDS
//   phase 2 (all 256 lanes) 8 (row, unit) cells per lane: nonlinearities / gate gradients, outputs to HBM, the new
//                           h (forward) or d(gates) (backward) split once into the three bf16 planes in LDS
// The step's HBM operands (x projection; saved gates, upstream gradient) are requested before phase 1 so that the
// MFMAs hide their latency.  The matrix pipe bounds a step (16 x 384 x 128 x 2 x 6 bf16 FLOP on one CU = 0.96 us);
// only 2 x ceil(B/16) CUs are busy, which leaves the rest of the chip to the weight-gradient stream.
#include <stdlib.h>

#include "ft_rnn.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

constexpr int NW = 4;               // waves per workgroup: one per SIMD, so that a wave may use all 512 registers
constexpr int MAXT = 6;             // forward: column tiles (16 gate columns) per wave -> G*H <= 384
constexpr int MAXKB = 4;            // forward: 32-k blocks of H                       -> H <= 128
constexpr int BT = 2;               // backward: output tiles (16 hidden units) per wave -> H <= 128
constexpr int BKB = 12;             // backward: 32-k blocks of G*H                     -> G*H <= 384
constexpr int PAIRS = 8;            // (row, unit) cells per lane: 16 * 128 / 256
#define FT_SOLO_KERNEL __global__ __launch_bounds__(NW * 64, 1) __attribute__((amdgpu_waves_per_eu(1, 1)))

__device__ __forceinline__ void split8(const float4& v0, const float4& v1, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
  const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
  u16x8 h, m, l;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const unsigned hb = __float_as_uint(f[i]) & 0xFFFF0000u;
    const float r1 = f[i] - __uint_as_float(hb);
    const unsigned mb = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(mb);
    h[i] = (unsigned short)(hb >> 16);
    m[i] = (unsigned short)(mb >> 16);
    l[i] = (unsigned short)(__float_as_uint(r2) >> 16);
  }
  hi = __builtin_bit_cast(bf16x8, h);
  mid = __builtin_bit_cast(bf16x8, m);
  lo = __builtin_bit_cast(bf16x8, l);
}
// one value -> its three planes (same truncation split)
__device__ __forceinline__ void split1(float x, unsigned short& hi, unsigned short& mid, unsigned short& lo) {
  const unsigned hb = __float_as_uint(x) & 0xFFFF0000u;
  const float r1 = x - __uint_as_float(hb);
  const unsigned mb = __float_as_uint(r1) & 0xFFFF0000u;
  const float r2 = r1 - __uint_as_float(mb);
  hi = (unsigned short)(hb >> 16);
  mid = (unsigned short)(mb >> 16);
  lo = (unsigned short)(__float_as_uint(r2) >> 16);
}
// the six product terms of the split, small ones first: (A plane, B plane)
__device__ constexpr int TERM_A[6] = {2, 0, 1, 1, 0, 0};
__device__ constexpr int TERM_B[6] = {0, 2, 1, 0, 1, 0};

// ---------------------------------------------------------------------------------------------------
// forward: gates_rec[16][G*H] = h[16][H] * W_hh^T ; wave w owns column tiles MAXT*w .. MAXT*w + MAXT-1
// LDS: hs = 3 planes [16][HS] bf16 (split h), gr = [16][GS] fp32 (recurrent pre-activations)
// ---------------------------------------------------------------------------------------------------
template <int G>
FT_SOLO_KERNEL void ft_rnn_fwd_solo_kernel(RnnFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int H = a.H, T = a.T, GH = G * H;
  const int HS = H + 8, GS = GH + 4;
  unsigned short* hs = reinterpret_cast<unsigned short*>(smem);
  float* gr = reinterpret_cast<float*>(smem + (size_t)3 * 16 * HS * sizeof(unsigned short));
  float* bs = gr + 16 * GS;                       // b_hh of this direction [G*H]
  int* Ls = reinterpret_cast<int*>(bs + GH);      // clamped lengths of the 16 rows
  const int hsh = 31 - __clz(H);                  // H is a power of two (host check): cell f -> (f >> hsh, f & (H-1))
  const int d = blockIdx.y, b0 = blockIdx.x * 16;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int q = lane >> 4, l15 = lane & 15;
  const int nkb = H / 32, ntiles = GH / 16;
  const long ldo = (long)a.ND * H;
  const long ldx = (long)a.ND * GH;

  // resident W_hh fragments: tile t -> gate column 16 t + l15 = row of W_hh [G*H, H]; lane (col, q) holds k = 32 kb + 8 q ..
  bf16x8 bw[MAXT][MAXKB][3];
#pragma unroll
  for (int nt = 0; nt < MAXT; ++nt) {
    const int tile = wave * MAXT + nt;
#pragma unroll
    for (int kb = 0; kb < MAXKB; ++kb) {
      const bool in = tile < ntiles && kb < nkb;
      const float* p = a.whh[d] + (in ? (long)(16 * tile + l15) * H + 32 * kb + 8 * q : 0);
      const float4 z = make_float4(0, 0, 0, 0);
      split8(in ? *reinterpret_cast<const float4*>(p) : z, in ? *reinterpret_cast<const float4*>(p + 4) : z,
             bw[nt][kb][0], bw[nt][kb][1], bw[nt][kb][2]);
    }
  }
  for (int i = tid; i < 3 * 16 * HS; i += NW * 64) hs[i] = 0;          // h_0 = 0

  // cells of this lane: pair i -> flat index f = tid + 256 i over [16 rows][H units] (f < 16 H, else idle)
  float hprev[PAIRS], cprev[PAIRS];
#pragma unroll
  for (int i = 0; i < PAIRS; ++i) {
    hprev[i] = 0.f;
    cprev[i] = 0.f;
  }
  for (int i = tid; i < GH; i += NW * 64) bs[i] = a.bhh[d][i];
  if (tid < 16) Ls[tid] = (b0 + tid < a.B) ? clamp_len(a.lens, b0 + tid, T) : 0;
  __syncthreads();

  for (int s = 0; s < T; ++s) {
    // ---- this step's x projection, requested now, consumed after the MFMA phase
    float xg[PAIRS][G];
#pragma unroll
    for (int i = 0; i < PAIRS; ++i) {
      const int f = tid + NW * 64 * i, row = (f >> hsh) & 15, un = f & (H - 1), L = f < 16 * H ? Ls[row] : 0;
#pragma unroll
      for (int g = 0; g < G; ++g) xg[i][g] = 0.f;
      if (s < L) {
        const int ct = d == 0 ? s : L - 1 - s;
        const float* xr = a.xp + ((long)ct * a.B + b0 + row) * ldx + (long)d * GH + un;
#pragma unroll
        for (int g = 0; g < G; ++g) xg[i][g] = xr[(long)g * H];
      }
    }
    // ---- phase 1: recurrent pre-activations of this wave's column tiles
    f32x4 acc[MAXT];
#pragma unroll
    for (int nt = 0; nt < MAXT; ++nt)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[nt][e] = 0.f;
    if (s > 0) {
#pragma unroll
      for (int kb = 0; kb < MAXKB; ++kb)
        if (kb < nkb) {
          bf16x8 a3[3];
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            a3[pl] = *reinterpret_cast<const bf16x8*>(hs + ((size_t)pl * 16 + l15) * HS + 32 * kb + 8 * q);
#pragma unroll
          for (int t = 0; t < 6; ++t)          // term-major: consecutive MFMAs feed different accumulators
#pragma unroll
            for (int nt = 0; nt < MAXT; ++nt)
              acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[TERM_A[t]], bw[nt][kb][TERM_B[t]], acc[nt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int nt = 0; nt < MAXT; ++nt) {
      const int tile = wave * MAXT + nt;
      if (tile < ntiles) {
#pragma unroll
        for (int e = 0; e < 4; ++e) gr[(4 * q + e) * GS + 16 * tile + l15] = acc[nt][e];
      }
    }
    __syncthreads();

    // ---- phase 2: cells
#pragma unroll
    for (int i = 0; i < PAIRS; ++i) {
      const int f = tid + NW * 64 * i, row = (f >> hsh) & 15, un = f & (H - 1);
      const bool in = f < 16 * H;
      const int L = in ? Ls[row] : 0;
      const bool act = s < L;
      float hnew = 0.f;
      if (act) {
        const int ct = d == 0 ? s : L - 1 - s;
        float hp[G];
#pragma unroll
        for (int g = 0; g < G; ++g) hp[g] = gr[row * GS + g * H + un] + bs[g * H + un];
        float sg[4], cnew = 0.f;
        if (G == 3) {
          const float r = ft_sigmoid(xg[i][0] + hp[0]);
          const float z = ft_sigmoid(xg[i][1] + hp[1]);
          const float n = ft_tanh(xg[i][2] + r * hp[2]);
          hnew = (1.f - z) * n + z * hprev[i];
          sg[0] = r; sg[1] = z; sg[2] = n; sg[3] = hp[2];
        } else {
          const float ig = ft_sigmoid(xg[i][0] + hp[0]);
          const float fg = ft_sigmoid(xg[i][1] + hp[1]);
          const float gg = ft_tanh(xg[i][2] + hp[2]);
          const float og = ft_sigmoid(xg[i][G - 1] + hp[G - 1]);
          cnew = fg * cprev[i] + ig * gg;
          cprev[i] = cnew;
          hnew = og * ft_tanh(cnew);
          sg[0] = ig; sg[1] = fg; sg[2] = gg; sg[3] = og;
        }
        hprev[i] = hnew;
        const long o = ((long)ct * a.B + b0 + row) * ldo + (long)d * H + un;
        a.out[o] = hnew;
        if (G == 4) a.cst[o] = cnew;
        if (a.gates) {
          float* gs = a.gates + (((long)ct * a.B + b0 + row) * a.ND + d) * 4 * H + un;
          gs[0] = sg[0]; gs[H] = sg[1]; gs[2 * H] = sg[2]; gs[3 * H] = sg[3];
        }
      }
      if (in) {        // a finished item's row is carried as zeros
        unsigned short h0, h1, h2;
        split1(hnew, h0, h1, h2);
        unsigned short* w = hs + (size_t)row * HS + un;
        w[0] = h0; w[(size_t)16 * HS] = h1; w[(size_t)32 * HS] = h2;
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------
// backward: dh_rec[16][H] = dgh[16][G*H] * W_hh ; wave w owns output tile w (16 hidden units), K = G*H
// LDS: ds = 3 planes [16][DS] bf16 (split d(gates) of the hidden projection), rr = [16][RS_] fp32 (dh_rec)
// ---------------------------------------------------------------------------------------------------
template <int G>
FT_SOLO_KERNEL void ft_rnn_bwd_solo_kernel(RnnBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int H = a.H, T = a.T, K = G * H;
  const int DS = K + 8, RS_ = H + 4;
  unsigned short* ds = reinterpret_cast<unsigned short*>(smem);
  float* rr = reinterpret_cast<float*>(smem + (size_t)3 * 16 * DS * sizeof(unsigned short));
  int* Ls = reinterpret_cast<int*>(rr + 16 * RS_);
  const int hsh = 31 - __clz(H);
  const int d = blockIdx.y, b0 = blockIdx.x * 16;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int q = lane >> 4, l15 = lane & 15;
  const int nkb = K / 32, ntiles = H / 16;
  const long ldg = (long)a.ND * K;
  const long ldo = (long)a.ND * H;

  // resident fragments: tile bt -> output unit n = 16 (BT wave + bt) + l15 ; lane (n, q) holds W_hh[k..k+7][n] =
  // whhT[n][k..k+7]
  bf16x8 bw[BT][BKB][3];
#pragma unroll
  for (int bt = 0; bt < BT; ++bt) {
    const int tile = wave * BT + bt;
#pragma unroll
    for (int kb = 0; kb < BKB; ++kb) {
      const bool in = tile < ntiles && kb < nkb;
      const float* p = a.whhT[d] + (in ? (long)(16 * tile + l15) * K + 32 * kb + 8 * q : 0);
      const float4 z = make_float4(0, 0, 0, 0);
      split8(in ? *reinterpret_cast<const float4*>(p) : z, in ? *reinterpret_cast<const float4*>(p + 4) : z,
             bw[bt][kb][0], bw[bt][kb][1], bw[bt][kb][2]);
    }
  }
  for (int i = tid; i < 3 * 16 * DS; i += NW * 64) ds[i] = 0;

  float carry[PAIRS];
#pragma unroll
  for (int i = 0; i < PAIRS; ++i) carry[i] = 0.f;
  if (tid < 16) Ls[tid] = (b0 + tid < a.B) ? clamp_len(a.lens, b0 + tid, T) : 0;
  __syncthreads();

  for (int s = 0; s < T; ++s) {
    // ---- saved activations / upstream gradient of this step, requested before the MFMA phase
    float gv[PAIRS][4], dov[PAIRS], cc[PAIRS], prev[PAIRS];
#pragma unroll
    for (int i = 0; i < PAIRS; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) gv[i][g] = 0.f;
      dov[i] = 0.f; cc[i] = 0.f; prev[i] = 0.f;
      const int f = tid + NW * 64 * i, row = (f >> hsh) & 15, L = f < 16 * H ? Ls[row] : 0;
      if (s < L) {
        const int cb = b0 + row, cun = f & (H - 1);
        const int ct = d == 0 ? L - 1 - s : s;
        const int tprev = d == 0 ? ct - 1 : ct + 1;
        const bool has_prev = tprev >= 0 && tprev < L;
        const long o = ((long)ct * a.B + cb) * ldo + (long)d * H + cun;
        const long op = ((long)tprev * a.B + cb) * ldo + (long)d * H + cun;
        const float* gs = a.gates + (((long)ct * a.B + cb) * a.ND + d) * 4 * H + cun;
#pragma unroll
        for (int g = 0; g < 4; ++g) gv[i][g] = gs[(long)g * H];
        dov[i] = a.dout[o];
        if (G == 3) {
          prev[i] = has_prev ? a.out[op] : 0.f;
        } else {
          cc[i] = a.cst[o];
          prev[i] = has_prev ? a.cst[op] : 0.f;
        }
      }
    }
    // ---- phase 1: recurrent part of d(h): two K halves x BT tiles = four independent accumulator chains
    f32x4 acc[BT][2];
#pragma unroll
    for (int bt = 0; bt < BT; ++bt)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[bt][hf][e] = 0.f;
    if (s > 0) {
#pragma unroll
      for (int kb = 0; kb < BKB; ++kb)
        if (kb < nkb) {
          bf16x8 a3[3];
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            a3[pl] = *reinterpret_cast<const bf16x8*>(ds + ((size_t)pl * 16 + l15) * DS + 32 * kb + 8 * q);
#pragma unroll
          for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int bt = 0; bt < BT; ++bt)
              acc[bt][kb & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[TERM_A[t]], bw[bt][kb][TERM_B[t]],
                                                                        acc[bt][kb & 1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) {
      const int tile = wave * BT + bt;
      if (tile < ntiles) {
#pragma unroll
        for (int e = 0; e < 4; ++e) rr[(4 * q + e) * RS_ + 16 * tile + l15] = acc[bt][0][e] + acc[bt][1][e];
      }
    }
    __syncthreads();

    // ---- phase 2: gate gradients
#pragma unroll
    for (int i = 0; i < PAIRS; ++i) {
      const int f = tid + NW * 64 * i, row = (f >> hsh) & 15, un = f & (H - 1);
      const bool in = f < 16 * H;
      const int L = in ? Ls[row] : 0;
      const bool act = s < L;
      float dgx[4] = {0.f, 0.f, 0.f, 0.f}, dgh2 = 0.f;
      if (act) {
        const float rec = rr[row * RS_ + un];
        if (G == 3) {
          const float dh = dov[i] + rec + carry[i];
          const float r = gv[i][0], z = gv[i][1], n = gv[i][2], hn = gv[i][3];
          const float dz = dh * (prev[i] - n) * z * (1.f - z);
          const float dn = dh * (1.f - z) * (1.f - n * n);
          const float dr = dn * hn * r * (1.f - r);
          dgx[0] = dr; dgx[1] = dz; dgx[2] = dn;
          dgh2 = dn * r;
          carry[i] = dh * z;
        } else {
          const float dh = dov[i] + rec;
          const float ig = gv[i][0], fg = gv[i][1], gg = gv[i][2], og = gv[i][3];
          const float tc = ft_tanh(cc[i]);
          const float dc = dh * og * (1.f - tc * tc) + carry[i];
          dgx[0] = dc * gg * ig * (1.f - ig);
          dgx[1] = dc * prev[i] * fg * (1.f - fg);
          dgx[2] = dc * ig * (1.f - gg * gg);
          dgx[G - 1] = dh * tc * og * (1.f - og);
          dgh2 = dgx[2];
          carry[i] = dc * fg;
        }
        const int ct = d == 0 ? L - 1 - s : s;
        float* dx = a.dxp + ((long)ct * a.B + b0 + row) * ldg + (long)d * K + un;
#pragma unroll
        for (int g = 0; g < G; ++g) dx[(long)g * H] = dgx[g];
        if (G == 3) {
          float* dhh = a.dhp + ((long)ct * a.B + b0 + row) * ldg + (long)d * K + un;
          dhh[0] = dgx[0]; dhh[H] = dgx[1]; dhh[2 * H] = dgh2;
        }
      }
      if (in) {        // a finished item's rows are carried as zeros
#pragma unroll
        for (int g = 0; g < G; ++g) {
          unsigned short h0, h1, h2;
          split1((G == 3 && g == 2) ? dgh2 : dgx[g], h0, h1, h2);
          unsigned short* w = ds + (size_t)row * DS + g * H + un;
          w[0] = h0; w[(size_t)16 * DS] = h1; w[(size_t)32 * DS] = h2;
        }
      }
    }
    __syncthreads();
  }
}

bool solo_enabled() {
  static const bool on = [] {
    const char* e = getenv("FT_RNN_SOLO");
    return !(e && e[0] == '0');
  }();
  return on;
}

// G*H <= 384 (column tiles / k-blocks a wave keeps resident), H a power of two in [32, 128] (bf16 MFMA k-blocks)
bool solo_fits(int G, int H) {
  return H >= 32 && (H & (H - 1)) == 0 && H <= 32 * MAXKB && H <= 16 * NW * BT && G * H <= 16 * NW * MAXT &&
         G * H <= 32 * BKB;
}

}  // namespace

int ft_rnn_fwd_solo(int G, const RnnFwdArgs& a, hipStream_t stream) {
  if (!solo_enabled() || !a.vec || !solo_fits(G, a.H) || a.T < 1) return -1;
  const size_t lds = (size_t)3 * 16 * (a.H + 8) * sizeof(unsigned short) + (size_t)16 * (G * a.H + 4) * sizeof(float) +
                     (size_t)G * a.H * sizeof(float) + 16 * sizeof(int);
  dim3 grid(ft_cdiv(a.B, 16), 2);
  if (G == 3)
    hipLaunchKernelGGL(ft_rnn_fwd_solo_kernel<3>, grid, dim3(NW * 64), lds, stream, a);
  else
    hipLaunchKernelGGL(ft_rnn_fwd_solo_kernel<4>, grid, dim3(NW * 64), lds, stream, a);
  return ft_check_launch("rnn_fwd_solo");
}

int ft_rnn_bwd_solo(int G, const RnnBwdArgs& a, hipStream_t stream) {
  if (!solo_enabled() || !a.vec || !solo_fits(G, a.H) || a.T < 1) return -1;
  const size_t lds = (size_t)3 * 16 * (G * a.H + 8) * sizeof(unsigned short) + (size_t)16 * (a.H + 4) * sizeof(float) +
                     16 * sizeof(int);
  dim3 grid(ft_cdiv(a.B, 16), 2);
  if (G == 3)
    hipLaunchKernelGGL(ft_rnn_bwd_solo_kernel<3>, grid, dim3(NW * 64), lds, stream, a);
  else
    hipLaunchKernelGGL(ft_rnn_bwd_solo_kernel<4>, grid, dim3(NW * 64), lds, stream, a);
  return ft_check_launch("rnn_bwd_solo");
}
