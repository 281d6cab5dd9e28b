// Exact-f32 MFMA GEMM family for gfx950 (v_mfma_f32_32x32x2_f32).
//
// All matmul-shaped work of the ForwardTacotron hot path funnels through two kernels:
//   ft_gemm_rows_kernel : C[M,N] (+)= sum_tap shift_tap(A)[M,K] * B_tap    (linear, conv fwd,
//                         conv bwd-data, RNN input projections)   -- NT or NN operand B
//   ft_gemm_tn_kernel   : dW_tap[m][n] = sum_rows A[r][m] * shift_tap(B)[r][n]   (all weight grads)
// "shift_tap" is the channels-last conv trick: a k-tap Conv1d over [B,T,C] is k GEMMs whose A rows
// are the input rows shifted by (tap - pad) inside each batch item, zero outside [0,T).
//
// Tiling: 256 threads = 4 wave64 in 2x2, each wave TMxTN tiles of 32x32 (f32 accumulators in
// AGPR/VGPR), BK = 32 per stage.  LDS tiles are K-major ([k][m]) so the MFMA operand fetch
// (lane l: row l&31, k = kk + (l>>5)) is a conflict-free ds_read_b32 across 32 consecutive floats.
// Measured ceiling of this tiling as a plain GEMM: 105-113 TFLOP/s of the 157 the f32 MFMA pipe offers
// (lab/gemm_lab.hip: operands in registers only 133-143, + LDS fetch 127-134, + LDS stores 120-130, + global
// loads 105-113); these kernels reach 87-102 with row maps, taps and fused epilogues.  The 128x128 NT launches are
// routed to the bf16-split kernel of ft_gemm_b3.hip instead (136-148 TFLOP/s fp32-equivalent); this file keeps the
// 64x64 tiles, the [K][N] operand form, unaligned / odd shapes and the TN (weight-gradient) kernel.
#include "ft_gemm.h"

namespace {

constexpr int BK = 32;

__device__ __forceinline__ float4 ld4(const float* p, int remaining, bool vec) {
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

// One K-stage of MFMAs from the K-major LDS tiles.  The operand fragments of k-pair ks+1 are requested
// BEFORE the MFMAs of k-pair ks are issued (explicit register double buffering): an MFMA occupies the
// matrix pipe for 64 cycles and the wave issues in order, so reads placed after the MFMA block would expose
// the LDS latency once per k-pair.
template <int TM, int TN, int LDA, int LDB>
__device__ __forceinline__ void mfma_stage(const float* ap, const float* bp, f32x16 (&acc)[TM][TN]) {
  float a0[TM], b0[TN], a1[TM], b1[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) a0[i] = ap[32 * i];
#pragma unroll
  for (int j = 0; j < TN; ++j) b0[j] = bp[32 * j];
#pragma unroll
  for (int ks = 0; ks < BK / 2; ks += 2) {
#pragma unroll
    for (int i = 0; i < TM; ++i) a1[i] = ap[2 * (ks + 1) * LDA + 32 * i];
#pragma unroll
    for (int j = 0; j < TN; ++j) b1[j] = bp[2 * (ks + 1) * LDB + 32 * j];
    __builtin_amdgcn_sched_barrier(0);     // keep the prefetch ABOVE the MFMA block (the scheduler sinks it)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b0[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (ks + 2 < BK / 2) {
#pragma unroll
      for (int i = 0; i < TM; ++i) a0[i] = ap[2 * (ks + 2) * LDA + 32 * i];
#pragma unroll
      for (int j = 0; j < TN; ++j) b0[j] = bp[2 * (ks + 2) * LDB + 32 * j];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b1[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// branch-free 16-B load: an invalid lane reads a safe address and its value is replaced by zeros with a
// select (no exec-mask branches in the staging code)
__device__ __forceinline__ float4 ld4_sel(const float* p, const float* safe, bool ok) {
  const float4 v = *reinterpret_cast<const float4*>(ok ? p : safe);
  return ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---------------------------------------------------------------------------------------------
// FAST: every task has 16-B aligned operands, K % 4 == 0 (and N % 4 == 0 for the NN form): staging uses
// ld4_sel only.  The generic form predicates every element (odd sizes, unaligned views).
template <int TM, int TN, bool BNC, bool FAST>
__global__ __launch_bounds__(256) void ft_gemm_rows_kernel(FtGemmBatch batch) {
  const bool zbatch = batch.t[0].nz > 1;           // strided-batch launch: one task, blockIdx.z = instance
  const FtGemmTask& T = batch.t[zbatch ? 0 : blockIdx.z];
  const float* TA = T.A;
  const float* TB = T.B;
  float* TC = T.C;
  if (zbatch) {
    const int z0 = blockIdx.z / T.nz1, z1 = blockIdx.z - z0 * T.nz1;
    TA += z0 * T.sA0 + z1 * T.sA1;
    TB += z0 * T.sB0 + z1 * T.sB1;
    TC += z0 * T.sC0 + z1 * T.sC1;
  }
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int LDA = BM + 1;
  constexpr int LDB = BNC ? BN + 4 : BN + 1;
  constexpr int PA = BM / 32;                 // float4 per thread for A
  constexpr int PB = BN / 32;                 // float4 per thread for B
  // two LDS stages: stage c+1 is written while stage c is multiplied -> ONE barrier per K-stage, and the
  // staging instructions issue in the shadow of the 64-cycle MFMAs
  constexpr int STAGE = BK * LDA + BK * LDB;
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  if (m0 >= T.M || n0 >= T.N) return;
  const int tid = threadIdx.x;
  const int kq = tid & 7, rr = tid >> 3;      // K-contiguous staging: 8 float4 along K, 32 rows / pass

  // Every task field the K loop needs lives in a local: left as T.x the compiler re-reads them with s_load inside
  // the loop, and each such read costs an `s_waitcnt lgkmcnt(0)` that also drains the LDS operand prefetch (SMEM and
  // LDS share the counter).  The locals are re-filled when the loader moves on to the next task of a CHAIN
  // (batch.chain tasks accumulate into task 0's output: C = sum_task sum_tap shift(A_task) * B_task,tap).
  const int tM = T.M, tN = T.N;
  const int nchain = batch.chain > 1 ? batch.chain : 1;
  int tK, taps, Tvalid, shift0, sstep, kch;
  long lda, ldb, btap, atst;
  bool avec, bvec;
  const float* curA;
  const float* curB;
  int a_t[PA];
  bool a_ok[PA];
  const float* a_row[PA];
  constexpr int NQ = BN / 4;                  // NN: float4 per k-row
  constexpr int KR = 256 / NQ;                // NN: k-rows per pass
  const int nq = tid % NQ, kr = tid / NQ;
  const float* b_row[PB];
  bool b_ok[PB];
  // row map of this thread's A rows: logical row -> (item, t); the pointer of (item, t + 0) is kept, a tap only adds a
  // wave-uniform offset and a range test
  auto setup = [&](const FtGemmTask& S, const float* SA, const float* SB) {
    curA = SA;
    curB = SB;
    tK = S.K;
    taps = S.taps;
    lda = S.lda;
    ldb = S.ldb;
    btap = S.b_tap_stride;
    Tvalid = S.amap.Tvalid;
    shift0 = S.amap.shift0;
    sstep = S.amap.shift_step;
    atst = S.amap.tstride;
    avec = S.a_vec;
    bvec = S.b_vec;
    kch = (tK + BK - 1) / BK;
    const int Tlog = S.amap.Tlog;
    const long abst = S.amap.bstride;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      int m = m0 + rr + 32 * p;
      a_ok[p] = m < tM;
      int b = m / Tlog;
      a_t[p] = m - b * Tlog;
      a_row[p] = SA + ((long)b * abst + (long)a_t[p] * atst) * lda + 4 * kq;
    }
#pragma unroll
    for (int p = 0; p < PB; ++p) {
      if constexpr (!BNC) {
        const int n = n0 + rr + 32 * p;
        b_ok[p] = n < tN;
        b_row[p] = SB + (long)n * ldb + 4 * kq;
      } else {
        b_ok[p] = n0 + 4 * nq < tN;
        b_row[p] = SB + (long)(kr + KR * p) * ldb + n0 + 4 * nq;
      }
    }
  };
  setup(T, TA, TB);
  int nch = 0;                                // K stages of the whole chain
  for (int i = 0; i < nchain; ++i) nch += batch.t[i].taps * ((batch.t[i].K + BK - 1) / BK);
  if (nchain == 1) nch = taps * kch;

  // loader cursor: (task, tap, k chunk) of the stage the next load_stage call fetches -- stages are requested in order
  int l_task = 0, l_tap = 0, l_kc = 0;
  float4 ra[PA], rb[PB];
  auto load_stage = [&](int) {
    const int j = l_tap;
    const int k0 = l_kc * BK;
    const int shift = shift0 + j * sstep;
    const int k = k0 + 4 * kq;
    const bool kok = k < tK;
    const long aoff = (long)shift * atst * lda + k0;           // wave-uniform
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      const int ts = a_t[p] + shift;
      const bool ok = a_ok[p] & (ts >= 0) & (ts < Tvalid);
      if constexpr (FAST) {
        ra[p] = ld4_sel(a_row[p] + aoff, curA, ok & kok);
        if (tK & 3) {                         // K % 4 != 0 (a_rowpad launches): whatever lies behind column K-1 is ignored
          if (k + 1 >= tK) ra[p].y = 0.f;
          if (k + 2 >= tK) ra[p].z = 0.f;
          if (k + 3 >= tK) ra[p].w = 0.f;
        }
      }
      else ra[p] = ld4(a_row[p] + aoff, ok ? tK - k : 0, avec);
    }
    if constexpr (!BNC) {
      const long boff = (long)j * btap + k0;                   // wave-uniform
#pragma unroll
      for (int p = 0; p < PB; ++p) {
        if constexpr (FAST) rb[p] = ld4_sel(b_row[p] + boff, curB, b_ok[p] & kok);
        else rb[p] = ld4(b_row[p] + boff, b_ok[p] ? tK - k : 0, bvec);
      }
    } else {
      const long boff = (long)j * btap + (long)k0 * ldb;       // wave-uniform
#pragma unroll
      for (int p = 0; p < PB; ++p) {
        const int kk = k0 + kr + KR * p;
        if constexpr (FAST) rb[p] = ld4_sel(b_row[p] + boff, curB, b_ok[p] & (kk < tK));
        else rb[p] = ld4(b_row[p] + boff, kk < tK ? tN - (n0 + 4 * nq) : 0, bvec);
      }
    }
    if (++l_kc == kch) {
      l_kc = 0;
      if (++l_tap == taps) {
        l_tap = 0;
        if (++l_task < nchain) setup(batch.t[l_task], batch.t[l_task].A, batch.t[l_task].B);
      }
    }
  };
  auto store_stage = [&](int buf) {
    float* As = smem + buf * STAGE;
    float* Bs = As + BK * LDA;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      int m = rr + 32 * p;
      As[(4 * kq + 0) * LDA + m] = ra[p].x;
      As[(4 * kq + 1) * LDA + m] = ra[p].y;
      As[(4 * kq + 2) * LDA + m] = ra[p].z;
      As[(4 * kq + 3) * LDA + m] = ra[p].w;
    }
    if constexpr (!BNC) {
#pragma unroll
      for (int p = 0; p < PB; ++p) {
        int n = rr + 32 * p;
        Bs[(4 * kq + 0) * LDB + n] = rb[p].x;
        Bs[(4 * kq + 1) * LDB + n] = rb[p].y;
        Bs[(4 * kq + 2) * LDB + n] = rb[p].z;
        Bs[(4 * kq + 3) * LDB + n] = rb[p].w;
      }
    } else {
      constexpr int NQ = BN / 4;
      constexpr int KR = 256 / NQ;
      const int nq = tid % NQ, kr = tid / NQ;
#pragma unroll
      for (int p = 0; p < PB; ++p)
        *reinterpret_cast<float4*>(&Bs[(kr + KR * p) * LDB + 4 * nq]) = rb[p];
    }
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int half = lane >> 5, l31 = lane & 31;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  load_stage(0);
  store_stage(0);
  if (nch > 1) load_stage(1);
  __syncthreads();
  for (int c = 0; c < nch; ++c) {
    const int cur = c & 1;
    if (c + 1 < nch) {
      store_stage(cur ^ 1);                  // stage c+1 (its registers were loaded one iteration ago)
      if (c + 2 < nch) load_stage(c + 2);
    }
    const float* ap = smem + cur * STAGE + half * LDA + wm * 32 * TM + l31;
    const float* bp = smem + cur * STAGE + BK * LDA + half * LDB + wn * 32 * TN + l31;
    mfma_stage<TM, TN, LDA, LDB>(ap, bp, acc);
    __syncthreads();
  }

  if (batch.hw_mode) {                        // highway gate / gate gradient in the epilogue (ft_gemm.h)
    ft_highway_epilogue<TM, TN>(batch, T, TC, acc, smem, m0, n0, tid);
    return;
  }
  // epilogue: lane holds column l31, rows (e&3) + 8*(e>>2) + 4*half of each 32x32 tile
  const float* ebias = T.bias;
  const float* escale = T.scale;
  const float* eshift = T.shift;
  const bool erelu = T.relu != 0, eacc = T.accumulate != 0;
  const long ldc = T.ldc, cbst = T.cmap.bstride, ctst = T.cmap.tstride;
  const int cTlog = T.cmap.Tlog;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * 32 * TN + 32 * j + l31;
      if (col >= tN) continue;
      const float bv = ebias ? ebias[col] : 0.f;
      const float sc = escale ? escale[col] : 1.f;
      const float sh = escale ? eshift[col] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 32 * TM + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * half;
        if (row >= tM) continue;
        long crow = row;
        if (cbst != 0) {                      // non-identity output layout (time-major); identity skips the division
          const int cb = row / cTlog;
          crow = (long)cb * cbst + (long)(row - cb * cTlog) * ctst;
        }
        float* cp = TC + crow * ldc + col;
        float v = acc[i][j][e] + bv;
        if (erelu) v = fmaxf(v, 0.f);
        if (escale) v = v * sc + sh;
        if (batch.relu_mask && !(batch.relu_mask[crow * ldc + col] > 0.f)) v = 0.f;
        if (eacc) v += *cp;              // residual / gradient accumulation happens last
        *cp = v;
      }
    }
}

// ---------------------------------------------------------------------------------------------
// TN: slab[z][m][n] = sum_{r in slice} A[map_a(r)][m] * B[map_b,tap(r)][n] ; z = tap*S + s
template <int TM, int TN, bool FAST>
__global__ __launch_bounds__(256) void ft_gemm_tn_kernel(FtGemmTNTask T, float* slab, int S, int rows_per_split) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int PA = BM / 32, PB = BN / 32;
  constexpr int AQ = BM / 4, AR = 256 / AQ;   // float4 per k-row, k-rows per pass
  constexpr int BQ = BN / 4, BR = 256 / BQ;
  constexpr int STAGE = BK * LDA + BK * LDB;    // two LDS stages, one barrier per K-stage (see rows kernel)
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

  const FtTnWho who = ft_tn_who(T, S, BM);
  const int m0 = who.mtile * BM, n0 = blockIdx.y * BN;
  const int zts = who.zts, s = who.s;
  const int zi = zts / T.taps, tap = zts - zi * T.taps;
  const float* TA = T.A;
  const float* TB = T.B;
  if (T.nz > 1) {
    const int z0 = zi / T.nz1, z1 = zi - z0 * T.nz1;
    TA += z0 * T.sA0 + z1 * T.sA1;
    TB += z0 * T.sB0 + z1 * T.sB1;
  }
  const int r_begin = s * rows_per_split;
  const int r_end = min(T.R, r_begin + rows_per_split);
  const int tid = threadIdx.x;
  const int aq = tid % AQ, ar = tid / AQ;
  const int bq = tid % BQ, br = tid / BQ;
  int bshift = T.bmap.shift0 + tap * T.bmap.shift_step;
  const int ashift = T.amap.shift0 + tap * T.amap.shift_step;
  int aTvalid = T.amap.Tvalid;
  if (who.kk > 0) {                           // conv-bank mode: this tile's member decides shift / valid rows
    bshift = tap - who.kk / 2;
    if (who.kk & 1) aTvalid = T.bankTodd;
  }

  // task fields as locals (see the rows kernel) and (item, t) of every staged row tracked incrementally: stages are
  // requested in strictly increasing order, BK rows apart, so the per-row integer division happens once
  const int tM = T.M, tN = T.N;
  const long lda = T.lda, ldb = T.ldb;
  const int aTlog = T.amap.Tlog, bTlog = T.bmap.Tlog, bTvalid = T.bmap.Tvalid;
  const long abst = T.amap.bstride, atst = T.amap.tstride, bbst = T.bmap.bstride, btst = T.bmap.tstride;
  const bool avec = T.a_vec, bvec = T.b_vec;
  const int am = m0 + 4 * aq, bn = n0 + 4 * bq;
  const bool am_ok = am < tM, bn_ok = bn < tN;
  int a_b[PA], a_t[PA], b_b[PB], b_t[PB];
#pragma unroll
  for (int p = 0; p < PA; ++p) {
    const int r = r_begin + ar + AR * p;
    a_b[p] = r / aTlog;
    a_t[p] = r - a_b[p] * aTlog;
  }
#pragma unroll
  for (int p = 0; p < PB; ++p) {
    const int r = r_begin + br + BR * p;
    b_b[p] = r / bTlog;
    b_t[p] = r - b_b[p] * bTlog;
  }
  int r_next = r_begin;                       // first row of the stage the next load_stage call fetches

  float4 ra[PA], rb[PB];
  auto load_stage = [&](int r0) {
    (void)r0;                                 // == r_next by construction
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      const int r = r_next + ar + AR * p;
      const int ts = a_t[p] + ashift;
      const bool ok = (r < r_end) & (ts >= 0) & (ts < aTvalid);
      const float* ptr = TA + ((long)a_b[p] * abst + (long)ts * atst) * lda + am;
      if constexpr (FAST) ra[p] = ld4_sel(ptr, TA, ok & am_ok);
      else ra[p] = ld4(ptr, ok ? tM - am : 0, avec);
      a_t[p] += BK;
      while (a_t[p] >= aTlog) {
        a_t[p] -= aTlog;
        ++a_b[p];
      }
    }
#pragma unroll
    for (int p = 0; p < PB; ++p) {
      const int r = r_next + br + BR * p;
      const int ts = b_t[p] + bshift;
      const bool ok = (r < r_end) & (ts >= 0) & (ts < bTvalid);
      const float* ptr = TB + ((long)b_b[p] * bbst + (long)ts * btst) * ldb + bn;
      if constexpr (FAST) rb[p] = ld4_sel(ptr, TB, ok & bn_ok);
      else rb[p] = ld4(ptr, ok ? tN - bn : 0, bvec);
      b_t[p] += BK;
      while (b_t[p] >= bTlog) {
        b_t[p] -= bTlog;
        ++b_b[p];
      }
    }
    r_next += BK;
  };
  auto store_stage = [&](int buf) {
    float* As = smem + buf * STAGE;
    float* Bs = As + BK * LDA;
#pragma unroll
    for (int p = 0; p < PA; ++p) *reinterpret_cast<float4*>(&As[(ar + AR * p) * LDA + 4 * aq]) = ra[p];
#pragma unroll
    for (int p = 0; p < PB; ++p) *reinterpret_cast<float4*>(&Bs[(br + BR * p) * LDB + 4 * bq]) = rb[p];
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int half = lane >> 5, l31 = lane & 31;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (r_begin < r_end) {
    load_stage(r_begin);
    store_stage(0);
    if (r_begin + BK < r_end) load_stage(r_begin + BK);
    __syncthreads();
    int cur = 0;
    for (int r0 = r_begin; r0 < r_end; r0 += BK, cur ^= 1) {
      if (r0 + BK < r_end) {
        store_stage(cur ^ 1);
        if (r0 + 2 * BK < r_end) load_stage(r0 + 2 * BK);
      }
      const float* ap = smem + cur * STAGE + half * LDA + wm * 32 * TM + l31;
      const float* bp = smem + cur * STAGE + BK * LDA + half * LDB + wn * 32 * TN + l31;
      mfma_stage<TM, TN, LDA, LDB>(ap, bp, acc);
      __syncthreads();
    }
  }
  float* out = slab + ((long)zts * S + s) * T.M * T.N;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * 32 * TN + 32 * j + l31;
      if (col >= T.N) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 32 * TM + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * half;
        if (row < T.M) out[(long)row * T.N + col] = acc[i][j][e];
      }
    }
}

// dst_z[m*ldm + n*ldn + tap*ldj] (+)= sum_s slab[(z*taps+tap)*S+s][m][n]   (fixed order -> reproducible)
// One thread per (instance, m, n) does ALL taps: the reads of a tap are coalesced over n, and with the torch conv layout
// (ldn = taps, ldj = 1) a thread's writes are `taps` adjacent floats, so a wave writes one contiguous run (one thread per
// (tap, m, n) wrote 4-byte pieces `taps` floats apart: 139 us for the prenet projection's 1.6 M weights).
__global__ void ft_splitk_reduce_kernel(const float* slab, float* dst, int M, int N, int taps, int S,
                                        long ldm, long ldn, long ldj, int accumulate, int nz, int nz1, long sD0,
                                        long sD1) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long plane = (long)M * N;
  long total = (long)nz * plane;
  if (idx >= total) return;
  int zi = (int)(idx / plane);
  long mn = idx - (long)zi * plane;
  int m = (int)(mn / N), n = (int)(mn - (long)m * N);
  float* d = dst + (zi / nz1) * sD0 + (zi % nz1) * sD1 + m * ldm + n * ldn;
  for (int tap = 0; tap < taps; ++tap) {
    const float* p = slab + ((long)(zi * taps + tap) * S) * plane + mn;
    float acc = 0.f;
    int s = 0;
    for (; s + 4 <= S; s += 4) {                     // four loads in flight, summed in slab order
      const float a0 = p[(long)s * plane], a1 = p[(long)(s + 1) * plane], a2 = p[(long)(s + 2) * plane],
                  a3 = p[(long)(s + 3) * plane];
      acc = (((acc + a0) + a1) + a2) + a3;
    }
    for (; s < S; ++s) acc += p[(long)s * plane];
    float* q = d + tap * ldj;
    *q = accumulate ? *q + acc : acc;
  }
}

// the same with one thread per (instance, tap, m, n): more threads, 4-byte writes `taps` floats apart -- the better form when
// a thread of the kernel above would walk a long taps x S chain (FastPitch's 9-tap convolutions: 25.02 vs 25.15 ms per step)
__global__ void ft_splitk_reduce_tap_kernel(const float* slab, float* dst, int M, int N, int taps, int S,
                                            long ldm, long ldn, long ldj, int accumulate, int nz, int nz1, long sD0,
                                            long sD1) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)nz * taps * M * N;
  if (idx >= total) return;
  int zt = (int)(idx / ((long)M * N));             // instance * taps + tap
  long mn = idx - (long)zt * M * N;
  int zi = zt / taps, tap = zt - zi * taps;
  int m = (int)(mn / N), n = (int)(mn - (long)m * N);
  const float* p = slab + ((long)zt * S) * M * N + mn;
  const long plane = (long)M * N;
  float acc = 0.f;
  int s = 0;
  for (; s + 4 <= S; s += 4) {                     // four loads in flight, summed in slab order
    const float a0 = p[(long)s * plane], a1 = p[(long)(s + 1) * plane], a2 = p[(long)(s + 2) * plane],
                a3 = p[(long)(s + 3) * plane];
    acc = (((acc + a0) + a1) + a2) + a3;
  }
  for (; s < S; ++s) acc += p[(long)s * plane];
  float* d = dst + (zi / nz1) * sD0 + (zi % nz1) * sD1 + m * ldm + n * ldn + tap * ldj;
  *d = accumulate ? *d + acc : acc;
}

// conv-bank mode: dst_kk[(co*N + n)*kk + j] = sum_s slab[j*S + s][(kk-1)*C + co][n], one thread per (m, n, tap j)
struct BankDst {
  float* p[FT_MAX_TASKS];
};
__global__ void ft_bank_wgrad_reduce_kernel(const float* __restrict__ slab, BankDst dst, int M, int N, int C, int S) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y;                        // tap
  if (idx >= (long)M * N) return;
  const int m = (int)(idx / N), n = (int)(idx - (long)m * N);
  const int member = m / C, kk = member + 1, co = m - member * C;
  if (j >= kk) return;
  const long plane = (long)M * N;
  const float* p = slab + ((long)j * S) * plane + idx;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;    // four loads in flight; summed in slab order below
  int s = 0;
  float acc = 0.f;
  for (; s + 4 <= S; s += 4) {
    a0 = p[(long)s * plane];
    a1 = p[(long)(s + 1) * plane];
    a2 = p[(long)(s + 2) * plane];
    a3 = p[(long)(s + 3) * plane];
    acc = (((acc + a0) + a1) + a2) + a3;
  }
  for (; s < S; ++s) acc += p[(long)s * plane];
  dst.p[member][((long)co * N + n) * kk + j] = acc;
}

struct TNPlan {
  int tm, tn, S, rows_per_split;
};

// FT_GEMM_B3_TN: 0 = weight gradients never take the bf16-split path, 1 = every tile size does, default: 128x128 only
int tn_b3_mode() {
  static const int mode = [] {
    const char* e = getenv("FT_GEMM_B3_TN");
    return e ? (e[0] == '1' ? 2 : (e[0] == '0' ? 0 : 1)) : 1;
  }();
  return ft_gemm_b3_enabled() ? mode : 0;
}

// split path usable (aligned operands): its 128x128 tile runs ~1.35x the f32 kernels' rate, so it is preferred whenever
// tiles x split-K can still fill the chip -- the extra slabs it needs cost a few microseconds of HBM traffic
TNPlan plan_tn(const FtGemmTNTask& t, bool b3_ok, long slots = 512) {
  TNPlan p;
  const int nz = t.nz > 1 ? t.nz : 1;
  long maxs = (t.R + 4 * BK - 1) / (4 * BK);      // at least 128 rows per split
  if (maxs < 1) maxs = 1;
  const long tiles128 = (long)ft_cdiv(t.M, 128) * ft_cdiv(t.N, 128) * t.taps * nz;
  bool small = tiles128 < 64 || t.M <= 64 || t.N <= 64;
  // ... as long as every split still has >= 1024 rows to amortise its prologue and its slab
  if (b3_ok && tn_b3_mode() >= 1 && t.M > 64 && t.N > 64 && tiles128 * (t.R / 1024) >= 256) small = false;
  p.tm = p.tn = small ? 1 : 2;
  int bm = 64 * p.tm;
  long tiles = (long)ft_cdiv(t.M, bm) * ft_cdiv(t.N, bm) * t.taps * nz;
  long want = tiles >= slots ? 1 : (slots + tiles - 1) / tiles;
  // the split-path kernel holds exactly 2 workgroups per CU (53 KB LDS, 220 VGPRs) = 512 slots: a grid slightly above
  // one full wave pays a second, nearly empty one (576 workgroups ran at 94 TF, 512 at 134) -> round DOWN there
  const bool split_path = b3_ok && tn_b3_mode() >= 1 && p.tm == 2;
  if (split_path && tiles < slots) want = slots / tiles;
  if (t.bankC > 0) {
    // conv-bank mode: member kk only has taps j < kk, i.e. (K+1)/(2K) of the grid does work; aim for ~4 waves of
    // workgroups so that the last, partial wave costs little
    const long K = t.taps;
    const long live = tiles * (K + 1) / (2 * K);
    want = (4 * slots + live - 1) / (live > 0 ? live : 1);
  }
  static const long force_s = [] {             // FT_TN_FORCE_S=<n>: tuning aid (lab/tn_split_lab.py)
    const char* e = getenv("FT_TN_FORCE_S");
    return e ? atol(e) : 0L;
  }();
  if (force_s > 0) want = force_s;
  long S = want < maxs ? want : maxs;
  if (S > 65535 / ((long)t.taps * nz)) S = 65535 / ((long)t.taps * nz);
  if (S < 1) S = 1;
  long rps = (t.R + S - 1) / S;
  rps = ((rps + BK - 1) / BK) * BK;
  if (rps < BK) rps = BK;
  S = (t.R + rps - 1) / rps;
  if (S < 1) S = 1;
  p.S = (int)S;
  p.rows_per_split = (int)rps;
  return p;
}

// FT_GEMM_LOG=1: every launch is timed with HIP events on its stream (synchronising) and one line per launch goes
// to stderr -- a profiling aid for tools/gemm_report.py, never enabled in a timed run.
struct GemmLog {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipStream_t stream = nullptr;
  static bool on() {
    static int v = -1;
    if (v < 0) {
      const char* e = getenv("FT_GEMM_LOG");
      v = (e && e[0] == '1') ? 1 : 0;
    }
    return v == 1;
  }
  void begin(hipStream_t s) {
    if (!on()) return;
    stream = s;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, s);
  }
  void end(const char* kind, long M, long N, long K, int taps, int inst, const char* variant, double flops) {
    if (!on() || !e0) return;
    (void)hipEventRecord(e1, stream);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    fprintf(stderr, "FTGEMM %s M=%ld N=%ld K=%ld taps=%d inst=%d %s us=%.1f TF=%.1f\n", kind, M, N, K, taps, inst,
            variant, ms * 1e3, flops / (ms * 1e-3) / 1e12);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
};

}  // namespace

bool ft_gemm_b3_enabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("FT_GEMM_B3");
    v = (e && e[0] == '0') ? 0 : 1;
  }
  return v == 1;
}

static int g_gemm_precision = 0;
int ft_gemm_precision() { return g_gemm_precision; }
extern "C" int ft_set_gemm_precision(int bf16) {
  const int old = g_gemm_precision;
  g_gemm_precision = bf16 ? 1 : 0;
  return old;
}

size_t ft_gemm_tn_workspace_floats(const FtGemmTNTask& t) {
  // the query does not know the operands' alignment yet: cover both plans the launcher may pick
  const TNPlan p0 = plan_tn(t, false), p1 = plan_tn(t, true);
  return (size_t)(p0.S > p1.S ? p0.S : p1.S) * t.taps * (t.nz > 1 ? t.nz : 1) * t.M * t.N;
}

// Split-K plan of an NT launch (FtGemmBatch.ksplit_slab): only single-task / chained products with few 128x128 output
// tiles, a long stage sequence (>= 128 stages of 16 k) and a plain epilogue; S ranges of >= 32 stages each, S * tiles <= the
// stream's slots
static int rows_ksplit_plan(const FtGemmBatch& b, int ntasks, hipStream_t stream) {
  static const bool enabled = [] {                   // FT_GEMM_KSPLIT=0: never split (A/B knob)
    const char* e = getenv("FT_GEMM_KSPLIT");
    return !(e && e[0] == '0');
  }();
  const bool chained = b.chain > 1;
  if (!enabled || !(chained ? b.chain == ntasks : ntasks == 1) || !ft_gemm_b3_enabled()) return 1;
  const FtGemmTask& t0 = b.t[0];
  if (t0.nz > 1 || t0.relu || t0.scale || t0.stat || b.hw_mode || b.relu_mask || t0.N % 4 != 0 || t0.M <= 64 || t0.N <= 64) return 1;
  if (!ft_rows_b3p_ok(b, ntasks)) return 1;
  long stages = 0;
  for (int i = 0; i < ntasks; ++i) stages += (long)b.t[i].taps * ft_cdiv(b.t[i].K, 16);
  const long tiles = (long)ft_cdiv(t0.M, 128) * ft_cdiv(t0.N, 128);
  const long slots = ft_stream_slots(stream);
  if (stages < 128) return 1;                      // (a short contraction gains nothing: launch + second pass)
  long S = slots / (tiles > 0 ? tiles : 1);
  if (S > stages / 32) S = stages / 32;
  if (S > 16) S = 16;
  if (S < 2) return 1;
  const long per = (stages + S - 1) / S;
  return (int)((stages + per - 1) / per);            // no empty range
}

size_t ft_rows_ksplit_floats(const FtGemmBatch& batch, int ntasks, hipStream_t stream) {
  const int S = rows_ksplit_plan(batch, ntasks, stream);
  return S > 1 ? (size_t)S * batch.t[0].M * batch.t[0].N : 0;
}

int ft_launch_gemm_rows(FtGemmBatch* batch, int ntasks, bool b_ncontig, hipStream_t stream) {
  FT_REQUIRE(ntasks >= 1 && ntasks <= FT_MAX_TASKS, "gemm_rows: bad task count %d", ntasks);
  int maxM = 0, maxN = 0;
  long tiles128 = 0;
  for (int i = 0; i < ntasks; ++i) {
    FtGemmTask& t = batch->t[i];
    FT_REQUIRE(t.M >= 0 && t.N >= 0 && t.K >= 0 && t.taps >= 1, "gemm_rows: bad dims");
    FT_REQUIRE(t.amap.Tlog > 0, "gemm_rows: bad row map");
    if (t.cmap.Tlog <= 0) t.cmap = ft_rowmap_identity(t.M);     // tasks built with memset(0): identity output
    if (t.nz <= 1) {
      t.nz = t.nz1 = 1;
      t.sA0 = t.sA1 = t.sB0 = t.sB1 = t.sC0 = t.sC1 = 0;
    }
    FT_REQUIRE(t.nz == 1 || ntasks == 1, "gemm_rows: strided batch needs a single task");
    FT_REQUIRE(t.nz1 >= 1 && t.nz % t.nz1 == 0, "gemm_rows: bad batch split");
    t.a_vec = (t.lda % 4 == 0) && (((uintptr_t)t.A) % 16 == 0) && (t.sA0 % 4 == 0) && (t.sA1 % 4 == 0);
    t.b_vec = (t.ldb % 4 == 0) && (t.b_tap_stride % 4 == 0) && (((uintptr_t)t.B) % 16 == 0) && (t.sB0 % 4 == 0) &&
              (t.sB1 % 4 == 0);
    if (t.M > maxM) maxM = t.M;
    if (t.N > maxN) maxN = t.N;
    tiles128 += (long)ft_cdiv(t.M, 128) * ft_cdiv(t.N, 128) * t.nz;
  }
  if (maxM == 0 || maxN == 0) return FT_OK;
  for (int i = ntasks; i < FT_MAX_TASKS; ++i) batch->t[i] = batch->t[0];
  const bool chained = batch->chain > 1;
  if (chained) {
    FT_REQUIRE(batch->chain == ntasks, "gemm_rows: chain must cover all tasks");
    for (int i = 0; i < ntasks; ++i)
      FT_REQUIRE(batch->t[i].M == batch->t[0].M && batch->t[i].N == batch->t[0].N && batch->t[i].nz == 1,
                 "gemm_rows: chained tasks must share M and N");
    tiles128 /= ntasks;
  } else {
    batch->chain = 0;
  }
  bool big = batch->force_tile ? batch->force_tile == 2 : ft_rows_tile_is_big(tiles128, maxM, maxN);
  batch->ksplit = 0;
  if (batch->ksplit_slab) {
    bool ok = !b_ncontig && ft_gemm_precision() == 0;
    for (int i = 0; i < ntasks && ok; ++i) {
      const FtGemmTask& t = batch->t[i];
      ok = (t.lda % 4 == 0) && (((uintptr_t)t.A) % 16 == 0) && (t.ldb % 4 == 0) && (t.b_tap_stride % 4 == 0) &&
           (((uintptr_t)t.B) % 16 == 0) && t.K % 4 == 0;
    }
    const int S = ok ? rows_ksplit_plan(*batch, ntasks, stream) : 1;
    if (S > 1) {
      batch->ksplit = S;
      big = true;
    }
  }
  const int bm = big ? 128 : 64;
  dim3 grid(ft_cdiv(maxM, bm), ft_cdiv(maxN, bm), chained ? 1 : (ntasks == 1 ? batch->t[0].nz : ntasks));
  if (batch->ksplit > 1) grid.z = batch->ksplit;
  FT_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "gemm_rows: grid too large");
  bool fast = true;
  for (int i = 0; i < ntasks; ++i) {
    const FtGemmTask& t = batch->t[i];
    const bool k_ok = t.K % 4 == 0 || (b_ncontig && t.a_rowpad && t.lda >= ((t.K + 3) & ~3));
    fast = fast && t.a_vec && t.b_vec && k_ok && (!b_ncontig || t.N % 4 == 0);
  }
#define FT_ROWS_LAUNCH(TM_, BNC_)                                                                              \
  do {                                                                                                         \
    if (fast)                                                                                                  \
      hipLaunchKernelGGL((ft_gemm_rows_kernel<TM_, TM_, BNC_, true>), grid, dim3(256), 0, stream, *batch);     \
    else                                                                                                       \
      hipLaunchKernelGGL((ft_gemm_rows_kernel<TM_, TM_, BNC_, false>), grid, dim3(256), 0, stream, *batch);    \
  } while (0)
  GemmLog log;
  log.begin(stream);
  // the 64x64 tile gains nothing from the split path (its staging per MFMA is twice the 128-tile's): f32 kernel there
  // bf16 precision mode: every NT-form fast launch takes the (one-plane) bf16 kernel, whatever its tile
  const bool b3 = fast && !b_ncontig && ((big && ft_gemm_b3_enabled()) || ft_gemm_precision() == 1);
  batch->stat_fused = b3 && big && !chained;      // only the 128x128 split kernel computes BatchNorm statistics
  if (b3) {
    (void)ft_launch_gemm_rows_b3(*batch, big, grid, stream);
  } else if (big) {
    if (b_ncontig) FT_ROWS_LAUNCH(2, true);
    else FT_ROWS_LAUNCH(2, false);
  } else {
    if (b_ncontig) FT_ROWS_LAUNCH(1, true);
    else FT_ROWS_LAUNCH(1, false);
  }
#undef FT_ROWS_LAUNCH
  if (GemmLog::on()) {
    double fl = 0;
    long sk = 0;
    for (int i = 0; i < ntasks; ++i) {
      const FtGemmTask& t = batch->t[i];
      fl += 2.0 * t.M * t.N * t.K * t.taps * t.nz;
      sk += (long)t.K * t.taps;
    }
    log.end(b_ncontig ? "rowsNN" : (b3 ? (ft_gemm_precision() == 1 ? "rowsBF" : "rowsB3") : "rowsNT"), maxM, maxN, sk, ntasks, batch->t[0].nz,
            big ? "128" : "64", fl);
  }
  int rc = ft_check_launch("gemm_rows");
  if (rc == FT_OK && batch->ksplit > 1) rc = ft_launch_ksplit_reduce(batch->ksplit_slab, batch->ksplit, batch->t[0], stream);
  return rc;
}

int ft_launch_slab_sum(const float* slab, float* dst, int M, int N, int S, long ldm, hipStream_t stream) {
  long total = (long)M * N;
  if (total == 0) return FT_OK;
  hipLaunchKernelGGL(ft_splitk_reduce_kernel, dim3(ft_cdiv(total, 256)), dim3(256), 0, stream, slab, dst, M, N, 1, S,
                     ldm, 1L, 0L, 0, 1, 1, 0L, 0L);
  return ft_check_launch("slab_sum");
}

int ft_launch_gemm_tn(const FtGemmTNTask& task_in, float* workspace, size_t workspace_floats,
                      hipStream_t stream) {
  FtGemmTNTask t = task_in;
  FT_REQUIRE(t.M >= 0 && t.N >= 0 && t.R >= 0 && t.taps >= 1, "gemm_tn: bad dims");
  if (t.M == 0 || t.N == 0) return FT_OK;
  if (t.nz <= 1) {
    t.nz = t.nz1 = 1;
    t.sA0 = t.sA1 = t.sB0 = t.sB1 = t.sD0 = t.sD1 = 0;
  }
  FT_REQUIRE(t.nz1 >= 1 && t.nz % t.nz1 == 0, "gemm_tn: bad batch split");
  t.a_vec = (t.lda % 4 == 0) && (((uintptr_t)t.A) % 16 == 0) && (t.sA0 % 4 == 0) && (t.sA1 % 4 == 0);
  t.b_vec = (t.ldb % 4 == 0) && (((uintptr_t)t.B) % 16 == 0) && (t.sB0 % 4 == 0) && (t.sB1 % 4 == 0);
  const bool m_ok = t.M % 4 == 0 || (t.rowpad && t.lda >= ((t.M + 3) & ~3));
  const bool n_ok = t.N % 4 == 0 || (t.rowpad && t.ldb >= ((t.N + 3) & ~3));
  const bool fast = t.a_vec && t.b_vec && m_ok && n_ok;
  TNPlan p = plan_tn(t, fast, ft_stream_slots(stream));       // a CU-limited stream has fewer slots: fewer, longer splits
  size_t need = (size_t)p.S * t.taps * t.nz * t.M * t.N;
  FT_REQUIRE(workspace && workspace_floats >= need, "gemm_tn: workspace too small (%zu < %zu floats)",
             workspace_floats, need);
  const int bm = 64 * p.tm;
  dim3 grid(ft_cdiv(t.M, bm), ft_cdiv(t.N, bm), p.S * t.taps * t.nz);
  if (t.bankC > 0) {
    FT_REQUIRE(t.bankC % bm == 0 && t.M % t.bankC == 0 && t.M / t.bankC <= FT_MAX_TASKS && t.taps == t.M / t.bankC &&
                   t.nz == 1 && !t.accumulate,
               "gemm_tn: bad conv-bank task");
    const int K = t.taps;
    grid = dim3((t.bankC / bm) * (K * (K + 1) / 2), ft_cdiv(t.N, bm), p.S);      // live (member, tap, tile) triples
  }
  FT_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "gemm_tn: grid too large");
  GemmLog log;
  log.begin(stream);
  // bf16-split TN form (r-pair packed LDS tiles, ft_gemm_b3.hip): on by default for the 128x128 tile; the 64x64 tile
  // has twice the staging per MFMA and stays on the f32 kernel unless FT_GEMM_B3_TN=1
  const bool b3 = fast && (ft_gemm_precision() == 1 || tn_b3_mode() == 2 || (tn_b3_mode() == 1 && p.tm == 2));
  if (b3) {
    (void)ft_launch_gemm_tn_b3(t, workspace, p.S, p.rows_per_split, p.tm, grid, stream);
  } else if (p.tm == 2) {
    if (fast) hipLaunchKernelGGL((ft_gemm_tn_kernel<2, 2, true>), grid, dim3(256), 0, stream, t, workspace, p.S, p.rows_per_split);
    else hipLaunchKernelGGL((ft_gemm_tn_kernel<2, 2, false>), grid, dim3(256), 0, stream, t, workspace, p.S, p.rows_per_split);
  } else {
    if (fast) hipLaunchKernelGGL((ft_gemm_tn_kernel<1, 1, true>), grid, dim3(256), 0, stream, t, workspace, p.S, p.rows_per_split);
    else hipLaunchKernelGGL((ft_gemm_tn_kernel<1, 1, false>), grid, dim3(256), 0, stream, t, workspace, p.S, p.rows_per_split);
  }
  int rc = ft_check_launch("gemm_tn");
  if (rc) return rc;
  long total = (long)t.nz * t.taps * t.M * t.N;
  if (t.bankC > 0) {
    BankDst bd;
    for (int i = 0; i < FT_MAX_TASKS; ++i) bd.p[i] = t.bank_dst[i];
    hipLaunchKernelGGL(ft_bank_wgrad_reduce_kernel, dim3(ft_cdiv((long)t.M * t.N, 256), t.taps), dim3(256), 0, stream,
                       workspace, bd, t.M, t.N, t.bankC, p.S);
    if (GemmLog::on()) {
      char var[32];
      snprintf(var, sizeof(var), "%d/S%d", 64 * p.tm, p.S);
      const int K = t.M / t.bankC;
      log.end("tnbank", t.M, t.N, t.R, t.taps, t.nz, var, 2.0 * t.bankC * t.N * t.R * (K * (K + 1) / 2));
    }
    return ft_check_launch("bank_wgrad_reduce");
  }
  if (t.taps * p.S <= 16)
    hipLaunchKernelGGL(ft_splitk_reduce_kernel, dim3(ft_cdiv((long)t.nz * t.M * t.N, 256)), dim3(256), 0, stream, workspace,
                       t.dst, t.M, t.N, t.taps, p.S, t.ldm, t.ldn, t.ldj, t.accumulate, t.nz, t.nz1, t.sD0, t.sD1);
  else
    hipLaunchKernelGGL(ft_splitk_reduce_tap_kernel, dim3(ft_cdiv(total, 256)), dim3(256), 0, stream, workspace, t.dst,
                       t.M, t.N, t.taps, p.S, t.ldm, t.ldn, t.ldj, t.accumulate, t.nz, t.nz1, t.sD0, t.sD1);
  if (GemmLog::on()) {
    char var[32];
    snprintf(var, sizeof(var), "%d/S%d", 64 * p.tm, p.S);
    log.end("tn", t.M, t.N, t.R, t.taps, t.nz, var, 2.0 * t.M * t.N * t.R * t.taps * t.nz);
  }
  return ft_check_launch("splitk_reduce");
}
