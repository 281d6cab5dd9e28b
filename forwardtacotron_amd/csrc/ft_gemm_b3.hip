// fp32 GEMMs on the bf16 matrix pipe ("bf16x3"): same task descriptors, row maps, taps, chains and epilogues as
// ft_gemm.hip, but every fp32 operand is split EXACTLY into three bf16 pieces while it is staged into LDS
//     x = hi + mid + lo      (truncation splits: each remainder is exact in fp32, 3 x 8 mantissa bits = fp32's 24)
// and a product keeps the six terms down to 2^-16 relative (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi), accumulated
// in fp32 by v_mfma_f32_32x32x16_bf16.  The dropped terms are <= 2^-23 |a||b| -- the size of fp32's own product
// rounding: measured against an f64 reference the result is as close as (slightly closer than) the
// v_mfma_f32_32x32x2_f32 path (lab/gemm_lab_b3.h: 1.9e-5 vs 2.3e-5 max error at K = 512, |C| up to 26).
// bf16 MFMA runs at 16x the fp32 MFMA rate and six of them replace one, so the matrix pipe has 2.67x fewer cycles to
// spend; what bounds the kernel instead is the staging work (split = ~5 VALU ops per element, LDS traffic).
//
// Layout: operand rows are K-contiguous in LDS, [row][plane hi|mid|lo][32 k] bf16 + 16 B pad = 208 B per row, so an
// MFMA fragment (8 consecutive k of one row) is ONE ds_read_b128 and a staged float4 becomes three ds_write_b64.
// One LDS stage of 32 k (53 KB for the 128x128 tile -> two workgroups per CU), two barriers per stage; the next
// stage's global loads are in flight during the MFMAs.  Only the NT form (both operands K-contiguous) exists: callers
// with an [K][N] operand hand in its transpose.
//
// NP = 1 instantiations are the GENUINE bf16 path (BASELINE configs[2], FastPitch "bs=32 bf16"): operands rounded to
// nearest bf16 while staged (v_cvt_pk_bf16_f32), ONE v_mfma_f32_32x32x16_bf16 per product, fp32 accumulation and fp32
// outputs -- what autocast-style bf16 matmuls compute.  Selected per process with ft_set_gemm_precision(1); the
// fp32-exact three-plane form (NP = 3) is the default everywhere else.
#include "ft_gemm.h"
#include "ft_split.h"

namespace {

constexpr int BK = 32;
constexpr int RS = 104;          // LDS row stride in bf16 elements: 3 planes x 32 + 8 pad

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 ld4_sel(const float* p, const float* safe, bool ok) {
  const float4 v = *reinterpret_cast<const float4*>(ok ? p : safe);
  return ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
}

// exact split of 4 floats into hi / mid / lo bf16 quadruples.  Round-to-nearest pieces through v_cvt_pk_bf16_f32 (two
// elements per instruction, already packed): hi = rn(x), r1 = x - hi (exact: hi shares x's leading bits), mid = rn(r1),
// r2 = r1 - mid (exact), lo = rn(r2) -- |x - hi - mid - lo| <= 2^-27 |x|, tighter than the truncation split's 2^-24, at
// 5.5 VALU instructions per element instead of 8 (and / sub / and / sub + three shift-packs): the split is what bounds
// this kernel, not the matrix pipe.
// (definition shared with the once-per-step weight split: ft_split.h)
// The file is built with -fno-slp-vectorize: the SLP pass packs the two subtractions into a v_pk_add_f32, which costs
// more than two v_sub_f32 beside MFMAs.
__device__ __forceinline__ unsigned rn_pack(float a, float b) { return ft_rn_pack(a, b); }
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  ft_split_pair(a, b, hi, mid, lo);
}
__device__ __forceinline__ void split3(const float4& v, u16x4& hi, u16x4& mid, u16x4& lo) {
  unsigned h[2], m[2], l[2];
  split_pair(v.x, v.y, h[0], m[0], l[0]);
  split_pair(v.z, v.w, h[1], m[1], l[1]);
  hi = __builtin_bit_cast(u16x4, uint2{h[0], h[1]});
  mid = __builtin_bit_cast(u16x4, uint2{m[0], m[1]});
  lo = __builtin_bit_cast(u16x4, uint2{l[0], l[1]});
}

__device__ __forceinline__ void store_split(unsigned short* row, const float4& v) {
  u16x4 h, m, l;
  split3(v, h, m, l);
  *reinterpret_cast<u16x4*>(row) = h;
  *reinterpret_cast<u16x4*>(row + 32) = m;
  *reinterpret_cast<u16x4*>(row + 64) = l;
}

// NP = 1: round-to-nearest bf16 of 4 floats -> plane 0 only
__device__ __forceinline__ void store_rn(unsigned short* row, const float4& v) {
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  bf16x4 r = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
  *reinterpret_cast<bf16x4*>(row) = r;
}

// epilogue shared by the two NT kernels: bias / ReLU / BatchNorm statistics / scale-shift / accumulate, same accumulator
// layout as ft_gemm_rows_kernel.  `smem` = the (now idle) LDS tiles, >= 4 KB.
template <int TM, int TN>
__device__ __forceinline__ void rows_b3_epilogue(const FtGemmTask& T, float* TC, f32x16 (&acc)[TM][TN], unsigned short* smem,
                                                 int m0, int n0, int tM, int tN, int tid, int mtile,
                                                 const float* relu_mask = nullptr) {
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int half = lane >> 5, l31 = lane & 31;
  const float* ebias = T.bias;
  const float* escale = T.scale;
  const float* eshift = T.shift;
  const bool erelu = T.relu != 0, eacc = T.accumulate != 0;
  const long ldc = T.ldc, cbst = T.cmap.bstride, ctst = T.cmap.tstride;
  const int cTlog = T.cmap.Tlog;
  double* const stat = (TM == 2 && TN == 2) ? T.stat : nullptr;
  const int sTlog = T.amap.Tlog, sTv = T.stat_tvalid;
  const int srow0 = m0 + wm * 32 * TM + 4 * half;          // first row of this lane; t = row % Tlog tracked without divisions
  const int stb = stat ? srow0 % sTlog : 0;
  const bool ssmall = sTlog < 32 * TM;
  double* sred = reinterpret_cast<double*>(smem);          // [wm 2][128 columns][2] doubles = 4 KB of the (now idle) tiles
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * 32 * TN + 32 * j + l31;
    const bool col_ok = col < tN;
    const float bv = (ebias && col_ok) ? ebias[col] : 0.f;
    const float sc = (escale && col_ok) ? escale[col] : 1.f;
    const float sh = (escale && col_ok) ? eshift[col] : 0.f;
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 32 * TM + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * half;
        if (row >= tM || !col_ok) continue;
        long crow = row;
        if (cbst != 0) {
          const int cb = row / cTlog;
          crow = (long)cb * cbst + (long)(row - cb * cTlog) * ctst;
        }
        float* cp = TC + crow * ldc + col;
        float v = acc[i][j][e] + bv;
        if (erelu) v = fmaxf(v, 0.f);
        if (stat) {
          int tt = stb + (row - srow0);                    // < 2 * Tlog unless Tlog is tiny
          if (ssmall) tt %= sTlog;
          else if (tt >= sTlog) tt -= sTlog;
          if (tt < sTv) {
            s0 += (double)v;
            s1 += (double)v * (double)v;
          }
        }
        if (escale) v = v * sc + sh;
        if (relu_mask && !(relu_mask[crow * ldc + col] > 0.f)) v = 0.f;
        if (eacc) v += *cp;
        *cp = v;
      }
    }
    if (stat) {            // lanes l31 and l31 + 32 hold the two row halves of a column: fold, then across the wm waves
      s0 += __shfl_xor(s0, 32, 64);
      s1 += __shfl_xor(s1, 32, 64);
      if (half == 0) {
        double* q = sred + ((wm * 128) + wn * 32 * TN + 32 * j + l31) * 2;
        q[0] = s0;
        q[1] = s1;
      }
    }
  }
  if (stat) {
    __syncthreads();
    if (tid < 128 && n0 + tid < tN) {
      double* o = stat + ((long)mtile * T.stat_ld + T.stat_col0 + n0 + tid) * 2;
      o[0] = sred[tid * 2] + sred[(128 + tid) * 2];
      o[1] = sred[tid * 2 + 1] + sred[(128 + tid) * 2 + 1];
    }
  }
}

template <int TM, int TN, int NP>
__global__ __launch_bounds__(256, 2) void ft_gemm_rows_b3_kernel(FtGemmBatch batch) {
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
  constexpr int PA = BM / 32, PB = BN / 32;
  __shared__ __attribute__((aligned(16))) unsigned short smem[(BM + BN) * RS];
  unsigned short* As = smem;
  unsigned short* Bs = smem + BM * RS;

  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  if (m0 >= T.M || n0 >= T.N) return;
  const int tid = threadIdx.x;
  const int kq = tid & 7, rr = tid >> 3;

  // task fields in locals (see ft_gemm.hip); re-filled when the loader moves to the next task of a chain
  const int tM = T.M, tN = T.N;
  const int nchain = batch.chain > 1 ? batch.chain : 1;
  int tK, taps, Tvalid, shift0, sstep, kch;
  long lda, btap, atst;
  const float* curA;
  const float* curB;
  int a_t[PA];
  bool a_ok[PA], b_ok[PB];
  const float* a_row[PA];
  const float* b_row[PB];
  auto setup = [&](const FtGemmTask& S, const float* SA, const float* SB) {
    curA = SA;
    curB = SB;
    tK = S.K;
    taps = S.taps;
    lda = S.lda;
    btap = S.b_tap_stride;
    Tvalid = S.amap.Tvalid;
    shift0 = S.amap.shift0;
    sstep = S.amap.shift_step;
    atst = S.amap.tstride;
    kch = (tK + BK - 1) / BK;
    const int Tlog = S.amap.Tlog;
    const long abst = S.amap.bstride, ldb = S.ldb;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      const int m = m0 + rr + 32 * p;
      a_ok[p] = m < tM;
      const int b = m / Tlog;
      a_t[p] = m - b * Tlog;
      a_row[p] = SA + ((long)b * abst + (long)a_t[p] * atst) * lda + 4 * kq;
    }
#pragma unroll
    for (int p = 0; p < PB; ++p) {
      const int n = n0 + rr + 32 * p;
      b_ok[p] = n < tN;
      b_row[p] = SB + (long)n * ldb + 4 * kq;
    }
  };
  setup(T, TA, TB);
  int nch = 0;
  for (int i = 0; i < nchain; ++i) nch += batch.t[i].taps * ((batch.t[i].K + BK - 1) / BK);
  if (nchain == 1) nch = taps * kch;

  int l_task = 0, l_tap = 0, l_kc = 0;           // loader cursor
  float4 ra[PA], rb[PB];
  auto load_stage = [&]() {
    const int j = l_tap;
    const int k0 = l_kc * BK;
    const int shift = shift0 + j * sstep;
    const bool kok = k0 + 4 * kq < tK;
    const long aoff = (long)shift * atst * lda + k0;
    const long boff = (long)j * btap + k0;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      const int ts = a_t[p] + shift;
      ra[p] = ld4_sel(a_row[p] + aoff, curA, a_ok[p] & (ts >= 0) & (ts < Tvalid) & kok);
    }
#pragma unroll
    for (int p = 0; p < PB; ++p) rb[p] = ld4_sel(b_row[p] + boff, curB, b_ok[p] & kok);
    if (++l_kc == kch) {
      l_kc = 0;
      if (++l_tap == taps) {
        l_tap = 0;
        if (++l_task < nchain) setup(batch.t[l_task], batch.t[l_task].A, batch.t[l_task].B);
      }
    }
  };
  auto store_stage = [&]() {
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      if constexpr (NP == 1) store_rn(As + (rr + 32 * p) * RS + 4 * kq, ra[p]);
      else store_split(As + (rr + 32 * p) * RS + 4 * kq, ra[p]);
    }
#pragma unroll
    for (int p = 0; p < PB; ++p) {
      if constexpr (NP == 1) store_rn(Bs + (rr + 32 * p) * RS + 4 * kq, rb[p]);
      else store_split(Bs + (rr + 32 * p) * RS + 4 * kq, rb[p]);
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

  const unsigned short* ap = As + (wm * 32 * TM + l31) * RS + 8 * half;
  const unsigned short* bp = Bs + (wn * 32 * TN + l31) * RS + 8 * half;
  load_stage();
  store_stage();
  if (nch > 1) load_stage();
  __syncthreads();
  for (int c = 0; c < nch; ++c) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {                  // two k-steps of 16 per stage
      bf16x8 a[TM][NP], b[TN][NP];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
          a[i][pl] = *reinterpret_cast<const bf16x8*>(ap + 32 * i * RS + 32 * pl + 16 * ks);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
          b[j][pl] = *reinterpret_cast<const bf16x8*>(bp + 32 * j * RS + 32 * pl + 16 * ks);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {                // small terms first
          if constexpr (NP == 1) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
            continue;
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][NP - 1], b[j][0], acc[i][j], 0, 0, 0);
          constexpr int P1 = NP > 1 ? 1 : 0, P2 = NP - 1;
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][P2], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][P1], b[j][P1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][P1], b[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][P1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    if (c + 1 < nch) {
      store_stage();                                  // its registers were loaded one iteration ago
      if (c + 2 < nch) load_stage();
      __syncthreads();
    }
  }

  if (batch.hw_mode) {                                  // highway gate / gate gradient in the epilogue (ft_gemm.h)
    ft_highway_epilogue<TM, TN>(batch, T, TC, acc, reinterpret_cast<float*>(smem), m0, n0, tid);
    return;
  }
  rows_b3_epilogue<TM, TN>(T, TC, acc, smem, m0, n0, tM, tN, tid, blockIdx.x, batch.relu_mask);
}

// ---------------------------------------------------------------------------------------------------
// The 128x128 NT kernel, software-pipelined ("b3p").  The two-barrier loop above tops out where the guide's "step-3
// structure" does (35 % of the bf16 pipe: MFMA busy 46-53 % in profiles/r02_pmc_report.txt): a wave alternates between
// an MFMA phase and a staging phase (split VALU + ds_write + two barriers) and the two workgroups of a CU drift into
// phase with each other.  Here a stage is 16 k deep and LDS holds TWO of them (2 x 24 KB, still two workgroups per CU),
// so one iteration = [issue the global loads of stage c+2] [fragments of stage c] [24 MFMAs, with the split + ds_write of
// stage c+1 issued in their shadow -- one scheduling region, no branch inside] [ONE barrier].  Plain VGPR loads stay in
// flight across __syncthreads() (only LDS-DMA would force its vmcnt(0)), so a load has two iterations to land.
// Layout: row = [plane][16 k] bf16, 96 B (48 B for NP = 1) -- an odd multiple of 16/32 B, which spreads 16 consecutive
// lanes' ds_read_b128 / ds_write_b128 over all 64 banks with no padding column.  A thread stages 8 k of one A row and one
// B row per stage (two float4 each, k0 + 4h and k0 + 8 + 4h so the wave's two loads are 32 contiguous bytes per row);
// the k ORDER inside a stage is therefore permuted, identically for A and B, which a dot product does not see.
//
// (A variant that staged pre-split weight operands -- split once per step instead of by every row tile -- was built in
// round 2, bit-identical and SLOWER: 0.320 -> 0.340 ms on the postnet bank; removed in round 3, lab/NOTES.md.)
template <int NP>
__global__ __launch_bounds__(256, 2) void ft_gemm_rows_b3p_kernel(FtGemmBatch batch) {
  constexpr int TM = 2, TN = 2, BM = 128, BN = 128, SK = 16;
  constexpr int RW = NP == 3 ? 48 : 24;            // row stride, bf16 elements
  constexpr int BUF = (BM + BN) * RW;
  // XCD-aware tile walk: workgroups are dealt to the 8 XCDs round-robin by linear id, each XCD has its own L2.  Within
  // every z slice (= task or batch instance: slices differ in work per tile, so each is spread over all XCDs) the
  // workgroups that share an XCD take a CONTIGUOUS range of the slice's (row tile, column tile) space, column tile
  // fastest: the workgroups resident on one XCD at a time share their A row tiles (and stream the same k window of B)
  // through that L2 instead of every XCD fetching every operand tile again (lab/gemm_b3p_lab.hip).  Bijective for any
  // grid: `xcd` only labels the workgroups of the slice that share an XCD.
  int bx, by;
  const int bz = blockIdx.z;
  {
    const int gy = gridDim.y;
    const int nwg = gridDim.x * gy;
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int xcd = lin & 7, q = nwg >> 3, r = nwg & 7;
    const int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
    bx = w / gy;
    by = w - bx * gy;
  }
  const bool zbatch = batch.t[0].nz > 1;
  const int ksplit = batch.ksplit > 1 ? batch.ksplit : 1;       // split-K: z = stage range (FtGemmBatch.ksplit_slab)
  const FtGemmTask& T = batch.t[(zbatch || ksplit > 1) ? 0 : bz];
  const float* TA = T.A;
  const float* TB = T.B;
  float* TC = T.C;
  if (zbatch) {
    const int z0 = bz / T.nz1, z1 = bz - z0 * T.nz1;
    TA += z0 * T.sA0 + z1 * T.sA1;
    TB += z0 * T.sB0 + z1 * T.sB1;
    TC += z0 * T.sC0 + z1 * T.sC1;
  }
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * BUF];

  const int m0 = bx * BM, n0 = by * BN;
  if (m0 >= T.M || n0 >= T.N) return;
  const int tid = threadIdx.x;
  const int h = tid & 1, rr = tid >> 1;            // k half of the stage, tile row (A and B)

  const int tM = T.M, tN = T.N;
  const int nchain = batch.chain > 1 ? batch.chain : 1;
  // Operands are fetched with buffer loads: the descriptor base is the tile's first row at the stage's (tap shift, k0) --
  // all scalar -- and a lane's voffset is constant per task.  A row outside the time window / the tile's M, N range or a
  // k beyond K gets an out-of-range voffset and the hardware returns zeros: no exec-masked branch per load (what hipcc
  // makes of "ok ? *p : 0", one basic block per load) and no select on the loaded value.
  constexpr unsigned OOB = 0x80000000u, NREC = 0x7fffffffu;
  int tK, taps, Tvalid, shift0, sstep, kch, a_t;
  long lda, btap, atst;
  const float* a_base;
  const float* b_base;
  bool a_ok;
  unsigned a_voff, vb, vbad0, vbad1;                 // bit 31 of a voffset = out of range = the load returns zeros
  unsigned va0, va1, vb0, vb1;                       // the voffsets of the stage under the cursor (two 16-B loads per row)
  bool ktail;                                        // K is not a multiple of the stage depth: the last chunk is masked
  const float* tapA;                                 // descriptor bases per (task, tap); the k position of a stage goes
  const float* tapB;                                 // into soffset
  int l_task = 0, l_tap = 0, l_kc = 0;             // loader cursor
  // per-tap state: everything a stage's loads need except k0 -- kept out of the per-stage path, which runs between a
  // barrier and the next stage's requests with the matrix pipe draining
  auto mask_tail = [&]() {
    va0 |= vbad0;
    va1 |= vbad1;
    vb0 |= vbad0;
    vb1 |= vbad1;
  };
  auto set_tap = [&]() {
    const int shift = shift0 + l_tap * sstep;
    const int ts = a_t + shift;
    va0 = (a_ok & (ts >= 0) & (ts < Tvalid)) ? a_voff : OOB;
    va1 = va0 + 32u;
    vb0 = vb;
    vb1 = vb + 32u;
    if (ktail && kch == 1) mask_tail();
    tapA = a_base + (long)shift * atst * lda;
    tapB = b_base + (long)l_tap * btap;
  };
  auto setup = [&](const FtGemmTask& S, const float* SA, const float* SB) {
    tK = S.K;
    taps = S.taps;
    lda = S.lda;
    btap = S.b_tap_stride;
    Tvalid = S.amap.Tvalid;
    shift0 = S.amap.shift0;
    sstep = S.amap.shift_step;
    atst = S.amap.tstride;
    kch = (tK + SK - 1) / SK;
    const int Tlog = S.amap.Tlog;
    const long abst = S.amap.bstride;
    const int bt0 = m0 / Tlog;
    long row0 = (long)bt0 * abst + (long)(m0 - bt0 * Tlog) * atst;            // the tile's lowest physical row (scalar):
    if (m0 - bt0 * Tlog + BM > Tlog && (long)(bt0 + 1) * abst < row0)         // its first one, or -- time-major layouts --
      row0 = (long)(bt0 + 1) * abst;                                          // the head of the next batch item
    a_base = SA + row0 * lda;
    const int m = m0 + rr;
    a_ok = m < tM;
    const int b = m / Tlog;
    a_t = m - b * Tlog;
    a_voff = (unsigned)((((long)b * abst + (long)a_t * atst) - row0) * lda * 4) + 16u * h;
    b_base = SB + (long)n0 * S.ldb;
    vb = n0 + rr < tN ? (unsigned)((long)rr * S.ldb * 4) + 16u * h : OOB;
    const int kl = (kch - 1) * SK;                   // only the last k chunk of a tap can reach past K
    ktail = tK % SK != 0;
    vbad0 = kl + 4 * h < tK ? 0u : OOB;
    vbad1 = kl + 8 + 4 * h < tK ? 0u : OOB;
    set_tap();
  };
  setup(T, TA, TB);
  int nch = 0;
  for (int i = 0; i < nchain; ++i) nch += batch.t[i].taps * ((batch.t[i].K + SK - 1) / SK);
  if (nchain == 1) nch = taps * kch;
  if (ksplit > 1) {                                  // this workgroup's range of the stage sequence: move the cursor there
    const int per = (nch + ksplit - 1) / ksplit;
    int rem = bz * per;
    nch = nch - rem < per ? nch - rem : per;
    if (nch < 0) nch = 0;
    while (l_task < nchain - 1) {
      const int n = batch.t[l_task].taps * ((batch.t[l_task].K + SK - 1) / SK);
      if (rem < n) break;
      rem -= n;
      ++l_task;
    }
    if (l_task > 0) setup(batch.t[l_task], batch.t[l_task].A, batch.t[l_task].B);
    l_tap = rem / kch;
    l_kc = rem - l_tap * kch;
    set_tap();
    if (ktail && kch > 1 && l_kc == kch - 1) mask_tail();
  }

  struct Regs { u32x4 a0, a1, b0, b1, b2; };
  // issue the four loads of the stage under the cursor.  No branch in here (the loads, the split of the older register
  // set and the MFMAs share one scheduling region, and the compiler's vmcnt counting stops at a branch).  A row outside
  // the tap's time window / the tile's M, N range or a k beyond K has bit 31 set in its voffset: out of range for the
  // descriptor, the hardware returns zeros -- no exec-masked branch per load (what hipcc makes of "ok ? *p : 0") and no
  // select on the loaded value.
  auto load_stage = [&](Regs& R) {
    // the descriptors are rebuilt from readfirstlane'd halves: a loop-carried descriptor is not provably uniform to
    // hipcc, which would wrap every load in a waterfall loop
    auto uniform = [](const float* q) {
      const unsigned long long u = (unsigned long long)q;
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
      return reinterpret_cast<float*>(((unsigned long long)hi << 32) | lo);
    };
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(uniform(tapA), 0, NREC, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(uniform(tapB), 0, NREC, 0x00020000);
    const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(l_kc * SK * 4);
    R.a0 = __builtin_amdgcn_raw_buffer_load_b128(rsA, va0, soff, 0);
    R.a1 = __builtin_amdgcn_raw_buffer_load_b128(rsA, va1, soff, 0);
    R.b0 = __builtin_amdgcn_raw_buffer_load_b128(rsB, vb0, soff, 0);
    R.b1 = __builtin_amdgcn_raw_buffer_load_b128(rsB, vb1, soff, 0);
    __builtin_amdgcn_sched_barrier(0);              // keep the requests at the top of the iteration (hipcc sinks them to
  };                                                // the bottom otherwise: zero prefetch distance)
  auto advance = [&]() {
    if (++l_kc == kch) {
      l_kc = 0;
      if (++l_tap == taps) {
        l_tap = 0;
        if (++l_task < nchain) setup(batch.t[l_task], batch.t[l_task].A, batch.t[l_task].B);
      } else {
        set_tap();
      }
    } else if (ktail && l_kc == kch - 1) {
      mask_tail();
    }
  };
  // 8 floats -> NP pieces of 8 bf16, one ds_write_b128 each
  auto put8 = [&](unsigned short* row, const u32x4& uu, const u32x4& vv) {
    const float4 u = __builtin_bit_cast(float4, uu), v = __builtin_bit_cast(float4, vv);
    if constexpr (NP == 1) {
      const uint4 w = make_uint4(rn_pack(u.x, u.y), rn_pack(u.z, u.w), rn_pack(v.x, v.y), rn_pack(v.z, v.w));
      *reinterpret_cast<uint4*>(row) = w;
    } else {
      unsigned hi[4], mid[4], lo[4];
      split_pair(u.x, u.y, hi[0], mid[0], lo[0]);
      split_pair(u.z, u.w, hi[1], mid[1], lo[1]);
      split_pair(v.x, v.y, hi[2], mid[2], lo[2]);
      split_pair(v.z, v.w, hi[3], mid[3], lo[3]);
      *reinterpret_cast<uint4*>(row) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
      *reinterpret_cast<uint4*>(row + 16) = make_uint4(mid[0], mid[1], mid[2], mid[3]);
      *reinterpret_cast<uint4*>(row + 32) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
  };
  auto store_stage = [&](const Regs& R, unsigned short* buf) {
    put8(buf + rr * RW + 8 * h, R.a0, R.a1);
    put8(buf + (BM + rr) * RW + 8 * h, R.b0, R.b1);
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

  const int aoff = (wm * 64 + l31) * RW + 8 * half;
  const int boff = (BM + wn * 64 + l31) * RW + 8 * half;
  // the split of the next stage goes into the shadow of this stage's MFMAs: 1 MFMA (8 passes) : a few VALU
  auto interleave = [&]() {
    __builtin_amdgcn_sched_group_barrier(0x100, 4 * NP, 0);  // the fragment reads
#pragma unroll
    for (int i = 0; i < TM * TN * (NP == 3 ? 6 : 1); ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, NP == 3 ? 5 : 8, 0);   // VALU
    }
  };
  // TWO fragment sets: the fragments of stage c+1 are read during the MFMAs of stage c (from the buffer the barrier at
  // the end of iteration c-1 published), so no MFMA waits for an LDS read behind a barrier.  Reads come BEFORE the next
  // stage's ds_writes in program order: the compiler cannot prove the two LDS buffers distinct and would otherwise keep
  // them behind the whole split + write block.
  bf16x8 fa[TM][NP], fb[TN][NP], ga[TM][NP], gb[TN][NP];
  auto read_into = [&](bf16x8 (&xa)[TM][NP], bf16x8 (&xb)[TN][NP], const unsigned short* buf) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) xa[i][pl] = *reinterpret_cast<const bf16x8*>(buf + aoff + 32 * i * RW + 16 * pl);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) xb[j][pl] = *reinterpret_cast<const bf16x8*>(buf + boff + 32 * j * RW + 16 * pl);
  };
  // term-major order: consecutive MFMAs go to different accumulators; each accumulator receives its six terms small-first
  auto mfma_on = [&](bf16x8 (&xa)[TM][NP], bf16x8 (&xb)[TN][NP]) {
    constexpr int PA_[6] = {2, 0, 1, 1, 0, 0}, PB_[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int term = 0; term < (NP == 3 ? 6 : 1); ++term)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int pa = NP == 3 ? PA_[term] : 0, pb = NP == 3 ? PB_[term] : 0;
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[i][pa], xb[j][pb], acc[i][j], 0, 0, 0);
        }
  };

  unsigned short* const buf0 = smem;
  unsigned short* const buf1 = smem + BUF;
  Regs R0, R1;
  if (nch > 0) {
  load_stage(R0);
  advance();
  if (nch > 1) {
    load_stage(R1);
    advance();
  }
  store_stage(R0, buf0);                             // stage 0
  __syncthreads();
  if (nch > 2) {
    load_stage(R0);                                  // stage 2
    advance();
  }
  read_into(fa, fb, buf0);
  if (nch > 1) store_stage(R1, buf1);                // stage 1
  __syncthreads();
  // iteration c (even): fragments of stage c in (fa, fb), stage c+1 in buf1, stage c+2 in R0 (in flight), cursor at c+3.
  //   first half : request c+3 -> R1 | read c+1 -> (ga, gb) | split + write c+2 -> buf0 | MFMA c     | barrier
  //   second half: request c+4 -> R0 | read c+2 -> (fa, fb) | split + write c+3 -> buf1 | MFMA c+1   | barrier
  int c = 0;
  for (; c + 4 < nch; c += 2) {
    load_stage(R1);
    read_into(ga, gb, buf1);
    store_stage(R0, buf0);
    mfma_on(fa, fb);
    interleave();
    __syncthreads();
    advance();
    load_stage(R0);
    read_into(fa, fb, buf0);
    store_stage(R1, buf1);
    mfma_on(ga, gb);
    interleave();
    __syncthreads();
    advance();
  }
  const int left = nch - c;                          // 1 .. 4 stages, same state as at the top of an iteration
  if (left == 4) {
    load_stage(R1);                                  // c+3
    read_into(ga, gb, buf1);
    store_stage(R0, buf0);                           // c+2
    mfma_on(fa, fb);
    __syncthreads();
    read_into(fa, fb, buf0);
    store_stage(R1, buf1);                           // c+3
    mfma_on(ga, gb);
    __syncthreads();
    read_into(ga, gb, buf1);
    mfma_on(fa, fb);
    mfma_on(ga, gb);
  } else if (left == 3) {
    read_into(ga, gb, buf1);
    store_stage(R0, buf0);                           // c+2
    mfma_on(fa, fb);
    __syncthreads();
    read_into(fa, fb, buf0);
    mfma_on(ga, gb);
    mfma_on(fa, fb);
  } else if (left == 2) {
    read_into(ga, gb, buf1);
    mfma_on(fa, fb);
    mfma_on(ga, gb);
  } else {
    mfma_on(fa, fb);
  }
  }                                                  // (nch > 0)
  __syncthreads();                                   // the epilogue's statistics scratch aliases the tiles
  if (ksplit > 1) {                                  // raw partial tile -> slab[z][M][N]; ft_ksplit_reduce_kernel finishes
    float* out = batch.ksplit_slab + (long)bz * tM * tN;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * 32 * TN + 32 * j + l31;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = m0 + wm * 32 * TM + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * half;
          if (row < tM && col < tN) out[(long)row * tN + col] = acc[i][j][e];
        }
    }
    return;
  }
  if (batch.hw_mode) {                                // highway gate / gate gradient in the epilogue (ft_gemm.h)
    ft_highway_epilogue<TM, TN>(batch, T, TC, acc, reinterpret_cast<float*>(smem), m0, n0, tid);
    return;
  }
  rows_b3_epilogue<TM, TN>(T, TC, acc, smem, m0, n0, tM, tN, tid, bx, batch.relu_mask);
}

// second launch of a split-K NT product: C[cmap(row)][col] (+)= sum_z slab[z][row][col] + bias, in range order
__global__ __launch_bounds__(256) void ft_ksplit_reduce_kernel(const float* __restrict__ slab, int S, FtGemmTask T) {
  const long idx = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  const long total = (long)T.M * T.N;
  if (idx >= total) return;
  const int row = (int)(idx / T.N), col = (int)(idx - (long)row * T.N);      // N % 4 == 0: the four columns share a row
  float4 a = *reinterpret_cast<const float4*>(slab + idx);
  for (int z = 1; z < S; ++z) {
    const float4 v = *reinterpret_cast<const float4*>(slab + (long)z * total + idx);
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  if (T.bias) {
    a.x += T.bias[col]; a.y += T.bias[col + 1]; a.z += T.bias[col + 2]; a.w += T.bias[col + 3];
  }
  long crow = row;
  if (T.cmap.bstride != 0) {
    const int cb = row / T.cmap.Tlog;
    crow = (long)cb * T.cmap.bstride + (long)(row - cb * T.cmap.Tlog) * T.cmap.tstride;
  }
  float* cp = T.C + crow * T.ldc + col;
  if (T.accumulate) {
    a.x += cp[0]; a.y += cp[1]; a.z += cp[2]; a.w += cp[3];
  }
  cp[0] = a.x; cp[1] = a.y; cp[2] = a.z; cp[3] = a.w;
}


// ---------------------------------------------------------------------------------------------------
// TN (weight gradients): slab[z][m][n] = sum_{r in slice} A[map_a(r)][m] * B[map_b,tap(r)][n].  The contraction index
// r is the ROW index of both operands, while a bf16 MFMA fragment wants 8 consecutive r of one column in a lane.  The
// LDS tiles are therefore kept r-PAIR packed: word [plane][r/2][m] holds the bf16 pieces of rows r (low half) and r+1
// (high half) of column m.  Staging stays coalesced on both sides -- a thread fetches rows r, r+1 of 4 adjacent columns
// (two 16-B loads), splits them and writes one 16-B piece per plane, conflict-free -- and a fragment is four 4-B reads
// one pair-row apart (row stride = BM + 8 words puts the two k-halves of a wave 32 banks apart).  [The first version
// wrote [m][plane][r] rows with 8-B pieces 832 B apart: a 16-way bank conflict that cost what the MFMAs saved.]
template <int TM, int TN, int NP>
__global__ __launch_bounds__(256, 2) void ft_gemm_tn_b3_kernel(FtGemmTNTask T, float* slab, int S, int rows_per_split) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int MQ = BM / 4, NQ = BN / 4;            // column quads per tile
  constexpr int QA = 256 / MQ, QB = 256 / NQ;        // pair-rows covered by one pass of the 256 threads
  constexpr int PA = 16 / QA, PB = 16 / QB;          // passes per stage (BK = 32 rows = 16 pair-rows)
  constexpr int RWA = BM + 8, RWB = BN + 8;          // pair-row stride in 32-bit words
  constexpr int PLA = 16 * RWA, PLB = 16 * RWB;      // plane stride
  __shared__ __attribute__((aligned(16))) unsigned smem[3 * PLA + 3 * PLB];
  unsigned* As = smem;
  unsigned* Bs = smem + 3 * PLA;

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
  const int tM = T.M, tN = T.N;
  const long lda = T.lda, ldb = T.ldb;
  const int aTlog = T.amap.Tlog, bTlog = T.bmap.Tlog, bTvalid = T.bmap.Tvalid;
  const long abst = T.amap.bstride, atst = T.amap.tstride, bbst = T.bmap.bstride, btst = T.bmap.tstride;
  const int ashift = T.amap.shift0 + tap * T.amap.shift_step;
  int bshift = T.bmap.shift0 + tap * T.bmap.shift_step;
  int aTvalid = T.amap.Tvalid;
  if (who.kk > 0) {                           // conv-bank mode (FtGemmTNTask): member of this tile
    bshift = tap - who.kk / 2;
    if (who.kk & 1) aTvalid = T.bankTodd;
  }

  const int amq = tid % MQ, aq0 = tid / MQ;          // this thread's column quad / first pair-row, per operand
  const int bmq = tid % NQ, bq0 = tid / NQ;
  const int am = m0 + 4 * amq, bn = n0 + 4 * bmq;
  const bool am_ok = am < tM, bn_ok = bn < tN;
  // (item, t) of every row this thread stages, tracked incrementally (stages come in order, BK rows apart)
  int a_b[PA][2], a_t[PA][2], b_b[PB][2], b_t[PB][2];
#pragma unroll
  for (int p = 0; p < PA; ++p)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = r_begin + 2 * (aq0 + QA * p) + i;
      a_b[p][i] = r / aTlog;
      a_t[p][i] = r - a_b[p][i] * aTlog;
    }
#pragma unroll
  for (int p = 0; p < PB; ++p)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = r_begin + 2 * (bq0 + QB * p) + i;
      b_b[p][i] = r / bTlog;
      b_t[p][i] = r - b_b[p][i] * bTlog;
    }
  int r_next = r_begin;

  float4 ra[PA][2], rb[PB][2];
  auto load_stage = [&]() {
#pragma unroll
    for (int p = 0; p < PA; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = r_next + 2 * (aq0 + QA * p) + i;
        const int ts = a_t[p][i] + ashift;
        const bool ok = (r < r_end) & (ts >= 0) & (ts < aTvalid) & am_ok;
        ra[p][i] = ld4_sel(TA + ((long)a_b[p][i] * abst + (long)ts * atst) * lda + am, TA, ok);
        a_t[p][i] += BK;
        while (a_t[p][i] >= aTlog) {
          a_t[p][i] -= aTlog;
          ++a_b[p][i];
        }
      }
#pragma unroll
    for (int p = 0; p < PB; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = r_next + 2 * (bq0 + QB * p) + i;
        const int ts = b_t[p][i] + bshift;
        const bool ok = (r < r_end) & (ts >= 0) & (ts < bTvalid) & bn_ok;
        rb[p][i] = ld4_sel(TB + ((long)b_b[p][i] * bbst + (long)ts * btst) * ldb + bn, TB, ok);
        b_t[p][i] += BK;
        while (b_t[p][i] >= bTlog) {
          b_t[p][i] -= bTlog;
          ++b_b[p][i];
        }
      }
    r_next += BK;
  };
  // rows (even, odd) of 4 columns -> three 16-B pieces of pair-packed bf16
  auto store_pair = [&](unsigned* dst, int plane_stride, const float4& ev, const float4& od) {
    const float e[4] = {ev.x, ev.y, ev.z, ev.w}, o[4] = {od.x, od.y, od.z, od.w};
    if constexpr (NP == 1) {              // bf16 path: one plane, round to nearest
      typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
      uint4 one;
      unsigned* po = reinterpret_cast<unsigned*>(&one);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bf16x2 pr = {(__bf16)e[c], (__bf16)o[c]};        // low half = even row, high half = odd row
        po[c] = __builtin_bit_cast(unsigned, pr);
      }
      *reinterpret_cast<uint4*>(dst) = one;
      return;
    }
    uint4 hi, mid, lo;
    unsigned* ph = reinterpret_cast<unsigned*>(&hi);
    unsigned* pm = reinterpret_cast<unsigned*>(&mid);
    unsigned* pl = reinterpret_cast<unsigned*>(&lo);
#pragma unroll
    for (int c = 0; c < 4; ++c) split_pair(e[c], o[c], ph[c], pm[c], pl[c]);       // word = odd row << 16 | even row
    *reinterpret_cast<uint4*>(dst) = hi;
    *reinterpret_cast<uint4*>(dst + plane_stride) = mid;
    *reinterpret_cast<uint4*>(dst + 2 * plane_stride) = lo;
  };
  auto store_stage = [&]() {
#pragma unroll
    for (int p = 0; p < PA; ++p) store_pair(As + (aq0 + QA * p) * RWA + 4 * amq, PLA, ra[p][0], ra[p][1]);
#pragma unroll
    for (int p = 0; p < PB; ++p) store_pair(Bs + (bq0 + QB * p) * RWB + 4 * bmq, PLB, rb[p][0], rb[p][1]);
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
    // fragment of sub-step ks: pair-rows 8 ks + 4 half + {0..3}, column (tile column) + l31
    const unsigned* ap = As + 4 * half * RWA + wm * 32 * TM + l31;
    const unsigned* bp = Bs + 4 * half * RWB + wn * 32 * TN + l31;
    const int nch = (r_end - r_begin + BK - 1) / BK;
    load_stage();
    store_stage();
    if (nch > 1) load_stage();
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 a[TM][NP], b[TN][NP];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int pl = 0; pl < NP; ++pl) {
            const unsigned* q = ap + pl * PLA + 8 * ks * RWA + 32 * i;
            const uint4 w = make_uint4(q[0], q[RWA], q[2 * RWA], q[3 * RWA]);
            a[i][pl] = __builtin_bit_cast(bf16x8, w);
          }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int pl = 0; pl < NP; ++pl) {
            const unsigned* q = bp + pl * PLB + 8 * ks * RWB + 32 * j;
            const uint4 w = make_uint4(q[0], q[RWB], q[2 * RWB], q[3 * RWB]);
            b[j][pl] = __builtin_bit_cast(bf16x8, w);
          }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            if constexpr (NP == 1) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
              continue;
            }
            constexpr int P1 = NP > 1 ? 1 : 0, P2 = NP - 1;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][P2], b[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][P2], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][P1], b[j][P1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][P1], b[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][P1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
          }
      }
      __syncthreads();
      if (c + 1 < nch) {
        store_stage();
        if (c + 2 < nch) load_stage();
        __syncthreads();
      }
    }
  }
  float* out = slab + ((long)zts * S + s) * tM * tN;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * 32 * TN + 32 * j + l31;
      if (col >= tN) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 32 * TM + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * half;
        if (row < tM) out[(long)row * tN + col] = acc[i][j][e];
      }
    }
}


// ---------------------------------------------------------------------------------------------------
// The 128x128 TN kernel, software-pipelined ("tn_b3p") -- the structure of ft_gemm_rows_b3p_kernel applied to the weight
// gradients: stages of 16 contraction rows (8 pair-rows), TWO LDS buffers (2 x 25.5 KB, two workgroups per CU) and TWO
// fragment sets, one barrier per stage, the split + ds_write of stage c+2 and the row bookkeeping of stage c+3 issued in
// the shadow of stage c's 24 MFMAs.  Operands arrive by buffer loads whose out-of-range voffset returns zeros: a row past
// the split's end, outside the tap's time window or a column quad beyond M / N just carries bit 31 -- the two-barrier
// kernel above wraps each of its 8 loads per stage in an exec-masked block with a `while` behind it.  A thread stages
// rows (2 pr, 2 pr + 1) of 4 adjacent columns of A and of B; it tracks t = row % Tlog of its two rows and their byte
// offsets incrementally (a stage advances by 16 rows: at most one wrap into the next item, hence Tlog >= 16), all with
// selects.  Same arithmetic in the same order as the kernel above (16 rows per MFMA step, six terms small-first): the
// slabs are bit-identical.  Requires amap.Tlog == bmap.Tlog >= 16 and operands addressable with 31-bit byte offsets from
// their base (checked by the launcher, which otherwise keeps the two-barrier kernel).
template <int NP>
__global__ __launch_bounds__(256, 2) void ft_gemm_tn_b3p_kernel(FtGemmTNTask T, float* slab, int S, int rows_per_split) {
  constexpr int TM = 2, TN = 2, BM = 128, BN = 128, SK = 16;
  constexpr int RW = BM + 8;                         // pair-row stride in 32-bit words (A and B tiles alike)
  constexpr int PL = 8 * RW;                         // plane stride: 8 pair-rows per stage
  constexpr int OPW = NP * PL;                       // one operand tile of one stage
  constexpr int BUF = 2 * OPW;
  __shared__ __attribute__((aligned(16))) unsigned smem[2 * BUF];

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
  const int tM = T.M, tN = T.N;
  const int Tlog = T.amap.Tlog;
  const int ashift = T.amap.shift0 + tap * T.amap.shift_step;
  int bshift = T.bmap.shift0 + tap * T.bmap.shift_step;
  int aTvalid = T.amap.Tvalid;
  const int bTvalid = T.bmap.Tvalid;
  if (who.kk > 0) {                           // conv-bank mode (FtGemmTNTask): member of this tile
    bshift = tap - who.kk / 2;
    if (who.kk & 1) aTvalid = T.bankTodd;
  }
  constexpr unsigned OOB = 0x80000000u, NREC = 0x7fffffffu;
  const int mq = tid & 31, pr = tid >> 5;            // column quad, pair-row of the stage
  const bool am_ok = m0 + 4 * mq < tM, bn_ok = n0 + 4 * mq < tN;
  // per staged row i = 0, 1: t within its item, byte offsets of (row, this thread's column quad) in A and B
  int t_[2];
  unsigned oa[2], ob[2];
  const unsigned a_step = (unsigned)(16L * T.amap.tstride * T.lda * 4);
  const unsigned a_wrap = (unsigned)(((long)T.amap.bstride + (16L - Tlog) * T.amap.tstride) * T.lda * 4);
  const unsigned b_step = (unsigned)(16L * T.bmap.tstride * T.ldb * 4);
  const unsigned b_wrap = (unsigned)(((long)T.bmap.bstride + (16L - Tlog) * T.bmap.tstride) * T.ldb * 4);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = r_begin + 2 * pr + i;
    const int b = r / Tlog;
    t_[i] = r - b * Tlog;
    oa[i] = (unsigned)(((long)b * T.amap.bstride + (long)(t_[i] + ashift) * T.amap.tstride) * T.lda * 4) + 16u * mq;
    ob[i] = (unsigned)(((long)b * T.bmap.bstride + (long)(t_[i] + bshift) * T.bmap.tstride) * T.ldb * 4) + 16u * mq;
  }
  int rows_left = r_end - r_begin - 2 * pr;          // rows of this thread's pair still inside the split (> 0: row 0 ok)
  unsigned va[2], vb[2];
  auto voffsets = [&]() {                            // validity of the rows under the cursor -> bit 31
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool in = rows_left > i;
      const bool wa = (unsigned)(t_[i] + ashift) < (unsigned)aTvalid;
      const bool wb = (unsigned)(t_[i] + bshift) < (unsigned)bTvalid;
      va[i] = (in & wa & am_ok) ? oa[i] : OOB;
      vb[i] = (in & wb & bn_ok) ? ob[i] : OOB;
    }
  };
  auto advance = [&]() {                             // the cursor moves 16 rows on
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int t1 = t_[i] + SK;
      const bool w = t1 >= Tlog;
      t_[i] = w ? t1 - Tlog : t1;
      oa[i] += w ? a_wrap : a_step;
      ob[i] += w ? b_wrap : b_step;
    }
    rows_left -= SK;
    voffsets();
  };
  voffsets();

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(TA + m0), 0, NREC, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(TB + n0), 0, NREC, 0x00020000);
  struct Regs { u32x4 a0, a1, b0, b1; };             // rows (even, odd) of A, of B
  auto load_stage = [&](Regs& R) {
    R.a0 = __builtin_amdgcn_raw_buffer_load_b128(rsA, va[0], 0, 0);
    R.a1 = __builtin_amdgcn_raw_buffer_load_b128(rsA, va[1], 0, 0);
    R.b0 = __builtin_amdgcn_raw_buffer_load_b128(rsB, vb[0], 0, 0);
    R.b1 = __builtin_amdgcn_raw_buffer_load_b128(rsB, vb[1], 0, 0);
    __builtin_amdgcn_sched_barrier(0);              // keep the requests at the top of the iteration
  };
  // rows (even, odd) of 4 columns -> NP 16-B pieces of pair-packed bf16 (word = odd row << 16 | even row)
  auto put_pair = [&](unsigned* dst, const u32x4& ev, const u32x4& od) {
    const float4 e = __builtin_bit_cast(float4, ev), o = __builtin_bit_cast(float4, od);
    if constexpr (NP == 1) {
      *reinterpret_cast<uint4*>(dst) = make_uint4(rn_pack(e.x, o.x), rn_pack(e.y, o.y), rn_pack(e.z, o.z), rn_pack(e.w, o.w));
    } else {
      unsigned hi[4], mid[4], lo[4];
      split_pair(e.x, o.x, hi[0], mid[0], lo[0]);
      split_pair(e.y, o.y, hi[1], mid[1], lo[1]);
      split_pair(e.z, o.z, hi[2], mid[2], lo[2]);
      split_pair(e.w, o.w, hi[3], mid[3], lo[3]);
      *reinterpret_cast<uint4*>(dst) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
      *reinterpret_cast<uint4*>(dst + PL) = make_uint4(mid[0], mid[1], mid[2], mid[3]);
      *reinterpret_cast<uint4*>(dst + 2 * PL) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
  };
  auto store_stage = [&](const Regs& R, unsigned* buf) {
    put_pair(buf + pr * RW + 4 * mq, R.a0, R.a1);
    put_pair(buf + OPW + pr * RW + 4 * mq, R.b0, R.b1);
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

  // fragment: pair-rows 4 half + {0..3} (= rows 8 half .. 8 half + 7 of the stage), column (tile column) + l31
  const int aoff = 4 * half * RW + wm * 64 + l31;
  const int boff = OPW + 4 * half * RW + wn * 64 + l31;
  bf16x8 fa[TM][NP], fb[TN][NP], ga[TM][NP], gb[TN][NP];
  auto read_into = [&](bf16x8 (&xa)[TM][NP], bf16x8 (&xb)[TN][NP], const unsigned* buf) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) {
        const unsigned* q = buf + aoff + pl * PL + 32 * i;
        xa[i][pl] = __builtin_bit_cast(bf16x8, make_uint4(q[0], q[RW], q[2 * RW], q[3 * RW]));
      }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) {
        const unsigned* q = buf + boff + pl * PL + 32 * j;
        xb[j][pl] = __builtin_bit_cast(bf16x8, make_uint4(q[0], q[RW], q[2 * RW], q[3 * RW]));
      }
  };
  // per accumulator the six terms small-first, as in the two-barrier kernel (bit-identical slabs); term-major so that
  // consecutive MFMAs go to different accumulators
  auto mfma_on = [&](bf16x8 (&xa)[TM][NP], bf16x8 (&xb)[TN][NP]) {
    constexpr int PA_[6] = {2, 0, 1, 1, 0, 0}, PB_[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int term = 0; term < (NP == 3 ? 6 : 1); ++term)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int pa = NP == 3 ? PA_[term] : 0, pb = NP == 3 ? PB_[term] : 0;
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[i][pa], xb[j][pb], acc[i][j], 0, 0, 0);
        }
  };
  // measured alternatives (LSTM shapes, same box): the compiler's own schedule -4 %, all fragment reads in front of the
  // MFMAs (the NT kernel's pattern) -3 %, reads and VALU in two alternating halves per MFMA: the same
  auto interleave = [&]() {
#pragma unroll
    for (int i = 0; i < TM * TN * (NP == 3 ? 6 : 1); ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                    // MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, NP == 3 ? 2 : 4, 0);      // DS reads of the next stage's fragments
      __builtin_amdgcn_sched_group_barrier(0x002, NP == 3 ? 6 : 12, 0);     // VALU: split + row bookkeeping
    }
  };

  const int nch = r_begin < r_end ? (r_end - r_begin + SK - 1) / SK : 0;
  if (nch > 0) {
    unsigned* const buf0 = smem;
    unsigned* const buf1 = smem + BUF;
    Regs R0, R1;
    load_stage(R0);
    advance();
    if (nch > 1) {
      load_stage(R1);
      advance();
    }
    store_stage(R0, buf0);                           // stage 0
    __syncthreads();
    if (nch > 2) {
      load_stage(R0);                                // stage 2
      advance();
    }
    read_into(fa, fb, buf0);
    if (nch > 1) store_stage(R1, buf1);              // stage 1
    __syncthreads();
    // iteration c (even): fragments of stage c in (fa, fb), stage c+1 in buf1, stage c+2 in R0 (in flight), cursor at c+3
    int c = 0;
    for (; c + 4 < nch; c += 2) {
      load_stage(R1);                                // c+3
      advance();
      read_into(ga, gb, buf1);
      store_stage(R0, buf0);                         // c+2
      mfma_on(fa, fb);
      interleave();
      __syncthreads();
      load_stage(R0);                                // c+4
      advance();
      read_into(fa, fb, buf0);
      store_stage(R1, buf1);                         // c+3
      mfma_on(ga, gb);
      interleave();
      __syncthreads();
    }
    const int left = nch - c;                        // 1 .. 4 stages, same state as at the top of an iteration
    if (left == 4) {
      load_stage(R1);                                // c+3
      read_into(ga, gb, buf1);
      store_stage(R0, buf0);                         // c+2
      mfma_on(fa, fb);
      __syncthreads();
      read_into(fa, fb, buf0);
      store_stage(R1, buf1);                         // c+3
      mfma_on(ga, gb);
      __syncthreads();
      read_into(ga, gb, buf1);
      mfma_on(fa, fb);
      mfma_on(ga, gb);
    } else if (left == 3) {
      read_into(ga, gb, buf1);
      store_stage(R0, buf0);                         // c+2
      mfma_on(fa, fb);
      __syncthreads();
      read_into(fa, fb, buf0);
      mfma_on(ga, gb);
      mfma_on(fa, fb);
    } else if (left == 2) {
      read_into(ga, gb, buf1);
      mfma_on(fa, fb);
      mfma_on(ga, gb);
    } else {
      mfma_on(fa, fb);
    }
  }
  float* out = slab + ((long)zts * S + s) * tM * tN;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * 32 * TN + 32 * j + l31;
      if (col >= tN) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 32 * TM + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * half;
        if (row < tM) out[(long)row * tN + col] = acc[i][j][e];
      }
    }
}

}  // namespace

static long g_tn_pipelined = 0;
extern "C" int ft_gemm_tn_pipelined_launches(void) { return (int)g_tn_pipelined; }

int ft_launch_gemm_tn_b3(const FtGemmTNTask& t, float* slab, int S, int rows_per_split, int tm, dim3 grid,
                         hipStream_t stream) {
  const bool bf16 = ft_gemm_precision() == 1;
  static const bool pipelined = [] {                 // FT_GEMM_TN_PIPE=0: the two-barrier 128x128 kernel (A/B knob)
    const char* e = getenv("FT_GEMM_TN_PIPE");
    return !(e && e[0] == '0');
  }();
  // short splits (few stages per workgroup) do not amortise the deeper prologue: 256 x 256 x 26912 at 224 rows per split
  // ran 54 -> 59 us on the pipelined kernel, the shapes with >= 1000 rows per split gain 17-25 % (lab/tn_pipe_ab.py)
  if (tm == 2 && pipelined && rows_per_split % 16 == 0 && rows_per_split >= 512 && t.amap.Tlog == t.bmap.Tlog &&
      t.amap.Tlog >= 16) {
    // the pipelined kernel addresses rows with 31-bit byte offsets from the operand's base (buffer loads) and walks them
    // with non-negative strides
    auto span_ok = [](const FtRowMap& m, long R, long ld, int taps) {
      if (m.bstride < 0 || m.tstride < 0) return false;
      const long items = (R + m.Tlog - 1) / m.Tlog;
      long lo = m.shift0 < 0 ? m.shift0 : 0, hi = m.shift0 > 0 ? m.shift0 : 0;
      const long last = (long)m.shift0 + (long)(taps - 1) * m.shift_step;
      lo = last < lo ? last : lo;
      hi = last > hi ? last : hi;
      lo -= taps;                                    // conv-bank mode: shifts tap - kk/2 within [-taps, taps]
      hi += taps;
      const long maxrow = (items + 1) * m.bstride + (m.Tlog + 16 + hi) * m.tstride;
      (void)lo;                                      // rows below the base only occur masked (t + shift < 0)
      return (maxrow + 1) * ld * 4 + 1024 < (1L << 31);
    };
    if (span_ok(t.amap, t.R, t.lda, t.taps) && span_ok(t.bmap, t.R, t.ldb, t.taps)) {
      if (bf16) hipLaunchKernelGGL((ft_gemm_tn_b3p_kernel<1>), grid, dim3(256), 0, stream, t, slab, S, rows_per_split);
      else hipLaunchKernelGGL((ft_gemm_tn_b3p_kernel<3>), grid, dim3(256), 0, stream, t, slab, S, rows_per_split);
      ++g_tn_pipelined;
      return FT_OK;
    }
  }
  if (tm == 2) {
    if (bf16) hipLaunchKernelGGL((ft_gemm_tn_b3_kernel<2, 2, 1>), grid, dim3(256), 0, stream, t, slab, S, rows_per_split);
    else hipLaunchKernelGGL((ft_gemm_tn_b3_kernel<2, 2, 3>), grid, dim3(256), 0, stream, t, slab, S, rows_per_split);
  } else {
    if (bf16) hipLaunchKernelGGL((ft_gemm_tn_b3_kernel<1, 1, 1>), grid, dim3(256), 0, stream, t, slab, S, rows_per_split);
    else hipLaunchKernelGGL((ft_gemm_tn_b3_kernel<1, 1, 3>), grid, dim3(256), 0, stream, t, slab, S, rows_per_split);
  }
  return FT_OK;
}

// NT, FAST (16-B aligned operands, K % 4 == 0) launches only; grid / tile choice made by ft_launch_gemm_rows
int ft_launch_ksplit_reduce(const float* slab, int S, const FtGemmTask& t, hipStream_t stream) {
  const long quads = ((long)t.M * t.N + 3) / 4;
  hipLaunchKernelGGL(ft_ksplit_reduce_kernel, dim3(ft_cdiv(quads, 256)), dim3(256), 0, stream, slab, S, t);
  return ft_check_launch("ksplit_reduce");
}

// may these tasks take the pipelined 128x128 kernel (the only one that implements split-K)?
bool ft_rows_b3p_ok(const FtGemmBatch& batch, int ntask) {
  static const bool pipelined = [] {                 // FT_GEMM_PIPE=0: the two-barrier 128x128 kernel (A/B knob)
    const char* e = getenv("FT_GEMM_PIPE");
    return !(e && e[0] == '0');
  }();
  // the pipelined kernel addresses a tile's rows with 32-bit byte offsets from the tile's first row (buffer loads)
  bool span_ok = true;
  for (int i = 0; i < (batch.t[0].nz > 1 ? 1 : ntask) && i < FT_MAX_TASKS; ++i) {
    const FtGemmTask& t = batch.t[i];
    const long tl = t.amap.Tlog, ts = t.amap.tstride, bs = t.amap.bstride;
    const long rows = (tl < 128 ? tl : 128) * ts + (128 / tl + 2) * bs + tl * ts;     // bound on a tile's physical row span
    span_ok = span_ok && ts >= 0 && bs >= 0 && rows * t.lda * 4 < (1L << 31) && 128L * t.ldb * 4 < (1L << 31);
  }
  return pipelined && span_ok;
}

int ft_launch_gemm_rows_b3(const FtGemmBatch& batch, bool big, dim3 grid, hipStream_t stream) {
  const bool bf16 = ft_gemm_precision() == 1;
  const int ntask = batch.chain > 1 ? batch.chain : (batch.ksplit > 1 ? 1 : (int)grid.z);
  const bool p_ok = ft_rows_b3p_ok(batch, ntask);
  FT_REQUIRE(batch.ksplit <= 1 || (big && p_ok), "gemm_rows: split-K planned for a launch the pipelined kernel cannot take");
  if (big && p_ok) {
    if (bf16) hipLaunchKernelGGL((ft_gemm_rows_b3p_kernel<1>), grid, dim3(256), 0, stream, batch);
    else hipLaunchKernelGGL((ft_gemm_rows_b3p_kernel<3>), grid, dim3(256), 0, stream, batch);
  } else if (big) {
    if (bf16) hipLaunchKernelGGL((ft_gemm_rows_b3_kernel<2, 2, 1>), grid, dim3(256), 0, stream, batch);
    else hipLaunchKernelGGL((ft_gemm_rows_b3_kernel<2, 2, 3>), grid, dim3(256), 0, stream, batch);
  } else {
    if (bf16) hipLaunchKernelGGL((ft_gemm_rows_b3_kernel<1, 1, 1>), grid, dim3(256), 0, stream, batch);
    else hipLaunchKernelGGL((ft_gemm_rows_b3_kernel<1, 1, 3>), grid, dim3(256), 0, stream, batch);
  }
  return FT_OK;
}
