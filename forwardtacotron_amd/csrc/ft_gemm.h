// Internal descriptors for the exact-f32 MFMA GEMM / shifted-row (conv) kernels.
#pragma once
#include "ft_common.h"

// logical row r = (b, t), b = r / Tlog, t = r % Tlog:
//   physical row = b*bstride + (t+shift)*tstride, valid iff 0 <= t+shift < Tvalid ; invalid rows read as zeros.
//   shift = shift0 + tap*shift_step.
// batch-major [B,Tbuf,C]: bstride = Tbuf, tstride = 1 ; time-major [T,B,C]: bstride = 1, tstride = B.
struct FtRowMap {
  int Tlog, bstride, tstride, Tvalid, shift0, shift_step;
};

static inline FtRowMap ft_rowmap_identity(int rows) {
  int n = rows > 0 ? rows : 1;
  FtRowMap m = {n, 0, 1, n, 0, 0};
  return m;
}
// rows = B*T logical (b,t) positions stored time-major ([T,B,C]) when tm_B > 0, else plain row-major
static inline FtRowMap ft_rowmap_layout(int rows, int tm_B) {
  if (tm_B <= 0) return ft_rowmap_identity(rows);
  int T = rows / tm_B;
  FtRowMap m = {T > 0 ? T : 1, 1, tm_B, T, 0, 0};
  return m;
}

// C[cmap(M),N] (+)= sum_{tap} Amap_tap(A)[M,K] * B_tap   (+bias, relu, affine)
//   B_NCONTIG = false: B_tap[n][k] at B + tap*b_tap_stride + n*ldb + k   ("NT")
//   B_NCONTIG = true : B_tap[k][n] at B + tap*b_tap_stride + k*ldb + n   ("NN")
struct FtGemmTask {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  const float* scale;   // optional per-column affine applied AFTER relu (eval-mode BatchNorm fold)
  const float* shift;
  long lda, ldb, ldc, b_tap_stride;
  int M, N, K, taps;
  FtRowMap amap;
  FtRowMap cmap;        // output row permutation (Tlog/bstride/tstride only); identity by default
  int relu, accumulate, a_vec, b_vec;
  // NN form only: every row of A is readable up to the next multiple of 4 of K (values there are ignored), so a K that
  // is not a multiple of 4 can still take the 16-B-load path
  int a_rowpad;
  // strided batch (attention: one GEMM per (batch item, head)); only for single-task launches.
  // instance z = blockIdx.z -> (z0, z1) = (z / nz1, z % nz1); X += z0*sX0 + z1*sX1 (floats)
  int nz, nz1;
  long sA0, sA1, sB0, sB1, sC0, sC1;
  // BatchNorm statistics from the epilogue (128x128 bf16-split kernel only; see ft_launch_gemm_rows): per M-tile and
  // column, (sum, sum of squares) of the stored values (after bias / ReLU) over the tile's rows with t = row % amap.Tlog
  // < stat_tvalid, as doubles at stat[((long)mtile * stat_ld + stat_col0 + col) * 2 + {0,1}] -- the layout of the
  // chunked partials ft_bn.hip finalizes.  stat = null: off.
  double* stat;
  int stat_ld, stat_col0, stat_tvalid;
};
#define FT_MAX_TASKS 16
struct FtGemmBatch {
  FtGemmTask t[FT_MAX_TASKS];
  // chain > 1: tasks 0..chain-1 are ONE product, accumulated in registers: C = sum_i sum_tap shift(A_i) * B_i,tap with
  // task 0's M, N, output and epilogue (conv-bank data gradient: every member adds into the same dx)
  int chain;
  int stat_fused;       // out: 1 if the launched kernel produced the tasks' BatchNorm statistics
  // Highway epilogues (HighwayNetwork, common_layers.py:35-40) of a single-task or chained launch with identity output
  // rows -- the north-star's "Highway fused": the gate never runs as a kernel of its own.
  //  hw_mode 1 (forward): B is the 32-row interleave of W1 / W2 (ft_highway_pack), N = 2 * hw_C; output column n is unit
  //    (n / 64) * 32 + n % 32 of y1 (n % 64 < 32) or of y2, so a wave's two 32-column tiles -- or, in the 64-column
  //    tiling, the two waves of a tile row -- hold y1 and y2 of the same units.  The epilogue writes
  //    out = g * relu(y1) + (1 - g) * x, g = sigmoid(y2), to C (ldc) and, if hw_x12 is set, the pre-activations
  //    [M, 2 * hw_C] = y1 | y2 (what the backward needs).  Biases hw_b1 / hw_b2 (task bias unused).
  //  hw_mode 2 (backward): the product (+ C when accumulate is set) is d(out) of the layer BELOW the one whose data
  //    gradient is being formed; the epilogue turns it into that layer's gate gradients at once:
  //    hw_d12 [M, 2 * hw_C] = d * g * (y1 > 0) | d * (relu(y1) - x) * g * (1 - g),  C = d * (1 - g)  (y1 | y2 = hw_x12).
  // ReLU gradient mask in the epilogue (single-task / chained launches, identity output rows): C = (product) where
  // relu_mask[row * ldc + col] > 0, else 0 -- the data gradient of a convolution that follows a conv + ReLU goes straight
  // through that ReLU's derivative (FFTBlock conv2 -> conv1, common_layers.py:178-180), no element-wise pass in between
  const float* relu_mask;
  // force_tile: 0 = the launcher picks the tile (and with it the kernel: 128x128 bf16-split or 64x64 f32) from the
  // launch's own size; 1 = 64, 2 = 128 -- a GEMM issued in row chunks then rounds exactly like the same GEMM issued
  // whole (ft_*_layer_fwd; ft_rows_tile_is_big gives the whole launch's choice)
  int force_tile;
  // Split-K of a single-task or chained NT launch with FEW output tiles and a LONG contraction (the token-side data
  // gradients: 4,096 rows): ksplit_slab != null offers ft_rows_ksplit_floats() floats of scratch; the launcher then may cut
  // the stage sequence into ksplit ranges (grid z), every range writing its raw partial tile to slab[z][M][N], and a second
  // launch adds the partials in range order and applies bias / accumulate / the output row map.  ksplit is set by the
  // launcher (0 / 1: not split).  Sums the same products in another order than the unsplit launch: callers are gradients.
  float* ksplit_slab;
  int ksplit;
  int hw_mode, hw_C;
  const float* hw_x;
  const float* hw_b1;
  const float* hw_b2;
  float* hw_x12;
  float* hw_d12;
};

// The highway epilogues (see FtGemmBatch) for an accumulator tile in the 32x32 MFMA layout shared by the row kernels:
// wave (wm, wn) of 2 x 2, lane (half, l31) holds column l31 and rows (e & 3) + 8 * (e >> 2) + 4 * half of each tile.
// lds: >= 2 * 32 * 33 floats of idle LDS (TN == 1 only), safe to overwrite when this is called.  Every thread of the
// workgroup must call it (TN == 1 synchronises).
template <int TM, int TN>
__device__ __forceinline__ void ft_highway_epilogue(const FtGemmBatch& bt, const FtGemmTask& T, float* TC,
                                                    f32x16 (&acc)[TM][TN], float* lds, int m0, int n0, int tid) {
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int half = lane >> 5, l31 = lane & 31;
  const int C = bt.hw_C, tM = T.M;
  const long ldc = T.ldc;
  const float* hx = bt.hw_x;
  if (bt.hw_mode == 1) {
    float* x12 = bt.hw_x12;
    if constexpr (TN == 2) {
      const int unit = ((n0 + wn * 64) >> 6) * 32 + l31;              // both 32-column tiles of this wave: the same units
      const bool uok = unit < C;
      const float b1 = uok ? bt.hw_b1[unit] : 0.f, b2 = uok ? bt.hw_b2[unit] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = m0 + wm * 32 * TM + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * half;
          if (row >= tM || !uok) continue;
          const float y1 = acc[i][0][e] + b1, y2 = acc[i][1][e] + b2;
          const float g = ft_sigmoid(y2);
          if (x12) {
            x12[(long)row * 2 * C + unit] = y1;
            x12[(long)row * 2 * C + C + unit] = y2;
          }
          TC[(long)row * ldc + unit] = g * fmaxf(y1, 0.f) + (1.f - g) * hx[(long)row * C + unit];
        }
    } else {
      static_assert(TN == 1 || TN == 2, "highway epilogue: 32- or 64-column waves");
      // 64-column tiles: wave wn = 0 holds y1, wave wn = 1 y2 of the same 32 units -> the gate travels through LDS
      const int unit = (n0 >> 6) * 32 + l31;
      const bool uok = unit < C;
      float* gl = lds + wm * (32 * TM) * 33;
      if (wn == 1) {
        const float b2 = uok ? bt.hw_b2[unit] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int rl = 32 * i + (e & 3) + 8 * (e >> 2) + 4 * half;
            const int row = m0 + wm * 32 * TM + rl;
            const float y2 = acc[i][0][e] + b2;
            gl[rl * 33 + l31] = ft_sigmoid(y2);
            if (x12 && row < tM && uok) x12[(long)row * 2 * C + C + unit] = y2;
          }
      }
      __syncthreads();
      if (wn == 0) {
        const float b1 = uok ? bt.hw_b1[unit] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int rl = 32 * i + (e & 3) + 8 * (e >> 2) + 4 * half;
            const int row = m0 + wm * 32 * TM + rl;
            if (row >= tM || !uok) continue;
            const float y1 = acc[i][0][e] + b1;
            const float g = gl[rl * 33 + l31];
            if (x12) x12[(long)row * 2 * C + unit] = y1;
            TC[(long)row * ldc + unit] = g * fmaxf(y1, 0.f) + (1.f - g) * hx[(long)row * C + unit];
          }
      }
    }
    return;
  }
  // mode 2
  const float* x12 = bt.hw_x12;
  float* d12 = bt.hw_d12;
  const bool eacc = T.accumulate != 0;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * 32 * TN + 32 * j + l31;
      if (col >= T.N) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 32 * TM + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * half;
        if (row >= tM) continue;
        float* cp = TC + (long)row * ldc + col;
        float dv = acc[i][j][e];
        if (eacc) dv += *cp;
        const float y1 = x12[(long)row * 2 * C + col], y2 = x12[(long)row * 2 * C + C + col];
        const float g = ft_sigmoid(y2);
        d12[(long)row * 2 * C + col] = y1 > 0.f ? dv * g : 0.f;
        d12[(long)row * 2 * C + C + col] = dv * (fmaxf(y1, 0.f) - hx[(long)row * C + col]) * g * (1.f - g);
        *cp = dv * (1.f - g);
      }
    }
}


// out_tap[m][n] = sum_r Amap(A)[r][m] * Bmap_tap(B)[r][n],  r over R logical rows   ("TN", split over rows)
// result written to dst[m*ldm + n*ldn + tap*ldj]  (deterministic slab + reduce)
struct FtGemmTNTask {
  const float* A;
  const float* B;
  float* dst;
  long lda, ldb;
  long ldm, ldn, ldj;
  int M, N, R, taps;
  FtRowMap amap, bmap;
  int a_vec, b_vec, accumulate;
  // rows of A / B are readable up to the next multiple of 4 of M / N (those columns only reach masked outputs)
  int rowpad;
  // strided batch (see FtGemmTask); dst += z0*sD0 + z1*sD1
  int nz, nz1;
  long sA0, sA1, sB0, sB1, sD0, sD1;
  // conv-bank mode (bankC > 0): A's M columns are the K = M / bankC members of a CBHG conv bank side by side, member
  // kk = m / bankC + 1 has kernel size kk: tap j < kk only, B row shift j - kk/2, A rows t < (kk odd ? bankTodd :
  // amap.Tvalid); the result goes to bank_dst[kk-1] in torch layout [bankC][N][kk].  bankC % tile == 0.
  int bankC, bankTodd;
  float* bank_dst[FT_MAX_TASKS];
};

// Workgroup -> (M-tile, tap, split) of a TN launch.  Plain tasks: blockIdx.x = M-tile, blockIdx.z = (instance*taps +
// tap)*S + split.  Conv-bank mode: member kk only has taps j < kk, so the grid enumerates the LIVE (member, tap, tile)
// triples along x (kk ascending, then tap, then the member's M-tiles) and blockIdx.z = split -- no empty workgroups,
// equal work per workgroup (the round-robin XCD assignment stays balanced), and neighbours share their dy tile.
struct FtTnWho {
  int mtile, zts, s, kk;      // zts = instance*taps + tap ; kk = member (bank mode) or 0
};
__device__ __forceinline__ FtTnWho ft_tn_who(const FtGemmTNTask& T, int S, int BM) {
  FtTnWho w;
  if (T.bankC > 0) {
    const int tpm = T.bankC / BM;                 // M-tiles per member
    const int q = blockIdx.x / tpm, sub = blockIdx.x - q * tpm;
    int kk = 1, first = 0;                        // first = kk(kk-1)/2 = index of (kk, tap 0)
    while (first + kk <= q) {
      first += kk;
      ++kk;
    }
    w.kk = kk;
    w.zts = q - first;
    w.mtile = (kk - 1) * tpm + sub;
    w.s = blockIdx.z;
  } else {
    w.kk = 0;
    w.mtile = blockIdx.x;
    w.zts = blockIdx.z / S;
    w.s = blockIdx.z - w.zts * S;
  }
  return w;
}

// the launcher's tile choice for a launch whose tasks have `tiles128` 128x128 output tiles in total
static inline bool ft_rows_tile_is_big(long tiles128, int maxM, int maxN) { return tiles128 >= 192 && maxN > 64 && maxM > 64; }
static_assert(sizeof(FtGemmBatch) <= 4080, "FtGemmBatch travels as a kernel argument (4 KB limit)");
int ft_launch_gemm_rows(FtGemmBatch* batch, int ntasks, bool b_ncontig, hipStream_t stream);
// scratch (floats) a launch of these tasks can use for split-K (0: it would not be split)
size_t ft_rows_ksplit_floats(const FtGemmBatch& batch, int ntasks, hipStream_t stream);
int ft_launch_ksplit_reduce(const float* slab, int S, const FtGemmTask& t, hipStream_t stream);
bool ft_rows_b3p_ok(const FtGemmBatch& batch, int ntask);
// fp32 on the bf16 matrix pipe (exact 3-way operand split, ft_gemm_b3.hip); NT + FAST launches, FT_GEMM_B3=0 disables
int ft_launch_gemm_rows_b3(const FtGemmBatch& batch, bool big, dim3 grid, hipStream_t stream);
int ft_launch_gemm_tn_b3(const FtGemmTNTask& t, float* slab, int S, int rows_per_split, int tm, dim3 grid,
                         hipStream_t stream);
// ft_capi_core.hip: workgroup slots (2 per usable CU) of a stream -- 512 unless it is CU-limited
void ft_note_stream_slots(hipStream_t s, int slots);
int ft_stream_slots(hipStream_t s);
bool ft_gemm_b3_enabled();
// 0 = fp32-exact (default), 1 = bf16 operands / fp32 accumulate on every NT-form fast launch (ft_set_gemm_precision)
int ft_gemm_precision();
int ft_launch_gemm_tn(const FtGemmTNTask& task, float* workspace, size_t workspace_floats,
                      hipStream_t stream);
size_t ft_gemm_tn_workspace_floats(const FtGemmTNTask& task);
// dst[m*ldm + n] = sum_{s<S} slab[s][m][n]  (fixed order)
int ft_launch_slab_sum(const float* slab, float* dst, int M, int N, int S, long ldm, hipStream_t stream);
