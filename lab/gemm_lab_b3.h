// lab: fp32 GEMM on the bf16 matrix pipe.  Every fp32 operand is split EXACTLY into three bf16 pieces
// (x = hi + mid + lo, 8 mantissa bits each, truncation splits so each remainder is exact in fp32) while it is staged
// into LDS; the product keeps the six terms down to 2^-16 relative (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi),
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  Dropped terms are <= 2^-23 of |a||b| -- the size of fp32's own
// product rounding.  bf16 MFMA runs at 16x the fp32 MFMA rate, six of them replace one: 2.67x fewer matrix cycles.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

// truncation split of 4 floats -> hi, mid, lo as 4 packed bf16 each
__device__ __forceinline__ void split3(const float4& v, u16x4& hi, u16x4& mid, u16x4& lo) {
  const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned b0 = __float_as_uint(f[i]);
    const unsigned h = b0 & 0xFFFF0000u;
    const float r1 = f[i] - __uint_as_float(h);
    const unsigned b1 = __float_as_uint(r1);
    const unsigned m = b1 & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(m);
    hi[i] = (unsigned short)(h >> 16);
    mid[i] = (unsigned short)(m >> 16);
    lo[i] = (unsigned short)(__float_as_uint(r2) >> 16);
  }
}

// LDS: per operand row, three planes of BK=32 bf16 (64 B each) + 16 B pad -> row stride 208 B = 104 shorts
template <int DUMMY>
__global__ __launch_bounds__(256) void k_b3(const float* __restrict__ A, const float* __restrict__ B,
                                            float* __restrict__ C, int M, int N, int K) {
  constexpr int TM = 2, TN = 2, BM = 128, BN = 128, RS = 104;          // RS: row stride in shorts
  constexpr int STAGE = (BM + BN) * RS;                                 // shorts
  constexpr int NBUF = DUMMY == 1 ? 1 : 2;        // 1: single LDS stage (53 KB -> two workgroups per CU)
  __shared__ __attribute__((aligned(16))) unsigned short smem[NBUF * STAGE];
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int tid = threadIdx.x, kq = tid & 7, rr = tid >> 3;
  const int nch = K / BK;
  float4 ra[4], rb[4];
  const float* Ag = A + (long)(m0 + rr) * K + 4 * kq;
  const float* Bg = B + (long)(n0 + rr) * K + 4 * kq;
  const long rs = 32L * K;
  auto load_stage = [&](int c) {
#pragma unroll
    for (int p = 0; p < 4; ++p) ra[p] = *reinterpret_cast<const float4*>(Ag + p * rs + c * BK);
#pragma unroll
    for (int p = 0; p < 4; ++p) rb[p] = *reinterpret_cast<const float4*>(Bg + p * rs + c * BK);
  };
  auto store_stage = [&](int buf) {
    unsigned short* As = smem + buf * STAGE;
    unsigned short* Bs = As + BM * RS;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      u16x4 h, m, l;
      split3(ra[p], h, m, l);
      unsigned short* row = As + (rr + 32 * p) * RS + 4 * kq;
      *reinterpret_cast<u16x4*>(row) = h;
      *reinterpret_cast<u16x4*>(row + 32) = m;
      *reinterpret_cast<u16x4*>(row + 64) = l;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      u16x4 h, m, l;
      split3(rb[p], h, m, l);
      unsigned short* row = Bs + (rr + 32 * p) * RS + 4 * kq;
      *reinterpret_cast<u16x4*>(row) = h;
      *reinterpret_cast<u16x4*>(row + 32) = m;
      *reinterpret_cast<u16x4*>(row + 64) = l;
    }
  };
  const int wave = tid >> 6, lane = tid & 63, wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
  f32x16 acc[TM][TN];
  for (int i = 0; i < TM; ++i)
    for (int j = 0; j < TN; ++j)
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  load_stage(0);
  store_stage(0);
  if (nch > 1) load_stage(1);
  __syncthreads();
  for (int c = 0; c < nch; ++c) {
    const int cur = NBUF == 2 ? (c & 1) : 0;
    if (NBUF == 2 && c + 1 < nch) {
      store_stage(cur ^ 1);
      if (c + 2 < nch) load_stage(c + 2);
    }
    const unsigned short* ap = smem + cur * STAGE + (wm * 64 + l31) * RS + 8 * half;
    const unsigned short* bp = smem + cur * STAGE + BM * RS + (wn * 64 + l31) * RS + 8 * half;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {                  // two k-steps of 16 per stage
      bf16x8 a[TM][3], b[TN][3];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          a[i][pl] = *reinterpret_cast<const bf16x8*>(ap + 32 * i * RS + 32 * pl + 16 * ks);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          b[j][pl] = *reinterpret_cast<const bf16x8*>(bp + 32 * j * RS + 32 * pl + 16 * ks);
      // small terms first
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    if (NBUF == 1 && c + 1 < nch) {
      store_stage(0);                       // registers were loaded one iteration ago
      if (c + 2 < nch) load_stage(c + 2);
      __syncthreads();
    }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * 64 + 32 * j + l31;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * half;
        C[(long)row * N + col] = acc[i][j][e];
      }
    }
}

// one LDS stage of 32 k, TWO register sets: the global loads of stage c+2 are in flight while stages c and c+1 are
// multiplied (with the 2.67x shorter MFMA phase one stage of prefetch no longer covers the L2/HBM latency)
template <int DUMMY>
__global__ __launch_bounds__(256, 2) void k_b3p(const float* __restrict__ A, const float* __restrict__ B,
                                                float* __restrict__ C, int M, int N, int K) {
  constexpr int TM = 2, TN = 2, BM = 128, BN = 128, RS = 104;
  __shared__ __attribute__((aligned(16))) unsigned short smem[(BM + BN) * RS];
  unsigned short* As = smem;
  unsigned short* Bs = smem + BM * RS;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int tid = threadIdx.x, kq = tid & 7, rr = tid >> 3;
  const int nch = K / BK;
  float4 ra[2][4], rb[2][4];
  const float* Ag = A + (long)(m0 + rr) * K + 4 * kq;
  const float* Bg = B + (long)(n0 + rr) * K + 4 * kq;
  const long rs = 32L * K;
#define LAB_LOAD(set, c)                                                                                 \
  {                                                                                                      \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) ra[set][p] = *reinterpret_cast<const float4*>(Ag + p * rs + (c) * BK); \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) rb[set][p] = *reinterpret_cast<const float4*>(Bg + p * rs + (c) * BK); \
  }
#define LAB_STORE(set)                                                                                   \
  {                                                                                                      \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                                      \
      u16x4 h, m, l;                                                                                     \
      split3(ra[set][p], h, m, l);                                                                       \
      unsigned short* row = As + (rr + 32 * p) * RS + 4 * kq;                                            \
      *reinterpret_cast<u16x4*>(row) = h;                                                                \
      *reinterpret_cast<u16x4*>(row + 32) = m;                                                           \
      *reinterpret_cast<u16x4*>(row + 64) = l;                                                           \
    }                                                                                                    \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                                      \
      u16x4 h, m, l;                                                                                     \
      split3(rb[set][p], h, m, l);                                                                       \
      unsigned short* row = Bs + (rr + 32 * p) * RS + 4 * kq;                                            \
      *reinterpret_cast<u16x4*>(row) = h;                                                                \
      *reinterpret_cast<u16x4*>(row + 32) = m;                                                           \
      *reinterpret_cast<u16x4*>(row + 64) = l;                                                           \
    }                                                                                                    \
  }
  const int wave = tid >> 6, lane = tid & 63, wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
  f32x16 acc[TM][TN];
  for (int i = 0; i < TM; ++i)
    for (int j = 0; j < TN; ++j)
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const unsigned short* ap = As + (wm * 64 + l31) * RS + 8 * half;
  const unsigned short* bp = Bs + (wn * 64 + l31) * RS + 8 * half;
  auto compute = [&]() {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[TM][3], b[TN][3];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) a[i][pl] = *reinterpret_cast<const bf16x8*>(ap + 32 * i * RS + 32 * pl + 16 * ks);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) b[j][pl] = *reinterpret_cast<const bf16x8*>(bp + 32 * j * RS + 32 * pl + 16 * ks);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
        }
    }
  };
  // stage c lives in register set c & 1
  LAB_LOAD(0, 0)
  if (nch > 1) LAB_LOAD(1, 1)
  LAB_STORE(0)
  if (nch > 2) LAB_LOAD(0, 2)
  __syncthreads();
  for (int c = 0; c < nch; c += 2) {
    compute();                                  // stage c
    __syncthreads();
    if (c + 1 < nch) {
      LAB_STORE(1)                              // stage c+1
      if (c + 3 < nch) LAB_LOAD(1, c + 3)
      __syncthreads();
      compute();                                // stage c+1
      __syncthreads();
      if (c + 2 < nch) {
        LAB_STORE(0)                            // stage c+2
        if (c + 4 < nch) LAB_LOAD(0, c + 4)
        __syncthreads();
      }
    }
  }
#undef LAB_LOAD
#undef LAB_STORE
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * 64 + 32 * j + l31;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * half;
        C[(long)row * N + col] = acc[i][j][e];
      }
    }
}

// split in the MFMA shadow: the 8 staged float4 of stage c+1 are split between the MFMA groups of stage c (one float4
// per group of 6 MFMAs), so after the barrier only the 24 ds_write_b64 remain
template <int DUMMY>
__global__ __launch_bounds__(256, 2) void k_b3i(const float* __restrict__ A, const float* __restrict__ B,
                                                float* __restrict__ C, int M, int N, int K) {
  constexpr int TM = 2, TN = 2, BM = 128, BN = 128, RS = 104;
  __shared__ __attribute__((aligned(16))) unsigned short smem[(BM + BN) * RS];
  unsigned short* As = smem;
  unsigned short* Bs = smem + BM * RS;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int tid = threadIdx.x, kq = tid & 7, rr = tid >> 3;
  const int nch = K / BK;
  float4 rv[8];                                  // 0..3: A rows rr+32p, 4..7: B rows
  u16x4 ph[8], pm[8], pl[8];
  const float* Ag = A + (long)(m0 + rr) * K + 4 * kq;
  const float* Bg = B + (long)(n0 + rr) * K + 4 * kq;
  const long rs = 32L * K;
  auto load_stage = [&](int c) {
#pragma unroll
    for (int p = 0; p < 4; ++p) rv[p] = *reinterpret_cast<const float4*>(Ag + p * rs + c * BK);
#pragma unroll
    for (int p = 0; p < 4; ++p) rv[4 + p] = *reinterpret_cast<const float4*>(Bg + p * rs + c * BK);
  };
  auto write_planes = [&]() {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      unsigned short* row = (p < 4 ? As : Bs) + (rr + 32 * (p & 3)) * RS + 4 * kq;
      *reinterpret_cast<u16x4*>(row) = ph[p];
      *reinterpret_cast<u16x4*>(row + 32) = pm[p];
      *reinterpret_cast<u16x4*>(row + 64) = pl[p];
    }
  };
  const int wave = tid >> 6, lane = tid & 63, wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
  f32x16 acc[TM][TN];
  for (int i = 0; i < TM; ++i)
    for (int j = 0; j < TN; ++j)
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const unsigned short* ap = As + (wm * 64 + l31) * RS + 8 * half;
  const unsigned short* bp = Bs + (wn * 64 + l31) * RS + 8 * half;
  load_stage(0);
#pragma unroll
  for (int p = 0; p < 8; ++p) split3(rv[p], ph[p], pm[p], pl[p]);
  write_planes();
  if (nch > 1) load_stage(1);
  __syncthreads();
  for (int c = 0; c < nch; ++c) {
    const bool more = c + 1 < nch;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[TM][3], b[TN][3];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int pl_ = 0; pl_ < 3; ++pl_) a[i][pl_] = *reinterpret_cast<const bf16x8*>(ap + 32 * i * RS + 32 * pl_ + 16 * ks);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int pl_ = 0; pl_ < 3; ++pl_) b[j][pl_] = *reinterpret_cast<const bf16x8*>(bp + 32 * j * RS + 32 * pl_ + 16 * ks);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
          const int slot = ks * 4 + i * 2 + j;
          if (more) split3(rv[slot], ph[slot], pm[slot], pl[slot]);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    if (more) {
      write_planes();
      if (c + 2 < nch) load_stage(c + 2);
      __syncthreads();
    }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * 64 + 32 * j + l31;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * half;
        C[(long)row * N + col] = acc[i][j][e];
      }
    }
}

// double-buffered, 16 k per stage: LDS row = 3 planes x 16 bf16 (+16 B pad) = 112 B -> 2 x 28 KB per workgroup
template <int DUMMY>
__global__ __launch_bounds__(256, 2) void k_b3d(const float* __restrict__ A, const float* __restrict__ B,
                                                float* __restrict__ C, int M, int N, int K) {
  constexpr int TM = 2, TN = 2, BM = 128, BN = 128, RS = 56, KS = 16;
  constexpr int STAGE = (BM + BN) * RS;
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * STAGE];
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int tid = threadIdx.x, kq = tid & 3, rr = tid >> 2;
  const int nch = K / KS;
  float4 ra[2], rb[2];
  const float* Ag = A + (long)(m0 + rr) * K + 4 * kq;
  const float* Bg = B + (long)(n0 + rr) * K + 4 * kq;
  const long rs = 64L * K;
  auto load_stage = [&](int c) {
#pragma unroll
    for (int p = 0; p < 2; ++p) ra[p] = *reinterpret_cast<const float4*>(Ag + p * rs + c * KS);
#pragma unroll
    for (int p = 0; p < 2; ++p) rb[p] = *reinterpret_cast<const float4*>(Bg + p * rs + c * KS);
  };
  auto store_stage = [&](int buf) {
    unsigned short* As = smem + buf * STAGE;
    unsigned short* Bs = As + BM * RS;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      u16x4 h, m, l;
      split3(ra[p], h, m, l);
      unsigned short* row = As + (rr + 64 * p) * RS + 4 * kq;
      *reinterpret_cast<u16x4*>(row) = h;
      *reinterpret_cast<u16x4*>(row + 16) = m;
      *reinterpret_cast<u16x4*>(row + 32) = l;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      u16x4 h, m, l;
      split3(rb[p], h, m, l);
      unsigned short* row = Bs + (rr + 64 * p) * RS + 4 * kq;
      *reinterpret_cast<u16x4*>(row) = h;
      *reinterpret_cast<u16x4*>(row + 16) = m;
      *reinterpret_cast<u16x4*>(row + 32) = l;
    }
  };
  const int wave = tid >> 6, lane = tid & 63, wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
  f32x16 acc[TM][TN];
  for (int i = 0; i < TM; ++i)
    for (int j = 0; j < TN; ++j)
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  load_stage(0);
  store_stage(0);
  if (nch > 1) load_stage(1);
  __syncthreads();
  for (int c = 0; c < nch; ++c) {
    const int cur = c & 1;
    const unsigned short* ap = smem + cur * STAGE + (wm * 64 + l31) * RS + 8 * half;
    const unsigned short* bp = smem + cur * STAGE + BM * RS + (wn * 64 + l31) * RS + 8 * half;
    bf16x8 a[TM][3], b[TN][3];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) a[i][pl] = *reinterpret_cast<const bf16x8*>(ap + 32 * i * RS + 16 * pl);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) b[j][pl] = *reinterpret_cast<const bf16x8*>(bp + 32 * j * RS + 16 * pl);
    if (c + 1 < nch) {
      store_stage(cur ^ 1);
      if (c + 2 < nch) load_stage(c + 2);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
      }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * 64 + 32 * j + l31;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * half;
        C[(long)row * N + col] = acc[i][j][e];
      }
    }
}

static void lab_b3(float* A, float* B, float* C, int M, int N, int K, double fl) {
  // full-mantissa pseudo-random operands in [-1, 1)
  std::vector<float> ha((size_t)M * K), hb((size_t)N * K);
  unsigned long long s = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (float)((double)(long long)(s >> 11) / (double)(1ll << 52) - 1.0);
  };
  for (auto& v : ha) v = rnd();
  for (auto& v : hb) v = rnd();
  (void)hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  dim3 grid(M / 128, N / 128);
  std::vector<float> c32((size_t)M * N), c3((size_t)M * N);
  hipLaunchKernelGGL(k_base<0>, grid, dim3(256), 0, 0, A, B, C, M, N, K);
  (void)hipMemcpy(c32.data(), C, c32.size() * 4, hipMemcpyDeviceToHost);
  (void)hipMemset(C, 0, (size_t)M * N * 4);
  printf("bf16x3 split    : %.1f TF (fp32-equivalent)\n",
         tf([&] { hipLaunchKernelGGL(k_b3<0>, grid, dim3(256), 0, 0, A, B, C, M, N, K); }, fl));
  printf("bf16x3, 2 stages of 16 k     : %.1f TF\n",
         tf([&] { hipLaunchKernelGGL(k_b3d<0>, grid, dim3(256), 0, 0, A, B, C, M, N, K); }, fl));
  printf("bf16x3, split in the MFMA shadow: %.1f TF\n",
         tf([&] { hipLaunchKernelGGL(k_b3i<0>, grid, dim3(256), 0, 0, A, B, C, M, N, K); }, fl));
  printf("bf16x3, 1 LDS stage, 2-deep register prefetch: %.1f TF\n",
         tf([&] { hipLaunchKernelGGL(k_b3p<0>, grid, dim3(256), 0, 0, A, B, C, M, N, K); }, fl));
  {
    std::vector<float> cp((size_t)M * N);
    (void)hipMemcpy(cp.data(), C, cp.size() * 4, hipMemcpyDeviceToHost);
    double md = 0;
    for (size_t i = 0; i < cp.size(); i += 101) md = fmax(md, fabs(cp[i] - c32[i]));
    printf("  (max |diff| vs the f32 MFMA result, sampled: %.3e)\n", md);
  }
  printf("bf16x3, 1 LDS stage (2 WG/CU): %.1f TF\n",
         tf([&] { hipLaunchKernelGGL(k_b3<1>, grid, dim3(256), 0, 0, A, B, C, M, N, K); }, fl));
  (void)hipMemcpy(c3.data(), C, c3.size() * 4, hipMemcpyDeviceToHost);
  // accuracy of both against a double-precision reference on sampled entries
  double e32 = 0, e3 = 0, scale = 0;
  for (int t = 0; t < 4000; ++t) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    const size_t r = (s >> 20) % M, cc = (s >> 44) % N;
    double ref = 0;
    for (int k = 0; k < K; ++k) ref += (double)ha[r * K + k] * (double)hb[cc * K + k];
    e32 = fmax(e32, fabs(c32[r * N + cc] - ref));
    e3 = fmax(e3, fabs(c3[r * N + cc] - ref));
    scale = fmax(scale, fabs(ref));
  }
  printf("  max |err| vs f64 over 4000 samples (|C| up to %.2f): f32 MFMA %.3e   bf16x3 %.3e\n", scale, e32, e3);
}
