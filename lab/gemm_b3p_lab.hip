// Ablation lab for the pipelined bf16x3 NT GEMM (ft_gemm_rows_b3p_kernel's structure on a plain M,N,K problem):
// which part of the iteration holds the MFMA pipe at ~40 % busy?   hipcc -O3 --offload-arch=gfx950 lab/gemm_b3p_lab.hip -o lab/gemm_b3p_lab.bin
// VAR bit 0: no global loads in the loop   bit 1: no split (raw bits stored)   bit 2: no ds_write   bit 3: no ds_read
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned rn_pack(float a, float b) {
  const bf16x2v p = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = rn_pack(a, b);
  const float a1 = a - __uint_as_float(hi << 16), b1 = b - __uint_as_float(hi & 0xFFFF0000u);
  mid = rn_pack(a1, b1);
  const float a2 = a1 - __uint_as_float(mid << 16), b2 = b1 - __uint_as_float(mid & 0xFFFF0000u);
  lo = rn_pack(a2, b2);
}

// TM x TN 32x32 accumulators per wave, WM x WN waves per workgroup
template <int VAR, int TM, int TN, int WM, int WN, int OCC, int REMAP = 0, int FPF = 0>
__global__ __launch_bounds__(64 * WM * WN, OCC) void k_b3p(const float* __restrict__ A, const float* __restrict__ B,
                                                            float* __restrict__ C, int M, int N, int K) {
  constexpr int NT = 64 * WM * WN, BM = 32 * TM * WM, BN = 32 * TN * WN, SK = 16, RW = 48, BUF = (BM + BN) * RW;
  constexpr int RPT = (BM + BN) * 2 / NT;          // (row, k-half) pieces per thread per stage
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * BUF];
  int bx = blockIdx.x, by = blockIdx.y;
  if constexpr (REMAP) {            // XCD c (= linear id % 8) walks a contiguous range of tiles, column tile fastest
    const int nwg = gridDim.x * gridDim.y, L = blockIdx.x + gridDim.x * blockIdx.y;
    const int xcd = L & 7, q = nwg >> 3, r = nwg & 7;
    const int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
    by = w % (int)gridDim.y;
    bx = w / (int)gridDim.y;
  }
  const int m0 = bx * BM, n0 = by * BN, tid = threadIdx.x;
  const int h = tid & 1;
  const int nch = K / SK;
  struct Regs { u32x4 v[RPT][2]; };
  unsigned voff[RPT];
  const float* base[RPT];
#pragma unroll
  for (int p = 0; p < RPT; ++p) {
    const int r = (tid >> 1) + p * (NT / 2);       // 0 .. BM+BN-1
    base[p] = r < BM ? A + (long)(((VAR & 16) ? 0 : m0) + r) * K : B + (long)(n0 + r - BM) * K;
    voff[p] = 16u * h;
  }
  auto load_stage = [&](Regs& R, int c) {
    if constexpr (VAR & 1) return;
#pragma unroll
    for (int p = 0; p < RPT; ++p) {
      const float* q = base[p] + c * SK + 4 * h;
      R.v[p][0] = *reinterpret_cast<const u32x4*>(q);
      R.v[p][1] = *reinterpret_cast<const u32x4*>(q + 8);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto store_stage = [&](const Regs& R, unsigned short* buf) {
#pragma unroll
    for (int p = 0; p < RPT; ++p) {
      const int r = (tid >> 1) + p * (NT / 2);
      unsigned short* row = buf + r * RW + 8 * h;
      const float4 u = __builtin_bit_cast(float4, R.v[p][0]), v = __builtin_bit_cast(float4, R.v[p][1]);
      unsigned hi[4], mid[4], lo[4];
      if constexpr (VAR & 2) {
        for (int e = 0; e < 4; ++e) { hi[e] = R.v[p][0][e]; mid[e] = R.v[p][1][e]; lo[e] = R.v[p][0][e] ^ R.v[p][1][e]; }
      } else {
        split_pair(u.x, u.y, hi[0], mid[0], lo[0]);
        split_pair(u.z, u.w, hi[1], mid[1], lo[1]);
        split_pair(v.x, v.y, hi[2], mid[2], lo[2]);
        split_pair(v.z, v.w, hi[3], mid[3], lo[3]);
      }
      if constexpr (VAR & 4) {
        asm volatile("" ::"v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]), "v"(mid[0]), "v"(mid[1]), "v"(mid[2]), "v"(mid[3]),
                     "v"(lo[0]), "v"(lo[1]), "v"(lo[2]), "v"(lo[3]));
      } else {
        *reinterpret_cast<uint4*>(row) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        *reinterpret_cast<uint4*>(row + 16) = make_uint4(mid[0], mid[1], mid[2], mid[3]);
        *reinterpret_cast<uint4*>(row + 32) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
      }
    }
  };
  const int wave = tid >> 6, lane = tid & 63, wm = wave / WN, wn = wave % WN, half = lane >> 5, l31 = lane & 31;
  f32x16 acc[TM][TN];
  for (int i = 0; i < TM; ++i)
    for (int j = 0; j < TN; ++j)
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int aoff = (wm * 32 * TM + l31) * RW + 8 * half;
  const int boff = (BM + wn * 32 * TN + l31) * RW + 8 * half;
  bf16x8 fa[TM][3], fb[TN][3], ga[TM][3], gb[TN][3];
  if constexpr (VAR & 8) {
    for (int i = 0; i < TM; ++i) for (int pl = 0; pl < 3; ++pl) for (int e = 0; e < 8; ++e) fa[i][pl][e] = (__bf16)(float)(lane + e + pl);
    for (int j = 0; j < TN; ++j) for (int pl = 0; pl < 3; ++pl) for (int e = 0; e < 8; ++e) fb[j][pl][e] = (__bf16)(float)(lane - e + pl);
  }
  auto read_into = [&](bf16x8 (&xa)[TM][3], bf16x8 (&xb)[TN][3], const unsigned short* buf) {
    if constexpr (VAR & 8) return;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) xa[i][pl] = *reinterpret_cast<const bf16x8*>(buf + aoff + 32 * i * RW + 16 * pl);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) xb[j][pl] = *reinterpret_cast<const bf16x8*>(buf + boff + 32 * j * RW + 16 * pl);
  };
  auto read_frags = [&](const unsigned short* buf) { read_into(fa, fb, buf); };
  auto mfma_on = [&](bf16x8 (&xa)[TM][3], bf16x8 (&xb)[TN][3]) {
    constexpr int PA_[6] = {2, 0, 1, 1, 0, 0}, PB_[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int term = 0; term < 6; ++term)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[i][PA_[term]], xb[j][PB_[term]], acc[i][j], 0, 0, 0);
  };
  auto mfma_frags = [&]() { mfma_on(fa, fb); };
  auto interleave = [&]() {
    if constexpr ((VAR & 8) == 0) __builtin_amdgcn_sched_group_barrier(0x100, 3 * (TM + TN), 0);
#pragma unroll
    for (int i = 0; i < TM * TN * 6; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, (RPT * 44 + TM * TN * 6 - 1) / (TM * TN * 6), 0);
    }
  };
  unsigned short* const buf0 = smem;
  unsigned short* const buf1 = smem + BUF;
  Regs R0, R1;
  if constexpr (VAR & 1) {
    for (int p = 0; p < RPT; ++p) for (int q = 0; q < 2; ++q) for (int e = 0; e < 4; ++e) { R0.v[p][q][e] = tid * 977 + e; R1.v[p][q][e] = tid * 31 + e; }
  }
  load_stage(R0, 0);
  load_stage(R1, 1);
  store_stage(R0, buf0);
  __syncthreads();
  int c = 0;
  if constexpr (FPF) {
    // iteration c: MFMA(stage c) from the fragment set read one iteration earlier; reads stage c+1 (buf[(c+1)&1]) for the
    // next iteration; splits + writes stage c+2 into buf[c&1]; requests stage c+3
    load_stage(R0, 2);                 // R0's stage 0 is already in buf0: R0 <- stage 2, R1 = stage 1
    read_into(fa, fb, buf0);           // stage 0
    store_stage(R1, buf1);             // stage 1
    __syncthreads();
    for (; c + 5 < nch; c += 2) {
      load_stage(R1, c + 3);
      read_into(ga, gb, buf1);         // stage c+1
      store_stage(R0, buf0);           // stage c+2
      mfma_on(fa, fb);                 // stage c
      interleave();
      __syncthreads();
      load_stage(R0, c + 4);
      read_into(fa, fb, buf0);         // stage c+2
      store_stage(R1, buf1);           // stage c+3
      mfma_on(ga, gb);                 // stage c+1
      interleave();
      __syncthreads();
    }
  } else {
  for (; c + 3 < nch; c += 2) {
    load_stage(R0, c + 2);
    read_frags(buf0);
    store_stage(R1, buf1);
    mfma_frags();
    interleave();
    __syncthreads();
    load_stage(R1, c + 3);
    read_frags(buf1);
    store_stage(R0, buf0);
    mfma_frags();
    interleave();
    __syncthreads();
  }
  }
  // (tail stages skipped in the lab: timing only; K/16 even and >= 4)
  for (int i = 0; i < TM; ++i)
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * 32 * TN + 32 * j + l31;
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 32 * TM + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * half;
        C[(long)row * N + col] = acc[i][j][e];
      }
    }
}

// ---- variant 2: full-line loads (8 lanes x 16 B per row), a loaded float4 feeds TWO stages: (x,y) the even one, (z,w)
// the odd one; fragment prefetch; XCD remap
template <int VAR, int OCC>
__global__ __launch_bounds__(256, OCC) void k_v2(const float* __restrict__ A, const float* __restrict__ B,
                                                 float* __restrict__ C, int M, int N, int K) {
  constexpr int TM = 2, TN = 2, WN = 2, NT = 256, BM = 128, BN = 128, RW = 48, BUF = (BM + BN) * RW;
  constexpr int RPT = (BM + BN) * 8 / NT;          // float4 per thread per double stage (32 k)
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * BUF];
  int bx = blockIdx.x, by = blockIdx.y;
  {
    const int nwg = gridDim.x * gridDim.y, L = blockIdx.x + gridDim.x * blockIdx.y;
    const int xcd = L & 7, q = nwg >> 3, r = nwg & 7;
    const int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
    by = w % (int)gridDim.y;
    bx = w / (int)gridDim.y;
  }
  const int m0 = bx * BM, n0 = by * BN, tid = threadIdx.x;
  const int q8 = tid & 7;
  const int nds = K / 32;
  struct Regs { u32x4 v[RPT]; };
  // rows (tid >> 3) + 32 p: p < RPT/2 are A rows, the rest B rows (BM = BN = 128, 32 rows per pass)
  const unsigned voff0 = (unsigned)(((tid >> 3) * K + 4 * q8) * 4);
  const unsigned pstep = (unsigned)(32 * K * 4);
  auto load_half = [&](Regs& R, int d, int hf) {    // half of the double stage's loads: hf 0 = A rows, 1 = B rows
    if constexpr (VAR & 1) return;
    const float* bp = (hf == 0 ? A + (long)m0 * K : B + (long)n0 * K) + d * 32;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bp), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int p = 0; p < RPT / 2; ++p)
      R.v[hf * (RPT / 2) + p] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff0 + p * pstep, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto store_stage = [&](const Regs& R, unsigned short* buf, int odd) {
#pragma unroll
    for (int p = 0; p < RPT; ++p) {
      const int r = (tid >> 3) + p * (NT / 8);
      unsigned* row = reinterpret_cast<unsigned*>(buf + r * RW) + q8;
      const float a = __uint_as_float(R.v[p][2 * odd]), b = __uint_as_float(R.v[p][2 * odd + 1]);
      unsigned hi, mid, lo;
      if constexpr (VAR & 2) { hi = R.v[p][2 * odd]; mid = R.v[p][2 * odd + 1]; lo = hi ^ mid; }
      else split_pair(a, b, hi, mid, lo);
      row[0] = hi;
      row[8] = mid;
      row[16] = lo;
    }
  };
  const int wave = tid >> 6, lane = tid & 63, wm = wave / WN, wn = wave % WN, half = lane >> 5, l31 = lane & 31;
  f32x16 acc[TM][TN];
  for (int i = 0; i < TM; ++i)
    for (int j = 0; j < TN; ++j)
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int aoff = (wm * 32 * TM + l31) * RW + 8 * half;
  const int boff = (BM + wn * 32 * TN + l31) * RW + 8 * half;
  bf16x8 fa[TM][3], fb[TN][3], ga[TM][3], gb[TN][3];
  auto read_into = [&](bf16x8 (&xa)[TM][3], bf16x8 (&xb)[TN][3], const unsigned short* buf) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) xa[i][pl] = *reinterpret_cast<const bf16x8*>(buf + aoff + 32 * i * RW + 16 * pl);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) xb[j][pl] = *reinterpret_cast<const bf16x8*>(buf + boff + 32 * j * RW + 16 * pl);
  };
  auto mfma_on = [&](bf16x8 (&xa)[TM][3], bf16x8 (&xb)[TN][3]) {
    constexpr int PA_[6] = {2, 0, 1, 1, 0, 0}, PB_[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int term = 0; term < 6; ++term)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[i][PA_[term]], xb[j][PB_[term]], acc[i][j], 0, 0, 0);
  };
  auto interleave = [&]() {
    __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
    for (int i = 0; i < 24; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
  };
  unsigned short* const buf0 = smem;
  unsigned short* const buf1 = smem + BUF;
  Regs R0, R1;
  if constexpr (VAR & 1) {
    for (int p = 0; p < RPT; ++p) for (int e = 0; e < 4; ++e) { R0.v[p][e] = tid * 977 + e + p; R1.v[p][e] = tid * 31 + e + p; }
  }
  // double stage D = stages 2D (even, (x,y) halves) and 2D+1 (odd).  iteration c: MFMA(c) from the fragment set read at c-1,
  // read stage c+1, write stage c+2, and the loads of double stage (c+2)/2 + 1 are requested half at c even, half at c odd
  load_half(R0, 0, 0);
  load_half(R0, 0, 1);
  load_half(R1, 1, 0);
  load_half(R1, 1, 1);
  store_stage(R0, buf0, 0);            // stage 0
  __syncthreads();
  read_into(fa, fb, buf0);
  store_stage(R0, buf1, 1);            // stage 1
  __syncthreads();
  // now: frags(0) in fa/fb, stage 1 in buf1, R0 free, R1 = double stage 1 (stages 2, 3)
  for (int d = 0; d + 3 < nds; d += 2) {
    // stages 2d, 2d+1 (R1 holds double stage d+1; R0 receives d+2)
    load_half(R0, d + 2, 0);
    read_into(ga, gb, buf1);           // stage 2d+1
    store_stage(R1, buf0, 0);          // stage 2d+2
    mfma_on(fa, fb);                   // stage 2d
    interleave();
    __syncthreads();
    load_half(R0, d + 2, 1);
    read_into(fa, fb, buf0);           // stage 2d+2
    store_stage(R1, buf1, 1);          // stage 2d+3
    mfma_on(ga, gb);                   // stage 2d+1
    interleave();
    __syncthreads();
    // stages 2d+2, 2d+3 (R0 holds double stage d+2; R1 receives d+3)
    load_half(R1, d + 3, 0);
    read_into(ga, gb, buf1);           // stage 2d+3
    store_stage(R0, buf0, 0);          // stage 2d+4
    mfma_on(fa, fb);                   // stage 2d+2
    interleave();
    __syncthreads();
    load_half(R1, d + 3, 1);
    read_into(fa, fb, buf0);           // stage 2d+4
    store_stage(R0, buf1, 1);          // stage 2d+5
    mfma_on(ga, gb);                   // stage 2d+3
    interleave();
    __syncthreads();
  }
  for (int i = 0; i < TM; ++i)
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * 32 * TN + 32 * j + l31;
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 32 * TM + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * half;
        C[(long)row * N + col] = acc[i][j][e];
      }
    }
}

template <int VAR, int OCC>
void run2(const char* name, const float* A, const float* B, float* C, int M, int N, int K) {
  dim3 grid(M / 128, N / 128), block(256);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_v2<VAR, OCC>), grid, block, 0, 0, A, B, C, M, N, K);
  (void)hipEventRecord(e0);
  const int n = 10;
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL((k_v2<VAR, OCC>), grid, block, 0, 0, A, B, C, M, N, K);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= n;
  printf("%-46s tile 128x128 waves 4 occ %d: %8.1f us  %6.1f TF(f32-eq)  %6.0f TF(bf16)\n", name, OCC, ms * 1e3,
         2.0 * M * N * K / ms / 1e9, 12.0 * M * N * K / ms / 1e9);
  fflush(stdout);
}

template <int VAR, int TM, int TN, int WM, int WN, int OCC, int REMAP = 0, int FPF = 0>
void run(const char* name, const float* A, const float* B, float* C, int M, int N, int K) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  dim3 grid(M / BM, N / BN), block(64 * WM * WN);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_b3p<VAR, TM, TN, WM, WN, OCC, REMAP, FPF>), grid, block, 0, 0, A, B, C, M, N, K);
  hipEventRecord(e0);
  const int n = 10;
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL((k_b3p<VAR, TM, TN, WM, WN, OCC, REMAP, FPF>), grid, block, 0, 0, A, B, C, M, N, K);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= n;
  printf("%-46s tile %3dx%3d waves %d occ %d: %8.1f us  %6.1f TF(f32-eq)  %6.0f TF(bf16)\n", name, BM, BN, WM * WN, OCC, ms * 1e3,
         2.0 * M * N * K / ms / 1e9, 12.0 * M * N * K / ms / 1e9);
  fflush(stdout);
}

int main() {
  const int M = 27136 /* 212 x 128 = 106 x 256 */, N = 2048, K = 1024;
  float *A, *B, *C;
  hipMalloc(&A, (size_t)M * K * 4);
  hipMalloc(&B, (size_t)N * K * 4);
  hipMalloc(&C, (size_t)M * N * 4);
  std::vector<float> h((size_t)M * K);
  srand(1);
  for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
  hipMemcpy(B, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
  // (the first launch series runs ~15 % slow -- clocks / caches warming up: it is repeated)
  run<0, 2, 2, 2, 2, 2>("two-buffer pipeline, no remap (warm-up)", A, B, C, M, N, K);
  run<0, 2, 2, 2, 2, 2>("two-buffer pipeline, no remap", A, B, C, M, N, K);
  run<0, 2, 2, 2, 2, 2, 1>("+ XCD-contiguous tile walk", A, B, C, M, N, K);
  run<0, 2, 2, 2, 2, 2, 0, 1>("+ fragment prefetch (no remap)", A, B, C, M, N, K);
  run<0, 2, 2, 2, 2, 2, 1, 1>("+ fragment prefetch + remap", A, B, C, M, N, K);
  run2<0, 2>("v2: full-line loads, pair stages, fpf, remap", A, B, C, M, N, K);
  run<16, 2, 2, 2, 2, 2, 1, 1>("fpf + remap, every WG reads A tile 0 (L2-hot A)", A, B, C, M, N, K);
  run<0, 2, 4, 2, 2, 1, 1, 1>("wave tile 64x128 (128x256), 4 waves, 1 wg/CU", A, B, C, M, N, K);
  run<0, 4, 2, 2, 2, 1, 1, 1>("wave tile 128x64 (256x128), 4 waves, 1 wg/CU", A, B, C, M, N, K);
  // ablations: registers hold synthetic, loop-invariant data in the "no loads" arms -- the split is hoisted with them and
  // the MFMAs chew on regular bit patterns at a higher clock: upper bounds of the structure, not prices of the loads
  run2<2, 2>("v2 without the split (raw bits stored)", A, B, C, M, N, K);
  run2<1, 2>("v2 without global loads (and without split: hoisted)", A, B, C, M, N, K);
  run<15, 2, 2, 2, 2, 2>("MFMAs + barriers only", A, B, C, M, N, K);
  return 0;
}
