// Fused self-attention for the FastPitch variant in its bf16 matmul mode (BASELINE configs[2]):
//   nn.MultiheadAttention(d, heads, dropout)(x, x, x, key_padding_mask)   (models/common_layers.py:148-185, :172-174)
// between the in-projection and the out-projection: QK^T -> key-padding-masked softmax -> attention dropout -> PV in ONE
// kernel, flash style -- the [B, heads, T, T] score / probability tensors (181 MB per frame-side layer at the benchmark
// shape) never exist in HBM; the backward recomputes the probabilities from Q, K and the forward's per-row log-sum-exp.
// Operands are the fp32 rows of the interleaved projection buffer qkv [B, T, 3d] (q | k | v, head h at columns h*hd),
// rounded to bf16 while they are staged (what every bf16-mode GEMM of the library does); products on
// v_mfma_f32_32x32x16_bf16, softmax statistics, accumulation and outputs in fp32.
//
// Orientation (cdna_hip_programming.md, "An accumulator tile as the next MFMA's operand"): the score tile is computed
// TRANSPOSED, X = K Q^T [32 keys x 32 queries], so that the query sits on the lane and a lane's 16 registers are keys:
//   * the softmax reductions over the keys are in-lane (plus one exchange between the two lane halves);
//   * P (bf16) is, as it stands, the B operand of O^T += V^T P -- the key order inside a 16-key step is permuted
//     (element j of lane half h = key 16 s + 8 (j >> 2) + 4 h + (j & 3)), so the V^T fragments are gathered in that same
//     order, with the transposing LDS read ds_read_b64_tr_b16 from the row-major V tile.
// Workgroup = 4 waves = 128 queries of one (batch item, head); a wave owns 32 queries; keys in blocks of 64; K / V tiles
// row-major bf16 in LDS (row stride HD*2 + 16 bytes), next block's global loads in flight during the current block's
// MFMAs.  Attention dropout re-derives the counter-based mask of ft_softmax_fwd (index = flat index into [B,h,T,T]).
//
// Backward: ft_attn_bwd_dq_kernel (query on the lane, like the forward: dQ^T += K^T dS) and ft_attn_bwd_dkv_kernel (key
// on the lane: dV^T += dO^T P, dK^T += Q^T dS), each recomputing S and dP = dO V^T; no atomics, nothing summed across
// workgroups: bitwise reproducible.
#include "ft_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int KB = 64;                       // keys per block
constexpr float LOG2E = 1.44269504088896341f;

__device__ __forceinline__ int crow(int r, int hf) { return (r & 3) + 8 * (r >> 2) + 4 * hf; }

__device__ __forceinline__ bf16x8 cvt8(const float4& a, const float4& b) {
  return bf16x8{(__bf16)a.x, (__bf16)a.y, (__bf16)a.z, (__bf16)a.w, (__bf16)b.x, (__bf16)b.y, (__bf16)b.z, (__bf16)b.w};
}

// 8 consecutive floats of a row (zeros if !ok) -> one bf16 fragment
__device__ __forceinline__ bf16x8 load_frag(const float* p, bool ok) {
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 a = ok ? *reinterpret_cast<const float4*>(p) : z;
  const float4 b = ok ? *reinterpret_cast<const float4*>(p + 4) : z;
  return cvt8(a, b);
}

// row-major [64 rows][HD] bf16 tile image, row stride RS bytes
template <int HD>
struct Tile {
  static constexpr int RS = HD * 2 + 16;
  static constexpr int BYTES = KB * RS;
  static constexpr int F4 = KB * HD / 4 / 256;          // float4 per thread per tile
  // global -> registers (fp32), rows beyond T read as zeros
  __device__ static void load(float4 (&r)[F4], const float* base, long ld, int row0, int T, int tid) {
#pragma unroll
    for (int i = 0; i < F4; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx / (HD / 4), c4 = idx - row * (HD / 4);
      const int g = row0 + row;
      r[i] = g < T ? *reinterpret_cast<const float4*>(base + (long)g * ld + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __device__ static void store(unsigned char* tile, const float4 (&r)[F4], int tid) {
#pragma unroll
    for (int i = 0; i < F4; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx / (HD / 4), c4 = idx - row * (HD / 4);
      const bf16x4 v = {(__bf16)r[i].x, (__bf16)r[i].y, (__bf16)r[i].z, (__bf16)r[i].w};
      *reinterpret_cast<bf16x4*>(tile + row * RS + 8 * c4) = v;
    }
  }
  // A / B fragment by rows: lane (row l31, half hf) holds columns 16*ks + 8*hf .. + 7 of tile row `row`
  __device__ static bf16x8 row_frag(const unsigned char* tile, int row, int ks, int hf) {
    return *reinterpret_cast<const bf16x8*>(tile + row * RS + 32 * ks + 16 * hf);
  }
  // transposed fragment: lane (column c0 + (lane & 31), half hf) holds tile rows r0 + 8 (j >> 2) + 4 hf + (j & 3), j = 0..7
  // -- the k order of an accumulator tile used as the other operand (file header)
  __device__ static bf16x8 tr_frag(const unsigned char* tile, int r0, int c0, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const int row = r0 + 4 * (g >> 1) + (i >> 2), col = c0 + 16 * (g & 1) + 4 * (i & 3);
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const unsigned char* p = tile + row * RS + 2 * col;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 8 * RS));
    const s16x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return __builtin_bit_cast(bf16x8, v);
  }
};

// registers 8s .. 8s+7 of a 32x32 accumulator tile -> the bf16 fragment of k-step s
__device__ __forceinline__ bf16x8 acc_frag(const f32x16& a, int s) {
  return s == 0 ? bf16x8{(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3], (__bf16)a[4], (__bf16)a[5], (__bf16)a[6],
                         (__bf16)a[7]}
                : bf16x8{(__bf16)a[8], (__bf16)a[9], (__bf16)a[10], (__bf16)a[11], (__bf16)a[12], (__bf16)a[13],
                         (__bf16)a[14], (__bf16)a[15]};
}

__device__ __forceinline__ unsigned long long pad_mask64(const unsigned char* kp, int k0, int T, int lane) {
  const int k = k0 + lane;
  const bool masked = k >= T || (kp && kp[k] != 0);
  return __ballot(masked);
}

// ---------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256, HD > 64 ? 1 : 2) void ft_attn_fwd_kernel(const float* __restrict__ qkv,
                                                             const unsigned char* __restrict__ key_pad,
                                                             float* __restrict__ att, float* __restrict__ lse2, int T,
                                                             int nh, int dmodel, float scale, float p_drop, uint64_t seed) {
  typedef Tile<HD> TL;
  constexpr int DT = HD / 32, KS = HD / 16;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TL::BYTES];
  unsigned char* Kt = smem;
  unsigned char* Vt = smem + TL::BYTES;
  const int inst = blockIdx.y, b = inst / nh, h = inst - b * nh;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hf = lane >> 5;
  const int myq = blockIdx.x * 128 + wave * 32 + l31;
  const long ld = 3L * dmodel;
  const float* qbase = qkv + (long)b * T * ld + h * HD;
  const float* kbase = qbase + dmodel;
  const float* vbase = qbase + 2 * dmodel;
  const unsigned char* kp = key_pad ? key_pad + (long)b * T : nullptr;
  const float c = scale * LOG2E;
  const bool drop = p_drop > 0.f;
  const float inv_keep = drop ? 1.0f / (1.0f - p_drop) : 1.0f;

  bf16x8 qf[KS];                               // B operand of X = K Q^T: lane (query l31, hf): d = 16 ks + 8 hf + j
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qf[ks] = load_frag(qbase + (long)myq * ld + 16 * ks + 8 * hf, myq < T);

  f32x16 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
  float m = -INFINITY, l = 0.f;

  const int nkb = (T + KB - 1) / KB;
  float4 rk[TL::F4], rv[TL::F4];
  TL::load(rk, kbase, ld, 0, T, tid);
  TL::load(rv, vbase, ld, 0, T, tid);
  for (int kb = 0; kb < nkb; ++kb) {
    __syncthreads();                           // the previous block's fragment reads are done
    TL::store(Kt, rk, tid);
    TL::store(Vt, rv, tid);
    __syncthreads();
    if (kb + 1 < nkb) {                        // next block's loads fly during this block's MFMAs
      TL::load(rk, kbase, ld, (kb + 1) * KB, T, tid);
      TL::load(rv, vbase, ld, (kb + 1) * KB, T, tid);
    }
    const unsigned long long pm = pad_mask64(kp, kb * KB, T, lane);
    // X[key][query] = K Q^T
    f32x16 x[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) x[kt][e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        x[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TL::row_frag(Kt, 32 * kt + l31, ks, hf), qf[ks], x[kt], 0, 0, 0);
    }
    // online softmax over the keys (in-lane + the other lane half)
    float mloc = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int kk = 32 * kt + crow(e, hf);
        const float v = ((pm >> kk) & 1ull) ? -INFINITY : x[kt][e] * c;
        x[kt][e] = v;
        mloc = fmaxf(mloc, v);
      }
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float mnew = fmaxf(m, mloc);
    const float msafe = mnew == -INFINITY ? 0.f : mnew;
    const float alpha = exp2f(m - msafe);       // m = -inf: 0
    m = mnew;
    float lsum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float p = __builtin_amdgcn_exp2f(x[kt][e] - msafe);      // v_exp_f32 (the ocml form was 7 % of the kernel)
        lsum += p;
        if (drop) {
          const long idx = ((long)inst * T + myq) * T + (kb * KB + 32 * kt + crow(e, hf));
          p = ft_dropout_keep(seed, idx, p_drop) ? p * inv_keep : 0.f;
        }
        x[kt][e] = p;
      }
    l = l * alpha + lsum;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[dt][e] *= alpha;
    // O^T[d][query] += V^T[d][key] P[key][query]
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 pf = acc_frag(x[kt], s);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TL::tr_frag(Vt, 32 * kt + 16 * s, 32 * dt, lane), pf, o[dt], 0, 0, 0);
      }
  }
  l += __shfl_xor(l, 32, 64);
  if (myq < T) {
    const float inv = 1.0f / l;
    float* orow = att + ((long)b * T + myq) * dmodel + h * HD;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 v = make_float4(o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
        *reinterpret_cast<float4*>(orow + 32 * dt + 8 * g + 4 * hf) = v;
      }
    if (hf == 0) lse2[(long)inst * T + myq] = m + log2f(l);
  }
}

// delta[inst][q] = sum_d dO[q][d] * O[q][d]   (one thread per (query, head); rows of [B,T,dmodel])
__global__ void ft_attn_delta_kernel(const float* __restrict__ dout, const float* __restrict__ out, float* __restrict__ delta,
                                     int B, int T, int nh, int HD) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * nh * T) return;
  const int q = (int)(i % T);
  const int inst = (int)(i / T), b = inst / nh, h = inst - b * nh;
  const long off = ((long)b * T + q) * (long)(nh * HD) + h * HD;
  float s = 0.f;
  for (int d = 0; d < HD; d += 4) {
    const float4 a = *reinterpret_cast<const float4*>(dout + off + d), o = *reinterpret_cast<const float4*>(out + off + d);
    s += a.x * o.x + a.y * o.y + a.z * o.z + a.w * o.w;
  }
  delta[i] = s;
}

// ---------------------------------------------------------------------------------------------------
// backward, dQ: query on the lane (the forward's orientation).  Per key block: X = K Q^T, Y = V dO^T (= dP^T),
// P = exp2(c X - lse2[query]), dS = P (keep/(1-p) Y - delta[query]), dQ^T[d][query] += K^T[d][key] dS[key][query].
// ---------------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256, HD > 64 ? 1 : 2) void ft_attn_bwd_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ datt,
                                                                const unsigned char* __restrict__ key_pad,
                                                                const float* __restrict__ lse2,
                                                                const float* __restrict__ delta, float* __restrict__ dqkv,
                                                                int T, int nh, int dmodel, float scale, float p_drop,
                                                                uint64_t seed) {
  typedef Tile<HD> TL;
  constexpr int DT = HD / 32, KS = HD / 16;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TL::BYTES];
  unsigned char* Kt = smem;
  unsigned char* Vt = smem + TL::BYTES;
  const int inst = blockIdx.y, b = inst / nh, h = inst - b * nh;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hf = lane >> 5;
  const int myq = blockIdx.x * 128 + wave * 32 + l31;
  const bool qok = myq < T;
  const long ld = 3L * dmodel;
  const float* qbase = qkv + (long)b * T * ld + h * HD;
  const float* kbase = qbase + dmodel;
  const float* vbase = qbase + 2 * dmodel;
  const unsigned char* kp = key_pad ? key_pad + (long)b * T : nullptr;
  const float c = scale * LOG2E;
  const bool drop = p_drop > 0.f;
  const float inv_keep = drop ? 1.0f / (1.0f - p_drop) : 1.0f;

  bf16x8 qf[KS], gf[KS];                       // Q^T and dO^T fragments of this wave's queries
  const float* grow = datt + ((long)b * T + myq) * dmodel + h * HD;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    qf[ks] = load_frag(qbase + (long)myq * ld + 16 * ks + 8 * hf, qok);
    gf[ks] = load_frag(grow + 16 * ks + 8 * hf, qok);
  }
  const float L = qok ? lse2[(long)inst * T + myq] : 0.f;
  const float D = qok ? delta[(long)inst * T + myq] : 0.f;
  f32x16 dq[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) dq[dt][e] = 0.f;

  const int nkb = (T + KB - 1) / KB;
  float4 rk[TL::F4], rv[TL::F4];
  TL::load(rk, kbase, ld, 0, T, tid);
  TL::load(rv, vbase, ld, 0, T, tid);
  for (int kb = 0; kb < nkb; ++kb) {
    __syncthreads();
    TL::store(Kt, rk, tid);
    TL::store(Vt, rv, tid);
    __syncthreads();
    if (kb + 1 < nkb) {
      TL::load(rk, kbase, ld, (kb + 1) * KB, T, tid);
      TL::load(rv, vbase, ld, (kb + 1) * KB, T, tid);
    }
    const unsigned long long pm = pad_mask64(kp, kb * KB, T, lane);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      f32x16 x, y;
#pragma unroll
      for (int e = 0; e < 16; ++e) x[e] = 0.f, y[e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TL::row_frag(Kt, 32 * kt + l31, ks, hf), qf[ks], x, 0, 0, 0);
        y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TL::row_frag(Vt, 32 * kt + l31, ks, hf), gf[ks], y, 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int kk = 32 * kt + crow(e, hf);
        const bool masked = (pm >> kk) & 1ull;
        const float p = masked ? 0.f : __builtin_amdgcn_exp2f(x[e] * c - L);
        float g = y[e];
        if (drop) {
          const long idx = ((long)inst * T + myq) * T + (kb * KB + kk);
          g = ft_dropout_keep(seed, idx, p_drop) ? g * inv_keep : 0.f;
        }
        x[e] = scale * p * (g - D);            // dS[key][query]
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 sf = acc_frag(x, s);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
          dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TL::tr_frag(Kt, 32 * kt + 16 * s, 32 * dt, lane), sf, dq[dt], 0, 0, 0);
      }
    }
  }
  if (qok) {
    float* orow = dqkv + ((long)b * T + myq) * ld + h * HD;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(orow + 32 * dt + 8 * g + 4 * hf) =
            make_float4(dq[dt][4 * g], dq[dt][4 * g + 1], dq[dt][4 * g + 2], dq[dt][4 * g + 3]);
  }
}

// ---------------------------------------------------------------------------------------------------
// backward, dK and dV: key on the lane.  A wave owns 32 keys (K^T and V^T fragments resident), the workgroup sweeps the
// queries in blocks of 64: X = Q K^T [query][key] with -lse2[query] as the initial accumulator, Y = dO V^T with
// -delta[query]; P = exp2(X), Pd = keep/(1-p) P, dS = scale P (keep/(1-p) Y' ...); dV^T[d][key] += dO^T[d][query] Pd,
// dK^T[d][key] += Q^T[d][query] dS -- both sum over the accumulator tiles' row index: no LDS transpose.
// ---------------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256, 1) void ft_attn_bwd_dkv_kernel(const float* __restrict__ qkv, const float* __restrict__ datt,
                                                                 const unsigned char* __restrict__ key_pad,
                                                                 const float* __restrict__ lse2,
                                                                 const float* __restrict__ delta, float* __restrict__ dqkv,
                                                                 int T, int nh, int dmodel, float scale, float p_drop,
                                                                 uint64_t seed) {
  typedef Tile<HD> TL;
  constexpr int DT = HD / 32, KS = HD / 16;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TL::BYTES];
  __shared__ float sL[KB], sD[KB];
  unsigned char* Qt = smem;
  unsigned char* Gt = smem + TL::BYTES;        // dO tile
  const int inst = blockIdx.y, b = inst / nh, h = inst - b * nh;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hf = lane >> 5;
  const int myk = blockIdx.x * 128 + wave * 32 + l31;
  const bool kok = myk < T;
  const long ld = 3L * dmodel;
  const float* qbase = qkv + (long)b * T * ld + h * HD;
  const float* kbase = qbase + dmodel;
  const float* vbase = qbase + 2 * dmodel;
  const float* gbase = datt + (long)b * T * dmodel + h * HD;
  const bool kmask = !kok || (key_pad && key_pad[(long)b * T + myk] != 0);
  const float c = scale * LOG2E;
  const bool drop = p_drop > 0.f;
  const float inv_keep = drop ? 1.0f / (1.0f - p_drop) : 1.0f;

  bf16x8 kf[KS], vf[KS];                       // B operands: lane (key l31, hf): d = 16 ks + 8 hf + j
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    kf[ks] = load_frag(kbase + (long)myk * ld + 16 * ks + 8 * hf, kok);
    vf[ks] = load_frag(vbase + (long)myk * ld + 16 * ks + 8 * hf, kok);
  }
  f32x16 dk[DT], dv[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) dk[dt][e] = 0.f, dv[dt][e] = 0.f;

  const int nqb = (T + KB - 1) / KB;
  float4 rq[TL::F4], rg[TL::F4];
  TL::load(rq, qbase, ld, 0, T, tid);
  TL::load(rg, gbase, dmodel, 0, T, tid);
  for (int qb = 0; qb < nqb; ++qb) {
    __syncthreads();
    TL::store(Qt, rq, tid);
    TL::store(Gt, rg, tid);
    if (tid < KB) {
      const int q = qb * KB + tid;
      sL[tid] = q < T ? lse2[(long)inst * T + q] : INFINITY;      // exp2(x - inf) = 0: rows beyond T contribute nothing
      sD[tid] = q < T ? delta[(long)inst * T + q] : 0.f;
    }
    __syncthreads();
    if (qb + 1 < nqb) {
      TL::load(rq, qbase, ld, (qb + 1) * KB, T, tid);
      TL::load(rg, gbase, dmodel, (qb + 1) * KB, T, tid);
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      f32x16 x, y;
#pragma unroll
      for (int e = 0; e < 16; ++e) x[e] = 0.f, y[e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TL::row_frag(Qt, 32 * qt + l31, ks, hf), kf[ks], x, 0, 0, 0);
        y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TL::row_frag(Gt, 32 * qt + l31, ks, hf), vf[ks], y, 0, 0, 0);
      }
      f32x16 pd;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int qq = 32 * qt + crow(e, hf);                       // query (tile row) of this register
        const float p = kmask ? 0.f : __builtin_amdgcn_exp2f(x[e] * c - sL[qq]);
        float keepf = 1.f;
        if (drop) {
          const long idx = ((long)inst * T + (qb * KB + qq)) * T + myk;
          keepf = ft_dropout_keep(seed, idx, p_drop) ? inv_keep : 0.f;
        }
        pd[e] = p * keepf;
        x[e] = scale * p * (y[e] * keepf - sD[qq]);                 // dS[query][key]
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 pf = acc_frag(pd, s), sf = acc_frag(x, s);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TL::tr_frag(Gt, 32 * qt + 16 * s, 32 * dt, lane), pf, dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TL::tr_frag(Qt, 32 * qt + 16 * s, 32 * dt, lane), sf, dk[dt], 0, 0, 0);
        }
      }
    }
  }
  if (kok) {
    float* krow = dqkv + ((long)b * T + myk) * ld + dmodel + h * HD;
    float* vrow = krow + dmodel;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<float4*>(krow + 32 * dt + 8 * g + 4 * hf) =
            make_float4(dk[dt][4 * g], dk[dt][4 * g + 1], dk[dt][4 * g + 2], dk[dt][4 * g + 3]);
        *reinterpret_cast<float4*>(vrow + 32 * dt + 8 * g + 4 * hf) =
            make_float4(dv[dt][4 * g], dv[dt][4 * g + 1], dv[dt][4 * g + 2], dv[dt][4 * g + 3]);
      }
  }
}

}  // namespace

extern "C" {

size_t ft_attn_workspace(int B, int T, int nheads) { return (size_t)B * nheads * T * sizeof(float); }

int ft_attn_fwd(const float* qkv, const unsigned char* key_pad, float* att, float* lse2, int B, int T, int nheads, int hd,
                float scale, float p_drop, uint64_t seed, void* stream) {
  FT_REQUIRE(hd == 64 || hd == 128, "attn_fwd: head_dim %d (64 or 128)", hd);
  FT_REQUIRE(B >= 0 && T >= 0 && nheads >= 1 && p_drop >= 0.f && p_drop < 1.f, "attn_fwd: bad arguments");
  FT_REQUIRE(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)att % 16) == 0, "attn_fwd: buffers must be 16-byte aligned");
  if (B == 0 || T == 0) return FT_OK;
  const dim3 grid(ft_cdiv(T, 128), B * nheads);
  FT_REQUIRE(grid.y <= 65535, "attn_fwd: too many (batch, head) instances");
  const int dmodel = nheads * hd;
  if (hd == 128)
    hipLaunchKernelGGL(ft_attn_fwd_kernel<128>, grid, dim3(256), 0, (hipStream_t)stream, qkv, key_pad, att, lse2, T, nheads,
                       dmodel, scale, p_drop, seed);
  else
    hipLaunchKernelGGL(ft_attn_fwd_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, qkv, key_pad, att, lse2, T, nheads,
                       dmodel, scale, p_drop, seed);
  return ft_check_launch("attn_fwd");
}

int ft_attn_bwd(const float* qkv, const float* att, const float* datt, const unsigned char* key_pad, const float* lse2,
                float* dqkv, int B, int T, int nheads, int hd, float scale, float p_drop, uint64_t seed, void* workspace,
                size_t workspace_bytes, void* stream) {
  FT_REQUIRE(hd == 64 || hd == 128, "attn_bwd: head_dim %d (64 or 128)", hd);
  FT_REQUIRE(B >= 0 && T >= 0 && nheads >= 1 && p_drop >= 0.f && p_drop < 1.f, "attn_bwd: bad arguments");
  FT_REQUIRE(workspace && workspace_bytes >= ft_attn_workspace(B, T, nheads), "attn_bwd: workspace too small");
  if (B == 0 || T == 0) return FT_OK;
  hipStream_t s = (hipStream_t)stream;
  float* delta = (float*)workspace;
  const long n = (long)B * nheads * T;
  hipLaunchKernelGGL(ft_attn_delta_kernel, dim3(ft_cdiv(n, 256)), dim3(256), 0, s, datt, att, delta, B, T, nheads, hd);
  const dim3 grid(ft_cdiv(T, 128), B * nheads);
  FT_REQUIRE(grid.y <= 65535, "attn_bwd: too many (batch, head) instances");
  const int dmodel = nheads * hd;
  if (hd == 128) {
    hipLaunchKernelGGL(ft_attn_bwd_dq_kernel<128>, grid, dim3(256), 0, s, qkv, datt, key_pad, lse2, delta, dqkv, T, nheads,
                       dmodel, scale, p_drop, seed);
    hipLaunchKernelGGL(ft_attn_bwd_dkv_kernel<128>, grid, dim3(256), 0, s, qkv, datt, key_pad, lse2, delta, dqkv, T, nheads,
                       dmodel, scale, p_drop, seed);
  } else {
    hipLaunchKernelGGL(ft_attn_bwd_dq_kernel<64>, grid, dim3(256), 0, s, qkv, datt, key_pad, lse2, delta, dqkv, T, nheads,
                       dmodel, scale, p_drop, seed);
    hipLaunchKernelGGL(ft_attn_bwd_dkv_kernel<64>, grid, dim3(256), 0, s, qkv, datt, key_pad, lse2, delta, dqkv, T, nheads,
                       dmodel, scale, p_drop, seed);
  }
  return ft_check_launch("attn_bwd");
}

}  // extern "C"
