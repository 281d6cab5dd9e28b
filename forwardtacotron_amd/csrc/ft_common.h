// Common helpers for the forwardtacotron_amd HIP kernels (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FT_OK 0
#define FT_ERR_ARG 1
#define FT_ERR_HIP 2

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

void ft_set_error(const char* fmt, ...);
int ft_check_launch(const char* what);

#define FT_REQUIRE(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      ft_set_error(__VA_ARGS__);         \
      return FT_ERR_ARG;                 \
    }                                    \
  } while (0)

// device address of the current device's sticky recurrence-fault word (ft_rnn_persist.hip); read by the optimizer kernels
unsigned* ft_rnn_fault_word();

static inline int ft_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// accurate (ocml) forms: the recurrences run 841 dependent steps and must hold 1e-4 abs
__device__ __forceinline__ float ft_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float ft_tanh(float x) { return tanhf(x); }
// hardware-transcendental forms (v_exp_f32 / v_rcp_f32, ~1 ulp each): absolute error <= ~2e-7, a fifth of the
// instructions of the ocml forms above.  The persistent recurrences' cell update sits on the dependent path of every
// time step (one cell per lane, nothing to hide it behind).
__device__ __forceinline__ float ft_sigmoid_fast(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float ft_tanh_fast(float x) {
  const float e = __builtin_amdgcn_exp2f(-2.88539008177792681f * fabsf(x));      // exp(-2|x|) in (0, 1]
  return copysignf((1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e), x);
}

__device__ __forceinline__ float ft_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ double ft_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// counter-based dropout mask (ft_dropout and the kernels that fuse it): element i of a tensor is kept iff u(seed, i) >= p;
// the backward re-derives the same mask from the seed, so no mask tensor is stored.  32-bit mixing (three multiplies:
// the attention kernels draw one decision per score element, forward and twice in the backward -- the 64-bit finalizer
// used until round 3 was 19 % of the fused attention forward); index and seed enter in full.
__device__ __forceinline__ uint32_t ft_hash32(uint64_t seed, uint64_t i) {
  uint32_t h = (uint32_t)i ^ ((uint32_t)(i >> 32) * 0x9E3779B1u);
  h = (h ^ (uint32_t)seed) * 0x85EBCA6Bu;
  h ^= h >> 15;
  h = (h + (uint32_t)(seed >> 32)) * 0xC2B2AE35u;
  h ^= h >> 13;
  h *= 0x27D4EB2Fu;
  h ^= h >> 16;
  return h;
}
__device__ __forceinline__ bool ft_dropout_keep(uint64_t seed, long i, float p) {
  const uint32_t h = ft_hash32(seed, (uint64_t)i);
  return (float)(h >> 8) * (1.0f / 16777216.0f) >= p;
}
