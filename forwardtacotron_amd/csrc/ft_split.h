// The exact three-way bf16 split of fp32 operands of the GEMM kernels that split while staging (ft_gemm_b3.hip): ONE
// definition for every place a value is split.
#pragma once
#include "ft_common.h"

typedef __bf16 ft_bf16x2 __attribute__((ext_vector_type(2)));
// round-to-nearest-even bf16 of two floats, packed (low half = a): v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned ft_rn_pack(float a, float b) {
  const ft_bf16x2 p = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, p);
}
// hi = rn(x), r1 = x - hi (exact: hi shares x's leading bits), mid = rn(r1), r2 = r1 - mid (exact), lo = rn(r2):
// |x - hi - mid - lo| <= 2^-27 |x|.  The packed value is made opaque (an empty asm: no instruction) before it is taken
// apart: with the conversion visible, "hi << 16" becomes a SECOND v_cvt_pk_bf16_f32 of (a, 0) -- one more VALU per pair
// and level.  [Real inline-asm instructions are not an option in the GEMM loop: sched_group_barrier does not count them
// as VALU and the MFMA interleave falls apart.]
__device__ __forceinline__ void ft_split_pair(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = ft_rn_pack(a, b);
  asm("" : "+v"(hi));
  const float a1 = a - __uint_as_float(hi << 16), b1 = b - __uint_as_float(hi & 0xFFFF0000u);
  mid = ft_rn_pack(a1, b1);
  asm("" : "+v"(mid));
  const float a2 = a1 - __uint_as_float(mid << 16), b2 = b1 - __uint_as_float(mid & 0xFFFF0000u);
  lo = ft_rn_pack(a2, b2);
}
