// Element-wise / data-movement kernels (HBM-bound): conv weight packing, embedding, highway gates,
// maxpool, residual, conditioning projections, layout changes.  See include/fwdtaco_hip.h.
#include "ft_common.h"

namespace {

// w[Cout][Cin][k] -> wp[k][Cout][Cin]
__global__ void ft_pack_conv_w_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int k) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)Cout * Cin * k;
  if (idx >= total) return;
  int j = (int)(idx / ((long)Cout * Cin));
  long rem = idx - (long)j * Cout * Cin;     // co*Cin + ci
  wp[idx] = w[rem * k + j];
}

}  // namespace

extern "C" {

int ft_conv_pack_weight(const float* w, float* wp, int Cout, int Cin, int k, void* stream) {
  FT_REQUIRE(Cout >= 0 && Cin >= 0 && k >= 1, "conv_pack_weight: bad dims");
  long total = (long)Cout * Cin * k;
  if (total == 0) return FT_OK;
  hipLaunchKernelGGL(ft_pack_conv_w_kernel, dim3(ft_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, wp, Cout,
                     Cin, k);
  return ft_check_launch("conv_pack_weight");
}

}  // extern "C"
