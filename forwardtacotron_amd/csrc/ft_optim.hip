// clip_grad_norm_ + Adam over FLAT fp32 buffers (trainer/forward_trainer.py:95-99 ; train_forward.py:76).
// The trainer keeps every parameter / gradient / Adam moment in one contiguous buffer each, so the whole
// optimiser is three launches: sum-of-squares partials, clip coefficient, fused Adam update.  HBM-bound:
// the Adam kernel reads p,g,m,v and writes p,m,v once (28 B per parameter), 16-B lanes.
#include "ft_common.h"

namespace {

__global__ __launch_bounds__(256) void ft_sumsq_partial_kernel(const float* __restrict__ g, long n,
                                                               double* __restrict__ partial) {
  __shared__ double red[4];
  double acc = 0.0;
  const long n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 v = g4[i];
    acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    float v = g[(n4 << 2) + threadIdx.x];
    acc += (double)v * v;
  }
  acc = ft_wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// norm = sqrt(sum)*pre_scale ; coef = pre_scale * min(1, max_norm / (norm + 1e-6))   (max_norm <= 0: no clipping)
// fault: the device's sticky recurrence-fault word (ft_rnn_persist.hip).  If a persistent recurrence timed out since the
// word was last cleared, the gradients of this step are garbage: out[2] = 1 makes ft_adam_kernel skip the update, the
// coefficient is 0 and the reported norm NaN, so that nothing downstream can mistake the step for a valid one.
// remote: the data-parallel fault lane (ft_fault_lane_set on every rank, SUM all-reduced with the gradients): > 0 means
// SOME rank's recurrence faulted, and then every rank's all-reduced gradient is garbage -- every rank skips alike
// (out[2] = 2 where only another rank faulted).
__global__ __launch_bounds__(256) void ft_clip_coef_kernel(const double* __restrict__ partial, int nblocks,
                                                           float max_norm, float pre_scale,
                                                           const unsigned* __restrict__ fault,
                                                           const float* __restrict__ remote,
                                                           float* __restrict__ out) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) s += partial[i];   // fixed assignment -> reproducible
  s = ft_wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x != 0) return;
  s = red[0] + red[1] + red[2] + red[3];
  float norm = (float)sqrt(s) * pre_scale;
  float c = 1.0f;
  if (max_norm > 0.f) {
    c = max_norm / (norm + 1e-6f);
    if (c > 1.0f) c = 1.0f;
  }
  const bool mine = fault && __hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
  const bool bad = mine || (remote && !(remote[0] == 0.f));      // (a NaN lane counts as a fault)
  out[0] = bad ? 0.f : c * pre_scale;
  out[1] = bad ? __uint_as_float(0x7fc00000u) : norm;
  out[2] = mine ? 1.f : (bad ? 2.f : 0.f);
  out[3] = 0.f;
}

__global__ void ft_fault_lane_kernel(const unsigned* __restrict__ fault, float* __restrict__ lane) {
  if (threadIdx.x < 4)
    lane[threadIdx.x] = (threadIdx.x == 0 && fault &&
                         __hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ? 1.f : 0.f;
}

// dst <- snapshot (4-byte words) iff the step was marked faulted: the buffers a forward pass updates in place (BatchNorm
// running statistics, num_batches_tracked, `step`) go back to their values from before the faulted step
__global__ __launch_bounds__(256) void ft_guarded_restore_kernel(unsigned* __restrict__ dst,
                                                                 const unsigned* __restrict__ snap, long n,
                                                                 const float* __restrict__ coef) {
  if (coef[2] == 0.f) return;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = snap[i];
}

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float b1, float b2, float eps,
                                      float step_size, float inv_sqrt_bc2) {
  m = m + (g - m) * (1.0f - b1);                    // exp_avg.lerp_(grad, 1-beta1)
  v = v * b2 + (1.0f - b2) * g * g;
  float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
  p = p - step_size * (m / denom);
}

__global__ __launch_bounds__(256) void ft_adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                      float* __restrict__ m, float* __restrict__ v, long n, float lr,
                                                      float b1, float b2, float eps, float bc1, float bc2,
                                                      const float* __restrict__ coef) {
  if (coef && coef[2] != 0.f) return;             // recurrence fault (ft_clip_coef_kernel): leave p, m, v untouched
  const float c = coef ? coef[0] : 1.0f;
  const float step_size = lr / bc1;
  const float isb2 = 1.0f / sqrtf(bc2);
  const long n4 = n >> 2;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    adam1(pp.x, gg.x * c, mm.x, vv.x, b1, b2, eps, step_size, isb2);
    adam1(pp.y, gg.y * c, mm.y, vv.y, b1, b2, eps, step_size, isb2);
    adam1(pp.z, gg.z * c, mm.z, vv.z, b1, b2, eps, step_size, isb2);
    adam1(pp.w, gg.w * c, mm.w, vv.w, b1, b2, eps, step_size, isb2);
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  } else if (i < n4 + (n & 3)) {
    long k = (n4 << 2) + (i - n4);
    float pp = p[k], mm = m[k], vv = v[k];
    adam1(pp, g[k] * c, mm, vv, b1, b2, eps, step_size, isb2);
    p[k] = pp; m[k] = mm; v[k] = vv;
  }
}

}  // namespace

extern "C" {

size_t ft_grad_norm_workspace(void) { return 2048 * sizeof(double); }

int ft_clip_grad_norm(const float* grads, long n, float max_norm, float pre_scale, const float* fault_lane,
                      float* coef_and_norm, void* workspace, size_t workspace_bytes, void* stream) {
  FT_REQUIRE(n >= 0, "clip_grad_norm: bad n");
  FT_REQUIRE(((uintptr_t)grads % 16) == 0, "clip_grad_norm: gradient buffer must be 16-byte aligned");
  FT_REQUIRE(workspace && workspace_bytes >= ft_grad_norm_workspace(), "clip_grad_norm: workspace too small");
  int nb = ft_cdiv((n >> 2) + 1, 256 * 4);
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(ft_sumsq_partial_kernel, dim3(nb), dim3(256), 0, s, grads, n, (double*)workspace);
  hipLaunchKernelGGL(ft_clip_coef_kernel, dim3(1), dim3(256), 0, s, (const double*)workspace, nb, max_norm, pre_scale,
                     ft_rnn_fault_word(), fault_lane, coef_and_norm);
  return ft_check_launch("clip_grad_norm");
}

int ft_fault_lane_set(float* lane, void* stream) {
  FT_REQUIRE(lane && ((uintptr_t)lane % 16) == 0, "fault_lane_set: lane must be 4 floats, 16-byte aligned");
  hipLaunchKernelGGL(ft_fault_lane_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ft_rnn_fault_word(), lane);
  return ft_check_launch("fault_lane_set");
}

int ft_guarded_restore(void* dst, const void* snapshot, long nwords, const float* coef, void* stream) {
  FT_REQUIRE(nwords >= 0 && coef, "guarded_restore: bad arguments");
  if (nwords == 0) return FT_OK;
  long nb = ft_cdiv(nwords, 256);
  if (nb > 256) nb = 256;
  hipLaunchKernelGGL(ft_guarded_restore_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, (unsigned*)dst,
                     (const unsigned*)snapshot, nwords, coef);
  return ft_check_launch("guarded_restore");
}

int ft_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
                 float beta2, float eps, long step, const float* coef, void* stream) {
  FT_REQUIRE(n >= 0 && step >= 1, "adam_step: bad n/step");
  FT_REQUIRE(((uintptr_t)params % 16) == 0 && ((uintptr_t)grads % 16) == 0 && ((uintptr_t)exp_avg % 16) == 0 &&
                 ((uintptr_t)exp_avg_sq % 16) == 0,
             "adam_step: buffers must be 16-byte aligned");
  if (n == 0) return FT_OK;
  float bc1 = 1.0f - (float)pow((double)beta1, (double)step);
  float bc2 = 1.0f - (float)pow((double)beta2, (double)step);
  long threads = (n >> 2) + (n & 3);
  hipLaunchKernelGGL(ft_adam_kernel, dim3(ft_cdiv(threads, 256)), dim3(256), 0, (hipStream_t)stream, params, grads,
                     exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, bc1, bc2, coef);
  return ft_check_launch("adam_step");
}

}  // extern "C"
