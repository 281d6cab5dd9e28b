// The PRODUCT NT kernels (ft_gemm_b3.hip, included as source) timed from a bare HIP harness with the lab kernel's
// problem, to separate kernel time from the Python wrapper.  hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize
//   -I forwardtacotron_amd/csrc -I include lab/gemm_prod_lab.hip -o lab/gemm_prod_lab.bin
#include "../forwardtacotron_amd/csrc/ft_gemm_b3.hip"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
int ft_gemm_precision() { return 0; }
int ft_fail(const char*, ...) { return -1; }

static void run(const char* name, const float* A, const float* B, float* C, int M, int N, int K) {
  FtGemmBatch batch;
  memset(&batch, 0, sizeof(batch));
  FtGemmTask& t = batch.t[0];
  t.A = A; t.B = B; t.C = C; t.lda = K; t.ldb = K; t.ldc = N; t.M = M; t.N = N; t.K = K; t.taps = 1;
  t.amap = ft_rowmap_identity(M); t.cmap = ft_rowmap_identity(M); t.nz = t.nz1 = 1; t.a_vec = t.b_vec = 1;
  for (int i = 1; i < FT_MAX_TASKS; ++i) batch.t[i] = batch.t[0];
  dim3 grid((M + 127) / 128, (N + 127) / 128, 1);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) ft_launch_gemm_rows_b3(batch, true, grid, 0);
  (void)hipEventRecord(e0);
  const int n = 10;
  for (int i = 0; i < n; ++i) ft_launch_gemm_rows_b3(batch, true, grid, 0);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= n;
  printf("%-30s M %d N %d K %d: %8.1f us  %6.1f TF(f32-eq)\n", name, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
  fflush(stdout);
}

int main() {
  const int M = 27136, N = 2048, K = 1024;
  float *A, *B, *C;
  (void)hipMalloc(&A, (size_t)M * 4096 * 4);
  (void)hipMalloc(&B, (size_t)4096 * 4096 * 4);
  (void)hipMalloc(&C, (size_t)M * 4096 * 4);
  std::vector<float> h((size_t)M * 4096);
  srand(1);
  for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  (void)hipMemcpy(A, h.data(), (size_t)M * 4096 * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(B, h.data(), (size_t)4096 * 4096 * 4, hipMemcpyHostToDevice);
  const char* e = getenv("FT_GEMM_PIPE");
  run(e && e[0] == '0' ? "product two-barrier" : "product pipelined", A, B, C, M, N, K);
  run("same, M = 26912", A, B, C, 26912, N, K);
  run("K = 512", A, B, C, 26912, N, 512);
  run("N = 512, K = 4096", A, B, C, 26912, 512, 4096);
  run("N = 4096, K = 512", A, B, C, 26912, 4096, 512);
  return 0;
}
