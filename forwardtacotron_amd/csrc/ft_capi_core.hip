// C-ABI glue, part 1: error handling, GEMM-shaped ops (linear / channels-last conv / conv bank) and
// the LengthRegulator.  Declarations + reference citations live in include/fwdtaco_hip.h.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/fwdtaco_hip.h"
#include <mutex>
#include <utility>
#include <vector>

#include "ft_gemm.h"

static thread_local char g_err[512] = "";

void ft_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int ft_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    ft_set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
    return FT_ERR_HIP;
  }
  return FT_OK;
}

int ft_lr_scan_impl(float*, int, int, int*, int*, hipStream_t);
int ft_lr_expand_impl(const float*, const int*, float*, int*, int, int, int, int, hipStream_t, int = 0, const float* = nullptr);
int ft_lr_bwd_impl(const float*, const int*, float*, int, int, int, int, hipStream_t, int = 0, float* = nullptr);

// ft_bn.hip: statistics partials by the stand-alone column pass (C++ linkage), and the partial-buffer size query
int ft_bn_stat_partials(const float* y, int B, int Tbuf, int C, int group, double* partial, hipStream_t s);
extern "C" size_t ft_conv_stats_workspace(int B, int Tbuf, int C);

// Workgroup slots of the 2-per-CU GEMM kernels on a stream (512 on an unrestricted one): the weight-gradient planner sizes
// its one-wave grids by it (ft_gemm.hip: plan_tn).
namespace {
std::mutex g_slots_mu;
std::vector<std::pair<hipStream_t, int>> g_stream_slots;
}  // namespace
void ft_note_stream_slots(hipStream_t s, int slots) {
  std::lock_guard<std::mutex> lk(g_slots_mu);
  for (auto& e : g_stream_slots)
    if (e.first == s) {
      e.second = slots;
      return;
    }
  g_stream_slots.push_back({s, slots});
}
int ft_stream_slots(hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_slots_mu);
  for (const auto& e : g_stream_slots)
    if (e.first == s) return e.second;
  return 512;
}

extern "C" {

const char* ft_last_error(void) { return g_err; }
int ft_abi_version(void) { return FWDTACO_ABI_VERSION; }

int ft_device_info(int* cu_count, int* is_gfx950) {
  hipDeviceProp_t p;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) {
    ft_set_error("ft_device_info: no HIP device");
    return FT_ERR_HIP;
  }
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (is_gfx950) *is_gfx950 = strncmp(p.gcnArchName, "gfx950", 6) == 0;
  return FT_OK;
}

// A stream whose kernels may only use the first `cus_per_xcd` CUs of every XCD (bit i of a HIP CU mask = CU i / 8 of
// XCD i % 8 on this part: lab/cumask_probe2.hip).  The trainer runs its weight-gradient side stream on such a stream:
// those GEMMs are one resident wave of long-running workgroups that fill every CU's register file, so the small
// dependent kernels of the step's critical stream found no slot until a whole GEMM had finished, whatever their priority.
int ft_stream_create_cu_limited(int cus_per_xcd, void** stream) {
  FT_REQUIRE(stream != nullptr, "ft_stream_create_cu_limited: null result pointer");
  int cus = 0, gfx950 = 0;
  if (ft_device_info(&cus, &gfx950) != FT_OK) return FT_ERR_HIP;
  const int per = cus / 8;
  FT_REQUIRE(cus % 8 == 0 && per <= 32 && cus_per_xcd >= 1 && cus_per_xcd <= per,
             "ft_stream_create_cu_limited: %d CUs per XCD requested, the device has %d CUs", cus_per_xcd, cus);
  uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 8 * cus_per_xcd; ++i) mask[i >> 5] |= 1u << (i & 31);
  hipStream_t s = nullptr;
  if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) {
    ft_set_error("ft_stream_create_cu_limited: hipExtStreamCreateWithCUMask failed");
    return FT_ERR_HIP;
  }
  *stream = (void*)s;
  ft_note_stream_slots(s, 2 * 8 * cus_per_xcd);
  return FT_OK;
}
int ft_stream_destroy(void* stream) {
  if (stream) ft_note_stream_slots((hipStream_t)stream, 512);
  if (stream && hipStreamDestroy((hipStream_t)stream) != hipSuccess) {
    ft_set_error("ft_stream_destroy: hipStreamDestroy failed");
    return FT_ERR_HIP;
  }
  return FT_OK;
}

// ------------------------------------------------------------------------------------------------
static int check_tm(const char* what, int rows, int tm_B) {
  if (tm_B > 0 && rows % tm_B != 0) {
    ft_set_error("%s: rows (%d) not a multiple of the time-major batch (%d)", what, rows, tm_B);
    return FT_ERR_ARG;
  }
  return FT_OK;
}

int ft_linear_fwd(const float* x, long ldx, const float* w, const float* bias, float* y, long ldy, int rows, int in_f,
                  int out_f, int relu, int accumulate, int x_tm_B, int y_tm_B, void* stream) {
  if (check_tm("linear_fwd", rows, x_tm_B) || check_tm("linear_fwd", rows, y_tm_B)) return FT_ERR_ARG;
  FtGemmBatch b;
  memset(&b, 0, sizeof(b));
  FtGemmTask& t = b.t[0];
  t.A = x; t.B = w; t.C = y; t.bias = bias;
  t.lda = ldx; t.ldb = in_f; t.ldc = ldy; t.b_tap_stride = 0;
  t.M = rows; t.N = out_f; t.K = in_f; t.taps = 1;
  t.amap = ft_rowmap_layout(rows, x_tm_B);
  t.cmap = ft_rowmap_layout(rows, y_tm_B);
  t.relu = relu; t.accumulate = accumulate;
  return ft_launch_gemm_rows(&b, 1, false, (hipStream_t)stream);
}

static int linear_multi_fwd(const float* x, long ldx, int ntasks, const float* const* w, const float* const* bias,
                            float* y, long ldy, const int* col_offset, const int* out_f, int rows, int in_f, int relu,
                            int x_tm_B, int y_tm_B, long as_rows, void* stream);

int ft_linear_multi_fwd(const float* x, long ldx, int ntasks, const float* const* w, const float* const* bias,
                        float* y, long ldy, const int* col_offset, const int* out_f, int rows, int in_f, int relu,
                        int x_tm_B, int y_tm_B, void* stream) {
  return linear_multi_fwd(x, ldx, ntasks, w, bias, y, ldy, col_offset, out_f, rows, in_f, relu, x_tm_B, y_tm_B, 0, stream);
}
int ft_linear_multi_fwd_as(const float* x, long ldx, int ntasks, const float* const* w, const float* const* bias,
                           float* y, long ldy, const int* col_offset, const int* out_f, int rows, int in_f, long as_rows,
                           void* stream) {
  FT_REQUIRE(as_rows >= 1, "linear_multi_fwd_as: as_rows (%ld) must be positive", as_rows);
  return linear_multi_fwd(x, ldx, ntasks, w, bias, y, ldy, col_offset, out_f, rows, in_f, 0, 0, 0, as_rows, stream);
}

static int linear_multi_fwd(const float* x, long ldx, int ntasks, const float* const* w, const float* const* bias,
                            float* y, long ldy, const int* col_offset, const int* out_f, int rows, int in_f, int relu,
                            int x_tm_B, int y_tm_B, long as_rows, void* stream) {
  FT_REQUIRE(ntasks >= 1 && ntasks <= FT_MAX_TASKS, "linear_multi_fwd: ntasks %d out of range", ntasks);
  if (check_tm("linear_multi_fwd", rows, x_tm_B) || check_tm("linear_multi_fwd", rows, y_tm_B)) return FT_ERR_ARG;
  FtGemmBatch b;
  memset(&b, 0, sizeof(b));
  if (as_rows > 0) {      // the tile (= the kernel, = the rounding) a launch over as_rows rows of the same layers takes
    long tiles = 0;
    int maxN = 0;
    for (int i = 0; i < ntasks; ++i) {
      tiles += (long)ft_cdiv(as_rows, 128) * ft_cdiv(out_f[i], 128);
      if (out_f[i] > maxN) maxN = out_f[i];
    }
    b.force_tile = ft_rows_tile_is_big(tiles, as_rows > 0x7fffffffL ? 0x7fffffff : (int)as_rows, maxN) ? 2 : 1;
  }
  for (int i = 0; i < ntasks; ++i) {
    FtGemmTask& t = b.t[i];
    t.A = x; t.B = w[i]; t.C = y + col_offset[i]; t.bias = bias ? bias[i] : nullptr;
    t.lda = ldx; t.ldb = in_f; t.ldc = ldy;
    t.M = rows; t.N = out_f[i]; t.K = in_f; t.taps = 1;
    t.amap = ft_rowmap_layout(rows, x_tm_B);
    t.cmap = ft_rowmap_layout(rows, y_tm_B);
    t.relu = relu;
  }
  return ft_launch_gemm_rows(&b, ntasks, false, (hipStream_t)stream);
}

int ft_linear_bwd_data(const float* dy, long lddy, const float* w, float* dx, long lddx, int rows, int in_f,
                       int out_f, int accumulate, int dy_tm_B, int dx_tm_B, int w_transposed, void* stream) {
  if (check_tm("linear_bwd_data", rows, dy_tm_B) || check_tm("linear_bwd_data", rows, dx_tm_B)) return FT_ERR_ARG;
  FtGemmBatch b;
  memset(&b, 0, sizeof(b));
  FtGemmTask& t = b.t[0];
  t.A = dy; t.B = w; t.C = dx;
  t.lda = lddy; t.ldb = w_transposed ? out_f : in_f; t.ldc = lddx;     // w^T [in_f][out_f]: both operands K-contiguous
  t.M = rows; t.N = in_f; t.K = out_f; t.taps = 1;
  t.amap = ft_rowmap_layout(rows, dy_tm_B);
  t.cmap = ft_rowmap_layout(rows, dx_tm_B);
  t.accumulate = accumulate;
  return ft_launch_gemm_rows(&b, 1, !w_transposed, (hipStream_t)stream);
}

// HighwayNetwork with the gate in the GEMM epilogue (ft_gemm.h: FtGemmBatch.hw_mode)
int ft_highway_fwd(const float* x, const float* w12i, const float* b1, const float* b2, float* out, float* x12, int rows,
                   int C, void* stream) {
  FT_REQUIRE(C > 0 && C % 32 == 0, "highway_fwd: the fused form needs C %% 32 == 0 (C = %d)", C);
  FT_REQUIRE(rows >= 0 && x && w12i && b1 && b2 && out, "highway_fwd: null operand");
  if (rows == 0) return FT_OK;
  FtGemmBatch b;
  memset(&b, 0, sizeof(b));
  FtGemmTask& t = b.t[0];
  t.A = x; t.B = w12i; t.C = out;
  t.lda = C; t.ldb = C; t.ldc = C;
  t.M = rows; t.N = 2 * C; t.K = C; t.taps = 1;
  t.amap = ft_rowmap_identity(rows);
  t.cmap = ft_rowmap_identity(rows);
  b.hw_mode = 1; b.hw_C = C; b.hw_x = x; b.hw_b1 = b1; b.hw_b2 = b2; b.hw_x12 = x12;
  return ft_launch_gemm_rows(&b, 1, false, (hipStream_t)stream);
}

int ft_highway_bwd_data(const float* d12, const float* w1, const float* w2, int w_transposed, float* dx, int rows, int C,
                        const float* below_x12, const float* below_x, float* below_d12, void* stream) {
  FT_REQUIRE(C > 0 && rows >= 0 && d12 && w1 && w2 && dx, "highway_bwd_data: bad arguments");
  FT_REQUIRE((below_x12 != nullptr) == (below_x != nullptr) && (below_x != nullptr) == (below_d12 != nullptr),
             "highway_bwd_data: below_x12 / below_x / below_d12 go together");
  if (rows == 0) return FT_OK;
  FtGemmBatch b;
  memset(&b, 0, sizeof(b));
  for (int i = 0; i < 2; ++i) {
    FtGemmTask& t = b.t[i];
    t.A = d12 + (long)i * C; t.B = i ? w2 : w1; t.C = dx;
    t.lda = 2 * C; t.ldb = C; t.ldc = C;
    t.M = rows; t.N = C; t.K = C; t.taps = 1;
    t.amap = ft_rowmap_identity(rows);
    t.cmap = ft_rowmap_identity(rows);
    t.accumulate = 1;
  }
  b.chain = 2;
  if (below_d12) {
    b.hw_mode = 2; b.hw_C = C; b.hw_x = below_x; b.hw_x12 = const_cast<float*>(below_x12); b.hw_d12 = below_d12;
  }
  return ft_launch_gemm_rows(&b, 2, !w_transposed, (hipStream_t)stream);
}

// dx (+)= sum_i dy_i * w_i : several Linear layers that read the same input (highway W1/W2, the two directions of a
// recurrence's input projection) hand their data gradients back in ONE chained launch
static void data_multi_batch(FtGemmBatch& b, int ntasks, const float* const* dy, long lddy, const float* const* w, float* dx,
                             long lddx, int rows, int in_f, int out_f, int accumulate, int dy_tm_B, int dx_tm_B,
                             int w_transposed) {
  memset(&b, 0, sizeof(b));
  for (int i = 0; i < ntasks; ++i) {
    FtGemmTask& t = b.t[i];
    t.A = dy ? dy[i] : nullptr; t.B = w ? w[i] : nullptr; t.C = dx;
    t.lda = lddy; t.ldb = w_transposed ? out_f : in_f; t.ldc = lddx;
    t.M = rows; t.N = in_f; t.K = out_f; t.taps = 1;
    t.amap = ft_rowmap_layout(rows, dy_tm_B);
    t.cmap = ft_rowmap_layout(rows, dx_tm_B);
    t.accumulate = accumulate;
  }
  b.chain = ntasks > 1 ? ntasks : 0;
}

int ft_linear_bwd_data_multi(int ntasks, const float* const* dy, long lddy, const float* const* w, float* dx, long lddx,
                             int rows, int in_f, int out_f, int accumulate, int dy_tm_B, int dx_tm_B, int w_transposed,
                             void* stream) {
  FT_REQUIRE(ntasks >= 1 && ntasks <= FT_MAX_TASKS, "linear_bwd_data_multi: ntasks %d out of range", ntasks);
  if (check_tm("linear_bwd_data_multi", rows, dy_tm_B) || check_tm("linear_bwd_data_multi", rows, dx_tm_B))
    return FT_ERR_ARG;
  FtGemmBatch b;
  data_multi_batch(b, ntasks, dy, lddy, w, dx, lddx, rows, in_f, out_f, accumulate, dy_tm_B, dx_tm_B, w_transposed);
  return ft_launch_gemm_rows(&b, ntasks, !w_transposed, (hipStream_t)stream);
}

// The same product with scratch for split-K: few output tiles (token-side rows) and a long contraction (the recurrent
// layers' 2 x 4H pre-activation gradients) leave most of the chip idle in one pass over K
size_t ft_linear_bwd_data_multi_workspace(int ntasks, int rows, int in_f, int out_f, void* stream) {
  if (ntasks < 1 || ntasks > FT_MAX_TASKS) return 0;
  FtGemmBatch b;
  data_multi_batch(b, ntasks, nullptr, out_f, nullptr, nullptr, in_f, rows, in_f, out_f, 0, 0, 0, 1);
  return ft_rows_ksplit_floats(b, ntasks, (hipStream_t)stream) * sizeof(float);
}
int ft_linear_bwd_data_multi_ws(int ntasks, const float* const* dy, long lddy, const float* const* w, float* dx, long lddx,
                                int rows, int in_f, int out_f, int accumulate, int dy_tm_B, int dx_tm_B, int w_transposed,
                                void* workspace, size_t workspace_bytes, void* stream) {
  FT_REQUIRE(ntasks >= 1 && ntasks <= FT_MAX_TASKS, "linear_bwd_data_multi: ntasks %d out of range", ntasks);
  if (check_tm("linear_bwd_data_multi", rows, dy_tm_B) || check_tm("linear_bwd_data_multi", rows, dx_tm_B))
    return FT_ERR_ARG;
  FtGemmBatch b;
  data_multi_batch(b, ntasks, dy, lddy, w, dx, lddx, rows, in_f, out_f, accumulate, dy_tm_B, dx_tm_B, w_transposed);
  if (workspace && w_transposed &&
      workspace_bytes >= ft_rows_ksplit_floats(b, ntasks, (hipStream_t)stream) * sizeof(float))
    b.ksplit_slab = static_cast<float*>(workspace);
  return ft_launch_gemm_rows(&b, ntasks, !w_transposed, (hipStream_t)stream);
}

static FtGemmTNTask linear_bw_task(const float* dy, long lddy, const float* x, long ldx, float* dw, int rows,
                                   int in_f, int out_f, int B, int T, int x_shift, int accumulate, int dy_tm,
                                   int x_tm) {
  FtGemmTNTask t;
  memset(&t, 0, sizeof(t));
  t.A = dy; t.B = x; t.dst = dw;
  t.lda = lddy; t.ldb = ldx;
  t.ldm = in_f; t.ldn = 1; t.ldj = 0;
  t.M = out_f; t.N = in_f; t.R = rows; t.taps = 1;
  if (x_shift == 0 && !dy_tm && !x_tm) {
    t.amap = ft_rowmap_identity(rows);
    t.bmap = ft_rowmap_identity(rows);
  } else {          // both maps share the logical (b, t) decomposition with Tlog = T
    int Tl = T > 0 ? T : 1;
    FtRowMap am = {Tl, dy_tm ? 1 : Tl, dy_tm ? B : 1, Tl, 0, 0};
    FtRowMap bm = {Tl, x_tm ? 1 : Tl, x_tm ? B : 1, Tl, x_shift, 0};
    t.amap = am;
    t.bmap = bm;
  }
  t.accumulate = accumulate;
  return t;
}

size_t ft_linear_bwd_weight_workspace(int rows, int in_f, int out_f) {
  FtGemmTNTask t = linear_bw_task(nullptr, out_f, nullptr, in_f, nullptr, rows, in_f, out_f, 1, rows, 0, 0, 0, 0);
  return ft_gemm_tn_workspace_floats(t) * sizeof(float);
}

int ft_linear_bwd_weight(const float* dy, long lddy, const float* x, long ldx, float* dw, int rows, int in_f,
                         int out_f, int B, int T, int x_shift, int accumulate, int dy_time_major, int x_time_major,
                         void* workspace, size_t workspace_bytes, void* stream) {
  FT_REQUIRE((x_shift == 0 && !dy_time_major && !x_time_major) || (long)B * T == rows,
             "linear_bwd_weight: rows != B*T with a row shift / time-major operand");
  FtGemmTNTask t = linear_bw_task(dy, lddy, x, ldx, dw, rows, in_f, out_f, B, T, x_shift, accumulate, dy_time_major,
                                  x_time_major);
  return ft_launch_gemm_tn(t, (float*)workspace, workspace_bytes / sizeof(float), (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// channels-last Conv1d
static void conv_fwd_task(FtGemmTask& t, const float* x, long ldx, const float* wp, float* y, long ldy, int B,
                          int T, int Cin, int Cout, int k, int Tout, int relu) {
  memset(&t, 0, sizeof(t));
  t.A = x; t.B = wp; t.C = y;
  t.lda = ldx; t.ldb = Cin; t.ldc = ldy; t.b_tap_stride = (long)Cout * Cin;
  t.M = B * Tout; t.N = Cout; t.K = Cin; t.taps = k;
  FtRowMap m = {Tout > 0 ? Tout : 1, T, 1, T, -(k / 2), 1};
  t.amap = m;
  t.relu = relu;
}

int ft_conv1d_fwd(const float* x, long ldx, const float* wp, const float* scale, const float* shift, float* y,
                  long ldy, int B, int T, int Cin, int Cout, int k, int Tout, int relu, int accumulate,
                  void* stream) {
  FT_REQUIRE(k >= 1 && Tout >= 0 && Tout <= T + 1, "conv1d_fwd: bad k/Tout");
  FtGemmBatch b;
  memset(&b, 0, sizeof(b));
  conv_fwd_task(b.t[0], x, ldx, wp, y, ldy, B, T, Cin, Cout, k, Tout, relu);
  b.t[0].scale = scale;
  b.t[0].shift = shift;
  b.t[0].accumulate = accumulate;
  return ft_launch_gemm_rows(&b, 1, false, (hipStream_t)stream);
}

int ft_conv_bank_fwd(const float* x, long ldx, const float* wp_all, const float* scale, const float* shift,
                     float* ybank, int B, int T, int Cin, int C, int K, int Tout, int relu, void* stream) {
  FT_REQUIRE(K >= 1 && K <= FT_MAX_TASKS, "conv_bank_fwd: K=%d unsupported (max %d)", K, FT_MAX_TASKS);
  FT_REQUIRE(Tout == T || Tout == T + 1, "conv_bank_fwd: Tout must be T or T+1");
  FtGemmBatch b;
  memset(&b, 0, sizeof(b));
  long woff = 0;
  for (int i = 0; i < K; ++i) {
    int k = i + 1;
    // member i (k taps) is task K - 1 - i: workgroups are dispatched in task order, and the members with the most taps
    // should start first (longest-processing-time order: the launch used to end on a tail of the K-tap member's tiles)
    FtGemmTask& t = b.t[K - 1 - i];
    conv_fwd_task(t, x, ldx, wp_all + woff, ybank + (long)i * C, (long)K * C, B, T, Cin, C, k, Tout, relu);
    t.scale = scale ? scale + (long)i * C : nullptr;
    t.shift = shift ? shift + (long)i * C : nullptr;
    woff += (long)k * C * Cin;
  }
  return ft_launch_gemm_rows(&b, K, false, (hipStream_t)stream);
}

// conv (+ReLU) that also leaves the BatchNorm statistics partials of its output behind: from the GEMM's own epilogue
// when the launch takes the 128x128 split kernel, by the stand-alone column pass otherwise

// FT_BN_FUSED_STATS=0: always the stand-alone statistics pass (A/B aid)
static bool stats_in_epilogue() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("FT_BN_FUSED_STATS");
    v = (e && e[0] == '0') ? 0 : 1;
  }
  return v == 1;
}

// Token-side convolutions (B*T = 4096 rows here) have too few 128x128 output tiles to fill the chip and used to run on
// the 64x64 f32 kernel at 45-70 TF (prenet projection, 4096 x 256 x (3 x 4096): 376 us).  With k >= 2 taps the taps are
// INDEPENDENT tasks of one launch instead -- tiles x k >= 192 workgroups on the pipelined split kernel, each writing its
// own fp32 slab -- followed by one ordered sum (+ReLU) over the k slabs: deterministic, the taps are merely added in tap
// order after instead of inside the accumulator.  FT_CONV_TAP_SPLIT=0 disables.
static bool conv_tap_split(int B, int Tout, int Cout, int k) {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("FT_CONV_TAP_SPLIT");
    on = (e && e[0] == '0') ? 0 : 1;
  }
  const long tiles = (long)ft_cdiv((long)B * Tout, 128) * ft_cdiv(Cout, 128);
  return on && k >= 2 && k <= FT_MAX_TASKS && Cout > 64 && Cout % 4 == 0 && (long)B * Tout > 64 && tiles < 192 &&
         tiles * k >= 192;
}
static size_t conv_stats_bytes_aligned(int B, int Tout, int Cout) {
  return (ft_conv_stats_workspace(B, Tout, Cout) + 255) & ~(size_t)255;
}

__global__ __launch_bounds__(256) void ft_tap_slab_sum4_kernel(const float4* __restrict__ slab, float4* __restrict__ y,
                                                               long total4, int S, int relu) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  float4 a = slab[i];
  for (int s = 1; s < S; ++s) {
    const float4 v = slab[(long)s * total4 + i];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  if (relu) {
    a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
  }
  y[i] = a;
}

size_t ft_conv1d_fwd_stats_workspace(int B, int Tout, int Cout, int k) {
  size_t n = conv_stats_bytes_aligned(B, Tout, Cout);
  if (conv_tap_split(B, Tout, Cout, k)) n += (size_t)k * B * Tout * Cout * sizeof(float);
  return n;
}

int ft_conv1d_fwd_stats(const float* x, long ldx, const float* wp, float* y, long ldy, int B, int T, int Cin, int Cout,
                        int k, int Tout, int relu, double* partial, size_t partial_bytes, int* nchunks,
                        void* stream) {
  FT_REQUIRE(k >= 1 && Tout >= 1 && Tout <= T + 1 && nchunks && partial, "conv1d_fwd_stats: bad arguments");
  FT_REQUIRE(ldy == Cout, "conv1d_fwd_stats: y must be contiguous [B,Tout,Cout]");
  FT_REQUIRE(partial_bytes >= ft_conv_stats_workspace(B, Tout, Cout), "conv1d_fwd_stats: partial buffer too small");
  FtGemmBatch b;
  memset(&b, 0, sizeof(b));
  if (conv_tap_split(B, Tout, Cout, k) && partial_bytes >= ft_conv1d_fwd_stats_workspace(B, Tout, Cout, k) &&
      ((uintptr_t)y % 16 == 0)) {
    float* slab = reinterpret_cast<float*>(reinterpret_cast<char*>(partial) + conv_stats_bytes_aligned(B, Tout, Cout));
    const long mn = (long)B * Tout * Cout;
    for (int j = 0; j < k; ++j) {
      FtGemmTask& t = b.t[j];
      conv_fwd_task(t, x, ldx, wp + (long)j * Cout * Cin, slab + (long)j * mn, Cout, B, T, Cin, Cout, k, Tout, 0);
      t.taps = 1;
      t.amap.shift0 = -(k / 2) + j;
      t.amap.shift_step = 0;
    }
    int rc = ft_launch_gemm_rows(&b, k, false, (hipStream_t)stream);
    if (rc) return rc;
    hipLaunchKernelGGL(ft_tap_slab_sum4_kernel, dim3(ft_cdiv(mn / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)slab, (float4*)y, mn / 4, k, relu);
    *nchunks = ft_bn_stat_partials(y, B, Tout, Cout, 0, partial, (hipStream_t)stream);
    return ft_check_launch("conv1d_fwd_stats (tap split)");
  }
  conv_fwd_task(b.t[0], x, ldx, wp, y, ldy, B, T, Cin, Cout, k, Tout, relu);
  b.t[0].stat = stats_in_epilogue() ? partial : nullptr;
  b.t[0].stat_ld = Cout;
  b.t[0].stat_col0 = 0;
  b.t[0].stat_tvalid = Tout;
  int rc = ft_launch_gemm_rows(&b, 1, false, (hipStream_t)stream);
  if (rc) return rc;
  if (b.stat_fused && stats_in_epilogue()) {
    *nchunks = ft_cdiv((long)B * Tout, 128);
    return FT_OK;
  }
  *nchunks = ft_bn_stat_partials(y, B, Tout, Cout, 0, partial, (hipStream_t)stream);
  return ft_check_launch("conv1d_fwd_stats");
}

int ft_conv_bank_fwd_stats(const float* x, long ldx, const float* wp_all, float* ybank, int B, int T, int Cin, int C,
                           int K, int relu, double* partial, size_t partial_bytes, int* nchunks, void* stream) {
  FT_REQUIRE(K >= 1 && K <= FT_MAX_TASKS && nchunks && partial, "conv_bank_fwd_stats: bad arguments");
  const int Tout = T + 1;           // the training-mode bank buffer: even-k members own T+1 rows (common_layers.py:97-99)
  FT_REQUIRE(partial_bytes >= ft_conv_stats_workspace(B, Tout, K * C), "conv_bank_fwd_stats: partial buffer too small");
  FtGemmBatch b;
  memset(&b, 0, sizeof(b));
  long woff = 0;
  for (int i = 0; i < K; ++i) {
    int k = i + 1;
    FtGemmTask& t = b.t[K - 1 - i];                   // most taps first (see ft_conv_bank_fwd)
    conv_fwd_task(t, x, ldx, wp_all + woff, ybank + (long)i * C, (long)K * C, B, T, Cin, C, k, Tout, relu);
    t.stat = stats_in_epilogue() ? partial : nullptr;
    t.stat_ld = K * C;
    t.stat_col0 = i * C;
    t.stat_tvalid = (k & 1) ? T : T + 1;              // BatchNorm of an odd-k member sees T rows
    woff += (long)k * C * Cin;
  }
  int rc = ft_launch_gemm_rows(&b, K, false, (hipStream_t)stream);
  if (rc) return rc;
  if (b.stat_fused && stats_in_epilogue()) {
    *nchunks = ft_cdiv((long)B * Tout, 128);
    return FT_OK;
  }
  *nchunks = ft_bn_stat_partials(ybank, B, Tout, K * C, C, partial, (hipStream_t)stream);
  return ft_check_launch("conv_bank_fwd_stats");
}

static int conv1d_bwd_data_impl(const float* dy, long lddy, const float* wp, float* dx, long lddx, int B, int T, int Cin,
                                int Cout, int k, int Tbuf, int Tvalid, int accumulate, int wp_transposed,
                                const float* relu_mask, void* stream);

int ft_conv1d_bwd_data(const float* dy, long lddy, const float* wp, float* dx, long lddx, int B, int T, int Cin,
                       int Cout, int k, int Tbuf, int Tvalid, int accumulate, int wp_transposed, void* stream) {
  return conv1d_bwd_data_impl(dy, lddy, wp, dx, lddx, B, T, Cin, Cout, k, Tbuf, Tvalid, accumulate, wp_transposed, nullptr,
                              stream);
}

int ft_conv1d_bwd_data_relu(const float* dy, long lddy, const float* wp, const float* y, float* dx, long lddx, int B,
                            int T, int Cin, int Cout, int k, int wp_transposed, void* stream) {
  FT_REQUIRE(y != nullptr, "conv1d_bwd_data_relu: null mask source");
  return conv1d_bwd_data_impl(dy, lddy, wp, dx, lddx, B, T, Cin, Cout, k, T, T, 0, wp_transposed, y, stream);
}

static int conv1d_bwd_data_impl(const float* dy, long lddy, const float* wp, float* dx, long lddx, int B, int T, int Cin,
                                int Cout, int k, int Tbuf, int Tvalid, int accumulate, int wp_transposed,
                                const float* relu_mask, void* stream) {
  FT_REQUIRE(k >= 1 && Tvalid <= Tbuf, "conv1d_bwd_data: bad k/Tvalid");
  FtGemmBatch b;
  memset(&b, 0, sizeof(b));
  b.relu_mask = relu_mask;
  FtGemmTask& t = b.t[0];
  t.A = dy; t.B = wp; t.C = dx;
  t.lda = lddy; t.ldb = wp_transposed ? Cout : Cin; t.ldc = lddx; t.b_tap_stride = (long)Cout * Cin;
  t.M = B * T; t.N = Cin; t.K = Cout; t.taps = k;
  FtRowMap m = {T > 0 ? T : 1, Tbuf, 1, Tvalid, k / 2, -1};     // dy row = t - tap + pad
  t.amap = m;
  t.accumulate = accumulate;
  return ft_launch_gemm_rows(&b, 1, !wp_transposed, (hipStream_t)stream);
}

// data gradient of the whole conv bank in ONE launch: dx[b,t,:] = sum_k sum_tap dy_k[b, t - tap + k/2, :] * W_k,tap
// (member k's rows t >= Tvalid_k of dy are not part of its output).  Chained: the K tasks are accumulated in registers.
// With few output tiles (prenet: 4096 x 256 = 64 tiles of 128x128) a chain would leave 3/4 of the CUs idle, so there the
// members run as K independent tasks into per-member partials and an ordered sum follows.
static bool bank_bwd_partials(int B, int T, int Cin, int K) {
  const long tiles = (long)ft_cdiv((long)B * T, 128) * ft_cdiv(Cin, 128);
  return K >= 2 && Cin > 64 && (long)B * T > 64 && tiles < 192 && tiles * K >= 192;
}

size_t ft_conv_bank_bwd_data_workspace(int B, int T, int Cin, int K) {
  return bank_bwd_partials(B, T, Cin, K) ? (size_t)K * B * T * Cin * sizeof(float) : 0;
}

int ft_conv_bank_bwd_data(const float* dy, long lddy, const float* wp_all, float* dx, long lddx, int B, int T, int Cin,
                          int C, int K, int Tbuf, int wp_transposed, void* workspace, size_t workspace_bytes,
                          void* stream) {
  FT_REQUIRE(K >= 1 && K <= FT_MAX_TASKS, "conv_bank_bwd_data: K=%d unsupported (max %d)", K, FT_MAX_TASKS);
  FT_REQUIRE(Tbuf == T || Tbuf == T + 1, "conv_bank_bwd_data: Tbuf must be T or T+1");
  const bool partials = bank_bwd_partials(B, T, Cin, K);
  if (partials)
    FT_REQUIRE(workspace && workspace_bytes >= ft_conv_bank_bwd_data_workspace(B, T, Cin, K),
               "conv_bank_bwd_data: workspace too small");
  float* part = static_cast<float*>(workspace);
  FtGemmBatch b;
  memset(&b, 0, sizeof(b));
  long woff = 0;
  for (int i = 0; i < K; ++i) {
    const int k = i + 1;
    FtGemmTask& t = b.t[i];
    t.A = dy + (long)i * C; t.B = wp_all + woff;
    t.C = partials ? part + (long)i * B * T * Cin : dx;
    t.lda = lddy; t.ldb = wp_transposed ? C : Cin; t.ldc = partials ? Cin : lddx; t.b_tap_stride = (long)C * Cin;
    t.M = B * T; t.N = Cin; t.K = C; t.taps = k;
    const int Tvalid = (k % 2 == 0) ? Tbuf : T;            // even kernels produce T+1 rows when the buffer has them
    FtRowMap m = {T > 0 ? T : 1, Tbuf, 1, Tvalid < Tbuf ? Tvalid : Tbuf, k / 2, -1};
    t.amap = m;
    woff += (long)k * C * Cin;
  }
  b.chain = partials ? 0 : K;
  if (partials && wp_transposed) {
    // better than per-member partials (member k does k taps: sixteen-fold imbalance between the tasks of one launch):
    // ONE chained product whose stage sequence is cut into equal ranges (split-K, FtGemmBatch.ksplit_slab)
    FtGemmBatch c = b;
    for (int i = 0; i < K; ++i) {
      c.t[i].C = dx;
      c.t[i].ldc = lddx;
    }
    c.chain = K;
    const size_t need = ft_rows_ksplit_floats(c, K, (hipStream_t)stream) * sizeof(float);
    if (need > 0 && need <= workspace_bytes) {
      c.ksplit_slab = part;
      return ft_launch_gemm_rows(&c, K, false, (hipStream_t)stream);
    }
  }
  int rc = ft_launch_gemm_rows(&b, K, !wp_transposed, (hipStream_t)stream);
  if (rc || !partials) return rc;
  return ft_launch_slab_sum(part, dx, B * T, Cin, K, lddx, (hipStream_t)stream);
}

static FtGemmTNTask conv_bw_task(const float* dy, long lddy, const float* x, long ldx, float* dw, int B, int T,
                                 int Cin, int Cout, int k, int Tbuf, int Tvalid) {
  FtGemmTNTask t;
  memset(&t, 0, sizeof(t));
  t.A = dy; t.B = x; t.dst = dw;
  t.lda = lddy; t.ldb = ldx;
  t.ldm = (long)Cin * k; t.ldn = k; t.ldj = 1;       // torch layout [Cout][Cin][k]
  t.M = Cout; t.N = Cin; t.R = B * Tvalid; t.taps = k;
  int Tl = Tvalid > 0 ? Tvalid : 1;
  FtRowMap am = {Tl, Tbuf, 1, Tvalid, 0, 0};
  FtRowMap bm = {Tl, T, 1, T, -(k / 2), 1};
  t.amap = am;
  t.bmap = bm;
  return t;
}

// weight gradients of ALL members of a conv bank in one launch (FtGemmTNTask conv-bank mode): dy [B,Tbuf,K*C] is the
// gradient of the bank buffer, x [B,T,Cin] the bank's input, dw[i] the torch-layout [C][Cin][i+1] gradient of member i
static FtGemmTNTask bank_bw_task(const float* dy, long lddy, const float* x, long ldx, int B, int T, int Cin, int C,
                                 int K, int Tbuf) {
  FtGemmTNTask t;
  memset(&t, 0, sizeof(t));
  t.A = dy; t.B = x;
  t.lda = lddy; t.ldb = ldx;
  t.M = K * C; t.N = Cin; t.R = B * Tbuf; t.taps = K;
  FtRowMap am = {Tbuf, Tbuf, 1, Tbuf, 0, 0};
  FtRowMap bm = {Tbuf, T, 1, T, 0, 1};
  t.amap = am;
  t.bmap = bm;
  t.bankC = C;
  t.bankTodd = T;
  return t;
}

size_t ft_conv_bank_bwd_weight_workspace(int B, int T, int Cin, int C, int K, int Tbuf) {
  FtGemmTNTask t = bank_bw_task(nullptr, (long)K * C, nullptr, Cin, B, T, Cin, C, K, Tbuf);
  return ft_gemm_tn_workspace_floats(t) * sizeof(float);
}

int ft_conv_bank_bwd_weight(const float* dy, long lddy, const float* x, long ldx, float* const* dw, int B, int T,
                            int Cin, int C, int K, int Tbuf, void* workspace, size_t workspace_bytes, void* stream) {
  FT_REQUIRE(K >= 1 && K <= FT_MAX_TASKS, "conv_bank_bwd_weight: K=%d unsupported (max %d)", K, FT_MAX_TASKS);
  FT_REQUIRE(Tbuf == T || Tbuf == T + 1, "conv_bank_bwd_weight: Tbuf must be T or T+1");
  FT_REQUIRE(C % 128 == 0, "conv_bank_bwd_weight: C must be a multiple of 128 (use ft_conv1d_bwd_weight per member)");
  FtGemmTNTask t = bank_bw_task(dy, lddy, x, ldx, B, T, Cin, C, K, Tbuf);
  for (int i = 0; i < K; ++i) {
    FT_REQUIRE(dw[i] != nullptr, "conv_bank_bwd_weight: null dw[%d]", i);
    t.bank_dst[i] = dw[i];
  }
  return ft_launch_gemm_tn(t, (float*)workspace, workspace_bytes / sizeof(float), (hipStream_t)stream);
}

size_t ft_conv1d_bwd_weight_workspace(int B, int T, int Cin, int Cout, int k, int Tvalid) {
  FtGemmTNTask t = conv_bw_task(nullptr, Cout, nullptr, Cin, nullptr, B, T, Cin, Cout, k, Tvalid, Tvalid);
  return ft_gemm_tn_workspace_floats(t) * sizeof(float);
}

int ft_conv1d_bwd_weight(const float* dy, long lddy, const float* x, long ldx, float* dw, int B, int T, int Cin,
                         int Cout, int k, int Tbuf, int Tvalid, void* workspace, size_t workspace_bytes,
                         void* stream) {
  FT_REQUIRE(k >= 1 && Tvalid <= Tbuf, "conv1d_bwd_weight: bad k/Tvalid");
  FtGemmTNTask t = conv_bw_task(dy, lddy, x, ldx, dw, B, T, Cin, Cout, k, Tbuf, Tvalid);
  return ft_launch_gemm_tn(t, (float*)workspace, workspace_bytes / sizeof(float), (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// Conv1d WITH bias (FFTBlock conv1/conv2, common_layers.py:161-164)
int ft_conv1d_bias_fwd(const float* x, long ldx, const float* wp, const float* bias, float* y, long ldy, int B, int T,
                       int Cin, int Cout, int k, int relu, void* stream) {
  FT_REQUIRE(k >= 1 && (k % 2) == 1, "conv1d_bias_fwd: odd kernel sizes only");
  FtGemmBatch b;
  memset(&b, 0, sizeof(b));
  conv_fwd_task(b.t[0], x, ldx, wp, y, ldy, B, T, Cin, Cout, k, T, relu);
  b.t[0].bias = bias;
  return ft_launch_gemm_rows(&b, 1, false, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// strided-batch GEMMs: instance z -> (z0, z1) = (z / nb1, z % nb1), X_z = X + z0*sX0 + z1*sX1
static void set_batch(FtGemmTask& t, int nb0, int nb1, long sA0, long sA1, long sB0, long sB1, long sC0, long sC1) {
  t.nz = nb0 * nb1; t.nz1 = nb1;
  t.sA0 = sA0; t.sA1 = sA1; t.sB0 = sB0; t.sB1 = sB1; t.sC0 = sC0; t.sC1 = sC1;
}

int ft_bgemm_nt(const float* A, long lda, long sA0, long sA1, const float* Bm, long ldb, long sB0, long sB1, float* C,
                long ldc, long sC0, long sC1, int M, int N, int K, int nb0, int nb1, void* stream) {
  FT_REQUIRE(nb0 >= 1 && nb1 >= 1, "bgemm_nt: bad batch");
  FtGemmBatch b;
  memset(&b, 0, sizeof(b));
  FtGemmTask& t = b.t[0];
  t.A = A; t.B = Bm; t.C = C;
  t.lda = lda; t.ldb = ldb; t.ldc = ldc;
  t.M = M; t.N = N; t.K = K; t.taps = 1;
  t.amap = ft_rowmap_identity(M);
  set_batch(t, nb0, nb1, sA0, sA1, sB0, sB1, sC0, sC1);
  return ft_launch_gemm_rows(&b, 1, false, (hipStream_t)stream);
}

int ft_bgemm_nn(const float* A, long lda, long sA0, long sA1, const float* Bm, long ldb, long sB0, long sB1, float* C,
                long ldc, long sC0, long sC1, int M, int N, int K, int nb0, int nb1, int a_rows_padded, void* stream) {
  FT_REQUIRE(nb0 >= 1 && nb1 >= 1, "bgemm_nn: bad batch");
  FtGemmBatch b;
  memset(&b, 0, sizeof(b));
  FtGemmTask& t = b.t[0];
  t.A = A; t.B = Bm; t.C = C;
  t.lda = lda; t.ldb = ldb; t.ldc = ldc;
  t.M = M; t.N = N; t.K = K; t.taps = 1;
  t.amap = ft_rowmap_identity(M);
  t.a_rowpad = a_rows_padded;
  set_batch(t, nb0, nb1, sA0, sA1, sB0, sB1, sC0, sC1);
  return ft_launch_gemm_rows(&b, 1, true, (hipStream_t)stream);
}

static FtGemmTNTask bgemm_tn_task(const float* A, long lda, long sA0, long sA1, const float* Bm, long ldb, long sB0,
                                  long sB1, float* C, long ldc, long sC0, long sC1, int M, int N, int R, int nb0,
                                  int nb1) {
  FtGemmTNTask t;
  memset(&t, 0, sizeof(t));
  t.A = A; t.B = Bm; t.dst = C;
  t.lda = lda; t.ldb = ldb;
  t.ldm = ldc; t.ldn = 1; t.ldj = 0;
  t.M = M; t.N = N; t.R = R; t.taps = 1;
  t.amap = ft_rowmap_identity(R);
  t.bmap = ft_rowmap_identity(R);
  t.nz = nb0 * nb1; t.nz1 = nb1;
  t.sA0 = sA0; t.sA1 = sA1; t.sB0 = sB0; t.sB1 = sB1; t.sD0 = sC0; t.sD1 = sC1;
  return t;
}

size_t ft_bgemm_tn_workspace(int M, int N, int R, int nb0, int nb1) {
  FtGemmTNTask t = bgemm_tn_task(nullptr, M, 0, 0, nullptr, N, 0, 0, nullptr, N, 0, 0, M, N, R, nb0, nb1);
  return ft_gemm_tn_workspace_floats(t) * sizeof(float);
}

int ft_bgemm_tn(const float* A, long lda, long sA0, long sA1, const float* Bm, long ldb, long sB0, long sB1, float* C,
                long ldc, long sC0, long sC1, int M, int N, int R, int nb0, int nb1, int rows_padded, void* workspace,
                size_t workspace_bytes, void* stream) {
  FT_REQUIRE(nb0 >= 1 && nb1 >= 1, "bgemm_tn: bad batch");
  FtGemmTNTask t = bgemm_tn_task(A, lda, sA0, sA1, Bm, ldb, sB0, sB1, C, ldc, sC0, sC1, M, N, R, nb0, nb1);
  t.rowpad = rows_padded;
  return ft_launch_gemm_tn(t, (float*)workspace, workspace_bytes / sizeof(float), (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
int ft_lr_scan(float* dur, int B, int Tx, int* cum, int* total, void* stream) {
  return ft_lr_scan_impl(dur, B, Tx, cum, total, (hipStream_t)stream);
}
int ft_lr_expand(const float* x, const int* cum, float* y, int* src_idx, int B, int Tx, int Tm, int C,
                 void* stream) {
  return ft_lr_expand_impl(x, cum, y, src_idx, B, Tx, Tm, C, (hipStream_t)stream);
}
int ft_lr_bwd(const float* dy, const int* cum, float* dx, int B, int Tx, int Tm, int C, void* stream) {
  return ft_lr_bwd_impl(dy, cum, dx, B, Tx, Tm, C, (hipStream_t)stream);
}
int ft_lr_expand_tm(const float* x, const int* cum, const float* pad_row, float* y, int B, int Tx, int Tm, int C,
                    void* stream) {
  return ft_lr_expand_impl(x, cum, y, nullptr, B, Tx, Tm, C, (hipStream_t)stream, 1, pad_row);
}
int ft_lr_bwd_tm(const float* dy, const int* cum, float* dx, float* dtail, int B, int Tx, int Tm, int C, void* stream) {
  return ft_lr_bwd_impl(dy, cum, dx, B, Tx, Tm, C, (hipStream_t)stream, 1, dtail);
}

}  // extern "C"
