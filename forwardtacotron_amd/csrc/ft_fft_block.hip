// Whole FFTBlocks per C call (include/fwdtaco_hip.h: ft_fft_blocks_fwd / ft_fft_blocks_bwd): host-side sequencing of the
// library's own entry points -- no kernel lives here.  What it removes is the interpreter between the launches: a
// FastPitch train step is ~900 launches for ~13 ms of GPU work, and issued one ctypes call at a time (tensor
// allocation, argument marshalling, autograd bookkeeping: ~15 us each) it was bound by the host at 15 ms.
#include <mutex>

#include <stdlib.h>

#include "ft_common.h"
#include "fwdtaco_hip.h"

namespace {

#define FT_TRY(call)            \
  do {                          \
    const int rc_ = (call);     \
    if (rc_ != FT_OK) return rc_; \
  } while (0)

// one fork event per device: record on the main stream, wait on the weight-gradient stream (a wait captures the record
// that precedes it, so the event can be re-recorded by the next call at once)
hipEvent_t fork_event() {
  static std::mutex mu;
  static hipEvent_t ev[16] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
  std::lock_guard<std::mutex> lk(mu);
  if (!ev[dev] && hipEventCreateWithFlags(&ev[dev], hipEventDisableTiming) != hipSuccess) ev[dev] = nullptr;
  return ev[dev];
}

int block_fwd(const FtFFTBlock& b, void* st) {
  const int rows = b.B * b.T, d = b.d;
  const int hd = d / b.nheads;
  const float scale = 1.0f / sqrtf((float)hd);
  FT_TRY(ft_linear_fwd(b.x, d, b.in_w, b.in_b, b.qkv, 3 * d, rows, d, 3 * d, 0, 0, 0, 0, st));
  FT_TRY(ft_attn_fwd(b.qkv, b.key_pad, b.att, b.lse2, b.B, b.T, b.nheads, hd, scale, b.p_drop, b.seed_attn, st));
  FT_TRY(ft_linear_fwd(b.att, d, b.out_w, b.out_b, b.sa, d, rows, d, d, 0, 0, 0, 0, st));
  FT_TRY(ft_layernorm_fwd(b.x, b.sa, b.n1_g, b.n1_b, b.s1, b.y1, b.mean1, b.rstd1, rows, d, b.eps1, b.p_drop, b.seed_ln1, st));
  FT_TRY(ft_conv1d_bias_fwd(b.y1, d, b.c1_wp, b.c1_b, b.h1, b.dfft, b.B, b.T, d, b.dfft, b.k1, 1, st));
  FT_TRY(ft_conv1d_bias_fwd(b.h1, b.dfft, b.c2_wp, b.c2_b, b.h2, d, b.B, b.T, b.dfft, d, b.k2, 0, st));
  FT_TRY(ft_layernorm_fwd(b.y1, b.h2, b.n2_g, b.n2_b, b.s2, b.y2, b.mean2, b.rstd2, rows, d, b.eps2, b.p_drop, b.seed_ln2, st));
  return FT_OK;
}

// the data path of one block's backward, on the main stream
int block_bwd_data(const FtFFTBlock& b, const FtFFTBlockGrads& g, void* ws, size_t ws_bytes, void* st) {
  const int rows = b.B * b.T, d = b.d, f = b.dfft;
  const int hd = d / b.nheads;
  const float scale = 1.0f / sqrtf((float)hd);
  // y2 = LN2(y1 + drop(h2)): d_y1 (residual path), d_h2 (own buffer: conv1's data gradient is accumulated into d_y1 while
  // conv2's weight gradient still reads d_h2 on the other stream)
  FT_TRY(ft_layernorm_bwd(g.dy2, b.s2, b.n2_g, b.mean2, b.rstd2, g.d_y1, g.t2, g.d_h2, rows, d, b.p_drop, b.seed_ln2, st));
  // conv2's data gradient through conv1's ReLU: the mask is the GEMM's epilogue (g.d_h1 is not used any more)
  FT_TRY(ft_conv1d_bwd_data_relu(g.d_h2, d, b.c2_wpt, b.h1, g.g_h1, f, b.B, b.T, f, d, b.k2, 1, st));
  FT_TRY(ft_conv1d_bwd_data(g.g_h1, f, b.c1_wpt, g.d_y1, d, b.B, b.T, d, f, b.k1, b.T, b.T, 1, 1, st));     // += residual path
  FT_TRY(ft_layernorm_bwd(g.d_y1, b.s1, b.n1_g, b.mean1, b.rstd1, g.d_h, g.t1, g.d_sa, rows, d, b.p_drop, b.seed_ln1, st));
  FT_TRY(ft_linear_bwd_data(g.d_sa, d, b.out_wT, g.datt, d, rows, d, d, 0, 0, 0, 1, st));
  FT_TRY(ft_attn_bwd(b.qkv, b.att, g.datt, b.key_pad, b.lse2, g.dqkv, b.B, b.T, b.nheads, hd, scale, b.p_drop, b.seed_attn,
                     ws, ws_bytes, st));
  // d(x) = d_h (residual path) + dqkv W_in
  FT_TRY(ft_linear_bwd_data(g.dqkv, 3 * d, b.in_wT, g.d_h, d, rows, d, 3 * d, 1, 0, 0, 1, st));
  if (g.dx != g.d_h) {
    FT_REQUIRE(false, "fft_blocks_bwd: dx must alias d_h (the in-projection's data gradient accumulates into it)");
  }
  return FT_OK;
}

// the weight gradients of one block (weight-gradient stream: four split-K GEMMs)
int block_bwd_weights(const FtFFTBlock& b, const FtFFTBlockGrads& g, void* ws, size_t wsb, void* st) {
  const int rows = b.B * b.T, d = b.d, f = b.dfft;
  FT_TRY(ft_conv1d_bwd_weight(g.d_h2, d, b.h1, f, g.g_c2_w, b.B, b.T, f, d, b.k2, b.T, b.T, ws, wsb, st));
  FT_TRY(ft_conv1d_bwd_weight(g.g_h1, f, b.y1, d, g.g_c1_w, b.B, b.T, d, f, b.k1, b.T, b.T, ws, wsb, st));
  FT_TRY(ft_linear_bwd_weight(g.d_sa, d, b.att, d, g.g_out_w, rows, d, d, 1, rows, 0, 0, 0, 0, ws, wsb, st));
  FT_TRY(ft_linear_bwd_weight(g.dqkv, 3 * d, b.x, d, g.g_in_w, rows, d, 3 * d, 1, rows, 0, 0, 0, 0, ws, wsb, st));
  return FT_OK;
}

// its bias and LayerNorm gradients: six column sums (twelve small launches) -- on a stream of their own, or they trail
// behind the weight-gradient GEMMs at the end of the step (1.9 ms of a 15 ms FastPitch step, lab/steady_segments_fp.py)
int block_bwd_sums(const FtFFTBlock& b, const FtFFTBlockGrads& g, void* ws, size_t wsb, void* st) {
  const int rows = b.B * b.T, d = b.d, f = b.dfft;
  FT_TRY(ft_colsum2(g.t2, g.dy2, d, g.g_n2_g, g.g_n2_b, rows, d, ws, wsb, st));
  FT_TRY(ft_colsum(g.d_h2, d, g.g_c2_b, rows, d, 1.0f, 0, ws, wsb, st));
  FT_TRY(ft_colsum(g.g_h1, f, g.g_c1_b, rows, f, 1.0f, 0, ws, wsb, st));
  FT_TRY(ft_colsum2(g.t1, g.d_y1, d, g.g_n1_g, g.g_n1_b, rows, d, ws, wsb, st));
  FT_TRY(ft_colsum(g.d_sa, d, g.g_out_b, rows, d, 1.0f, 0, ws, wsb, st));
  FT_TRY(ft_colsum(g.dqkv, 3 * d, g.g_in_b, rows, 3 * d, 1.0f, 0, ws, wsb, st));
  return FT_OK;
}

// the same eight sums as block_bwd_sums in ONE pair of launches (ft_colsum_batch), issued once the block's data path is done
int block_bwd_sums_batched(const FtFFTBlock& b, const FtFFTBlockGrads& g, void* ws, size_t wsb, void* st) {
  const int rows = b.B * b.T, d = b.d, f = b.dfft;
  const float* x[8] = {g.t2, g.dy2, g.d_h2, g.g_h1, g.t1, g.d_y1, g.d_sa, g.dqkv};
  const long ld[8] = {d, d, d, f, d, d, d, 3L * d};
  float* out[8] = {g.g_n2_g, g.g_n2_b, g.g_c2_b, g.g_c1_b, g.g_n1_g, g.g_n1_b, g.g_out_b, g.g_in_b};
  const int C[8] = {d, d, d, f, d, d, d, 3 * d};
  return ft_colsum_batch(8, x, ld, out, C, rows, ws, wsb, st);
}

}  // namespace

extern "C" {

int ft_fft_blocks_fwd(const FtFFTBlock* blocks, int n, void* stream) {
  FT_REQUIRE(blocks != nullptr && n >= 0, "fft_blocks_fwd: bad arguments");
  for (int i = 0; i < n; ++i) {
    const FtFFTBlock& b = blocks[i];
    FT_REQUIRE(b.d % b.nheads == 0 && (b.d / b.nheads == 64 || b.d / b.nheads == 128), "fft_blocks_fwd: head_dim 64 or 128");
    FT_TRY(block_fwd(b, stream));
  }
  return FT_OK;
}

size_t ft_fft_block_wgrad_workspace(int B, int T, int d, int dfft, int k1, int k2) {
  const int rows = B * T;
  size_t m = ft_conv1d_bwd_weight_workspace(B, T, d, dfft, k1, T);
  auto up = [&](size_t v) { if (v > m) m = v; };
  up(ft_conv1d_bwd_weight_workspace(B, T, dfft, d, k2, T));
  up(ft_linear_bwd_weight_workspace(rows, d, d));
  up(ft_linear_bwd_weight_workspace(rows, d, 3 * d));
  return m;
}

size_t ft_fft_block_sums_workspace(int B, int T, int d, int dfft) {
  const int rows = B * T;
  {
    const int C[8] = {d, d, d, dfft, d, d, d, 3 * d};
    const size_t batched = ft_colsum_batch_workspace(8, C, rows);
    if (batched > 0) return batched;            // (covers the per-sum calls too: it is their sum)
  }
  size_t m = ft_colsum_workspace(rows, 3 * d);
  auto up = [&](size_t v) { if (v > m) m = v; };
  up(ft_colsum_workspace(rows, dfft));
  up(2 * ft_colsum_workspace(rows, d));
  return m;
}

int ft_fft_blocks_bwd(const FtFFTBlock* blocks, const FtFFTBlockGrads* grads, int n, void* workspace,
                      size_t workspace_bytes, void* wgrad_workspace, size_t wgrad_workspace_bytes, void* sums_workspace,
                      size_t sums_workspace_bytes, void* stream, void* wgrad_stream, void* sums_stream) {
  FT_REQUIRE(blocks != nullptr && grads != nullptr && n >= 0, "fft_blocks_bwd: bad arguments");
  if (!wgrad_stream) wgrad_stream = stream;
  if (!sums_stream) sums_stream = wgrad_stream;
  FT_REQUIRE(sums_stream == wgrad_stream ? (sums_workspace == nullptr || sums_workspace == wgrad_workspace ||
                                            (char*)sums_workspace >= (char*)wgrad_workspace + wgrad_workspace_bytes ||
                                            (char*)sums_workspace + sums_workspace_bytes <= (char*)wgrad_workspace)
                                         : sums_workspace != wgrad_workspace,
             "fft_blocks_bwd: two streams must not share one workspace");
  const bool fork = wgrad_stream != stream || sums_stream != stream;
  hipEvent_t ev = fork ? fork_event() : nullptr;
  FT_REQUIRE(!fork || ev != nullptr, "fft_blocks_bwd: could not create the fork event");
  if (!sums_workspace) {
    sums_workspace = wgrad_workspace;
    sums_workspace_bytes = wgrad_workspace_bytes;
    sums_stream = wgrad_stream;               // one workspace: one stream
  }
  // the side streams follow the main stream's position: every weight gradient / column sum is issued as soon as the data
  // path has produced its operand (one fork per stage, four per block -- a fork per BLOCK left the weight-gradient stream
  // 1.9 ms behind the main stream at the end of a 15 ms step)
  auto follow = [&]() -> int {
    if (!fork) return FT_OK;
    bool ok = hipEventRecord(ev, (hipStream_t)stream) == hipSuccess;
    if (wgrad_stream != stream) ok = ok && hipStreamWaitEvent((hipStream_t)wgrad_stream, ev, 0) == hipSuccess;
    if (sums_stream != stream && sums_stream != wgrad_stream)
      ok = ok && hipStreamWaitEvent((hipStream_t)sums_stream, ev, 0) == hipSuccess;
    if (!ok) {
      ft_set_error("fft_blocks_bwd: stream fork failed");
      return FT_ERR_HIP;
    }
    return FT_OK;
  };
  static const bool per_stage = [] {                 // FT_FFT_FORK_PER_BLOCK=1: one fork per block (A/B knob)
    const char* e = getenv("FT_FFT_FORK_PER_BLOCK");
    return !(e && e[0] == '1');
  }();
  static const bool batch_sums = [] {                // FT_FFT_BATCH_SUMS=0: eight column-sum calls per block (A/B knob)
    const char* e = getenv("FT_FFT_BATCH_SUMS");
    return !(e && e[0] == '0');
  }();
  for (int i = n - 1; i >= 0; --i) {
    const FtFFTBlock& b = blocks[i];
    const FtFFTBlockGrads& g = grads[i];
    if (!per_stage) {
      FT_TRY(block_bwd_data(b, g, workspace, workspace_bytes, stream));
      FT_TRY(follow());
      FT_TRY(block_bwd_weights(b, g, wgrad_workspace, wgrad_workspace_bytes, wgrad_stream));
      FT_TRY(block_bwd_sums(b, g, sums_workspace, sums_workspace_bytes, sums_stream));
      continue;
    }
    const int rows = b.B * b.T, d = b.d, f = b.dfft;
    const int hd = d / b.nheads;
    const float scale = 1.0f / sqrtf((float)hd);
    void* const st = stream;
    void* const wst = wgrad_stream;
    void* const sst = sums_stream;
    // (the same launches with the same operands as block_bwd_data / _weights / _sums, interleaved)
    FT_TRY(ft_layernorm_bwd(g.dy2, b.s2, b.n2_g, b.mean2, b.rstd2, g.d_y1, g.t2, g.d_h2, rows, d, b.p_drop, b.seed_ln2, st));
    FT_TRY(follow());
    FT_TRY(ft_conv1d_bwd_weight(g.d_h2, d, b.h1, f, g.g_c2_w, b.B, b.T, f, d, b.k2, b.T, b.T, wgrad_workspace,
                                wgrad_workspace_bytes, wst));
    if (!batch_sums) {
      FT_TRY(ft_colsum2(g.t2, g.dy2, d, g.g_n2_g, g.g_n2_b, rows, d, sums_workspace, sums_workspace_bytes, sst));
      FT_TRY(ft_colsum(g.d_h2, d, g.g_c2_b, rows, d, 1.0f, 0, sums_workspace, sums_workspace_bytes, sst));
    }
    FT_TRY(ft_conv1d_bwd_data_relu(g.d_h2, d, b.c2_wpt, b.h1, g.g_h1, f, b.B, b.T, f, d, b.k2, 1, st));
    FT_TRY(follow());
    FT_TRY(ft_conv1d_bwd_weight(g.g_h1, f, b.y1, d, g.g_c1_w, b.B, b.T, d, f, b.k1, b.T, b.T, wgrad_workspace,
                                wgrad_workspace_bytes, wst));
    if (!batch_sums) FT_TRY(ft_colsum(g.g_h1, f, g.g_c1_b, rows, f, 1.0f, 0, sums_workspace, sums_workspace_bytes, sst));
    FT_TRY(ft_conv1d_bwd_data(g.g_h1, f, b.c1_wpt, g.d_y1, d, b.B, b.T, d, f, b.k1, b.T, b.T, 1, 1, st));     // += residual path
    FT_TRY(ft_layernorm_bwd(g.d_y1, b.s1, b.n1_g, b.mean1, b.rstd1, g.d_h, g.t1, g.d_sa, rows, d, b.p_drop, b.seed_ln1, st));
    FT_TRY(follow());
    FT_TRY(ft_linear_bwd_weight(g.d_sa, d, b.att, d, g.g_out_w, rows, d, d, 1, rows, 0, 0, 0, 0, wgrad_workspace,
                                wgrad_workspace_bytes, wst));
    if (!batch_sums) {
      FT_TRY(ft_colsum2(g.t1, g.d_y1, d, g.g_n1_g, g.g_n1_b, rows, d, sums_workspace, sums_workspace_bytes, sst));
      FT_TRY(ft_colsum(g.d_sa, d, g.g_out_b, rows, d, 1.0f, 0, sums_workspace, sums_workspace_bytes, sst));
    }
    FT_TRY(ft_linear_bwd_data(g.d_sa, d, b.out_wT, g.datt, d, rows, d, d, 0, 0, 0, 1, st));
    FT_TRY(ft_attn_bwd(b.qkv, b.att, g.datt, b.key_pad, b.lse2, g.dqkv, b.B, b.T, b.nheads, hd, scale, b.p_drop, b.seed_attn,
                       workspace, workspace_bytes, st));
    FT_TRY(follow());
    FT_TRY(ft_linear_bwd_weight(g.dqkv, 3 * d, b.x, d, g.g_in_w, rows, d, 3 * d, 1, rows, 0, 0, 0, 0, wgrad_workspace,
                                wgrad_workspace_bytes, wst));
    if (batch_sums) FT_TRY(block_bwd_sums_batched(b, g, sums_workspace, sums_workspace_bytes, sst));
    else FT_TRY(ft_colsum(g.dqkv, 3 * d, g.g_in_b, rows, 3 * d, 1.0f, 0, sums_workspace, sums_workspace_bytes, sst));
    FT_TRY(ft_linear_bwd_data(g.dqkv, 3 * d, b.in_wT, g.d_h, d, rows, d, 3 * d, 1, 0, 0, 1, st));
    FT_REQUIRE(g.dx == g.d_h, "fft_blocks_bwd: dx must alias d_h (the in-projection's data gradient accumulates into it)");
  }
  return FT_OK;
}

}  // extern "C"
