// Pre-split weight operands for the pipelined 128x128 NT GEMM (ft_gemm_b3.hip, BP = true).
//
// A weight matrix W [rows][ld] fp32 that GEMM launches use as their B operand is re-split into bf16 pieces by every one
// of the launch's row tiles (210 of them for a frame-side launch), although weights only change at the optimizer step.
// Here the pieces are produced ONCE per step: planes[row][c][piece][16] bf16, c = 16-k chunk, piece = hi | mid | lo of
// the exact split x = hi + mid + lo (ft_split.h: the same round-to-nearest pieces the kernel computes while staging, so
// a launch gives the same bits with or without planes), the 16 k of a chunk in the kernel's LDS order
//     position p = 8 h + i  <->  k = (i < 4 ? 4 h + i : 8 + 4 h + (i - 4)),   h = 0, 1
// (a staging thread owns half h of a row), chunks zero-filled beyond ld.  96 B per row and chunk.
//
// Registry.  The owner of the weights (hip.PackCache: raw 2-D weights, their transposes, tap-major conv packs, bank
// packs) registers each matrix with a planes buffer.  A launch whose B pointer falls into a registered matrix (whole rows
// apart, k offset a multiple of 16, same ld) gets the planes pointer instead -- but only if that matrix was refreshed
// since the last invalidation.  Matrices are refreshed on demand: the first launch that would have used one marks it
// WANTED; ft_planes_refresh (once per step, right behind the weight packs) splits every wanted matrix in ONE table-driven
// launch.  ft_planes_invalidate (end of the step: the optimizer is about to change the weights) makes every matrix
// stale again.  Results never depend on any of this -- only the time does.
//
// MEASURED (round 2, same-box A/B): OFF by default (FT_GEMM_PLANES=1 / ft_planes_enable turn it on).  The
// BP kernel is not faster: postnet bank forward 0.320 -> 0.340 ms, train step 24.38 -> 24.83 ms, FastPitch bf16 18.41 ->
// 18.55 ms.  Halving the split VALU buys nothing (the loop is not VALU-bound) while the third 16-B load per row and its
// four more registers push the NP = 3 kernel from 3 to 15 spilled registers at its 256-VGPR budget.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "ft_gemm.h"
#include "ft_split.h"

namespace {

struct Entry {
  const float* w;
  long rows, ld;
  char* planes;
  bool wanted, fresh;
};
struct Desc {                   // device copy of a wanted entry
  const float* w;
  char* planes;
  long first;                   // first (row, chunk) item of this entry in the launch
  int ld, chunks;
};

std::mutex g_mu;
std::vector<Entry> g_entries;   // sorted by w
bool g_table_dirty = true;
Desc* g_dev_table = nullptr;
int g_dev_cap = 0, g_dev_n = 0;
long g_items = 0;
long g_hits = 0, g_misses = 0;

__global__ __launch_bounds__(256) void ft_split_planes_kernel(const Desc* __restrict__ table, int n, long items) {
  const long it = (long)blockIdx.x * 256 + threadIdx.x;
  if (it >= items) return;
  int lo = 0, hi = n - 1;                         // entry of this item: last one with first <= it
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].first <= it) lo = mid;
    else hi = mid - 1;
  }
  const Desc d = table[lo];
  const long local = it - d.first;
  const long row = local / d.chunks;
  const int c = (int)(local - row * d.chunks);
  const float* src = d.w + row * d.ld + 16 * c;
  float x[16];
  const int left = d.ld - 16 * c;                 // ld % 4 == 0 (checked at registration): whole float4s
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (4 * q < left) v = *reinterpret_cast<const float4*>(src + 4 * q);
    x[4 * q] = v.x;
    x[4 * q + 1] = v.y;
    x[4 * q + 2] = v.z;
    x[4 * q + 3] = v.w;
  }
  // LDS order: positions 0..3 = k 0..3, 4..7 = k 8..11, 8..11 = k 4..7, 12..15 = k 12..15
  unsigned hi_[8], mid_[8], lo_[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) {                   // pair p = positions 2p, 2p + 1
    const int h = p >> 2, i = (2 * p) & 7;
    const int k = i < 4 ? 4 * h + i : 8 + 4 * h + (i - 4);
    ft_split_pair(x[k], x[k + 1], hi_[p], mid_[p], lo_[p]);
  }
  uint4* out = reinterpret_cast<uint4*>(d.planes + (row * d.chunks + c) * 96);
  out[0] = make_uint4(hi_[0], hi_[1], hi_[2], hi_[3]);
  out[1] = make_uint4(hi_[4], hi_[5], hi_[6], hi_[7]);
  out[2] = make_uint4(mid_[0], mid_[1], mid_[2], mid_[3]);
  out[3] = make_uint4(mid_[4], mid_[5], mid_[6], mid_[7]);
  out[4] = make_uint4(lo_[0], lo_[1], lo_[2], lo_[3]);
  out[5] = make_uint4(lo_[4], lo_[5], lo_[6], lo_[7]);
}

int g_enabled = -1;                // -1: not decided yet (FT_GEMM_PLANES, default off)
bool planes_enabled() {
  if (g_enabled < 0) {
    const char* e = getenv("FT_GEMM_PLANES");
    g_enabled = (e && e[0] == '1') ? 1 : 0;
  }
  return g_enabled == 1;
}

}  // namespace

// B pointer of a launch -> planes pointer of the same first row / k chunk, or nullptr (not registered, not aligned to
// rows / chunks, stale -- a stale hit marks the matrix wanted for the next refresh).  rows_needed: rows from the
// pointer's row that the launch may touch (taps included).
const void* ft_planes_lookup(const float* b, long ldb, long rows_needed) {
  if (!planes_enabled()) return nullptr;
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_entries.empty()) return nullptr;
  auto it = std::upper_bound(g_entries.begin(), g_entries.end(), b, [](const float* p, const Entry& e) { return p < e.w; });
  if (it == g_entries.begin()) return nullptr;
  Entry& e = *(it - 1);
  const long off = b - e.w;
  if (off < 0 || off >= e.rows * e.ld || ldb != e.ld) return nullptr;
  const long row = off / e.ld, k0 = off - row * e.ld;
  if (k0 % 16 != 0 || row + rows_needed > e.rows) return nullptr;
  if (!e.fresh) {
    if (!e.wanted) {
      e.wanted = true;
      g_table_dirty = true;
    }
    ++g_misses;
    return nullptr;
  }
  ++g_hits;
  const long chunks = (e.ld + 15) / 16;
  return e.planes + (row * chunks + k0 / 16) * 96;
}

extern "C" {

size_t ft_planes_bytes(long rows, long ld) {
  if (rows <= 0 || ld <= 0) return 0;
  return (size_t)rows * (size_t)((ld + 15) / 16) * 96;
}

// (one registry per process: one process drives one GPU -- the registry remembers the device it was first used on and
//  refuses another)
static int g_planes_dev = -1;
static bool planes_device_ok() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  if (g_planes_dev < 0) g_planes_dev = dev;
  return g_planes_dev == dev;
}

int ft_planes_register(const float* w, long rows, long ld, void* planes) {
  FT_REQUIRE(planes_device_ok(), "ft_planes_register: the planes registry belongs to device %d (one process per GPU)", g_planes_dev);
  FT_REQUIRE(w && planes && rows > 0 && ld > 0 && ld % 4 == 0, "ft_planes_register: rows %ld, ld %ld (ld %% 4 must be 0)", rows, ld);
  FT_REQUIRE(((uintptr_t)w & 15) == 0 && ((uintptr_t)planes & 15) == 0, "ft_planes_register: 16-byte alignment");
  std::lock_guard<std::mutex> lk(g_mu);
  for (const Entry& e : g_entries)
    FT_REQUIRE(w + rows * ld <= e.w || e.w + e.rows * e.ld <= w, "ft_planes_register: overlaps a registered matrix");
  const Entry e = {w, rows, ld, (char*)planes, false, false};
  g_entries.insert(std::upper_bound(g_entries.begin(), g_entries.end(), e, [](const Entry& a, const Entry& b) { return a.w < b.w; }), e);
  g_table_dirty = true;
  return FT_OK;
}

int ft_planes_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  const int old = planes_enabled() ? 1 : 0;
  if (on >= 0) g_enabled = on ? 1 : 0;
  return old;
}

int ft_planes_clear(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_entries.clear();
  g_table_dirty = true;
  return FT_OK;
}

int ft_planes_invalidate(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (Entry& e : g_entries) e.fresh = false;
  return FT_OK;
}

int ft_planes_refresh(void* stream) {
  if (!planes_enabled()) return FT_OK;
  std::lock_guard<std::mutex> lk(g_mu);
  hipStream_t s = (hipStream_t)stream;
  if (g_table_dirty) {
    std::vector<Desc> host;
    long items = 0;
    for (const Entry& e : g_entries)
      if (e.wanted) {
        const int chunks = (int)((e.ld + 15) / 16);
        host.push_back({e.w, e.planes, items, (int)e.ld, chunks});
        items += e.rows * chunks;
      }
    if ((int)host.size() > g_dev_cap) {
      // the previous table may still be read by a launch in flight: it is left to the allocator only after the stream
      // has drained (a table is rebuilt a handful of times in the life of a process)
      if (g_dev_table) {
        (void)hipStreamSynchronize(s);
        (void)hipFree(g_dev_table);
      }
      g_dev_cap = (int)host.size() + 64;
      if (hipMalloc(&g_dev_table, sizeof(Desc) * g_dev_cap) != hipSuccess) {
        g_dev_table = nullptr;
        g_dev_cap = 0;
        ft_set_error("ft_planes_refresh: hipMalloc failed");
        return FT_ERR_HIP;
      }
    }
    if (!host.empty()) {
      // pageable source: the copy is staged by the runtime before the call returns
      if (hipMemcpyAsync(g_dev_table, host.data(), sizeof(Desc) * host.size(), hipMemcpyHostToDevice, s) != hipSuccess ||
          hipStreamSynchronize(s) != hipSuccess) {
        ft_set_error("ft_planes_refresh: table upload failed");
        return FT_ERR_HIP;
      }
    }
    g_dev_n = (int)host.size();
    g_items = items;
    g_table_dirty = false;
  }
  if (g_dev_n > 0) {
    hipLaunchKernelGGL(ft_split_planes_kernel, dim3((unsigned)((g_items + 255) / 256)), dim3(256), 0, s, g_dev_table, g_dev_n,
                       g_items);
    for (Entry& e : g_entries)
      if (e.wanted) e.fresh = true;
    return ft_check_launch("split_planes");
  }
  return FT_OK;
}

int ft_planes_counters(long* hits, long* misses, long* registered, long* wanted) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (hits) *hits = g_hits;
  if (misses) *misses = g_misses;
  if (registered) *registered = (long)g_entries.size();
  if (wanted) {
    long n = 0;
    for (const Entry& e : g_entries) n += e.wanted ? 1 : 0;
    *wanted = n;
  }
  return FT_OK;
}

}  // extern "C"
