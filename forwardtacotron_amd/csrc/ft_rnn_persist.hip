// PERSISTENT bidirectional GRU / LSTM recurrences for gfx950: one launch runs all T steps.
//
// Why: the XCD L2s do not keep W_hh across kernel boundaries, so per-step launches re-stream the recurrent weights
// from Infinity Cache every step (measured 14.8 MB/step, profiles/r01_pmc_rnn_step.txt).  Here every wave keeps its
// W_hh fragments in REGISTERS for the whole sequence and only h (forward) or d(gates) (backward) travels between
// workgroups, through a small exchange buffer:
//   exchange layout  xb[parity][group = (dir, batch group of 16)][k/4][16][4]   (k = contraction index; a storing wave
//                    owns whole [16][4] blocks = two full 128-B lines, written by ONE store instruction)
//   producer: write-through (sc1) stores FIRST -> that wave's s_waitcnt vmcnt(0) -> its lane 0 adds 1 to the
//             arrival-counter shard of the block (agent scope).  The step's other outputs (h / c / saved gates,
//             d(pre-activations)) are plain stores issued AFTER the signal, off the critical path.
//   consumer: 16 lanes of wave 0 poll the 16 shards of the group with sc1 loads until each holds
//             step * (#producing waves of the shard); workgroup barrier; sc1 16-B loads of the operand rows.
//   (cdna_hip_programming.md Guideline 16 / MI355X_MICROARCH.md "Valid forms", row "each storing wave for itself".)
// XCD-LOCAL mode.  The agent-scope hand-off above costs two fabric round trips per step (write-through stores drop
// the line from the XCD's L2; counters live at the memory side).  When the grid is laid out so that every workgroup of
// a (direction, batch group) group sits in ONE XCD slot (blockIdx % 8 equal), the group's hand-off can stay inside
// that XCD's L2, which IS coherent for its own CUs: plain stores (the line stays in L2) -> the wave's s_waitcnt
// vmcnt(0) -> a plain store of (step + 1) to the wave's own flag word; consumers poll the flag words and read the
// operands with L1-bypassing (sc1) loads, served by the L2.  Placement is NOT assumed (HIP promises none): every
// workgroup publishes its HW_REG_XCC_ID with the agent-scope protocol during step 0; at step 1 -- still on the
// agent-scope protocol -- every workgroup of a group sees the same 'all ids equal' verdict and the group switches to
// the local protocol from then on, or stays on the agent-scope one for the whole launch.  Any placement is correct.
// Step s reads parity (s-1)&1 and writes parity s&1; a workgroup can only be one step ahead of the slowest producer it
// depends on, so two parities suffice.  Every spin is bounded: on timeout the workgroup raises the device's STICKY
// fault word (g_rnn_fault: no launch ever clears it, only ft_rnn_status does) and leaves; all others then time out at
// the same step, so the grid always drains.  ft_clip_grad_norm / ft_adam_step read that word on the device and skip
// the update, so a timed-out recurrence can never reach the parameters (trainer.TrainStep surfaces it).  All
// workgroups must be co-resident: the host admits a launch only while the persistent grids IN FLIGHT on the device
// (all streams, tracked with events) plus this one fit the occupancy query; otherwise (or with FT_RNN_PERSISTENT=0)
// the per-step kernels run.
//
// What the per-step time is made of (s_memtime phase profile, LSTM 512, B=32: poll 1.0 us, operand loads 0.8,
// MFMA 0.6, LDS reduce + barrier 0.6, cell 0.6, store drain 0.4, top 0.4) shaped this version:
//   * a workgroup covers 16 batch rows (not 32): the operand volume per CU per step (rows x K x 4 B, streamed
//     through one 64 B/clk L1 path) halves, and a (dir, 16-row) group has half as many producers to wait for;
//   * the x-projection / saved-activation operands of step s+1 are requested during step s (loads return in order, so
//     a same-step request would sit in front of the exchange loads with its HBM latency);
//   * 1-D grid with an XCD-aware decode: the 8 XCDs take contiguous ranges of (group, chunk), so a group's exchange
//     lines and the 128-B lines of out / gates / xp rows are shared inside one or two L2s instead of all eight.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "ft_rnn.h"

namespace {

constexpr int NSH = 16;                // arrival-counter shards per group
constexpr int CSTRIDE = 32;            // one counter per 128-B line
constexpr unsigned MAX_SPINS_DEFAULT = 1u << 18;
constexpr int MB = 16;                 // batch rows per workgroup
constexpr int NFLAG = 256;             // XCD-local mode: flag words per group (one per signalling wave)
constexpr int NXCC = 64;               // published XCC ids per group (one per workgroup)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

// exact 3-way bf16 split of 8 fp32 values (x = hi + mid + lo to 2^-27 |x|; see ft_gemm_b3.hip): the recurrent matmuls
// then run as six v_mfma_f32_16x16x32_bf16 per 32-k block instead of eight v_mfma_f32_16x16x4_f32 -- 2.67x fewer
// matrix-pipe cycles at fp32-class accuracy (B3 variants of the kernels below; need H % 32 == 0).  Round-to-nearest
// pieces through v_cvt_pk_bf16_f32 (two values per instruction, already packed): hi = rn(x), r1 = x - hi (exact),
// mid = rn(r1), r2 = r1 - mid (exact), lo = rn(r2).  One definition for every place a value is split -- operands
// split by the consumer (agent-scope hand-off) and values split by the producer (XCD-local granules) give the same bits.
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
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
__device__ __forceinline__ void split8(const float4& v0, const float4& v1, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
  unsigned h[4], m[4], l[4];
  split_pair(v0.x, v0.y, h[0], m[0], l[0]);
  split_pair(v0.z, v0.w, h[1], m[1], l[1]);
  split_pair(v1.x, v1.y, h[2], m[2], l[2]);
  split_pair(v1.z, v1.w, h[3], m[3], l[3]);
  hi = __builtin_bit_cast(bf16x8, u32x4{h[0], h[1], h[2], h[3]});
  mid = __builtin_bit_cast(bf16x8, u32x4{m[0], m[1], m[2], m[3]});
  lo = __builtin_bit_cast(bf16x8, u32x4{l[0], l[1], l[2], l[3]});
}
__device__ __forceinline__ void mfma6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4& acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0);      // small terms first
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
}
__device__ __forceinline__ float4 as_f4(const u32x4& v) {
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

struct Geom {
  int nchunks, nbg, total;             // chunks per group, batch groups, workgroups that have work
  int xcd_aware, sig_per_wave;
  int xcd_off;                         // group-aligned placement: physical XCD of logical slot 0
  unsigned max_spins;                  // bound of every arrival poll (ft_rnn_set_max_spins; tests force timeouts with it)
  int local_ok;                        // groups are XCD-slot aligned: the XCD-local hand-off may be used if placement agrees
  int gran;                            // XCD-local hand-off by granules where the kernel has them (FT_RNN_GRANULES=0: flag words)
  int fast;                            // cell update with the hardware exp / rcp forms (FT_RNN_FAST_CELL)
};
__device__ __forceinline__ float sig_(float x, bool fast) { return fast ? ft_sigmoid_fast(x) : ft_sigmoid(x); }
__device__ __forceinline__ float tanh_(float x, bool fast) { return fast ? ft_tanh_fast(x) : ft_tanh(x); }

// Diagnostic build only (hipcc -DFT_RNN_PROF, lab/build_prof.sh -> lab/libfwdtaco_prof.so, never the product library):
// s_memtime stamps at the seams of a time step, summed per phase over the launch for two waves of one workgroup and read
// back with ft_rnn_prof_read.  The stamps go to a buffer nothing else reads; no output depends on them.
#ifdef FT_RNN_PROF
__device__ unsigned long long g_prof[64];
#define PROF_DECL unsigned long long pt_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pl_ = __builtin_amdgcn_s_memtime()
#define PROF(i)                                              \
  {                                                          \
    const unsigned long long n_ = __builtin_amdgcn_s_memtime(); \
    pt_[i] += n_ - pl_;                                      \
    pl_ = n_;                                                \
  }
#define PROF_DUMP(sel, slot)                                              \
  if ((sel) && (threadIdx.x & 63) == 0)                                   \
    for (int i_ = 0; i_ < 12; ++i_) g_prof[(slot) * 12 + i_] = pt_[i_]
#else
#define PROF_DECL
#define PROF(i)
#define PROF_DUMP(sel, slot)
#endif

// sticky per-device fault word (one 128-B line of its own): set by any workgroup whose poll ran out, cleared only by
// ft_rnn_status; read on the device by the optimizer kernels (ft_optim.hip) through ft_rnn_fault_word().  Words 8 / 9
// of the line count the groups that ran XCD-local / on the agent-scope protocol (ft_rnn_mode_counts).
__device__ unsigned g_rnn_fault[32];

__device__ __forceinline__ float4 ld_sc1_b128(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16);     // aux 16 = sc1
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// wave 0 waits until every shard k holds >= step * n_k arrivals; returns false on timeout (wave-uniform)
__device__ __forceinline__ bool wait_arrivals(const unsigned* cnt, unsigned step, int nprod, int lane,
                                              unsigned max_spins) {
  bool ok = true;
  if (lane < NSH) {
    const unsigned nk = lane < nprod ? (unsigned)((nprod - lane + NSH - 1) / NSH) : 0u;
    const unsigned target = step * nk;
    unsigned spins = 0;
    if (max_spins == 0) ok = false;        // fault injection (ft_rnn_set_max_spins(-1)): every poll fails at once
    while (ok && __hip_atomic_load(cnt + lane * CSTRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > max_spins) {
        ok = false;
        break;
      }
    }
  }
  return __all(ok);
}

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xFu;
}

// XCD-local mode: EVERY wave polls, on its own, the flag words of the producers of its own operand slice (lane l polls
// the word f, or nothing if f is null) until each holds >= step.  The loads bypass L1 (sc1) and are served by the XCD's
// L2, where the producers' plain flag stores land; no sleep: an L2 poll costs the fabric nothing.
__device__ __forceinline__ bool poll_flag(const unsigned* f, unsigned step, unsigned max_spins) {
  bool ok = max_spins != 0;              // 0 = fault injection
  if (f) {
    unsigned spins = 0;
    while (ok && __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < step)
      if (++spins > max_spins) ok = false;
  }
  return __all(ok);
}

// XCD-local mode of the bf16-split kernels: the hand-off unit is an 8-byte GRANULE {hi, mid, lo, tag} -- the value already
// split into its three bf16 pieces by the PRODUCER (once, instead of by every consumer of every step) and a 16-bit tag
// = (step + 1) & 0xffff.  One 8-byte store is one indivisible memory transaction, so the tag validates the value it
// travels with: no drain, no flag store, no flag poll -- a consumer loads the granules of its operand slice (L1-bypassing
// loads, served by the XCD's L2), and reloads a block while any of its tags is not the awaited one.  Granules live in an
// area of their own behind the two fp32 parities (never written with anything else; zeroed per launch), in two parities
// like the fp32 exchange (a slot is rewritten every second step: a stale tag is the awaited one minus 2).
struct Gran {
  u32x4 v[4];           // the 8 granules of one lane's fragment (k = 8q .. 8q+7 of one row): v[j] = granules 2j, 2j+1
};
constexpr int GL_AUX = 16 | (int)0x80000000;        // sc1 + volatile (a poll must really re-load)
__device__ __forceinline__ void gran_load(Gran& g, __amdgpu_buffer_rsrc_t rs, unsigned offA, unsigned offB) {
  g.v[0] = __builtin_amdgcn_raw_buffer_load_b128(rs, offA, 0, GL_AUX);
  g.v[1] = __builtin_amdgcn_raw_buffer_load_b128(rs, offA + 16, 0, GL_AUX);
  g.v[2] = __builtin_amdgcn_raw_buffer_load_b128(rs, offB, 0, GL_AUX);
  g.v[3] = __builtin_amdgcn_raw_buffer_load_b128(rs, offB + 16, 0, GL_AUX);
}
__device__ __forceinline__ bool gran_ok(const Gran& g, unsigned want_hi) {      // want_hi = tag << 16
  unsigned x = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) x |= (g.v[j].y ^ want_hi) | (g.v[j].w ^ want_hi);
  return (x & 0xFFFF0000u) == 0u;
}
__device__ __forceinline__ void gran_unpack(const Gran& g, bf16x8 (&a3)[3]) {
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    h[j] = __builtin_amdgcn_perm(g.v[j].z, g.v[j].x, 0x05040100u);        // low halves:  hi(2j) | hi(2j+1) << 16
    m[j] = __builtin_amdgcn_perm(g.v[j].z, g.v[j].x, 0x07060302u);        // high halves: mid
    l[j] = __builtin_amdgcn_perm(g.v[j].w, g.v[j].y, 0x05040100u);        // low halves of the second dwords: lo
  }
  a3[0] = __builtin_bit_cast(bf16x8, u32x4{h[0], h[1], h[2], h[3]});
  a3[1] = __builtin_bit_cast(bf16x8, u32x4{m[0], m[1], m[2], m[3]});
  a3[2] = __builtin_bit_cast(bf16x8, u32x4{l[0], l[1], l[2], l[3]});
}
// one value -> its granule (the same round-to-nearest split as split8)
__device__ __forceinline__ unsigned long long gran_make(float x, unsigned tag) {
  const unsigned hi = rn_pack(x, 0.f) & 0xFFFFu;
  const float r1 = x - __uint_as_float(hi << 16);
  const unsigned mid = rn_pack(r1, 0.f) & 0xFFFFu;
  const float r2 = r1 - __uint_as_float(mid << 16);
  const unsigned lo = rn_pack(r2, 0.f) & 0xFFFFu;
  return (unsigned long long)(hi | (mid << 16)) | ((unsigned long long)(lo | (tag << 16)) << 32);
}
// waits (bounded) until the block's granules all carry the awaited tag; returns false on timeout (wave-uniform)
__device__ __forceinline__ bool gran_wait(Gran& g, __amdgpu_buffer_rsrc_t rs, unsigned offA, unsigned offB, unsigned want_hi,
                                          unsigned max_spins) {
  unsigned spins = 0;
  while (!__all(gran_ok(g, want_hi))) {
    if (++spins > max_spins) return false;
    gran_load(g, rs, offA, offB);
  }
  return max_spins != 0;                  // 0 = fault injection
}

// step 1, wave 0, after the agent-scope arrival wait: did every workgroup of the group report my XCC id?
__device__ __forceinline__ bool same_xcd(const unsigned* ids, int nchunks, unsigned mine, int lane) {
  bool same = true;
  for (int i = lane; i < nchunks; i += 64)
    same &= __hip_atomic_load(ids + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == mine;
  return __all(same);
}

// XCD-aware decode of the 1-D grid: dispatch is round-robin over the 8 XCDs, so XCD x runs work items
// [x*per, (x+1)*per).  Only locality depends on that; any placement is correct.
__device__ __forceinline__ bool decode(const Geom& g, int& d, int& bgp, int& chunk, int& grp) {
  const int lin = blockIdx.x, per = gridDim.x >> 3;
  const int w = g.xcd_aware ? (((lin & 7) + 8 - g.xcd_off) & 7) * per + (lin >> 3) : lin;
  if (w >= g.total) return false;
  grp = w / g.nchunks;
  chunk = w - grp * g.nchunks;
  d = grp / g.nbg;
  bgp = grp - d * g.nbg;
  return true;
}

// ---------------------------------------------------------------------------------------------------
// forward: workgroup = 16 batch rows x UB hidden units (all G gates = G*UB/16 column tiles), K = H over NW waves.
// UB = 8, or 16 for the 512-wide LSTM (32 workgroups per group = one XCD) and the 256-wide GRU: see fwd_persistent
// ---------------------------------------------------------------------------------------------------
template <int G, int NW, bool B3, int UB, int BC>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void ft_rnn_fwd_persist_kernel(RnnFwdArgs a, Geom geo, float* xb, unsigned* cnt,
                                                                     unsigned* fault, unsigned xb_bytes) {
  // UB hidden units (all G gates) per workgroup: G*UB columns = NT tiles of 16; cell waves = UB / 4 (one [16][4]
  // exchange block each)
  // BC = 32-k blocks of W_hh a wave keeps resident (bf16-split form): 1, 2 or 4, sized by the host to ceil(H/32 / NW)
  constexpr int NT = (G * UB + 15) / 16, BCH = BC, CW = UB / 4;
  // granule hand-off (XCD-local mode): two blocks of granules in flight (32 registers) whatever BC is; the 4-block form
  // (LSTM-512 on 4 waves) takes it too since round 3 (FT_RNN_GRAN4=0: flag words)
  constexpr bool GRAN = B3 && BC <= 2;
  static_assert(UB % 4 == 0 && CW <= NW, "cell waves");
  // partial tiles, double-buffered by step parity: in XCD-local mode no barrier separates a step's readers (cell
  // threads) from the next step's writers
  __shared__ float red2[2 * NW * NT * 16 * RLD];
  __shared__ int s_ok;
  int d, bgp, chunk, grp;
  if (!decode(geo, d, bgp, chunk, grp)) return;
  const int u0 = chunk * UB, b0 = bgp * MB;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int q = lane >> 4, l15 = lane & 15;
  const int H = a.H, T = a.T, nq = H / 4;
  const long ldo = (long)a.ND * H;
  const long ldx = (long)a.ND * G * H;
  const long grp_floats = (long)nq * MB * 4;
  const long par_floats = (long)2 * geo.nbg * grp_floats;                 // one parity
  const long base_floats = (long)grp * grp_floats;
  unsigned* mycnt = cnt + (long)grp * NSH * CSTRIDE;
  const int nprod = (geo.sig_per_wave ? CW : 1) * geo.nchunks;            // signalling waves per group
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(xb, 0, (int)xb_bytes, 0x00020000);
  // XCD-local mode state (file header): flag words and published XCC ids of this group
  unsigned* myflags = cnt + (long)2 * geo.nbg * NSH * CSTRIDE + (long)grp * NFLAG;
  unsigned* myxcc = cnt + (long)2 * geo.nbg * (NSH * CSTRIDE + NFLAG) + (long)grp * NXCC;
  const unsigned my_xcc = xcc_id() + 1u;
  __shared__ int s_local, s_fail;
  bool local = false;
  if (tid == 0) {
    s_local = 0;
    s_fail = 0;
    __hip_atomic_store(myxcc + chunk, my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // drained before wave 0's first signal
  }

  // ---- resident W_hh fragments of this wave: tile nt, column l15 -> (gate, unit) = ((nt*16+l15)/8, (nt*16+l15)%8)
  const int ngroups = H / 16;
  const int gpw = (ngroups + NW - 1) / NW;
  const int g0 = wave * gpw, g1 = min(ngroups, g0 + gpw);
  // B3: 32-k blocks [kb0, kb1) of this wave, W fragments pre-split into (hi, mid, lo) once
  const int nblk = H / 32;
  const int bpw = (nblk + NW - 1) / NW;
  const int kb0 = wave * bpw, kb1 = min(nblk, kb0 + bpw);
  float4 bv[NT][B3 ? 1 : GCH];
  bf16x8 bw[NT][B3 ? BCH : 1][3];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = nt * 16 + l15, gj = col / UB, ul = col - gj * UB;
    const bool valid = gj < G;
    const float* brow = a.whh[d] + (valid ? ((long)gj * H + u0 + ul) * H : 0);
    if constexpr (!B3) {
#pragma unroll
      for (int c = 0; c < GCH; ++c)
        bv[nt][c] = (valid && g0 + c < g1) ? *reinterpret_cast<const float4*>(brow + 16 * (g0 + c) + 4 * q)
                                           : make_float4(0, 0, 0, 0);
    } else {
#pragma unroll
      for (int c = 0; c < BCH; ++c) {
        const bool in = valid && kb0 + c < kb1;
        const float* p = brow + (in ? 32 * (kb0 + c) + 8 * q : 0);
        const float4 z = make_float4(0, 0, 0, 0);
        split8(in ? *reinterpret_cast<const float4*>(p) : z, in ? *reinterpret_cast<const float4*>(p + 4) : z,
               bw[nt][c][0], bw[nt][c][1], bw[nt][c][2]);
      }
    }
  }

  // XCD-local mode: this wave's operand slice is k-quads [q0, q0 + nq4) of h; the signalling wave (CW*chunk' + jq') that
  // produced quad i has flag index i
  const int pq0 = B3 ? 8 * kb0 : 4 * g0, pnq = B3 ? 8 * max(kb1 - kb0, 0) : 4 * max(g1 - g0, 0);
  const unsigned* pollf = lane < pnq ? myflags + pq0 + lane : nullptr;

  // ---- cell threads (waves 0 .. CW-1): wave jq owns units 4*jq..4*jq+3 of the chunk = one [16][4] exchange block
  const int jq = tid >> 6, ci = (tid >> 2) & 15, jj = tid & 3;
  const int cu = 4 * jq + jj;
  const int cb = b0 + ci, cun = u0 + cu;
  const bool sthr = tid < 64 * CW;
  const bool cthr = sthr && cb < a.B;
  const int L = cthr ? clamp_len(a.lens, cb, T) : 0;
  float bg[G];
#pragma unroll
  for (int g = 0; g < G; ++g) bg[g] = sthr ? a.bhh[d][g * H + cun] : 0.f;
  float hprev = 0.f, cprev = 0.f;
  float xg[G];
#pragma unroll
  for (int g = 0; g < G; ++g) xg[g] = 0.f;
  // Gated launches (RnnFwdArgs.gate): the x projection arrives in time chunks while this kernel runs.  ONE thread of the
  // workgroup keeps ahead of the cell threads' requests: at the top of step s it makes sure the chunk that holds the
  // rows of step s + 2 (requested during step s + 1, behind this step's barrier) is complete, polling the direction's
  // gate word if it has not seen it that far yet -- one sc1 load every few microseconds per workgroup (a first version
  // in which every cell thread polled kept 500 waves hammering one memory channel: the chunk GEMM they were waiting for
  // did not finish).  Direction 1 of a packed batch: item b starts at t = L_b - 1, i.e. (T - L_b) rows into the
  // direction's descending chunk order; the workgroup takes the deepest of its 16 rows.  The values are read past L1
  // (another kernel wrote them while this one runs).
  const bool gated = a.gate != nullptr;
  int gate_off = 0;
  unsigned gate_have = 1u;              // chunks of this direction seen complete (thread 0)
  if (gated && d == 1) {
    int lm = T;
    for (int i = 0; i < MB; ++i)
      if (b0 + i < a.B) lm = min(lm, clamp_len(a.lens, b0 + i, T));
    gate_off = T - lm;
  }
  auto gate_ahead = [&](int step) {     // thread 0: the rows the workgroup requests for `step` are complete
    const int pos = min(gate_off + step, T - 1);
    const unsigned need = (unsigned)pos / (unsigned)a.gate_cs;
    if (need >= gate_have) {
      bool ok = geo.max_spins != 0;       // 0 = fault injection
      unsigned spins = 0;
      while (ok && (gate_have = 1u + __hip_atomic_load(a.gate + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) <= need) {
        __builtin_amdgcn_s_sleep(127);
        if (++spins > geo.max_spins) ok = false;
      }
      if (!ok) s_fail = 1;
    }
  };
  if (gated) {
    __syncthreads();                    // s_fail is initialised
    if (tid == 0) {
      gate_ahead(0);
      gate_ahead(1);
    }
    __syncthreads();
  }
  auto xload = [&](int cn, float (&dst)[G]) {
    const float* xr = a.xp + ((long)cn * a.Bld + cb) * ldx + (long)d * G * H + cun;
    if (gated) {
#pragma unroll
      for (int g = 0; g < G; ++g)
        dst[g] = __hip_atomic_load(xr + (long)g * H, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
#pragma unroll
      for (int g = 0; g < G; ++g) dst[g] = xr[(long)g * H];
    }
  };
  if (cthr && 0 < L) xload(d == 0 ? 0 : L - 1, xg);

  PROF_DECL;
  for (int s = 0; s < T; ++s) {
    PROF(0);
    if (gated && tid == 0) gate_ahead(s + 2);
    const bool cact = cthr && s < L;
    const int ct = d == 0 ? s : L - 1 - s;
    float* red = red2 + (s & 1) * (NW * NT * 16 * RLD);
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[nt][e] = 0.f;
    float4 av[B3 ? 1 : GCH];
    float4 aw[B3 ? BCH : 1][2];
    // next step's x projection: requested BEHIND this step's hand-off loads (loads return in order: requested first it
    // would put its HBM latency in front of them); it is not needed before the next step's cell update
    float xn[G];
#pragma unroll
    for (int g = 0; g < G; ++g) xn[g] = 0.f;
    auto request_xn = [&]() {
      if (cthr && s + 1 < L) xload(d == 0 ? s + 1 : L - 2 - s, xn);
    };
    const bool gl = local;              // protocol of THIS step's operands (the mode may change below, at s == 1)
    // granule mode of this step (wave-uniform): XCD-local, bf16-split (file header).  Block c+1 is requested before
    // block c is consumed; a block whose tags are not all the awaited ones is re-loaded until they are.  The MFMA code
    // below is shared with the fp32-exchange forms (one site per (block, tile): two sites made hipcc hold W_hh twice).
    const bool gm = GRAN && s > 0 && gl && geo.gran;
    const unsigned want_hi = ((unsigned)s & 0xFFFFu) << 16;
    const unsigned gbase = (unsigned)(((long)2 * par_floats * 4) + ((long)((s - 1) & 1) * par_floats + base_floats) * 8);
    auto goffs = [&](int c, unsigned& oa, unsigned& ob) {
      const long quad = 8 * (kb0 + c) + 2 * q;
      oa = gbase + (unsigned)((quad * MB + l15) * 4 * 8);
      ob = gbase + (unsigned)(((quad + 1) * MB + l15) * 4 * 8);
    };
    Gran gr[2];
    if (s > 0 && !gm) {
      if (gl) {          // XCD-local, flag words: every wave waits for the producers of its own slice
        if (!poll_flag(pollf, (unsigned)s, geo.max_spins) && lane == 0) s_fail = 1;
      } else {
        if (wave == 0) {
          const bool ok = wait_arrivals(mycnt, (unsigned)s, nprod, lane, geo.max_spins);
          if (s == 1 && ok && geo.local_ok) {
            const bool same = same_xcd(myxcc, geo.nchunks, my_xcc, lane);
            if (lane == 0) s_local = same;
          }
          if (lane == 0) s_ok = ok;
        }
        __syncthreads();
        if (!s_ok) {
          if (tid == 0) atomicExch(fault, 1u);
          return;
        }
        if (s == 1) {
          local = s_local != 0;
          if (threadIdx.x == 0 && chunk == 0) atomicAdd(fault + (local ? 8 : 9), 1u);    // statistics: groups per mode
        }
      }
    }
    if (s > 0) {
      const long rbase = (long)((s - 1) & 1) * par_floats + base_floats;
      if constexpr (!B3) {
#pragma unroll
        for (int c = 0; c < GCH; ++c)
          if (g0 + c < g1) {
            const long quad = 4 * (g0 + c) + q;
            av[c] = ld_sc1_b128(rs, (unsigned)((rbase + (quad * MB + l15) * 4) * 4));
          }
        request_xn();
#pragma unroll
        for (int c = 0; c < GCH; ++c)
          if (g0 + c < g1) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) mfma4(av[c], bv[nt][c], acc[nt]);
          }
      } else {
        if (gm) {
          if constexpr (GRAN) {
            if (kb0 < kb1) {
              unsigned oa, ob;
              goffs(0, oa, ob);
              gran_load(gr[0], rs, oa, ob);
            }
          }
        } else {
#pragma unroll
          for (int c = 0; c < BCH; ++c)
            if (kb0 + c < kb1) {                      // lane (row l15, q): k = 32 blk + 8 q .. +7 = two exchange quads
              const long quad = 8 * (kb0 + c) + 2 * q;
              aw[c][0] = ld_sc1_b128(rs, (unsigned)((rbase + (quad * MB + l15) * 4) * 4));
              aw[c][1] = ld_sc1_b128(rs, (unsigned)((rbase + ((quad + 1) * MB + l15) * 4) * 4));
            }
          request_xn();
        }
        bool ok = true;
#pragma unroll
        for (int c = 0; c < BCH; ++c)
          if (kb0 + c < kb1) {
            bf16x8 a3[3];
            if (gm) {
              if constexpr (GRAN) {
                unsigned oa, ob;
                if (c + 1 < BCH && kb0 + c + 1 < kb1) {
                  goffs(c + 1, oa, ob);
                  gran_load(gr[(c + 1) & 1], rs, oa, ob);
                }
                goffs(c, oa, ob);
                ok = gran_wait(gr[c & 1], rs, oa, ob, want_hi, geo.max_spins) && ok;
                gran_unpack(gr[c & 1], a3);
              }
            } else {
              split8(aw[c][0], aw[c][1], a3[0], a3[1], a3[2]);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) mfma6(a3, bw[nt][c], acc[nt]);
          }
        if (gm) {
          if (!ok && lane == 0) s_fail = 1;
          request_xn();
        }
      }
    }
    if (s == 0) request_xn();
    PROF(1);
    store_partials<NT>(red, wave, lane, acc);
    PROF(2);
    __syncthreads();
    PROF(3);
    if (s_fail) {                                          // a wave's poll ran out (XCD-local mode): leave together
      if (tid == 0) atomicExch(fault, 1u);
      return;
    }

    float hnew = 0.f, cnew = 0.f, sg[4] = {0.f, 0.f, 0.f, 0.f};
    if (cact) {
      float hp[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int col = g * UB + cu;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[w * (NT * 16 * RLD) + ((col >> 4) * 16 + ci) * RLD + (col & 15)];
        hp[g] = v + bg[g];
      }
      if (G == 3) {
        const float r = sig_(xg[0] + hp[0], geo.fast);
        const float z = sig_(xg[1] + hp[1], geo.fast);
        const float n = tanh_(xg[2] + r * hp[2], geo.fast);
        hnew = (1.f - z) * n + z * hprev;
        sg[0] = r; sg[1] = z; sg[2] = n; sg[3] = hp[2];
      } else {
        const float ig = sig_(xg[0] + hp[0], geo.fast);
        const float fg = sig_(xg[1] + hp[1], geo.fast);
        const float gg = tanh_(xg[2] + hp[2], geo.fast);
        const float og = sig_(xg[G - 1] + hp[G - 1], geo.fast);
        cnew = fg * cprev + ig * gg;
        cprev = cnew;
        hnew = og * tanh_(cnew, geo.fast);
        sg[0] = ig; sg[1] = fg; sg[2] = gg; sg[3] = og;
      }
      hprev = hnew;
      // exchange block [CW*chunk + jq][16][4] of parity s&1, before anything else
      const long xe = (long)(s & 1) * par_floats + base_floats + (((long)CW * chunk + jq) * MB + ci) * 4 + jj;
      if (local && GRAN && geo.gran) {  // (granule mode: stored below, for EVERY row)
      } else if (local) {
        __hip_atomic_store(xb + xe, hnew, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);   // plain: stays in the XCD's L2
      } else {
        __hip_atomic_store(xb + xe, hnew, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // write-through (sc1)
      }
    }
    PROF(4);
    if (local && GRAN && geo.gran) {
      // granule {hi, mid, lo, tag = s + 1}: one 8-byte plain store per (row, unit), nothing else to signal.  EVERY row
      // slot is written every step -- rows beyond the batch and finished items as zeros -- because a consumer accepts a
      // block only once all its tags are the awaited ones
      if (sthr) {
        const long xe = (long)(s & 1) * par_floats + base_floats + (((long)CW * chunk + jq) * MB + ci) * 4 + jj;
        unsigned long long* gp = reinterpret_cast<unsigned long long*>(xb + 2 * par_floats) + xe;
        __hip_atomic_store(gp, gran_make(cact ? hnew : 0.f, (unsigned)(s + 1) & 0xFFFFu), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
    } else if (geo.sig_per_wave) {
      if (sthr) {                                          // the cell waves, wave-uniform
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
          if (local)
            __hip_atomic_store(myflags + CW * chunk + jq, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          else
            __hip_atomic_fetch_add(mycnt + ((CW * chunk + jq) % NSH) * CSTRIDE, 1u, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0)
        __hip_atomic_fetch_add(mycnt + (chunk % NSH) * CSTRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    PROF(5);
    if (cact) {                                          // the step's outputs proper: off the recurrence's path
      const long o = ((long)ct * a.Bld + cb) * ldo + (long)d * H + cun;
      a.out[o] = hnew;
      if (G == 4) a.cst[o] = cnew;
      if (a.gates) {
        float* gs = a.gates + (((long)ct * a.Bld + cb) * a.ND + d) * 4 * H + cun;
        gs[0] = sg[0]; gs[H] = sg[1]; gs[2 * H] = sg[2]; gs[3 * H] = sg[3];
      }
    } else if (cthr) {                                   // finished item: positions t >= L read as zeros (t = s, both directions)
      const long o = ((long)s * a.Bld + cb) * ldo + (long)d * H + cun;
      a.out[o] = 0.f;
      if (G == 4) a.cst[o] = 0.f;
    }
    PROF(6);
#pragma unroll
    for (int g = 0; g < G; ++g) xg[g] = xn[g];
    PROF(7);
  }
  PROF_DUMP(grp == 0 && chunk == 3 && wave == 0, 0);
  PROF_DUMP(grp == 0 && chunk == 3 && wave == NW - 1, 1);
}

// ---------------------------------------------------------------------------------------------------
// backward: workgroup = 16 batch rows x 16 hidden units, K = G*H over NW waves
// ---------------------------------------------------------------------------------------------------
template <int G, int NW, int GW, bool B3>
__global__ __launch_bounds__(NW * 64) void ft_rnn_bwd_persist_kernel(RnnBwdArgs a, Geom geo, float* xb, unsigned* cnt,
                                                                     unsigned* fault, unsigned xb_bytes) {
  __shared__ float red2[2 * NW * 16 * RLD];            // double-buffered by step parity (see the forward kernel)
  __shared__ int s_ok;
  int d, bgp, chunk, grp;
  if (!decode(geo, d, bgp, chunk, grp)) return;
  const int u0 = chunk * 16, b0 = bgp * MB;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int q = lane >> 4, l15 = lane & 15;
  const int H = a.H, T = a.T, K = G * H, nq = K / 4;
  const long ldg = (long)a.ND * K;
  const long ldo = (long)a.ND * H;
  const long grp_floats = (long)nq * MB * 4;
  const long par_floats = (long)2 * geo.nbg * grp_floats;
  const long base_floats = (long)grp * grp_floats;
  unsigned* mycnt = cnt + (long)grp * NSH * CSTRIDE;
  const int nprod = (geo.sig_per_wave ? 4 : 1) * geo.nchunks;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(xb, 0, (int)xb_bytes, 0x00020000);
  unsigned* myflags = cnt + (long)2 * geo.nbg * NSH * CSTRIDE + (long)grp * NFLAG;      // XCD-local mode (file header)
  unsigned* myxcc = cnt + (long)2 * geo.nbg * (NSH * CSTRIDE + NFLAG) + (long)grp * NXCC;
  const unsigned my_xcc = xcc_id() + 1u;
  __shared__ int s_local, s_fail;
  bool local = false;
  if (threadIdx.x == 0) {
    s_local = 0;
    s_fail = 0;
    __hip_atomic_store(myxcc + chunk, my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

  // ---- resident W_hh^T fragments: column l15 = unit u0+l15 ; K groups [g0,g1)
  const float* brow = a.whhT[d] + (long)(u0 + l15) * K;
  const int ngroups = K / 16;
  const int gpw = (ngroups + NW - 1) / NW;
  const int g0 = wave * gpw, g1 = min(ngroups, g0 + gpw);
  constexpr int BW = GW / 2;                        // B3: 32-k blocks per wave
  constexpr bool GRAN = B3 && BW <= 4;              // granule hand-off in XCD-local mode (see the forward kernel)
  const int nblk = K / 32;
  const int bpw = (nblk + NW - 1) / NW;
  const int kb0 = wave * bpw, kb1 = min(nblk, kb0 + bpw);
  float4 bv[B3 ? 1 : GW];
  bf16x8 bw[B3 ? BW : 1][3];
  if constexpr (!B3) {
#pragma unroll
    for (int c = 0; c < GW; ++c)
      bv[c] = (g0 + c < g1) ? *reinterpret_cast<const float4*>(brow + 16 * (g0 + c) + 4 * q) : make_float4(0, 0, 0, 0);
  } else {
#pragma unroll
    for (int c = 0; c < BW; ++c) {
      const bool in = kb0 + c < kb1;
      const float* p = brow + (in ? 32 * (kb0 + c) + 8 * q : 0);
      const float4 z = make_float4(0, 0, 0, 0);
      split8(in ? *reinterpret_cast<const float4*>(p) : z, in ? *reinterpret_cast<const float4*>(p + 4) : z, bw[c][0],
             bw[c][1], bw[c][2]);
    }
  }

  // XCD-local mode: this wave's operand slice is k-quads [q0, q0 + nq4) of d(gates), k = gate*H + unit; the signalling
  // wave (4*chunk' + j4') that produced a quad has flag index unit / 4 (it writes its 4 units for every gate)
  const int pq0 = B3 ? 8 * kb0 : 4 * g0, pnq = B3 ? 8 * max(kb1 - kb0, 0) : 4 * max(g1 - g0, 0);
  const unsigned* pollf = lane < pnq ? myflags + ((4 * (pq0 + lane)) % H) / 4 : nullptr;

  // ---- cell threads (first 256): wave j4 owns units 4*j4..4*j4+3 -> whole [16][4] exchange blocks, one per gate
  const int j4 = tid >> 6, ci = (tid >> 2) & 15, jj = tid & 3;
  const int cj = 4 * j4 + jj;
  const int cb = b0 + ci, cun = u0 + cj;
  const bool sthr = tid < 256;
  const bool cthr = sthr && cb < a.B;
  const int L = cthr ? clamp_len(a.lens, cb, T) : 0;
  float carry = 0.f;

  // saved activations / upstream gradient of one step (requested a step ahead)
  float gv[4] = {0.f, 0.f, 0.f, 0.f}, dov = 0.f, cc = 0.f, prev = 0.f;
  auto request = [&](int sn, float (&rgv)[4], float& rdo, float& rcc, float& rprev) {
#pragma unroll
    for (int g = 0; g < 4; ++g) rgv[g] = 0.f;
    rdo = 0.f; rcc = 0.f; rprev = 0.f;
    if (cthr && sn < L) {
      const int ct = d == 0 ? L - 1 - sn : sn;
      const int tprev = d == 0 ? ct - 1 : ct + 1;
      const bool has_prev = tprev >= 0 && tprev < L;
      const long o = ((long)ct * a.B + cb) * ldo + (long)d * H + cun;
      const long op = ((long)tprev * a.B + cb) * ldo + (long)d * H + cun;
      const float* gs = a.gates + (((long)ct * a.B + cb) * a.ND + d) * 4 * H + cun;
#pragma unroll
      for (int g = 0; g < 4; ++g) rgv[g] = gs[(long)g * H];
      rdo = a.dout[o];
      if (G == 3) {
        rprev = has_prev ? a.out[op] : 0.f;
      } else {
        rcc = a.cst[o];
        rprev = has_prev ? a.cst[op] : 0.f;
      }
    }
  };
  request(0, gv, dov, cc, prev);
  PROF_DECL;

  for (int s = 0; s < T; ++s) {
    PROF(0);
    const bool cact = cthr && s < L;
    const int ct = d == 0 ? L - 1 - s : s;
    float* red = red2 + (s & 1) * (NW * 16 * RLD);
    f32x4 acc[1];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[0][e] = 0.f;
    float4 av[B3 ? 1 : GW];
    float4 aw[B3 ? BW : 1][2];
    float ngv[4], ndo, ncc, nprev;
    const bool gl = local;              // protocol of THIS step's operands (the mode may change below, at s == 1)
    if (s > 0 && gl && GRAN && geo.gran) {
      // XCD-local, bf16-split: granules (see the forward kernel / file header)
      if constexpr (GRAN) {
        const unsigned want_hi = ((unsigned)s & 0xFFFFu) << 16;
        const unsigned gbase = (unsigned)(((long)2 * par_floats * 4) + ((long)((s - 1) & 1) * par_floats + base_floats) * 8);
        auto offs = [&](int c, unsigned& oa, unsigned& ob) {
          const long quad = 8 * (kb0 + c) + 2 * q;
          oa = gbase + (unsigned)((quad * MB + l15) * 4 * 8);
          ob = gbase + (unsigned)(((quad + 1) * MB + l15) * 4 * 8);
        };
        Gran gr[2];
        unsigned oa, ob;
        if (kb0 < kb1) {
          offs(0, oa, ob);
          gran_load(gr[0], rs, oa, ob);
        }
        bool ok = true;
#pragma unroll
        for (int c = 0; c < BW; ++c)
          if (kb0 + c < kb1) {
            if (c + 1 < BW && kb0 + c + 1 < kb1) {
              unsigned na, nb;
              offs(c + 1, na, nb);
              gran_load(gr[(c + 1) & 1], rs, na, nb);
            }
            offs(c, oa, ob);
            ok = gran_wait(gr[c & 1], rs, oa, ob, want_hi, geo.max_spins) && ok;
            bf16x8 a3[3];
            gran_unpack(gr[c & 1], a3);
            mfma6(a3, bw[c], acc[0]);
          }
        if (!ok && lane == 0) s_fail = 1;
      }
      request(s + 1, ngv, ndo, ncc, nprev);
    } else {
      if (s > 0) {
        if (gl) {        // XCD-local, flag words: every wave waits for the producers of its own slice
          if (!poll_flag(pollf, (unsigned)s, geo.max_spins) && lane == 0) s_fail = 1;
        } else {
          if (wave == 0) {
            const bool ok = wait_arrivals(mycnt, (unsigned)s, nprod, lane, geo.max_spins);
            if (s == 1 && ok && geo.local_ok) {
              const bool same = same_xcd(myxcc, geo.nchunks, my_xcc, lane);
              if (lane == 0) s_local = same;
            }
            if (lane == 0) s_ok = ok;
          }
          __syncthreads();
          if (!s_ok) {
            if (tid == 0) atomicExch(fault, 1u);
            return;
          }
          if (s == 1) {
            local = s_local != 0;
            if (threadIdx.x == 0 && chunk == 0) atomicAdd(fault + (local ? 8 : 9), 1u);    // statistics: groups per mode
          }
        }
        const long rbase = (long)((s - 1) & 1) * par_floats + base_floats;
        if constexpr (!B3) {
#pragma unroll
          for (int c = 0; c < GW; ++c)
            if (g0 + c < g1) {
              const long quad = 4 * (g0 + c) + q;
              av[c] = ld_sc1_b128(rs, (unsigned)((rbase + (quad * MB + l15) * 4) * 4));
            }
        } else {
#pragma unroll
          for (int c = 0; c < BW; ++c)
            if (kb0 + c < kb1) {
              const long quad = 8 * (kb0 + c) + 2 * q;
              aw[c][0] = ld_sc1_b128(rs, (unsigned)((rbase + (quad * MB + l15) * 4) * 4));
              aw[c][1] = ld_sc1_b128(rs, (unsigned)((rbase + ((quad + 1) * MB + l15) * 4) * 4));
            }
        }
      }
      request(s + 1, ngv, ndo, ncc, nprev);
      if (s > 0) {
        if constexpr (!B3) {
#pragma unroll
          for (int c = 0; c < GW; ++c)
            if (g0 + c < g1) mfma4(av[c], bv[c], acc[0]);
        } else {
#pragma unroll
          for (int c = 0; c < BW; ++c)
            if (kb0 + c < kb1) {
              bf16x8 a3[3];
              split8(aw[c][0], aw[c][1], a3[0], a3[1], a3[2]);
              mfma6(a3, bw[c], acc[0]);
            }
        }
      }
    }
    PROF(1);
    store_partials<1>(red, wave, lane, acc);
    PROF(2);
    __syncthreads();
    PROF(3);
    if (s_fail) {                                          // a wave's poll ran out (XCD-local mode): leave together
      if (tid == 0) atomicExch(fault, 1u);
      return;
    }

    float dgx[4] = {0.f, 0.f, 0.f, 0.f}, dgh2 = 0.f;
    if (cact) {
      float rec = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) rec += red[w * (16 * RLD) + ci * RLD + cj];
      if (G == 3) {
        const float dh = dov + rec + carry;
        const float r = gv[0], z = gv[1], n = gv[2], hn = gv[3];
        const float dz = dh * (prev - n) * z * (1.f - z);
        const float dn = dh * (1.f - z) * (1.f - n * n);
        const float dr = dn * hn * r * (1.f - r);
        dgx[0] = dr; dgx[1] = dz; dgx[2] = dn;
        dgh2 = dn * r;
        carry = dh * z;
      } else {
        const float dh = dov + rec;
        const float ig = gv[0], fg = gv[1], gg = gv[2], og = gv[3];
        const float tc = tanh_(cc, geo.fast);
        const float dc = dh * og * (1.f - tc * tc) + carry;
        dgx[0] = dc * gg * ig * (1.f - ig);
        dgx[1] = dc * prev * fg * (1.f - fg);
        dgx[2] = dc * ig * (1.f - gg * gg);
        dgx[G - 1] = dh * tc * og * (1.f - og);
        dgh2 = dgx[2];
        carry = dc * fg;
      }
    }
    PROF(4);
    if (cthr) {
      // exchange: k = g*H + cun -> block k/4 = (g*H + u0)/4 + j4, row ci, slot jj ; a finished item's rows are
      // published as zeros for the workgroups that still multiply them
      float* xw = xb + (long)(s & 1) * par_floats + base_floats;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float v = (G == 3 && g == 2) ? dgh2 : dgx[g];
        const long xe = (((long)(g * H + u0) / 4 + j4) * MB + ci) * 4 + jj;
        if (local && GRAN && geo.gran) continue;           // (granule mode: stored below, for every row)
        if (local) __hip_atomic_store(xw + xe, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        else __hip_atomic_store(xw + xe, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (local && GRAN && geo.gran) {
      if (sthr) {        // every (row, k) slot, every step: rows beyond the batch / finished items as zeros
        unsigned long long* gp = reinterpret_cast<unsigned long long*>(xb + 2 * par_floats) + (long)(s & 1) * par_floats +
                                 base_floats;
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const float v = cthr ? ((G == 3 && g == 2) ? dgh2 : dgx[g]) : 0.f;
          const long xe = (((long)(g * H + u0) / 4 + j4) * MB + ci) * 4 + jj;
          __hip_atomic_store(gp + xe, gran_make(v, (unsigned)(s + 1) & 0xFFFFu), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
      }
    } else if (geo.sig_per_wave) {
      if (sthr) {                                          // waves 0..3, wave-uniform
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
          if (local)
            __hip_atomic_store(myflags + 4 * chunk + j4, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          else
            __hip_atomic_fetch_add(mycnt + ((4 * chunk + j4) % NSH) * CSTRIDE, 1u, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0)
        __hip_atomic_fetch_add(mycnt + (chunk % NSH) * CSTRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    PROF(5);
    if (cact) {
      float* dx = a.dxp + ((long)ct * a.B + cb) * ldg + (long)d * K + cun;
#pragma unroll
      for (int g = 0; g < G; ++g) dx[(long)g * H] = dgx[g];
      if (G == 3) {
        float* dhh = a.dhp + ((long)ct * a.B + cb) * ldg + (long)d * K + cun;
        dhh[0] = dgx[0]; dhh[H] = dgx[1]; dhh[2 * H] = dgh2;
      }
    } else if (cthr) {       // finished item: zero gradient at t >= L (t = s in both directions)
      float* dx = a.dxp + ((long)s * a.B + cb) * ldg + (long)d * K + cun;
#pragma unroll
      for (int g = 0; g < G; ++g) dx[(long)g * H] = 0.f;
      if (G == 3) {
        float* dhh = a.dhp + ((long)s * a.B + cb) * ldg + (long)d * K + cun;
        dhh[0] = 0.f; dhh[H] = 0.f; dhh[2 * H] = 0.f;
      }
    }
    PROF(6);
#pragma unroll
    for (int g = 0; g < 4; ++g) gv[g] = ngv[g];
    dov = ndo; cc = ncc; prev = nprev;
    PROF(7);
  }
  PROF_DUMP(grp == 0 && chunk == 3 && wave == 0, 0);
  PROF_DUMP(grp == 0 && chunk == 3 && wave == 5, 1);
}

// ---------------------------------------------------------------------------------------------------
// backward, reduce-scatter form (bf16-split only): the same workgroup = 16 batch rows x 16 hidden units, but the
// recurrent product is formed on the PRODUCER side.  The all-gather form above makes every workgroup pull all G*H
// d(gates) of its 16 rows each step (LSTM-512: 128 KB per workgroup per step through one L1, plus the 3-way split of
// 32 K values) to contract them with its W_hh columns.  Here a workgroup contracts only ITS OWN 16 units' d(gates)
// ([16 rows] x [16 G k's], padded to 64) with its W_hh ROWS -- resident in registers, same 128 KB -- which yields a
// partial of ALL H outputs, and hands tile c' (16 units) of it to workgroup c':
//   exchange layout  xb[parity][group][consumer c'][producer p][unit 16][row 16]   (1 KB blocks; an MFMA result lane
//                    holds 4 rows of one unit = one 16-B write-through store per tile)
//   consumer: after the arrival wait its 256 cell threads each read ONE float per producer (p-th block + tid,
//             fully coalesced) and add them in producer order -> d(h) of (row, unit); cell math; the new d(gates)
//             go to LDS, every wave splits them (1 K values, not 32 K) and multiplies its output tiles.
// Per step a workgroup reads nchunks KB (32) instead of G*H*64 B (128 KB) and splits 1/32 of the values; the cross-
// workgroup hand-off (stores -> vmcnt(0) -> per-wave arrival -> poll -> loads) is the same as above.
// ---------------------------------------------------------------------------------------------------
template <int G, int NW, int NT>
__global__ __launch_bounds__(NW * 64) void ft_rnn_bwd_rs_kernel(RnnBwdArgs a, Geom geo, float* xb, unsigned* cnt,
                                                                unsigned* fault, unsigned xb_bytes) {
  constexpr int ALD = 68;                             // LDS row stride of the local d(gates) tile [16][64]
  constexpr int GWV = 4;                              // gathering waves = the cell waves
  constexpr int TPW = NW * NT / GWV;                  // incoming tiles per gathering wave (chunks per group = NW * NT)
  // PF: ONE wave beyond the cell waves (wave 4) fetches the NEXT step's saved activations for the cell threads, through
  // LDS.  Those loads come from HBM (441 MB of saved gates at the benchmark shape, ~4000 cycles) and vector-memory
  // operations complete in order -- per wave (vmcnt) and, as measured here, per CU: issued by the cell threads
  // themselves, a step ahead (until round 3: 7 four-byte loads per lane, 448 line requests per step), they sat in front
  // of the step's hand-off loads and of the vmcnt(0) before its flag store: 1.2 of the 4.0 us of an LSTM-512 step
  // (lab/rnn_phase_prof.py: 4.37 -> 2.98 us without them); issued by other waves at the END of a step they still held up
  // the cell waves' hand-off loads right behind them (4.2 us).  So: 7 coalesced 16-byte wave loads (112 line requests),
  // issued right after the step's hand-off loads have landed (barrier A), a whole MFMA phase before the next ones.
  constexpr bool PF = NW >= 8;
  constexpr int NIN = 7;                              // gates[4], dout, c (LSTM), previous h (GRU) / c (LSTM)
  __shared__ __attribute__((aligned(16))) float adg2[2 * 16 * ALD];   // double-buffered by step parity
  // per-wave partial sums of the incoming tiles (wave w adds the tiles of producers w*TPW .. w*TPW+TPW-1), [wave][unit][row]
  __shared__ __attribute__((aligned(16))) float rsum[GWV * 256];
  __shared__ float inb[PF ? 2 * NIN * 256 : 1];       // next-step inputs of the 256 cell threads, by step parity
  __shared__ int s_ok;
  int d, bgp, chunk, grp;
  if (!decode(geo, d, bgp, chunk, grp)) return;
  const int u0 = chunk * 16, b0 = bgp * MB;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int q = lane >> 4, l15 = lane & 15;
  const int H = a.H, T = a.T, K = G * H, P = geo.nchunks;
  const long ldg = (long)a.ND * K;
  const long ldo = (long)a.ND * H;
  const long grp_floats = (long)P * P * 256;
  const long par_floats = (long)2 * geo.nbg * grp_floats;
  const long base_floats = (long)grp * grp_floats;
  unsigned* mycnt = cnt + (long)grp * NSH * CSTRIDE;
  const int nprod = NW * P;                           // every wave signals its own stores
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(xb, 0, (int)xb_bytes, 0x00020000);
  unsigned* myflags = cnt + (long)2 * geo.nbg * NSH * CSTRIDE + (long)grp * NFLAG;      // XCD-local mode (file header)
  unsigned* myxcc = cnt + (long)2 * geo.nbg * (NSH * CSTRIDE + NFLAG) + (long)grp * NXCC;
  const unsigned my_xcc = xcc_id() + 1u;
  __shared__ int s_local, s_fail;
  bool local = false;
  if (threadIdx.x == 0) {
    s_local = 0;
    s_fail = 0;
    __hip_atomic_store(myxcc + chunk, my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

  // ---- resident W_hh rows of this workgroup's units, as B fragments: tile nt = consumer chunk wave*NT + nt, column
  //      l15 = output unit n; local k = g*16 + j  <->  W_hh[g*H + u0 + j][n] = whhT[n][g*H + u0 + j]
  bf16x8 bw[NT][2][3];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = (wave * NT + nt) * 16 + l15;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const int g = 2 * blk + (q >> 1), j0 = 8 * (q & 1);
      const bool in = g < G && n < H;
      const float* p = a.whhT[d] + (in ? (long)n * K + (long)g * H + u0 + j0 : 0);
      const float4 z = make_float4(0, 0, 0, 0);
      split8(in ? *reinterpret_cast<const float4*>(p) : z, in ? *reinterpret_cast<const float4*>(p + 4) : z,
             bw[nt][blk][0], bw[nt][blk][1], bw[nt][blk][2]);
    }
  }
  for (int i = tid; i < 2 * 16 * ALD; i += NW * 64) adg2[i] = 0.f;  // (GRU: k 48..63 stay zero)

  // XCD-local mode: cell wave w gathers the tiles the producer chunks p = w*TPW + i (i < TPW) computed for this
  // consumer chunk; such a tile came from producer wave chunk / NT -> flag NW*p + chunk/NT
  const unsigned* pollf =
      (wave < GWV && lane < TPW && wave * TPW + lane < P) ? myflags + NW * (wave * TPW + lane) + chunk / NT : nullptr;

  // ---- cell threads (first 256): thread = (unit cj, row ci), the order of an exchange block
  const int cj = tid >> 4, ci = tid & 15;
  const int cb = b0 + ci, cun = u0 + cj;
  const bool sthr = tid < 256;
  const bool cthr = sthr && cb < a.B;
  const int L = cthr ? clamp_len(a.lens, cb, T) : 0;
  float carry = 0.f;
  // ---- fetching lanes (PF: wave 4): lane = (row fr, unit quad fq) -> 16 bytes = units u0 + 4 fq .. + 3 of row fr
  const bool fwave = PF && wave == 4;
  const int fr = lane >> 2, fq = lane & 3;
  const int fb = b0 + fr;
  const int FL = (fwave && fb < a.B) ? clamp_len(a.lens, fb, T) : 0;

  float gv[4] = {0.f, 0.f, 0.f, 0.f}, dov = 0.f, cc = 0.f, prev = 0.f;
  auto request = [&](int sn, float (&rgv)[4], float& rdo, float& rcc, float& rprev) {
#pragma unroll
    for (int g = 0; g < 4; ++g) rgv[g] = 0.f;
    rdo = 0.f; rcc = 0.f; rprev = 0.f;
    if (cthr && sn < L) {
      const int ct = d == 0 ? L - 1 - sn : sn;
      const int tprev = d == 0 ? ct - 1 : ct + 1;
      const bool has_prev = tprev >= 0 && tprev < L;
      const long o = ((long)ct * a.B + cb) * ldo + (long)d * H + cun;
      const long op = ((long)tprev * a.B + cb) * ldo + (long)d * H + cun;
      const float* gs = a.gates + (((long)ct * a.B + cb) * a.ND + d) * 4 * H + cun;
#pragma unroll
      for (int g = 0; g < 4; ++g) rgv[g] = gs[(long)g * H];
      rdo = a.dout[o];
      if (G == 3) {
        rprev = has_prev ? a.out[op] : 0.f;
      } else {
        rcc = a.cst[o];
        rprev = has_prev ? a.cst[op] : 0.f;
      }
    }
  };
  // PF: the fetching wave holds the inputs of the NEXT step in fv (requested behind barrier A of a step, stored to inb
  // behind its own flag store at the end of that step); without PF the cell threads request their own, one step ahead
  float4 fv[NIN];
#pragma unroll
  for (int i = 0; i < NIN; ++i) fv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  auto fetch = [&](int sn) {
#pragma unroll
    for (int i = 0; i < NIN; ++i) fv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (fwave && sn < FL) {
      const int ct = d == 0 ? FL - 1 - sn : sn;
      const int tprev = d == 0 ? ct - 1 : ct + 1;
      const bool has_prev = tprev >= 0 && tprev < FL;
      const long o = ((long)ct * a.B + fb) * ldo + (long)d * H + u0 + 4 * fq;
      const long op = ((long)tprev * a.B + fb) * ldo + (long)d * H + u0 + 4 * fq;
      const float* gs = a.gates + (((long)ct * a.B + fb) * a.ND + d) * 4 * H + u0 + 4 * fq;
#pragma unroll
      for (int g = 0; g < 4; ++g) fv[g] = *reinterpret_cast<const float4*>(gs + (long)g * H);
      fv[4] = *reinterpret_cast<const float4*>(a.dout + o);
      if (G == 3) {
        if (has_prev) fv[6] = *reinterpret_cast<const float4*>(a.out + op);
      } else {
        fv[5] = *reinterpret_cast<const float4*>(a.cst + o);
        if (has_prev) fv[6] = *reinterpret_cast<const float4*>(a.cst + op);
      }
    }
  };
  auto publish = [&](int sn) {            // fetching wave: fv -> inb[parity of step sn][value][unit][row]
    if (fwave) {
      float* ib = inb + (sn & 1) * (NIN * 256) + (4 * fq) * 16 + fr;
#pragma unroll
      for (int i = 0; i < NIN; ++i) {
        ib[i * 256] = fv[i].x; ib[i * 256 + 16] = fv[i].y; ib[i * 256 + 32] = fv[i].z; ib[i * 256 + 48] = fv[i].w;
      }
    }
  };
  if constexpr (PF) {
    fetch(0);
    publish(0);
  } else {
    request(0, gv, dov, cc, prev);
  }
  __syncthreads();
  PROF_DECL;

  for (int s = 0; s < T; ++s) {
    PROF(0);
    const bool cact = cthr && s < L;
    const int ct = d == 0 ? L - 1 - s : s;
    float* adg = adg2 + (s & 1) * (16 * ALD);
    float rec = 0.f;
    if (s > 0) {
      if (local) {       // every wave waits for the producers of its own slice; a timeout is acted on at the next barrier
        if (!poll_flag(pollf, (unsigned)s, geo.max_spins) && lane == 0) s_fail = 1;
      } else {
        if (wave == 0) {
          const bool ok = wait_arrivals(mycnt, (unsigned)s, nprod, lane, geo.max_spins);
          if (s == 1 && ok && geo.local_ok) {
            const bool same = same_xcd(myxcc, P, my_xcc, lane);
            if (lane == 0) s_local = same;
          }
          if (lane == 0) s_ok = ok;
        }
        __syncthreads();
        if (!s_ok) {
          if (tid == 0) atomicExch(fault, 1u);
          return;
        }
        if (s == 1) {
          local = s_local != 0;
          if (threadIdx.x == 0 && chunk == 0) atomicAdd(fault + (local ? 8 : 9), 1u);    // statistics: groups per mode
        }
      }
      PROF(1);
      // The P incoming tiles, [unit 16][row 16] floats each.  Each cell wave takes TPW of them with one 16-byte load per
      // lane and tile (a whole 1-KB tile per wave instruction), adds them in producer order and leaves its partial in
      // LDS; the cell threads then add the four partials in wave order.
      if (wave < GWV) {
        const long rbase = (long)((s - 1) & 1) * par_floats + base_floats + (long)chunk * P * 256 + 4 * lane;
        float4 v[TPW];
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
          const int pp = wave * TPW + i;
          v[i] = pp < P ? ld_sc1_b128(rs, (unsigned)((rbase + (long)pp * 256) * 4)) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float4 acc4 = v[0];
#pragma unroll
        for (int i = 1; i < TPW; ++i) {
          acc4.x += v[i].x; acc4.y += v[i].y; acc4.z += v[i].z; acc4.w += v[i].w;
        }
        *reinterpret_cast<float4*>(rsum + wave * 256 + 4 * lane) = acc4;
      }
      __syncthreads();
      if (sthr) {
#pragma unroll
        for (int w = 0; w < GWV; ++w) rec += rsum[w * 256 + tid];
      }
    }
    if constexpr (PF) {
      fetch(s + 1);     // (behind barrier A: this step's hand-off loads have landed)
      if (sthr) {       // this step's saved activations, fetched during the previous step
        const float* ib = inb + (s & 1) * (NIN * 256) + tid;
#pragma unroll
        for (int g = 0; g < 4; ++g) gv[g] = ib[g * 256];
        dov = ib[4 * 256]; cc = ib[5 * 256]; prev = ib[6 * 256];
      }
    }
    float ngv[4] = {0.f, 0.f, 0.f, 0.f}, ndo = 0.f, ncc = 0.f, nprev = 0.f;
    if constexpr (!PF) request(s + 1, ngv, ndo, ncc, nprev);
#ifdef FT_RNN_PROF
    asm volatile("" : "+v"(rec));
#endif
    PROF(2);

    float dgx[4] = {0.f, 0.f, 0.f, 0.f}, dgh2 = 0.f;
    if (cact) {
      if (G == 3) {
        const float dh = dov + rec + carry;
        const float r = gv[0], z = gv[1], n = gv[2], hn = gv[3];
        const float dz = dh * (prev - n) * z * (1.f - z);
        const float dn = dh * (1.f - z) * (1.f - n * n);
        const float dr = dn * hn * r * (1.f - r);
        dgx[0] = dr; dgx[1] = dz; dgx[2] = dn;
        dgh2 = dn * r;
        carry = dh * z;
      } else {
        const float dh = dov + rec;
        const float ig = gv[0], fg = gv[1], gg = gv[2], og = gv[3];
        const float tc = tanh_(cc, geo.fast);
        const float dc = dh * og * (1.f - tc * tc) + carry;
        dgx[0] = dc * gg * ig * (1.f - ig);
        dgx[1] = dc * prev * fg * (1.f - fg);
        dgx[2] = dc * ig * (1.f - gg * gg);
        dgx[G - 1] = dh * tc * og * (1.f - og);
        dgh2 = dgx[2];
        carry = dc * fg;
      }
    }
    if (sthr) {         // a finished item's rows are published as zeros
#pragma unroll
      for (int g = 0; g < G; ++g) adg[ci * ALD + g * 16 + cj] = (G == 3 && g == 2) ? dgh2 : dgx[g];
    }
    PROF(3);
    __syncthreads();
    PROF(4);
    if (s_fail) {                                          // a wave's poll ran out (XCD-local mode): leave together
      if (tid == 0) atomicExch(fault, 1u);
      return;
    }

    if (s + 1 < T) {    // (the last step's product has no reader)
      bf16x8 a3[2][3];
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        const float* ap = adg + l15 * ALD + 32 * blk + 8 * q;
        split8(*reinterpret_cast<const float4*>(ap), *reinterpret_cast<const float4*>(ap + 4), a3[blk][0], a3[blk][1],
               a3[blk][2]);
      }
      PROF(5);
      const long wbase = (long)(s & 1) * par_floats + base_floats + (long)chunk * 256 + l15 * 16 + 4 * q;
      // (issuing the products term-major over the NT independent tiles instead of tile by tile -- no chain of dependent
      //  MFMAs -- measured 3.24 -> 3.37 us per step: the phase is not bound by the accumulator latency)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        mfma6(a3[0], bw[nt][0], acc);
        mfma6(a3[1], bw[nt][1], acc);
        const int cons = wave * NT + nt;
        if (cons < P) {
          u32x4 v = {__float_as_uint(acc[0]), __float_as_uint(acc[1]), __float_as_uint(acc[2]), __float_as_uint(acc[3])};
          const unsigned off = (unsigned)((wbase + (long)cons * P * 256) * 4);
          if (local) __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);       // plain: stays in the XCD's L2
          else __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 16);            // write-through (sc1)
        }
      }
      PROF(6);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      PROF(7);
      if (lane == 0) {
        if (local)
          __hip_atomic_store(myflags + NW * chunk + wave, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        else
          __hip_atomic_fetch_add(mycnt + ((NW * chunk + wave) % NSH) * CSTRIDE, 1u, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (cact) {
      float* dx = a.dxp + ((long)ct * a.B + cb) * ldg + (long)d * K + cun;
#pragma unroll
      for (int g = 0; g < G; ++g) dx[(long)g * H] = dgx[g];
      if (G == 3) {
        float* dhh = a.dhp + ((long)ct * a.B + cb) * ldg + (long)d * K + cun;
        dhh[0] = dgx[0]; dhh[H] = dgx[1]; dhh[2 * H] = dgh2;
      }
    } else if (cthr) {       // finished item: zero gradient at t >= L (t = s in both directions)
      float* dx = a.dxp + ((long)s * a.B + cb) * ldg + (long)d * K + cun;
#pragma unroll
      for (int g = 0; g < G; ++g) dx[(long)g * H] = 0.f;
      if (G == 3) {
        float* dhh = a.dhp + ((long)s * a.B + cb) * ldg + (long)d * K + cun;
        dhh[0] = 0.f; dhh[H] = 0.f; dhh[2 * H] = 0.f;
      }
    }
    if constexpr (PF) {
      // fetching wave, behind its flag store (the vmcnt(0) in front of it covered the fetch): hand the inputs of step
      // s + 1 to the cell threads
      publish(s + 1);
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) gv[g] = ngv[g];
      dov = ndo; cc = ncc; prev = nprev;
    }
    PROF(8);
  }
  PROF_DUMP(grp == 0 && chunk == 3 && wave == 0, 0);
  PROF_DUMP(grp == 0 && chunk == 3 && wave == 5, 1);
}

int g_persistent = -1;      // -1: take FT_RNN_PERSISTENT from the environment
bool persistent_enabled() {
  if (g_persistent < 0) {
    const char* e = getenv("FT_RNN_PERSISTENT");
    g_persistent = (e && e[0] == '0') ? 0 : 1;
  }
  return g_persistent == 1;
}

int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

int device_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
    if (cus <= 0) cus = 1;
  }
  return cus;
}

struct PersistWs {
  float* xb;
  unsigned* cnt;
  size_t sync_bytes, xb_bytes, total_bytes;
};
// workspace = [arrival counters | flag words | XCC ids | exchange buffer (fp32 parities, granule parities)]; the sync region (and, for the all-gather
// forms, the exchange buffer) is zeroed per call from the allocation's start (the fault word is NOT in here: it is the
// device-global g_rnn_fault, which launches never touch)
size_t sync_region_bytes(int ngrp) { return (size_t)ngrp * (NSH * CSTRIDE + NFLAG + NXCC) * sizeof(unsigned); }
PersistWs carve_ws(void* ws, int ngrp, int K) {
  PersistWs p;
  p.sync_bytes = sync_region_bytes(ngrp);
  p.cnt = (unsigned*)ws;
  // two fp32 parities, then the granule area of the XCD-local split kernels (two parities of 8-byte granules)
  p.xb_bytes = (size_t)3 * 2 * ngrp * (K / 4) * MB * 4 * sizeof(float);
  p.xb = (float*)((char*)ws + p.sync_bytes);
  p.total_bytes = p.sync_bytes + p.xb_bytes;
  return p;
}

// reduce-scatter backward: xb[parity][group][consumer][producer][256]
PersistWs carve_ws_rs(void* ws, int ngrp, int nchunks) {
  PersistWs p;
  p.sync_bytes = sync_region_bytes(ngrp);
  p.cnt = (unsigned*)ws;
  p.xb_bytes = (size_t)2 * ngrp * nchunks * nchunks * 256 * sizeof(float);
  p.xb = (float*)((char*)ws + p.sync_bytes);
  p.total_bytes = p.sync_bytes + p.xb_bytes;
  return p;
}

constexpr int MAX_DEV = 16;
unsigned g_max_spins = MAX_SPINS_DEFAULT;
long g_n_persistent = 0, g_n_refused = 0, g_n_waited = 0;

int current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) dev = 0;
  return dev;
}

// ---- admission: every workgroup of a persistent grid spins until the whole grid is resident, so two such grids on
// different streams (trunk recurrence on the step's main stream, a predictor's on the side stream) must BOTH fit
// beside each other, or each could hold the CUs the other is waiting for.  Workgroup i goes to XCD slot i % 8 and must
// find room on a CU of THAT XCD, and which physical XCD a launch's slot 0 is cannot be known, so the budget is per XCD
// for the worst case that every concurrent launch puts its fullest slot on the same XCD:
//     demand d = (workgroups of the fullest XCD slot) / (32 CUs x workgroups per CU, capped at 2)
// Launches of ONE stream execute in order (at most one of them resident), but the host enqueues far ahead of the
// device, so "unfinished" launches pile up per stream: a launch on stream s is admitted while
//     d + sum over the OTHER streams of (the largest d among that stream's unfinished launches) <= 0.75
// (a quarter is kept back: a CU left with room for half a workgroup helps nobody), or -- with nothing unfinished on
// other streams -- while d <= 1.  A launch that does not fit beside the others is NOT demoted to the per-step kernels
// (which round differently: results would depend on timing): its stream is made to wait, on the device, for the
// conflicting launches' events, and it runs persistent afterwards.  ft_rnn_note_join tells the bookkeeping about waits
// the caller inserted itself (the joined stream's earlier launches then precede everything the waiting stream does and
// no longer count against it).  One event per launch, retired by query.  Only a grid that does not fit the chip even
// alone (d > 1) runs per-step.  Non-persistent kernels (GEMMs, RCCL) only delay residency: they finish without us.
struct Flight {
  hipEvent_t ev;
  hipStream_t stream;
  double d;
  std::vector<hipStream_t> joined_by;      // streams that waited for `stream` after this launch was enqueued
};
std::mutex g_adm_mu;
std::vector<Flight> g_flights[MAX_DEV];
std::vector<hipEvent_t> g_ev_pool[MAX_DEV];

template <typename KernelT>
double xcd_demand(KernelT kernel, int block, int wgs_per_slot) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, 0) != hipSuccess || per_cu < 1) return -1.0;
  if (per_cu > 2) per_cu = 2;      // stay well inside what the dispatcher really admits
  return (double)wgs_per_slot / (per_cu * (device_cus() / 8.0));
}

// true: admitted (call admitted_launch_done after the launch); false: does not fit beside what other streams may run
bool admit(double d, hipStream_t stream) {
  if (d < 0.0 || !ft_rnn_fault_word()) return false;
  const int dev = current_device();
  std::lock_guard<std::mutex> lk(g_adm_mu);
  auto& fl = g_flights[dev];
  for (size_t i = 0; i < fl.size();) {
    if (hipEventQuery(fl[i].ev) == hipSuccess) {
      g_ev_pool[dev].push_back(fl[i].ev);
      fl[i] = std::move(fl.back());
      fl.pop_back();
    } else {
      ++i;
    }
  }
  (void)hipGetLastError();         // hipErrorNotReady of a pending event is not an error
  auto counts = [&](const Flight& f) {
    if (f.stream == stream) return false;
    for (hipStream_t j : f.joined_by)
      if (j == stream) return false;
    return true;
  };
  double others = 0.0;             // sum over other streams of their largest unfinished demand
  for (size_t i = 0; i < fl.size(); ++i) {
    if (!counts(fl[i])) continue;
    bool first = true;
    double mx = 0.0;
    for (size_t j = 0; j < fl.size(); ++j)
      if (fl[j].stream == fl[i].stream && counts(fl[j])) {
        if (j < i) first = false;
        if (fl[j].d > mx) mx = fl[j].d;
      }
    if (first) others += mx;
  }
  const double scale = env_int("FT_RNN_ADMIT_PCT", 100) / 100.0;
  if (d > 1.0 * scale) return false;             // does not fit the chip even alone: per-step kernels
  const bool fits = others == 0.0 || d + others <= 0.75 * scale;
  if (!fits) {
    // Not beside what the other streams may still be running: make THIS stream wait for those launches (device-side
    // wait on their events; the host does not block) and launch afterwards.  Always the same kernel, whatever the
    // timing -- a fallback to the per-step kernels here would make results depend on when the host happened to look
    // (they round differently: f32 MFMA against the exact bf16 split) -- and waits only follow enqueue order, so they
    // cannot form a cycle.
    for (Flight& f : fl)
      if (counts(f)) {
        (void)hipStreamWaitEvent(stream, f.ev, 0);
        f.joined_by.push_back(stream);
      }
    ++g_n_waited;
  }
  return true;
}

void admitted_launch_done(double d, hipStream_t stream) {
  const int dev = current_device();
  std::lock_guard<std::mutex> lk(g_adm_mu);
  hipEvent_t ev;
  if (!g_ev_pool[dev].empty()) {
    ev = g_ev_pool[dev].back();
    g_ev_pool[dev].pop_back();
  } else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
    return;
  }
  (void)hipEventRecord(ev, stream);
  g_flights[dev].push_back({ev, stream, d, {}});
  ++g_n_persistent;
}

// Grid layout of a persistent launch.  The 1-D grid is 8 x per: the decode hands XCD slot x the work items
// [x*per, (x+1)*per).  ALIGNED: per = whole (direction, batch group) groups, so that a group's hand-off stays inside
// one XCD (the XCD-local protocol then applies; surplus workgroups of the larger grid exit at once); successive
// launches start on different slots.  SPREAD: per = ceil(total / 8), groups straddle XCDs, agent-scope protocol only.
// The aligned layout is tried first; the spread one only runs when the aligned per-XCD demand exceeds a whole XCD (admit()
// makes a launch WAIT for conflicting ones rather than refuse it, so a demand <= 1 is always taken as aligned).
struct Layout {
  int grid, per, xcd_off, aligned;
};
Layout layout_aligned(const Geom& geo) {
  static int rotor = 0;
  Layout l = {0, 0, 0, 0};
  if (!geo.xcd_aware || !geo.sig_per_wave || !env_int("FT_RNN_XCDALIGN", 1) || geo.nchunks > NXCC) return l;
  const int ngroups = geo.total / geo.nchunks;
  l.per = geo.nchunks * ft_cdiv(ngroups, 8);
  l.grid = 8 * l.per;
  l.xcd_off = rotor & 7;
  l.aligned = 1;
  rotor += ngroups < 8 ? ngroups : 8;
  return l;
}
Layout layout_spread(const Geom& geo) {
  Layout l = {0, 0, 0, 0};
  l.per = ft_cdiv(geo.total, 8);
  l.grid = 8 * l.per;
  return l;
}

// picks the layout, checks admission, fills geo; -1.0 = not admitted in any layout
template <typename KernelT>
double plan_launch(KernelT kernel, int block, Geom& geo, int& grid, hipStream_t stream) {
  const Layout cand[2] = {layout_aligned(geo), layout_spread(geo)};
  for (int i = 0; i < 2; ++i) {
    if (cand[i].grid == 0) continue;
    const double d = xcd_demand(kernel, block, cand[i].per);
    const bool ok = d <= 1.0 && admit(d, stream);
    if (env_int("FT_RNN_DEBUG", 0))
      fprintf(stderr, "[ft_rnn] block %d total %d nchunks %d: %s layout, %d workgroups per XCD slot, demand %.3f -> %s\n", block,
              geo.total, geo.nchunks, cand[i].aligned ? "aligned" : "spread", cand[i].per, d, ok ? "admitted" : "refused");
    if (!ok) continue;
    grid = cand[i].grid;
    geo.xcd_off = cand[i].xcd_off;
    geo.local_ok = cand[i].aligned && env_int("FT_RNN_LOCAL", 1);
    geo.gran = env_int("FT_RNN_GRANULES", 1);
    geo.fast = env_int("FT_RNN_FAST_CELL", 1);
    return d;
  }
  ++g_n_refused;
  return -1.0;
}

// fill_probe (ft_rnn_fwd_xcd_fill): no launch -- report the share of an XCD's CUs the aligned layout of this kernel holds
double* g_fill_probe = nullptr;

template <int G, int NW, bool B3, int UB, int BC>
int launch_fwd_persist(const RnnFwdArgs& a, Geom geo, const PersistWs& p, hipStream_t stream) {
  if (g_fill_probe) {
    const int ngroups = geo.total / geo.nchunks;
    *g_fill_probe = xcd_demand(ft_rnn_fwd_persist_kernel<G, NW, B3, UB, BC>, NW * 64, geo.nchunks * ft_cdiv(ngroups, 8));
    return FT_OK;
  }
  int grid = 0;
  const double cus = plan_launch(ft_rnn_fwd_persist_kernel<G, NW, B3, UB, BC>, NW * 64, geo, grid, stream);
  if (cus < 0.0) return -1;
  if (BC > 2 && !env_int("FT_RNN_GRAN4", 1)) geo.gran = 0;
  (void)hipMemsetAsync(p.cnt, 0, p.total_bytes, stream);
  hipLaunchKernelGGL((ft_rnn_fwd_persist_kernel<G, NW, B3, UB, BC>), dim3(grid), dim3(NW * 64), 0, stream, a, geo, p.xb,
                     p.cnt, ft_rnn_fault_word(), (unsigned)p.xb_bytes);
  admitted_launch_done(cus, stream);
  return ft_check_launch("rnn_fwd_persistent");
}

template <int G>
int fwd_persistent(RnnFwdArgs a, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (!persistent_enabled() || !ws || a.T < 2) return -1;
  const int H = a.H, B = a.B;
  if (!a.vec || H % 16 != 0) return -1;
  const int ngroups = H / 16;
  int NW = ngroups > 16 ? 8 : 4;
  // The 512-wide LSTM: a (direction, batch group) group must fit ONE XCD to hand over through its L2, i.e. at most 32
  // workgroup-CUs.  Two forms do: (a) 16 units per 8-wave workgroup (32 workgroups, one per CU; 2 resident k-blocks per
  // wave, granule hand-off) -- every CU then pulls the group's h once per step instead of twice (two 8-unit workgroups
  // per CU: 64 KB per CU and step through one 64 B/clk path) -- and (b) 8 units per 4-wave workgroup (64 workgroups, two
  // per CU; 4 resident blocks per wave, flag words).  Round 2 measured (a) at 5.65 us per step against 3.77: it spilled
  // 57 registers.  With ONE MFMA site per (block, tile) for all hand-off protocols (round 3) it spills 4 and runs 3.51
  // against 3.88 in isolation, 22.8 -> 22.3 ms per train step (profiles/r03_rnn_ab.txt).  FT_RNN_FWD_UB16=0: form (b).
  const bool wide = NW == 8 && G == 4 && H / 8 > 32 && H % 32 == 0 && ft_cdiv(H / 32, 8) <= 2 && env_int("FT_RNN_B3", 1) &&
                    env_int("FT_RNN_FWD_UB16", 1);
  if (!wide && NW == 8 && G == 4 && H % 32 == 0 && ft_cdiv(H / 32, 4) <= 4 && env_int("FT_RNN_FWD_NW4", 1)) NW = 4;
  if (ft_cdiv(ngroups, NW) > GCH) return -1;
  const int bpw = H % 32 == 0 ? ft_cdiv(H / 32, NW) : 99;                 // 32-k blocks per wave in the split form
  const bool b3 = bpw <= 4 && env_int("FT_RNN_B3", 1);                    // matmul on the bf16 pipe (exact split)
  // GRU, H = 256 (the trunk's two GRUs; the postnet's runs 841 steps): 8 waves with ONE resident k-block each and 16
  // hidden units per workgroup -- 16 producers per group instead of 32, half the operand bytes per wave.  Same box,
  // us per step at T = 841 (lab/gru256_ab.py): 4 waves x 2 blocks x 8 units 2.03 | 8 x 1 x 8: 2.03-2.07 | 4 x 2 x 16:
  // 2.04 | 8 x 1 x 16: 1.78-1.81 | 8 x 1 x 32: 2.30.  FT_RNN_GRU_WIDE=0: the 4-wave 8-unit form.
  const bool gru_wide = G == 3 && H == 256 && b3 && NW == 4 && env_int("FT_RNN_GRU_WIDE", 1);
  Geom geo;
  geo.nchunks = H / ((wide || gru_wide) ? 16 : 8);
  geo.nbg = ft_cdiv(B, MB);
  geo.total = 2 * geo.nbg * geo.nchunks;
  geo.xcd_aware = env_int("FT_RNN_XCDMAP", 1);
  geo.sig_per_wave = env_int("FT_RNN_SIG", 1);
  geo.max_spins = g_max_spins;
  PersistWs p = carve_ws(ws, 2 * geo.nbg, H);
  if (ws_bytes < p.total_bytes || p.xb_bytes >= (1ull << 31)) return -1;
  geo.xcd_off = 0;
  geo.local_ok = 0;
  geo.gran = 0;
  geo.fast = 0;
  a.s = 0;
  if (wide) return launch_fwd_persist<G, 8, true, 16, 2>(a, geo, p, stream);
  if constexpr (G == 3) {
    if (gru_wide) return launch_fwd_persist<G, 8, true, 16, 1>(a, geo, p, stream);
  }
  if (b3) {
    if (NW == 8) return bpw <= 1 ? launch_fwd_persist<G, 8, true, 8, 1>(a, geo, p, stream)
                                 : launch_fwd_persist<G, 8, true, 8, 2>(a, geo, p, stream);     // (NW = 8: bpw <= 2)
    if (bpw <= 1) return launch_fwd_persist<G, 4, true, 8, 1>(a, geo, p, stream);
    if (bpw <= 2) return launch_fwd_persist<G, 4, true, 8, 2>(a, geo, p, stream);
    return launch_fwd_persist<G, 4, true, 8, 4>(a, geo, p, stream);
  }
  return NW == 8 ? launch_fwd_persist<G, 8, false, 8, 1>(a, geo, p, stream)
                 : launch_fwd_persist<G, 4, false, 8, 1>(a, geo, p, stream);
}

template <int G, int NW, int GW, bool B3>
int launch_bwd_persist(const RnnBwdArgs& a, Geom geo, const PersistWs& p, hipStream_t stream) {
  if (g_fill_probe) {
    *g_fill_probe = xcd_demand(ft_rnn_bwd_persist_kernel<G, NW, GW, B3>, NW * 64,
                               geo.nchunks * ft_cdiv(geo.total / geo.nchunks, 8));
    return FT_OK;
  }
  int grid = 0;
  const double cus = plan_launch(ft_rnn_bwd_persist_kernel<G, NW, GW, B3>, NW * 64, geo, grid, stream);
  if (cus < 0.0) return -1;
  (void)hipMemsetAsync(p.cnt, 0, p.total_bytes, stream);
  hipLaunchKernelGGL((ft_rnn_bwd_persist_kernel<G, NW, GW, B3>), dim3(grid), dim3(NW * 64), 0, stream, a, geo, p.xb, p.cnt,
                     ft_rnn_fault_word(), (unsigned)p.xb_bytes);
  admitted_launch_done(cus, stream);
  return ft_check_launch("rnn_bwd_persistent");
}

template <int G, int NW, int NT>
int launch_bwd_rs(const RnnBwdArgs& a, Geom geo, const PersistWs& p, hipStream_t stream) {
  if (g_fill_probe) {
    *g_fill_probe = xcd_demand(ft_rnn_bwd_rs_kernel<G, NW, NT>, NW * 64, geo.nchunks * ft_cdiv(geo.total / geo.nchunks, 8));
    return FT_OK;
  }
  int grid = 0;
  const double cus = plan_launch(ft_rnn_bwd_rs_kernel<G, NW, NT>, NW * 64, geo, grid, stream);
  if (cus < 0.0) return -1;
  // only the sync region (counters, flags, XCC ids) needs zeroing: every exchange block is written before it is read
  (void)hipMemsetAsync(p.cnt, 0, p.sync_bytes, stream);
  hipLaunchKernelGGL((ft_rnn_bwd_rs_kernel<G, NW, NT>), dim3(grid), dim3(NW * 64), 0, stream, a, geo, p.xb, p.cnt,
                     ft_rnn_fault_word(), (unsigned)p.xb_bytes);
  admitted_launch_done(cus, stream);
  return ft_check_launch("rnn_bwd_persistent_rs");
}

// reduce-scatter form: H/16 output tiles over NW waves, NT tiles each; -1 if it does not apply
template <int G>
int bwd_persistent_rs(RnnBwdArgs a, const Geom& geo, void* ws, size_t ws_bytes, hipStream_t stream) {
  const int tiles = a.H / 16;
  if (!env_int("FT_RNN_BWD_RS", 1) || !env_int("FT_RNN_B3", 1) || !geo.sig_per_wave) return -1;
  PersistWs p = carve_ws_rs(ws, 2 * geo.nbg, geo.nchunks);
  if (ws_bytes < p.total_bytes || p.xb_bytes >= (1ull << 31)) return -1;
  // measured (lab/rnn_step_us.py, B = 32, us/step, all-gather -> reduce-scatter): LSTM H=512 5.26 -> 4.67;
  // GRU H=256 3.15 -> 3.60, H=128 2.87 -> 2.95, H=64 2.70 -> 2.73: the form pays once the gathered operand is large
  // (G*H >= 1024 values per row); FT_RNN_BWD_RS=2 forces it wherever it applies
  const bool force = env_int("FT_RNN_BWD_RS", 1) == 2;
  if (!force && G * a.H < 1024) return -1;
  switch (tiles) {
    case 4: return launch_bwd_rs<G, 4, 1>(a, geo, p, stream);
    case 8: return launch_bwd_rs<G, 8, 1>(a, geo, p, stream);
    case 16: return launch_bwd_rs<G, 8, 2>(a, geo, p, stream);
    case 32: return launch_bwd_rs<G, 8, 4>(a, geo, p, stream);
    default: return -1;
  }
}

template <int G>
int bwd_persistent(RnnBwdArgs a, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (!persistent_enabled() || !ws || a.T < 2) return -1;
  const int H = a.H, B = a.B, K = G * H;
  if (!a.vec || H % 16 != 0) return -1;
  const int ngroups = K / 16;
  // waves x resident K groups per wave: 8 waves keep up to 16 groups each (256 registers per lane at 2 waves per
  // SIMD); 16 waves would cap a lane at 128 registers and spill the LSTM-512 working set
  const int NW = ngroups > 128 ? 16 : (ngroups > 16 ? 8 : 4);
  const int GW = ngroups > 64 ? 16 : 8;
  if (ft_cdiv(ngroups, NW) > GW) return -1;
  Geom geo;
  geo.nchunks = H / 16;
  geo.nbg = ft_cdiv(B, MB);
  geo.total = 2 * geo.nbg * geo.nchunks;
  geo.xcd_aware = env_int("FT_RNN_XCDMAP", 1);
  geo.sig_per_wave = env_int("FT_RNN_SIG", 1);
  geo.max_spins = g_max_spins;
  geo.xcd_off = 0;
  geo.local_ok = 0;
  geo.gran = 0;
  geo.fast = 0;
  a.s = 0;
  {
    const int rc = bwd_persistent_rs<G>(a, geo, ws, ws_bytes, stream);
    if (rc != -1) return rc;
  }
  PersistWs p = carve_ws(ws, 2 * geo.nbg, K);
  if (ws_bytes < p.total_bytes || p.xb_bytes >= (1ull << 31)) return -1;
  // (GRU H = 256, K = 768 on 16 waves with 2 of the 24 k-blocks each instead of 8 x 3: 11.9 us per step against 2.45 --
  //  a 1024-thread workgroup caps a lane at 128 registers; 12 waves x 2 blocks: 5.1 us; lab/gru256_ab.py)
  if (K % 32 == 0 && ft_cdiv(K / 32, NW) <= GW / 2 && env_int("FT_RNN_B3", 1)) {
    if (NW == 16) return launch_bwd_persist<G, 16, 16, true>(a, geo, p, stream);
    if (NW == 8) return GW == 16 ? launch_bwd_persist<G, 8, 16, true>(a, geo, p, stream)
                                 : launch_bwd_persist<G, 8, 8, true>(a, geo, p, stream);
    return launch_bwd_persist<G, 4, 8, true>(a, geo, p, stream);
  }
  if (NW == 16) return launch_bwd_persist<G, 16, 16, false>(a, geo, p, stream);
  if (NW == 8) return GW == 16 ? launch_bwd_persist<G, 8, 16, false>(a, geo, p, stream)
                               : launch_bwd_persist<G, 8, 8, false>(a, geo, p, stream);
  return launch_bwd_persist<G, 4, 8, false>(a, geo, p, stream);
}

}  // namespace

// device address of this device's sticky fault word (cached per device); nullptr on a HIP error
unsigned* ft_rnn_fault_word() {
  static unsigned* cache[MAX_DEV] = {};
  const int dev = current_device();
  if (!cache[dev]) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_rnn_fault)) != hipSuccess) return nullptr;
    cache[dev] = (unsigned*)p;
  }
  return cache[dev];
}

// Share of ONE XCD's CUs that the persistent forward of this shape occupies (aligned layout), or -1 if it would not run
// persistent.  A launch that fills whole XCDs (the 512-wide LSTM: 1.0) stops EVERY other kernel from being dispatched
// while it is resident -- workgroups are dealt to the XCDs round-robin and the dispatcher waits at the first one whose
// XCD has no room (profiles/r03_xcd_dispatch_probe.txt) -- so nothing it waits for may be launched beside it.
double ft_rnn_fwd_xcd_fill(int G, int B, int T, int H, void* ws, size_t ws_bytes) {
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  RnnFwdArgs a;
  memset(&a, 0, sizeof(a));
  a.B = B; a.T = T; a.H = H; a.ND = 2; a.Bld = B; a.vec = H % 4 == 0;
  double fill = -1.0;
  g_fill_probe = &fill;
  const int rc = G == 3 ? fwd_persistent<3>(a, ws, ws_bytes, nullptr) : fwd_persistent<4>(a, ws, ws_bytes, nullptr);
  g_fill_probe = nullptr;
  return rc == FT_OK ? fill : -1.0;
}

// the same for the persistent backward of this shape (what trainer.TrainStep._predictors_first budgets with)
double ft_rnn_bwd_xcd_fill(int G, int B, int T, int H, void* ws, size_t ws_bytes) {
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  RnnBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.B = B; a.T = T; a.H = H; a.ND = 2; a.vec = H % 4 == 0;
  double fill = -1.0;
  g_fill_probe = &fill;
  const int rc = G == 3 ? bwd_persistent<3>(a, ws, ws_bytes, nullptr) : bwd_persistent<4>(a, ws, ws_bytes, nullptr);
  g_fill_probe = nullptr;
  return rc == FT_OK ? fill : -1.0;
}

int ft_rnn_fwd_persistent(int G, RnnFwdArgs a, void* ws, size_t ws_bytes, hipStream_t stream) {
  return G == 3 ? fwd_persistent<3>(a, ws, ws_bytes, stream) : fwd_persistent<4>(a, ws, ws_bytes, stream);
}
int ft_rnn_bwd_persistent(int G, RnnBwdArgs a, void* ws, size_t ws_bytes, hipStream_t stream) {
  return G == 3 ? bwd_persistent<3>(a, ws, ws_bytes, stream) : bwd_persistent<4>(a, ws, ws_bytes, stream);
}

extern "C" {

size_t ft_rnn_workspace(int gates, int B, int H) {
  if (gates < 3 || gates > 4 || B <= 0 || H <= 0 || H % 16 != 0) return 0;
  const int ngrp = 2 * ft_cdiv(B, MB);
  const PersistWs f = carve_ws(nullptr, ngrp, H);
  const PersistWs b = carve_ws(nullptr, ngrp, gates * H);
  const PersistWs r = carve_ws_rs(nullptr, ngrp, H / 16);
  size_t m = f.total_bytes > b.total_bytes ? f.total_bytes : b.total_bytes;
  return r.total_bytes > m ? r.total_bytes : m;
}

/* share of ONE XCD's CUs the persistent recurrence of this shape holds while it runs (gates 3 | 4; backward != 0: the
 * BPTT kernel), in percent; -1: it would not run persistent.  100 = whole XCDs: no other kernel is dispatched beside it */
int ft_rnn_xcd_fill_pct(int gates, int backward, int B, int T, int H) {
  if ((gates != 3 && gates != 4) || B <= 0 || H <= 0) return -1;
  void* const ws = (void*)(uintptr_t)256;            // never dereferenced by the probe
  const double f = backward ? ft_rnn_bwd_xcd_fill(gates, B, T, H, ws, (size_t)-1)
                            : ft_rnn_fwd_xcd_fill(gates, B, T, H, ws, (size_t)-1);
  return f < 0.0 ? -1 : (int)(f * 100.0 + 0.5);
}
int ft_rnn_admit_budget_pct(void) { return (int)(0.75 * env_int("FT_RNN_ADMIT_PCT", 100) + 0.5); }

int ft_rnn_set_persistent(int enabled) {
  int old = persistent_enabled() ? 1 : 0;
  g_persistent = enabled ? 1 : 0;
  return old;
}

int ft_rnn_set_max_spins(int max_spins) {
  const int old = (int)g_max_spins;
  g_max_spins = max_spins > 0 ? (unsigned)max_spins : (max_spins < 0 ? 0u : MAX_SPINS_DEFAULT);
  return old;
}

int ft_rnn_note_join(void* waiting_stream, void* joined_stream) {
  const int dev = current_device();
  std::lock_guard<std::mutex> lk(g_adm_mu);
  for (Flight& f : g_flights[dev])
    if (f.stream == (hipStream_t)joined_stream) f.joined_by.push_back((hipStream_t)waiting_stream);
  return FT_OK;
}

int ft_rnn_counters(long* persistent_launches, long* refused_launches) {
  if (persistent_launches) *persistent_launches = g_n_persistent;
  if (refused_launches) *refused_launches = g_n_refused;
  return FT_OK;
}

int ft_rnn_waited_launches(void) { return (int)g_n_waited; }

int ft_rnn_mode_counts(long* xcd_local_groups, long* agent_scope_groups) {
  unsigned* w = ft_rnn_fault_word();
  unsigned v[2] = {0, 0};
  if (!w || hipDeviceSynchronize() != hipSuccess ||
      hipMemcpy(v, w + 8, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) {
    ft_set_error("rnn_mode_counts: HIP error while reading the counters");
    return FT_ERR_HIP;
  }
  if (xcd_local_groups) *xcd_local_groups = v[0];
  if (agent_scope_groups) *agent_scope_groups = v[1];
  return FT_OK;
}

#ifdef FT_RNN_PROF
int ft_rnn_prof_read(unsigned long long* out, int n) {
  return hipDeviceSynchronize() == hipSuccess &&
                 hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * (n < 64 ? n : 64)) == hipSuccess
             ? FT_OK
             : FT_ERR_HIP;
}
#endif

int ft_rnn_status(int clear) {
  unsigned* w = ft_rnn_fault_word();
  unsigned flag = 0;
  if (!w || hipDeviceSynchronize() != hipSuccess ||
      hipMemcpy(&flag, w, sizeof(flag), hipMemcpyDeviceToHost) != hipSuccess) {
    ft_set_error("rnn_status: HIP error while reading the fault word");
    return FT_ERR_HIP;
  }
  if (flag != 0) {
    if (clear) (void)hipMemset(w, 0, sizeof(flag));
    ft_set_error("persistent recurrence timed out waiting for another workgroup (grid not co-resident?); "
                 "set FT_RNN_PERSISTENT=0 to use the per-step kernels");
    return FT_ERR_HIP;
  }
  return FT_OK;
}

}  // extern "C"
