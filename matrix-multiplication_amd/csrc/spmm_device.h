// Device-side pieces shared by the SpMM translation units (spmm_csr.hip: one-pass kernels; spmm_panels.hip: column-panel
// passes; spmm_long.hip: rows beyond the long-row threshold; spmm_sched.hip: scheduled launches).  Not installed.
// Everything here has internal linkage (an anonymous namespace per including unit).
#ifndef MI_SPMM_DEVICE_H_
#define MI_SPMM_DEVICE_H_

#include "mi_common.h"
#include "mi_lanes.h"
#include "spmm_internal.h"

namespace {

using mi::f32x4;

__device__ __forceinline__ f32x4 fma4(float a, f32x4 x, f32x4 acc) {
  acc.x = __builtin_fmaf(a, x.x, acc.x);
  acc.y = __builtin_fmaf(a, x.y, acc.y);
  acc.z = __builtin_fmaf(a, x.z, acc.z);
  acc.w = __builtin_fmaf(a, x.w, acc.w);
  return acc;
}

// ---------------------------------------------------------------------------
// Rows beyond the long-row threshold (see "Skewed matrices" below) are skipped by the kernels of this file and
// LISTED by them on the way: the wave (or lane group) that meets such a row appends it to the list in the
// caller's workspace, so a product costs no separate scan of rowptr — main kernel + one follow-up launch that
// sums the listed rows (or finds the list empty and exits).
// Workspace (ints): [0] rows listed, [1] workgroup slots handed out, [2] partial-row slots handed out,
// [3] follow-up workgroups done; then cap_e entries of 8 ints {row, slot base, S, partial base, arrivals, –, –, –};
// then cap_s slot → entry indices; then (16-B aligned) cap_p × N floats of partial rows.
// ---------------------------------------------------------------------------
constexpr int kLongRow = mi::kLongRowThreshold;
constexpr int kLongChunk = 1024;
constexpr int kLongWaves = 16;
constexpr int kLongSplitShift = 15;  // one workgroup per 32768 non-zeros of a row …
constexpr int kLongSplitMax = 128;   // … up to 128 workgroups
constexpr int kLongEnt = 8;          // ints per list entry

using mi::LongArg;  // spmm_internal.h

// ---------------------------------------------------------------------------
// Structure-aware panels without a host round trip (round 5).  The L2-level panel plans are chosen from the SHAPE alone; on a
// matrix whose rows gather from a narrow band of B (banded / block-diagonal structure) one pass is already served by the
// caches and P panels only add passes (tools/plan_grid.py --pattern band1k: up to 2 × behind one pass).  When the caller
// gave a workspace, a probe launch ahead of the passes looks at kAdaptSlots windows of kAdaptWindow consecutive rows
// (first / last four columns of every fourth row) and writes one verdict per window: "the rows of B this window touches span
// ≤ 0.4 of B and ≤ 128 MiB" (uniform columns span all of B; a band of ± 1 K columns a few per cent of it).  Every workgroup of the panel kernels reads the verdicts (uniform scalar loads): with
// ≥ 7/8 of the windows local the FIRST pass takes every column (and the bias) and the other passes return at once — the
// one-pass chain, the same bits, decided on the device: no read-back, graph-capturable, the launches stay as they were.
// ---------------------------------------------------------------------------
constexpr int kAdaptSlots = mi::kAdaptVerdicts;
constexpr int kAdaptWindow = 2048;

__device__ __forceinline__ bool adapt_says_local(const int* __restrict__ verdicts) {  // wave-uniform
  int s = 0;
#pragma unroll
  for (int i = 0; i < kAdaptSlots; ++i) s += __builtin_amdgcn_readfirstlane(verdicts[i]);
  return 8 * s >= 7 * kAdaptSlots;
}

// One lane per long row.  The order of the list does not matter: every slot is a fixed function of (row, g).
__device__ __forceinline__ void long_list_append(const LongArg& la, int row, int len) {
  int* ws = la.ws;
  if (ws == nullptr) return;
  int S = len >> kLongSplitShift;
  S = S < 1 ? 1 : (S > kLongSplitMax ? kLongSplitMax : S);
  const int e = atomicAdd(&ws[0], 1);
  const int sb = atomicAdd(&ws[1], S);
  const int pb = S > 1 ? atomicAdd(&ws[2], S) : 0;
  // the caps hold for any rowptr consistent with nnz; a lying rowptr must not write out of bounds, and an
  // entry below the count must never hold garbage (its S = 0 makes every slot that points at it a no-op)
  if (e >= la.cap_e) return;
  const bool fits = sb + S <= la.cap_s && (S <= 1 || pb + S <= la.cap_p);
  int* ent = ws + 4 + kLongEnt * (long)e;
  ent[0] = row;
  ent[1] = sb;
  ent[2] = fits ? S : 0;
  ent[3] = pb;
  ent[4] = 0;  // workgroups of this row that have delivered their partial sums
  if (!fits) return;
  int* owner = ws + 4 + kLongEnt * (long)la.cap_e;
  for (int g = 0; g < S; ++g) owner[sb + g] = e;
}

// The last end − p < U entries of a row (wave-uniform p, end; col / val through the scalar unit): their gathers are
// issued TOGETHER — blocks of U/2, U/4, … 1 entries, every block's loads before the first block's FMAs — instead of
// one entry at a time with its latency exposed (a row of 20 entries at U = 8 used to end in four dependent trips to
// memory, as long as its two full batches took).  The FMAs run in entry order: the chain is unchanged.
template <int T, int U>
__device__ __forceinline__ void row_tail(const int* __restrict__ col, const float* __restrict__ val, const float* Bl,
                                         long ldb, int p, int end, f32x4 (&acc)[T]) {
  const int rem = end - p;  // 0 … U-1
  if (rem <= 0) return;
  f32x4 x[U > 1 ? U - 1 : 1][T];
  float v[U > 1 ? U - 1 : 1];
  int q = p;  // (compile-time slot of each block: U/2 entries at slots [0, U/2), U/4 at [U/2, 3U/4), …)
  mi::static_for<7>([&](auto k_) {  // blocks of U >> 1, U >> 2, …
    constexpr int blk = U >> (decltype(k_)::value + 1);
    if constexpr (blk >= 1) {
      constexpr int slot = U - 2 * blk;  // Σ of the larger blocks = U − 2·blk
      if (rem & blk) {
#pragma unroll
        for (int u = 0; u < blk; ++u) {
          const int c = col[q + u];
          v[slot + u] = val[q + u];
          const float* src = Bl + (long)c * ldb;
#pragma unroll
          for (int t = 0; t < T; ++t) x[slot + u][t] = *reinterpret_cast<const f32x4*>(src + t * 256);
        }
        q += blk;
      }
    }
  });
  mi::static_for<7>([&](auto k_) {
    constexpr int blk = U >> (decltype(k_)::value + 1);
    if constexpr (blk >= 1) {
      constexpr int slot = U - 2 * blk;
      if (rem & blk) {
#pragma unroll
        for (int u = 0; u < blk; ++u)
#pragma unroll
          for (int t = 0; t < T; ++t) acc[t] = fma4(v[slot + u], x[slot + u][t], acc[t]);
      }
    }
  });
}

template <int VEC>
struct Vec;
template <>
struct Vec<4> {
  static constexpr int width = 4;
  typedef f32x4 type;
  static __device__ __forceinline__ type zero() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
  static __device__ __forceinline__ type load(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  static __device__ __forceinline__ void store(float* p, type v) {
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
  }
  static __device__ __forceinline__ type fma(float a, type x, type acc) { return fma4(a, x, acc); }
};
// 44: four floats per lane at ANY 4-byte alignment (rows of B / C that do not start on 16 bytes: N % 4 != 0, odd leading
// dimensions, offset views) — dword-aligned global_load / store_dwordx4, which gfx950 serves (split where a request
// crosses a line).  The kernel shifts a row's last, partial quad back to end at column N − 1: it overlaps its neighbour,
// the shared columns are computed twice from the same chain and stored twice with the same bits.
template <>
struct Vec<44> {
  static constexpr int width = 4;
  typedef float type __attribute__((ext_vector_type(4), aligned(4)));
  static __device__ __forceinline__ type zero() { return type{0.f, 0.f, 0.f, 0.f}; }
  static __device__ __forceinline__ type load(const float* p) { return *reinterpret_cast<const type*>(p); }
  static __device__ __forceinline__ void store(float* p, type v) { __builtin_nontemporal_store(v, reinterpret_cast<type*>(p)); }
  static __device__ __forceinline__ type fma(float a, type x, type acc) {
    acc.x = __builtin_fmaf(a, x.x, acc.x);
    acc.y = __builtin_fmaf(a, x.y, acc.y);
    acc.z = __builtin_fmaf(a, x.z, acc.z);
    acc.w = __builtin_fmaf(a, x.w, acc.w);
    return acc;
  }
};
template <>
struct Vec<2> {
  static constexpr int width = 2;
  typedef float type __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ type zero() { return type{0.f, 0.f}; }
  static __device__ __forceinline__ type load(const float* p) { return *reinterpret_cast<const type*>(p); }
  static __device__ __forceinline__ void store(float* p, type v) {
    __builtin_nontemporal_store(v, reinterpret_cast<type*>(p));
  }
  static __device__ __forceinline__ type fma(float a, type x, type acc) {
    acc.x = __builtin_fmaf(a, x.x, acc.x);
    acc.y = __builtin_fmaf(a, x.y, acc.y);
    return acc;
  }
};
template <>
struct Vec<1> {
  static constexpr int width = 1;
  typedef float type;
  static __device__ __forceinline__ type zero() { return 0.f; }
  static __device__ __forceinline__ type load(const float* p) { return *p; }
  static __device__ __forceinline__ void store(float* p, type v) { __builtin_nontemporal_store(v, p); }
  static __device__ __forceinline__ type fma(float a, type x, type acc) { return __builtin_fmaf(a, x, acc); }
};

constexpr int kIntMin = -0x7fffffff - 1;

template <int G>
__device__ __forceinline__ int group_prefix_max(int x, int gl) {
#pragma unroll
  for (int d = 1; d < G; d <<= 1) {
    const int y = __shfl_up(x, d, G);
    if (gl >= d) x = x > y ? x : y;
  }
  return x;
}

}  // namespace

#endif  // MI_SPMM_DEVICE_H_
