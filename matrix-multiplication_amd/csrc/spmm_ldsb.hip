// CSR × dense with the WHOLE dense operand of an item resident in LDS (MI_SPMM_LDS_B) — batched products of many
// small matrices: pruned attention probabilities × V (BASELINE.json configs[4]: 384 items of 512×512 · 512×64, the
// reference reaches it through matmuls.cusparseMM / naiveSpMM and its per-slice recursion, matmuls.py:282-297).
//
// Why: the row-split kernels of spmm_csr.hip gather one 256-byte row of B per non-zero from the L2s.  With every B
// (128 KB) L2-resident that gather runs at ≈30 TB/s chip-wide — the L2s' limit — and the product takes 0.87 ms at
// 100 % kept, 0.124 ms at 10 % (tools/bench_attn_csr.py), the dense MFMA product 0.10 ms whatever the density.  The
// CUs' LDS serve ≈150 TB/s (ds_read_b128: 256 B per clock and CU): a workgroup that first copies its item's B into
// LDS (K·N·4 ≤ 128 KB of the 160 KB) gathers from there.
//
//   * persistent grid, one 16-wave workgroup per CU; a work unit = (item, block of rows); every workgroup takes a
//     CONTIGUOUS range of units, so consecutive units of one item reuse the staged B (384 items × 2 units on 256
//     CUs: 2 stagings per workgroup, not 3);
//   * G = N/4 (rounded up to a power of two) lanes per row, 64/G rows per wave, one float4 of the output row per
//     lane: the 16 lanes of a group read one whole B row per ds_read_b128 (64 distinct banks: conflict-free for any
//     mix of rows in the wave);
//   * a row's col / val are handed round its group as in spmm_group_kernel, but loaded 4·G entries at a time and
//     prefetched across rows (the L2-gather kernels lean on 16+ resident waves per SIMD instead: here LDS is full);
//   * every output element is the row's non-zeros in CSR order, one fmaf chain: the bits of every other plan.
// Rows beyond the long-row threshold are skipped (spmm_dispatch lists them and runs its follow-up kernel), batch = 1 only.
// Round 4, measured and NOT kept (384 × 512² × 64): v_pk_fma_f32 for the chain (0.385 → 0.466 ms at 100 % kept: a packed
// f32 FMA does not issue faster than its two halves), eight staging loads per lane in flight before their LDS writes
// (0.0590 → 0.0592 ms at 10 %: the plain loop was not the cost), one more row step of col / val look-ahead (0.0592 →
// 0.0619 at 10 %, 0.1116 → 0.1152 at 25 %).
#include <atomic>

#include "mi_common.h"
#include "mi_lanes.h"

namespace {

using mi::f32x4;
using mi::group_lane;
using mi::static_for;

__device__ __forceinline__ f32x4 fma4(float a, f32x4 x, f32x4 acc) {
  acc.x = __builtin_fmaf(a, x.x, acc.x);
  acc.y = __builtin_fmaf(a, x.y, acc.y);
  acc.z = __builtin_fmaf(a, x.z, acc.z);
  acc.w = __builtin_fmaf(a, x.w, acc.w);
  return acc;
}

constexpr int kWaves = 16;

#ifndef MI_LDSQ_FMAC_DPP
#define MI_LDSQ_FMAC_DPP 0
#endif
#ifndef MI_LDSQ_BATCH_MASK
#define MI_LDSQ_BATCH_MASK 3  // entries whose LDS reads the scheduler may batch, minus one (A/B in one process, tools/probes/ldsb_ab.py:
                              // 0 / 1 / 3 within 1 % of each other, 3 ahead by 0.5 %)
#endif

#ifdef MI_LDSB_TIMING
// developer build (tools/probes/ldsb_timing.py): wave 0 of every workgroup stamps the 100 MHz wall clock at its phase
// boundaries — entry, after each staging, after each unit, exit — into g_ldsq_stamps[workgroup][slot]
__device__ unsigned long long g_ldsq_stamps[512][16];
__device__ unsigned long long g_ldsq_wave_end[512][16][4];  // [workgroup][wave][unit]: when the wave left the unit's rows
#define LDSQ_WAVE_END(unit) do { if (lane == 0 && (unit) < 4) g_ldsq_wave_end[blockIdx.x & 511][wave][unit] = wall_clock64(); } while (0)
#define LDSQ_STAMP(slot) do { const int s_ = (slot); if (tid == 0 && s_ < 16) g_ldsq_stamps[blockIdx.x & 511][s_] = wall_clock64(); } while (0)
#else
#define LDSQ_STAMP(slot) do {} while (0)
#define LDSQ_WAVE_END(unit) do {} while (0)
#endif

template <int G, bool PERM>
__global__ __launch_bounds__(kWaves * 64) void spmm_ldsb_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ val,
    const float* __restrict__ B, float* __restrict__ C, int M, int K, int N, long ldb, long ldc, long strideB,
    long strideC, const float* __restrict__ bias, int ctiles, int units_per_item, int rows_per_unit,
    unsigned total_units, int long_thresh, const int* __restrict__ perm) {
  // N: width of one column tile (the whole row when ctiles = 1); PERM: entry p's value is val[perm[p]] (a template
  // parameter: as a run-time test inside load_chunk it cost 20 % — 0.385 → 0.462 ms at 100 % kept)
  extern __shared__ __attribute__((aligned(16))) f32x4 Bs[];  // [K][N / 4]
  constexpr int RPW = 64 / G;  // rows per wave
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gl = lane & (G - 1);
  const int nq = N >> 2;          // float4 per row of B / C
  const bool on = gl < nq;        // lanes beyond a row that is not a power of two wide idle
  const unsigned per = (total_units + gridDim.x - 1) / gridDim.x;
  const unsigned u0 = blockIdx.x * per, u1 = u0 + per < total_units ? u0 + per : total_units;
  long staged = -1;
  for (unsigned u = u0; u < u1; ++u) {  // workgroup-uniform
    // unit → (item, column tile, block of rows), blocks of one (item, tile) adjacent: they share the staged slice
    const long slice = u / (unsigned)units_per_item;  // item · ctiles + tile
    const int part = (int)(u % (unsigned)units_per_item);
    const long item = slice / ctiles;
    const int col0 = (int)(slice % ctiles) * N;
    if (slice != staged) {
      if (staged >= 0) __syncthreads();  // every wave is done with the previous slice of B
      const float* Bi = B + item * strideB + col0;
      const int total4 = K * nq;
      for (int i = tid; i < total4; i += kWaves * 64) {
        const int r = i / nq, q = i - r * nq;
        Bs[i] = *reinterpret_cast<const f32x4*>(Bi + (long)r * ldb + 4 * q);
      }
      if (tid < nq) Bs[total4 + tid] = f32x4{0.f, 0.f, 0.f, 0.f};  // row K: what the padding of a row's last chunk reads
      __syncthreads();
      staged = slice;
    }
    const int* rp = rowptr + item * ((long)M + 1);
    float* Ci = C + item * strideC + col0;
    const float* bias_t = bias ? bias + col0 : nullptr;
    const int r0 = part * rows_per_unit;
    const int r1 = r0 + rows_per_unit < M ? r0 + rows_per_unit : M;
    // Rows in a software pipeline: a group's col / val come in super-chunks of SC chunks (SC independent loads per
    // lane), and a row's bounds are loaded two rows ahead, its first super-chunk one row ahead, the next super-chunk
    // of a long row while the current one is processed — left to itself every chunk of G entries cost its own trip
    // to memory (≈3 k cycles per 64 non-zeros of a wave: the first version ran at the L2-gather kernel's speed).
    constexpr int SC = 4;
    constexpr int EPC = mi::LaneChunk<G, true>::ENTRIES;  // entries per chunk
    constexpr int STRIDE = kWaves * RPW;
    auto load_bounds = [&](int rb, int& st, int& en, bool& skip) {
      const int row = rb + lane / G;
      st = en = 0;
      if (row < r1) {
        st = rp[row];
        en = rp[row + 1];
      }
      skip = en - st > long_thresh;  // left to the listed-rows launch (spmm_heavy.hip)
      if (skip) en = st;
    };
    // an entry travels as {byte offset of its B row in LDS, value}; positions beyond the row's end read as value −0 on
    // the all-zero (+0) row K: fmaf(−0, +0, acc) = acc + (−0) leaves EVERY accumulator's bits — +0 stays +0, and a −0
    // accumulator (a row whose products all underflow negatively) stays −0, which padding with +0 would turn into +0
    // where every other plan and the oracle keep −0 — so the four-entry steps carry no predicate
    const int row_bytes = N * 4;
    auto load_chunk = [&](int p, int en, int (&c)[SC], float (&v)[SC]) {
#pragma unroll
      for (int j = 0; j < SC; ++j) {
        const int idx = p + j * EPC + (gl & (EPC - 1));
        c[j] = (idx < en ? col[idx] : K) * row_bytes;
        if (PERM) v[j] = idx < en ? val[perm[idx]] : -0.f;
        else v[j] = idx < en ? val[idx] : -0.f;
      }
    };
    // (lanes beyond a row that is not a power of two wide compute on column 0 and store nothing: no predicate in the loop)
    const char* Bbytes = reinterpret_cast<const char*>(Bs) + 16 * (on ? gl : 0);
    int rb = r0 + wave * RPW;
    int s0, e0, s1 = 0, e1 = 0;
    bool k0, k1 = false;
    int c0[SC], c1[SC];
    float v0[SC], v1[SC];
    load_bounds(rb, s0, e0, k0);
    load_chunk(s0, e0, c0, v0);
    if (rb + STRIDE < r1) load_bounds(rb + STRIDE, s1, e1, k1);
    for (; rb < r1; rb += STRIDE) {  // wave-uniform
      load_chunk(s1, e1, c1, v1);    // nothing to load (s1 = e1) when there is no next row
      int s2 = 0, e2 = 0;
      bool k2 = false;
      if (rb + 2 * STRIDE < r1) load_bounds(rb + 2 * STRIDE, s2, e2, k2);
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      int p = s0;
      while (p < e0) {  // trip count differs between groups
        int cn[SC];
        float vn[SC];
        load_chunk(p + SC * EPC, e0, cn, vn);
#pragma unroll
        for (int j = 0; j < SC; ++j) {
          const int left = e0 - (p + j * EPC);  // group-uniform
          // four entries per step, the last step of a row padded (see load_chunk); the lane index is a compile-time
          // constant so that the broadcast can be a DPP modifier
          static_for<(EPC + 3) / 4>([&](auto b_) {
            constexpr int b = 4 * decltype(b_)::value;
            if (b < left) {
              f32x4 x[4];
              float v[4];
              static_for<4>([&](auto e_) {
                constexpr int e = decltype(e_)::value;
                constexpr int I = b + e < EPC ? b + e : EPC - 1;
                const int off = b + e < EPC ? group_lane<G, I, true>(c0[j]) : K * row_bytes;  // (G < 4: the zero row)
                v[e] = b + e < EPC ? group_lane<G, I, true>(v0[j]) : -0.f;
                x[e] = *reinterpret_cast<const f32x4*>(Bbytes + off);
              });
#pragma unroll
              for (int e = 0; e < 4; ++e) acc = fma4(v[e], x[e], acc);
            }
          });
        }
        p += SC * EPC;
#pragma unroll
        for (int j = 0; j < SC; ++j) {
          c0[j] = cn[j];
          v0[j] = vn[j];
        }
      }
      const int row = rb + lane / G;
      if (row < r1 && !k0 && on) {
        if (bias_t) acc += *reinterpret_cast<const f32x4*>(bias_t + 4 * gl);
        __builtin_nontemporal_store(acc, reinterpret_cast<f32x4*>(Ci + (long)row * ldc + 4 * gl));
      }
      s0 = s1, e0 = e1, k0 = k1;
      s1 = s2, e1 = e2, k1 = k2;
#pragma unroll
      for (int j = 0; j < SC; ++j) {
        c0[j] = c1[j];
        v0[j] = v1[j];
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// The same product with FOUR lanes per row (a DPP quad) and Q float4 of the output row per lane (tile width W = 16·Q:
// 64, 32 or 16 columns) — "quad form", what AUTO runs at BERT's head size.  Why a second form: in spmm_ldsb_kernel a
// 16-lane group pays two DPP moves (column, value) per non-zero and ONE ds_read_b128; a quad pays one DPP move and Q
// DPP adds for Q reads, a wave carries 16 rows instead of 4, and a row's col / val arrive as 16-byte vector loads
// (16 entries per quad and instruction instead of 16 dwords), so the bookkeeping per non-zero — loads, bounds, row
// steps, stores — shrinks fourfold: a workgroup of 16 waves covers a 256-row unit in ONE row step (the 16-lane form:
// four, each waiting for its own bounds → col / val chain).  Same arithmetic: one fmaf chain per output element over
// the row's non-zeros in CSR order.
//   * LDS banking: ds_read_b128 is served 16 lanes at a time in the fixed groups {0–3,12–15,20–27}, {4–11,16–19,
//     28–31} (+32), i.e. four quads whose indices differ mod 4, and they read four DIFFERENT rows.  A quad's lanes read
//     64 contiguous bytes (16 banks); quad q takes the row's 64-byte pieces in the order (j + q) mod Q, so the four
//     quads of a service group are always on four different bank quarters: conflict-free for any mix of rows;
//   * an entry travels as {byte offset of its B row in LDS, value} in component e & 3 of lane e >> 2 of the quad's
//     chunk registers (16 entries) and is handed round by a quad permute; positions past the row's end are (row K, −0)
//     as in spmm_ldsb_kernel, so a chunk's steps carry no lane predicate — only wave-uniform early exits;
//   * the vector loads of col / val start at any entry (dword-aligned 16-byte loads) and are clamped to the arrays'
//     last 16 bytes; the one lane in the whole launch that this shifts rotates its components back;
//   * bounds two row steps ahead, the first chunk of the next row step one ahead, chunk k + 1 of a row while chunk k is
//     processed (16 entry steps of four waves per SIMD: ≈1.5 µs) — all of it ACROSS units and stagings (registers
//     only), so the staging of B hides the first trip to memory for col / val.  ONE look-ahead register set serves
//     both: behind a row's last chunk it holds the next row step's first one (a wave-uniform choice of the address,
//     the same load sequence on either path).  (Separate sets for the row's next chunks and the next row's spilled.)
//   * B is staged by LDS-DMA, C rows leave as buffer stores, every global load is unconditional on a clamped address:
//     see the comments at those places — each was measured (the first version, with predicated loads, ran at half the
//     16-lane form's speed).
// ---------------------------------------------------------------------------------------------
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));

struct QuadChunk {  // 16 entries of a row as its quad holds them: lane e >> 2, component e & 3
  i32x4 c;          // as loaded: column indices; finished: byte offsets of the B rows in LDS
  f32x4 v;
};

template <int Q, bool PERM, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void spmm_ldsq_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ val,
    const float* __restrict__ B, float* __restrict__ C, int M, int K, long ldb, long ldc, long strideB,
    long strideC, const float* __restrict__ bias, int ctile_shift, int units_per_item, int rows_per_unit,
    unsigned total_units, int long_thresh, const int* __restrict__ perm, int last4_hint, long rp_last) {
  // last4: the last index a 16-byte load of col / val may start at = (entries the arrays hold) − 4.  The caller's count
  // (last4_hint = nnz_total − 4 ≥ 0) is an UPPER BOUND by contract (include/mi_spmm.h: rowptr's last entry ≤ nnz ≤ array
  // length); a count BELOW the offsets' last entry (a stale estimate, a shard's local count) must not move the clamp into
  // the data, so the clamp is taken from whichever is larger — the arrays hold at least rowptr[rp_last] entries.
  // 1 << ctile_shift column tiles
  const int true_last4 = rowptr[rp_last] - 4;
  const int last4 = last4_hint > true_last4 ? last4_hint : true_last4;
  extern __shared__ __attribute__((aligned(16))) f32x4 Bs[];  // [K + 1][W / 4]
  constexpr int W = 16 * Q, ROWB = 4 * W, RPW = 16, STRIDE = WAVES * RPW;
  // WAVES: 16 at Q = 4; the narrow tiles (Q = 2, 1) run 6–11 % faster as 8-wave workgroups (two 128-row steps per unit:
  // 96 × 1024² × 64 at 10 % kept 0.0577 → 0.0543 ms, 48 × 2048² × 64 0.154 → 0.137), Q = 4 does not (100 % kept 0.299 → 0.315)
  static_assert((Q & (Q - 1)) == 0, "the rotation needs a power of two");  // (Q < 4: rows narrower than the 64 banks —
  // the quads of a service group meet on a bank quarter when their rows' offsets agree mod 256 B: up to 4-way, by the data)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qd = lane >> 2, gl = lane & 3;
  const unsigned per = (total_units + gridDim.x - 1) / gridDim.x;
  const unsigned u0 = blockIdx.x * per, u1 = u0 + per < total_units ? u0 + per : total_units;
  typedef const f32x4 __attribute__((address_space(3))) * LdsRow;
  const unsigned lds0 = (unsigned)(unsigned long)((LdsRow)Bs);
  unsigned lb[Q];  // lane's address inside a B row for its j-th piece (LDS base included: + entry offset = one DPP add)
#pragma unroll
  for (int j = 0; j < Q; ++j) lb[j] = lds0 + 64 * ((j + qd) & (Q - 1)) + 16 * gl;
  const int zero_off = K * ROWB;

  // The wave's row steps in unit order — (unit, its slice and part, first row of the step) — wave-uniform and advanced
  // without divisions (there is no scalar divide: a uniform quotient costs ≈40 vector instructions).
  struct Cursor {
    unsigned u;
    int slice, part, rb;
  };
  auto unit_r1 = [&](int part) {
    const long e = ((long)part + 1) * rows_per_unit;
    return e < M ? (int)e : M;
  };
  auto settle = [&](Cursor& c) {  // onto the first existing step at or behind (u, rb); u = u1 when there is none
    while (c.u < u1 && c.rb >= unit_r1(c.part)) {
      ++c.u;
      if (++c.part == units_per_item) c.part = 0, ++c.slice;
      c.rb = c.part * rows_per_unit + wave * RPW;
    }
  };
  // Every global load below is UNCONDITIONAL on a clamped address and sits in straight-line code: vmcnt retires in order
  // and the compiler counts a wait from the memory operations on every path — one predicated load (a branch round it)
  // turns the waits into vmcnt(0) and puts the latency of the loads just issued in front of work that needs older ones.
  auto issue_bounds = [&](const Cursor& c, int& st, int& en) {  // (no row here: the same word twice, an empty row)
    // (uniform 64-bit base + 32-bit lane offset: the scalar-base form of the load, no 64-bit vector arithmetic)
    const bool unit = c.u < u1;
    const bool there = unit && c.rb + qd < unit_r1(c.part);
    const char* rp = reinterpret_cast<const char*>(rowptr + (unit ? (long)(c.slice >> ctile_shift) * ((long)M + 1) : 0L));
    const unsigned o = there ? 4u * (unsigned)(c.rb + qd) : 0u;
    st = *reinterpret_cast<const int*>(rp + o);
    en = *reinterpret_cast<const int*>(rp + (there ? o + 4u : 0u));
  };
  auto finish_bounds = [&](int& st, int& en, bool& skip) {  // (skip: left to the listed-rows launch (spmm_heavy.hip))
    skip = en - st > long_thresh;
    if (skip) en = st;
  };
  auto issue_chunk = [&](int p, QuadChunk& q) {
    const int idx = p + 4 * gl;
    const unsigned a = 4u * (unsigned)(idx < last4 ? idx : last4);  // bytes (nnz < 2^29 … checked by the launcher)
    q.c = *reinterpret_cast<const i32x4_u*>(reinterpret_cast<const char*>(col) + a);
    if (PERM) {
      const i32x4 pm = *reinterpret_cast<const i32x4_u*>(reinterpret_cast<const char*>(perm) + a);
      const char* vb = reinterpret_cast<const char*>(val);
      q.v = f32x4{*reinterpret_cast<const float*>(vb + 4u * (unsigned)pm.x), *reinterpret_cast<const float*>(vb + 4u * (unsigned)pm.y),
                  *reinterpret_cast<const float*>(vb + 4u * (unsigned)pm.z), *reinterpret_cast<const float*>(vb + 4u * (unsigned)pm.w)};
    } else {
      q.v = *reinterpret_cast<const f32x4_u*>(reinterpret_cast<const char*>(val) + a);
    }
  };
  auto finish_chunk = [&](int p, int en, QuadChunk& q) {
    const int idx = p + 4 * gl;
    const int left = en - idx;   // entries of this lane inside the row
    const int sh = idx - last4;  // 1 … 3: the load was clamped to the arrays' last 16 bytes and starts sh entries early
    if (__builtin_amdgcn_ballot_w64(left > 0 && sh > 0) != 0) {  // wave-uniform; at most one lane of the whole launch
      if (sh > 0) {
        q.c = sh == 1 ? i32x4{q.c.y, q.c.z, q.c.w, 0} : sh == 2 ? i32x4{q.c.z, q.c.w, 0, 0} : i32x4{q.c.w, 0, 0, 0};
        q.v = sh == 1 ? f32x4{q.v.y, q.v.z, q.v.w, 0.f} : sh == 2 ? f32x4{q.v.z, q.v.w, 0.f, 0.f} : f32x4{q.v.w, 0.f, 0.f, 0.f};
      }
    }
    // positions past the row's end: value −0 on the all-zero row K (see spmm_ldsb_kernel)
    q.c.x = left > 0 ? q.c.x * ROWB : zero_off;
    q.c.y = left > 1 ? q.c.y * ROWB : zero_off;
    q.c.z = left > 2 ? q.c.z * ROWB : zero_off;
    q.c.w = left > 3 ? q.c.w * ROWB : zero_off;
    q.v.x = left > 0 ? q.v.x : -0.f;
    q.v.y = left > 1 ? q.v.y : -0.f;
    q.v.z = left > 2 ? q.v.z : -0.f;
    q.v.w = left > 3 ? q.v.w : -0.f;
  };

  Cursor c0, c1, c2;
  c0.u = u0;
  c0.slice = (int)(u0 / (unsigned)units_per_item);
  c0.part = (int)(u0 % (unsigned)units_per_item);
  c0.rb = c0.part * rows_per_unit + wave * RPW;
  settle(c0);
  c1 = c0;
  c1.rb += STRIDE;
  settle(c1);
  int s0, e0, s1, e1, s2, e2;
  bool k0, k1;
  issue_bounds(c0, s0, e0);
  issue_bounds(c1, s1, e1);
  finish_bounds(s0, e0, k0);
  QuadChunk r0, rn;  // the chunk being processed; the next one of the wave's chunk stream (this row's, or the next row's first)
  issue_chunk(s0, r0);
  finish_bounds(s1, e1, k1);

  long staged = -1;
  [[maybe_unused]] int stamp = 0;
  LDSQ_STAMP(stamp++);
  for (unsigned u = u0; u < u1; ++u) {  // workgroup-uniform
    const long slice = u / (unsigned)units_per_item;
    const int part = (int)(u % (unsigned)units_per_item);
    const long item = slice >> ctile_shift;
    const int col0 = (int)(slice & ((1 << ctile_shift) - 1)) * W;
    if (slice != staged) {
      if (staged >= 0) __syncthreads();  // every wave is done with the previous slice of B
      // LDS-DMA: a wave-instruction copies 1 KiB (four rows of the slice) straight into the image — no registers, so all
      // of a wave's eight pieces are in flight at once (through registers the compiler waited for every 16-byte load
      // before its ds_write: eight trips to memory one behind the other, 6–15 µs per staging under load)
      const float* Bi = B + item * strideB + col0;
      constexpr int nq = W / 4;
      const int total4 = K * nq;
      for (int f0 = wave * 64; f0 < total4; f0 += WAVES * 64) {  // wave-uniform
        const int f = f0 + lane;
        if (f < total4)
          __builtin_amdgcn_global_load_lds(reinterpret_cast<const __attribute__((address_space(1))) void*>(
                                               reinterpret_cast<unsigned long>(Bi + (long)(f / nq) * ldb + 4 * (f % nq))),
                                           (__attribute__((address_space(3))) void*)(Bs + f0), 16, 0, 0);
      }
      if (tid < nq) Bs[total4 + tid] = f32x4{0.f, 0.f, 0.f, 0.f};  // row K: what the padding reads
      __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): the pieces have landed (expcnt / lgkmcnt left alone)
      __syncthreads();
      staged = slice;
      LDSQ_STAMP(stamp++);
    }
    const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(C + item * strideC + col0, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t brsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias ? bias + col0 : B), 0, 0x7fffffff, 0x00020000);
    const int r1 = unit_r1(part);
    while (c0.u == u) {  // wave-uniform: this wave's row steps inside the unit
      // the bounds of the step two ahead go out first; the next step's first chunk follows this row's last one
      c2 = c1;
      c2.rb += STRIDE;
      settle(c2);
      issue_bounds(c2, s2, e2);

      f32x4 acc[Q];
#pragma unroll
      for (int j = 0; j < Q; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      int p = s0;
      for (;;) {  // wave-uniform trip count: the longest row of the wave's sixteen, in chunks of 16 entries
        const bool more = __builtin_amdgcn_ballot_w64(p + 16 < e0) != 0;
        issue_chunk(more ? p + 16 : s1, rn);  // (uniform choice: one load sequence on every path)
        finish_chunk(p, e0, r0);
        const int left = e0 - p;
        static_for<4>([&](auto b_) {
          constexpr int b = 4 * decltype(b_)::value;
          if (b == 0 || __builtin_amdgcn_ballot_w64(b < left) != 0) {  // wave-uniform; a row that has ended reads padding
            static_for<4>([&](auto e_) {
              constexpr int e = b + decltype(e_)::value;
              constexpr int src = e >> 2, comp = e & 3;
              const unsigned off = (unsigned)group_lane<4, src, true>(r0.c[comp]);
              f32x4 x[Q];
#pragma unroll
              for (int j = 0; j < Q; ++j) x[j] = *(LdsRow)(unsigned long)(off + lb[j]);
#if MI_LDSQ_FMAC_DPP
              // developer A/B: the value's quad broadcast as a DPP modifier of sixteen v_fmac_f32 instead of one
              // v_mov_b32_dpp + eight v_pk_fma_f32 — measured 15 % SLOWER (0.340 vs 0.297 ms at 100 % kept): the packed
              // FMA issues at full rate (SQ_ACTIVE_INST_VALU counts one quad-cycle for it)
#pragma unroll
              for (int j = 0; j < Q; ++j) {
                float a0 = acc[j].x, a1 = acc[j].y, a2 = acc[j].z, a3 = acc[j].w;
                asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[%3,%3,%3,%3] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a0) : "v"(r0.v[comp]), "v"(x[j].x), "n"(src));
                asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[%3,%3,%3,%3] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a1) : "v"(r0.v[comp]), "v"(x[j].y), "n"(src));
                asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[%3,%3,%3,%3] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a2) : "v"(r0.v[comp]), "v"(x[j].z), "n"(src));
                asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[%3,%3,%3,%3] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a3) : "v"(r0.v[comp]), "v"(x[j].w), "n"(src));
                acc[j] = f32x4{a0, a1, a2, a3};
              }
#else
              const float v = group_lane<4, src, true>(r0.v[comp]);
#pragma unroll
              for (int j = 0; j < Q; ++j) acc[j] = fma4(v, x[j], acc[j]);
#endif
              // (a fence for the scheduler every MI_LDSQ_BATCH_MASK + 1 entries: left alone it hoists sixteen entries' reads and spills)
              if constexpr ((e & MI_LDSQ_BATCH_MASK) == MI_LDSQ_BATCH_MASK) __builtin_amdgcn_sched_barrier(0);
            });
          }
        });
        r0 = rn;
        p += 16;
        if (!more) break;
      }
      const int row = c0.rb + qd;
      if (row < r1 && !k0) {
        // buffer stores — the item's C as a scalar descriptor + a 32-bit lane offset (a piece's byte offset inside the
        // row is lb[j] without the LDS base): as global stores the compiler kept four 64-bit lane addresses per unit
        // alive through the row loop and spilled them
        const unsigned crow = (unsigned)row * (unsigned)ldc * 4u - lds0;
#pragma unroll
        for (int j = 0; j < Q; ++j) {
          f32x4 o = acc[j];
          if (bias)
            o += __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(brsrc, (int)(lb[j] - lds0), 0, 0));
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), crsrc, (int)(crow + lb[j]), 0, 2 /* nt */);
        }
      }
      c0 = c1, c1 = c2;
      s0 = s1, e0 = e1, k0 = k1;
      s1 = s2, e1 = e2;
      finish_bounds(s1, e1, k1);
    }
    LDSQ_WAVE_END((int)(u - u0));
    LDSQ_STAMP(stamp++);
  }
}


// ---------------------------------------------------------------------------------------------
// SDDMM on a batched CSR pattern with the item's B resident in LDS (round 4): out[p] = <dC[item][row(p)], B[item][col[p]]>
// — the gradient of the stored values of a batched CSR tensor (pruned attention probabilities: d probs = dctx · Vᵀ on
// the pattern).  matmuls ran it as ONE SDDMM on the block-diagonal matrix of the batch (sddmm_group_kernel, convert.hip):
// every non-zero gathers a 256-byte row of V from the L2s — 0.2 ms at 384 × 512² × 64 and 10–25 % kept, the largest
// item of that backward.  Here a workgroup stages its item's B once (as spmm_ldsb_kernel does, same units, same
// persistent grid) and the gathers come from LDS.  Arithmetic = sddmm_group_kernel's, bit for bit: G = N/4 lanes per
// row, lane l chains columns 4l … 4l+3 of the product, the tree levels whose partner never had columns add +0, and the G
// sums of a chunk of G entries go through one joint xor tree (distances G/2 … 1) that leaves entry l's sum in lane l.
// A chunk's columns are loaded one chunk ahead, a row's bounds and its dC row one row step ahead.
// ---------------------------------------------------------------------------------------------
template <int G>  // ≤ 16: a group's columns sit in one 16-lane DPP row
__global__ __launch_bounds__(kWaves * 64) void sddmm_ldsb_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ dC,
    const float* __restrict__ B, float* __restrict__ out, int M, int K, int N, long lddc, long strideDC, long ldb,
    long strideB, int units_per_item, int rows_per_unit, unsigned total_units) {
  extern __shared__ __attribute__((aligned(16))) f32x4 Bs[];  // [K][N / 4]
  constexpr int RPW = 64 / G;
  constexpr int U = G < 8 ? G : 8;  // gathers in flight per lane
  constexpr int STRIDE = kWaves * RPW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gl = lane & (G - 1);
  const int nq = N >> 2;
  const bool on = gl < nq;
  const unsigned per = (total_units + gridDim.x - 1) / gridDim.x;
  const unsigned u0 = blockIdx.x * per, u1 = u0 + per < total_units ? u0 + per : total_units;
  long staged = -1;
  const int row_bytes = N * 4;
  const char* Bbytes = reinterpret_cast<const char*>(Bs) + 16 * (on ? gl : 0);
  for (unsigned u = u0; u < u1; ++u) {  // workgroup-uniform
    const long item = u / (unsigned)units_per_item;
    const int part = (int)(u % (unsigned)units_per_item);
    if (item != staged) {
      if (staged >= 0) __syncthreads();  // every wave is done with the previous item's B
      const float* Bi = B + item * strideB;
      const int total4 = K * nq;
      for (int i = tid; i < total4; i += kWaves * 64) {
        const int r = i / nq, q = i - r * nq;
        Bs[i] = *reinterpret_cast<const f32x4*>(Bi + (long)r * ldb + 4 * q);
      }
      __syncthreads();
      staged = item;
    }
    const int* rp = rowptr + item * ((long)M + 1);
    const float* dCi = dC + item * strideDC;
    const int r0 = part * rows_per_unit;
    const int r1 = r0 + rows_per_unit < M ? r0 + rows_per_unit : M;
    auto load_row = [&](int rb, int& st, int& en, f32x4& x) {
      const int row = rb + lane / G;
      st = en = 0;
      x = f32x4{0.f, 0.f, 0.f, 0.f};
      if (row < r1) {
        st = rp[row];
        en = rp[row + 1];
        if (on && st < en) x = *reinterpret_cast<const f32x4*>(dCi + (long)row * lddc + 4 * gl);
      }
    };
    int rb = r0 + wave * RPW;
    int s0, e0, s1 = 0, e1 = 0;
    f32x4 x0, x1 = f32x4{0.f, 0.f, 0.f, 0.f};
    load_row(rb, s0, e0, x0);
    int c0 = s0 + gl < e0 ? col[s0 + gl] : 0;  // past the row's end: row 0 of B (its sum is never stored)
    for (; rb < r1; rb += STRIDE) {  // wave-uniform
      if (rb + STRIDE < r1) load_row(rb + STRIDE, s1, e1, x1);
      else s1 = e1 = 0;
      const bool empty_row = s0 >= e0;
      for (int p0 = s0; p0 < e0; p0 += G) {  // trip count differs between groups
        const int cnt = e0 - p0 < G ? e0 - p0 : G;  // group-uniform
        const int mycol = c0;
        // the next chunk's columns — of this row, or behind its last chunk the first of the next row — are in flight meanwhile
        const int pn = p0 + G;
        if (pn < e0) c0 = pn + gl < e0 ? col[pn + gl] : 0;
        else c0 = s1 + gl < e1 ? col[s1 + gl] : 0;
        float s[G];
        static_for<G / U>([&](auto b_) {
          constexpr int i = U * decltype(b_)::value;
          if (i < cnt) {
            f32x4 y[U];
            static_for<U>([&](auto u_) {
              constexpr int uu = decltype(u_)::value;
              const int off = group_lane<G, i + uu, true>(mycol) * row_bytes;
              y[uu] = *reinterpret_cast<const f32x4*>(Bbytes + off);
            });
#pragma unroll
            for (int uu = 0; uu < U; ++uu) {
              float acc = 0.f;
              if (on) {
                acc = __builtin_fmaf(x0.x, y[uu].x, acc);
                acc = __builtin_fmaf(x0.y, y[uu].y, acc);
                acc = __builtin_fmaf(x0.z, y[uu].z, acc);
                acc = __builtin_fmaf(x0.w, y[uu].w, acc);
              }
#pragma unroll
              for (int w = 32; w >= G; w >>= 1) acc = acc + 0.0f;  // the tree levels whose partner never had columns
              s[i + uu] = acc;
            }
          } else {
#pragma unroll
            for (int uu = 0; uu < U; ++uu) s[i + uu] = 0.f;
          }
        });
        // joint xor tree over the group: after the last level lane l holds the sum of entry l
#pragma unroll
        for (int w = G / 2; w >= 1; w >>= 1) {
          const bool hi = (gl & w) != 0;
#pragma unroll
          for (int k = 0; k < w; ++k) {
            const float keep = hi ? s[k + w] : s[k];
            const float send = hi ? s[k] : s[k + w];
            s[k] = keep + __shfl_xor(send, w, 64);
          }
        }
        if (gl < cnt) out[p0 + gl] = s[0];
      }
      if (empty_row) c0 = s1 + gl < e1 ? col[s1 + gl] : 0;  // (a row without entries prefetched nothing)
      s0 = s1, e0 = e1, x0 = x1;
    }
  }
}


// ---------------------------------------------------------------------------------------------
// The batched SDDMM in quad form (N = 64): as spmm_ldsq_kernel — four lanes per row, a wave carries 16 rows, 16-byte
// loads of the columns, LDS-DMA staging, one 256-row step per unit — with sddmm_ldsb_kernel's arithmetic bit for bit.
// That kernel's order: 16 lanes l chain columns 4l … 4l+3 (p[l]), every chain takes "+0" twice (the tree levels 32 and
// 16, whose partners never had columns), then a xor tree over the distances 8, 4, 2, 1.  Here lane g of the quad holds
// the four pieces 16c + 4g … of the dC row, i.e. the chains l = g + 4c: the levels 8 and 4 pair chains of ONE lane
// ((p[c=0] + p[c=2]) + (p[c=1] + p[c=3]), whatever the bank rotation: addition commutes), the levels 2 and 1 are two
// quad-permute adds.  The "+0"s: replacing −0 leaves by +0 changes a tree's result only when EVERY leaf is −0 (the only
// sum that gives −0) — so one "+0" on the result is the same bits as one on every leaf.
// Results of a chunk are held until the next chunk's loads are out: a store between a load and its wait would make the
// wait cover the store.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWaves * 64) void sddmm_ldsq_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ dC,
    const float* __restrict__ B, float* __restrict__ out, int M, int K, long lddc, long strideDC, long ldb,
    long strideB, int units_per_item, int rows_per_unit, unsigned total_units, int last4_hint, long rp_last, int ktile_shift,
    int ktile_rows) {
  const int true_last4 = rowptr[rp_last] - 4;  // see spmm_ldsq_kernel: the caller's count is an upper bound, never a clamp below the data
  const int last4 = last4_hint > true_last4 ? last4_hint : true_last4;
  // 1 << ktile_shift tiles of ktile_rows rows of B per item (K beyond the 512 rows the image holds: an entry needs ONE row of
  // B, so a pass per tile computes the entries whose column lies in it — exact whatever the order of columns inside a row)
  extern __shared__ __attribute__((aligned(16))) f32x4 Bs[];  // [min(K, ktile_rows)][16]
  constexpr int Q = 4, W = 64, ROWB = 4 * W, RPW = 16, STRIDE = kWaves * RPW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qd = lane >> 2, gl = lane & 3;
  const unsigned per = (total_units + gridDim.x - 1) / gridDim.x;
  const unsigned u0 = blockIdx.x * per, u1 = u0 + per < total_units ? u0 + per : total_units;
  typedef const f32x4 __attribute__((address_space(3))) * LdsRow;
  const unsigned lds0 = (unsigned)(unsigned long)((LdsRow)Bs);
  unsigned lb[Q];
#pragma unroll
  for (int j = 0; j < Q; ++j) lb[j] = lds0 + 64 * ((j + qd) & (Q - 1)) + 16 * gl;

  struct Cursor {
    unsigned u;
    int slice, part, rb;
  };
  auto unit_r1 = [&](int part) {
    const long e = ((long)part + 1) * rows_per_unit;
    return e < M ? (int)e : M;
  };
  auto settle = [&](Cursor& c) {
    while (c.u < u1 && c.rb >= unit_r1(c.part)) {
      ++c.u;
      if (++c.part == units_per_item) c.part = 0, ++c.slice;
      c.rb = c.part * rows_per_unit + wave * RPW;
    }
  };
  // a step's bounds, and its dC row (the lane's four pieces); no row here: the same word twice / row 0 of item 0
  auto issue_bounds = [&](const Cursor& c, int& st, int& en) {
    const bool unit = c.u < u1;
    const bool there = unit && c.rb + qd < unit_r1(c.part);
    const char* rp = reinterpret_cast<const char*>(rowptr + (unit ? (long)(c.slice >> ktile_shift) * ((long)M + 1) : 0L));
    const unsigned o = there ? 4u * (unsigned)(c.rb + qd) : 0u;
    st = *reinterpret_cast<const int*>(rp + o);
    en = *reinterpret_cast<const int*>(rp + (there ? o + 4u : 0u));
  };
  auto issue_x = [&](const Cursor& c, f32x4 (&x)[Q]) {
    const bool unit = c.u < u1;
    const bool there = unit && c.rb + qd < unit_r1(c.part);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(dC + (unit ? (long)(c.slice >> ktile_shift) * strideDC : 0L)), 0, 0x7fffffff, 0x00020000);
    const unsigned xo = (there ? (unsigned)(c.rb + qd) * (unsigned)lddc * 4u : 0u) - lds0;
#pragma unroll
    for (int j = 0; j < Q; ++j) x[j] = __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(xr, (int)(xo + lb[j]), 0, 0));
  };
  auto issue_chunk = [&](int p, i32x4& c) {
    const int idx = p + 4 * gl;
    c = *reinterpret_cast<const i32x4_u*>(reinterpret_cast<const char*>(col) + 4u * (unsigned)(idx < last4 ? idx : last4));
  };
  // → byte offsets of the B rows in the staged tile + which of the lane's four entries are computed in this pass (inside
  // the row and, with tiles, a column of this tile); the others read row 0 of the image and their sum is not stored
  auto finish_chunk = [&](int p, int en, int tile_lo, int tile_n, i32x4& c) -> unsigned {
    const int idx = p + 4 * gl;
    const int left = en - idx;
    const int sh = idx - last4;  // see spmm_ldsq_kernel
    if (__builtin_amdgcn_ballot_w64(left > 0 && sh > 0) != 0) {
      if (sh > 0) c = sh == 1 ? i32x4{c.y, c.z, c.w, 0} : sh == 2 ? i32x4{c.z, c.w, 0, 0} : i32x4{c.w, 0, 0, 0};
    }
    unsigned vm = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int rel = c[t] - tile_lo;
      const bool ok = left > t && (unsigned)rel < (unsigned)tile_n;
      c[t] = ok ? rel * ROWB : 0;
      vm |= ok ? 1u << t : 0u;
    }
    return vm;
  };
  auto store_chunk = [&](int p, unsigned vm, const f32x4& o) {
    char* dst = reinterpret_cast<char*>(out) + 4u * (unsigned)(p + 4 * gl);
    if (vm == 15u) {
      *reinterpret_cast<f32x4_u*>(dst) = o;
    } else if (vm != 0u) {
      if (vm & 1u) *reinterpret_cast<float*>(dst) = o.x;
      if (vm & 2u) *reinterpret_cast<float*>(dst + 4) = o.y;
      if (vm & 4u) *reinterpret_cast<float*>(dst + 8) = o.z;
      if (vm & 8u) *reinterpret_cast<float*>(dst + 12) = o.w;
    }
  };

  Cursor c0, c1, c2;
  c0.u = u0;
  c0.slice = (int)(u0 / (unsigned)units_per_item);
  c0.part = (int)(u0 % (unsigned)units_per_item);
  c0.rb = c0.part * rows_per_unit + wave * RPW;
  settle(c0);
  c1 = c0;
  c1.rb += STRIDE;
  settle(c1);
  int s0, e0, s1, e1, s2, e2;
  f32x4 x0[Q], x1[Q];
  issue_bounds(c0, s0, e0);
  issue_bounds(c1, s1, e1);
  issue_x(c0, x0);
  i32x4 r0, rn;
  issue_chunk(s0, r0);

  long staged = -1;
  for (unsigned u = u0; u < u1; ++u) {  // workgroup-uniform
    const long slice = u / (unsigned)units_per_item;
    const long item = slice >> ktile_shift;
    const int tile_lo = (int)(slice & ((1 << ktile_shift) - 1)) * ktile_rows;
    const int tile_n = K - tile_lo < ktile_rows ? (K - tile_lo > 0 ? K - tile_lo : 0) : ktile_rows;
    if (slice != staged) {
      if (staged >= 0) __syncthreads();
      const float* Bi = B + item * strideB + (long)tile_lo * ldb;
      constexpr int nq = W / 4;
      const int total4 = tile_n * nq;
      for (int f0 = wave * 64; f0 < total4; f0 += kWaves * 64) {  // wave-uniform; LDS-DMA, see spmm_ldsq_kernel
        const int f = f0 + lane;
        if (f < total4)
          __builtin_amdgcn_global_load_lds(reinterpret_cast<const __attribute__((address_space(1))) void*>(
                                               reinterpret_cast<unsigned long>(Bi + (long)(f / nq) * ldb + 4 * (f % nq))),
                                           (__attribute__((address_space(3))) void*)(Bs + f0), 16, 0, 0);
      }
      __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
      __syncthreads();
      staged = slice;
    }
    while (c0.u == u) {  // wave-uniform: this wave's row steps inside the unit
      // the bounds of the step two ahead and the dC row of the next step go out first
      c2 = c1;
      c2.rb += STRIDE;
      settle(c2);
      issue_bounds(c2, s2, e2);
      issue_x(c1, x1);
      int p = s0;
      int pend_p = 0;
      unsigned pend_vm = 0;  // the chunk whose results wait in `o` (no bit set: none)
      f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
      for (;;) {  // wave-uniform trip count
        const bool more = __builtin_amdgcn_ballot_w64(p + 16 < e0) != 0;
        store_chunk(pend_p, pend_vm, o);
        issue_chunk(more ? p + 16 : s1, rn);
        const unsigned vm = finish_chunk(p, e0, tile_lo, tile_n, r0);
        static_for<4>([&](auto b_) {
          constexpr int b = 4 * decltype(b_)::value;
          // wave-uniform: some quad's lane b / 4 holds an entry of this pass (a row's entries of one tile are a run when
          // its columns ascend, so a pass skips most steps of the other tiles' entries)
          if (__builtin_amdgcn_ballot_w64(gl == b / 4 && vm != 0u) != 0) {
            static_for<4>([&](auto e_) {
              constexpr int e = b + decltype(e_)::value;
              constexpr int src = e >> 2, comp = e & 3;
              const unsigned off = (unsigned)group_lane<4, src, true>(r0[comp]);
              f32x4 y[Q];
#pragma unroll
              for (int j = 0; j < Q; ++j) y[j] = *(LdsRow)(unsigned long)(off + lb[j]);
              float part[Q];
#pragma unroll
              for (int j = 0; j < Q; ++j) {
                float a = __builtin_fmaf(x0[j].x, y[j].x, 0.f);
                a = __builtin_fmaf(x0[j].y, y[j].y, a);
                a = __builtin_fmaf(x0[j].z, y[j].z, a);
                part[j] = __builtin_fmaf(x0[j].w, y[j].w, a);
              }
              float t = (part[0] + part[2]) + (part[1] + part[3]);                                   // levels 8, 4
              t = t + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x4e, 0xf, 0xf, true));  // 2
              t = t + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xb1, 0xf, 0xf, true));  // 1
              if (gl == src) o[comp] = t + 0.0f;
              if constexpr ((e & 1) == 1) __builtin_amdgcn_sched_barrier(0);
            });
          }
        });
        pend_p = p;
        pend_vm = vm;
        r0 = rn;
        p += 16;
        if (!more) break;
      }
      store_chunk(pend_p, pend_vm, o);
      c0 = c1, c1 = c2;
      s0 = s1, e0 = e1;
      s1 = s2, e1 = e2;
#pragma unroll
      for (int j = 0; j < Q; ++j) x0[j] = x1[j];
    }
  }
}

}  // namespace

namespace mi {

// Column tiles: when K·N·4 exceeds the 128 KB an LDS image may take, the columns are cut into 2 or 4 tiles of W = N / tiles
// (W ≥ 32: 8-lane groups, the narrowest with DPP broadcasts — 2048 × 64 as four tiles of 16 ran 25 % behind the group
// kernel) and a unit covers one tile — every output element still sees its row's non-zeros in CSR
// order (no assumption on the order of columns inside a row, which cutting K would need); the row's col / val are read
// once per tile.  1024 tokens × 64: two tiles of 32 columns (0.114 → 0.090 ms at 10 % kept); 1024 × 128: four of 32 (0.207 → 0.162).
static int ldsb_column_tiles(int32_t K, int32_t N) {
  if (N < 4 || N % 4 != 0 || N > 256 || K < 1) return 0;
  for (int tiles = 1; tiles <= 4; tiles *= 2) {
    const int w = N / tiles;
    if (N % tiles != 0 || w % 4 != 0 || (tiles > 1 && w < 32)) return 0;
    if ((long)K * w * 4 <= 128L * 1024) return tiles;
  }
  return 0;
}

// Quad form (spmm_ldsq_kernel): column tiles of 64, 32 or 16 — the widest whose image (K + 1 rows) fits — whatever the
// 16-lane form's tiling: 512 rows of B at 64 columns, 1024 at 32, 2048 at 16 (attention over 2048 tokens: four tiles of
// 16, a shape the 16-lane form leaves to the L2 gathers); at most 8 tiles.  0: the form does not cover the shape.
static int ldsq_tile_width(int32_t K, int32_t N) {
  if (N < 16 || N % 16 != 0 || K < 1) return 0;
  for (int w = 64; w >= 16; w /= 2)
    if (N % w == 0 && N / w <= 8 && ((N / w) & (N / w - 1)) == 0 && ((long)K + 1) * w * 4 <= 132L * 1024) return w;
  return 0;
}

// whether the plan can take the problem at all (shape only; the caller checked the vec4 requirements)
bool spmm_ldsb_fits(int32_t K, int32_t N) { return ldsb_column_tiles(K, N) > 0 || ldsq_tile_width(K, N) > 0; }
int spmm_ldsb_tiles(int32_t K, int32_t N) {
  const int w = ldsq_tile_width(K, N);
  return w > 0 ? N / w : ldsb_column_tiles(K, N);
}

// 0: the 16-lane form only, 1: the quad form wherever it covers the shape, -1 (default): by rule (developer A/B and tests)
static std::atomic<int> g_ldsb_form{-1};

int launch_spmm_ldsb(const int32_t* rowptr, const int32_t* col, const float* val, const float* B, float* C,
                     int32_t batch, int32_t M, int32_t K, int32_t N, int64_t ldb, int64_t ldc, int64_t strideB,
                     int64_t strideC, const float* bias, int long_thresh, hipStream_t s, const int32_t* perm,
                     int64_t nnz_total) {
  static const int cus = [] {  // thread-safe one-time query (every device of a node has the same CU count)
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
      (void)hipGetLastError();
      n = 256;
    }
    return n;
  }();
  // Quad form (spmm_ldsq_kernel)
  const int form = g_ldsb_form.load(std::memory_order_relaxed);
  const int qw = ldsq_tile_width(K, N);
  if (form != 0 && qw != 0 && nnz_total >= 4 && nnz_total < (1LL << 29) && (int64_t)M * ldc < (1LL << 29)) {
    const int qtiles = N / qw;
    // Units of whole row steps (256 rows with 16 waves, 128 with 8), as many as keep every CU at work.  The narrow tiles
    // run 8 waves (see the kernel); the 64-column tile takes 8 where the finer units suit the persistent grid better —
    // priced as the busiest workgroup's rows (≈3 ns + 0.66 ns per non-zero each; the 8-wave build ≈5 % more) plus its
    // stagings (≈35 ns per KB of image), from tools/bench_ldsb_forms.py: 384 items of 256 rows run 3 units of 128 rows per
    // CU instead of 2 of 256 on three quarters of the CUs (0.053 → 0.047 ms at 50 % kept), 96 items of 1024 × 512 keep
    // their 2 units of 256 rows — one staging per workgroup instead of two (0.013 against 0.018 ms at 2 % kept).
    struct Units {
      int upi, rpu;
      long total, per;
      double span;  // ns, the busiest workgroup
    };
    const double row_ns = 3.0 + 0.66 * (double)nnz_total / ((double)batch * (double)M);
    const double stage_ns = 35.0 * ((double)K + 1) * qw * 4 / 1024.0;
    auto plan_units = [&](int step, double price) {
      Units u;
      u.upi = 1;
      while ((long)batch * qtiles * u.upi < 3L * cus && ((long)M + u.upi * 2 - 1) / (u.upi * 2) >= step) u.upi *= 2;
      u.rpu = (int)((((long)M + u.upi - 1) / u.upi + step - 1) / step * step);
      u.upi = (int)(((long)M + u.rpu - 1) / u.rpu);
      u.total = (long)batch * qtiles * u.upi;
      u.per = (u.total + cus - 1) / cus;
      long stagings = 1;  // the most slices a workgroup's contiguous units touch
      const long groups = (u.total + u.per - 1) / u.per;
      for (long w = 0; w < groups && w < 4096; ++w) {
        const long a = w * u.per, b = a + u.per < u.total ? a + u.per : u.total;
        const long n = (b - 1) / u.upi - a / u.upi + 1;
        stagings = n > stagings ? n : stagings;
      }
      u.span = (double)u.per * u.rpu * row_ns * price + (double)stagings * stage_ns;
      return u;
    };
    const Units u16 = plan_units(256, 1.0), u8 = plan_units(128, 1.05);
    const bool eight = qw != 64 || u8.span < u16.span;
    const Units& un = eight ? u8 : u16;
    const int upi = un.upi, rpu = un.rpu;
    const long total = un.total, per = un.per;
    if (total > 0x7fffffffL) return MI_ERANGE;
    const unsigned grid = (unsigned)((total + per - 1) / per);
    const size_t lds = ((size_t)K + 1) * qw * 4;
    const int shift = qtiles == 8 ? 3 : qtiles == 4 ? 2 : qtiles == 2 ? 1 : 0;
#define MI_LDSQ(Q_, WAVES_)                                                                                           \
  do {                                                                                                                \
    auto k = perm ? spmm_ldsq_kernel<Q_, true, WAVES_> : spmm_ldsq_kernel<Q_, false, WAVES_>;                         \
    if (lds > 64 * 1024) MI_HIP_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL(k, dim3(grid), dim3(WAVES_ * 64), lds, s, rowptr, col, val, B, C, M, K, (long)ldb, (long)ldc,  \
                       (long)strideB, (long)strideC, bias, shift, upi, rpu, (unsigned)total, long_thresh, perm,         \
                       (int)(nnz_total - 4), (long)batch * ((long)M + 1) - 1);                                         \
  } while (0)
    if (qw == 64 && !eight) MI_LDSQ(4, 16);
    else if (qw == 64) MI_LDSQ(4, 8);
    else if (qw == 32) MI_LDSQ(2, 8);
    else MI_LDSQ(1, 8);
#undef MI_LDSQ
    return check_launch();
  }
  const int ctiles = ldsb_column_tiles(K, N);
  if (ctiles == 0) return 1;  // only the quad form covers the shape and it was pinned off / out of its index range: not taken
  const int W = N / ctiles;
  // units: blocks of rows of one item, sized so that all CUs get work (two blocks per item for 384 items on 256 CUs),
  // never below 64 rows
  int units_per_item = 1;
  while ((long)batch * ctiles * units_per_item < 3L * cus && ((long)M + units_per_item * 2 - 1) / (units_per_item * 2) >= 64)
    units_per_item *= 2;
  const int rows_per_unit = (int)(((long)M + units_per_item - 1) / units_per_item);
  const long total = (long)batch * ctiles * units_per_item;
  if (total > 0x7fffffffL) return MI_ERANGE;
  const unsigned grid = (unsigned)(total < cus ? total : cus);
  const size_t lds = ((size_t)K + 1) * W * 4;  // + the all-zero row
  const int G = pow2_ceil(W / 4);
#define MI_LDSB(G_)                                                                                                   \
  do {                                                                                                                \
    auto k = perm ? spmm_ldsb_kernel<G_, true> : spmm_ldsb_kernel<G_, false>;                                         \
    if (lds > 64 * 1024) MI_HIP_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL(k, dim3(grid), dim3(kWaves * 64), lds, s, rowptr, col, val, B, C, M, K, W, (long)ldb, (long)ldc, \
                       (long)strideB, (long)strideC, bias, ctiles, units_per_item, rows_per_unit, (unsigned)total,        \
                       long_thresh, perm);                                                                             \
  } while (0)
  switch (G) {
    case 1: MI_LDSB(1); break;
    case 2: MI_LDSB(2); break;
    case 4: MI_LDSB(4); break;
    case 8: MI_LDSB(8); break;
    case 16: MI_LDSB(16); break;
    case 32: MI_LDSB(32); break;
    default: MI_LDSB(64); break;
  }
#undef MI_LDSB
  return check_launch();
}


// whether the LDS-resident SDDMM takes a batched problem (shape only; the caller checked alignment): what
// sddmm_group_kernel covers (N ≤ 64, N % 4 == 0) with an item's B fitting the LDS image
bool sddmm_ldsb_fits(int32_t K, int32_t N) {
  return N >= 4 && N <= 64 && N % 4 == 0 && K >= 1 && ((long)K * N * 4 <= 128L * 1024 || (N == 64 && K <= 4096));
}

int launch_sddmm_ldsb(const int32_t* rowptr, const int32_t* col, const float* dC, const float* B, float* out,
                      int32_t batch, int32_t M, int32_t K, int32_t N, int64_t lddc, int64_t strideDC, int64_t ldb,
                      int64_t strideB, hipStream_t s, int64_t nnz_total) {
  if (!sddmm_ldsb_fits(K, N)) return MI_EINVAL;
  const int form = g_ldsb_form.load(std::memory_order_relaxed);
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
      (void)hipGetLastError();
      n = 256;
    }
    return n;
  }();
  if (form != 0 && N == 64 && K <= 4096 && nnz_total >= 4 && nnz_total < (1LL << 29) && (int64_t)M * lddc < (1LL << 29)) {
    // the quad form (sddmm_ldsq_kernel): units of whole 256-row steps, as launch_spmm_ldsb sizes them; B beyond the 512 rows
    // the image holds goes in as 2 / 4 / 8 tiles of 512 rows, a pass each
    const int ktiles = K <= 512 ? 1 : K <= 1024 ? 2 : K <= 2048 ? 4 : 8;
    int upi = 1;
    while ((long)batch * ktiles * upi < 3L * cus && ((long)M + upi * 2 - 1) / (upi * 2) >= 256) upi *= 2;
    const int rpu = (int)((((long)M + upi - 1) / upi + 255) / 256 * 256);
    upi = (int)(((long)M + rpu - 1) / rpu);
    const long total = (long)batch * ktiles * upi;
    if (total > 0x7fffffffL) return MI_ERANGE;
    const long per = (total + cus - 1) / cus;
    const unsigned grid = (unsigned)((total + per - 1) / per);
    const size_t lds = (size_t)(K < 512 ? K : 512) * 64 * 4;
    if (lds > 64 * 1024)
      MI_HIP_TRY(hipFuncSetAttribute((const void*)sddmm_ldsq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(sddmm_ldsq_kernel, dim3(grid), dim3(kWaves * 64), lds, s, rowptr, col, dC, B, out, M, K, (long)lddc,
                       (long)strideDC, (long)ldb, (long)strideB, upi, rpu, (unsigned)total, (int)(nnz_total - 4),
                       (long)batch * ((long)M + 1) - 1, ktiles == 8 ? 3 : ktiles == 4 ? 2 : ktiles == 2 ? 1 : 0, 512);
    return check_launch();
  }
  if ((long)K * N * 4 > 128L * 1024) return 1;  // only the quad form's tiles cover this B, and it cannot run: not taken
  int units_per_item = 1;
  while ((long)batch * units_per_item < 3L * cus && ((long)M + units_per_item * 2 - 1) / (units_per_item * 2) >= 64)
    units_per_item *= 2;
  const int rows_per_unit = (int)(((long)M + units_per_item - 1) / units_per_item);
  const long total = (long)batch * units_per_item;
  if (total > 0x7fffffffL) return MI_ERANGE;
  const unsigned grid = (unsigned)(total < cus ? total : cus);
  const size_t lds = (size_t)K * N * 4;
  const int G = pow2_ceil(N / 4);
#define MI_SDDMM_LDSB(G_)                                                                                            \
  do {                                                                                                                \
    auto k = sddmm_ldsb_kernel<G_>;                                                                                   \
    if (lds > 64 * 1024) MI_HIP_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL(k, dim3(grid), dim3(kWaves * 64), lds, s, rowptr, col, dC, B, out, M, K, N, (long)lddc,         \
                       (long)strideDC, (long)ldb, (long)strideB, units_per_item, rows_per_unit, (unsigned)total);      \
  } while (0)
  switch (G) {
    case 1: MI_SDDMM_LDSB(1); break;
    case 2: MI_SDDMM_LDSB(2); break;
    case 4: MI_SDDMM_LDSB(4); break;
    case 8: MI_SDDMM_LDSB(8); break;
    default: MI_SDDMM_LDSB(16); break;
  }
#undef MI_SDDMM_LDSB
  return check_launch();
}

}  // namespace mi

extern "C" {

#if defined(MI_LDSB_TIMING) || defined(MI_LDSB_PROBE)
// developer builds of this file alone (tools/probes/ldsb_timing.py, tools/probes/ldsb_ab.py): the plan's launcher without
// the dispatcher; stamps_out = null: no synchronisation, nothing read back
int mi_ldsb_probe(const int32_t* rowptr, const int32_t* col, const float* val, const float* B, float* C, int32_t batch,
                  int32_t M, int32_t K, int32_t N, int64_t nnz, void* stamps_out) {
  const int st = mi::launch_spmm_ldsb(rowptr, col, val, B, C, batch, M, K, N, N, N, (int64_t)K * N, (int64_t)M * N, nullptr,
                                      0x7fffffff, nullptr, nullptr, nnz);
  if (st != MI_OK || !stamps_out) return st;
  MI_HIP_TRY(hipDeviceSynchronize());
#ifdef MI_LDSB_TIMING
  MI_HIP_TRY(hipMemcpyFromSymbol(stamps_out, HIP_SYMBOL(g_ldsq_stamps), sizeof(g_ldsq_stamps)));
  MI_HIP_TRY(hipMemcpyFromSymbol(static_cast<char*>(stamps_out) + sizeof(g_ldsq_stamps), HIP_SYMBOL(g_ldsq_wave_end),
                                 sizeof(g_ldsq_wave_end)));
#endif
  return MI_OK;
}
#endif

int mi_spmm_ldsb_set_form(int form) {
  if (form < -1 || form > 1) return MI_EINVAL;
  mi::g_ldsb_form.store(form, std::memory_order_relaxed);
  return MI_OK;
}

int mi_sddmm_csr_batched_f32(const int32_t* rowptr, const int32_t* col, int64_t nnz_total, int32_t batch, int32_t M,
                             int32_t K, int32_t N, const float* dC, int64_t lddc, int64_t strideDC, const float* B,
                             int64_t ldb, int64_t strideB, float* out_val, mi_stream_t stream) {
  if (batch < 0 || M < 0 || K < 0 || N < 0 || nnz_total < 0 || strideDC < 0 || strideB < 0) return MI_EINVAL;
  if (nnz_total > 0x7fffffffLL) return MI_ERANGE;
  if (batch == 0 || M == 0 || nnz_total == 0) return MI_OK;
  if (!rowptr || !col || !out_val || !dC || !B || lddc < N || ldb < N) return MI_EINVAL;
  const bool vec = N % 4 == 0 && lddc % 4 == 0 && ldb % 4 == 0 && strideDC % 4 == 0 && strideB % 4 == 0 &&
                   mi::aligned16(dC) && mi::aligned16(B);
  // taken where the forward's LDS-resident-B plan is (enough rows to fill the chip, rows of a few non-zeros at least)
  if (!vec || !mi::sddmm_ldsb_fits(K, N) || (long)batch * M < 16384 || nnz_total < 4L * batch * M) return 1;
  return mi::launch_sddmm_ldsb(rowptr, col, dC, B, out_val, batch, M, K, N, lddc, strideDC, ldb, strideB,
                               static_cast<hipStream_t>(stream), nnz_total);
}

}  // extern "C"
