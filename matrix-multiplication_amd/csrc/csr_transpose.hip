// Device CSR transpose for gfx950 (include/mi_spmm.h: mi_csr_transpose_f32, mi_csr_transpose_batched_f32)
//   A (M×K, CSR) → Aᵀ (K×M, CSR), columns of Aᵀ (= rows of A) ascending inside every row, stable.
// Used by the backward pass grad_B = Aᵀ·dC; no counterpart in the reference (its backward
// re-sparsifies a strided view, matmuls.py:319-325, SURVEY.md §8a defect 1).
//
// A transpose is a stable sort of the entries by column.  This file does it with hand-written
// least-significant-digit counting passes over the column index (10 bits per pass: two passes for
// K·batch ≤ 2²⁰, e.g. the 1M × 1M matrix of BASELINE config C3), built for the memory system
// rather than around a generic sort:
//   * the entry travels as 8 bytes {remaining key bits | row, value}; the first pass reads the CSR
//     arrays directly (row ids are filled into LDS row by row, no expanded row array, no per-entry
//     search) and the last pass writes t_col / t_val directly — no pack / unpack passes;
//   * a workgroup owns a tile of 8192 consecutive entries: 16 waves rank their 512 entries each with
//     wave-private 16-bit LDS counters (a lane's rank = the counter + the number of lanes of the same
//     step holding the same digit below it, found by collision: lanes tag the counter word with their
//     id and read it back — in entry order → stable, no atomics), the per-wave counts are prefixed per
//     digit, the tile is reordered by digit in LDS and written out so that every digit's entries
//     leave as one contiguous run;
//   * the scatter kernel is persistent and software-pipelined over its tiles: while a tile is ranked,
//     the previous tile streams out of LDS and the next one is loaded, slice by slice (see
//     tr_scatter_staged_kernel for what that takes from the compiler's wait counting);
//   * per-(tile, digit) counts live in a tile-major table that is prefixed column-wise by three
//     small launches (digit-major order of the scan, coalesced accesses);
//   * for two passes the tiles of the last pass never straddle two low-digit bins, so the row
//     offsets of Aᵀ fall out of the scanned table:  t_rowptr[hi·2^b + lo] = offset(hi, first tile of
//     bin lo) — no histogram over K columns, no atomics anywhere, results are deterministic.
// Larger key spaces (K·batch > 2²⁰) take three passes and keep the full key to find the row
// boundaries with one extra pass; entries that do not fit the 8-byte form ((remaining key bits) +
// (row bits) > 32) or an 11-bit digit take a simpler path that scatters straight from registers.
#include <atomic>

#include <type_traits>

#include "mi_common.h"
#include "csr_transpose_internal.h"

namespace {

constexpr int TR_TILE = 8192;
constexpr int TR_THREADS = 1024;
constexpr int TR_WAVES = TR_THREADS / 64;
constexpr int TR_PER = TR_TILE / TR_THREADS;  // entries per thread
constexpr int TR_GROUPS = 256;                // tile groups of the column-wise table scan
#ifndef MI_TR_COUNT_THREADS
#define MI_TR_COUNT_THREADS 256
#endif
constexpr int TR_COUNT_THREADS = MI_TR_COUNT_THREADS;  // count kernel (512: same, 1024: first pass 133 -> 187 µs)
constexpr int TR_GRID = 256;                  // persistent scatter workgroups: one per CU of an MI355X
// One-sweep plan: the tiles of a pass are cut into TR_LB_GROUPS contiguous groups; a group's per-digit base comes from
// the histogram launch, offsets inside a group by decoupled look-back over at most its own tiles.  XCD x (workgroups
// with blockIdx % 8 == x) works on groups 4x … 4x+3 — a contiguous eighth of the tiles, as in the table plan — with
// 8 workgroups per group, so a look-back usually finds a finished prefix within 8 tiles.
constexpr int TR_LB_GROUPS = 32;
constexpr unsigned TR_LB_AGG = 1u << 30, TR_LB_PREFIX = 2u << 30, TR_LB_VALUE = (1u << 30) - 1u;
constexpr int TR_LB_WALK = 8;                 // predecessors read in one go
constexpr int TR_LB_SPIN_LIMIT = 1 << 22;     // polls before a look-back gives up (seconds; a correct run needs a handful)

__host__ __device__ inline int bits_for(unsigned long long n) {  // bits needed for values 0 … n-1
  int b = 0;
  while (b < 63 && (1ULL << b) < n) ++b;
  return b < 1 ? 1 : b;
}

struct TrArgs {
  // problem
  const int* rowptr;  // [batch][M+1], global offsets
  const int* col;
  const float* val;
  long nnz;
  int batch, M, K;
  // pass description
  int shift, bits;       // digit = (keyfield >> shift) & ((1 << bits) - 1)
  int row_bits;          // packed form: a = (keyfield << row_bits) | row
  int drop_after_first;  // two-pass packed: low digit dropped from the key after the first pass
  // tiling of this pass's input
  const int2* desc;  // per tile {start, len}, or nullptr: tile t = [t·TR_TILE, …)
  int ntiles;        // grid size (an upper bound when desc is given; unused tiles have len 0)
  const int* used_tiles;  // device count of the tiles in desc that are in use (they come first), or nullptr
  // tables
  int* table;            // [ntiles + 1][1 << bits], counts then (after the scan) global offsets
  int* tile_row;         // first pass: flat rowptr index of the row holding the tile's first entry
  // intermediate arrays (input of non-first passes / output of non-last passes)
  const uint2* in_packed;
  uint2* out_packed;
  const unsigned *in_key, *in_row;
  const float* in_val;
  unsigned *out_key, *out_row;
  float* out_val;
  // final output
  int* t_col;
  float* t_val;
  unsigned* keys_out;  // three passes: full key per output position (row boundaries), else nullptr
  uint2* dump;         // [2·TR_GRID]: where a workgroup's not-yet-existing previous tile "streams out" to
  // one-sweep plan (decoupled look-back, see tr_hist_kernel): no per-(tile, digit) table is scanned ahead of the scatter
  unsigned* status;       // [ntiles][1 << bits] look-back words of this pass: flag (bits 31-30) | value
  const int* grp_first;   // [TR_LB_GROUPS + 1] first tile of every tile group of this pass
  const int* grp_base;    // [TR_LB_GROUPS][1 << bits] global offset of (group, digit) at the group's first tile
  int* tickets;           // [TR_LB_GROUPS] next tile of each group (relative to its first)
  const int* tile_bin;    // last pass: the low digit (bin) every tile belongs to
  const int* first_tile;  // last pass: first tile of every bin
  int* rowoff;            // last pass: [bins][1 << bits] offset of (bin, digit) = row offset of Aᵀ for that column
  unsigned* lb_dump;      // [TR_THREADS] where lanes without a digit publish to
  int* errflag;           // set when a look-back spin gives up (never on a correct run)
};

// flat rowptr index i (in [lo, hi]) of the row that holds entry e: the last i with rowptr[i] ≤ e.
// (In the batched layout an item's end slot equals the next item's first slot, so "the last i"
// is never an end slot for e < nnz.)
__device__ __forceinline__ int row_of(const int* __restrict__ rowptr, int lo, int hi, int e) {
  while (lo < hi) {
    const int mid = lo + ((hi - lo + 1) >> 1);
    if (rowptr[mid] <= e) lo = mid; else hi = mid - 1;
  }
  return lo;
}

struct Entry {
  unsigned key;  // key field of this pass (digit = (key >> shift) & mask)
  unsigned row;
  float val;
};

// Loads of block-uniform metadata through the scalar unit.  The tables were written by EARLIER launches,
// so they are constant for this kernel; the compiler cannot see that (the kernel stores to global
// memory) and would use vector loads — whose s_waitcnt vmcnt(0) also waits for every older store.
typedef const __attribute__((address_space(4))) int* tr_kptr;
typedef int tr_int2v __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) tr_int2v* tr_kptr2;
__device__ __forceinline__ int sload(const int* p) { return *(tr_kptr)(unsigned long long)p; }
__device__ __forceinline__ int2 sload(const int2* p) {
  const tr_int2v v = *(tr_kptr2)(unsigned long long)p;
  return make_int2(v.x, v.y);
}

// tile bounds of workgroup t
__device__ __forceinline__ void tile_bounds(const TrArgs& a, int t, long& start, int& len) {
  if (a.desc) {
    const int2 d = a.desc[t];
    start = d.x;
    len = d.y;
  } else {
    start = (long)t * TR_TILE;
    const long rest = a.nnz - start;
    len = rest < 0 ? 0 : (rest < TR_TILE ? (int)rest : TR_TILE);
  }
}

template <bool FIRST, bool PACKED>
__device__ __forceinline__ Entry load_entry(const TrArgs& a, long p, int row_lo, int row_hi) {
  Entry e;
  if (FIRST) {
    const int idx = row_of(a.rowptr, row_lo, row_hi, (int)p);
    unsigned c = (unsigned)a.col[p];
    unsigned r = (unsigned)idx;
    if (a.batch > 1) {
      const unsigned item = (unsigned)idx / (unsigned)(a.M + 1);
      r = (unsigned)idx - item * (unsigned)(a.M + 1);
      c += item * (unsigned)a.K;
    }
    e.key = c;
    e.row = r;
    e.val = a.val[p];
  } else if (PACKED) {
    const uint2 w = a.in_packed[p];
    e.key = w.x >> a.row_bits;
    e.row = w.x & ((1u << a.row_bits) - 1u);
    e.val = __builtin_bit_cast(float, w.y);
  } else {
    e.key = a.in_key[p];
    e.row = a.in_row[p];
    e.val = a.in_val[p];
  }
  return e;
}

// ---------------------------------------------------------------------------------------------
// Count: per-tile digit histogram (LDS atomics), written tile-major.  The first pass also records
// the row that holds the tile's first entry (one thread's binary search, hidden behind the rest).
// ---------------------------------------------------------------------------------------------
template <bool FIRST, bool PACKED>
__global__ __launch_bounds__(TR_COUNT_THREADS) void tr_count_kernel(TrArgs a) {
  extern __shared__ int hist[];
  const int nb = 1 << a.bits;
  const int t = blockIdx.x;
  for (int d = threadIdx.x; d < nb; d += TR_COUNT_THREADS) hist[d] = 0;
  long start;
  int len;
  tile_bounds(a, t, start, len);
  if (FIRST && threadIdx.x == 0) {
    const int last = a.batch * (a.M + 1) - 1;
    a.tile_row[t] = start < a.nnz ? row_of(a.rowptr, 0, last, (int)start) : last;
    if (t == (int)gridDim.x - 1) a.tile_row[t + 1] = last;
  }
  __syncthreads();
  const unsigned mask = (unsigned)nb - 1u;
  if (FIRST && a.batch == 1) {
    for (int i = threadIdx.x; i < len; i += 8 * TR_COUNT_THREADS) {  // eight independent loads in flight per thread
      unsigned k[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) k[u] = i + u * TR_COUNT_THREADS < len ? (unsigned)a.col[start + i + u * TR_COUNT_THREADS] : 0u;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i + u * TR_COUNT_THREADS < len) atomicAdd(&hist[(k[u] >> a.shift) & mask], 1);
    }
  } else if (FIRST) {
    // batched: the key includes the item, i.e. needs the row; two threads bound the tile's rows first
    __shared__ int row_bounds[2];
    const int last = a.batch * (a.M + 1) - 1;
    if (threadIdx.x < 2 && len > 0)
      row_bounds[threadIdx.x] = row_of(a.rowptr, 0, last, (int)(threadIdx.x == 0 ? start : start + len - 1));
    __syncthreads();
    for (int i = threadIdx.x; i < len; i += TR_COUNT_THREADS) {
      const Entry e = load_entry<true, PACKED>(a, start + i, row_bounds[0], row_bounds[1]);
      atomicAdd(&hist[(e.key >> a.shift) & mask], 1);
    }
  } else {
    for (int i = threadIdx.x; i < len; i += 8 * TR_COUNT_THREADS) {
      unsigned k[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        k[u] = i + u * TR_COUNT_THREADS < len ? (PACKED ? (a.in_packed[start + i + u * TR_COUNT_THREADS].x >> a.row_bits) : a.in_key[start + i + u * TR_COUNT_THREADS]) : 0u;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i + u * TR_COUNT_THREADS < len) atomicAdd(&hist[(k[u] >> a.shift) & mask], 1);
    }
  }
  __syncthreads();
  int* out = a.table + (long)t * nb;
  for (int d = threadIdx.x; d < nb; d += TR_COUNT_THREADS) out[d] = hist[d];
}

// ---------------------------------------------------------------------------------------------
// Column-wise scan of the tile-major table:  table[t][d] ← base[d] + Σ_{t' < t} table[t'][d],
// base = exclusive scan over d of the column sums — i.e. the exclusive scan in (digit major, tile
// minor) order.  rows = ntiles + 1 (the extra all-zero row ends up holding every digit's end).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tr_group_sums_kernel(const int* __restrict__ table, int rows, int nb,
                                                            int* __restrict__ gsum) {
  const int d = blockIdx.y * 256 + threadIdx.x;
  if (d >= nb) return;
  const int per = (rows + TR_GROUPS - 1) / TR_GROUPS;
  const int t0 = blockIdx.x * per, t1 = t0 + per < rows ? t0 + per : rows;
  int s = 0;
#pragma unroll 8
  for (int t = t0; t < t1; ++t) s += table[(long)t * nb + d];
  gsum[blockIdx.x * nb + d] = s;
}

// One workgroup per 64 digits: wave w scans groups [w·G/16, (w+1)·G/16) of its digits (lane = digit),
// the waves' partial sums meet in LDS.  gsum[g][d] ← Σ_{g' < g} gsum[g'][d];  dtot[d] ← Σ_g gsum[g][d].
// (As one workgroup walking the groups one after the other this took 19–31 µs per pass.)
__global__ __launch_bounds__(1024) void tr_group_bases_kernel(int* __restrict__ gsum, int nb, int* __restrict__ dtot) {
  constexpr int PER = TR_GROUPS / 16;
  __shared__ int part[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int d = blockIdx.x * 64 + lane;
  int v[PER];
  int tot = 0;
  if (d < nb) {
#pragma unroll
    for (int u = 0; u < PER; ++u) v[u] = gsum[(wave * PER + u) * nb + d];
#pragma unroll
    for (int u = 0; u < PER; ++u) tot += v[u];
  }
  part[wave][lane] = tot;
  __syncthreads();
  int before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const int p = part[w][lane];
    if (w < wave) before += p;
    all += p;
  }
  if (d < nb) {
    int run = before;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      gsum[(wave * PER + u) * nb + d] = run;
      run += v[u];
    }
    if (wave == 0) dtot[d] = all;
  }
}

__global__ __launch_bounds__(256) void tr_apply_kernel(int* __restrict__ table, int rows, int nb,
                                                       const int* __restrict__ gsum, const int* __restrict__ dtot) {
  // base of this thread's digit: Σ dtot[d'] over d' < d (digits in chunks of 256, this block's chunk last)
  __shared__ int wsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int carry = 0, base = 0;
  for (int c = 0; c <= (int)blockIdx.y; ++c) {
    const int dd = c * 256 + threadIdx.x;
    const int tot = dd < nb ? dtot[dd] : 0;
    int incl = tot;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
      const int v = __shfl_up(incl, s, 64);
      if (lane >= s) incl += v;
    }
    __syncthreads();
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int wbase = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int v = wsum[w];
      if (w < wave) wbase += v;
      all += v;
    }
    base = carry + wbase + incl - tot;
    carry += all;
  }
  const int d = blockIdx.y * 256 + threadIdx.x;
  if (d >= nb) return;
  const int per = (rows + TR_GROUPS - 1) / TR_GROUPS;
  const int t0 = blockIdx.x * per, t1 = t0 + per < rows ? t0 + per : rows;
  int run = gsum[blockIdx.x * nb + d] + base;
#pragma unroll 8
  for (int t = t0; t < t1; ++t) {
    const int v = table[(long)t * nb + d];
    table[(long)t * nb + d] = run;
    run += v;
  }
}

// ---------------------------------------------------------------------------------------------
// Tiles of the last of two passes: bin lo of the first pass occupies positions
// [off(lo, tile 0), off(lo + 1, tile 0)) of the intermediate array; cut every bin into tiles of its
// own so that no tile holds two low digits.  first_tile[lo] (and [nb0] = total) + desc[t].
// One workgroup; tiles are written cooperatively bin by bin.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void tr_bin_tiles_kernel(const int* __restrict__ table0, int nb0, long nnz,
                                                            int max_tiles, int* __restrict__ first_tile,
                                                            int2* __restrict__ desc) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int d0 = 0; d0 < nb0; d0 += 1024) {
    const int lo = d0 + threadIdx.x;
    int nt = 0;
    long b0 = 0, b1 = 0;
    if (lo < nb0) {
      b0 = table0[lo];  // row 0 of the scanned table: offset of (lo, tile 0)
      b1 = lo + 1 < nb0 ? table0[lo + 1] : nnz;
      nt = (int)((b1 - b0 + TR_TILE - 1) / TR_TILE);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = nt;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
      const int v = __shfl_up(incl, s, 64);
      if (lane >= s) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int wbase = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const int v = wsum[w];
      if (w < wave) wbase += v;
      all += v;
    }
    const int first = carry_s + wbase + incl - nt;
    if (lo < nb0) {
      first_tile[lo] = first;
      for (int j = 0; j < nt && first + j < max_tiles; ++j) {
        const long s = b0 + (long)j * TR_TILE;
        desc[first + j] = make_int2((int)s, (int)(b1 - s < TR_TILE ? b1 - s : TR_TILE));
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) carry_s += all;
    __syncthreads();
  }
  const int total = carry_s;
  if (threadIdx.x == 0) first_tile[nb0] = total;
  for (int t = total + threadIdx.x; t < max_tiles; t += 1024) desc[t] = make_int2(0, 0);
}


// ---------------------------------------------------------------------------------------------
// One-sweep plan (two packed, staged passes; one matrix; nnz < 2³⁰): the columns are read ONCE for counting.
// tr_hist_kernel is the first pass's count kernel made persistent — per-tile low-digit histograms into the tile-major
// table (scanned as before; the first scatter launch is the table plan's) — that ALSO counts, in the same sweep,
//   h2g[g][hi]  entries with high digit hi whose LOW digit lies in bin group g (bins g·2^gshift …): the last pass walks
//               the bins in order, so a group of bins is a contiguous range of its tiles → (after tr_lb_setup_kernel)
//               the global offset of (group g, digit hi) at the group's first tile.
// Inside a group the last pass finds a tile's offsets by decoupled look-back (tr_scatter_staged_kernel<false, true, 1>):
// the second count pass over the intermediate array (a second read of all 8-byte entries), its three scan launches
// and the bin-tiles launch are gone.  (Tried and dropped: giving every workgroup of the FIRST pass a contiguous run of
// tiles with a running offset in registers — no table, no look-back — ran 0.73 ms against 0.58: with 256 runs × 1024
// digits open at once the partially written lines no longer meet their neighbours in the L2s; the XCD-interleaved
// tile order of the table plan stays.)
// Persistent: workgroup w counts tiles [w·tpw, (w+1)·tpw), the next tile's columns in flight while one is counted.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TR_THREADS) void tr_hist_kernel(TrArgs a, int bits0, int bits1, int gshift,
                                                             int* __restrict__ h2g) {
  extern __shared__ int hist_lds[];
  const int nb0 = 1 << bits0, nb1 = 1 << bits1;
  const int G = ((nb0 - 1) >> gshift) + 1;
  int* h1 = hist_lds;        // [nb0] this tile
  int* h2 = hist_lds + nb0;  // [G][nb1] this workgroup's whole run
  const int tid = threadIdx.x, lane = tid & 63;
  const int ntiles = a.ntiles;
  const int tpw = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t0 = blockIdx.x * tpw, t1 = t0 + tpw < ntiles ? t0 + tpw : ntiles;
  for (int i = tid; i < nb0 + G * nb1; i += TR_THREADS) hist_lds[i] = 0;
  __syncthreads();
  if (t0 >= t1) return;
  const int last = a.batch * (a.M + 1) - 1;
  if (tid < 64) {  // the row holding every tile's first entry (a binary search per lane; wave 0 joins the counting late)
    for (int i = lane; i < t1 - t0; i += 64) {
      const long st = (long)(t0 + i) * TR_TILE;
      a.tile_row[t0 + i] = st < a.nnz ? row_of(a.rowptr, 0, last, (int)st) : last;
    }
    if (t1 == ntiles && lane == 0) a.tile_row[ntiles] = last;
  }
  const unsigned m0 = (unsigned)nb0 - 1u, m1 = (unsigned)nb1 - 1u;
  unsigned c[TR_PER], cn[TR_PER];
  auto fetch = [&](int t, unsigned* dst) {
#pragma unroll
    for (int k = 0; k < TR_PER; ++k) {
      const long q = (long)t * TR_TILE + k * TR_THREADS + tid;
      dst[k] = q < a.nnz ? (unsigned)a.col[q] : ~0u;
    }
  };
  fetch(t0, c);
  for (int t = t0; t < t1; ++t) {
    if (t + 1 < t1) fetch(t + 1, cn);
#pragma unroll
    for (int k = 0; k < TR_PER; ++k) {
      if (c[k] != ~0u) {
        const unsigned lo = c[k] & m0, hi = (c[k] >> bits0) & m1;
        atomicAdd(&h1[lo], 1);
        atomicAdd(&h2[(lo >> gshift) * nb1 + hi], 1);
      }
    }
    __syncthreads();
    for (int i = tid; i < nb0; i += TR_THREADS) {
      a.table[(long)t * nb0 + i] = h1[i];
      h1[i] = 0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TR_PER; ++k) c[k] = cn[k];
  }
  for (int i = tid; i < G * nb1; i += TR_THREADS) {
    const int v = h2[i];
    if (v != 0) atomicAdd(&h2g[i], v);
  }
}

// exclusive scan of one value per thread over a 1024-thread workgroup (wsum: 16 ints of LDS); total → all threads
__device__ __forceinline__ int tr_block_scan_1024(int v, int* wsum, int& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) {
    const int u = __shfl_up(incl, s, 64);
    if (lane >= s) incl += u;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int wbase = 0, all = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const int u = wsum[w];
    if (w < wave) wbase += u;
    all += u;
  }
  __syncthreads();
  total = all;
  return wbase + incl - v;
}

// One workgroup, after the first pass's table is scanned (its row 0 = every low digit's global offset): the bins of the
// intermediate array cut into the last pass's tiles (as tr_bin_tiles_kernel does) with every tile's bin, the last
// pass's per-(bin group, high digit) offsets (h2g, in place) and the first tile of each of its tile groups.
__global__ __launch_bounds__(1024) void tr_lb_setup_kernel(int bits0, int bits1, int gshift, long nnz, int max_tiles1,
                                                           const int* __restrict__ table0, int* __restrict__ h2g,
                                                           int* __restrict__ hibase, int* __restrict__ first_tile,
                                                           int2* __restrict__ desc, int* __restrict__ tile_bin,
                                                           int* __restrict__ grp_first2) {
  __shared__ int wsum[16];
  __shared__ int s_first[1025];
  const int nb0 = 1 << bits0, nb1 = 1 << bits1;
  const int G = ((nb0 - 1) >> gshift) + 1;
  const int tid = threadIdx.x;
  int total = 0;
  // high digit first: its loads are in flight while the bins are cut
  int h2v[TR_LB_GROUPS];
#pragma unroll
  for (int g = 0; g < TR_LB_GROUPS; ++g) h2v[g] = (tid < nb1 && g < G) ? h2g[g * nb1 + tid] : 0;
  int lobase = 0, cnt = 0;
  if (tid < nb0) {
    lobase = table0[tid];
    cnt = (tid + 1 < nb0 ? table0[tid + 1] : (int)nnz) - lobase;
  }
  // the bins of the intermediate array as tiles of the last pass (no tile holds two low digits)
  const int nt = tid < nb0 ? (cnt + TR_TILE - 1) / TR_TILE : 0;
  int total_tiles = 0;
  const int first = tr_block_scan_1024(nt, wsum, total_tiles);
  if (tid < nb0) {
    s_first[tid] = first;
    first_tile[tid] = first;
    for (int j = 0; j < nt && first + j < max_tiles1; ++j) {
      const int st = lobase + j * TR_TILE;
      desc[first + j] = make_int2(st, cnt - j * TR_TILE < TR_TILE ? cnt - j * TR_TILE : TR_TILE);
      tile_bin[first + j] = tid;
    }
  }
  if (tid == 0) first_tile[nb0] = total_tiles;
  for (int t = total_tiles + tid; t < max_tiles1; t += 1024) desc[t] = make_int2(0, 0);
  int cnt2 = 0;
#pragma unroll
  for (int g = 0; g < TR_LB_GROUPS; ++g) {
    const int v = h2v[g];
    h2v[g] = cnt2;
    cnt2 += v;
  }
  const int hb = tr_block_scan_1024(cnt2, wsum, total);
  if (tid < nb1) {
    hibase[tid] = hb;
#pragma unroll
    for (int g = 0; g < TR_LB_GROUPS; ++g)
      if (g < G) h2g[g * nb1 + tid] = h2v[g] + hb;
  }
  if (tid == 0) hibase[nb1] = (int)nnz;
  __syncthreads();  // s_first complete
  if (tid <= TR_LB_GROUPS) {
    const int bin = tid << gshift;
    grp_first2[tid] = (tid < G && bin < nb0) ? s_first[bin] : total_tiles;
  }
}

// Row offsets of Aᵀ for the one-sweep plan: rowoff[lo][hi], written by the last pass at every bin's first tile, is the
// offset of column hi·2^bits0 + lo; an empty bin has no tile, its columns start where the next non-empty bin's do.
__global__ __launch_bounds__(256) void tr_rowptr_lb_kernel(const int* __restrict__ rowoff, const int* __restrict__ first_tile,
                                                           const int* __restrict__ hibase, int bits0, int bits1, int batch,
                                                           int K, long nnz, int* __restrict__ t_rowptr) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)batch * ((long)K + 1);
  if (idx >= total) return;
  const long b = idx / ((long)K + 1);
  const long key = b * K + (idx - b * ((long)K + 1));
  int v;
  if (key >= (long)batch * K) {
    v = (int)nnz;
  } else {
    const int nb0 = 1 << bits0;
    int lo = (int)(key & (nb0 - 1));
    const int hi = (int)(key >> bits0);
    while (lo < nb0 && first_tile[lo] == first_tile[lo + 1]) ++lo;
    v = lo < nb0 ? rowoff[((long)lo << bits1) + hi] : hibase[hi + 1];
  }
  t_rowptr[idx] = v;
}

// ---------------------------------------------------------------------------------------------
// Scatter, generic form (entries that do not fit 8 bytes, 11-bit digits, the last of three passes):
// 16 waves × 512 entries per tile, ranked as described above and scattered straight from registers.
// Persistent — one workgroup per CU walks a fixed list of tiles, no inter-workgroup dependence — so the
// next tile's entries are in flight while the current tile's stores drain.  The main path is
// tr_scatter_staged_kernel below.
// ---------------------------------------------------------------------------------------------
#ifdef MI_TR_TIMING
// per-phase cycle sums of wave 0 (tools/probes/tr_probe.cpp).  The sums are kept in scalar registers and
// written once when the workgroup ends: an atomic per stamp sits in the same in-order memory queue as
// the tile's loads and stores and slows exactly what is being measured.
__device__ unsigned long long g_tr_phase[16];  // [0, 8): later passes, [8, 16): first pass
#define TR_STAMP_INIT unsigned long long stamp_ = __builtin_readcyclecounter(), acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define TR_STAMP(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); acc_[k] += now_ - stamp_; stamp_ = now_; } while (0)
#define TR_STAMP_FLUSH(first) do { if (threadIdx.x == 0) for (int k_ = 0; k_ < 8; ++k_) atomicAdd(&g_tr_phase[k_ + ((first) ? 8 : 0)], acc_[k_]); } while (0)
#else
#define TR_STAMP_INIT do {} while (0)
#define TR_STAMP(k) do {} while (0)
#define TR_STAMP_FLUSH(first) do {} while (0)
#endif

// An entry as the scatter kernel carries it: already in the form it leaves the pass in, plus its
// digit — 3 registers on the main path (a 1024-thread workgroup has 128 VGPRs per lane; the first
// form kept key, row, value, digit and rank apart and spilled to scratch inside the rank loop, which
// is what made a tile take 24–39 µs).
struct Slot {
  unsigned a;    // LAST: row | PACKED: the packed word | else: key as the next pass wants it
  unsigned b;    // !PACKED: row;  LAST with keys_out: the full key
  float val;
  unsigned d;    // digit of this pass; the rank is kept in the upper half once known
};

template <bool FIRST, bool LAST, bool PACKED>
__device__ __forceinline__ Slot load_slot(const TrArgs& a, long p, int row_lo, int row_hi, unsigned mask) {
  const Entry e = load_entry<FIRST, PACKED>(a, p, row_lo, row_hi);
  Slot s;
  s.d = (e.key >> a.shift) & mask;
  s.val = e.val;
  const unsigned keep = a.drop_after_first ? (e.key >> a.bits) : e.key;  // two passes: the low digit is dropped
  s.b = LAST ? e.key : e.row;
  if (LAST) s.a = e.row;
  else if (PACKED) s.a = (keep << a.row_bits) | e.row;
  else s.a = keep;
  return s;
}

template <bool FIRST, bool LAST, bool PACKED>
__global__ __launch_bounds__(TR_THREADS) void tr_scatter_kernel(TrArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nb = 1 << a.bits;
  unsigned short* cntw = reinterpret_cast<unsigned short*>(smem);  // [TR_WAVES][nb]
  int* binstart = reinterpret_cast<int*>(smem + (size_t)TR_WAVES * nb * 2);  // [nb] tile-local start of a digit
  int* gadj = binstart + nb;                                                  // [nb] global offset − binstart
  // first pass: flat rowptr index of the row holding each entry of the tile, filled row by row
  unsigned* rowid = reinterpret_cast<unsigned*>(gadj + nb);  // [TR_TILE] (FIRST)
  __shared__ int wsum[16];
  __shared__ int carry_s;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned mask = (unsigned)nb - 1u;
  unsigned short* mycnt = cntw + wave * nb;
  constexpr int GPT = 2;  // digits per thread in the prefix phase (nb ≤ 2048)

  Slot e[TR_PER];
  int goff[GPT];
  int len = 0;
  long start = 0;
  int row_lo = 0, row_hi = 0, rp0 = 0, rp1 = 0;  // first pass: the tile's rows; this lane's row bounds
  // fetch a tile's entries and its row of the offset table into registers
  auto fetch = [&](int t) {
    tile_bounds(a, t, start, len);
    if (FIRST) {
      // the rows come later, from LDS (rowid): a per-entry binary search in rowptr here cost 8 × ~8
      // dependent global loads per lane and was the larger half of this kernel's time.  Lane l of
      // wave w prefetches the bounds of row row_lo + 16·l + w (the first 1024 rows of the tile).
      row_lo = a.tile_row[t];
      row_hi = a.tile_row[t + 1];
      const int r = row_lo + lane * TR_WAVES + wave;
      if (r <= row_hi) {  // the last slot (row_hi of the last tile) has no successor: it holds no entry either
        const int last_slot = a.batch * (a.M + 1) - 1;
        rp0 = a.rowptr[r];
        rp1 = a.rowptr[r < last_slot ? r + 1 : last_slot];
      }
#pragma unroll
      for (int c = 0; c < TR_PER; ++c) {
        const int i = wave * (TR_TILE / TR_WAVES) + c * 64 + lane;
        if (i < len) {
          e[c].a = (unsigned)a.col[start + i];
          e[c].val = a.val[start + i];
        }
      }
    } else {
#pragma unroll
      for (int c = 0; c < TR_PER; ++c) {
        const int i = wave * (TR_TILE / TR_WAVES) + c * 64 + lane;
        if (i < len) e[c] = load_slot<FIRST, LAST, PACKED>(a, start + i, 0, 0, mask);
      }
    }
#pragma unroll
    for (int g = 0; g < GPT; ++g) {
      const int d = g * TR_THREADS + tid;
      goff[g] = d < nb ? a.table[(long)t * nb + d] : 0;
    }
  };

  // XCD-aware tile order: workgroup b runs on XCD b % 8 (private L2).  Give every XCD a CONTIGUOUS
  // range of tiles and let its 32 workgroups walk it together: at any time an XCD then writes 32
  // consecutive tiles' runs of every digit — pieces that are adjacent in memory — so whole lines
  // assemble in ITS L2 before they are written back.
  int t = blockIdx.x, t_end = a.ntiles, t_step = gridDim.x;
  if (gridDim.x == TR_GRID) {
    const int per = (a.ntiles + 7) / 8, xcd = blockIdx.x & 7;
    t = xcd * per + (blockIdx.x >> 3);
    t_end = (xcd + 1) * per < a.ntiles ? (xcd + 1) * per : a.ntiles;
    t_step = TR_GRID / 8;
  }
  TR_STAMP_INIT;
  if (t < t_end) fetch(t);
  for (; t < t_end; t += t_step) {
    const int cur_len = len;  // block-uniform
    if (cur_len > 0) {
      TR_STAMP(0);
      for (int i = tid; i < TR_WAVES * nb / 2; i += TR_THREADS) reinterpret_cast<unsigned*>(cntw)[i] = 0u;
      if (tid == 0) carry_s = 0;
      if (FIRST) {
        // every row of the tile writes its flat index over its own entries (wave-uniform bounds, 64
        // positions per instruction); each position of the tile belongs to exactly one row
        const long t_lo = start, t_hi = start + cur_len;
        auto fill = [&](int r, long b, long en) {
          b = b < t_lo ? t_lo : b;
          en = en > t_hi ? t_hi : en;
          for (long q = b + lane; q < en; q += 64) rowid[q - t_lo] = (unsigned)r;
        };
        const int span = row_hi - row_lo + 1;
        for (int k = 0; k < 64 && k * TR_WAVES + wave < span; ++k)
          fill(row_lo + k * TR_WAVES + wave, __builtin_amdgcn_readlane(rp0, k), __builtin_amdgcn_readlane(rp1, k));
        for (int r = row_lo + 64 * TR_WAVES + wave; r <= row_hi; r += TR_WAVES)  // tiles spanning > 1024 rows
          fill(r, a.rowptr[r], a.rowptr[r < a.batch * (a.M + 1) - 1 ? r + 1 : r]);
      }
      __syncthreads();  // counters zeroed (rows known); the previous tile's LDS reads are complete
      if (FIRST) {
#pragma unroll
        for (int c = 0; c < TR_PER; ++c) {
          const int i = wave * (TR_TILE / TR_WAVES) + c * 64 + lane;
          if (i < cur_len) {
            unsigned idx = rowid[i], key = e[c].a, r = idx;
            if (a.batch > 1) {
              const unsigned item = idx / (unsigned)(a.M + 1);
              r = idx - item * (unsigned)(a.M + 1);
              key += item * (unsigned)a.K;
            }
            e[c].d = (key >> a.shift) & mask;
            const unsigned keep = a.drop_after_first ? (key >> a.bits) : key;
            e[c].b = LAST ? key : r;
            e[c].a = LAST ? r : (PACKED ? ((keep << a.row_bits) | r) : keep);
          }
        }
      }
      TR_STAMP(1);

      // rank inside the wave: lanes holding the same digit ("peers") found with one ballot per digit
      // bit; a lane's rank is the wave's running count of the digit + the number of peers below it
#pragma unroll
      for (int c = 0; c < TR_PER; ++c) {
        const int i = wave * (TR_TILE / TR_WAVES) + c * 64 + lane;
        const bool valid = i < cur_len;
        const unsigned d = valid ? e[c].d : 0u;
        unsigned long long peers = __ballot(valid);
        for (int b = 0; b < a.bits; ++b) {
          const bool bit = (d >> b) & 1u;
          const unsigned long long m = __ballot(bit);
          peers &= bit ? m : ~m;
        }
        const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(peers >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)peers, 0));
        int base = 0;
        if (valid) base = mycnt[d];
        if (valid && below == 0) mycnt[d] = (unsigned short)(base + __builtin_popcountll(peers));
        e[c].d = d | ((unsigned)(base + below) << 16);
      }
      TR_STAMP(2);
      __syncthreads();
      TR_STAMP(3);

      // per digit: exclusive prefix over the waves (in place), tile total; then the exclusive scan of
      // the totals over the digits and the digit's global offset
#pragma unroll
      for (int g = 0; g < GPT; ++g) {
        if (g * TR_THREADS < nb) {  // block-uniform
          const int d = g * TR_THREADS + tid;
          int tot = 0;
          if (d < nb) {
            int v[TR_WAVES];
#pragma unroll
            for (int w = 0; w < TR_WAVES; ++w) v[w] = cntw[w * nb + d];
#pragma unroll
            for (int w = 0; w < TR_WAVES; ++w) {
              cntw[w * nb + d] = (unsigned short)tot;
              tot += v[w];
            }
          }
          int incl = tot;
#pragma unroll
          for (int s2 = 1; s2 < 64; s2 <<= 1) {
            const int v = __shfl_up(incl, s2, 64);
            if (lane >= s2) incl += v;
          }
          if (lane == 63) wsum[wave] = incl;
          __syncthreads();
          int wbase = 0, all = 0;
#pragma unroll
          for (int w = 0; w < 16; ++w) {
            const int v = wsum[w];
            if (w < wave) wbase += v;
            all += v;
          }
          if (d < nb) {
            const int bs = carry_s + wbase + incl - tot;
            binstart[d] = bs;
            gadj[d] = goff[g] - bs;
          }
          __syncthreads();
          if (tid == 0) carry_s += all;
          __syncthreads();
        }
      }

      TR_STAMP(4);
      {
        // straight from registers (wide keys / rows, 11-bit digits): correct for every size, but the
        // stores of a wave go to up to 64 different places
#pragma unroll
        for (int c = 0; c < TR_PER; ++c) {
          const int i = wave * (TR_TILE / TR_WAVES) + c * 64 + lane;
          if (i < cur_len) {
            const unsigned d = e[c].d & 0xffffu;
            const long dst = (long)gadj[d] + binstart[d] + cntw[wave * nb + d] + (long)(e[c].d >> 16);
            if (LAST) {
              a.t_col[dst] = (int)e[c].a;
              a.t_val[dst] = e[c].val;
              if (a.keys_out) a.keys_out[dst] = e[c].b;
            } else if (PACKED) {
              a.out_packed[dst] = make_uint2(e[c].a, __builtin_bit_cast(unsigned, e[c].val));
            } else {
              a.out_key[dst] = e[c].a;
              a.out_row[dst] = e[c].b;
              a.out_val[dst] = e[c].val;
            }
          }
        }
        __syncthreads();  // binstart / gadj / cntw are rewritten by the next tile
        if (t + t_step < t_end) fetch(t + t_step); else len = 0;
      }
    } else {
      if (t + t_step < t_end) fetch(t + t_step); else len = 0;
    }
  }
  TR_STAMP_FLUSH(FIRST);
}

// ---------------------------------------------------------------------------------------------
// Staged scatter (8-byte entries, digits of ≤ 10 bits): the main path.  The tile is ranked, then
// reordered by digit in LDS so that every digit's entries leave as one contiguous run.  One workgroup
// fills a CU's LDS, so there is no second workgroup to hide memory time behind: the kernel is
// software-pipelined over the tiles of a persistent workgroup instead.  While tile t is processed,
//     * tile t−1 streams out of LDS slice by slice — two slices at the top, some between the ranking
//       steps, some in the prefix phase (`sorted`, `sorted_d` are rewritten only when tile t is staged;
//       the per-digit offsets `gadj` are double-buffered), and
//     * tile t+1 is loaded slice by slice between the ranking steps, into the raw registers tile t gave
//       up when it was converted (first pass: its columns; the values of tile t itself are loaded one
//       phase ahead, they are not needed before the tile is staged),
// so loads, stores and ranking share the CU.  What it is up against (tools/probes/store_piece_probe.cpp,
// tr_probe.cpp): the runs a tile leaves in are 32–64 bytes at arbitrary offsets, and an XCD's L2 accepts
// such pieces at ≈5 per clock for its 32 CUs — ≈0.3 ms (first pass) and ≈0.46 ms (last pass: two 4-byte
// arrays) for the C3 matrix whatever the kernel does; a wave whose store finds the queue full stalls
// with its ranking work, so only the other waves' work overlaps.  Issued as blocks (all loads, then
// all stores) a tile took 35 k cycles, 47 k before any pipelining, ≈26 k now.
// The compiler's waits: vmcnt retires in order on gfx9 and a wait is computed from the number of younger
// memory operations on EVERY path, so all loads and stores here are unconditional — indices are clamped
// into the tile (lanes beyond a short tile's end repeat its last entry: same address, same bytes; a
// workgroup's first tile "streams out" to a dump slot) — block-uniform metadata comes through the
// scalar unit, and the first tile's loads are waited for before the loop.  One predicated store, or a
// vector load of a tile descriptor, turned every wait into vmcnt(0): load latency, store drain and
// ranking one after the other.
// ---------------------------------------------------------------------------------------------
#ifndef MI_TR_INTERLEAVE
#define MI_TR_INTERLEAVE 1
#endif
// Where slice k of the previous tile streams out: 0 = top of the tile, 1+c = after ranking step c,
// 9 / 10 = prefix phase before its first / second barrier.  (C3 shape, whole transpose, one box: all
// eight as one block before the ranking 2.16 ms, one per ranking step 2.04 ms, spread over the three
// phases 1.93–1.97 ms with ≈2 % between the spreads tried; boxes differ by more than that.)
#ifndef MI_TR_SLICE_PLAN
#define MI_TR_SLICE_PLAN {0, 0, 3, 7, 9, 9, 10, 10}
#endif
#ifndef MI_TR_LB_SKEW
#define MI_TR_LB_SKEW 0
#endif
#ifndef MI_TR_LB_EARLY  // 1: a tile's look-back loads are issued when it has been staged, 0: at the top of the next tile
#define MI_TR_LB_EARLY 1
#endif
#ifndef MI_TR_ABL  // timing-only builds (tools/probes/tr_time.cpp): 1 = no global stores, 3 = no tile loads either
#define MI_TR_ABL 0
#endif
// LB = true: the one-sweep plan.  There is no scanned table; a tile's per-digit global offsets are found by decoupled
// look-back inside its tile group (TR_LB_GROUPS contiguous groups per pass, each with a per-digit base from
// tr_lb_setup_kernel).  Every digit of every tile has one 32-bit status word {flag, value} written with ONE agent-scope
// (sc1) store and read with agent-scope loads — a self-contained granule, no ordering with any other data is needed:
//   * in its prefix phase a tile publishes AGG | (its count of the digit);
//   * at the top of the workgroup's NEXT tile — the staged tile only streams out from then on — thread d reads the words
//     of the TR_LB_WALK tiles before it in one go (issued before the row fill, consumed after it), adds counts until it
//     meets a PREFIX word or the group's first tile, re-polls a word it found empty, and publishes
//     PREFIX | (count of the digit in the group up to and including this tile) for the tiles behind it;
//   * tiles are handed out by per-group tickets (one returning atomic per tile, by thread 0, a tile ahead like the
//     loads): whoever holds a ticket is running, tickets ascend, and a workgroup publishes a tile's AGG word before it
//     waits for anything that tile's successors could hold — so every wait is on a lower-numbered tile of a RESIDENT
//     workgroup and the lowest unfinished tile can always finish: no dependence on the grid being co-resident or on
//     dispatch order.  A workgroup whose group is used up helps the next group.  A poll that does not succeed within
//     TR_LB_SPIN_LIMIT tries sets *errflag and goes on with a wrong offset rather than hang the GPU.
//   * last pass: the first tile of a bin also records its offsets as the row offsets of Aᵀ (rowoff[bin][digit]).
template <bool FIRST, bool LAST, bool LB>
__global__ __launch_bounds__(TR_THREADS) void tr_scatter_staged_kernel(TrArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nb = 1 << a.bits;
  unsigned short* cntw = reinterpret_cast<unsigned short*>(smem);  // [TR_WAVES][nb]
  int* binstart = reinterpret_cast<int*>(smem + (size_t)TR_WAVES * nb * 2);  // [nb] tile-local start of a digit
  int* gadj2 = binstart + nb;  // [2][nb] global offset − binstart, of this tile and of the one streaming out
  uint2* sorted = reinterpret_cast<uint2*>(gadj2 + 2 * nb);                   // [TR_TILE]
  unsigned short* sorted_d = reinterpret_cast<unsigned short*>(sorted + TR_TILE);  // [TR_TILE]
  unsigned* rowid = reinterpret_cast<unsigned*>(sorted_d + TR_TILE);               // [TR_TILE] (FIRST)
  int* lb_tot = reinterpret_cast<int*>(sorted_d + TR_TILE);  // [nb] (LB, a last pass: no rowid) the staged tile's counts
  __shared__ int wsum[16];
  __shared__ int s_next[4];
  __shared__ int s_grp;  // (LB) the tile group thread 0 draws tickets from

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // in a scalar register: loops over "my rows" stay scalar
  const unsigned mask = (unsigned)nb - 1u;
  const unsigned rmask = (1u << a.row_bits) - 1u;
  unsigned short* mycnt = cntw + wave * nb;
  const int last_slot = a.batch * (a.M + 1) - 1;

  // the tile in flight: raw words (first pass: column, value bits; later: the packed entry) + its metadata
  unsigned rx[TR_PER], ry[TR_PER];
  int n_goff = 0, n_len = 0, n_row_lo = 0, n_row_hi = 0, n_rp0 = 0, n_rp1 = 0;
  long n_start = 0;
  auto fetch_meta = [&](int t) {
    if (a.desc) {
      const int2 d = sload(a.desc + t);
      n_start = d.x;
      n_len = d.y;
    } else {
      n_start = (long)t * TR_TILE;
      const long rest = a.nnz - n_start;
      n_len = rest < TR_TILE ? (int)rest : TR_TILE;
    }
    if (FIRST) {
      // rows come from LDS later (rowid); lane l of wave w prefetches the bounds of row row_lo + 16·l + w
      n_row_lo = sload(a.tile_row + t);
      n_row_hi = sload(a.tile_row + t + 1);
      int r = n_row_lo + lane * TR_WAVES + wave;
      r = r < n_row_hi ? r : n_row_hi;
      n_rp0 = a.rowptr[r];
      n_rp1 = a.rowptr[r < last_slot ? r + 1 : last_slot];
    }
    if (!LB) {
      n_goff = a.table[(long)t * nb + (tid & (nb - 1))];
    }
  };
  // slice c of the NEXT tile's keys (first pass: columns; later: the whole packed entry) …
  auto fetch_slice = [&](int c) {
    int i = wave * (TR_TILE / TR_WAVES) + c * 64 + lane;
    i = i < n_len ? i : 0;
    if (MI_TR_ABL & 2) {
      rx[c] = (unsigned)(i * 2654435761u) >> 7;
      ry[c] = 0;
      return;
    }
    if (FIRST) {
      rx[c] = (unsigned)a.col[n_start + i];
    } else {
      const uint2 w = a.in_packed[n_start + i];
      rx[c] = w.x;
      ry[c] = w.y;
    }
  };
  // … and, first pass only, slice c of the CURRENT tile's values: they are not needed before the tile is
  // staged, so they are loaded one phase (not one tile) ahead and cost no second set of registers
  auto fetch_values = [&](int c, long start, int len) {
    int i = wave * (TR_TILE / TR_WAVES) + c * 64 + lane;
    i = i < len ? i : 0;
    if (MI_TR_ABL & 2) return;
    ry[c] = __builtin_bit_cast(unsigned, a.val[start + i]);
  };
  // slice k of the tile staged in LDS (length plen) leaves; with no such tile (plen = 0) to the dump slot
  int plen = 0;
  const int* gadj_out = gadj2;  // offsets of the tile that is streaming out
  constexpr int slice_at[TR_PER] = MI_TR_SLICE_PLAN;
  auto store_slice = [&](int k) {
    const int i0 = k * TR_THREADS + tid;
    const int i = i0 < plen ? i0 : (plen > 0 ? plen - 1 : 0);
    const uint2 w = sorted[i];
    const long dst = (long)gadj_out[sorted_d[i] & mask] + i;
    if (MI_TR_ABL & 1) {
      asm volatile("" ::"v"(w.x), "v"(w.y), "v"(dst));
      return;
    }
    if (LAST) {
      int* pc = plen > 0 ? a.t_col + dst : reinterpret_cast<int*>(a.dump) + 4 * blockIdx.x;
      float* pv = plen > 0 ? a.t_val + dst : reinterpret_cast<float*>(a.dump) + 4 * blockIdx.x + 1;
      *pc = (int)w.x;
      *pv = __builtin_bit_cast(float, w.y);
    } else {
      uint2* po = plen > 0 ? a.out_packed + dst : a.dump + 2 * blockIdx.x;
      *po = w;  // plain stores: the runs of neighbouring tiles merge in the XCD's L2
    }           // (non-temporal stores measured 3.9 ms vs 2.6 ms for the whole transpose)
  };

  // ---- one-sweep plan: tickets and look-back ----
  static_assert(!(LB && FIRST), "the look-back form keeps its counts where the first pass keeps its row ids");
  if (LB && tid == 0) s_grp = (int)((blockIdx.x & 7u) * (TR_LB_GROUPS / 8) + ((blockIdx.x >> 3) & (TR_LB_GROUPS / 8 - 1)));
  // thread 0: a ticket of this workgroup's group is ISSUED at the top of a tile (one returning atomic) and turned into a
  // tile at the tile's end, a whole tile later — two tiles ahead of its use, so the atomic's round trip (≈1–3 µs from a
  // streaming CU) is never waited for.  A group that is used up sends the workgroup on to the next group that still
  // has tiles (then with a wait, rarely); −1 when none is left.
  int aq_tk = 0;
  auto acquire_issue = [&]() {
    const int my_grp = __builtin_amdgcn_readfirstlane(s_grp);
    const int gf = sload(a.grp_first + my_grp), gl = sload(a.grp_first + my_grp + 1);
    aq_tk = gl > gf ? atomicAdd(&a.tickets[my_grp], 1) : 0x7fffffff;
  };
  auto acquire_finish = [&](int slot) {
    int tile = -1, grp = 0;
    int my_grp = __builtin_amdgcn_readfirstlane(s_grp);
    const int aq_gf = sload(a.grp_first + my_grp), aq_gl = sload(a.grp_first + my_grp + 1);
    if (aq_tk < aq_gl - aq_gf) {
      tile = aq_gf + aq_tk;
      grp = my_grp;
    } else {
      for (int tries = 1; tries < TR_LB_GROUPS; ++tries) {
        my_grp = my_grp + 1 < TR_LB_GROUPS ? my_grp + 1 : 0;
        const int gf = a.grp_first[my_grp], gl = a.grp_first[my_grp + 1];
        if (gl > gf) {
          const int k = atomicAdd(&a.tickets[my_grp], 1);
          if (k < gl - gf) {
            tile = gf + k;
            grp = my_grp;
            break;
          }
        }
      }
    }
    s_grp = my_grp;
    s_next[slot] = tile;
    s_next[slot + 1] = grp;
  };
  // (threads beyond nb shadow digit 0 and publish to a dump word)
  unsigned lbv[TR_LB_WALK];
  int lb_base = 0;
  int p_tile = 0, p_grp = 0;  // the tile staged in LDS: id, its group (its per-digit counts: lb_tot)
  // read the status words of the TR_LB_WALK tiles before tile p (clamped into its group) and the group's base
  auto lookback_issue = [&]() {
    const int dd = tid < nb ? tid : 0;
    const int p_gf = sload(a.grp_first + p_grp);
#pragma unroll
    for (int j = 0; j < TR_LB_WALK; ++j) {
      int k = p_tile - 1 - j;
      k = k < p_gf ? p_gf : k;
      // (32-bit element offsets from ONE base pointer: the status area is far below 2³² words)
      lbv[j] = __hip_atomic_load(a.status + ((unsigned)k * (unsigned)nb + (unsigned)dd), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    lb_base = a.grp_base[(unsigned)p_grp * (unsigned)nb + (unsigned)dd];
  };
  // finish tile p's look-back, publish its inclusive prefix, set its offsets (gadj) and, last pass, the row offsets
  auto lookback_resolve = [&](int* gadj) {
    const int dd = tid < nb ? tid : 0;
    const int p_gf = sload(a.grp_first + p_grp);
    unsigned sum = 0;
    int k = p_tile - 1;
    bool done = k < p_gf, stalled = false;
#pragma unroll
    for (int j = 0; j < TR_LB_WALK; ++j) {
      if (!done && !stalled) {
        const unsigned w = lbv[j], f = w >> 30;
        if (f == 0) {
          stalled = true;  // not published yet: poll it below
        } else {
          sum += w & TR_LB_VALUE;
          --k;
          done = f == 2 || k < p_gf;
        }
      }
    }
    int spins = 0;
    while (__any(!done)) {  // wave-uniform; in step with its neighbours a tile does not come here
      if (!done) {
        const unsigned w = __hip_atomic_load(a.status + ((unsigned)k * (unsigned)nb + (unsigned)dd), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned f = w >> 30;
        if (f != 0) {
          sum += w & TR_LB_VALUE;
          --k;
          done = f == 2 || k < p_gf;
        }
      }
      if (++spins > TR_LB_SPIN_LIMIT) {
        if (!done) *a.errflag = 1;
        break;
      }
      if (spins > 4) __builtin_amdgcn_s_sleep(8);
    }
    int excl = lb_base + (int)sum;
    const int p_tot = lb_tot[dd];
    // a run must lie inside the output whatever went wrong upstream (a store outside it would fault the GPU)
    if ((unsigned)excl > (unsigned)((int)a.nnz - p_tot)) {
      excl = 0;
      *a.errflag = 2;
    }
    unsigned* pub = tid < nb ? a.status + ((unsigned)p_tile * (unsigned)nb + (unsigned)tid) : a.lb_dump + tid;
    // (relative to the group's first tile, like the counts: a reader adds its group's base itself)
    __hip_atomic_store(pub, TR_LB_PREFIX | (sum + (unsigned)p_tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < nb) {
      gadj[tid] = excl - binstart[tid];
      if (LAST) {
        const int p_bin = sload(a.tile_bin + p_tile);
        if (sload(a.first_tile + p_bin) == p_tile) a.rowoff[((long)p_bin << a.bits) + tid] = excl;
      }
    }
  };

  // XCD-aware tile order (see tr_scatter_kernel): every XCD walks a contiguous range of the tiles in use
  const int ntiles = a.used_tiles ? sload(a.used_tiles) : a.ntiles;
  int t = blockIdx.x, t_end = ntiles, t_step = gridDim.x;
  int cur_grp = 0, n1_tile = -1, n1_grp = 0;  // one-sweep last pass: the group of the tile in hand; the tile after it
  if (LB) {
    // (MI_TR_LB_SKEW: the workgroups of a group start a fraction of a tile apart, ≈1.4 µs each, so that a tile's
    // predecessors published their counts well before it looks back; measured: no gain — off.)
#if MI_TR_LB_SKEW
    for (int i = 0; i < (int)(blockIdx.x >> 5); ++i) __builtin_amdgcn_s_sleep(52);
#endif
    if (tid == 0) {
      acquire_issue();
      acquire_finish(0);
      if (s_next[0] >= 0) {
        acquire_issue();
        acquire_finish(2);
      } else {
        s_next[2] = -1;
        s_next[3] = 0;
      }
    }
    __syncthreads();
    t = __builtin_amdgcn_readfirstlane(s_next[0]);
    cur_grp = __builtin_amdgcn_readfirstlane(s_next[1]);
    n1_tile = __builtin_amdgcn_readfirstlane(s_next[2]);
    n1_grp = __builtin_amdgcn_readfirstlane(s_next[3]);
    t_end = 0x7fffffff;
    t_step = 0;
    if (t < 0) return;
    __syncthreads();  // s_next is rewritten by the next acquire
  } else if (gridDim.x == TR_GRID) {
    const int per = (ntiles + 7) / 8, xcd = blockIdx.x & 7;
    t = xcd * per + (blockIdx.x >> 3);
    t_end = (xcd + 1) * per < ntiles ? (xcd + 1) * per : ntiles;
    t_step = TR_GRID / 8;
  }
  TR_STAMP_INIT;
  if (t < t_end) {
    fetch_meta(t);
#pragma unroll
    for (int c = 0; c < TR_PER; ++c) fetch_slice(c);
    // the first tile's loads are complete before the loop is entered (an empty asm that "uses" them): the
    // loop's waits must see the same pending state from here as from its own back edge
#pragma unroll
    for (int c = 0; c < TR_PER; ++c) {
      if (FIRST) asm volatile("" : "+v"(rx[c]));
      else asm volatile("" : "+v"(rx[c]), "+v"(ry[c]));
    }
    asm volatile("" : "+v"(n_goff), "+v"(n_rp0), "+v"(n_rp1));
  }
  static_assert(TR_PER == 8, "MI_TR_SLICE_PLAN places eight slices");
  static_assert(TR_GROUPS % 16 == 0, "tr_group_bases_kernel: sixteen waves share the groups");
  int tile_no = 0;
  for (; t < t_end && t >= 0;) {
    const int cur_len = n_len;  // block-uniform, ≥ 1: every tile below `ntiles` is in use
    const long start = n_start;
    const int row_lo = n_row_lo, row_hi = n_row_hi;
    TR_STAMP(0);
    int* gadj = gadj2 + ((tile_no & 1) ? nb : 0);
    if (LB) {
      // (the staged tile's predecessors' words were requested when it was staged); the ticket after next
#if !MI_TR_LB_EARLY
      if (plen > 0) lookback_issue();
#endif
      if (tid == 0 && n1_tile >= 0) acquire_issue();
    } else {
#if MI_TR_INTERLEAVE
#pragma unroll
      for (int k = 0; k < TR_PER; ++k)
        if (slice_at[k] == 0) store_slice(k);
#endif
    }
    // every wave zeroes ITS counters (nobody else touches them between the prefix phase and here)
    for (int i = lane; i < nb / 2; i += 64) reinterpret_cast<unsigned*>(mycnt)[i] = 0u;
    if (FIRST) {
      // every row of the tile writes its flat index over its own entries (wave-uniform bounds, 64
      // positions per instruction); each position of the tile belongs to exactly one row
      const int rp0 = n_rp0, rp1 = n_rp1;
      const int t_lo = (int)start;  // nnz < 2³¹: entry offsets fit an int
      auto fill = [&](int r, int b, int en) {
        b = (b < t_lo ? t_lo : b) - t_lo;
        en = (en - t_lo > cur_len ? cur_len : en - t_lo);
        for (int q = b + lane; q < en; q += 64) rowid[q] = (unsigned)r;
      };
      const int span = row_hi - row_lo + 1;
      for (int k = 0; k < 64 && k * TR_WAVES + wave < span; ++k)
        fill(row_lo + k * TR_WAVES + wave, __builtin_amdgcn_readlane(rp0, k), __builtin_amdgcn_readlane(rp1, k));
      for (int r = row_lo + 64 * TR_WAVES + wave; r <= row_hi; r += TR_WAVES)  // tiles spanning > 1024 rows
        fill(r, sload(a.rowptr + r), sload(a.rowptr + (r < last_slot ? r + 1 : last_slot)));
    }
    if (FIRST) __syncthreads();  // rows known
    const int next_tile = n1_tile, next_grp = n1_grp;
    if (LB && plen > 0) lookback_resolve(const_cast<int*>(gadj_out));  // the staged tile's offsets
    TR_STAMP(7);
    Slot e[TR_PER];
#pragma unroll
    for (int c = 0; c < TR_PER; ++c) {
      const int i = wave * (TR_TILE / TR_WAVES) + c * 64 + lane;
      e[c].a = 0u;
      e[c].d = 0u;
      if (!FIRST) e[c].val = __builtin_bit_cast(float, ry[c]);
      if (i < cur_len) {
        unsigned key, r;
        if (FIRST) {
          const unsigned idx = rowid[i];
          key = rx[c];
          r = idx;
          if (a.batch > 1) {
            const unsigned item = idx / (unsigned)(a.M + 1);
            r = idx - item * (unsigned)(a.M + 1);
            key += item * (unsigned)a.K;
          }
        } else {
          key = rx[c] >> a.row_bits;
          r = rx[c] & rmask;
        }
        e[c].d = (key >> a.shift) & mask;
        const unsigned keep = a.drop_after_first ? (key >> a.bits) : key;  // two passes: the low digit is dropped
        e[c].a = LAST ? r : ((keep << a.row_bits) | r);
      }
    }
    const int goff = n_goff;
    const bool more = LB ? next_tile >= 0 : t + t_step < t_end;  // block-uniform
    if (more) fetch_meta(LB ? next_tile : t + t_step);
    TR_STAMP(1);
    if (LB) {
      __syncthreads();  // the staged tile's offsets (gadj_out) are complete
#if MI_TR_INTERLEAVE
#pragma unroll
      for (int k = 0; k < TR_PER; ++k)
        if (slice_at[k] == 0) store_slice(k);
#endif
    }

    // Rank inside the wave.  A lane's rank among equal digits = the wave's running count of the digit
    // (wave-private 16-bit LDS word: count in bits 0-9, ≤ 512) + the number of "peers" (lanes of this
    // step with the same digit) below it.  With 1024 digits and 64 lanes most lanes have no peer, so
    // the peers are found by collision instead of one ballot per digit bit: every lane tags its digit's
    // word with its lane id (bits 10-15) and reads it back; a lane that reads another id has a peer.
    // One ballot per COLLIDING digit value (≈2 per step for uniform digits) hands all its holders
    // their peer mask.  Stable (ranks follow lane order), no atomics.
    // Between the steps: one slice of the previous tile out, one slice of the next tile in.
#if !MI_TR_INTERLEAVE
    if (more) {
#pragma unroll
      for (int c = 0; c < TR_PER; ++c) fetch_slice(c);
    }
    if (FIRST) {
#pragma unroll
      for (int c = 0; c < TR_PER; ++c) fetch_values(c, start, cur_len);
    }
#pragma unroll
    for (int c = 0; c < TR_PER; ++c) store_slice(c);
#endif
#pragma unroll
    for (int c = 0; c < TR_PER; ++c) {
      const int i = wave * (TR_TILE / TR_WAVES) + c * 64 + lane;
      const bool valid = i < cur_len;
      const unsigned d = e[c].d;
      unsigned w0 = 0;
      bool lost = false;
      if (valid) {
        w0 = mycnt[d] & 0x3ffu;
        mycnt[d] = (unsigned short)(w0 | ((unsigned)lane << 10));
        // the read-back must really go to LDS (to this thread alone it would be its own store)
        asm volatile("" ::: "memory");
        lost = (unsigned)(mycnt[d] >> 10) != (unsigned)lane;
      }
      unsigned long long coll = __ballot(lost);
      unsigned long long peers = 0;  // the mask of a lane's peers, itself included; 0 = no peer
      while (coll) {  // wave-uniform
        const int holder = __builtin_ctzll(coll);
        const unsigned dv = (unsigned)__builtin_amdgcn_readlane((int)d, holder);
        const bool mine = valid && d == dv;
        const unsigned long long m = __ballot(mine);
        if (mine) peers = m;
        coll &= ~m;
      }
      const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(peers >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)peers, 0));
      const int holders = __builtin_popcountll(peers);
      if (valid && below == 0) mycnt[d] = (unsigned short)(w0 + (holders > 1 ? holders : 1));
      e[c].d = d | ((w0 + (unsigned)below) << 16);
#if MI_TR_INTERLEAVE
#pragma unroll
      for (int k = 0; k < TR_PER; ++k)
        if (slice_at[k] == 1 + c) store_slice(k);
      if (more) fetch_slice(c);
      if (FIRST) fetch_values(c, start, cur_len);
#endif
    }
    TR_STAMP(2);
    __syncthreads();  // all counts in; every thread is past its reads of the previous tile's staging
    TR_STAMP(3);

    // per digit: exclusive prefix over the waves (in place), tile total; then the exclusive scan of
    // the totals over the digits and the digit's global offset
    {
      const int d = tid;
      int tot = 0;
      if (d < nb) {
        int v[TR_WAVES];
#pragma unroll
        for (int w = 0; w < TR_WAVES; ++w) v[w] = cntw[w * nb + d];
#pragma unroll
        for (int w = 0; w < TR_WAVES; ++w) {
          cntw[w * nb + d] = (unsigned short)tot;
          tot += v[w];
        }
      }
      if (LB) {
        // this tile's count of the digit, for the tiles behind it (one self-contained word, agent scope)
        unsigned* pub = d < nb ? a.status + ((unsigned)t * (unsigned)nb + (unsigned)d) : a.lb_dump + tid;
        __hip_atomic_store(pub, TR_LB_AGG | (unsigned)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d < nb) lb_tot[d] = tot;  // read again when this tile's look-back is resolved (top of the next tile)
      }
      int incl = tot;
#pragma unroll
      for (int s2 = 1; s2 < 64; s2 <<= 1) {
        const int v = __shfl_up(incl, s2, 64);
        if (lane >= s2) incl += v;
      }
      if (lane == 63) wsum[wave] = incl;
#if MI_TR_INTERLEAVE
#pragma unroll
      for (int k = 0; k < TR_PER; ++k)
        if (slice_at[k] == 9) store_slice(k);
#endif
      __syncthreads();
      int wbase = 0;
#pragma unroll
      for (int w = 0; w < 16; ++w) {
        const int v = wsum[w];
        if (w < wave) wbase += v;
      }
      if (d < nb) {
        const int bs = wbase + incl - tot;
        binstart[d] = bs;
        if (!LB) gadj[d] = goff - bs;
      }
#if MI_TR_INTERLEAVE
#pragma unroll
      for (int k = 0; k < TR_PER; ++k)
        if (slice_at[k] == 10) store_slice(k);
#endif
      __syncthreads();
    }

    TR_STAMP(4);
    // reorder the tile by digit in LDS; it streams out during the next tile's ranking
#pragma unroll
    for (int c = 0; c < TR_PER; ++c) {
      const int i = wave * (TR_TILE / TR_WAVES) + c * 64 + lane;
      if (i < cur_len) {
        const unsigned d = e[c].d & 0xffffu;
        const int s2 = binstart[d] + cntw[wave * nb + d] + (int)(e[c].d >> 16);
        sorted[s2] = make_uint2(e[c].a, FIRST ? ry[c] : __builtin_bit_cast(unsigned, e[c].val));
        sorted_d[s2] = (unsigned short)d;
      }
    }
    plen = cur_len;
    gadj_out = gadj;
    if (LB) {
      p_tile = t;
      p_grp = cur_grp;
#if MI_TR_LB_EARLY
      lookback_issue();  // in flight across the barrier below and the top of the next tile
#endif
      cur_grp = next_grp;
      t = next_tile;
      if (tid == 0) {
        if (next_tile >= 0) {
          acquire_finish(0);  // the ticket issued at the top of this tile
        } else {
          s_next[0] = -1;
          s_next[1] = 0;
        }
      }
    } else {
      t += t_step;
    }
    ++tile_no;
    TR_STAMP(5);
    __syncthreads();  // the tile is staged (and, first pass: rowid may be refilled)
    if (LB) {
      n1_tile = __builtin_amdgcn_readfirstlane(s_next[0]);
      n1_grp = __builtin_amdgcn_readfirstlane(s_next[1]);
    }
    TR_STAMP(6);
  }
  // the last tile of this workgroup
  if (plen > 0) {
    if (LB) {
#if !MI_TR_LB_EARLY
      lookback_issue();
#endif
      lookback_resolve(const_cast<int*>(gadj_out));
      __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < TR_PER; ++k) store_slice(k);
  }
  TR_STAMP_FLUSH(FIRST);
}

// Row offsets of Aᵀ, [batch][K+1] with global offsets, from the scanned tables (≤ 2 passes).
//   one pass : offset(key) = table0[0][key]
//   two      : offset(key) = table1[first_tile[lo]][hi]
__global__ __launch_bounds__(256) void tr_rowptr_kernel(const int* __restrict__ table0, const int* __restrict__ table1,
                                                        const int* __restrict__ first_tile, int bits0, int bits1,
                                                        int passes, int batch, int K, long nnz,
                                                        int* __restrict__ t_rowptr) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)batch * ((long)K + 1);
  if (idx >= total) return;
  const long b = idx / ((long)K + 1);
  const long key = b * K + (idx - b * ((long)K + 1));
  int v;
  if (key >= (long)batch * K) {
    v = (int)nnz;
  } else if (passes == 1) {
    v = table0[key];
  } else {
    const int lo = (int)(key & ((1L << bits0) - 1)), hi = (int)(key >> bits0);
    v = table1[(long)first_tile[lo] * (1L << bits1) + hi];
  }
  t_rowptr[idx] = v;
}

// Three passes: row offsets from the run boundaries of the sorted full keys.
__global__ __launch_bounds__(256) void tr_boundaries_kernel(const unsigned* __restrict__ keys, long nnz, int batch,
                                                            int K, int* __restrict__ t_rowptr) {
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  if (q >= nnz) return;
  const long key = keys[q];
  const long prev = q > 0 ? (long)keys[q - 1] : -1;
  // every key in (prev, key] starts at q; a key's slot is key + (its item) in the [batch][K+1] layout
  for (long k = prev + 1; k <= key; ++k) {
    const long b = k / K;
    t_rowptr[k + b] = (int)q;
    if (k - b * K == 0 && b > 0) t_rowptr[k + b - 1] = (int)q;  // the previous item's end slot
  }
  if (q == nnz - 1) {
    const long allk = (long)batch * K;
    for (long k = key + 1; k <= allk; ++k) {
      if (k < allk) {
        const long b = k / K;
        t_rowptr[k + b] = (int)nnz;
        if (k - b * K == 0 && b > 0) t_rowptr[k + b - 1] = (int)nnz;
      } else {
        t_rowptr[allk + batch - 1] = (int)nnz;
      }
    }
  }
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct TrPlan {
  int passes, key_bits, row_bits;
  int bits[3], shift[3];
  bool packed, staged[3], drop;
  long ntiles0, ntiles_max[3];
};

TrPlan make_plan(int batch, int M, int K, long nnz) {
  TrPlan p{};
  p.key_bits = bits_for((unsigned long long)batch * (unsigned long long)(K > 0 ? K : 1));
  p.row_bits = bits_for((unsigned long long)(M > 0 ? M : 1));
  if (p.key_bits <= 10) {
    p.passes = 1;
    p.bits[0] = p.key_bits;
  } else if (p.key_bits <= 20) {
    p.passes = 2;
    p.bits[0] = (p.key_bits + 1) / 2;
    p.bits[1] = p.key_bits - p.bits[0];
  } else {
    p.passes = 3;
    p.bits[0] = (p.key_bits + 2) / 3;
    p.bits[1] = (p.key_bits - p.bits[0] + 1) / 2;
    p.bits[2] = p.key_bits - p.bits[0] - p.bits[1];
  }
  p.drop = p.passes == 2;  // the key field loses its low digit after the first of two passes
  const int carried_bits = p.passes == 2 ? p.key_bits - p.bits[0] : p.key_bits;  // key bits an intermediate entry keeps
  p.packed = p.passes == 1 || carried_bits + p.row_bits <= 32;
  int sh = 0;
  for (int i = 0; i < p.passes; ++i) {
    p.shift[i] = (p.drop && i == 1) ? 0 : sh;
    sh += p.bits[i];
    // LDS staging needs the 8-byte entry and a 10-bit digit; the last of three passes must also hand
    // out full keys, which the staged form does not carry
    p.staged[i] = p.packed && p.bits[i] <= 10 && !(p.passes == 3 && i == 2);
  }
  p.ntiles0 = (nnz + TR_TILE - 1) / TR_TILE;
  for (int i = 0; i < 3; ++i) p.ntiles_max[i] = p.ntiles0;
  if (p.passes == 2) p.ntiles_max[1] = p.ntiles0 + (1L << p.bits[0]);
  return p;
}

struct TrWs {
  size_t inter[2], tables[3], lbz, lbz_bytes, gsum, tile_row, desc, first_tile, keys, dump, total;
  size_t h2g, tickets, errflag;                          // inside the zeroed region lbz
  size_t hibase, grp_first, tile_bin, rowoff, lb_dump;   // one-sweep plan, not zeroed
};

// Which plan transposes a matrix (process-wide; same bits): AUTO takes the one-sweep plan where it applies.
std::atomic<int> g_tr_plan{MI_TRANSPOSE_PLAN_AUTO};

// The one-sweep plan covers: one matrix, two packed + staged passes, 30-bit offsets, enough tiles to fill the chip.
bool one_sweep_applies(const TrPlan& p, int batch, long nnz) {
  return batch == 1 && p.passes == 2 && p.packed && p.staged[0] && p.staged[1] && p.bits[0] >= 5 && nnz < (1L << 30) &&
         p.ntiles0 >= 2 * TR_GRID;
}

TrWs ws_layout(const TrPlan& p, long nnz) {
  TrWs w{};
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off += align256(bytes);
    return o;
  };
  const size_t entry = p.packed ? 8 : 12;
  w.inter[0] = take(p.passes >= 2 ? (size_t)nnz * entry : 0);
  w.inter[1] = take(p.passes >= 3 ? (size_t)nnz * entry : 0);
  for (int i = 0; i < 3; ++i)
    w.tables[i] = take(i < p.passes ? (size_t)(p.ntiles_max[i] + 1) * ((size_t)1 << p.bits[i]) * 4 : 0);
  // one-sweep plan: histograms, tickets and the error flag follow the tables (= its status words): ONE memset zeroes all
  const bool lb = p.passes == 2;
  w.lbz = off;
  w.h2g = take(lb ? (size_t)TR_LB_GROUPS * ((size_t)1 << p.bits[1]) * 4 : 0);
  w.tickets = take(lb ? (size_t)TR_LB_GROUPS * 4 : 0);
  w.errflag = take(lb ? 4 : 0);
  w.lbz_bytes = off - w.lbz;
  w.hibase = take(lb ? (((size_t)1 << p.bits[1]) + 1) * 4 : 0);
  w.grp_first = take(lb ? (size_t)(TR_LB_GROUPS + 1) * 4 : 0);
  w.tile_bin = take(lb ? (size_t)p.ntiles_max[1] * 4 : 0);
  w.rowoff = take(lb ? ((size_t)1 << (p.bits[0] + p.bits[1])) * 4 : 0);
  w.lb_dump = take(lb ? (size_t)TR_THREADS * 4 : 0);
  w.gsum = take((size_t)(TR_GROUPS + 1) * 2048 * 4);  // group sums + per-digit totals
  w.tile_row = take((size_t)(p.ntiles0 + 2) * 4);
  w.desc = take(p.passes == 2 ? (size_t)p.ntiles_max[1] * 8 : 0);
  w.first_tile = take(p.passes == 2 ? (((size_t)1 << p.bits[0]) + 1) * 4 : 0);
  w.keys = take(p.passes == 3 ? (size_t)nnz * 4 : 0);
  w.dump = take((size_t)TR_GRID * 16);
  w.total = off;
  return w;
}

template <bool FIRST, bool LAST>
int launch_scatter_one_sweep(const TrArgs& a, hipStream_t s) {
  const int nb = 1 << a.bits;
  size_t lds = (size_t)TR_WAVES * nb * 2 + (size_t)nb * 8 + (size_t)nb * 4 + (size_t)TR_TILE * 8 + (size_t)TR_TILE * 2;
  if (FIRST) lds += (size_t)TR_TILE * 4;  // rowid
  if (!FIRST) lds += (size_t)nb * 4;      // lb_tot
  auto k = tr_scatter_staged_kernel<FIRST, LAST, true>;
  if (lds > 64 * 1024) MI_HIP_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k, dim3(TR_GRID), dim3(TR_THREADS), lds, s, a);
  return mi::check_launch();
}

template <bool FIRST, bool LAST>
int launch_scatter(const TrArgs& a, bool packed, bool staged, hipStream_t s) {
  const int nb = 1 << a.bits;
  size_t lds = (size_t)TR_WAVES * nb * 2 + (size_t)nb * 8;
  if (staged) lds += (size_t)nb * 4 + (size_t)TR_TILE * 8 + (size_t)TR_TILE * 2;
  if (FIRST) lds += (size_t)TR_TILE * 4;  // rowid
  const dim3 grid((unsigned)(a.ntiles < TR_GRID ? a.ntiles : TR_GRID));
#define MI_TR(K_)                                                                                                  \
  do {                                                                                                             \
    auto k = K_;                                                                                                   \
    if (lds > 64 * 1024) MI_HIP_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL(k, grid, dim3(TR_THREADS), lds, s, a);                                                       \
  } while (0)
  if (packed && staged) MI_TR((tr_scatter_staged_kernel<FIRST, LAST, false>));
  else if (packed) MI_TR((tr_scatter_kernel<FIRST, LAST, true>));
  else MI_TR((tr_scatter_kernel<FIRST, LAST, false>));
#undef MI_TR
  return mi::check_launch();
}


int transpose_impl(const int32_t* rowptr, const int32_t* col, const float* val, int64_t nnz, int32_t batch, int32_t M,
                   int32_t K, int32_t* t_rowptr, int32_t* t_col, float* t_val, void* workspace, size_t workspace_bytes,
                   hipStream_t s) {
  if (batch < 0 || M < 0 || K < 0 || nnz < 0) return MI_EINVAL;
  if (nnz > 0x7fffffffLL || (long)batch * ((long)M + 1) > 0x7fffffffLL || (long)batch * (long)K > 0x7fffffffLL ||
      (long)batch * ((long)K + 1) > 0x7fffffffLL)
    return MI_ERANGE;
  if (batch == 0) return MI_OK;
  if (!t_rowptr) return MI_EINVAL;
  if (nnz == 0 || K == 0) {
    MI_HIP_TRY(hipMemsetAsync(t_rowptr, 0, sizeof(int32_t) * (size_t)batch * ((size_t)K + 1), s));
    return MI_OK;
  }
  if (!rowptr || !col || !val || !t_col || !t_val || M == 0) return MI_EINVAL;
  // small items: one workgroup per item, inside its LDS (no workspace needed; the size query still covers the general plan,
  // which MI_TRANSPOSE_PLAN_TABLES / _ONE_SWEEP pin)
  if (g_tr_plan.load(std::memory_order_relaxed) == MI_TRANSPOSE_PLAN_AUTO) {
    const int waves = mi::tr_item_lds_waves(nnz, batch, M, K);
    if (waves > 0) return mi::launch_tr_item_lds(waves, nnz, rowptr, col, val, batch, M, K, t_rowptr, t_col, t_val, s);
  }
  const TrPlan p = make_plan(batch, M, K, nnz);
  const TrWs w = ws_layout(p, nnz);
  if (workspace_bytes < w.total) return MI_ENOMEM;
  if (!workspace || !mi::aligned16(workspace)) return MI_EINVAL;
  char* base = static_cast<char*>(workspace);

  TrArgs a{};
  a.rowptr = rowptr;
  a.col = col;
  a.val = val;
  a.nnz = nnz;
  a.batch = batch;
  a.M = M;
  a.K = K;
  a.row_bits = p.row_bits;
  a.tile_row = reinterpret_cast<int*>(base + w.tile_row);
  a.t_col = t_col;
  a.t_val = t_val;
  a.keys_out = p.passes == 3 ? reinterpret_cast<unsigned*>(base + w.keys) : nullptr;
  a.dump = reinterpret_cast<uint2*>(base + w.dump);
  int* gsum = reinterpret_cast<int*>(base + w.gsum);
  int* tables[3];
  for (int i = 0; i < 3; ++i) tables[i] = reinterpret_cast<int*>(base + w.tables[i]);
  int* first_tile = reinterpret_cast<int*>(base + w.first_tile);
  int2* desc = reinterpret_cast<int2*>(base + w.desc);

  auto set_inter = [&](int which, bool as_input) {
    char* q = base + w.inter[which];
    if (p.packed) {
      if (as_input) a.in_packed = reinterpret_cast<const uint2*>(q);
      else a.out_packed = reinterpret_cast<uint2*>(q);
    } else {
      unsigned* k = reinterpret_cast<unsigned*>(q);
      unsigned* r = k + nnz;
      float* v = reinterpret_cast<float*>(r + nnz);
      if (as_input) {
        a.in_key = k;
        a.in_row = r;
        a.in_val = v;
      } else {
        a.out_key = k;
        a.out_row = r;
        a.out_val = v;
      }
    }
  };

  const int plan = g_tr_plan.load(std::memory_order_relaxed);
  if (plan == MI_TRANSPOSE_PLAN_ONE_SWEEP && !one_sweep_applies(p, batch, nnz)) return MI_EINVAL;
  // AUTO: where it pays — measured on MI355X (tools/bench_transpose_plans.py): 110 M non-zeros (13 422 tiles) 1.73 → 1.56 ms,
  // 6 M / 10.5 M non-zeros (733 / 1 283 tiles) 0.147 → 0.196 / 0.258 → 0.330 ms: the look-back costs ≈1 µs per tile and
  // workgroup, the launches it replaces ≈19 ns per tile of the whole problem
  if (plan != MI_TRANSPOSE_PLAN_TABLES && one_sweep_applies(p, batch, nnz) &&
      (plan == MI_TRANSPOSE_PLAN_ONE_SWEEP || p.ntiles0 >= 16 * TR_GRID)) {
    const int nb0 = 1 << p.bits[0], nb1 = 1 << p.bits[1];
    const int gshift = p.bits[0] - 5;                      // 32 groups of bins for the last pass
    const int ntiles0 = (int)p.ntiles0;
    int* h2g = reinterpret_cast<int*>(base + w.h2g);
    int* tickets = reinterpret_cast<int*>(base + w.tickets);
    int* hibase = reinterpret_cast<int*>(base + w.hibase);
    int* grp_first = reinterpret_cast<int*>(base + w.grp_first);
    int* tile_bin = reinterpret_cast<int*>(base + w.tile_bin);
    // status words of the last pass (its table area), h2, tickets, error flag: zero; so is the first table's extra row
    MI_HIP_TRY(hipMemsetAsync(base + w.tables[1], 0, (w.lbz + w.lbz_bytes) - w.tables[1], s));
    MI_HIP_TRY(hipMemsetAsync(tables[0] + (size_t)ntiles0 * nb0, 0, (size_t)nb0 * 4, s));
    a.ntiles = ntiles0;
    a.lb_dump = reinterpret_cast<unsigned*>(base + w.lb_dump);
    a.errflag = reinterpret_cast<int*>(base + w.errflag);
    // first pass: by the low digit, CSR arrays → packed intermediate — the table plan's, its count launch also counting
    // the last pass's per-(bin group, high digit) totals
    a.shift = p.shift[0];
    a.bits = p.bits[0];
    a.drop_after_first = 1;
    a.desc = nullptr;
    a.used_tiles = nullptr;
    a.table = tables[0];
    set_inter(0, false);
    const size_t hist_lds = ((size_t)nb0 + (size_t)TR_LB_GROUPS * nb1) * 4;
    MI_HIP_TRY(hipFuncSetAttribute((const void*)tr_hist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hist_lds));
    hipLaunchKernelGGL(tr_hist_kernel, dim3(TR_GRID), dim3(TR_THREADS), hist_lds, s, a, p.bits[0], p.bits[1], gshift, h2g);
    {
      const int rows = ntiles0 + 1;
      const dim3 sg(TR_GROUPS, (unsigned)((nb0 + 255) / 256));
      hipLaunchKernelGGL(tr_group_sums_kernel, sg, dim3(256), 0, s, tables[0], rows, nb0, gsum);
      int* dtot = gsum + (size_t)TR_GROUPS * 2048;
      hipLaunchKernelGGL(tr_group_bases_kernel, dim3((unsigned)((nb0 + 63) / 64)), dim3(1024), 0, s, gsum, nb0, dtot);
      hipLaunchKernelGGL(tr_apply_kernel, sg, dim3(256), 0, s, tables[0], rows, nb0, gsum, dtot);
    }
    int st = mi::check_launch();
    if (st != MI_OK) return st;
    st = launch_scatter<true, false>(a, true, true, s);
    if (st != MI_OK) return st;
    hipLaunchKernelGGL(tr_lb_setup_kernel, dim3(1), dim3(1024), 0, s, p.bits[0], p.bits[1], gshift, (long)nnz,
                       (int)p.ntiles_max[1], tables[0], h2g, hibase, first_tile, desc, tile_bin, grp_first);
    // last pass: by the high digit, one tile never holds two bins; offsets by look-back inside 32 groups of bins
    a.shift = p.shift[1];
    a.bits = p.bits[1];
    a.drop_after_first = 0;
    a.desc = desc;
    a.ntiles = (int)p.ntiles_max[1];
    set_inter(0, true);
    a.status = reinterpret_cast<unsigned*>(tables[1]);
    a.grp_first = grp_first;
    a.grp_base = h2g;
    a.tickets = tickets;
    a.tile_bin = tile_bin;
    a.first_tile = first_tile;
    a.rowoff = reinterpret_cast<int*>(base + w.rowoff);
    st = launch_scatter_one_sweep<false, true>(a, s);
    if (st != MI_OK) return st;
    const long total = (long)batch * ((long)K + 1);
    hipLaunchKernelGGL(tr_rowptr_lb_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a.rowoff, first_tile, hibase,
                       p.bits[0], p.bits[1], batch, K, (long)nnz, t_rowptr);
    return mi::check_launch();
  }

  for (int pass = 0; pass < p.passes; ++pass) {
    const bool first = pass == 0, last = pass == p.passes - 1;
    const int nb = 1 << p.bits[pass];
    a.shift = p.shift[pass];
    a.bits = p.bits[pass];
    a.drop_after_first = (p.drop && first) ? 1 : 0;
    a.table = tables[pass];
    a.desc = nullptr;
    a.used_tiles = nullptr;
    a.ntiles = (int)p.ntiles0;
    if (!first) set_inter(pass - 1, true);
    if (!last) set_inter(pass, false);
    if (p.passes == 2 && pass == 1) {
      // tiles that do not straddle the low-digit bins (→ row offsets straight from the table)
      hipLaunchKernelGGL(tr_bin_tiles_kernel, dim3(1), dim3(1024), 0, s, tables[0], 1 << p.bits[0], (long)nnz,
                         (int)p.ntiles_max[1], first_tile, desc);
      a.desc = desc;
      a.used_tiles = first_tile + (1 << p.bits[0]);
      a.ntiles = (int)p.ntiles_max[1];
    }
    const int rows = a.ntiles + 1;
    // the extra last row must read as zeros before the scan
    MI_HIP_TRY(hipMemsetAsync(tables[pass] + (size_t)a.ntiles * nb, 0, (size_t)nb * 4, s));
    const size_t hist_lds = (size_t)nb * 4;
    if (first) {
      if (p.packed) hipLaunchKernelGGL((tr_count_kernel<true, true>), dim3((unsigned)a.ntiles), dim3(TR_COUNT_THREADS), hist_lds, s, a);
      else hipLaunchKernelGGL((tr_count_kernel<true, false>), dim3((unsigned)a.ntiles), dim3(TR_COUNT_THREADS), hist_lds, s, a);
    } else {
      if (p.packed) hipLaunchKernelGGL((tr_count_kernel<false, true>), dim3((unsigned)a.ntiles), dim3(TR_COUNT_THREADS), hist_lds, s, a);
      else hipLaunchKernelGGL((tr_count_kernel<false, false>), dim3((unsigned)a.ntiles), dim3(TR_COUNT_THREADS), hist_lds, s, a);
    }
    const dim3 sg(TR_GROUPS, (unsigned)((nb + 255) / 256));
    hipLaunchKernelGGL(tr_group_sums_kernel, sg, dim3(256), 0, s, tables[pass], rows, nb, gsum);
    int* dtot = gsum + (size_t)TR_GROUPS * 2048;
    hipLaunchKernelGGL(tr_group_bases_kernel, dim3((unsigned)((nb + 63) / 64)), dim3(1024), 0, s, gsum, nb, dtot);
    hipLaunchKernelGGL(tr_apply_kernel, sg, dim3(256), 0, s, tables[pass], rows, nb, gsum, dtot);
    int st = mi::check_launch();
    if (st != MI_OK) return st;
    if (first && last) st = launch_scatter<true, true>(a, p.packed, p.staged[pass], s);
    else if (first) st = launch_scatter<true, false>(a, p.packed, p.staged[pass], s);
    else if (last) st = launch_scatter<false, true>(a, p.packed, p.staged[pass], s);
    else st = launch_scatter<false, false>(a, p.packed, p.staged[pass], s);
    if (st != MI_OK) return st;
  }
  if (p.passes <= 2) {
    const long total = (long)batch * ((long)K + 1);
    hipLaunchKernelGGL(tr_rowptr_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, tables[0],
                       p.passes == 2 ? tables[1] : tables[0], first_tile, p.bits[0], p.bits[1], p.passes, batch, K,
                       (long)nnz, t_rowptr);
  } else {
    hipLaunchKernelGGL(tr_boundaries_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, s, a.keys_out, (long)nnz,
                       batch, K, t_rowptr);
  }
  return mi::check_launch();
}

}  // namespace

extern "C" {

size_t mi_csr_transpose_batched_workspace_bytes(int32_t batch, int32_t M, int32_t K, int64_t nnz) {
  if (batch <= 0 || M < 0 || K < 0 || nnz < 0) return 0;
  return ws_layout(make_plan(batch, M, K, nnz), nnz).total;
}

size_t mi_csr_transpose_workspace_bytes(int32_t M, int32_t K, int64_t nnz) {
  return mi_csr_transpose_batched_workspace_bytes(1, M, K, nnz);
}

int mi_csr_transpose_f32(const int32_t* rowptr, const int32_t* col, const float* val, int64_t nnz, int32_t M,
                         int32_t K, int32_t* t_rowptr, int32_t* t_col, float* t_val, void* workspace,
                         size_t workspace_bytes, mi_stream_t stream) {
  return transpose_impl(rowptr, col, val, nnz, 1, M, K, t_rowptr, t_col, t_val, workspace, workspace_bytes,
                        static_cast<hipStream_t>(stream));
}

int mi_csr_transpose_set_plan(int plan) {
  if (plan != MI_TRANSPOSE_PLAN_AUTO && plan != MI_TRANSPOSE_PLAN_TABLES && plan != MI_TRANSPOSE_PLAN_ONE_SWEEP) return MI_EINVAL;
  g_tr_plan.store(plan, std::memory_order_relaxed);
  return MI_OK;
}

int mi_csr_transpose_one_sweep_applies(int32_t batch, int32_t M, int32_t K, int64_t nnz) {
  if (batch <= 0 || M <= 0 || K <= 0 || nnz <= 0 || nnz > 0x7fffffffLL) return 0;
  return one_sweep_applies(make_plan(batch, M, K, nnz), batch, nnz) ? 1 : 0;
}

int mi_csr_transpose_auto_takes_one_sweep(int32_t batch, int32_t M, int32_t K, int64_t nnz) {
  if (batch <= 0 || M <= 0 || K <= 0 || nnz <= 0 || nnz > 0x7fffffffLL) return 0;
  const int plan = g_tr_plan.load(std::memory_order_relaxed);
  if (plan == MI_TRANSPOSE_PLAN_TABLES) return 0;
  if (plan == MI_TRANSPOSE_PLAN_AUTO && mi::tr_item_lds_waves(nnz, batch, M, K) > 0) return 0;
  const TrPlan p = make_plan(batch, M, K, nnz);
  return one_sweep_applies(p, batch, nnz) && (plan == MI_TRANSPOSE_PLAN_ONE_SWEEP || p.ntiles0 >= 16 * TR_GRID) ? 1 : 0;
}

int mi_csr_transpose_check(const void* workspace, size_t workspace_bytes, int32_t batch, int32_t M, int32_t K,
                           int64_t nnz, mi_stream_t stream) {
  if (batch <= 0 || M < 0 || K < 0 || nnz < 0 || nnz > 0x7fffffffLL) return MI_EINVAL;
  const TrPlan p = make_plan(batch, M, K, nnz);
  if (!one_sweep_applies(p, batch, nnz)) return MI_OK;  // only the one-sweep plan has anything to report
  const TrWs w = ws_layout(p, nnz);
  if (!workspace || workspace_bytes < w.total) return MI_EINVAL;
  int flag = 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  MI_HIP_TRY(hipMemcpyAsync(&flag, static_cast<const char*>(workspace) + w.errflag, sizeof(int), hipMemcpyDeviceToHost, s));
  MI_HIP_TRY(hipStreamSynchronize(s));
  return flag == 0 ? MI_OK : MI_EHIP;
}

#ifdef MI_TR_ITEM_TIMING
int mi_tr_item_stamps(unsigned long long* out8) {  // reads and clears the phase counters (synchronises the device)
  MI_HIP_TRY(hipDeviceSynchronize());
  MI_HIP_TRY(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_tr_item_stamp), 8 * sizeof(unsigned long long)));
  unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  MI_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_tr_item_stamp), zero, sizeof(zero)));
  return MI_OK;
}
#endif

int mi_csr_transpose_batched_in_lds(int64_t nnz, int32_t batch, int32_t M, int32_t K) {
  return g_tr_plan.load(std::memory_order_relaxed) == MI_TRANSPOSE_PLAN_AUTO && mi::tr_item_lds_waves(nnz, batch, M, K) > 0 ? 1 : 0;
}

int mi_csr_transpose_batched_f32(const int32_t* rowptr, const int32_t* col, const float* val, int64_t nnz,
                                 int32_t batch, int32_t M, int32_t K, int32_t* t_rowptr, int32_t* t_col,
                                 float* t_val, void* workspace, size_t workspace_bytes, mi_stream_t stream) {
  return transpose_impl(rowptr, col, val, nnz, batch, M, K, t_rowptr, t_col, t_val, workspace, workspace_bytes,
                        static_cast<hipStream_t>(stream));
}

}  // extern "C"

