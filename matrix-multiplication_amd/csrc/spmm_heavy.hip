// Rows with hundreds to hundreds of thousands of entries, summed at a CU's gather rate instead of one wave's: the heavy rows
// of a row schedule (spmm_sched.hip) and the rows beyond the long-row threshold (spmm_long.hip lists them) — what decides the
// time of a product that is over in a fraction of a millisecond (an adjacency matrix with power-law degrees: 1 % of the rows
// hold a sixth of the entries, a handful of rows ten thousand each).
//
// The ARITHMETIC is fixed elsewhere and untouched here.  A row of ≤ 8192 entries is ONE fmaf chain per output element in CSR
// order (include/mi_spmm.h; reference src/naive_sparse_mm.cu:60-92); a longer row is the fixed split the oracle restates
// (oracle_spmm_csr_long_f32; spmm_long.hip): 16·S chains over its 1024-entry chunks dealt round-robin, the chains of a group
// added in order, the S groups added in order.  Both make every plan, every schedule and every GPU count give the same bits.
// Left to one wave, a chain advances at (gathers in flight) / (memory latency): 8 rows of B per ≈ 1.4 µs.  The chain itself
// is cheap; what a long row lacks is memory-level parallelism.  So here a whole 8-wave workgroup works on ONE chain at a time:
//   * the loader waves (seven) gather the B rows of the next `E` entries (224 at 64 columns) into one of two LDS images —
//     registers first, two chunks ahead, every load of a chunk in flight at once, then ds_write;
//   * the first wave (one lane per output element) walks the previous image in entry order: one LDS read + one FMA per entry,
//     the chain — while the other waves' loads of the next chunks are in the air.  One barrier per chunk.
// and the COLUMNS of a row are dealt to workgroups 64 at a time (grid.y): a CU gathers ≈ 30 GB/s from the Infinity Cache
// whatever it does (the guide's figure: 33), so a 7 806-entry row of 128 columns is 4 MB = 138 µs on one CU and half that on
// two — columns are independent chains, the split costs nothing in bits.
// Measured (tools/probes/skew_trace.py, 170 K rows × 128, Pareto lengths): DESIGN.md §3.6.  History of the round: every wave
// gathering AND the first waves chaining, 191 µs for the 93 heavy rows of that matrix (the phases of a step added up: chain
// 1.15, LDS stores 0.67, exposed loads 0.5 µs per 128 entries); roles split, all 128 columns in one workgroup, 138 µs
// (gather-rate bound: 1.7 µs per 96 entries = 29 GB/s); tried and dropped: a float4 per chain lane (1.7 µs per chunk for the
// chain alone), s_setprio for the workgroup beside the ordinary launch (no change: the contention is not for issue slots).
// A branch around each load costs a factor of 1.6: hipcc then waits `vmcnt(0)` before every one of them (8 dependent trips
// to memory per chunk) — every slot is made valid instead.  New relative to the reference (one warp per row and 32 columns,
// whatever the row).
#include <cstdlib>

#include "spmm_device.h"
#include "spmm_internal.h"

namespace {

using mi::LongArg;

// four floats of a row of B / C / a partial row at ANY 4-byte alignment (N % 4 != 0, odd leading dimensions, offset views): a
// dword-aligned global_load / store_dwordx4, which gfx950 serves; where the address is 16-byte aligned it is the same instruction
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int kStageThreads = 512;
constexpr int kStageCols = 64;                     // columns per workgroup when rows are few: one chain wave, seven loader waves
constexpr int kStageColsWide = 128;                // … when there are heavy rows enough to fill the chip: two and six
constexpr int kStageFloats = 16384;                // floats per LDS image (64 KB); two images
constexpr int kStageMaxE = 256;                    // entries per chunk at most (cols · (E + 4) floats fill an image)

// The sub-chunks of one workgroup's work, in the order BOTH roles walk them: chain w = 0 … nch−1 of the group takes the
// L-entry chunks first + w·L, + stride, + 2·stride, … below `end`, each in sub-chunks of ≤ E entries.  A heavy row is one chain
// over [start, end): nch = 1, L = stride = "never".  Scalar code, the same in every wave.
struct StageIt {
  long p, ce, cb;
  int w;
};
struct StageRange {
  long first, end, L, stride;
  int nch, E;
};
__device__ __forceinline__ void stage_begin(StageIt& it, const StageRange& r) {
  it.w = r.first < r.end ? 0 : r.nch;  // (chains' first chunks ascend: when chain w has none, no later chain has)
  it.cb = r.first;
  it.p = r.first;
  it.ce = r.first + r.L < r.end ? r.first + r.L : r.end;
}
// the next sub-chunk: entries [p0, p0 + count) of chain w0; count 0 when the work is exhausted (and from then on)
__device__ __forceinline__ int stage_next(StageIt& it, const StageRange& r, long& p0, int& w0) {
  if (it.w >= r.nch) return 0;
  p0 = it.p;
  w0 = it.w;
  const long left = it.ce - it.p;
  const int count = left < r.E ? (int)left : r.E;
  it.p += count;
  if (it.p >= it.ce) {
    it.cb += r.stride;
    if (it.cb >= r.end) {
      ++it.w;
      it.cb = r.first + it.w * r.L;
      if (it.cb >= r.end) it.w = r.nch;
    }
    it.p = it.cb;
    it.ce = it.cb + r.L < r.end ? it.cb + r.L : r.end;
  }
  return count;
}

// One workgroup's sum over `r` for columns [0, ncols) of Bp (already offset to the workgroup's first column): returns, in thread
// tid < ncols, chain 0 + chain 1 + … + chain nch−1 (left to right; a chain without entries adds + 0.0f, as a wave without chunks
// does in the one-wave-per-chain form) — garbage in every other thread.  Called by the whole workgroup; ends with a barrier (the
// images are free for the next call).
//
// The LDS image of a chunk is TRANSPOSED: column c's entries lie side by side (pitch EP = E + 4 floats), so the chain lane
// takes four entries with one ds_read_b128 — a single wave issuing ds_read_b32 gets a fifth of the LDS rate (the guide's LDS
// table), and with one read per entry the chain, not the gather, set the pace (16 ns per entry; measured).  EP = 4·odd keeps those
// reads free of bank conflicts.  The loaders pay with four ds_write_b32 per float4; their lanes are dealt 8 entries × 8 quads
// per wave-instruction — whole 128-byte lines of B on the way in, 2-way conflicts (free for that store) on the way to LDS.
// The matrix values travel with the rows of B (one loader lane per entry stores them behind the images; the chain lanes read
// them four at a time, a broadcast).  Taking them from memory through the scalar unit instead was tried: every 64 bytes a new
// line, ≈ 500 cycles exposed per 32 entries — 13 ns per entry against 16 before the transposition.
template <int KQ>
__device__ __forceinline__ float staged_sum(const int* __restrict__ col, const float* __restrict__ val,
                                            const float* __restrict__ Bp, long ldb, int ncols, const StageRange r) {
  // two images of kStageFloats floats, then the two chunks' values — addressed by integer offsets into the one array (a pointer picked by the buffer index
  // would become a table of generic pointers to LDS, which the back end refuses)
  extern __shared__ __attribute__((aligned(16))) float stage_lds[];
  const int tid = threadIdx.x;
  const int E = r.E, EP = r.E + 4;
  StageIt it;
  stage_begin(it, r);
  long p0 = 0;
  int w0 = 0;

  const int CW = (ncols + 63) >> 6;  // chain waves: one lane per output element
  if (tid < CW * 64) {
    // ---- chain role: this lane's output element over the images in turn; entry order, one fmaf per entry.  16 entries per
    // batch, two register sets: batch i + 1 is read while batch i's FMAs run.
    float tot = 0.f, acc1 = 0.f;
    int cur = 0;
    const bool mine = tid < ncols;
#define MI_STAGE_READ16(Y_, S_, E_)                                                                                  \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) Y_[i] = lds4[src4 + ((E_) >> 2) + i];                                  \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) S_[i] = lds4[vals4 + ((E_) >> 2) + i];
#define MI_STAGE_FMA16(Y_, S_)                                 \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {              \
    acc1 = __builtin_fmaf(S_[i].x, Y_[i].x, acc1);             \
    acc1 = __builtin_fmaf(S_[i].y, Y_[i].y, acc1);             \
    acc1 = __builtin_fmaf(S_[i].z, Y_[i].z, acc1);             \
    acc1 = __builtin_fmaf(S_[i].w, Y_[i].w, acc1);             \
  }
    // chains cur … UPTO_ − 1 are complete: fold them into the running total, left to right
#define MI_STAGE_FOLD(UPTO_)                                   \
  {                                                            \
    tot = cur == 0 ? acc1 : tot + acc1;                        \
    for (int k_ = cur + 1; k_ < (UPTO_); ++k_) tot += 0.f;     \
    acc1 = 0.f;                                                \
    cur = (UPTO_);                                             \
  }
    int b = 0;
    for (int count = stage_next(it, r, p0, w0); count > 0; count = stage_next(it, r, p0, w0)) {
      __syncthreads();  // image b is whole
      if (w0 != cur) MI_STAGE_FOLD(w0)
      if (mine) {
        // (in units of four floats: the compiler must SEE the 16-byte alignment to emit ds_read_b128)
        const int src4 = b * (kStageFloats / 4) + tid * ((E >> 2) + 1), src = 4 * src4;
        const f32x4* lds4 = reinterpret_cast<const f32x4*>(stage_lds);
        const int vals4 = (2 * kStageFloats + b * kStageMaxE) / 4;
        const int full = count & ~15;
        f32x4 ya[4], yb[4];
        f32x4 sa[4], sb[4];
        int e = 0;
        if (full > 0) { MI_STAGE_READ16(ya, sa, 0) }
        for (; e + 32 <= full; e += 32) {
          MI_STAGE_READ16(yb, sb, e + 16)
          MI_STAGE_FMA16(ya, sa)
          if (e + 32 < full) { MI_STAGE_READ16(ya, sa, e + 32) }
          MI_STAGE_FMA16(yb, sb)
        }
        if (e < full) {
          MI_STAGE_FMA16(ya, sa)
          e += 16;
        }
        for (; e < count; ++e) acc1 = __builtin_fmaf(stage_lds[4 * vals4 + e], stage_lds[src + e], acc1);
      }
      b ^= 1;
    }
    MI_STAGE_FOLD(r.nch)
#undef MI_STAGE_READ16
#undef MI_STAGE_FMA16
#undef MI_STAGE_FOLD
    __syncthreads();  // the last image has been read
    return tot;
  }

  // ---- loader role: this thread's float4s of a chunk, the same slots for every chunk.  Quad q = lt + LT·k lies in a block of
  // 8 entries × NQ float4 columns: entry 8·(q / 8NQ) + (q mod 8), float4 column (q mod 8NQ) / 8 — a wave-instruction covers 8
  // entries × 8 quads (one 128-byte line of each of 8 rows of B).  EVERY slot is made valid (a quad beyond the chunk's E·NQ
  // repeats the last one; an entry beyond a short chunk repeats its last entry): the loads and stores below are then
  // straight-line code.  That matters — with a branch around each load hipcc cannot count what is in flight and puts
  // `s_waitcnt vmcnt(0)` in front of every one of them: eight dependent trips to memory per chunk instead of one (measured: × 1.6).
  const int lt = tid - CW * 64;
  const int LT = kStageThreads - CW * 64;  // loader threads
  const int NQ = (ncols + 3) >> 2;  // the last quad of a width that is no multiple of four is shifted back to end at the last column: it
                                    // overlaps its neighbour, the shared columns land twice in the image with the same bits (ncols ≥ 4)
  int qe[KQ], qc[KQ];
#pragma unroll
  for (int k = 0; k < KQ; ++k) {
    int q = lt + LT * k;
    q = q < E * NQ ? q : E * NQ - 1;
    const int blk = q / (8 * NQ), rr = q - blk * 8 * NQ;
    qe[k] = 8 * blk + (rr & 7);
    qc[k] = 4 * (rr >> 3) + 4 <= ncols ? 4 * (rr >> 3) : ncols - 4;  // first column of the quad
  }
  // B rows travel two chunks ahead of the chain (register sets xa / xb, taking turns), col / val three chunks ahead
  f32x4 xa[KQ], xb[KQ];
  float xva[KQ], xvb[KQ];
  int cn[KQ];
  float vn[KQ];
#define MI_STAGE_FETCH(P0_, COUNT_)                                           \
  _Pragma("unroll") for (int k = 0; k < KQ; ++k) {                            \
    const long i_ = (P0_) + (qe[k] < (COUNT_) ? qe[k] : (COUNT_) - 1);        \
    cn[k] = col[i_];                                                          \
    vn[k] = val[i_];                                                          \
  }
#define MI_STAGE_ISSUE(X_, XV_)                                                      \
  _Pragma("unroll") for (int k = 0; k < KQ; ++k) {                                   \
    X_[k] = *reinterpret_cast<const f32x4_u*>(Bp + (long)cn[k] * ldb + qc[k]);       \
    XV_[k] = vn[k];                                                                  \
  }
  // (entries of the image beyond a short chunk's count receive copies of its last entry: the chain never reads them)
#define MI_STAGE_LAND(X_, XV_, IMG_, VALS_)                            \
  _Pragma("unroll") for (int k = 0; k < KQ; ++k) {                     \
    const int at_ = (IMG_) + qc[k] * EP + qe[k];                       \
    stage_lds[at_] = X_[k].x;                                          \
    stage_lds[at_ + EP] = X_[k].y;                                     \
    stage_lds[at_ + 2 * EP] = X_[k].z;                                 \
    stage_lds[at_ + 3 * EP] = X_[k].w;                                 \
    if (qc[k] == 0) stage_lds[(VALS_) + qe[k]] = XV_[k]; /* one lane per entry: the first quad's */ \
  }
  // One step: chunk A lands from register set X_ in image I_; the barrier; chunk C (two ahead) takes the freed registers,
  // col / val of chunk D (three ahead) follow.
#define MI_STAGE_STEP(X_, XV_, I_)                                                          \
  {                                                                                         \
    MI_STAGE_LAND(X_, XV_, (I_) * kStageFloats, 2 * kStageFloats + (I_) * kStageMaxE)       \
    __syncthreads(); /* image I_ is whole; the chain is done with the other image */        \
    int cD = 0;                                                                             \
    long pD = 0;                                                                            \
    if (cC > 0) {                                                                           \
      MI_STAGE_ISSUE(X_, XV_)                                                               \
      cD = stage_next(it, r, pD, w0);                                                       \
      if (cD > 0) MI_STAGE_FETCH(pD, cD)                                                    \
    }                                                                                       \
    cA = cB;                                                                                \
    cB = cC;                                                                                \
    cC = cD;                                                                                \
  }
  long pA = 0, pB = 0, pC = 0;
  int cA = stage_next(it, r, pA, w0);
  int cB = cA > 0 ? stage_next(it, r, pB, w0) : 0;
  int cC = cB > 0 ? stage_next(it, r, pC, w0) : 0;
  if (cA > 0) {
    MI_STAGE_FETCH(pA, cA)
    MI_STAGE_ISSUE(xa, xva)
    if (cB > 0) {
      MI_STAGE_FETCH(pB, cB)
      MI_STAGE_ISSUE(xb, xvb)
      if (cC > 0) MI_STAGE_FETCH(pC, cC)
    }
  }
  while (cA > 0) {
    MI_STAGE_STEP(xa, xva, 0)
    if (cA <= 0) break;
    MI_STAGE_STEP(xb, xvb, 1)
  }
#undef MI_STAGE_FETCH
#undef MI_STAGE_ISSUE
#undef MI_STAGE_LAND
#undef MI_STAGE_STEP
  __syncthreads();  // (the chain's closing barrier)
  return 0.f;
}

// How many columns a workgroup takes: 64 while that leaves workgroups scarce (a few rows: every CU that joins adds its gather
// rate); once the rows alone fill the chip the product is bound by the memory system, and 256-byte pieces of rows fetched by
// different CUs at different times cost it (185 long rows beside config C3's shape, nothing else running: 64 columns 17.1 ms,
// 128 17.4, 256 19.0 — the chunks get short —, one wave per chain over whole rows 16.8).  So with ≥ 128 units the heavy slots
// take 128 columns per workgroup, and the listed rows go one wave per chain (below).  The same rule on the host (heavy slots)
// and on the device (listed rows).
__host__ __device__ __forceinline__ bool stage_is_bulk(long units, int N) {
  return units * ((N + 255) / 256) >= 128;
}

// What one launch of spmm_staged_rows_kernel works on: the listed rows beyond the long-row threshold (blocks [0, list_blocks),
// grid-stride over the list) and / or the heavy slots of a schedule (the blocks behind them, one per slot and column part).
struct StagedList {
  int* ws;  // nullptr: no list in this launch
  int cap_e, cap_s;
  float* partial;
  int reset, E, force;  // force: 0 the rule, 1 staged, 2 one wave per chain (measurements)
  int blocks;
};
struct StagedHeavy {
  LongArg la;  // order / nslots: the heavy slots (nslots 0: none); thresh, ws: rows beyond the threshold are skipped and listed
  int cols, E, parts;
};

// Heavy rows of a schedule: unit (slot, part) sums columns [cols·part, cols·part + cols) of row order[slot] — one chain.
template <int KQ>
__device__ __forceinline__ void staged_heavy_unit(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                  const float* __restrict__ val, const float* __restrict__ B,
                                                  float* __restrict__ C, int N, long ldb, long ldc,
                                                  const float* __restrict__ bias, const StagedHeavy& h, int unit) {
  const int slot = unit / h.parts, part = unit - slot * h.parts;
  if (slot >= h.la.nslots) return;
  const int row = h.la.order[slot];
  const int start = __builtin_amdgcn_readfirstlane(rowptr[row]), end = __builtin_amdgcn_readfirstlane(rowptr[row + 1]);
  if (end - start > h.la.thresh) {  // left to the long-row launch
    if (threadIdx.x == 0 && part == 0) long_list_append(h.la, row, end - start);
    return;
  }
  int n0 = part * h.cols;
  int ncols = N - n0 < h.cols ? N - n0 : h.cols;
  if (ncols < 4) n0 = N - 4, ncols = 4;  // a last part of 1 … 3 columns moves back over its neighbour's (computed twice, the same bits)
  const StageRange r = {start, end, 1L << 40, 1L << 40, 1, h.E};
  const float tot = staged_sum<KQ>(col, val, B + n0, ldb, ncols, r);
  if ((int)threadIdx.x < ncols) {
    float out = tot;
    if (bias) out += bias[n0 + threadIdx.x];
    __builtin_nontemporal_store(out, C + (long)row * ldc + n0 + threadIdx.x);
  }
}

// One wave per chain (the form of the 16-wave kernel of rounds 2 – 5, for eight waves): group g of a row's S — chains 16g … 16g + 15, two
// per wave one after the other, sixteen gathers of whole rows of B in flight per wave — for every column, 256 per pass; the
// sixteen chain sums meet in LDS and are added in order.  S = 1: the row of C (+ bias); else the group's partial row.
__device__ __forceinline__ void wave_chain_group(const int* __restrict__ col, const float* __restrict__ val,
                                                 const float* __restrict__ B, float* __restrict__ C, int N, long ldb, long ldc,
                                                 const float* __restrict__ bias, float* __restrict__ partial, int row,
                                                 long start, long end, int g, int S, int pb) {
  extern __shared__ __attribute__((aligned(16))) float stage_lds[];
  f32x4* part = reinterpret_cast<f32x4*>(stage_lds);  // [16][64]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long stride = (long)kLongWaves * S * kLongChunk;
  for (int n0 = 0; n0 < N; n0 += 256) {
    const bool on = n0 + lane * 4 < N;
    const int c0 = n0 + lane * 4 + 4 <= N ? n0 + lane * 4 : N - 4;  // the partial last quad, shifted back (N ≥ 4)
    for (int h = 0; h < 2; ++h) {
      const int w = 8 * h + wave;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (long cb = start + ((long)g * kLongWaves + w) * kLongChunk; cb < end; cb += stride) {
        const long ce = cb + kLongChunk < end ? cb + kLongChunk : end;
        for (long p = cb; p < ce; p += 64) {
          const long idx = p + lane;
          const int myc = idx < ce ? col[idx] : 0;
          const float myv = idx < ce ? val[idx] : 0.f;
          const int cnt = ce - p < 64 ? (int)(ce - p) : 64;
          int i = 0;
#define MI_WAVE_BATCH(U_)                                                                                         \
  for (; i + (U_) <= cnt; i += (U_)) {                                                                            \
    f32x4 x[U_];                                                                                                  \
    float v[U_];                                                                                                  \
    _Pragma("unroll") for (int u = 0; u < (U_); ++u) {                                                            \
      const int c = __builtin_amdgcn_readlane(myc, i + u);                                                        \
      v[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myv), i + u));           \
      if (on) x[u] = *reinterpret_cast<const f32x4_u*>(B + (long)c * ldb + c0);                                   \
    }                                                                                                             \
    _Pragma("unroll") for (int u = 0; u < (U_); ++u) if (on) acc = fma4(v[u], x[u], acc);                         \
  }
          MI_WAVE_BATCH(16)
          MI_WAVE_BATCH(4)
          MI_WAVE_BATCH(1)
#undef MI_WAVE_BATCH
        }
      }
      part[w * 64 + lane] = acc;
    }
    __syncthreads();
    if (wave == 0 && on) {
      f32x4 tot = part[lane];
#pragma unroll
      for (int w = 1; w < kLongWaves; ++w) tot += part[w * 64 + lane];
      if (S == 1) {
        if (bias) tot += *reinterpret_cast<const f32x4_u*>(bias + c0);
        __builtin_nontemporal_store(tot, reinterpret_cast<f32x4_u*>(C + (long)row * ldc + c0));
      } else {
        *reinterpret_cast<f32x4_u*>(partial + (long)(pb + g) * N + c0) = tot;
      }
    }
    __syncthreads();
  }
}

// Rows beyond the long-row threshold, from the list in `ws` (spmm_device.h: long_list_append; layout in spmm_long.hip): a unit of
// work is (slot t, column part) — group g of its row's S, chains 16g … 16g + 15 one after the other at the CU's gather rate,
// added in that order, for one part of the columns.  S = 1: that is the row's sum.  Else the S group sums meet in the workspace
// and whichever of the S · parts units delivers LAST adds them in order g = 0 … S−1 for every column (an arrival counter per
// row; agent-scope release by every deliverer, acquire by the last) — the arithmetic of the 16-wave kernel of rounds 2 – 5 (one wave
// per chain), which this one replaces for every width.
template <int KQ>
__device__ __forceinline__ void staged_list_units(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                  const float* __restrict__ val, const float* __restrict__ B,
                                                  float* __restrict__ C, int N, long ldb, long ldc,
                                                  const float* __restrict__ bias, const StagedList& l, int block) {
  __shared__ int last_flag;
  int* __restrict__ ws = l.ws;
  float* __restrict__ partial = l.partial;
  const int listed = ws[0], handed = ws[1], handed_p = ws[2];
  if (listed == 0 && handed == 0 && handed_p == 0) return;  // no long row: nothing to sum, nothing to reset
  const int count = listed < l.cap_e ? listed : l.cap_e;
  const int slots = handed < l.cap_s ? handed : l.cap_s;
  const int* owner = ws + 4 + kLongEnt * (long)l.cap_e;
  // few units: 64 columns each, the chains staged through LDS.  Enough to fill the chip: every column, one wave per chain.
  const bool bulk = l.force > 0 ? l.force == 2 : stage_is_bulk(slots, N);
  const int E = l.E;
  const int parts = bulk ? 1 : (N + kStageCols - 1) / kStageCols;
  for (long u = block; u < (long)slots * parts; u += l.blocks) {
    const int t = (int)(u / parts), part = (int)(u - (long)t * parts);
    const int e = owner[t];
    if ((unsigned)e >= (unsigned)count) continue;  // slot of a dropped entry
    int* ent = ws + 4 + kLongEnt * (long)e;
    const int row = ent[0], S = ent[2], pb = ent[3];
    const int g = t - ent[1];
    if ((unsigned)g >= (unsigned)S) continue;  // not a slot of that entry
    const long start = __builtin_amdgcn_readfirstlane(rowptr[row]), end = __builtin_amdgcn_readfirstlane(rowptr[row + 1]);
    if (bulk) {
      wave_chain_group(col, val, B, C, N, ldb, ldc, bias, partial, row, start, end, g, S, pb);
    } else {
      int n0 = part * kStageCols;
      int ncols = N - n0 < kStageCols ? N - n0 : kStageCols;
      if (ncols < 4) n0 = N - 4, ncols = 4;  // (as the heavy slots' last part)
      const StageRange r = {start + (long)g * kLongWaves * kLongChunk, end, kLongChunk, (long)kLongWaves * S * kLongChunk,
                            kLongWaves, E};
      const float tot = staged_sum<KQ>(col, val, B + n0, ldb, ncols, r);
      if (S == 1) {
        if ((int)threadIdx.x < ncols) {
          float out = tot;
          if (bias) out += bias[n0 + threadIdx.x];
          __builtin_nontemporal_store(out, C + (long)row * ldc + n0 + threadIdx.x);
        }
      } else if ((int)threadIdx.x < ncols) {
        partial[(long)(pb + g) * N + n0 + threadIdx.x] = tot;
      }
    }
    if (S == 1) continue;
    // Deliver: the chain waves are the only ones that stored partial sums.  Their stores are drained and written back at agent
    // scope, the workgroup meets, and only then does one lane take an arrival ticket (cdna_hip_programming.md Guideline 16:
    // fence before the ticket, with the explicit wait hipcc may drop).  The workgroup that draws the last of the S · parts
    // tickets acquires and adds the S partial rows in order g = 0 … S−1, then the bias.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      const int ticket = __hip_atomic_fetch_add(&ent[4], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last_flag = ticket == S * parts - 1;
      if (ticket == S * parts - 1) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    __syncthreads();
    if (last_flag) {
      for (int c = threadIdx.x; c < N; c += blockDim.x) {
        // sc1 loads: served by L2 / memory, never by a line this CU cached before the other workgroups wrote
        float sum = __hip_atomic_load(partial + (long)pb * N + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int gg = 1; gg < S; ++gg)
          sum += __hip_atomic_load(partial + (long)(pb + gg) * N + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (bias) sum += bias[c];
        C[(long)row * ldc + c] = sum;
      }
      if (threadIdx.x == 0) ent[4] = 0;  // a prepared list serves the next product too
    }
    __syncthreads();  // last_flag is rewritten by the next unit
  }
  if (l.reset) {
    // reset != 0: the list was built for this product only — the workgroup that finishes last zeroes the four counters (every
    // read of them by this workgroup is done: they were read into registers at the top)
    __syncthreads();
    if (threadIdx.x == 0) {
      const int done = __hip_atomic_fetch_add(&ws[3], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (done == l.blocks - 1) {
        __hip_atomic_store(&ws[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ws[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ws[2], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ws[3], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// ONE launch for both kinds of work (a prepared list beside a schedule's heavy slots: launched one behind the other on a
// stream, the list's two workgroups kept 93 heavy rows waiting for 170 µs): the list's blocks first — its units are the longest.
template <int KQ>
__global__ __launch_bounds__(kStageThreads) void spmm_staged_rows_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ val,
    const float* __restrict__ B, float* __restrict__ C, int N, long ldb, long ldc, const float* __restrict__ bias,
    StagedList l, StagedHeavy h) {
  const int block = blockIdx.x;
  if (block < l.blocks) staged_list_units<KQ>(rowptr, col, val, B, C, N, ldb, ldc, bias, l, block);
  else staged_heavy_unit<KQ>(rowptr, col, val, B, C, N, ldb, ldc, bias, h, block - l.blocks);
}

// entries per chunk and float4s per loader thread for rows of N columns dealt `cols` to a workgroup: an image holds the chunk
// (cols · (E + 4) ≤ 16384 floats), the loader threads (512 − 64 per chain wave) carry it in KQ float4s each, E a multiple of 16
// (the chain's batches).  64 columns: E = 224, KQ = 8; 128: E = 96; 256: E = 32; 32: E = 256 (2048 of 3584 slots carry: KQ = 4
// would halve the chunk); ≤ 16: E = 256, KQ = 4.
struct StageShape {
  int parts, E, KQ;
  size_t lds;
};
StageShape stage_shape(int N, int cols_per_wg) {
  StageShape sh;
  sh.parts = (N + cols_per_wg - 1) / cols_per_wg;
  const int cols = N < cols_per_wg ? N : cols_per_wg;
  const int nq = (cols + 3) / 4;
  const long loaders = kStageThreads - 64 * ((cols + 63) / 64);
  int E = kStageMaxE;
  if ((long)(E + 4) * cols > kStageFloats) E = kStageFloats / cols - 4;
  sh.KQ = N <= 16 ? 4 : 8;  // (one rule for both column widths: the kernels are instantiated per KQ)
  if ((long)E * nq > loaders * sh.KQ) E = (int)(loaders * sh.KQ / nq);
  sh.E = E / 16 * 16;
  sh.lds = (size_t)(2 * kStageFloats + 2 * kStageMaxE) * sizeof(float);
  return sh;
}

// MI_STAGE_COLS = 64 | 128 pins the heavy slots' column width (measurements; the bits do not depend on it)
int forced_cols() {
  static const int v = [] {
    const char* e = getenv("MI_STAGE_COLS");
    const int x = e ? atoi(e) : 0;
    return x == 64 || x == 128 ? x : 0;
  }();
  return v;
}
// MI_LONG_ROWS_FORM = staged | wave pins how the listed rows are summed (measurements)
int forced_list_form() {
  static const int v = [] {
    const char* e = getenv("MI_LONG_ROWS_FORM");
    return e == nullptr ? 0 : (e[0] == 's' ? 1 : (e[0] == 'w' ? 2 : 0));
  }();
  return v;
}
}  // namespace

namespace mi {

// The listed rows beyond the threshold (ws != nullptr) and / or the heavy slots of a schedule (la.nslots > 0), one launch.
int launch_staged_rows(int* ws, const LongWs& lw, bool reset, const LongArg& heavy, const int32_t* rowptr, const int32_t* col,
                       const float* val, const float* B, float* C, int32_t N, int64_t ldb, int64_t ldc, const float* bias,
                       hipStream_t s) {
  if (N < 4) return MI_EINVAL;  // (narrower products keep the narrow kernel's own order: the dispatcher never sends them)
  const int force = forced_cols();
  StagedList l = {};
  StagedHeavy h = {};
  h.la = heavy;
  h.parts = 1;
  const StageShape narrow = stage_shape(N, kStageCols), wide = stage_shape(N, kStageColsWide);
  if (ws != nullptr) {
    l.ws = ws, l.cap_e = (int)lw.cap_e, l.cap_s = (int)lw.cap_s;
    l.partial = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + lw.partial_off);
    l.reset = reset ? 1 : 0, l.E = narrow.E, l.force = forced_list_form();
    // three workgroups per CU (one is resident at a time: 130 KB of LDS), fewer when the list cannot be that long; grid-stride
    // over the units.  With no long row every workgroup reads three zeros and exits.
    const long most = lw.cap_s * (long)narrow.parts;
    l.blocks = (int)(most < 768 ? most : 768);
  }
  long heavy_blocks = 0;
  if (heavy.order != nullptr && heavy.nslots > 0) {
    h.cols = force > 0 ? force : (stage_is_bulk(heavy.nslots, N) ? kStageColsWide : kStageCols);
    const StageShape sh = h.cols == kStageCols ? narrow : wide;
    h.E = sh.E, h.parts = sh.parts;
    heavy_blocks = (long)heavy.nslots * sh.parts;
  } else {
    h.la.nslots = 0;
  }
  const long grid = l.blocks + heavy_blocks;
  if (grid <= 0) return MI_OK;
  if (grid > 0x7fffffffL) return MI_ERANGE;
#define MI_STAGED(KQ_)                                                                                                      \
  do {                                                                                                                      \
    MI_HIP_TRY(hipFuncSetAttribute((const void*)spmm_staged_rows_kernel<KQ_>, hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                   (int)narrow.lds));                                                                       \
    hipLaunchKernelGGL(spmm_staged_rows_kernel<KQ_>, dim3((unsigned)grid), dim3(kStageThreads), narrow.lds, s, rowptr, col, \
                       val, B, C, N, (long)ldb, (long)ldc, bias, l, h);                                                     \
  } while (0)
  if (narrow.KQ == 4) MI_STAGED(4);
  else MI_STAGED(8);
#undef MI_STAGED
  return check_launch();
}

int launch_heavy_rows(const int32_t* rowptr, const int32_t* col, const float* val, int32_t M, int32_t N, const float* B,
                      int64_t ldb, float* C, int64_t ldc, const float* bias, LongArg la, hipStream_t s) {
  (void)M;
  if (la.order == nullptr || la.nslots <= 0) return MI_OK;
  return launch_staged_rows(nullptr, LongWs{}, false, la, rowptr, col, val, B, C, N, ldb, ldc, bias, s);
}

int launch_long_rows_staged(int* ws, const LongWs& lw, const int32_t* rowptr, const int32_t* col, const float* val,
                            const float* B, float* C, int32_t N, int64_t ldb, int64_t ldc, const float* bias, bool reset,
                            hipStream_t s) {
  LongArg none = {0x7fffffff, 0, 0, 0, nullptr, nullptr, nullptr, 0};
  return launch_staged_rows(ws, lw, reset, none, rowptr, col, val, B, C, N, ldb, ldc, bias, s);
}

}  // namespace mi
