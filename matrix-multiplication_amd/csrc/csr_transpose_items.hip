// CSR transpose of a batch of SMALL items, one workgroup per item inside its LDS (mi_csr_transpose_batched_f32's plan for
// pruned-attention-sized items; contract: include/mi_spmm.h; reference: the backward of a batched CSR operand does not exist
// there, matmuls.py:245-256).  Split from csr_transpose.hip (the general and one-sweep plans) in round 6: a unit of its own.
#include "mi_common.h"
#include "csr_transpose_internal.h"

namespace {

// ---------------------------------------------------------------------------------------------
// Batches of SMALL items (round 5): one workgroup transposes one item inside its LDS — pruned attention hands over a new
// pattern on every step (384 items of 512 × 512 at 10 % kept: 10 M entries), and the general plan above, built for 10⁸
// entries of one matrix, took 0.27 ms for them (tools/probes/attn_fresh_pieces.py) — more than the three products of the
// step together.  Here: the item's rows are dealt to the WAVES waves in contiguous blocks; (1) every wave counts its block's
// entries per column into its own row of an LDS table [WAVES][K] (ds_add); (2) one pass over the table turns it into
// cursors — column k of wave w starts at (entries of columns < k) + (entries of column k in waves < w), the column starts
// go out as the item's transposed offsets; (3) every wave walks its rows IN ORDER, a row's ≤ 64 entries per instruction:
// position = cursor[w][col]++ (an LDS read and write — the columns of one instruction are distinct when the row ascends
// strictly, which is checked on the spot; a chunk that does not — unsorted rows, duplicate columns — takes its positions
// one lane after the other, in lane order).  Entries of a column therefore land by ascending row, ties in CSR order: the
// stable order of the general plan and of oracle_csr_transpose.
// The placed entries (row, value) do NOT go to memory one by one: a first version stored them straight to t_col / t_val —
// 4-byte stores, 64 different lines per instruction, every 128-byte line completed by 32 stores spread over the whole sweep,
// with 384 items' 80 MB of half-written lines thrashing the 4 MiB L2s: 0.24 ms, no faster than the general plan.  They
// are STAGED in LDS and leave as whole lines: step (3) runs once per column PASS — the columns are cut into passes whose
// entries fit the staging area (`cap` entries of 8 bytes), a pass places only its own columns' entries at
// (position − the pass's first position) in the staging area, and after a barrier the workgroup copies the pass's
// contiguous piece of t_col / t_val out with coalesced 16-byte stores.  A pass re-reads the item's col / val from the
// L2s (they were read by step 1); two passes at 10 % kept.  A pass that does not fit its share of the staging area
// (skewed columns) falls back to the direct stores for its entries: slower, same result.
// Loads travel eight rows (or eight 64-entry chunks) at a time.  No workspace, no inter-workgroup hand-off.
// ---------------------------------------------------------------------------------------------
// lane l ← lane l − 1 (lane 0 keeps its own value): a DPP wave shift — no trip through the LDS crossbar, which the placement
// loop below keeps busy enough
__device__ __forceinline__ int wave_shr1(int x) { return __builtin_amdgcn_update_dpp(x, x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false); }

#ifdef MI_TR_ITEM_TIMING  // developer probe (tools/probes/tr_item_timing.py): cycles per phase, summed over workgroups (thread 0)
__device__ unsigned long long g_tr_item_stamp[8];
#define TR_ITEM_STAMP(k)                                                                   \
  do {                                                                                     \
    const unsigned long long now_ = __builtin_readcyclecounter();                          \
    if (threadIdx.x == 0) atomicAdd(&g_tr_item_stamp[k], now_ - stamp_);                   \
    stamp_ = now_;                                                                         \
  } while (0)
#else
#define TR_ITEM_STAMP(k) do {} while (0)
#endif

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void tr_item_lds_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                                 const float* __restrict__ val, int M, int K,
                                                                 int* __restrict__ t_rowptr, int* __restrict__ t_col,
                                                                 float* __restrict__ t_val, int cap, int nseg, int nnz_total) {
  extern __shared__ __attribute__((aligned(16))) int tr_lds[];
  // [cap] staged rows, [cap] staged values (both 16-byte aligned: cap % 4 == 0), [WAVES][K] counters → cursors,
  // [K + 1] column starts of the item (relative), WAVES + 1 ints of scan scratch
  constexpr int T = WAVES * 64;
  int* st_row = tr_lds;
  float* st_val = reinterpret_cast<float*>(tr_lds + cap);
  int* cnt = tr_lds + 2 * (long)cap;
  int* starts = cnt + (long)WAVES * K;
  int* scratch = starts + K + 1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // grid = items × nseg: workgroup (item, seg) counts the whole item (steps 1, 2 — it needs every column's start) and
  // places the seg-th column pass; the last one also takes the passes beyond nseg, should the item need more than the
  // launcher sized for (a larger or skewed item)
  const long item = blockIdx.x / (unsigned)nseg;
  const int seg = (int)(blockIdx.x % (unsigned)nseg);
  const int* rp = rowptr + item * ((long)M + 1);
  const int rows_per_wave = (M + WAVES - 1) / WAVES;
  const int r0 = wave * rows_per_wave < M ? wave * rows_per_wave : M;
  const int r1 = r0 + rows_per_wave < M ? r0 + rows_per_wave : M;
  const int base = rp[0];
#ifdef MI_TR_ITEM_TIMING
  unsigned long long stamp_ = __builtin_readcyclecounter();
#endif
  // The kernel is a chain of dependent trips to memory (row bounds → entries → … ) with the CU to itself, so the chain is
  // kept short: the wave's row bounds arrive as ONE vector load, the counting pass's entries and the first placement
  // group's entries are requested together right behind it.
  // (an item without entries never USES what it loads — every `has` is false — but the loads are issued: the index must lie
  // inside the caller's arrays.  For an empty item at the END of the batch `base` is nnz_total, one past them: clamp to the
  // last entry of the whole batch — nnz_total > 0 here, the launcher returns early on an empty batch)
  const int last_e = rp[M] > base ? rp[M] - 1 : (base < nnz_total ? base : nnz_total - 1);
  const bool rp_in_reg = r1 - r0 <= 63;
  const int rpv = rp_in_reg ? rp[r0 + lane <= r1 ? r0 + lane : r1] : 0;
  auto bound = [&](int row) {  // rp[min(row, r1)], wave-uniform
    const int rr = row < r1 ? row : r1;
    return rp_in_reg ? __builtin_amdgcn_readlane(rpv, rr - r0) : rp[rr];
  };
  struct Group {
    int b[9];
    int c[8];
    float v[8];
  };
  auto fetch = [&](int r, Group& gq) {
#pragma unroll
    for (int u = 0; u < 9; ++u) gq.b[u] = bound(r + u);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      // UNCONDITIONAL loads on an index clamped to the item's entries: a predicated load puts a branch round it, and the
      // compiler then waits for EVERY outstanding load at the join — the next group's too, which were to stay in flight
      const int idx = gq.b[u] + lane;
      const int at = idx < last_e ? idx : last_e;
      const int cl = col[at];
      const float vl = val[at];
      const bool has = idx < gq.b[u + 1];
      gq.c[u] = has ? cl : -1;
      gq.v[u] = has ? vl : 0.f;
    }
  };
  Group g0, g1;
  bool g0_ready = false;
  if (r0 < r1) {
    fetch(r0, g0);
    g0_ready = true;
  }
  for (int i = tid; i < WAVES * K; i += T) cnt[i] = 0;
  __syncthreads();
  TR_ITEM_STAMP(0);  // bounds + first fetch issued, table zeroed
  int* mine = cnt + (long)wave * K;
  {  // (1) the wave's entries are contiguous: [rp[r0], rp[r1])
    const int e0 = bound(r0), e1 = bound(r1);
    for (int p = e0; p < e1; p += 8 * 64) {
      int c[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = p + u * 64 + lane;
        const int cl = col[idx < last_e ? idx : last_e];
        c[u] = idx < e1 ? cl : -1;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if ((unsigned)c[u] < (unsigned)K) atomicAdd(&mine[c[u]], 1);  // (a column outside [0, K) is dropped here and below)
    }
  }
  __syncthreads();
  TR_ITEM_STAMP(1);  // counted
  {  // (2) counters → cursors (positions relative to the item); a thread owns a contiguous range of columns
    const int cpt = (K + T - 1) / T;
    const int k_lo = tid * cpt < K ? tid * cpt : K, k_hi = k_lo + cpt < K ? k_lo + cpt : K;
    int sum = 0;
    for (int k = k_lo; k < k_hi; ++k)
      for (int w = 0; w < WAVES; ++w) sum += cnt[(long)w * K + k];
    int incl = sum;  // inclusive scan of `sum` over the wave, then over the workgroup
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int y = __shfl_up(incl, d, 64);
      if (lane >= d) incl += y;
    }
    if (lane == 63) scratch[wave] = incl;
    __syncthreads();
    if (tid == 0) {
      int run = 0;
      for (int w = 0; w < WAVES; ++w) {
        const int x = scratch[w];
        scratch[w] = run;
        run += x;
      }
      scratch[WAVES] = run;  // entries of the item with a column in range
    }
    __syncthreads();
    int run = scratch[wave] + incl - sum;  // entries of the columns before k_lo
    int* t_rp = t_rowptr + item * ((long)K + 1);
    for (int k = k_lo; k < k_hi; ++k) {
      if (seg == 0) t_rp[k] = base + run;
      starts[k] = run;
      for (int w = 0; w < WAVES; ++w) {
        const int x = cnt[(long)w * K + k];
        cnt[(long)w * K + k] = run;
        run += x;
      }
    }
    if (tid == T - 1) {
      if (seg == 0) t_rp[K] = rp[M];  // (= base + every entry of the item when all columns are in range)
      starts[K] = scratch[WAVES];
    }
  }
  __syncthreads();
  TR_ITEM_STAMP(2);  // cursors
  // (3) column passes: [ka, kb) = as many columns from ka on as fit the staging area (at least one); this workgroup
  // places pass number `seg` (and, if it is the last workgroup of the item, every pass behind it)
  int ka = 0, pass = 0;
  while (ka < K) {  // workgroup-uniform
    const int pa = starts[ka];
    int lo = ka + 1, hi = K;  // the last kb in (ka, K] with starts[kb] − pa ≤ cap (kb = ka + 1 even if that column alone is too long)
    while (lo < hi) {
      const int mid = lo + ((hi - lo + 1) >> 1);
      if (starts[mid] - pa <= cap - 4) lo = mid; else hi = mid - 1;  // (− 4: the piece may start up to 3 entries into the area)
    }
    const int kb = lo;
    const int pb = starts[kb];
    const bool take = pass == seg || (seg == nseg - 1 && pass > seg);
    // the staged piece starts `sh` entries into the staging area, sh = the destination's offset from a 16-byte boundary: an
    // entry is then 16-byte aligned in LDS exactly when it is in memory, and the copy-out moves whole aligned quads
    const int sh = (int)((((unsigned long long)(t_col + base + pa)) >> 2) & 3);
    const bool staged = pb - pa <= cap - 4;  // false: one column longer than the staging area — its entries go out directly
    // (the pass body is compiled twice, for staged and for direct output: with one body and a run-time choice of the
    // destination hipcc merges the two stores of an entry into ONE flat store through a selected pointer — and a flat
    // store into LDS is several times slower than a ds_write)
    auto run_pass = [&](auto staged_c) {
      constexpr bool kStaged = decltype(staged_c)::value;
      auto emit = [&](bool ok, int pos, int row, float vv) {
        if (!ok) return;
        if constexpr (kStaged) {
          st_row[pos - pa + sh] = row;
          st_val[pos - pa + sh] = vv;
        } else {
          t_col[base + pos] = row;
          t_val[base + pos] = vv;
        }
      };
      // one ≤ 64-entry chunk of a row (prev_last: the last column of the row's previous chunk, −1 at its start)
      auto place = [&](int cc, float vv, bool has, int n, int row, int prev_last) {
        const bool ok = has && cc >= ka && cc < kb;  // this pass's columns
        int before = __shfl_up(cc, 1, 64);
        if (lane == 0) before = prev_last;
        int pos = 0;
        if (__ballot(has && cc <= before) == 0ull) {  // strictly ascending: the lanes' cursors are distinct words
          if (ok) {
            pos = mine[cc];
            mine[cc] = pos + 1;
          }
        } else {  // lane order by hand (wave-uniform loop; rare)
          for (int i = 0; i < n; ++i) {
            const int ci = __builtin_amdgcn_readlane(cc, i);
            if (ci < ka || ci >= kb) continue;
            const int pi = mine[ci];  // every lane reads the same word
            if (lane == i) {
              pos = pi;
              mine[ci] = pi + 1;
            }
          }
        }
        emit(ok, pos, row, vv);
      };
      // Rows in order, eight rows per group, and the NEXT group's first chunks already in flight while a group is placed
      // (two register sets, alternating): a pass is otherwise a chain of exposed memory latencies — with the staging area
      // a workgroup has its CU to itself, so nothing else hides them.  The wave's row bounds sit in one register (lane j:
      // rp[r0 + j]) where its block has ≤ 63 rows, so a group's bounds cost no memory access either.
      auto settle = [&](int r, const Group& gq) {
        const int* b = gq.b;
        // Fast form: every row of the group fits one chunk and ascends strictly — straight-line code for the eight rows'
        // cursor updates (the LDS executes a wave's instructions in order, so a column that two rows share gets its
        // positions in row order; inside one instruction the columns are distinct).
        bool fast = true;
        unsigned long long bad = 0ull;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          fast = fast && (b[u + 1] - b[u] <= 64);
          const int before = wave_shr1(gq.c[u]);
          bad |= __ballot(gq.c[u] >= 0 && lane > 0 && gq.c[u] <= before);
        }
        if (fast && bad == 0ull) {
          int pos[8];
          bool ok[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            ok[u] = gq.c[u] >= ka && gq.c[u] < kb;  // (absent lanes carry −1)
            pos[u] = 0;
            // a plain read and a plain write (the lanes' words are distinct; the next row's read follows this row's write in
            // the wave's LDS order): a RETURNING LDS atomic measured ≈ 1 lane per cycle for the whole CU — 64 cycles per
            // instruction, 33 k cycles per pass and workgroup, four fifths of the kernel (tools/probes/tr_item_timing.py)
            if (ok[u]) {
              pos[u] = mine[gq.c[u]];
              mine[gq.c[u]] = pos[u] + 1;
            }
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) emit(ok[u], pos[u], r + u, gq.v[u]);
        } else {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            int prev_last = -1;
            int cc = gq.c[u];
            float vv = gq.v[u];
            for (int p = b[u]; p < b[u + 1]; p += 64) {
              const int idx = p + lane;
              const bool hh = idx < b[u + 1];
              if (p > b[u]) {
                cc = hh ? col[idx] : -1;
                vv = hh ? val[idx] : 0.f;
              }
              place(cc, vv, hh, b[u + 1] - p < 64 ? b[u + 1] - p : 64, r + u, prev_last);
              prev_last = __builtin_amdgcn_readlane(cc, 63);
            }
          }
        }
      };
      if (r0 < r1 && !g0_ready) fetch(r0, g0);  // (a later pass of the same workgroup: the first group again)
      g0_ready = false;
      // (the fetches are UNCONDITIONAL — rows beyond the block read as empty, their loads are clamped: a fetch under a
      // branch leaves the number of loads in flight unknown at the join and the compiler then drains them all, vmcnt(0),
      // before the older group is used — which is exactly the overlap this loop exists for)
      for (int r = r0; r < r1; r += 16) {
        fetch(r + 8, g1);
        settle(r, g0);
        fetch(r + 16, g0);
        settle(r + 8, g1);
      }
    };
    if (take && pb > pa) {
      if (staged) run_pass(std::true_type{});
      else run_pass(std::false_type{});
    }
    if (take && staged && pb > pa) {
      __syncthreads();  // the pass's entries are all staged
      TR_ITEM_STAMP(3);  // placed
      // copy out: entries [pa, pb) of the item → t_col / t_val [base + pa, base + pb); 16-byte stores between aligned ends
      const int n = pb - pa;
      int* dc = t_col + base + pa - sh;    // 16-byte aligned; entry j of the staging area ↔ dc[j], j in [sh, sh + n)
      float* dv = t_val + base + pa - sh;  // aligned too when t_val shares t_col's alignment (two arrays at the same offset)
      const bool vec = ((((unsigned long long)dc) ^ ((unsigned long long)dv)) & 15ull) == 0;
      if (vec) {
        const int q0 = sh ? 1 : 0, q1 = (sh + n) >> 2;  // whole quads [q0, q1); entries before / behind them one by one
        for (int i = q0 + tid; i < q1; i += T) {
          *reinterpret_cast<int4*>(dc + 4 * i) = *reinterpret_cast<const int4*>(st_row + 4 * i);
          *reinterpret_cast<float4*>(dv + 4 * i) = *reinterpret_cast<const float4*>(st_val + 4 * i);
        }
        const int head_end = q0 * 4 < sh + n ? q0 * 4 : sh + n;  // [sh, head_end)
        if (tid < head_end - sh) {
          dc[sh + tid] = st_row[sh + tid];
          dv[sh + tid] = st_val[sh + tid];
        }
        const int tail0 = q1 * 4 > head_end ? q1 * 4 : head_end;  // [tail0, sh + n)
        if (tid < sh + n - tail0) {
          dc[tail0 + tid] = st_row[tail0 + tid];
          dv[tail0 + tid] = st_val[tail0 + tid];
        }
      } else {
        for (int i = sh + tid; i < sh + n; i += T) {
          dc[i] = st_row[i];
          dv[i] = st_val[i];
        }
      }
      __syncthreads();  // the staging area is free for the next pass
      TR_ITEM_STAMP(4);  // copied out
    }
    ka = kb;
    ++pass;
  }
}

// Does the LDS plan take this batch?  Items small enough for one workgroup's table, and enough of them (or little enough
// work) that one workgroup per item is not a serial tail.  Returns the number of waves (16, 8 or 4) or 0.
size_t tr_item_fixed_bytes(int waves, int32_t K) { return ((size_t)waves * K + K + 1 + waves + 1) * sizeof(int); }

int tr_item_lds_waves_impl(int64_t nnz, int32_t batch, int32_t M, int32_t K) {
  if (batch <= 0 || M <= 0 || K <= 0 || nnz <= 0) return 0;
  if (nnz / batch > 262144 || (batch < 64 && nnz > 131072)) return 0;
  for (int waves : {16, 8, 4})
    if (tr_item_fixed_bytes(waves, K) <= 48u * 1024) {  // table + column starts ≤ 48 KiB, the rest stages
      // every pass re-reads the item and every workgroup of an item re-counts it: beyond four passes the general plan is
      // ahead (96 items of 1024² at 25 % kept — 19 passes — took 6.5 ms here)
      const long room = (150L * 1024 - (long)tr_item_fixed_bytes(waves, K)) / 8;
      return (nnz + batch - 1) / batch + 64 <= 4 * room ? waves : 0;
    }
  return 0;
}

int launch_tr_item_lds_impl(int waves, int64_t nnz, const int32_t* rowptr, const int32_t* col, const float* val, int32_t batch,
                       int32_t M, int32_t K, int32_t* t_rowptr, int32_t* t_col, float* t_val, hipStream_t s) {
  // staging area: the whole (average) item where that fits 78 KiB per workgroup beside the table — two workgroups per CU,
  // one pass; else 150 KiB per workgroup and as few passes as that allows (an item larger than the average just takes
  // one pass more: the kernel cuts its passes from the item's own column starts)
  const long per_item = (nnz + batch - 1) / batch;
  const size_t fixed = tr_item_fixed_bytes(waves, K);
  long cap = per_item + 64, passes = 1;
  if ((long)fixed + cap * 8 > 78L * 1024) {
    const long room = (150L * 1024 - (long)fixed) / 8;
    passes = (cap + room - 1) / room;
    cap = (per_item + passes - 1) / passes + 64;
    if (cap > room) cap = room;
  }
  cap = (cap + 3) / 4 * 4;
  const size_t lds = fixed + (size_t)cap * 8;
  // one workgroup per (item, pass): a pass re-counts the item (cheap) but the passes of an item run side by side, and the
  // grid fills the CUs' rounds more evenly (384 items in 2 passes: 768 workgroups = 3 whole rounds of 256)
  const int nseg = (int)(passes < 16 ? passes : 16);
  if ((long)batch * nseg > 0x7fffffffL) return MI_ERANGE;
#define MI_TR_ITEM(W_)                                                                                                       \
  do {                                                                                                                       \
    auto k = tr_item_lds_kernel<W_>;                                                                                         \
    if (lds > 64 * 1024) MI_HIP_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL(k, dim3((unsigned)(batch * nseg)), dim3(W_ * 64), lds, s, rowptr, col, val, M, K, t_rowptr, t_col, t_val, \
                       (int)cap, nseg, (int)nnz);                                                                            \
  } while (0)
  if (waves == 16) MI_TR_ITEM(16);
  else if (waves == 8) MI_TR_ITEM(8);
  else MI_TR_ITEM(4);
#undef MI_TR_ITEM
  return mi::check_launch();
}

}  // namespace

namespace mi {
int tr_item_lds_waves(int64_t nnz, int32_t batch, int32_t M, int32_t K) { return ::tr_item_lds_waves_impl(nnz, batch, M, K); }
int launch_tr_item_lds(int waves, int64_t nnz, const int32_t* rowptr, const int32_t* col, const float* val, int32_t batch,
                       int32_t M, int32_t K, int32_t* t_rowptr, int32_t* t_col, float* t_val, hipStream_t s) {
  return ::launch_tr_item_lds_impl(waves, nnz, rowptr, col, val, batch, M, K, t_rowptr, t_col, t_val, s);
}
}  // namespace mi
