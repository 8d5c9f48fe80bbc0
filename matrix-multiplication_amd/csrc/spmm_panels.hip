// Column-panel passes of the row-split SpMM (gfx950): K cut into panels whose slice of B a cache level can hold, one launch
// per panel, C carried through memory — the one-wave-per-row panel kernel (N = 256 / 512 / 1024), the lane-group panel
// kernel (every other width) and the device-side locality probe that lets the L2-level passes collapse on banded matrices.
// Contract: include/mi_spmm.h (mi_spmm_csr_f32; MI_SPMM_PANELS_*, MI_SPMM_GROUP_PANELS_*); reference kernel:
// src/naive_sparse_mm.cu:24-101 (one pass, any N).
#include "spmm_device.h"
#include "spmm_internal.h"

namespace {

using mi::LongArg;

__global__ __launch_bounds__(256) void spmm_locality_probe_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                                  int M, long ldb, double b_bytes, int* __restrict__ verdicts,
                                                                  const int* __restrict__ order) {
  __shared__ int s_mn[4], s_mx[4];
  const int w = blockIdx.x;
  const int win = M < kAdaptWindow ? M : kAdaptWindow;
  const long first = kAdaptSlots > 1 ? (long)w * (M - win) / (kAdaptSlots - 1) : 0;
  int mn = 0x7fffffff, mx = -1;
  // every fourth row of the window, two rows per thread, all of a thread's loads of one kind in flight together: the launch is
  // two dependent trips to memory (offsets, then columns) long — ≈ 3 µs ahead of a product of ≥ 50 µs
  constexpr int kRows = 2;
  int s0[kRows], n0[kRows];
#pragma unroll
  for (int i = 0; i < kRows; ++i) {
    const int r = 4 * ((int)threadIdx.x + 256 * i);
    s0[i] = 0, n0[i] = 0;
    if (r < win) {
      const long row = order ? order[first + r] : first + r;  // a schedule: windows of consecutive SLOTS — what runs together
      s0[i] = rowptr[row];
      n0[i] = rowptr[row + 1] - s0[i];
    }
  }
#pragma unroll
  for (int i = 0; i < kRows; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // rows need not ascend: a few entries from either end
      if (j < n0[i]) {
        const int a = col[s0[i] + j], b = col[s0[i] + n0[i] - 1 - j];
        mn = a < mn ? a : mn;
        mn = b < mn ? b : mn;
        mx = a > mx ? a : mx;
        mx = b > mx ? b : mx;
      }
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const int a = __shfl_xor(mn, d, 64), b = __shfl_xor(mx, d, 64);
    mn = a < mn ? a : mn;
    mx = b > mx ? b : mx;
  }
  if ((threadIdx.x & 63) == 0) s_mn[threadIdx.x >> 6] = mn, s_mx[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 4; ++i) {
      mn = s_mn[i] < mn ? s_mn[i] : mn;
      mx = s_mx[i] > mx ? s_mx[i] : mx;
    }
    const double span_bytes = mx >= mn ? ((double)mx - (double)mn + 1.0) * (double)ldb * 4.0 : 0.0;
    verdicts[w] = (span_bytes <= 0.4 * b_bytes && span_bytes <= 128.0 * 1048576.0) ? 1 : 0;
  }
}

// ---------------------------------------------------------------------------
// Column-panel pass (Infinity-Cache blocking), N == 256.  When B is larger than
// the 256 MiB Infinity Cache a uniformly random gather misses it ~(1 - 256MiB/|B|)
// of the time.  Cutting K into P panels whose B slice fits the cache and running
// one launch per panel (all CUs work on the same panel at the same time) turns
// the gathers into cache hits; the price is that C is carried through memory
// between passes ((2P-1) row passes instead of 1) and col is scanned P times.
// A pass handles the nonzeros with c_lo <= col < c_hi in CSR order on top of the
// previous pass's C, so for rows whose columns do not descend (torch CSR, the pinned
// generator) the per-element fmaf chain is exactly the CSR-order chain of the one-pass
// kernel: bit-identical results.
// Rows whose columns DO descend somewhere (legal CSR: the reference's COO→CSR keeps the
// input order inside a row, src/sparse_mm.cu:110-134) would be summed panel by panel, i.e.
// in another order.  Every pass therefore checks, on the col entries it scans anyway
// (one ds_bpermute + compare + ballot per 64 entries), whether the row's columns ascend; the
// verdict is a function of the row alone, so all passes agree without any flag in memory:
// the FIRST pass recomputes such a row from scratch over all its non-zeros in plain CSR order
// (+ bias) and the later passes leave it untouched.
// ---------------------------------------------------------------------------
template <bool FIRST, int T, int U, bool ADAPT = false>
__global__ __launch_bounds__(256) void spmm_wave_row_panel_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ col,
    const float* __restrict__ val, const float* __restrict__ B, float* __restrict__ C,
    int M, long ldb, long ldc, int c_lo, int c_hi, const float* __restrict__ bias, int last_pass, int ctiles,
    unsigned row_blocks, LongArg la) {
  if constexpr (ADAPT) {  // (the Infinity-Cache level — config C3 — runs the ADAPT = false build: nothing added there)
    if (adapt_says_local(la.adapt)) {
      if (!FIRST) return;
      c_lo = 0, c_hi = 0x7fffffff, last_pass = 1;
    }
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned rb = blockIdx.x;
  bool lists = FIRST;  // the first pass over the first column tile lists the long rows it skips
  if (ctiles > 1) {
    // wide N: column tiles of 256·T columns dealt XCD-aware exactly as in spmm_group_kernel, so the
    // (row panel × column tile) slice of B this pass gathers from stays in the XCD's L2
    const unsigned xcd = blockIdx.x % 8, idx = blockIdx.x / 8;
    const int tile = (int)(xcd + 8 * (idx / row_blocks));
    if (tile >= ctiles) return;
    lists = FIRST && tile == 0;
    rb = idx % row_blocks;
    B += (long)tile * (256 * T);
    C += (long)tile * (256 * T);
    if (bias) bias += (long)tile * (256 * T);
  }
  const long slot = (long)rb * 4 + wave;
  if (slot >= (la.order ? la.nslots : M)) return;
  const long row = la.order ? la.order[slot] : slot;  // (wave-uniform) a schedule's slot → row map
  const float* Bl = B + lane * 4;
  float* Cl = C + row * ldc + lane * 4;
  const int start = rowptr[row];
  const int end = rowptr[row + 1];
  if (end - start > la.thresh) {  // left to the listed-rows launch (spmm_heavy.hip) (in every pass)
    if (lists && lane == 0) long_list_append(la, (int)row, end - start);
    return;
  }
  f32x4 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t)
    acc[t] = FIRST ? f32x4{0.f, 0.f, 0.f, 0.f}
                   : __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Cl + t * 256));
  const unsigned width = (unsigned)(c_hi - c_lo);
  bool descends = false;   // wave-uniform: some column of the row is smaller than its predecessor
  int prev_last = -1;      // last column of the previous chunk

  for (int p = start; p < end; p += 64) {
    const int idx = p + lane;
    const int myc = idx < end ? col[idx] : 0x7fffffff;
    int before = __shfl_up(myc, 1, 64);
    if (lane == 0) before = prev_last;
    descends |= __ballot(myc < before) != 0ull;
    if (descends) break;  // no point in gathering on: the first pass redoes the row, the others drop it
    prev_last = __builtin_amdgcn_readlane(myc, 63);  // 0x7fffffff past the end: only the last chunk has such lanes
    const bool in = (unsigned)(myc - c_lo) < width && idx < end;
    const float myv = in ? val[idx] : 0.f;
    unsigned long long mask = __ballot(in);  // this chunk's nonzeros that fall in the panel
    // batches of up to U entries of this panel: every gather of a batch is issued before its first FMA — the last,
    // partial batch of a chunk too (it used to go one entry at a time, each with its latency exposed)
    while (mask) {
      const int n = __builtin_popcountll(mask);  // wave-uniform
      f32x4 x[U][T];
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (u < n) {
          const int i = __builtin_ctzll(mask);
          mask &= mask - 1;
          const int c = __builtin_amdgcn_readlane(myc, i);
          v[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myv), i));
          const float* src = Bl + (long)c * ldb;
#pragma unroll
          for (int t = 0; t < T; ++t) x[u][t] = *reinterpret_cast<const f32x4*>(src + t * 256);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (u < n) {
#pragma unroll
          for (int t = 0; t < T; ++t) acc[t] = fma4(v[u], x[u][t], acc[t]);
        }
      }
    }
  }
  bool add_bias = bias != nullptr && last_pass != 0;
  if (descends) {
    if (!FIRST) return;  // the first pass wrote the whole row
    // plain CSR-order chain over every non-zero of the row, whatever its panel
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int p = start; p < end; p += 64) {
      const int idx = p + lane;
      const int myc = idx < end ? col[idx] : 0;
      const float myv = idx < end ? val[idx] : 0.f;
      const int cnt = (end - p) < 64 ? (end - p) : 64;
      int i = 0;
      for (; i + 4 <= cnt; i += 4) {
        f32x4 x[4][T];
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int c = __builtin_amdgcn_readlane(myc, i + u);
          v[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myv), i + u));
          const float* src = Bl + (long)c * ldb;
#pragma unroll
          for (int t = 0; t < T; ++t) x[u][t] = *reinterpret_cast<const f32x4*>(src + t * 256);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int t = 0; t < T; ++t) acc[t] = fma4(v[u], x[u][t], acc[t]);
      }
      for (; i < cnt; ++i) {
        const int c = __builtin_amdgcn_readlane(myc, i);
        const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myv), i));
        const float* src = Bl + (long)c * ldb;
#pragma unroll
        for (int t = 0; t < T; ++t)
          acc[t] = fma4(v, *reinterpret_cast<const f32x4*>(src + t * 256), acc[t]);
      }
    }
    add_bias = bias != nullptr;
  }
  if (add_bias) {
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] += *reinterpret_cast<const f32x4*>(bias + lane * 4 + t * 256);
  }
#pragma unroll
  for (int t = 0; t < T; ++t)
    __builtin_nontemporal_store(acc[t], reinterpret_cast<f32x4*>(Cl + t * 256));
}

template <int T, int U>
int launch_panels_t(int panels, const int* rowptr, const int* col, const float* val, const float* B,
                    float* C, int M, int K, long ldb, long ldc, const float* bias, LongArg la,
                    hipStream_t s, int ctiles = 1) {
  const long row_blocks = ((la.order ? (long)la.nslots : (long)M) + 3) / 4;
  const long blocks = ctiles > 1 ? 8L * ((ctiles + 7) / 8) * row_blocks : row_blocks;
  if (blocks > 0x7fffffffL) return MI_ERANGE;
  if (blocks == 0) return MI_OK;
  const long kp = ((long)K + panels - 1) / panels;
  for (int q = 0; q < panels; ++q) {
    const int lo = (int)(q * kp);
    const int hi = (int)((q + 1) * kp < K ? (q + 1) * kp : K);
#define MI_PANEL_PASS(FIRST_, ADAPT_)                                                                                          \
  hipLaunchKernelGGL((spmm_wave_row_panel_kernel<FIRST_, T, U, ADAPT_>), dim3((unsigned)blocks), dim3(256), 0, s, rowptr, col, \
                     val, B, C, M, ldb, ldc, lo, hi, bias, q == panels - 1 ? 1 : 0, ctiles, (unsigned)row_blocks, la)
    if (la.adapt != nullptr && ctiles == 1) {
      if (q == 0) MI_PANEL_PASS(true, true);
      else MI_PANEL_PASS(false, true);
    } else {
      if (q == 0) MI_PANEL_PASS(true, false);
      else MI_PANEL_PASS(false, false);
    }
#undef MI_PANEL_PASS
  }
  return mi::check_launch();
}

// ---------------------------------------------------------------------------
// Column-panel passes for the lane-group kernel (round 5): N ≤ 128 with B beyond the Infinity Cache (N = 64 at
// K ≥ 3 M, N = 128 at K ≥ 1.5 M: a 256- or 512-byte row gathered at random from 1 GiB and more).  Same idea as
// spmm_wave_row_panel_kernel — K cut into P panels whose slice of B the cache can hold, one launch per panel, all
// CUs on the same panel at the same time, C carried through memory — with another way of staying exact: a pass
// takes the entries whose RUNNING MAXIMUM of the columns so far (m_i = max_{j ≤ i} col_j, a prefix maximum over the
// row) falls in its panel.  m is non-decreasing, so the passes cut every row into P contiguous index ranges in CSR
// order, whatever the order of its columns: the per-element fmaf chain is the one-pass chain for every legal CSR
// input, with no descent check and no recomputation (for a sorted row m_i = col_i and a pass gathers exactly its own
// panel's rows of B; in an unsorted row an entry may be gathered in a later panel's pass — slower, same bits).
// A pass scans the row's columns from its start (4 bytes per non-zero and pass against 4N + 8 of gathers) and stops
// at the first chunk that ends beyond its panel.
// G = 16 (N ≤ 64) or 32 (N ≤ 128) lanes per row, float4 per lane; grid = ⌈M / (4·64/G)⌉, block = 256.
// ---------------------------------------------------------------------------

template <bool FIRST, int G, int T>
__global__ __launch_bounds__(256) void spmm_group_panel_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ val,
    const float* __restrict__ B, float* __restrict__ C, int M, int N, long ldb, long ldc, int c_lo, int c_hi,
    const float* __restrict__ bias, int last_pass, LongArg la) {
  // T > 1 (G = 64 only): T tiles of 256 columns per lane — the widths between and beyond the one-wave-per-row panel kernel's
  // 256 / 512 / 1024 (N = 160 … 1024, any multiple of 4): the same passes for every N the path takes
  static_assert(T == 1 || G == 64, "column tiles only with a whole wave per row");
  if (la.adapt != nullptr && adapt_says_local(la.adapt)) {  // see spmm_locality_probe_kernel
    if (!FIRST) return;
    c_lo = kIntMin, c_hi = 0x7fffffff, last_pass = 1;
  }
  constexpr int RPW = 64 / G;
  constexpr int UI = T >= 3 ? 2 : 4;  // gathers in flight per group and batch (T float4 each)
  const int lane = threadIdx.x & 63;
  const int gl = lane & (G - 1);
  const int gshift = lane & ~(G - 1);  // first lane of this group inside the wave
  const long slot = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + (lane / G);
  const bool live = slot < (la.order ? la.nslots : M);
  const long row = (la.order && live) ? la.order[slot] : slot;  // a schedule's slot → row map
  int coff[T];
  bool on[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    coff[t] = (t * G + gl) * 4;
    on[t] = coff[t] < N;
  }
  int start = 0, end = 0;
  if (live) {
    start = rowptr[row];
    end = rowptr[row + 1];
  }
  const bool skipped = end - start > la.thresh;  // left to the listed-rows launch (spmm_heavy.hip) (in every pass)
  if (skipped) {
    if (FIRST && gl == 0) long_list_append(la, (int)row, end - start);
    end = start;
  }
  float* dst = C + row * ldc;
  f32x4 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!FIRST && live && !skipped && on[t]) acc[t] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dst + coff[t]));
  }
  int prev_max = kIntMin;  // running maximum of the columns of the chunks behind
  constexpr unsigned long long gmask = G == 64 ? ~0ull : ((1ull << G) - 1ull);
  for (int p = start; p < end; p += G) {  // trip count differs between groups
    const int idx = p + gl;
    const bool there = idx < end;
    const int myc = there ? col[idx] : kIntMin;
    int m = group_prefix_max<G>(myc, gl);
    m = m > prev_max ? m : prev_max;
    prev_max = __shfl(m, G - 1, G);
    const bool below = there && m < c_lo;
    const bool inq = there && m >= c_lo && m < c_hi;
    const int i0 = __builtin_popcountll((__ballot(below) >> gshift) & gmask);       // group-uniform: first entry of this pass
    const int i1 = i0 + __builtin_popcountll((__ballot(inq) >> gshift) & gmask);    // … one past its last
    if (i1 > i0) {
      const float myv = inq ? val[idx] : 0.f;
      mi::static_for<G / UI>([&](auto b_) {
        constexpr int b = UI * decltype(b_)::value;
        if (b >= i0 && b + UI <= i1) {
          f32x4 x[UI][T];
          float v[UI];
          mi::static_for<UI>([&](auto u_) {
            constexpr int u = decltype(u_)::value;
            const int c = mi::group_lane<G, b + u, false>(myc);
            v[u] = mi::group_lane<G, b + u, false>(myv);
            const float* src = B + (long)c * ldb;
#pragma unroll
            for (int t = 0; t < T; ++t)
              if (on[t]) x[u][t] = *reinterpret_cast<const f32x4*>(src + coff[t]);
          });
#pragma unroll
          for (int u = 0; u < UI; ++u)
#pragma unroll
            for (int t = 0; t < T; ++t)
              if (on[t]) acc[t] = fma4(v[u], x[u][t], acc[t]);
        } else if (b + UI > i0 && b < i1) {
          mi::static_for<UI>([&](auto u_) {
            constexpr int u = decltype(u_)::value;
            const int c = mi::group_lane<G, b + u, false>(myc);
            const float v = mi::group_lane<G, b + u, false>(myv);
            if (b + u >= i0 && b + u < i1) {
              const float* src = B + (long)c * ldb;
#pragma unroll
              for (int t = 0; t < T; ++t)
                if (on[t]) acc[t] = fma4(v, *reinterpret_cast<const f32x4*>(src + coff[t]), acc[t]);
            }
          });
        }
      });
    }
    if (prev_max >= c_hi) break;  // every later entry belongs to a later pass
  }
  if (live && !skipped) {
#pragma unroll
    for (int t = 0; t < T; ++t)
      if (on[t]) {
        if (bias && last_pass) acc[t] += *reinterpret_cast<const f32x4*>(bias + coff[t]);
        __builtin_nontemporal_store(acc[t], reinterpret_cast<f32x4*>(dst + coff[t]));
      }
  }
}

template <int G, int T>
int launch_group_panels_t(int panels, const int* rowptr, const int* col, const float* val, const float* B, float* C,
                          int M, int K, int N, long ldb, long ldc, const float* bias, LongArg la, hipStream_t s) {
  constexpr int rows_per_block = 4 * (64 / G);
  const long blocks = ((la.order ? (long)la.nslots : (long)M) + rows_per_block - 1) / rows_per_block;
  if (blocks > 0x7fffffffL) return MI_ERANGE;
  if (blocks == 0) return MI_OK;
  const long kp = ((long)K + panels - 1) / panels;
  for (int q = 0; q < panels; ++q) {
    const int lo = (int)(q * kp);
    // the last pass takes whatever is left (columns ≥ K of a lying matrix included: every entry is summed exactly once)
    const int hi = q == panels - 1 ? 0x7fffffff : (int)((q + 1) * kp);
    if (q == 0)
      hipLaunchKernelGGL((spmm_group_panel_kernel<true, G, T>), dim3((unsigned)blocks), dim3(256), 0, s, rowptr, col, val, B, C,
                         M, N, ldb, ldc, kIntMin, hi, bias, q == panels - 1 ? 1 : 0, la);  // (first pass: from the smallest int, as lo is unused)
    else
      hipLaunchKernelGGL((spmm_group_panel_kernel<false, G, T>), dim3((unsigned)blocks), dim3(256), 0, s, rowptr, col, val, B, C,
                         M, N, ldb, ldc, lo, hi, bias, q == panels - 1 ? 1 : 0, la);
  }
  return mi::check_launch();
}

}  // namespace

namespace mi {

int launch_locality_probe(const int32_t* rowptr, const int32_t* col, int32_t M, int64_t ldb, double b_bytes, int* verdicts,
                          const int32_t* order, hipStream_t s) {
  hipLaunchKernelGGL(spmm_locality_probe_kernel, dim3(kAdaptSlots), dim3(256), 0, s, rowptr, col, M, (long)ldb, b_bytes, verdicts,
                     order);
  return check_launch();
}

int launch_panels(int panels, const int* rowptr, const int* col, const float* val, const float* B,
                  float* C, int M, int K, int N, long ldb, long ldc, const float* bias, LongArg la,
                  hipStream_t s) {
  if (N == 256) return launch_panels_t<1, 8>(panels, rowptr, col, val, B, C, M, K, ldb, ldc, bias, la, s);
  if (N == 512) return launch_panels_t<2, 4>(panels, rowptr, col, val, B, C, M, K, ldb, ldc, bias, la, s);
  return launch_panels_t<4, 2>(panels, rowptr, col, val, B, C, M, K, ldb, ldc, bias, la, s);
}

int launch_coltile_panels(int panels, const int* rowptr, const int* col, const float* val, const float* B, float* C, int M,
                          int K, int N, long ldb, long ldc, const float* bias, LongArg la, hipStream_t s) {
  return launch_panels_t<1, 8>(panels, rowptr, col, val, B, C, M, K, ldb, ldc, bias, la, s, N / 256);
}

int group_panel_count(int variant) {
  static const int kCount[] = {2, 3, 4, 6, 8};
  return kCount[variant - MI_SPMM_GROUP_PANELS_2];
}

int launch_group_panels(int panels, const int* rowptr, const int* col, const float* val, const float* B, float* C, int M,
                        int K, int N, long ldb, long ldc, const float* bias, LongArg la, hipStream_t s) {
  if (N <= 64) return launch_group_panels_t<16, 1>(panels, rowptr, col, val, B, C, M, K, N, ldb, ldc, bias, la, s);
  if (N <= 128) return launch_group_panels_t<32, 1>(panels, rowptr, col, val, B, C, M, K, N, ldb, ldc, bias, la, s);
  if (N <= 256) return launch_group_panels_t<64, 1>(panels, rowptr, col, val, B, C, M, K, N, ldb, ldc, bias, la, s);
  if (N <= 512) return launch_group_panels_t<64, 2>(panels, rowptr, col, val, B, C, M, K, N, ldb, ldc, bias, la, s);
  if (N <= 768) return launch_group_panels_t<64, 3>(panels, rowptr, col, val, B, C, M, K, N, ldb, ldc, bias, la, s);
  return launch_group_panels_t<64, 4>(panels, rowptr, col, val, B, C, M, K, N, ldb, ldc, bias, la, s);
}

}  // namespace mi
