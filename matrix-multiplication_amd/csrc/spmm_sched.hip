// Row schedules — the inspector's answer to degree-skewed matrices and to matrices whose locality is hidden by their row order
// ("row-length-binned / nnz-balanced block descriptors … skew-robust variant … optional column-blocked
// reordering to raise the cache hit rate").  The reference's kernel descends from merge-spmm (src/naive_sparse_mm.cu:20-21) and
// its inspector builds a restructured image of A once, compacting each block's footprint of B (src/sparse_mm.cu:62-68,137-368);
// here the CSR arrays stay as they are and the inspector builds ONE small thing per matrix: the order in which rows are handed
// to waves.
//
//   order[slot] = row,   rows by DESCENDING length class (classes ≈ 12 % wide: exact below 32 entries, eight per octave above)
//                        and — where that recovers locality the row order hides — by the column their entries centre on
//
// What that buys, with every row's fmaf chain untouched (the same bits as the unscheduled product, for every plan):
//   * longest first — no long row is left for the end of the grid (a row is one wave's serial chain);
//   * rows that share a wave (lane-group kernels: 2 – 16 rows per wave) or a workgroup have about the same length: no lane
//     group idles while its neighbour finishes;
//   * the rows beyond `heavy_len` (a few, holding a large share of the entries) are the first `heavy` slots: the dispatcher
//     launches them apart, a whole workgroup per row (spmm_heavy.hip), beside the launch of the rest (spmm_csr.hip);
//   * LOCALITY: a banded or community-structured matrix whose rows arrive shuffled gathers like a uniformly random one — the
//     waves that run together touch all of B.  Within a length class the rows are put in the order of their median column
//     (a 16-bit column key), so that waves running together gather from one neighbourhood of B again.  Taken only when the
//     inspector MEASURES that it helps: the rows of B a window of 2048 consecutive slots touches must span less than half of
//     what a window of the natural order spans (and the natural order must not be local already).
// Built on the device by stable counting passes over tiles of 2048 slots (tile histogram, scan of the bucket × tile table,
// scatter: tiles keep their order inside a bucket, so a later pass by class keeps the earlier pass's column order); creating a
// schedule reads the class table and two window statistics back (≈ 1.5 KiB) and therefore synchronises the stream —
// inspection time, like cusparse_inspect's checks.
#include <map>
#include <mutex>
#include <new>

#include "spmm_device.h"
#include "spmm_internal.h"

namespace {

constexpr int kClasses = 256;
constexpr int kTileRows = 2048;  // slots per workgroup of a counting pass
constexpr int kWindows = 16;     // windows of kTileRows consecutive slots the locality statistic looks at
constexpr int kHeavyRowsMost = 128;  // the rule's heavy launch takes at most this many rows (set_heavy)
static_assert(kClasses + 1 + 2 * 3 * kWindows < MI_SCHEDULE_HOST_INTS, "the class table and both window statistics fit host_out");

// Length class, larger = longer: exact below 32, then eight classes per octave (2^e … 2^(e+1)) up to 2^31.
__host__ __device__ __forceinline__ int length_class(int len) {
  if (len < 32) return len < 0 ? 0 : len;
  const int e = 31 - __builtin_clz((unsigned)len);  // 5 … 30
  return 32 + (e - 5) * 8 + ((len >> (e - 3)) & 7);  // ≤ 32 + 25·8 + 7 = 239
}
// smallest length of a class (host: turning a length threshold into a slot count)
int class_floor(int c) {
  if (c < 32) return c;
  const int e = 5 + (c - 32) / 8, m = (c - 32) % 8;
  return (1 << e) + (m << (e - 3));
}

enum { kKeyClass = 0, kKeyColumn = 1, kKeyColumnLow = 2 };

// The bucket of a row under a pass's key: kKeyClass — kClasses − 1 − length class (bucket order = descending length);
// kKeyColumn / kKeyColumnLow — the high / low byte of the row's median entry's column in 65536ths of K (an empty row: 0; columns are not assumed sorted — the
// "median" of an unsorted row is just one of its columns, which costs locality, never correctness).
template <int KEY>
__device__ __forceinline__ int row_bucket(const int* __restrict__ rowptr, const int* __restrict__ col, int K, int row) {
  const int s = rowptr[row], e = rowptr[row + 1];
  if constexpr (KEY == kKeyClass) return kClasses - 1 - length_class(e - s);
  if (e <= s || K <= 0) return 0;
  const long c = col[s + ((e - s) >> 1)];
  // a 16-bit column key, c · 65536 / K, in two stable passes: its low byte first (kKeyColumnLow), then its high byte
  long k16 = c < 0 ? 0 : c * (kClasses * kClasses) / K;
  k16 = k16 < kClasses * kClasses ? k16 : kClasses * kClasses - 1;
  return KEY == kKeyColumnLow ? (int)(k16 & (kClasses - 1)) : (int)(k16 >> 8);
}

// A stable counting pass over tiles of kTileRows slots of `in_order` (nullptr: slot i is row i).
// (1) table[bucket][tile] = rows of the tile in the bucket
template <int KEY>
__global__ __launch_bounds__(256) void sched_tile_hist_kernel(const int* __restrict__ rowptr, const int* __restrict__ col, int M,
                                                              int K, const int* __restrict__ in_order, int* __restrict__ table,
                                                              int ntiles) {
  __shared__ int cnt[kClasses];
  cnt[threadIdx.x] = 0;
  __syncthreads();
  const long s0 = (long)blockIdx.x * kTileRows;
#pragma unroll
  for (int i = 0; i < kTileRows / 256; ++i) {
    const long slot = s0 + (long)i * 256 + threadIdx.x;
    if (slot < M) atomicAdd(&cnt[row_bucket<KEY>(rowptr, col, K, in_order ? in_order[slot] : (int)slot)], 1);
  }
  __syncthreads();
  table[(long)threadIdx.x * ntiles + blockIdx.x] = cnt[threadIdx.x];
}

// (2) exclusive scan of the n = kClasses · ntiles table entries in place (bucket-major: all tiles of bucket 0, then bucket 1, …);
// start[b] (may be null) = slots before bucket b, start[kClasses] = total.  One workgroup.
__global__ __launch_bounds__(1024) void sched_table_scan_kernel(int* __restrict__ table, int ntiles, int* __restrict__ start) {
  __shared__ int part[1024];
  const long n = (long)kClasses * ntiles;
  const long per = (n + 1023) / 1024;
  const long lo = (long)threadIdx.x * per < n ? (long)threadIdx.x * per : n;
  const long hi = lo + per < n ? lo + per : n;
  int sum = 0;
  for (long i = lo; i < hi; ++i) sum += table[i];
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {  // inclusive scan of the per-thread sums
    const int add = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
    __syncthreads();
    part[threadIdx.x] += add;
    __syncthreads();
  }
  int run = part[threadIdx.x] - sum;
  for (long i = lo; i < hi; ++i) {
    const int c = table[i];
    table[i] = run;
    if (start != nullptr && i % ntiles == 0) start[i / ntiles] = run;
    run += c;
  }
  if (start != nullptr && threadIdx.x == 1023) start[kClasses] = part[1023];
}

// (3) slot of a row = table[bucket][tile] + its rank among the tile's rows of the bucket (ranks from LDS counters: the order of
// 2048 neighbours inside a bucket is immaterial — no result depends on it — the order of the TILES is kept)
template <int KEY>
__global__ __launch_bounds__(256) void sched_tile_scatter_kernel(const int* __restrict__ rowptr, const int* __restrict__ col, int M,
                                                                 int K, const int* __restrict__ in_order,
                                                                 const int* __restrict__ table, int ntiles,
                                                                 int* __restrict__ out_order) {
  __shared__ int cnt[kClasses];
  constexpr int PER = kTileRows / 256;
  cnt[threadIdx.x] = 0;
  __syncthreads();
  const long s0 = (long)blockIdx.x * kTileRows;
  int row[PER], bucket[PER], rank[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const long slot = s0 + (long)i * 256 + threadIdx.x;
    bucket[i] = -1;
    if (slot < M) {
      row[i] = in_order ? in_order[slot] : (int)slot;
      bucket[i] = row_bucket<KEY>(rowptr, col, K, row[i]);
      rank[i] = atomicAdd(&cnt[bucket[i]], 1);
    }
  }
#pragma unroll
  for (int i = 0; i < PER; ++i)
    if (bucket[i] >= 0) out_order[table[(long)bucket[i] * ntiles + blockIdx.x] + rank[i]] = row[i];
}

// The locality statistic: for kWindows windows of kTileRows consecutive slots (spread evenly over the slots), the FOOTPRINT of
// the window in B — sample up to eight evenly spaced entries of every fourth row, histogram their columns in 256ths of K, and
// count the buckets it takes to hold three quarters of the samples (a uniformly random window: ≈ 192 of 256; a band of ± 1 K
// columns of 1 M: 1 – 2; a community matrix with a tenth of its entries anywhere: its community's bucket — a min / max span
// would call that window as wide as B).  stats[w] = {footprint buckets, mean row span (max − min of the samples), rows sampled}.
__global__ __launch_bounds__(256) void sched_window_stats_kernel(const int* __restrict__ rowptr, const int* __restrict__ col, int M,
                                                                 int K, const int* __restrict__ order, int* __restrict__ stats) {
  __shared__ int hist[kClasses], sorted[kClasses];
  __shared__ unsigned long long s_sum[4];
  __shared__ int s_n[4];
  const int w = blockIdx.x;
  const int win = M < kTileRows ? M : kTileRows;
  const long first = kWindows > 1 ? (long)w * (M - win) / (kWindows - 1) : 0;
  hist[threadIdx.x] = 0;
  __syncthreads();
  int n = 0;
  unsigned long long sum = 0;
  for (int r = 4 * (int)threadIdx.x; r < win; r += 4 * 256) {
    const int row = order ? order[first + r] : (int)(first + r);
    const int s = rowptr[row], e = rowptr[row + 1];
    if (e <= s) continue;
    int lo = 0x7fffffff, hi = -1;
    const int take = e - s < 8 ? e - s : 8;
    for (int j = 0; j < take; ++j) {
      const int c = col[s + (int)((long)j * (e - s) / take)];
      lo = c < lo ? c : lo;
      hi = c > hi ? c : hi;
      const long bq = c < 0 ? 0 : (long)c * kClasses / K;
      atomicAdd(&hist[bq < kClasses ? (int)bq : kClasses - 1], 1);
    }
    sum += (unsigned long long)(hi - lo);
    ++n;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    sum += __shfl_xor(sum, d, 64);
    n += __shfl_xor(n, d, 64);
  }
  if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = sum, s_n[threadIdx.x >> 6] = n;
  __syncthreads();
  // counts in descending order: the rank of this bucket = buckets with a larger count (ties: the lower index first)
  const int mine = hist[threadIdx.x];
  int rank = 0;
  for (int b = 0; b < kClasses; ++b) {
    const int o = hist[b];
    rank += (o > mine || (o == mine && b < (int)threadIdx.x)) ? 1 : 0;
  }
  sorted[rank] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    long total = 0;
    for (int b = 0; b < kClasses; ++b) total += sorted[b];
    long run = 0;
    int need = 0;
    while (need < kClasses && 4 * run < 3 * total) run += sorted[need++];
    for (int i = 1; i < 4; ++i) sum += s_sum[i], n += s_n[i];
    stats[3 * w] = need;
    stats[3 * w + 1] = n > 0 ? (int)(sum / (unsigned long long)n) : 0;
    stats[3 * w + 2] = n;
  }
}

// one stable pass: in_order (or identity) → out_order by KEY; leaves the scanned table in `table` and, for the class pass,
// start[] on the device
template <int KEY>
hipError_t counting_pass(const int32_t* rowptr, const int32_t* col, int32_t M, int32_t K, const int32_t* in_order,
                         int32_t* out_order, int* table, int* start, hipStream_t s) {
  const int ntiles = (int)(((long)M + kTileRows - 1) / kTileRows);
  hipLaunchKernelGGL(sched_tile_hist_kernel<KEY>, dim3((unsigned)ntiles), dim3(256), 0, s, rowptr, col, M, K, in_order, table, ntiles);
  hipLaunchKernelGGL(sched_table_scan_kernel, dim3(1), dim3(1024), 0, s, table, ntiles, start);
  hipLaunchKernelGGL(sched_tile_scatter_kernel<KEY>, dim3((unsigned)ntiles), dim3(256), 0, s, rowptr, col, M, K, in_order, table,
                     ntiles, out_order);
  return hipGetLastError();
}

}  // namespace

struct mi_spmm_schedule {
  mi::RowSchedule rs;
  int64_t nnz;
  int32_t n_width;           // the dense width the heavy length was chosen for
  int32_t start[kClasses + 1];  // host copy: slots before bucket b
  int32_t longest;           // smallest length of the longest non-empty class (a lower bound of the longest row)
  int32_t classes;           // non-empty classes
  bool locality;             // the order also follows the rows' median columns (it measurably recovers locality)
  int32_t span_natural, span_scheduled, span_row;  // ‰ of B: a window's footprint in natural order / in this order; mean row span, ‰ of K (-1: not measured)
};

namespace {

// Rows longer than this get the launch of their own.  A wave walks a row at (gathers in flight) / (memory latency): ≈ 6 GB/s of
// B rows, i.e. a row of L entries takes L · 4N / 6e9 s while the whole product takes ≈ nnz · 4N / 6e12 s on the chip — a row
// beyond nnz / 1000 entries alone outlasts the product.  A quarter of that, not below 128 entries (shorter rows gain nothing
// from more gathers in flight), not below four times the mean length (a heavy row is an OUTLIER: where every row has hundreds of
// entries the waves that run together are alike and nothing waits for one of them) and not above the long-row threshold (those
// rows have a kernel of their own).
int32_t heavy_length_for(int64_t nnz, int32_t rows) {
  int64_t h = nnz / 4096;
  const int64_t mean4 = rows > 0 ? 4 * nnz / rows : 0;
  h = h < mean4 ? mean4 : h;
  return (int32_t)(h < 128 ? 128 : (h > mi::kLongRowThreshold ? mi::kLongRowThreshold : h));
}

// ONE side stream per device for every schedule of the process, created with the first schedule that needs it and never
// destroyed.  A stream per schedule ran out of hardware queues: the runtime deals streams round-robin to a handful of them, and
// with a dozen schedules alive (tools/bench_degree_skew.py keeps the automatic ones of its earlier cases) a new schedule's side
// stream shared its queue with the caller's stream — the launches it was made to run side by side ran one behind the other (reddit-like
// unclipped: 44.8 ms in the full sweep, 39.2 on its own).  Sharing costs unrelated products a false order between their ordinary
// launches, never correctness: every product forks into the stream and joins out of it by its own events.
hipStream_t shared_side_stream(hipError_t* err) {
  static std::mutex m;
  static std::map<int, hipStream_t> streams;
  int dev = 0;
  *err = hipGetDevice(&dev);
  if (*err != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(m);
  auto it = streams.find(dev);
  if (it != streams.end()) return it->second;
  hipStream_t s = nullptr;
  *err = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  if (*err != hipSuccess) return nullptr;
  streams[dev] = s;
  return s;
}

int ensure_side_stream(mi_spmm_schedule* sc) {
  hipError_t e = hipSuccess;
  if (sc->rs.side == nullptr) sc->rs.side = shared_side_stream(&e);
  if (e == hipSuccess && sc->rs.fork == nullptr) e = hipEventCreateWithFlags(&sc->rs.fork, hipEventDisableTiming);
  if (e == hipSuccess && sc->rs.join == nullptr) e = hipEventCreateWithFlags(&sc->rs.join, hipEventDisableTiming);
  return e == hipSuccess ? MI_OK : mi::record_hip_error(e);
}

void set_heavy(mi_spmm_schedule* sc, int32_t heavy_len, bool by_rule) {
  // slots before the first class that holds a row of ≤ heavy_len entries: every row there is longer than heavy_len
  int c = length_class(heavy_len);  // rows of class c may be ≤ heavy_len: not heavy
  // The heavy launch is for the FEW outliers (one workgroup per row and 64 columns, one resident per CU): where the rule's
  // length leaves more than kHeavyRowsMost rows above it, the boundary moves up, class by class, to the longest ones
  // (not beyond the long-row threshold's class).  7 680 rows of mean 482 with a
  // tenth of them beyond nnz / 4096 ran 1.37 × slower with 700 heavy rows than unscheduled (tools/plan_grid.py --pattern degskew).
  if (by_rule) {
    const int c_most = length_class(mi::kLongRowThreshold);
    while (c < c_most && sc->start[kClasses - 1 - c] > kHeavyRowsMost) ++c;
    if (c > length_class(heavy_len)) heavy_len = class_floor(c + 1) - 1;  // the longest length that is not heavy
  }
  sc->rs.heavy_len = heavy_len;
  sc->rs.heavy = sc->start[kClasses - 1 - c];  // buckets 0 … (kClasses − 2 − c) = classes above c
  // Is the order worth its indirection?  It costs the locality of consecutive rows (rowptr reads, C rows written side by side):
  // 4 M rows of 1 … 8 entries ran 11 – 16 % SLOWER scheduled, 2.4 M rows of ≤ 100 entries around a mean of 50 ran 5 % faster
  // (profiles/r06_degree_skew.log).  Active with heavy rows, or with rows that are long enough to be gather-bound (mean ≥ 16)
  // and differ in length.
  // … or when the order recovers locality the row order hides (mi_spmm_schedule_create measured it).
  // "Differ in length" is asked of the ENTRIES, not of one outlier row: at least 2 % of them in rows of ≥ 1.5 × the mean (config
  // C3's Binomial lengths — mean 105, a longest row near 155 among a million — must stay inactive whatever its luckiest row).
  const double mean = sc->rs.rows > 0 ? (double)sc->nnz / (double)sc->rs.rows : 0.0;
  double long_entries = 0.0;
  for (int b = 0; b < kClasses; ++b) {
    const int floor_b = class_floor(kClasses - 1 - b);
    if ((double)floor_b >= 1.5 * mean) long_entries += (double)(sc->start[b + 1] - sc->start[b]) * (double)floor_b;
  }
  sc->rs.active = sc->rs.heavy > 0 || sc->locality || (mean >= 16.0 && long_entries >= 0.02 * (double)sc->nnz);
}

}  // namespace

extern "C" {

size_t mi_spmm_schedule_workspace_bytes(int32_t M) {
  const size_t ntiles = M > 0 ? ((size_t)M + kTileRows - 1) / kTileRows : 1;
  // bucket × tile table, the class starts, a second order (the column pass's), two window statistics
  return (kClasses * ntiles + (kClasses + 1) + (size_t)(M > 0 ? M : 0) + 2 * 3 * kWindows + 16) * sizeof(int);
}

// The build in two halves, so that a caller that must not synchronise (custom_mm's automatic schedules for the plain entry
// points) can let the tables travel behind an event: _begin enqueues everything on the stream — both candidate orders, the class
// table and the window statistics copied to `host_out` (MI_SCHEDULE_HOST_INTS ints; pinned memory for the copy to be
// asynchronous) — and _finish, once that copy has landed, reads the tables on the host, picks the order and creates the object.
int mi_spmm_schedule_begin(const int32_t* rowptr, const int32_t* col, int32_t M, int32_t K, int64_t nnz, int32_t N,
                           int32_t* order, int32_t* order_locality, void* workspace, size_t workspace_bytes,
                           int32_t* host_out, mi_stream_t stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (M < 0 || K < 0 || nnz < 0 || N < 0 || !host_out) return MI_EINVAL;
  if (nnz > 0x7fffffffLL) return MI_ERANGE;
  if (M > 0 && (!rowptr || !order || !workspace)) return MI_EINVAL;
  if (workspace_bytes < mi_spmm_schedule_workspace_bytes(M)) return MI_ENOMEM;
  for (int i = 0; i < MI_SCHEDULE_HOST_INTS; ++i) host_out[i] = 0;
  host_out[MI_SCHEDULE_HOST_INTS - 1] = 0;  // 1: the locality order was built and measured
  if (M == 0) return MI_OK;
  const size_t ntiles = ((size_t)M + kTileRows - 1) / kTileRows;
  int* table = static_cast<int*>(workspace);
  int* start = table + kClasses * ntiles;
  int* order1 = start + kClasses + 1;
  int* stats = order1 + M;  // [2][kWindows][3]
  // a column pass is worth trying on a matrix with columns to look at and enough rows for windows to mean something
  const bool try_locality = order_locality != nullptr && col != nullptr && K > 0 && nnz > 0 && M >= 4 * kTileRows;
  hipError_t e = hipSuccess;
  if (try_locality) {
    hipLaunchKernelGGL(sched_window_stats_kernel, dim3(kWindows), dim3(256), 0, s, rowptr, col, M, K, (const int*)nullptr, stats);
    // least significant key first: the column key's low byte, its high byte, then the length class — every pass keeps the
    // order of the one before inside its buckets (tiles keep their order; a tile of the previous pass's output holds one or a
    // few neighbouring key values, so the arbitrary order inside a tile costs nothing)
    e = counting_pass<kKeyColumnLow>(rowptr, col, M, K, nullptr, order_locality, table, nullptr, s);
    if (e == hipSuccess) e = counting_pass<kKeyColumn>(rowptr, col, M, K, order_locality, order1, table, nullptr, s);
    if (e == hipSuccess) e = counting_pass<kKeyClass>(rowptr, col, M, K, order1, order_locality, table, start, s);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(sched_window_stats_kernel, dim3(kWindows), dim3(256), 0, s, rowptr, col, M, K, (const int*)order_locality,
                         stats + 3 * kWindows);
      e = hipGetLastError();
    }
    if (e == hipSuccess)
      e = hipMemcpyAsync(host_out + kClasses + 1, stats, 2 * 3 * kWindows * sizeof(int), hipMemcpyDeviceToHost, s);
  }
  // by length class alone: neighbours in the row order stay neighbours inside a class
  if (e == hipSuccess) e = counting_pass<kKeyClass>(rowptr, col, M, K, nullptr, order, table, start, s);
  if (e == hipSuccess) e = hipMemcpyAsync(host_out, start, (kClasses + 1) * sizeof(int), hipMemcpyDeviceToHost, s);
  if (e != hipSuccess) return mi::record_hip_error(e);
  host_out[MI_SCHEDULE_HOST_INTS - 1] = try_locality ? 1 : 0;  // (a host-side mark: not part of what the copies write)
  return MI_OK;
}

int mi_spmm_schedule_finish(const int32_t* host_out, int32_t* order, int32_t* order_locality, int32_t M, int32_t K, int64_t nnz,
                            int32_t N, mi_spmm_schedule_t** out) {
  if (!out) return MI_EINVAL;
  *out = nullptr;
  if (!host_out || M < 0 || K < 0 || nnz < 0 || N < 0 || (M > 0 && !order)) return MI_EINVAL;
  mi_spmm_schedule* sc = new (std::nothrow) mi_spmm_schedule();
  if (!sc) return MI_ENOMEM;
  sc->rs = mi::RowSchedule{order, M, 0, 0, true, nullptr, nullptr, nullptr};
  sc->nnz = nnz;
  sc->n_width = N;
  sc->locality = false;
  sc->span_natural = sc->span_scheduled = sc->span_row = -1;
  for (int b = 0; b <= kClasses; ++b) sc->start[b] = M > 0 ? host_out[b] : 0;
  if (M > 0 && host_out[MI_SCHEDULE_HOST_INTS - 1] == 1 && order_locality != nullptr && K > 0) {
    const int32_t* h_stats = host_out + kClasses + 1;
    double nat = 0, neu = 0, row = 0;
    int wn = 0;
    for (int w = 0; w < kWindows; ++w) {
      if (h_stats[3 * w + 2] == 0 && h_stats[3 * (kWindows + w) + 2] == 0) continue;
      nat += h_stats[3 * w], neu += h_stats[3 * (kWindows + w)], row += h_stats[3 * w + 1];
      ++wn;
    }
    if (wn > 0) {
      sc->span_natural = (int32_t)(1000.0 * nat / wn / kClasses);   // footprints, ‰ of B
      sc->span_scheduled = (int32_t)(1000.0 * neu / wn / kClasses);
      sc->span_row = (int32_t)(1000.0 * row / wn / K);
      // the natural order is not local already (three quarters of a window's gathers need > 40 % of B) and the locality order
      // more than halves the footprint
      sc->locality = sc->span_natural > 400 && 2 * sc->span_scheduled < sc->span_natural;
      if (sc->locality) sc->rs.order = order_locality;
    }
  }
  sc->longest = 0, sc->classes = 0;
  for (int b = kClasses - 1; b >= 0; --b) {
    if (sc->start[b + 1] > sc->start[b]) {
      sc->classes++;
      sc->longest = class_floor(kClasses - 1 - b);
    }
  }
  set_heavy(sc, heavy_length_for(nnz, M), true);
  // the side stream of the ordinary launch beside the heavy one, and its fork / join events — for a schedule that has heavy or
  // long rows to run beside the rest; an inactive schedule creates nothing (an extra stream is not free: config C3 read 1.7 %
  // slower with one merely EXISTING in the process, A/B on one box)
  if (sc->rs.active && (sc->rs.heavy > 0 || sc->longest + sc->longest / 8 + 1 > mi::kLongRowThreshold)) {
    const int st = ensure_side_stream(sc);
    if (st != MI_OK) {
      mi_spmm_schedule_destroy(sc);
      return st;
    }
  }
  *out = sc;
  return MI_OK;
}

// begin + synchronise + finish.  `order` must hold 2·M ints when col is given (the second half takes the locality order; the
// schedule points at whichever half it picked).
int mi_spmm_schedule_create(const int32_t* rowptr, const int32_t* col, int32_t M, int32_t K, int64_t nnz, int32_t N,
                            int32_t* order, void* workspace, size_t workspace_bytes, mi_stream_t stream,
                            mi_spmm_schedule_t** out) {
  if (!out) return MI_EINVAL;
  *out = nullptr;
  int32_t host[MI_SCHEDULE_HOST_INTS];
  int32_t* order_locality = (col != nullptr && order != nullptr) ? order + (M > 0 ? M : 0) : nullptr;
  int st = mi_spmm_schedule_begin(rowptr, col, M, K, nnz, N, order, order_locality, workspace, workspace_bytes, host, stream);
  if (st != MI_OK) return st;
  const hipError_t e = hipStreamSynchronize(static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return mi::record_hip_error(e);
  return mi_spmm_schedule_finish(host, order, order_locality, M, K, nnz, N, out);
}

int mi_spmm_schedule_destroy(mi_spmm_schedule_t* sc) {
  if (!sc) return MI_OK;
  if (sc->rs.fork) (void)hipEventDestroy(sc->rs.fork);
  if (sc->rs.join) (void)hipEventDestroy(sc->rs.join);
  delete sc;
  return MI_OK;
}

int mi_spmm_schedule_info(const mi_spmm_schedule_t* sc, int64_t* info) {
  if (!sc || !info) return MI_EINVAL;
  info[0] = sc->rs.rows;
  info[1] = sc->rs.heavy;
  info[2] = sc->rs.heavy_len;
  info[3] = sc->classes;
  info[4] = sc->longest;
  info[5] = (sc->rs.side != nullptr ? 1 : 0) | (sc->rs.active ? 2 : 0) | (sc->locality ? 4 : 0);
  info[6] = sc->nnz;
  info[7] = sc->n_width;
  info[8] = sc->span_natural;
  info[9] = sc->span_scheduled;
  info[10] = sc->span_row;
  info[11] = 0;
  return MI_OK;
}

int mi_spmm_schedule_set_heavy(mi_spmm_schedule_t* sc, int32_t heavy_len, int use_side_stream) {
  if (!sc || heavy_len < 0) return MI_EINVAL;
  set_heavy(sc, heavy_len, false);
  sc->rs.active = true;  // a pinned heavy length is a request to run scheduled
  if (!use_side_stream && sc->rs.side) {  // (tests, A/B: the heavy launch in line, ahead of the rest)
    sc->rs.side = nullptr;  // (the stream is the process's, not the schedule's)
  }
  if (use_side_stream) return ensure_side_stream(sc);
  return MI_OK;
}

int mi_spmm_csr_scheduled_f32(const mi_spmm_schedule_t* sc, int variant, const int32_t* rowptr, const int32_t* col,
                              const float* val, int64_t nnz, int32_t M, int32_t K, int32_t N, const float* B, int64_t ldb,
                              const float* bias, float* C, int64_t ldc, int long_rows, void* workspace,
                              size_t workspace_bytes, mi_stream_t stream) {
  if (!sc) return MI_EINVAL;
  if (long_rows == MI_LONG_ROWS_SPLIT && workspace == nullptr && nnz > mi::kLongRowThreshold) return MI_EINVAL;
  return mi::spmm_dispatch(variant, rowptr, col, val, nnz, 1, M, K, N, B, ldb, 0, C, ldc, 0, bias, workspace, workspace_bytes,
                           static_cast<hipStream_t>(stream), long_rows, &sc->rs);
}

}  // extern "C"
