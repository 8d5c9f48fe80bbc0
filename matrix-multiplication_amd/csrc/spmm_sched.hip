// Row schedules — the inspector's answer to degree-skewed matrices ("row-length-binned / nnz-balanced block
// descriptors … skew-robust variant").  The reference's kernel descends from merge-spmm (src/naive_sparse_mm.cu:20-21) and its
// inspector builds a restructured image of A once (src/sparse_mm.cu:137-368); here the CSR arrays stay as they are and the
// inspector builds ONE small thing per matrix: the order in which rows are handed to waves.
//
//   order[slot] = row,   rows by DESCENDING length class (classes ≈ 12 % wide: exact below 32 entries, eight per octave above)
//
// What that buys, with every row's fmaf chain untouched (the same bits as the unscheduled product, for every plan):
//   * longest first — no long row is left for the end of the grid (a row is one wave's serial chain);
//   * rows that share a wave (lane-group kernels: 2 – 16 rows per wave) or a workgroup have about the same length: no lane
//     group idles while its neighbour finishes;
//   * the rows beyond `heavy_len` (a few, holding a large share of the entries) are the first `heavy` slots: the dispatcher
//     launches them apart with more gathers in flight per row, on the schedule's side stream beside the rest (spmm_csr.hip).
// Built on the device by three small launches (class histogram, scan, scatter); creating a schedule reads 1 KiB back (the class
// table) and therefore synchronises the stream — inspection time, like cusparse_inspect's checks.
#include <new>

#include "spmm_device.h"
#include "spmm_internal.h"

namespace {

constexpr int kClasses = 256;
constexpr int kTileRows = 2048;  // rows per workgroup of the scatter pass

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

// bucket b = kClasses − 1 − class: bucket order = descending length
__global__ __launch_bounds__(256) void sched_hist_kernel(const int* __restrict__ rowptr, int M, int* __restrict__ hist) {
  __shared__ int h[kClasses];
  h[threadIdx.x] = 0;
  __syncthreads();
  for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < M; r += (long)gridDim.x * blockDim.x)
    atomicAdd(&h[kClasses - 1 - length_class(rowptr[r + 1] - rowptr[r])], 1);
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

// start[b] = rows in buckets before b (start[kClasses] = M); cursor = a working copy for the scatter pass
__global__ __launch_bounds__(256) void sched_scan_kernel(const int* __restrict__ hist, int* __restrict__ start,
                                                         int* __restrict__ cursor) {
  __shared__ int s[kClasses];
  s[threadIdx.x] = hist[threadIdx.x];
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int b = 0; b < kClasses; ++b) {
      const int n = s[b];
      s[b] = run;
      run += n;
    }
    start[kClasses] = run;
  }
  __syncthreads();
  start[threadIdx.x] = s[threadIdx.x];
  cursor[threadIdx.x] = s[threadIdx.x];
}

// A tile of kTileRows consecutive rows per workgroup: ranks inside the tile from LDS counters, one global reservation per
// (tile, bucket).  Within a bucket the tiles land in arrival order and a tile's rows in LDS-atomic order: neighbours stay
// neighbours (locality of rowptr reads and C writes), the exact order is immaterial — no result depends on it.
__global__ __launch_bounds__(256) void sched_scatter_kernel(const int* __restrict__ rowptr, int M, int* __restrict__ cursor,
                                                            int* __restrict__ order) {
  __shared__ int cnt[kClasses], base[kClasses];
  constexpr int PER = kTileRows / 256;
  cnt[threadIdx.x] = 0;
  __syncthreads();
  const long r0 = (long)blockIdx.x * kTileRows;
  int bucket[PER], rank[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const long r = r0 + (long)i * 256 + threadIdx.x;
    bucket[i] = -1;
    if (r < M) {
      bucket[i] = kClasses - 1 - length_class(rowptr[r + 1] - rowptr[r]);
      rank[i] = atomicAdd(&cnt[bucket[i]], 1);
    }
  }
  __syncthreads();
  base[threadIdx.x] = cnt[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], cnt[threadIdx.x]) : 0;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PER; ++i)
    if (bucket[i] >= 0) order[base[bucket[i]] + rank[i]] = (int)(r0 + (long)i * 256 + threadIdx.x);
}

}  // namespace

struct mi_spmm_schedule {
  mi::RowSchedule rs;
  int64_t nnz;
  int32_t n_width;           // the dense width the heavy length was chosen for
  int32_t start[kClasses + 1];  // host copy: slots before bucket b
  int32_t longest;           // smallest length of the longest non-empty class (a lower bound of the longest row)
  int32_t classes;           // non-empty classes
};

namespace {

// Rows longer than this get the launch of their own.  A wave walks a row at (gathers in flight) / (memory latency): ≈ 6 GB/s of
// B rows, i.e. a row of L entries takes L · 4N / 6e9 s while the whole product takes ≈ nnz · 4N / 6e12 s on the chip — a row
// beyond nnz / 1000 entries alone outlasts the product.  A quarter of that, not below 128 entries (shorter rows gain nothing
// from more gathers in flight) and not above the long-row threshold (those rows have a kernel of their own).
int32_t heavy_length_for(int64_t nnz) {
  const int64_t h = nnz / 4096;
  return (int32_t)(h < 128 ? 128 : (h > mi::kLongRowThreshold ? mi::kLongRowThreshold : h));
}

void set_heavy(mi_spmm_schedule* sc, int32_t heavy_len) {
  // slots before the first class that holds a row of ≤ heavy_len entries: every row there is longer than heavy_len
  const int c = length_class(heavy_len);  // rows of this class may be ≤ heavy_len: not heavy
  sc->rs.heavy_len = heavy_len;
  sc->rs.heavy = sc->start[kClasses - 1 - c];  // buckets 0 … (kClasses − 2 − c) = classes above c
  // Is the order worth its indirection?  It costs the locality of consecutive rows (rowptr reads, C rows written side by side):
  // 4 M rows of 1 … 8 entries ran 11 – 16 % SLOWER scheduled, 2.4 M rows of ≤ 100 entries around a mean of 50 ran 5 % faster
  // (profiles/r06_degree_skew.log).  Active with heavy rows, or with rows that are long enough to be gather-bound (mean ≥ 16)
  // and differ in length (longest ≥ 1.5 × mean).
  const double mean = sc->rs.rows > 0 ? (double)sc->nnz / (double)sc->rs.rows : 0.0;
  sc->rs.active = sc->rs.heavy > 0 || (mean >= 16.0 && (double)sc->longest >= 1.5 * mean);
}

}  // namespace

extern "C" {

size_t mi_spmm_schedule_workspace_bytes(int32_t M) {
  (void)M;
  return (size_t)(3 * kClasses + 1) * sizeof(int);
}

int mi_spmm_schedule_create(const int32_t* rowptr, int32_t M, int64_t nnz, int32_t N, int32_t* order, void* workspace,
                            size_t workspace_bytes, mi_stream_t stream, mi_spmm_schedule_t** out) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!out) return MI_EINVAL;
  *out = nullptr;
  if (M < 0 || nnz < 0 || N < 0) return MI_EINVAL;
  if (nnz > 0x7fffffffLL) return MI_ERANGE;
  if (M > 0 && (!rowptr || !order || !workspace)) return MI_EINVAL;
  if (workspace_bytes < mi_spmm_schedule_workspace_bytes(M)) return MI_ENOMEM;
  mi_spmm_schedule* sc = new (std::nothrow) mi_spmm_schedule();
  if (!sc) return MI_ENOMEM;
  sc->rs = mi::RowSchedule{order, M, 0, 0, true, nullptr, nullptr, nullptr};
  sc->nnz = nnz;
  sc->n_width = N;
  for (int b = 0; b <= kClasses; ++b) sc->start[b] = 0;
  if (M > 0) {
    int* hist = static_cast<int*>(workspace);
    int* start = hist + kClasses;
    int* cursor = start + kClasses + 1;
    hipError_t e = hipMemsetAsync(hist, 0, kClasses * sizeof(int), s);
    if (e == hipSuccess) {
      const long want = ((long)M + 255) / 256;
      hipLaunchKernelGGL(sched_hist_kernel, dim3((unsigned)(want < 2048 ? want : 2048)), dim3(256), 0, s, rowptr, M, hist);
      hipLaunchKernelGGL(sched_scan_kernel, dim3(1), dim3(256), 0, s, hist, start, cursor);
      hipLaunchKernelGGL(sched_scatter_kernel, dim3((unsigned)(((long)M + kTileRows - 1) / kTileRows)), dim3(256), 0, s, rowptr,
                         M, cursor, order);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(sc->start, start, (kClasses + 1) * sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
      delete sc;
      return mi::record_hip_error(e);
    }
  }
  sc->longest = 0, sc->classes = 0;
  for (int b = kClasses - 1; b >= 0; --b) {
    if (sc->start[b + 1] > sc->start[b]) {
      sc->classes++;
      sc->longest = class_floor(kClasses - 1 - b);
    }
  }
  set_heavy(sc, heavy_length_for(nnz));
  // the side stream of the heavy launch and its fork / join events
  hipError_t e = hipStreamCreateWithFlags(&sc->rs.side, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&sc->rs.fork, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&sc->rs.join, hipEventDisableTiming);
  if (e != hipSuccess) {
    mi_spmm_schedule_destroy(sc);
    return mi::record_hip_error(e);
  }
  *out = sc;
  return MI_OK;
}

int mi_spmm_schedule_destroy(mi_spmm_schedule_t* sc) {
  if (!sc) return MI_OK;
  if (sc->rs.fork) (void)hipEventDestroy(sc->rs.fork);
  if (sc->rs.join) (void)hipEventDestroy(sc->rs.join);
  if (sc->rs.side) (void)hipStreamDestroy(sc->rs.side);
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
  info[5] = (sc->rs.side != nullptr ? 1 : 0) | (sc->rs.active ? 2 : 0);
  info[6] = sc->nnz;
  info[7] = sc->n_width;
  return MI_OK;
}

int mi_spmm_schedule_set_heavy(mi_spmm_schedule_t* sc, int32_t heavy_len, int use_side_stream) {
  if (!sc || heavy_len < 0) return MI_EINVAL;
  set_heavy(sc, heavy_len);
  sc->rs.active = true;  // a pinned heavy length is a request to run scheduled
  if (!use_side_stream && sc->rs.side) {  // (tests, A/B: the heavy launch in line, ahead of the rest)
    (void)hipStreamDestroy(sc->rs.side);
    sc->rs.side = nullptr;
  }
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
