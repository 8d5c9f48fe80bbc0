// Format-side kernels of the SpMM path (include/mi_spmm.h):
//   dense 2-D transpose, column-major SpMM executor, device dense→CSR
//   (count / scan / fill, batched), column sums, SDDMM.  (CSR transpose: csr_transpose.hip.)
// Reference counterparts are cited at each entry point.
#include <cstring>

#include "mi_common.h"
#include "mi_lanes.h"

namespace {

// --------------------------------------------------------------------------
// Dense transpose: dst[c, r] = src[r, c]; 64×64 tiles through padded LDS so
// both the global read and the global write are 256-B coalesced.
// --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, int rows,
                                                        int cols, long ld_src,
                                                        float* __restrict__ dst, long ld_dst,
                                                        int tiles_c) {
  __shared__ float tile[64][65];
  const int tr = blockIdx.x / tiles_c, tc = blockIdx.x % tiles_c;
  const int r0 = tr * 64, c0 = tc * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + ty + i * 4, c = c0 + tx;
    if (r < rows && c < cols) tile[ty + i * 4][tx] = src[(long)r * ld_src + c];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ty + i * 4, r = r0 + tx;
    if (c < cols && r < rows) dst[(long)c * ld_dst + r] = tile[tx][ty + i * 4];
  }
}

int launch_transpose(const float* src, int rows, int cols, long ld_src, float* dst, long ld_dst,
                     hipStream_t s) {
  if (rows == 0 || cols == 0) return MI_OK;
  const long tiles_r = (rows + 63) / 64, tiles_c = (cols + 63) / 64;
  if (tiles_r * tiles_c > 0x7fffffffL) return MI_ERANGE;
  hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)(tiles_r * tiles_c)), dim3(256), 0, s, src,
                     rows, cols, ld_src, dst, ld_dst, (int)tiles_c);
  return mi::check_launch();
}

// --------------------------------------------------------------------------
// Exclusive scan of n int32 counts, 2048 per block, three launches.
// scatter variant: element i = (b, r) with r < rows goes to
// out[b*(rows+1) + r]; the item's end slot out[b*(rows+1)+rows] gets the
// inclusive value — the "rowptr of rowptrs" layout of the batched CSR.
// --------------------------------------------------------------------------
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = 256 * SCAN_ITEMS;

__device__ __forceinline__ int block_exclusive_scan_256(int v, int* total) {
  __shared__ int wave_sums[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  if (lane == 63) wave_sums[wave] = incl;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int s = wave_sums[w];
    if (w < wave) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

__global__ __launch_bounds__(256) void scan_block_sums_kernel(const int* __restrict__ in, long n,
                                                              int* __restrict__ block_sums) {
  const long base = (long)blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  int s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i)
    if (base + i < n) s += in[base + i];
  int total;
  block_exclusive_scan_256(s, &total);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// Single block: exclusive scan of block_sums in place.
__global__ __launch_bounds__(256) void scan_of_sums_kernel(int* __restrict__ block_sums, long nb) {
  int carry = 0;
  for (long b0 = 0; b0 < nb; b0 += 256) {
    const long i = b0 + threadIdx.x;
    const int v = i < nb ? block_sums[i] : 0;
    int total;
    const int ex = block_exclusive_scan_256(v, &total);
    if (i < nb) block_sums[i] = carry + ex;
    carry += total;
  }
}

__global__ __launch_bounds__(256) void scan_scatter_kernel(const int* __restrict__ in, long n,
                                                           const int* __restrict__ block_sums,
                                                           int* __restrict__ out, int rows) {
  const long base = (long)blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  int v[SCAN_ITEMS];
  int s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    v[i] = base + i < n ? in[base + i] : 0;
    s += v[i];
  }
  int total;
  int run = block_sums[blockIdx.x] + block_exclusive_scan_256(s, &total);
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    const long idx = base + i;
    if (idx < n) {
      const long b = idx / rows;
      const int r = (int)(idx - b * rows);
      out[b * ((long)rows + 1) + r] = run;
      if (r == rows - 1) out[b * ((long)rows + 1) + rows] = run + v[i];
    }
    run += v[i];
  }
}

size_t scan_workspace_ints(long n) { return (size_t)((n + SCAN_TILE - 1) / SCAN_TILE) + 1; }

// counts[n] → out (rowptr-of-rowptrs with `rows` rows per item); n = batch*rows.
int launch_scan_rowptr(const int* counts, long n, int rows, int* out, int* block_sums,
                       hipStream_t s) {
  const long nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (nb > 0x7fffffffL) return MI_ERANGE;
  hipLaunchKernelGGL(scan_block_sums_kernel, dim3((unsigned)nb), dim3(256), 0, s, counts, n,
                     block_sums);
  hipLaunchKernelGGL(scan_of_sums_kernel, dim3(1), dim3(256), 0, s, block_sums, nb);
  hipLaunchKernelGGL(scan_scatter_kernel, dim3((unsigned)nb), dim3(256), 0, s, counts, n,
                     block_sums, out, rows);
  return mi::check_launch();
}

// --------------------------------------------------------------------------
// Dense → CSR: one wave per dense row.  VEC = 4: a lane reads 16 B (4 columns),
// the row is swept 256 columns per wave-instruction; VEC = 1 for rows that are
// not 16-B aligned.  Ranks come from four ballots (one per lane component) so
// columns stay ascending.
// --------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void count_nonzeros_kernel(const float* __restrict__ dense,
                                                             long total_rows, int rows, int cols,
                                                             long ld, long stride,
                                                             int* __restrict__ counts) {
  const int lane = threadIdx.x & 63;
  const long id = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (id >= total_rows) return;
  const long b = id / rows;
  const float* src = dense + b * stride + (id - b * rows) * ld;
  int cnt = 0;
  if (VEC == 4) {
    for (int c0 = 0; c0 < cols; c0 += 256) {
      const int c = c0 + lane * 4;
      mi::f32x4 x = mi::f32x4{0.f, 0.f, 0.f, 0.f};
      if (c + 3 < cols) {
        x = __builtin_nontemporal_load(reinterpret_cast<const mi::f32x4*>(src + c));
      } else {
        if (c + 0 < cols) x.x = src[c + 0];
        if (c + 1 < cols) x.y = src[c + 1];
        if (c + 2 < cols) x.z = src[c + 2];
      }
      cnt += (x.x != 0.0f) + (x.y != 0.0f) + (x.z != 0.0f) + (x.w != 0.0f);
    }
#pragma unroll
    for (int w = 32; w >= 1; w >>= 1) cnt += __shfl_xor(cnt, w, 64);
  } else {
    for (int c0 = 0; c0 < cols; c0 += 64) {
      const int c = c0 + lane;
      const bool nz = c < cols && src[c] != 0.0f;
      cnt += __builtin_popcountll(__ballot(nz));
    }
  }
  if (lane == 0) counts[id] = cnt;
}

__device__ __forceinline__ int lanes_below(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

template <int VEC>
__global__ __launch_bounds__(256) void fill_csr_kernel(const float* __restrict__ dense,
                                                       long total_rows, int rows, int cols, long ld,
                                                       long stride, const int* __restrict__ rowptr,
                                                       int* __restrict__ col,
                                                       float* __restrict__ val) {
  const int lane = threadIdx.x & 63;
  const long id = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (id >= total_rows) return;
  const long b = id / rows;
  const int r = (int)(id - b * rows);
  const float* src = dense + b * stride + (long)r * ld;
  long out = rowptr[b * ((long)rows + 1) + r];
  if (VEC == 4) {
    for (int c0 = 0; c0 < cols; c0 += 256) {
      const int c = c0 + lane * 4;
      mi::f32x4 x = mi::f32x4{0.f, 0.f, 0.f, 0.f};
      if (c + 3 < cols) {
        x = __builtin_nontemporal_load(reinterpret_cast<const mi::f32x4*>(src + c));
      } else {
        if (c + 0 < cols) x.x = src[c + 0];
        if (c + 1 < cols) x.y = src[c + 1];
        if (c + 2 < cols) x.z = src[c + 2];
      }
      const bool n0 = x.x != 0.0f, n1 = x.y != 0.0f, n2 = x.z != 0.0f, n3 = x.w != 0.0f;
      const unsigned long long m0 = __ballot(n0), m1 = __ballot(n1), m2 = __ballot(n2), m3 = __ballot(n3);
      long o = out + lanes_below(m0) + lanes_below(m1) + lanes_below(m2) + lanes_below(m3);
      if (n0) { col[o] = c;     val[o] = x.x; ++o; }
      if (n1) { col[o] = c + 1; val[o] = x.y; ++o; }
      if (n2) { col[o] = c + 2; val[o] = x.z; ++o; }
      if (n3) { col[o] = c + 3; val[o] = x.w; }
      out += __builtin_popcountll(m0) + __builtin_popcountll(m1) + __builtin_popcountll(m2) +
             __builtin_popcountll(m3);
    }
  } else {
    for (int c0 = 0; c0 < cols; c0 += 64) {
      const int c = c0 + lane;
      const float x = c < cols ? src[c] : 0.0f;
      const bool nz = c < cols && x != 0.0f;
      const unsigned long long mask = __ballot(nz);
      if (nz) {
        col[out + lanes_below(mask)] = c;
        val[out + lanes_below(mask)] = x;
      }
      out += __builtin_popcountll(mask);
    }
  }
}

size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

// --------------------------------------------------------------------------
// Column sums  dst[j] = Σ_r src[r, j]  (bias gradient of the FC layers).  Workgroup (ct, rc): one
// column tile × one chunk of rows; a lane owns 4 consecutive columns (16-B loads, 1 KB per wave and
// row) when the layout allows, the 4 waves take rows rc0 + w, rc0 + w + 4, … four at a time in
// separate accumulators; wave sums are added in wave order through LDS and the chunks by a second
// launch.  Fixed order, no atomics.  16384 × 3072 (201 MB): 0.041 ms ≈ 4.9 TB/s (torch.sum 0.039 ms;
// tools/bench_fc.py).
// --------------------------------------------------------------------------
// rows per chunk: 64 for up to 64 Ki rows (thousands of workgroups on a [tokens, features]
// gradient), growing so that the second pass never has more than ≈1024 partial rows to add
int colsum_chunk_rows(int rows) {
  int c = 64;
  while ((long)c * 1024 < rows) c *= 2;
  return c;
}

// VEC: lane ↔ 4 consecutive columns (16-B loads, 256 columns per wave and row); else lane ↔ column.
template <bool VEC>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ src, int rows, int n,
                                                             long ld, int chunk_rows, float* __restrict__ partial) {
  constexpr int W = VEC ? 4 : 1;
  __shared__ float part[4][64 * W];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int colj = (blockIdx.x * 64 + lane) * W;
  const int r0 = blockIdx.y * chunk_rows;
  const int r1 = r0 + chunk_rows < rows ? r0 + chunk_rows : rows;
  float acc[4][W];  // four rows in flight per wave
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int c = 0; c < W; ++c) acc[u][c] = 0.f;
  if (colj < n) {
    for (int r = r0 + wave; r < r1; r += 16) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rr = r + 4 * u;
        if (rr < r1) {
          const float* p = src + (long)rr * ld + colj;
          if constexpr (VEC) {
            const mi::f32x4 v = *reinterpret_cast<const mi::f32x4*>(p);
            acc[u][0] += v.x;
            acc[u][1] += v.y;
            acc[u][2] += v.z;
            acc[u][3] += v.w;
          } else {
            acc[u][0] += p[0];
          }
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < W; ++c) part[wave][lane * W + c] = ((acc[0][c] + acc[1][c]) + acc[2][c]) + acc[3][c];
  __syncthreads();
  if (wave == 0 && colj < n) {
#pragma unroll
    for (int c = 0; c < W; ++c) {
      const int i = lane * W + c;
      partial[(long)blockIdx.y * n + colj + c] = ((part[0][i] + part[1][i]) + part[2][i]) + part[3][i];
    }
  }
}

// dst[j] = Σ_c partial[c][j]: 64 columns per workgroup, the 4 waves take chunks w, w+4, …
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, int chunks, int n,
                                                           float* __restrict__ dst) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  float a0 = 0.f, a1 = 0.f;
  if (j < n) {
    int c = wave;
    for (; c + 4 < chunks; c += 8) {
      a0 += partial[(long)c * n + j];
      a1 += partial[(long)(c + 4) * n + j];
    }
    if (c < chunks) a0 += partial[(long)c * n + j];
  }
  part[wave][lane] = a0 + a1;
  __syncthreads();
  if (wave == 0 && j < n) dst[j] = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
}

// SDDMM on A's pattern (the gradient of C = A·B with respect to A's stored values):
//   out[p] = Σ_j dC[row(p), j] · B[col[p], j]
// Summation order (restated by oracle_sddmm_csr_f32): lane l of a 64-lane wave chains the products
// of columns j = 256t + 4l + c (t ascending, c = 0…3, j < N) with fmaf, and the 64 lane sums are
// combined by the xor tree 32, 16, 8, 4, 2, 1.
//
// One wave per row of A; 256·T columns of the dC row sit in registers at a time (16 B per lane and
// 256 columns; wider rows are walked in chunks, each lane's chain simply continuing).  The
// row's non-zeros are taken 64 at a time: eight 1-KB rows of B are gathered per step (the same
// pipeline as the forward kernel), each leaving one partial sum per lane, and the 64 × 64 partial
// sums of a batch are reduced TOGETHER: at xor distance w the two halves of the remaining values
// are exchanged between the lanes with bit w clear and set, so every level halves the number of
// live registers and after six levels lane i holds the finished out[p0 + i] — 63 exchanges for 64
// results instead of 6 × 64, and one coalesced 256-B store.  The pairing is exactly the butterfly's,
// so the bits are the same as a per-non-zero butterfly would give.
//
// PANEL (Infinity-Cache blocking, as in the forward product's two-panel plan, spmm_csr.hip): a launch
// handles only the non-zeros with c_lo ≤ column < c_hi, so that all CUs gather from one slice of a B far
// larger than the cache at the same time.  An output value depends on ONE row of B, so nothing is carried
// between the panel launches (unlike the forward product's C): each launch compacts the chunk's in-panel
// entries (ballot, in CSR order), runs the same gathers and the same joint tree on them, and hands every
// result back to the lane that holds its non-zero.  Per-value arithmetic is untouched → same bits.
template <int T, bool VEC, bool PANEL>
__global__ __launch_bounds__(256) void sddmm_kernel(const int* __restrict__ rowptr,
                                                    const int* __restrict__ col, int M, int N,
                                                    const float* __restrict__ dC, long lddc,
                                                    const float* __restrict__ B, long ldb,
                                                    float* __restrict__ out, int c_lo, int c_hi) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (row >= M) return;
  const int start = rowptr[row], end = rowptr[row + 1];
  if (start == end) return;

  // 4 consecutive columns per lane and pass (zeros beyond N, which the chain below never uses)
  auto load4 = [&](const float* src, int t) {
    const int j = 256 * t + 4 * lane;
    mi::f32x4 v = mi::f32x4{0.f, 0.f, 0.f, 0.f};
    if (VEC) {
      if (j < N) v = *reinterpret_cast<const mi::f32x4*>(src + j);
    } else {
      if (j + 0 < N) v.x = src[j + 0];
      if (j + 1 < N) v.y = src[j + 1];
      if (j + 2 < N) v.z = src[j + 2];
      if (j + 3 < N) v.w = src[j + 3];
    }
    return v;
  };
  constexpr int U = 8;
  const float* xrow = dC + row * lddc;
  for (int p0 = start; p0 < end; p0 += 64) {
    int cnt = end - p0 < 64 ? end - p0 : 64;
    const int mycol = lane < cnt ? col[p0 + lane] : 0;
    unsigned long long pmask = ~0ull;  // PANEL: lanes whose non-zero belongs to this launch
    bool mine = lane < cnt;
    if (PANEL) {
      mine = mine && (unsigned)(mycol - c_lo) < (unsigned)(c_hi - c_lo);
      pmask = __ballot(mine);
      cnt = __builtin_popcountll(pmask);  // results are computed for the compacted entries 0 … cnt-1
      if (cnt == 0) continue;
    }
    float s[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) s[i] = 0.f;
    // columns in chunks of 256·T (one chunk when N ≤ 256·T): each lane's chain simply continues
    for (int t0 = 0; 256 * t0 < N; t0 += T) {
      mi::f32x4 x[T];
#pragma unroll
      for (int t = 0; t < T; ++t) x[t] = load4(xrow, t0 + t);
      unsigned long long walk = pmask;  // PANEL: compacted entry i+u is the (i+u)-th set bit
#pragma unroll
      for (int i = 0; i < 64; i += U) {
        if (i < cnt) {
          mi::f32x4 y[U][T];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            int src_lane = i + u;
            if (PANEL) {
              src_lane = walk ? __builtin_ctzll(walk) : 0;  // past the last entry: any lane (its sum is never stored)
              walk &= walk - 1;
            }
            const float* brow = B + (long)__builtin_amdgcn_readlane(mycol, src_lane) * ldb;  // row 0 when past cnt
#pragma unroll
            for (int t = 0; t < T; ++t) y[u][t] = load4(brow, t0 + t);
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            float acc = s[i + u];
#pragma unroll
            for (int t = 0; t < T; ++t) {
              // columns at or beyond N are not part of the chain (adding their +0 would turn a -0 sum into +0)
              const int j = 256 * (t0 + t) + 4 * lane;
              if (VEC) {
                if (j < N) {  // N % 4 == 0: all four or none
                  acc = __builtin_fmaf(x[t].x, y[u][t].x, acc);
                  acc = __builtin_fmaf(x[t].y, y[u][t].y, acc);
                  acc = __builtin_fmaf(x[t].z, y[u][t].z, acc);
                  acc = __builtin_fmaf(x[t].w, y[u][t].w, acc);
                }
              } else {
                if (j + 0 < N) acc = __builtin_fmaf(x[t].x, y[u][t].x, acc);
                if (j + 1 < N) acc = __builtin_fmaf(x[t].y, y[u][t].y, acc);
                if (j + 2 < N) acc = __builtin_fmaf(x[t].z, y[u][t].z, acc);
                if (j + 3 < N) acc = __builtin_fmaf(x[t].w, y[u][t].w, acc);
              }
            }
            s[i + u] = acc;
          }
        }
      }
    }
    // joint xor tree: level w pairs value k with value k + w; lanes with bit w set keep the upper one
#pragma unroll
    for (int w = 32; w >= 1; w >>= 1) {
      const bool hi = (lane & w) != 0;
#pragma unroll
      for (int k = 0; k < w; ++k) {
        const float keep = hi ? s[k + w] : s[k];
        const float send = hi ? s[k] : s[k + w];
        s[k] = keep + __shfl_xor(send, w, 64);
      }
    }
    if (PANEL) {
      // lane i holds the result of compacted entry i; the lane holding that non-zero is the i-th set bit
      const int rank = lanes_below(pmask);
      const float v = __shfl(s[0], rank, 64);
      if (mine) out[p0 + lane] = v;
    } else if (lane < cnt) {
      out[p0 + lane] = s[0];
    }
  }
}

// The same sums for narrow rows (N ≤ 64, N % 4 == 0 — attention heads): G = N/4 (rounded up to a power of two) lanes
// per row of A instead of a whole wave of which only N/4 lanes had columns, 64/G rows per wave.  Lane l of a group
// chains columns 4l … 4l+3 — the chain of lane l of the wave above — and the joint xor tree runs over the distances
// G/2 … 1; the levels 32 … G of the 64-lane tree only ever added the +0 of a lane without columns, which changes a
// partial sum in one case, −0 → +0 (a chain of products that all underflow): `+ 0.0f` per skipped level restates
// exactly that.  Same bits as sddmm_kernel and the oracle; 4× the gathers in flight at N = 64 (block-diagonal batch of
// 384 × 512² at 10 %: 0.43 → see profiles/r03_attention_csr.log).  A row's columns travel to its group through
// compile-time lane broadcasts (mi_lanes.h).
template <int G>  // ≤ 16: a group's columns sit in one 16-lane DPP row
__global__ __launch_bounds__(256) void sddmm_group_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                          int M, int N, const float* __restrict__ dC, long lddc,
                                                          const float* __restrict__ B, long ldb,
                                                          float* __restrict__ out) {
  constexpr int RPW = 64 / G;
  constexpr int U = G < 8 ? G : 8;  // gathers in flight per lane
  const int lane = threadIdx.x & 63, gl = lane & (G - 1);
  const long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / G;
  const int nq = N >> 2;
  const bool on = gl < nq;
  int start = 0, end = 0;
  if (row < M) {
    start = rowptr[row];
    end = rowptr[row + 1];
  }
  mi::f32x4 x = mi::f32x4{0.f, 0.f, 0.f, 0.f};
  if (on && start < end) x = *reinterpret_cast<const mi::f32x4*>(dC + row * lddc + 4 * gl);
  for (int p0 = start; p0 < end; p0 += G) {  // trip count differs between groups
    const int cnt = end - p0 < G ? end - p0 : G;  // group-uniform
    const int mycol = gl < cnt ? col[p0 + gl] : 0;  // past the row's end: row 0 of B (its sum is never stored)
    float s[G];
    mi::static_for<G / U>([&](auto b_) {
      constexpr int i = U * decltype(b_)::value;
      if (i < cnt) {
        mi::f32x4 y[U];
        mi::static_for<U>([&](auto u_) {
          constexpr int u = decltype(u_)::value;
          const float* brow = B + (long)mi::group_lane<G, i + u, true>(mycol) * ldb;
          y[u] = on ? *reinterpret_cast<const mi::f32x4*>(brow + 4 * gl) : mi::f32x4{0.f, 0.f, 0.f, 0.f};
        });
#pragma unroll
        for (int u = 0; u < U; ++u) {
          float acc = 0.f;
          if (on) {
            acc = __builtin_fmaf(x.x, y[u].x, acc);
            acc = __builtin_fmaf(x.y, y[u].y, acc);
            acc = __builtin_fmaf(x.z, y[u].z, acc);
            acc = __builtin_fmaf(x.w, y[u].w, acc);
          }
#pragma unroll
          for (int w = 32; w >= G; w >>= 1) acc = acc + 0.0f;  // the tree levels whose partner never had columns
          s[i + u] = acc;
        }
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) s[i + u] = 0.f;
      }
    });
    // joint xor tree over the group (see sddmm_kernel): after the last level lane l holds the sum of entry l
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
}


// dst[p] = src[perm[p]] — the values of a CSR tensor carried into its cached transposed pattern (matmuls' backward where the
// product's plan takes no permutation itself).  Four entries per lane: the perm words and the results move as 16-byte
// vectors, the gathers are single dwords (a permutation has no locality to offer).
template <bool VEC>
__global__ __launch_bounds__(256) void gather_perm_kernel(const float* __restrict__ src, const int* __restrict__ perm, long n,
                                                          float* __restrict__ dst) {
  const long p = 4 * ((long)blockIdx.x * 256 + threadIdx.x);
  if (p >= n) return;
  if (VEC && p + 4 <= n) {
    const int4 q = *reinterpret_cast<const int4*>(perm + p);
    float4 v;
    v.x = src[q.x], v.y = src[q.y], v.z = src[q.z], v.w = src[q.w];
    *reinterpret_cast<float4*>(dst + p) = v;
  } else {
    for (long i = p; i < n && i < p + 4; ++i) dst[i] = src[perm[i]];
  }
}

}  // namespace


extern "C" {

int mi_transpose_f32(const float* src, int32_t rows, int32_t cols, int64_t ld_src, float* dst,
                     int64_t ld_dst, mi_stream_t stream) {
  if (rows < 0 || cols < 0) return MI_EINVAL;
  if (rows == 0 || cols == 0) return MI_OK;
  if (!src || !dst || ld_src < cols || ld_dst < rows) return MI_EINVAL;
  return launch_transpose(src, rows, cols, ld_src, dst, ld_dst, static_cast<hipStream_t>(stream));
}

// Column-major executor (round-1 form): bring B to row-major [K,N] and C back
// from row-major [M,N] with two coalesced tile transposes around the row-split
// kernel.  The transposes move 8·(K+M)·N bytes against the gather's 4·N·nnz, so
// they are a small fraction for rows longer than a few nonzeros.
size_t mi_spmm_colmajor_workspace_bytes(int32_t M, int32_t K, int32_t N) {
  if (M < 0 || K < 0 || N < 0) return 0;
  return align_up((size_t)K * N * 4) + align_up((size_t)M * N * 4);
}

// Native form: when the product is one the LDS-slab plan serves (moderate density, enough tiles), the slab
// kernel reads X = Bᵀ and writes Y = Cᵀ directly (transposing slab loads, transposed tile store) — no
// transposed copies, the workspace stays untouched.  Same CSR-order chain per element as every other plan,
// so the bits do not depend on which form ran.  The transposing (scalar) LDS writes cost the slab kernel
// ≈5 % (measured, 4096² × 16384 at 10 %: 3.13 vs 2.97 ms for the row-major kernel alone); the two tile
// transposes cost their 16·(K+M)·N bytes at ≈3.7 TB/s (0.55 ms there): the native form is taken where that
// is the larger price — in practice wherever the slab plan runs (3072 × 768 × 16384 at 10 %: 0.475 vs
// 0.44 + 0.30 ms).
int mi_spmm_colmajor_native_form(int64_t nnz, int32_t M, int32_t K, int32_t N, const float* B, int64_t ldb,
                                 const float* C, int64_t ldc) {
  if (M <= 0 || N <= 0 || K < 4 || nnz <= 0) return 0;
  if (N % 4 != 0 || ldb % 4 != 0 || ldc % 4 != 0 || !mi::aligned16(B) || !mi::aligned16(C)) return 0;
  // the plan of the row-major product on 16-byte aligned operands of leading dimension N
  alignas(16) static const float probe[4] = {0.f, 0.f, 0.f, 0.f};
  if (mi_spmm_csr_f32_plan(nnz, M, K, N, probe, N, probe, N) != MI_SPMM_SLAB) return 0;
  const double wgs = (double)(((long)M + 127) / 128) * (double)(((long)N + 255) / 256);
  const double density = (double)nnz / ((double)M * (double)K);
  const double t_slab = (wgs <= 256.0 ? 1.0 : wgs / 256.0) * (double)(((long)K + 63) / 64) * (2.35e-6 + 34e-6 * density);
  const double t_transposes = 16.0 * ((double)K + (double)M) * (double)N / 3.7e12;
  return 0.06 * t_slab < t_transposes ? 1 : 0;
}

int mi_spmm_colmajor_form(int64_t nnz, int32_t M, int32_t K, int32_t N, const float* B, int64_t ldb,
                          const float* C, int64_t ldc, const void* workspace) {
  if (mi_spmm_colmajor_native_form(nnz, M, K, N, B, ldb, C, ldc)) return 1;
  if (nnz > 0 && K > 0 && M > 0 && N > 0 && workspace &&
      mi::launch_spmm_wave_row_colmajor_out(nullptr, nullptr, nullptr, nnz, M, K, N, static_cast<const float*>(workspace),
                                            N, nullptr, ldc, false, nullptr) == MI_OK)
    return 2;
  return 0;
}

int mi_spmm_csr_colmajor_sched_f32(const mi_spmm_schedule_t* schedule, const int32_t* rowptr, const int32_t* col,
                                   const float* val, int64_t nnz, int32_t M, int32_t K, int32_t N, const float* B,
                                   int64_t ldb, float* C, int64_t ldc, int long_rows, void* long_rows_workspace,
                                   size_t long_rows_workspace_bytes, void* workspace, size_t workspace_bytes,
                                   mi_stream_t stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  // an ACTIVE row schedule (a degree-skewed matrix: include/mi_spmm.h "Row schedules") runs the row-major product between the
  // two operand transposes — the form that can hand rows to waves in the schedule's order; the LDS-slab form on column-major
  // operands stays (a long row is ordinary work there), the fused-output kernel (16 consecutive rows per workgroup) does not
  bool scheduled = false;
  if (schedule != nullptr) {
    int64_t info[8];
    if (mi_spmm_schedule_info(schedule, info) == MI_OK) scheduled = (info[5] & 2) != 0 && info[0] == M;
  }
  if (M < 0 || K < 0 || N < 0 || nnz < 0) return MI_EINVAL;
  if (M == 0 || N == 0) return MI_OK;
  if (!rowptr || !C || ldc < M) return MI_EINVAL;
  if (K > 0 && (!B || ldb < K)) return MI_EINVAL;
  if (workspace_bytes < mi_spmm_colmajor_workspace_bytes(M, K, N)) return MI_ENOMEM;
  if (!workspace || !mi::aligned16(workspace)) return MI_EINVAL;
  if (long_rows == MI_LONG_ROWS_NONE && mi_spmm_colmajor_native_form(nnz, M, K, N, B, ldb, C, ldc))
    return mi::launch_spmm_slab_colmajor(rowptr, col, val, B, C, M, K, N, ldb, ldc, s);
  float* Bt = static_cast<float*>(workspace);  // [K, N] row-major
  float* Ct = reinterpret_cast<float*>(static_cast<char*>(workspace) + align_up((size_t)K * N * 4));
  // column-major K×N with ldb  ==  row-major [N, ldb]; its transpose is [K, N].
  int st = launch_transpose(B, N, K, ldb, Bt, N, s);
  if (st != MI_OK) return st;
  if (long_rows == MI_LONG_ROWS_NONE && nnz > 0 && K > 0 && !scheduled) {
    // one wave per row with the output transpose fused into its epilogue, where that is the plan
    st = mi::launch_spmm_wave_row_colmajor_out(rowptr, col, val, nnz, M, K, N, Bt, N, C, ldc, true, s);
    if (st <= MI_OK) return st;
  }
  st = scheduled ? mi_spmm_csr_scheduled_f32(schedule, MI_SPMM_AUTO, rowptr, col, val, nnz, M, K, N, Bt, N, nullptr, Ct, N, long_rows,
                                             long_rows_workspace, long_rows_workspace_bytes, stream)
                 : mi_spmm_csr_ex_f32(rowptr, col, val, nnz, M, K, N, Bt, N, nullptr, Ct, N, long_rows, long_rows_workspace,
                                      long_rows_workspace_bytes, stream);
  if (st != MI_OK) return st;
  // row-major [M, N] → row-major [N, ldc]  ==  column-major M×N with ldc.
  return launch_transpose(Ct, M, N, N, C, ldc, s);
}

int mi_spmm_csr_colmajor_ex_f32(const int32_t* rowptr, const int32_t* col, const float* val,
                                int64_t nnz, int32_t M, int32_t K, int32_t N, const float* B,
                                int64_t ldb, float* C, int64_t ldc, int long_rows, void* long_rows_workspace,
                                size_t long_rows_workspace_bytes, void* workspace, size_t workspace_bytes,
                                mi_stream_t stream) {
  return mi_spmm_csr_colmajor_sched_f32(nullptr, rowptr, col, val, nnz, M, K, N, B, ldb, C, ldc, long_rows, long_rows_workspace,
                                        long_rows_workspace_bytes, workspace, workspace_bytes, stream);
}

int mi_spmm_csr_colmajor_f32(const int32_t* rowptr, const int32_t* col, const float* val,
                             int64_t nnz, int32_t M, int32_t K, int32_t N, const float* B,
                             int64_t ldb, float* C, int64_t ldc, void* workspace,
                             size_t workspace_bytes, mi_stream_t stream) {
  return mi_spmm_csr_colmajor_ex_f32(rowptr, col, val, nnz, M, K, N, B, ldb, C, ldc, MI_LONG_ROWS_NONE, nullptr, 0,
                                     workspace, workspace_bytes, stream);
}

size_t mi_dense_to_csr_workspace_bytes(int32_t batch, int32_t rows) {
  if (batch < 0 || rows < 0) return 0;
  const long n = (long)batch * rows;
  return align_up((size_t)n * 4) + align_up(scan_workspace_ints(n) * 4);
}

// Replaces dense_to_csr's cusparseDenseToSparse_{bufferSize,analysis}
// (reference src/baseline_mm.cu:232-247) for a whole batch in three launches.
int mi_dense_to_csr_count(const float* dense, int32_t batch, int32_t rows, int32_t cols,
                          int64_t ld, int64_t stride, int32_t* rowptr, void* workspace,
                          size_t workspace_bytes, mi_stream_t stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (batch < 0 || rows < 0 || cols < 0) return MI_EINVAL;
  if (batch == 0) return MI_OK;
  if (!rowptr) return MI_EINVAL;
  const long n = (long)batch * rows;
  if (n == 0) {  // rows == 0: every item is just its end slot
    MI_HIP_TRY(hipMemsetAsync(rowptr, 0, sizeof(int32_t) * (size_t)batch, s));
    return MI_OK;
  }
  if ((cols > 0 && !dense) || ld < cols || stride < 0) return MI_EINVAL;
  if (workspace_bytes < mi_dense_to_csr_workspace_bytes(batch, rows)) return MI_ENOMEM;
  if (!workspace) return MI_EINVAL;
  int* counts = static_cast<int*>(workspace);
  int* block_sums = reinterpret_cast<int*>(static_cast<char*>(workspace) + align_up((size_t)n * 4));
  const long blocks = (n + 3) / 4;
  if (blocks > 0x7fffffffL) return MI_ERANGE;
  if (ld % 4 == 0 && stride % 4 == 0 && mi::aligned16(dense))
    hipLaunchKernelGGL(count_nonzeros_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, dense, n, rows,
                       cols, ld, stride, counts);
  else
    hipLaunchKernelGGL(count_nonzeros_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, dense, n, rows,
                       cols, ld, stride, counts);
  int st = mi::check_launch();
  if (st != MI_OK) return st;
  return launch_scan_rowptr(counts, n, rows, rowptr, block_sums, s);
}

// Replaces cusparseDenseToSparse_convert (reference src/baseline_mm.cu:249-258).
int mi_dense_to_csr_fill(const float* dense, int32_t batch, int32_t rows, int32_t cols, int64_t ld,
                         int64_t stride, const int32_t* rowptr, int32_t* col, float* val,
                         mi_stream_t stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (batch < 0 || rows < 0 || cols < 0) return MI_EINVAL;
  const long n = (long)batch * rows;
  if (n == 0 || cols == 0) return MI_OK;
  if (!dense || !rowptr || !col || !val || ld < cols || stride < 0) return MI_EINVAL;
  const long blocks = (n + 3) / 4;
  if (blocks > 0x7fffffffL) return MI_ERANGE;
  if (ld % 4 == 0 && stride % 4 == 0 && mi::aligned16(dense))
    hipLaunchKernelGGL(fill_csr_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, dense, n, rows, cols,
                       ld, stride, rowptr, col, val);
  else
    hipLaunchKernelGGL(fill_csr_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, dense, n, rows, cols,
                       ld, stride, rowptr, col, val);
  return mi::check_launch();
}

size_t mi_colsum_workspace_bytes(int32_t rows, int32_t n) {
  if (rows < 0 || n < 0) return 0;
  const size_t cr = (size_t)colsum_chunk_rows(rows);
  const size_t chunks = ((size_t)rows + cr - 1) / cr;
  return align_up((chunks ? chunks : 1) * (size_t)n * sizeof(float));
}

int mi_colsum_f32(const float* src, int32_t rows, int32_t n, int64_t ld, float* dst, void* workspace,
                  size_t workspace_bytes, mi_stream_t stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (rows < 0 || n < 0) return MI_EINVAL;
  if (n == 0) return MI_OK;
  if (!dst) return MI_EINVAL;
  if (rows == 0) {
    MI_HIP_TRY(hipMemsetAsync(dst, 0, sizeof(float) * (size_t)n, s));
    return MI_OK;
  }
  if (!src || ld < n || !workspace) return MI_EINVAL;
  if (workspace_bytes < mi_colsum_workspace_bytes(rows, n)) return MI_ENOMEM;
  const int cr = colsum_chunk_rows(rows);
  const int chunks = (rows + cr - 1) / cr;
  if (chunks > 65535) return MI_ERANGE;
  float* partial = static_cast<float*>(workspace);
  if (n % 4 == 0 && ld % 4 == 0 && mi::aligned16(src))
    hipLaunchKernelGGL(colsum_partial_kernel<true>, dim3((unsigned)((n + 255) / 256), (unsigned)chunks), dim3(256), 0,
                       s, src, rows, n, (long)ld, cr, partial);
  else
    hipLaunchKernelGGL(colsum_partial_kernel<false>, dim3((unsigned)((n + 63) / 64), (unsigned)chunks), dim3(256), 0,
                       s, src, rows, n, (long)ld, cr, partial);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, s, partial, chunks, n, dst);
  return mi::check_launch();
}

int mi_gather_f32(const float* src, const int32_t* perm, int64_t n, float* dst, mi_stream_t stream) {
  if (n < 0) return MI_EINVAL;
  if (n == 0) return MI_OK;
  if (!src || !perm || !dst) return MI_EINVAL;
  const long quads = (n + 3) / 4;
  const long blocks = (quads + 255) / 256;
  if (blocks > 0x7fffffffL) return MI_ERANGE;
  const bool vec = mi::aligned16(perm) && mi::aligned16(dst);
  if (vec) hipLaunchKernelGGL(gather_perm_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), src, perm, (long)n, dst);
  else hipLaunchKernelGGL(gather_perm_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), src, perm, (long)n, dst);
  return mi::check_launch();
}

int mi_sddmm_csr_f32(const int32_t* rowptr, const int32_t* col, int64_t nnz, int32_t M, int32_t K,
                     int32_t N, const float* dC, int64_t lddc, const float* B, int64_t ldb,
                     float* out_val, mi_stream_t stream) {
  if (M < 0 || N < 0 || K < 0 || nnz < 0) return MI_EINVAL;
  if (M == 0 || nnz == 0) return MI_OK;
  if (!rowptr || !col || !out_val) return MI_EINVAL;
  if (N > 0 && (!dC || !B || lddc < N || ldb < N)) return MI_EINVAL;
  const long blocks = ((long)M + 3) / 4;
  if (blocks > 0x7fffffffL) return MI_ERANGE;
  const bool vec = N % 4 == 0 && lddc % 4 == 0 && ldb % 4 == 0 && mi::aligned16(dC) && mi::aligned16(B);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int T = N <= 256 ? 1 : 2;  // wider rows: chunks of 512 columns
#define MI_SDDMM(T_, V_)                                                                                    \
  hipLaunchKernelGGL((sddmm_kernel<T_, V_, false>), dim3((unsigned)blocks), dim3(256), 0, s, rowptr, col, M, N, dC, \
                     (long)lddc, B, (long)ldb, out_val, 0, 0)
  // B far beyond the 256 MiB Infinity Cache and rows long enough to re-touch a panel: one launch per column panel of K,
  // as many panels as the forward product's gate takes (spmm_csr.hip ic_panels, fitted in round 5: slices of ≈ 0.5 – 0.7 GiB,
  // P ≈ |B| / 683 MiB from {2, 3, 4, 6, 8}; none beyond ≈ 6 GiB) — a pass here carries nothing through memory, it re-reads
  // the row's col entries and its dC row, so 8 P entries per row suffice
  // (N ≤ 64 stays with the lane-group kernel below: the one-wave-per-row kernel of the panel passes has columns for only
  // N / 4 of its lanes — 4 M × 64 at 100 per row ran 30.7 ms in panels, half the forward product's rate)
  if (vec && T == 1 && N > 64 && (long)K * ldb * 4 >= (768L << 20) && nnz * (long)N >= 8L * K * ldb) {
    const double want = (double)K * (double)ldb * 4.0 / (683.0 * 1048576.0);
    int panels = 2;
    // (measured, tools/probes/sddmm_regime.py: three panels at 2 GiB −2 % against two, six at 4 GiB +1 %: beyond three nothing is gained here)
    for (int p : {2, 3})
      if ((p - want < 0 ? want - p : p - want) <= (panels - want < 0 ? want - panels : panels - want)) panels = p;
    if (want <= 9.0 && nnz >= 8L * panels * M) {
      for (int q = 0; q < panels; ++q) {
        const int lo = (int)((long)K * q / panels), hi = (int)((long)K * (q + 1) / panels);
        hipLaunchKernelGGL((sddmm_kernel<1, true, true>), dim3((unsigned)blocks), dim3(256), 0, s, rowptr, col, M, N, dC,
                           (long)lddc, B, (long)ldb, out_val, lo, hi);
      }
      return mi::check_launch();
    }
  }
  // B beyond the L2s but far from that regime (6 MiB < |B| ≤ 128 MiB): panels of ≈4 MiB keep the gathers in L2,
  // as in the forward product (spmm_csr.hip: l2_panels) — and here a pass carries nothing, it only re-reads the
  // row's col entries and its dC row.  16384² × 256 at 10 %: 1.60 → 1.0 ms.
  if (vec && N == 256) {  // measured at N = 256 (1-KB row gathers) only
    const double b_bytes = (double)K * (double)ldb * 4.0;
    if (b_bytes > 6.0 * 1024 * 1024 && b_bytes <= 128.0 * 1024 * 1024) {
      int panels = (int)((b_bytes + (4 << 20) - 1) / (4 << 20));
      panels = panels > 8 ? 8 : panels;
      while (panels >= 2 && nnz < 32L * panels * M) --panels;  // a pass re-reads the dC row: it needs ≥ 32 non-zeros per row
      if (panels >= 2) {
        for (int q = 0; q < panels; ++q) {
          const int lo = (int)((long)K * q / panels), hi = (int)((long)K * (q + 1) / panels);
          hipLaunchKernelGGL((sddmm_kernel<1, true, true>), dim3((unsigned)blocks), dim3(256), 0, s, rowptr, col, M, N, dC,
                             (long)lddc, B, (long)ldb, out_val, lo, hi);
        }
        return mi::check_launch();
      }
    }
  }
  if (vec && N >= 4 && N <= 64) {  // narrow rows: N/4 lanes per row, several rows per wave (one DPP row at most)
    const int G = mi::pow2_ceil(N / 4);
    const long gblocks = ((long)M + 4 * (64 / G) - 1) / (4 * (64 / G));
#define MI_SDDMM_G(G_)                                                                                               \
  hipLaunchKernelGGL((sddmm_group_kernel<G_>), dim3((unsigned)gblocks), dim3(256), 0, s, rowptr, col, M, N, dC, (long)lddc, \
                     B, (long)ldb, out_val)
    switch (G) {
      case 1: MI_SDDMM_G(1); break;
      case 2: MI_SDDMM_G(2); break;
      case 4: MI_SDDMM_G(4); break;
      case 8: MI_SDDMM_G(8); break;
      default: MI_SDDMM_G(16); break;
    }
#undef MI_SDDMM_G
    return mi::check_launch();
  }
  if (vec) {
    if (T == 1) MI_SDDMM(1, true);
    else MI_SDDMM(2, true);
  } else {
    if (T == 1) MI_SDDMM(1, false);
    else MI_SDDMM(2, false);
  }
#undef MI_SDDMM
  return mi::check_launch();
}

}  // extern "C"
