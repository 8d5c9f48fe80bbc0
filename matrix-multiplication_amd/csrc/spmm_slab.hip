// Moderate-density CSR × dense for gfx950:  C = A·B with B staged through LDS in k-slabs.
//
// The row-split kernels (spmm_csr.hip) gather one 1-KB piece of a B row per non-zero and 256
// columns through the texture path, ≈16 cycles of a CU's 64 B/clk load pipe each; at the
// densities of pruned network weights (3–30 % non-zeros; the regime of the reference's
// dense-vs-sparse sweep, benchmarks/random_tensor_benchmark.py:70-73) every B row is wanted by
// many rows of A at once, so it pays to bring it on chip ONCE per 128 rows:
//
//  * a workgroup (16 waves) owns 128 rows of A × one 256-column tile of C; wave w keeps the
//    accumulators of its 8 rows in registers for the whole k loop (lane l ↔ columns 4l…4l+3);
//  * B is walked in slabs of 64 k-rows: the slab's 64 × 1 KB go global → registers → LDS (two
//    LDS buffers, one barrier per slab, two register sets so a slab's loads are issued two
//    iterations before they are needed);
//  * each row keeps a 64-entry window of its (column, value) pairs in two VGPRs (lane j ↔ entry
//    base+j).  Columns inside a row ascend, so the entries that fall into the current slab are a
//    run starting at the row's cursor: one ballot counts them, v_readlane broadcasts each
//    (column, value), and the B row comes from LDS with one ds_read_b128 per lane (256 B/clk
//    instead of 64) feeding the fmaf chain;
//  * non-zeros are consumed in CSR order (slabs ascend, entries inside a slab ascend), so the
//    result is bit-identical to the row-split kernels and the oracle.
//
// CSR rows whose columns do NOT ascend are legal input (the reference's COO→CSR keeps the input
// order inside a row, src/sparse_mm.cu:110-134): every window load checks the order, and a row
// that breaks it is recomputed after the slab loop by the plain CSR-order gather chain from
// global memory — same bits, just not fast.
#include <limits.h>

#include "mi_common.h"

namespace {

using mi::f32x4;

constexpr int kSlab = 64;         // k rows of B per slab
constexpr int kWaves = 16;        // waves per workgroup
constexpr int kRows = 8;          // rows of A per wave
constexpr int kStage = kSlab / kWaves;  // float4 per thread and slab
constexpr int kTileCols = 256;    // columns of C per workgroup
// LDS: two slabs of kSlab × 1 KB and one all-zero row (the target of padding slots, below)
constexpr int kLdsVec = 2 * kSlab * 64 + 64;
constexpr size_t kLdsBytes = (size_t)kLdsVec * sizeof(f32x4);

__device__ __forceinline__ f32x4 fma4(float a, f32x4 x, f32x4 acc) {
  acc.x = __builtin_fmaf(a, x.x, acc.x);
  acc.y = __builtin_fmaf(a, x.y, acc.y);
  acc.z = __builtin_fmaf(a, x.z, acc.z);
  acc.w = __builtin_fmaf(a, x.w, acc.w);
  return acc;
}

__device__ __forceinline__ float readlane_f(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// TR (column-major operands, the executor behind cusparse_mmul_opt / tiledspmm_mm): B is given as the
// caller's activations X = Bᵀ, row-major [N, ldb] (element B[k][n] = X[n·ldb + k]), and C is produced
// as Y = Cᵀ, row-major [N, ldc] — no transposed copies in memory:
//  * the slab loader reads 64-byte pieces of X rows (a wave instruction: 16 rows × 4 k-quads; the four
//    instructions of a wave cover 256 contiguous bytes of each of its 16 rows) and writes them TRANSPOSED
//    into the same [k][n] LDS image, scalar by scalar; LDS rows are padded to 65 float4, which with that
//    lane mapping makes a write instruction 2-way conflicted (16-way on unpadded rows) while the compute
//    side keeps its ONE plain ds_read_b128 per lane (a first version XOR-swizzled the column groups
//    instead: two more vector instructions per non-zero on an issue-bound kernel, +17 %);
//  * the epilogue stages the 128 × 256 tile through the (now free) slab buffers as [n][m] and writes
//    512-byte row segments of Y.
// grid: 1-D over (column tile, row block), dealt XCD-contiguously (below).
template <bool TR>
__global__ __launch_bounds__(kWaves * 64) void spmm_slab_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ val,
    const float* __restrict__ B, float* __restrict__ C, int M, int K, int N, long ldb, long ldc,
    const float* __restrict__ bias, int ctiles, unsigned row_blocks, int long_thresh) {
  extern __shared__ __attribute__((aligned(16))) f32x4 lds[];  // [2][kSlab][64] + zero row [64]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // XCD-contiguous work order (workgroup b runs on XCD b % 8): XCD x takes a contiguous eighth of
  // the (column tile, row block) list, tile-major, so the workgroups streaming one K × 256 slice of
  // B share an L2 — and every XCD gets the same amount of work whatever the number of column
  // tiles.  Bijective for any grid size; speed only.
  const unsigned total = gridDim.x, q8 = total / 8, rem = total % 8, xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
  const unsigned work = xcd * q8 + (xcd < rem ? xcd : rem) + slot;
  const int tile = (int)(work / row_blocks);
  const unsigned rb = work % row_blocks;
  const int c0 = tile * kTileCols;
  const int ncols = N - c0 < kTileCols ? N - c0 : kTileCols;
  const bool on = lane * 4 < ncols;  // N % 4 == 0: a lane's four columns are all in or all out
  const int row0 = (int)rb * (kWaves * kRows) + wave * kRows;
  constexpr int kRowVec = TR ? 65 : 64;  // float4 per LDS row (TR: padded, see above)
  constexpr int kZeroRow = 2 * kSlab;    // row index of the zero row
  if (wave == 0) lds[kZeroRow * kRowVec + lane] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- per-row state: window of 64 entries, cursor, accumulators
  int base[kRows], endp[kRows], pos[kRows], next[kRows];
  int wcol[kRows];
  float wval[kRows];
  f32x4 acc[kRows];
  unsigned unsorted = 0;  // bit i: row i's columns do not ascend → recomputed at the end
  unsigned skip = 0;      // bit i: row beyond M, or left to the long-row kernel
  auto load_window = [&](int i, int prev_last) {
    const int e = base[i] + lane;
    const bool v = e < endp[i];
    wcol[i] = v ? col[e] : INT_MAX;
    wval[i] = v ? val[e] : 0.f;
    // ascending inside the window and across the window boundary?
    int before = __shfl_up(wcol[i], 1, 64);
    if (lane == 0) before = prev_last;
    if (__any(v && wcol[i] < before)) unsorted |= 1u << i;
  };
#pragma unroll
  for (int i = 0; i < kRows; ++i) {
    const int r = row0 + i;
    acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    int s = 0, e = 0;
    if (r < M) {
      s = rowptr[r];
      e = rowptr[r + 1];
      if (e - s > long_thresh) {
        e = s;
        skip |= 1u << i;
      }
    } else {
      skip |= 1u << i;
    }
    base[i] = s;
    endp[i] = e;
    pos[i] = 0;
    load_window(i, INT_MIN);
    next[i] = ((unsorted >> i) & 1u) ? INT_MAX : __builtin_amdgcn_readlane(wcol[i], 0);
  }

  // ---- slab staging: kSlab rows × 64 float4 = kStage float4 per thread (slab row = wave + 16·v),
  // two register sets so a slab's loads have two whole iterations to land
  const int nslab = (K + kSlab - 1) / kSlab;
  const float* bcol = B + c0 + (on ? lane * 4 : 0);
  auto load_slab = [&](f32x4 (&st)[kStage], int s) {
#pragma unroll
    for (int v = 0; v < kStage; ++v) {
      if (TR) {
        // wave w, lane l, piece v: row n = 16·w + (l & 15) of X (column of B), k quad = 4·v + (l >> 4)
        int n = c0 + wave * 16 + (lane & 15);
        n = n < N ? n : N - 1;                 // columns past N belong to lanes that are off
        const int k0 = s * kSlab + (4 * v + (lane >> 4)) * 4;
        // a quad wholly past K is never referenced (any valid address will do); the PARTIAL last quad (K % 4 != 0) is
        // read as X[n][K-4 … K-1] — the padding behind the last row of X need not exist — and shifted so that
        // component j still holds k0 + j, which is where store_slab puts it (round 2 stored the clamped quad at the
        // rows of k0: wrong B values for the last K % 4 columns of A whenever ldb was padded beyond K)
        const int kk = k0 + 3 < K ? k0 : (K >= 4 ? K - 4 : 0);
        f32x4 x = *reinterpret_cast<const f32x4*>(B + (long)n * ldb + kk);
        if (k0 < K && k0 + 3 >= K && K >= 4) {
          const int d = k0 - kk;  // 1 … 3: component j of the wanted quad is component j + d of the loaded one
          x = d == 1 ? f32x4{x.y, x.z, x.w, 0.f} : d == 2 ? f32x4{x.z, x.w, 0.f, 0.f} : f32x4{x.w, 0.f, 0.f, 0.f};
        }
        st[v] = x;
      } else {
        int kr = s * kSlab + wave + kWaves * v;
        kr = kr < K ? kr : K - 1;  // rows past K are never referenced: any valid address will do
        st[v] = *reinterpret_cast<const f32x4*>(bcol + (long)kr * ldb);
      }
    }
  };
  auto store_slab = [&](const f32x4 (&st)[kStage], int buf) {
#pragma unroll
    for (int v = 0; v < kStage; ++v) {
      if (TR) {
        const int nl = wave * 16 + (lane & 15), kq = 4 * v + (lane >> 4);  // local column, k quad
        float* base = reinterpret_cast<float*>(lds);
        const float comp[4] = {st[v].x, st[v].y, st[v].z, st[v].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) base[(buf * kSlab + 4 * kq + j) * (kRowVec * 4) + nl] = comp[j];
      } else {
        lds[(buf * kSlab + wave + kWaves * v) * kRowVec + lane] = st[v];
      }
    }
  };
  // this lane's 4-column group of LDS row r
  auto at = [&](int r) { return lds[r * kRowVec + lane]; };

  // One slab: every row consumes its entries with column < slab_end from LDS buffer `cur`, then
  // the registers holding slab s+1 go to the other buffer and are refilled with slab s+3.
  auto do_slab = [&](int s, int cur, f32x4 (&st)[kStage]) {
    const int k0 = s * kSlab;
    const int slab_end = k0 + kSlab;
    const int rowbase = cur * kSlab - k0;  // LDS row of column c: rowbase + c
#pragma unroll
    for (int i = 0; i < kRows; ++i) {
      // next[i] = column at the row's cursor (INT_MAX when the row is used up or set aside): a
      // scalar compare decides whether the row has anything in this slab
      while (next[i] < slab_end) {
        // entries of this row inside the slab: a run of lanes starting at the cursor
        const bool in = lane >= pos[i] && wcol[i] < slab_end;
        const int n = __popcll(__ballot(in));
        int j = 0;
        for (; j + 4 <= n; j += 4) {  // four B rows in flight
          int r[4];
          float v[4];
          f32x4 x[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            r[u] = rowbase + __builtin_amdgcn_readlane(wcol[i], pos[i] + j + u);
            v[u] = readlane_f(wval[i], pos[i] + j + u);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) x[u] = at(r[u]);
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[i] = fma4(v[u], x[u], acc[i]);
        }
        // the last 1–3, two at a time; an odd count is padded with the product (-0.0)·(+0.0) from
        // the zero row, which leaves every bit of the accumulator as it is (x + (-0) = x, also ±0)
        for (; j < n; j += 2) {
          const int p0 = pos[i] + j, p1 = p0 + 1 < 63 ? p0 + 1 : 63;
          const bool two = j + 1 < n;
          const int r0 = rowbase + __builtin_amdgcn_readlane(wcol[i], p0);
          const int c1 = __builtin_amdgcn_readlane(wcol[i], p1);
          const int r1 = two ? rowbase + c1 : kZeroRow;
          const float v0 = readlane_f(wval[i], p0);
          const float v1r = readlane_f(wval[i], p1);
          const float v1 = two ? v1r : -0.0f;
          const f32x4 x0 = at(r0);
          const f32x4 x1 = at(r1);
          acc[i] = fma4(v0, x0, acc[i]);
          acc[i] = fma4(v1, x1, acc[i]);
        }
        pos[i] += n;
        if (pos[i] < 64) {
          next[i] = __builtin_amdgcn_readlane(wcol[i], pos[i]);  // INT_MAX beyond the row's end
          break;
        }
        if (base[i] + 64 >= endp[i]) {
          next[i] = INT_MAX;
          break;
        }
        // window used up and the row goes on: next 64 entries (may continue in this slab)
        const int last = __builtin_amdgcn_readlane(wcol[i], 63);
        base[i] += 64;
        pos[i] = 0;
        load_window(i, last);
        next[i] = ((unsorted >> i) & 1u) ? INT_MAX : __builtin_amdgcn_readlane(wcol[i], 0);
      }
    }
    if (s + 1 < nslab) {
      store_slab(st, cur ^ 1);  // its last readers passed the previous barrier
      if (s + 3 < nslab) load_slab(st, s + 3);
    }
    __syncthreads();
  };

  f32x4 sta[kStage], stb[kStage];
  load_slab(sta, 0);
  store_slab(sta, 0);
  if (nslab > 2) load_slab(sta, 2);  // slab t travels in sta for even t, stb for odd t
  if (nslab > 1) load_slab(stb, 1);
  __syncthreads();
  for (int s = 0; s < nslab; s += 2) {
    do_slab(s, 0, stb);
    if (s + 1 < nslab) do_slab(s + 1, 1, sta);
  }

  // ---- rows whose columns do not ascend: the plain CSR-order chain from global memory
  if (unsorted) {
#pragma unroll
    for (int i = 0; i < kRows; ++i) {
      if (!((unsorted >> i) & 1u)) continue;
      const int r = row0 + i;
      f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
      const int s = rowptr[r], e = rowptr[r + 1];
      for (int p = s; p < e; p += 64) {
        const int q = p + lane;
        const int myc = q < e ? col[q] : 0;
        const float myv = q < e ? val[q] : 0.f;
        const int cnt = e - p < 64 ? e - p : 64;
        for (int j = 0; j < cnt; ++j) {
          const int cc = __builtin_amdgcn_readlane(myc, j);
          const float vv = readlane_f(myv, j);
          if (on) {
            f32x4 bx;
            if (TR) {
              const float* xp = B + (long)(c0 + lane * 4) * ldb + cc;
              bx = f32x4{xp[0], xp[ldb], xp[2 * ldb], xp[3 * ldb]};
            } else {
              bx = *reinterpret_cast<const f32x4*>(bcol + (long)cc * ldb);
            }
            a = fma4(vv, bx, a);
          }
        }
      }
      acc[i] = a;
    }
  }

  // ---- epilogue
  f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
  if (bias && on) bv = *reinterpret_cast<const f32x4*>(bias + c0 + lane * 4);
  if (TR) {
    // Y[n][m]: stage the tile as [n][128] floats (m contiguous) in the slab buffers — the 4-row groups of
    // a column n sit at position q XOR ((n >> 2) & 31), so the 64 lanes of a staging write (64 different
    // n, the same q) spread over the banks — then every 32 lanes write one 512-byte row segment of Y
    constexpr int SLD = kWaves * kRows;  // 128
    float* tile = reinterpret_cast<float*>(lds);
    __syncthreads();  // (the k loop ended with a barrier; rows recomputed above did not touch LDS)
    if (on) {
      const float comp[kRows][4] = {{acc[0].x, acc[0].y, acc[0].z, acc[0].w}, {acc[1].x, acc[1].y, acc[1].z, acc[1].w},
                                    {acc[2].x, acc[2].y, acc[2].z, acc[2].w}, {acc[3].x, acc[3].y, acc[3].z, acc[3].w},
                                    {acc[4].x, acc[4].y, acc[4].z, acc[4].w}, {acc[5].x, acc[5].y, acc[5].z, acc[5].w},
                                    {acc[6].x, acc[6].y, acc[6].z, acc[6].w}, {acc[7].x, acc[7].y, acc[7].z, acc[7].w}};
      const float bcomp[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int n = lane * 4 + c, sw = (n >> 2) & 31;
        float* row = tile + n * SLD;
        const float b = bias ? bcomp[c] : 0.f;
        *reinterpret_cast<f32x4*>(row + 4 * ((wave * 2) ^ sw)) = f32x4{comp[0][c] + b, comp[1][c] + b, comp[2][c] + b, comp[3][c] + b};
        *reinterpret_cast<f32x4*>(row + 4 * ((wave * 2 + 1) ^ sw)) = f32x4{comp[4][c] + b, comp[5][c] + b, comp[6][c] + b, comp[7][c] + b};
      }
    }
    __syncthreads();
    const int mbase = (int)rb * (kWaves * kRows);
    // 32 lanes cover the 128 rows (m) of one n: thread t → n = t / 32 + 32·pass, m quad = t % 32
    for (int nn = (int)threadIdx.x / 32; nn < ncols; nn += kWaves * 2) {
      const int mq = (int)threadIdx.x % 32, mg = mq * 4;
      const f32x4 v = *reinterpret_cast<const f32x4*>(tile + nn * SLD + 4 * (mq ^ ((nn >> 2) & 31)));
      float* yrow = C + (long)(c0 + nn) * ldc + mbase + mg;
      if (mbase + mg + 3 < M) {
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(yrow));
      } else {
        if (mbase + mg + 0 < M) yrow[0] = v.x;
        if (mbase + mg + 1 < M) yrow[1] = v.y;
        if (mbase + mg + 2 < M) yrow[2] = v.z;
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < kRows; ++i) {
    if ((skip >> i) & 1u) continue;
    if (on) {
      f32x4 o = acc[i];
      if (bias) o += bv;
      __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(C + (long)(row0 + i) * ldc + c0 + lane * 4));
    }
  }
}

}  // namespace

namespace mi {

// Requirements (checked by the caller, spmm_csr.hip): N % 4 == 0, ldb % 4 == 0, ldc % 4 == 0,
// B / C / bias 16-byte aligned, K > 0.
int launch_spmm_slab(const int32_t* rowptr, const int32_t* col, const float* val, const float* B, float* C,
                     int32_t M, int32_t K, int32_t N, int64_t ldb, int64_t ldc, const float* bias,
                     int long_thresh, hipStream_t s) {
  const long row_blocks = ((long)M + kWaves * kRows - 1) / (kWaves * kRows);
  const int ctiles = (N + kTileCols - 1) / kTileCols;
  const long blocks = (long)ctiles * row_blocks;
  if (blocks > 0x7fffffffL) return MI_ERANGE;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(spmm_slab_kernel<false>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
  if (attr != hipSuccess) return record_hip_error(attr);
  hipLaunchKernelGGL(spmm_slab_kernel<false>, dim3((unsigned)blocks), dim3(kWaves * 64), kLdsBytes, s, rowptr, col, val, B,
                     C, M, K, N, (long)ldb, (long)ldc, bias, ctiles, (unsigned)row_blocks, long_thresh);
  return check_launch();
}

// Column-major operands: X = Bᵀ row-major [N, ldx ≥ K], Y = Cᵀ row-major [N, ldy ≥ M] (see the kernel's TR
// notes).  Requirements (checked by the caller, convert.hip): N % 4 == 0, K ≥ 4, ldx % 4 == 0, ldy % 4 == 0,
// X / Y 16-byte aligned, no row left to the long-row kernel (it writes row-major C).
int launch_spmm_slab_colmajor(const int32_t* rowptr, const int32_t* col, const float* val, const float* X, float* Y,
                              int32_t M, int32_t K, int32_t N, int64_t ldx, int64_t ldy, hipStream_t s) {
  const long row_blocks = ((long)M + kWaves * kRows - 1) / (kWaves * kRows);
  const int ctiles = (N + kTileCols - 1) / kTileCols;
  const long blocks = (long)ctiles * row_blocks;
  if (blocks > 0x7fffffffL) return MI_ERANGE;
  constexpr size_t kLdsBytesTr = (size_t)(2 * kSlab * 65 + 65) * sizeof(f32x4);  // padded rows
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(spmm_slab_kernel<true>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytesTr);
  if (attr != hipSuccess) return record_hip_error(attr);
  hipLaunchKernelGGL(spmm_slab_kernel<true>, dim3((unsigned)blocks), dim3(kWaves * 64), kLdsBytesTr, s, rowptr, col, val, X,
                     Y, M, K, N, (long)ldx, (long)ldy, (const float*)nullptr, ctiles, (unsigned)row_blocks, 0x7fffffff);
  return check_launch();
}

}  // namespace mi
