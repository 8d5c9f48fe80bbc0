// Row-split CSR × dense SpMM for gfx950 (MI355X), fp32.
//
// What it computes (contract of include/mi_spmm.h, mi_spmm_csr_f32):
//   C[r, j] = Σ_{p = rowptr[r]}^{rowptr[r+1]-1} val[p] · B[col[p], j]
// accumulated per output element with fused multiply-add in CSR order — the
// per-element summation order of the reference's spmm_kernel
// (src/naive_sparse_mm.cu:60-92), so results do not depend on launch geometry
// and the row-sharded multi-GPU result is bit-identical to the 1-GPU one.
//
// Design (not a port of the reference's 32-lane, 4-byte-per-lane kernel):
//  * wave64 owns whole output rows; a lane owns 16 B (float4) of the row, so a
//    B-row gather for N = 256 is exactly one global_load_dwordx4 wave
//    instruction (1 KiB, fully coalesced) and col/val are read ONCE per row
//    (the reference re-reads them ⌈N/32⌉ times, naive_sparse_mm.cu:39,116).
//  * col/val of a row are wave-uniform, so the wave-per-row kernels read them
//    through the scalar unit (s_load) and form the B-row address as
//    scalar-base + lane-offset; no shuffles (the reference spends 64
//    __shfl_sync per 32 nonzeros, naive_sparse_mm.cu:75-80).
//  * the kernel is a random 1 KiB-row gather from a table far larger than the
//    256 MiB Infinity Cache: memory-level parallelism is what matters, so U
//    independent B-row loads are issued before the first FMA consumes one.
//  * C rows are written once with non-temporal 16-B stores.
//  * narrower N: G = N/4 lanes per row and 64/G rows per wave (col/val are
//    broadcast inside the G-lane group with ds_bpermute); arbitrary N or
//    unaligned operands: the same kernel with one float per lane.
#include "spmm_device.h"
#include "spmm_internal.h"

namespace {

using mi::LongArg;
using mi::Shape;

// ---------------------------------------------------------------------------
// One wave per row, N == 256·T exactly.  col/val through the scalar unit.
// grid = (⌈M/4⌉, batch), block = 256 (4 waves = 4 rows).
// ---------------------------------------------------------------------------
template <int T, int U>
__global__ __launch_bounds__(256) void spmm_wave_row_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ col,
    const float* __restrict__ val, const float* __restrict__ B, float* __restrict__ C,
    int M, long ldb, long ldc, long strideB, long strideC, const float* __restrict__ bias,
    LongArg la) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long slot = (long)blockIdx.x * 4 + wave;
  if (slot >= (la.order ? la.nslots : M)) return;
  const long row = la.order ? la.order[slot] : slot;  // (wave-uniform: a scalar load) a schedule's slot → row map
  const long item = blockIdx.y;
  const int* rp = rowptr + item * ((long)M + 1);
  const float* Bl = B + item * strideB + lane * 4;
  float* Cl = C + item * strideC + row * ldc + lane * 4;

  int p = rp[row];
  const int end = rp[row + 1];
  if (end - p > la.thresh) {  // left to the listed-rows launch (spmm_heavy.hip)
    if (lane == 0) long_list_append(la, (int)row, end - p);
    return;
  }

  f32x4 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (; p + U <= end; p += U) {
    int c[U];
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      c[u] = col[p + u];
      v[u] = val[p + u];
    }
    f32x4 x[U][T];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float* src = Bl + (long)c[u] * ldb;
#pragma unroll
      for (int t = 0; t < T; ++t) x[u][t] = *reinterpret_cast<const f32x4*>(src + t * 256);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int t = 0; t < T; ++t) acc[t] = fma4(v[u], x[u][t], acc[t]);
    }
  }
  row_tail<T, U>(col, val, Bl, ldb, p, end, acc);
  if (bias) {  // fused epilogue: + bias[j] after the chain (one extra rounding, like `out += bias`)
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] += *reinterpret_cast<const f32x4*>(bias + lane * 4 + t * 256);
  }
#pragma unroll
  for (int t = 0; t < T; ++t)
    __builtin_nontemporal_store(acc[t], reinterpret_cast<f32x4*>(Cl + t * 256));
}

// ---------------------------------------------------------------------------
// The same chain, result written COLUMN-major (Ccm[n·ldc + m]) — the column-major executor's output
// transpose fused into the epilogue (convert.hip: mi_spmm_csr_colmajor_ex_f32).  16 waves = 16
// consecutive rows per workgroup; the 16 × N results meet in LDS (rows padded by 4 floats: the four
// row-quads a store instruction reads sit 16 banks apart) and leave as N pieces of 16 consecutive m
// (64 bytes; one float4 per thread).  grid = ⌈M/16⌉, block = 1024, LDS = 16·(N+4)·4 bytes.
// ---------------------------------------------------------------------------
template <int T, int U>
__global__ __launch_bounds__(1024) void spmm_wave_row_ct_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ val,
    const float* __restrict__ B, float* __restrict__ Ccm, int M, long ldb, long ldc, int vec_ok) {
  extern __shared__ __attribute__((aligned(16))) float ct_tile[];
  constexpr int NP = 256 * T + 4;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long m0 = (long)blockIdx.x * 16;
  const long row = m0 + wave;
  f32x4 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (row < M) {
    const float* Bl = B + lane * 4;
    int p = rowptr[row];
    const int end = rowptr[row + 1];
    for (; p + U <= end; p += U) {
      int c[U];
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        c[u] = col[p + u];
        v[u] = val[p + u];
      }
      f32x4 x[U][T];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float* src = Bl + (long)c[u] * ldb;
#pragma unroll
        for (int t = 0; t < T; ++t) x[u][t] = *reinterpret_cast<const f32x4*>(src + t * 256);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = fma4(v[u], x[u][t], acc[t]);
      }
    }
    for (; p < end; ++p) {
      const int c = col[p];
      const float v = val[p];
      const float* src = Bl + (long)c * ldb;
#pragma unroll
      for (int t = 0; t < T; ++t)
        acc[t] = fma4(v, *reinterpret_cast<const f32x4*>(src + t * 256), acc[t]);
    }
  }
#pragma unroll
  for (int t = 0; t < T; ++t) *reinterpret_cast<f32x4*>(ct_tile + wave * NP + t * 256 + lane * 4) = acc[t];
  __syncthreads();
  const int q = threadIdx.x & 3;
  const long m = m0 + 4 * q;
#pragma unroll
  for (int i = 0; i < T; ++i) {
    const int n = (threadIdx.x >> 2) + 256 * i;
    const float* src = ct_tile + (4 * q) * NP + n;
    const f32x4 o = {src[0], src[NP], src[2 * NP], src[3 * NP]};
    float* dst = Ccm + (long)n * ldc + m;
    if (vec_ok && m + 3 < M) {
      __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(dst));
    } else {
      if (m < M) dst[0] = o.x;
      if (m + 1 < M) dst[1] = o.y;
      if (m + 2 < M) dst[2] = o.z;
      if (m + 3 < M) dst[3] = o.w;
    }
  }
}

// ---------------------------------------------------------------------------
// One wave per row, N == 256, col/val fetched 64 at a time with one coalesced
// vector load each and handed to the scalar unit with v_readlane.
// ---------------------------------------------------------------------------
template <int U>
__global__ __launch_bounds__(256) void spmm_wave_row_vl_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ col,
    const float* __restrict__ val, const float* __restrict__ B, float* __restrict__ C,
    int M, long ldb, long ldc, long strideB, long strideC, const float* __restrict__ bias,
    LongArg la) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long row = (long)blockIdx.x * 4 + wave;
  if (row >= M) return;
  const long item = blockIdx.y;
  const int* rp = rowptr + item * ((long)M + 1);
  const float* Bl = B + item * strideB + lane * 4;
  float* Cl = C + item * strideC + row * ldc + lane * 4;

  const int start = rp[row];
  const int end = rp[row + 1];
  if (end - start > la.thresh) {  // left to the listed-rows launch (spmm_heavy.hip)
    if (lane == 0) long_list_append(la, (int)row, end - start);
    return;
  }
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int p = start; p < end; p += 64) {
    const int idx = p + lane;
    const int myc = idx < end ? col[idx] : 0;
    const float myv = idx < end ? val[idx] : 0.f;
    const int cnt = (end - p) < 64 ? (end - p) : 64;  // wave-uniform
    int i = 0;
    for (; i + U <= cnt; i += U) {
      f32x4 x[U];
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = __builtin_amdgcn_readlane(myc, i + u);
        v[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myv), i + u));
        x[u] = *reinterpret_cast<const f32x4*>(Bl + (long)c * ldb);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc = fma4(v[u], x[u], acc);
    }
    for (; i < cnt; ++i) {
      const int c = __builtin_amdgcn_readlane(myc, i);
      const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myv), i));
      acc = fma4(v, *reinterpret_cast<const f32x4*>(Bl + (long)c * ldb), acc);
    }
  }
  if (bias) acc += *reinterpret_cast<const f32x4*>(bias + lane * 4);
  __builtin_nontemporal_store(acc, reinterpret_cast<f32x4*>(Cl));
}

// ---------------------------------------------------------------------------
// G lanes per row (G a power of two ≤ 64), 64/G rows per wave, VEC floats per
// lane per tile, T tiles per pass; columns beyond G·VEC·T are covered by an
// outer pass loop (col/val re-read once per pass).  Handles every N.
// grid = (⌈M / (4·64/G)⌉, batch), block = 256.
// ---------------------------------------------------------------------------
template <int G, int VEC, int T>
__global__ __launch_bounds__(256) void spmm_group_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ col,
    const float* __restrict__ val, const float* __restrict__ B, float* __restrict__ C,
    int M, int N, long ldb, long ldc, long strideB, long strideC, const float* __restrict__ bias,
    int ctiles, int tile_cols, unsigned row_blocks, LongArg la) {
  typedef Vec<VEC> V;
  typedef typename V::type vec_t;
  constexpr int RPW = 64 / G;  // rows per wave
  constexpr int UI = G < 4 ? G : 4;  // B-row loads in flight per group
  const int lane = threadIdx.x & 63;
  const int gl = lane & (G - 1);
  const long item = blockIdx.y;
  const int* rp = rowptr + item * ((long)M + 1);
  const float* Bi = B + item * strideB;
  float* Ci = C + item * strideC;
  unsigned rb = blockIdx.x;
  bool lists = true;  // the first column tile lists the long rows it skips
  if (ctiles > 1) {
    // XCD-aware column tiling (wide N): workgroup b runs on XCD b % 8, whose private 4 MiB L2
    // should hold the K × tile_cols slice of B it gathers from.  XCD x takes column tiles
    // x, x+8, x+16, … one after the other, all row blocks of a tile before the next tile, so the
    // CUs of an XCD share one slice at a time.  Per-element arithmetic is unchanged (each output
    // element still sees its row's non-zeros in CSR order); placement affects speed only.
    const unsigned xcd = blockIdx.x % 8, idx = blockIdx.x / 8;
    const int tile = (int)(xcd + 8 * (idx / row_blocks));
    if (tile >= ctiles) return;
    lists = tile == 0;
    rb = idx % row_blocks;
    const int c0 = tile * tile_cols;
    Bi += c0;
    Ci += c0;
    if (bias) bias += c0;
    N = N - c0 < tile_cols ? N - c0 : tile_cols;
  }
  const long slot = ((long)rb * 4 + (threadIdx.x >> 6)) * RPW + (lane / G);
  const bool live = slot < (la.order ? la.nslots : M);
  const long row = (la.order && live) ? la.order[slot] : slot;  // a schedule's slot → row map

  int start = 0, end = 0;
  if (live) {
    start = rp[row];
    end = rp[row + 1];
  }
  const bool skipped = end - start > la.thresh;  // left to the listed-rows launch (spmm_heavy.hip)
  if (skipped) {
    if (lists && gl == 0 && item == 0) long_list_append(la, (int)row, end - start);
    end = start;
  }

  constexpr int W = V::width;  // floats per lane and tile
  for (int n0 = 0; n0 < N; n0 += G * W * T) {  // wave-uniform pass loop
    vec_t acc[T];
    bool on[T];
    int coff[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      acc[t] = V::zero();
      coff[t] = n0 + (t * G + gl) * W;
      on[t] = coff[t] < N;
      if (VEC == 44 && on[t] && coff[t] + 4 > N) coff[t] = N - 4;  // the partial last quad, shifted back (N ≥ 4: the launcher checks)
    }
    if constexpr (G < 4) {
      // 1- or 2-lane groups: a chunk of G entries would leave only G gathers in flight, so take
      // 4 entries (4/G chunks) per round; entry e of the round sits in chunk e/G, lane e%G.
      constexpr int CHK = 4 / G;
      for (int p = start; p < end; p += 4) {  // trip count differs between groups
        int myc[CHK];
        float myv[CHK];
#pragma unroll
        for (int u = 0; u < CHK; ++u) {
          const int idx = p + u * G + gl;
          myc[u] = idx < end ? col[idx] : 0;
          myv[u] = idx < end ? val[idx] : 0.f;
        }
        const int cnt = end - p;  // entries of this round that exist (≥ 4 except in the last round)
        vec_t x[4][T];
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = __shfl(myc[e / G], e % G, G);
          v[e] = __shfl(myv[e / G], e % G, G);
          const float* src = Bi + (long)c * ldb;
#pragma unroll
          for (int t = 0; t < T; ++t)
            if (on[t] && e < cnt) x[e][t] = V::load(src + coff[t]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int t = 0; t < T; ++t)
            if (on[t] && e < cnt) acc[t] = V::fma(v[e], x[e][t], acc[t]);
        }
      }
    } else {
    // a chunk of EPC entries per coalesced load, handed round the group with compile-time lane indices so that the
    // broadcast is a DPP modifier / a scalar readlane instead of two ds_bpermute per non-zero (mi_lanes.h)
    constexpr int EPC = mi::LaneChunk<G, false>::ENTRIES;
    for (int p = start; p < end; p += EPC) {  // trip count differs between groups
      const int idx = p + (gl & (EPC - 1));
      const int myc = idx < end ? col[idx] : 0;
      const float myv = idx < end ? val[idx] : 0.f;
      const int cnt = (end - p) < EPC ? (end - p) : EPC;  // group-uniform
      mi::static_for<EPC / UI>([&](auto b_) {
        constexpr int b = UI * decltype(b_)::value;
        if (b + UI <= cnt) {
          vec_t x[UI][T];
          float v[UI];
          mi::static_for<UI>([&](auto u_) {
            constexpr int u = decltype(u_)::value;
            const int c = mi::group_lane<G, b + u, false>(myc);
            v[u] = mi::group_lane<G, b + u, false>(myv);
            const float* src = Bi + (long)c * ldb;
#pragma unroll
            for (int t = 0; t < T; ++t)
              if (on[t]) x[u][t] = V::load(src + coff[t]);
          });
#pragma unroll
          for (int u = 0; u < UI; ++u) {
#pragma unroll
            for (int t = 0; t < T; ++t)
              if (on[t]) acc[t] = V::fma(v[u], x[u][t], acc[t]);
          }
        } else if (b < cnt) {
          mi::static_for<UI>([&](auto u_) {
            constexpr int u = decltype(u_)::value;
            if (b + u < cnt) {
              const int c = mi::group_lane<G, b + u, false>(myc);
              const float v = mi::group_lane<G, b + u, false>(myv);
              const float* src = Bi + (long)c * ldb;
#pragma unroll
              for (int t = 0; t < T; ++t)
                if (on[t]) acc[t] = V::fma(v, V::load(src + coff[t]), acc[t]);
            }
          });
        }
      });
    }
    }
    if (live && !skipped) {
      float* dst = Ci + row * ldc;
#pragma unroll
      for (int t = 0; t < T; ++t)
        if (on[t]) V::store(dst + coff[t], bias ? acc[t] + V::load(bias + coff[t]) : acc[t]);
    }
  }
}

// rows a launch covers: all M, or the slots of a schedule
inline long slots_of(const LongArg& la, int M) { return la.order ? la.nslots : M; }

template <int G, int VEC, int T>
int launch_group(const int* rowptr, const int* col, const float* val, const float* B,
                 float* C, int M, int N, long ldb, long ldc, long strideB, long strideC,
                 int batch, const float* bias, LongArg la, hipStream_t s) {
  constexpr int rows_per_block = 4 * (64 / G);
  const long blocks = (slots_of(la, M) + rows_per_block - 1) / rows_per_block;
  if (blocks > 0x7fffffffL) return MI_ERANGE;
  if (blocks == 0) return MI_OK;
  hipLaunchKernelGGL((spmm_group_kernel<G, VEC, T>), dim3((unsigned)blocks, (unsigned)batch),
                     dim3(256), 0, s, rowptr, col, val, B, C, M, N, ldb, ldc, strideB, strideC, bias, 1, N,
                     (unsigned)blocks, la);
  return mi::check_launch();
}

// Column-tiled launch of the float4 group kernel: tile_cols = 4·G columns per tile.
template <int G>
int launch_coltile(const int* rowptr, const int* col, const float* val, const float* B, float* C, int M,
                   int N, long ldb, long ldc, const float* bias, LongArg la, hipStream_t s) {
  constexpr int rows_per_block = 4 * (64 / G);
  constexpr int tile_cols = 4 * G;
  const long row_blocks = ((long)M + rows_per_block - 1) / rows_per_block;
  const int ctiles = (N + tile_cols - 1) / tile_cols;
  const long blocks = 8L * ((ctiles + 7) / 8) * row_blocks;  // every XCD gets the same count; extras exit
  if (blocks > 0x7fffffffL) return MI_ERANGE;
  hipLaunchKernelGGL((spmm_group_kernel<G, 4, 1>), dim3((unsigned)blocks, 1u), dim3(256), 0, s, rowptr, col, val,
                     B, C, M, N, ldb, ldc, 0L, 0L, bias, ctiles, tile_cols, (unsigned)row_blocks, la);
  return mi::check_launch();
}

template <int VEC>
int dispatch_group(const int* rowptr, const int* col, const float* val, const float* B,
                   float* C, int M, int N, long ldb, long ldc, long strideB, long strideC,
                   int batch, const float* bias, LongArg la, hipStream_t s) {
  const int nv = (N + VEC - 1) / VEC;  // vector columns
  const int G = nv >= 64 ? 64 : mi::pow2_ceil(nv);
#define MI_GROUP(G_, T_) \
  return launch_group<G_, VEC, T_>(rowptr, col, val, B, C, M, N, ldb, ldc, strideB, strideC, batch, bias, \
                                   la, s)
  switch (G) {
    case 1: MI_GROUP(1, 1);
    case 2: MI_GROUP(2, 1);
    case 4: MI_GROUP(4, 1);
    case 8: MI_GROUP(8, 1);
    case 16: MI_GROUP(16, 1);
    case 32: MI_GROUP(32, 1);
    default: break;
  }
  const int tiles = (nv + 63) / 64;
  if (tiles <= 1) MI_GROUP(64, 1);
  if (tiles == 2) MI_GROUP(64, 2);
  MI_GROUP(64, 4);
#undef MI_GROUP
}

// any N ≥ 4 at any 4-byte alignment: quads per row = ⌈N / 4⌉ (the last one shifted back when N % 4 != 0)
int dispatch_group_u4(const int* rowptr, const int* col, const float* val, const float* B, float* C, int M, int N, long ldb,
                      long ldc, long strideB, long strideC, int batch, const float* bias, LongArg la, hipStream_t s) {
  const int nv = (N + 3) / 4;
  const int G = nv >= 64 ? 64 : mi::pow2_ceil(nv);
#define MI_GROUP_U(G_, T_) \
  return launch_group<G_, 44, T_>(rowptr, col, val, B, C, M, N, ldb, ldc, strideB, strideC, batch, bias, la, s)
  switch (G) {
    case 1: MI_GROUP_U(1, 1);
    case 2: MI_GROUP_U(2, 1);
    case 4: MI_GROUP_U(4, 1);
    case 8: MI_GROUP_U(8, 1);
    case 16: MI_GROUP_U(16, 1);
    case 32: MI_GROUP_U(32, 1);
    default: break;
  }
  const int tiles = (nv + 63) / 64;
  if (tiles <= 1) MI_GROUP_U(64, 1);
  if (tiles == 2) MI_GROUP_U(64, 2);
  MI_GROUP_U(64, 4);
#undef MI_GROUP_U
}

template <int T, int U>
int launch_wave_row(const int* rowptr, const int* col, const float* val, const float* B,
                    float* C, int M, long ldb, long ldc, long strideB, long strideC,
                    int batch, const float* bias, LongArg la, hipStream_t s) {
  const long blocks = (slots_of(la, M) + 3) / 4;
  if (blocks > 0x7fffffffL) return MI_ERANGE;
  if (blocks == 0) return MI_OK;
  hipLaunchKernelGGL((spmm_wave_row_kernel<T, U>), dim3((unsigned)blocks, (unsigned)batch),
                     dim3(256), 0, s, rowptr, col, val, B, C, M, ldb, ldc, strideB, strideC, bias, la);
  return mi::check_launch();
}

template <int U>
int launch_wave_row_vl(const int* rowptr, const int* col, const float* val, const float* B,
                       float* C, int M, long ldb, long ldc, long strideB, long strideC,
                       int batch, const float* bias, LongArg la, hipStream_t s) {
  const long blocks = ((long)M + 3) / 4;
  if (blocks > 0x7fffffffL) return MI_ERANGE;
  hipLaunchKernelGGL((spmm_wave_row_vl_kernel<U>), dim3((unsigned)blocks, (unsigned)batch),
                     dim3(256), 0, s, rowptr, col, val, B, C, M, ldb, ldc, strideB, strideC, bias, la);
  return mi::check_launch();
}

// ---------------------------------------------------------------------------
// Narrow outputs, N < 4 (SpMV-like; matmuls promotes `A @ vector` to N = 1).  Lanes-over-columns
// leaves a row to a single lane there, so this kernel turns the wave around: one wave per row,
// lane l walks the row's non-zeros l, l+64, l+128, … (coalesced col/val loads, one 4-byte gather
// of B per non-zero and column) keeping N partial sums, and a xor-butterfly (32, 16, …, 1) adds the
// 64 partial rows.  This is the classic wavefront shuffle reduction; its summation order (64
// lane-strided chains, then the butterfly) is fixed and stated in oracle_spmm_csr_f32, so results
// stay bit-identical to the oracle.  grid = (⌈M/4⌉, batch).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void spmm_narrow_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ val,
    const float* __restrict__ B, float* __restrict__ C, int M, int N, long ldb, long ldc, long strideB,
    long strideC, const float* __restrict__ bias) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const long item = blockIdx.y;
  const int* rp = rowptr + item * ((long)M + 1);
  const float* Bi = B + item * strideB;
  const int start = rp[row], end = rp[row + 1];
  float acc[3] = {0.f, 0.f, 0.f};
  for (int p = start + lane; p < end; p += 64) {
    const float v = val[p];
    const float* brow = Bi + (long)col[p] * ldb;
#pragma unroll
    for (int j = 0; j < 3; ++j)
      if (j < N) acc[j] = __builtin_fmaf(v, brow[j], acc[j]);
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    if (j < N) {
      float s = acc[j];
#pragma unroll
      for (int w = 32; w >= 1; w >>= 1) s += __shfl_xor(s, w, 64);
      if (lane == 0) C[item * strideC + row * ldc + j] = bias ? s + bias[j] : s;
    }
  }
}

int launch_variant(int variant, const Shape& sh, const int32_t* rowptr, const int32_t* col, const float* val,
                   int64_t nnz, int32_t batch, int32_t M, int32_t K, int32_t N, const float* B,
                   int64_t ldb, int64_t strideB, float* C, int64_t ldc, int64_t strideC,
                   const float* bias, LongArg la, hipStream_t s) {
  const bool vec4_ok = sh.vec4_ok, vec2_ok = sh.vec2_ok, wave_ok = sh.wave_ok;

#define MI_WAVE(T_, U_) \
  return launch_wave_row<T_, U_>(rowptr, col, val, B, C, M, ldb, ldc, strideB, strideC, batch, bias, la, s)
  switch (variant) {
    case MI_SPMM_WAVE_ROW_U4:
      if (!wave_ok) return MI_EINVAL;
      if (N == 256) MI_WAVE(1, 4);
      if (N == 512) MI_WAVE(2, 4);
      MI_WAVE(4, 2);
    case MI_SPMM_WAVE_ROW_U8:
      if (!wave_ok) return MI_EINVAL;
      if (N == 256) MI_WAVE(1, 8);
      if (N == 512) MI_WAVE(2, 4);
      MI_WAVE(4, 2);
    case MI_SPMM_WAVE_ROW_U16:
      if (!wave_ok) return MI_EINVAL;
      if (N == 256) MI_WAVE(1, 16);
      if (N == 512) MI_WAVE(2, 8);
      MI_WAVE(4, 4);
    case MI_SPMM_WAVE_ROW_VL:
      if (!(vec4_ok && N == 256)) return MI_EINVAL;
      return launch_wave_row_vl<8>(rowptr, col, val, B, C, M, ldb, ldc, strideB, strideC, batch, bias, la, s);
    case MI_SPMM_PANELS_2: case MI_SPMM_PANELS_3: case MI_SPMM_PANELS_4: case MI_SPMM_PANELS_5:
    case MI_SPMM_PANELS_6: case MI_SPMM_PANELS_8: {
      if (!(wave_ok && batch == 1)) return MI_EINVAL;
      static const int kPanels[] = {2, 3, 4, 5, 6, 8};
      return mi::launch_panels(kPanels[variant - MI_SPMM_PANELS_2], rowptr, col, val, B, C, M, K, N, ldb, ldc, bias, la, s);
    }
    case MI_SPMM_GROUP_PANELS_2: case MI_SPMM_GROUP_PANELS_3: case MI_SPMM_GROUP_PANELS_4: case MI_SPMM_GROUP_PANELS_6:
    case MI_SPMM_GROUP_PANELS_8:
      if (!(vec4_ok && batch == 1 && N <= 1024)) return MI_EINVAL;
      return mi::launch_group_panels(mi::group_panel_count(variant), rowptr, col, val, B, C, M, K, N, ldb, ldc, bias, la, s);
    case MI_SPMM_COLTILE_PANELS: {
      if (!(vec4_ok && batch == 1 && N % 256 == 0 && N >= 256)) return MI_EINVAL;
      int panels = mi::coltile_panels(M, K, N, ldb, nnz);
      if (panels == 0) panels = 3;  // forced by the caller
      return mi::launch_coltile_panels(panels, rowptr, col, val, B, C, M, K, N, ldb, ldc, bias, la, s);
    }
    case MI_SPMM_COLTILE: {
      if (!(vec4_ok && batch == 1)) return MI_EINVAL;
      int w = mi::coltile_width(M, K, N, ldb);
      if (w == 0) w = N >= 256 ? 256 : (N >= 128 ? 128 : 64);  // forced by the caller: any width works
      if (w == 256) return launch_coltile<64>(rowptr, col, val, B, C, M, N, ldb, ldc, bias, la, s);
      if (w == 128) return launch_coltile<32>(rowptr, col, val, B, C, M, N, ldb, ldc, bias, la, s);
      return launch_coltile<16>(rowptr, col, val, B, C, M, N, ldb, ldc, bias, la, s);
    }
    case MI_SPMM_GROUP_VEC4:
      if (!vec4_ok) return MI_EINVAL;
      return dispatch_group<4>(rowptr, col, val, B, C, M, N, ldb, ldc, strideB, strideC, batch, bias, la, s);
    case MI_SPMM_GROUP_VEC2:
      if (!vec2_ok) return MI_EINVAL;
      return dispatch_group<2>(rowptr, col, val, B, C, M, N, ldb, ldc, strideB, strideC, batch, bias, la, s);
    case MI_SPMM_GROUP_VEC4U:
      if (N < 4) return MI_EINVAL;
      return dispatch_group_u4(rowptr, col, val, B, C, M, N, ldb, ldc, strideB, strideC, batch, bias, la, s);
    case MI_SPMM_GROUP_SCALAR:
      return dispatch_group<1>(rowptr, col, val, B, C, M, N, ldb, ldc, strideB, strideC, batch, bias, la, s);
    case MI_SPMM_SLAB:
      if (!(vec4_ok && batch == 1 && K > 0)) return MI_EINVAL;
      return mi::launch_spmm_slab(rowptr, col, val, B, C, M, K, N, ldb, ldc, bias, la.thresh, s);
    case MI_SPMM_LDS_B:
      if (!(vec4_ok && mi::spmm_ldsb_fits(K, N))) return MI_EINVAL;
    {
      const int st = mi::launch_spmm_ldsb(rowptr, col, val, B, C, batch, M, K, N, ldb, ldc, strideB, strideC, bias, la.thresh,
                                          s, nullptr, nnz);
      if (st != 1) return st;  // 1: a shape only the quad form covers, with that form unavailable → same bits from the L2s
      return dispatch_group<4>(rowptr, col, val, B, C, M, N, ldb, ldc, strideB, strideC, batch, bias, la, s);
    }
    case MI_SPMM_NARROW: {
      if (N >= 4) return MI_EINVAL;
      const long blocks = ((long)M + 3) / 4;
      if (blocks > 0x7fffffffL) return MI_ERANGE;
      hipLaunchKernelGGL(spmm_narrow_kernel, dim3((unsigned)blocks, (unsigned)batch), dim3(256), 0, s, rowptr, col,
                         val, B, C, M, N, ldb, ldc, strideB, strideC, bias);
      return mi::check_launch();
    }
    default:
      return MI_EINVAL;
  }
#undef MI_WAVE
}

// Can this plan take its rows through a schedule's slot → row map?  (The kernels of this file and of spmm_panels.hip that
// walk rows one wave or one lane group at a time; the column-tiled launches, the slab / LDS-resident plans and N < 4 keep
// their own row order.)
bool variant_takes_order(int variant) {
  switch (variant) {
    case MI_SPMM_WAVE_ROW_U4: case MI_SPMM_WAVE_ROW_U8: case MI_SPMM_WAVE_ROW_U16:
    case MI_SPMM_GROUP_VEC4: case MI_SPMM_GROUP_VEC2: case MI_SPMM_GROUP_SCALAR: case MI_SPMM_GROUP_VEC4U:
    case MI_SPMM_PANELS_2: case MI_SPMM_PANELS_3: case MI_SPMM_PANELS_4: case MI_SPMM_PANELS_5: case MI_SPMM_PANELS_6:
    case MI_SPMM_PANELS_8:
    case MI_SPMM_GROUP_PANELS_2: case MI_SPMM_GROUP_PANELS_3: case MI_SPMM_GROUP_PANELS_4: case MI_SPMM_GROUP_PANELS_6:
    case MI_SPMM_GROUP_PANELS_8:
      return true;
    default:
      return false;
  }
}

}  // namespace

namespace mi {

int spmm_dispatch(int variant, const int32_t* rowptr, const int32_t* col, const float* val,
                  int64_t nnz, int32_t batch, int32_t M, int32_t K, int32_t N, const float* B,
                  int64_t ldb, int64_t strideB, float* C, int64_t ldc, int64_t strideC,
                  const float* bias, void* workspace, size_t workspace_bytes, hipStream_t s,
                  int long_mode, const RowSchedule* sched) {
  if (M < 0 || K < 0 || N < 0 || nnz < 0 || batch < 0) return MI_EINVAL;
  if (variant < 0 || variant >= MI_SPMM_VARIANT_COUNT) return MI_EINVAL;
  if (long_mode < MI_LONG_ROWS_AUTO || long_mode > MI_LONG_ROWS_AUTO_ZEROED) return MI_EINVAL;
  if (nnz > 0x7fffffffLL) return MI_ERANGE;  // int32 rowptr entries
  if (batch > 65535) return MI_ERANGE;       // grid.y
  if (M == 0 || N == 0 || batch == 0) return MI_OK;
  if (!rowptr || !C) return MI_EINVAL;
  if (nnz > 0 && (!col || !val || !B)) return MI_EINVAL;
  if (ldb < N || ldc < N) return MI_EINVAL;
  if (sched != nullptr && (sched->rows != M || sched->order == nullptr || batch != 1)) return MI_EINVAL;

  Shape sh = classify(N, ldb, ldc, strideB, strideC, B, C);
  if (bias && !mi::aligned16(bias)) sh.vec4_ok = sh.wave_ok = false;
  if (bias && (reinterpret_cast<uintptr_t>(bias) & 7u)) sh.vec2_ok = false;
  if (variant == MI_SPMM_AUTO) variant = choose_variant(sh, nnz, batch, M, K, N, ldb);
  if (sched != nullptr && (!sched->active || !variant_takes_order(variant))) sched = nullptr;  // (this plan keeps its own row order: same bits)

  // long rows get their own kernel when a workspace is there to list them (no row can be long
  // unless the matrix has more than kLongRow non-zeros)
  // (the slab plan needs no split: a long row there is 8 rows' worth of ordinary work for its wave,
  // not a serial tail — and a matrix dense enough to have thousands of long rows would drown the
  // one-workgroup-per-row kernel; its rows all keep the plain CSR order)
  // MI_LONG_ROWS_SPLIT / _NONE pin the rule whatever the plan (a row shard must sum its rows the way
  // the whole matrix would: sharded.py); N < 4 keeps the narrow kernel's own order in every mode.
  const bool pinned_split = long_mode == MI_LONG_ROWS_SPLIT || long_mode == MI_LONG_ROWS_PREPARED;
  const bool split = workspace != nullptr && batch == 1 && nnz > kLongRow && variant != MI_SPMM_NARROW &&
                     long_mode != MI_LONG_ROWS_NONE && (variant != MI_SPMM_SLAB || pinned_split);
  int* ws = static_cast<int*>(workspace);
  const LongWs lw = long_ws_layout(nnz, N);
  LongArg la = {0x7fffffff, 0, 0, 0, nullptr, nullptr, nullptr, 0};
  const bool prepared = long_mode == MI_LONG_ROWS_PREPARED;
  // L2-level panel plans with a workspace at hand: the probe's verdicts decide on the device whether the passes stay
  // passes (spmm_locality_probe_kernel).  The Infinity-Cache level (B ≥ 768 MiB: config C3) is left alone.
  const bool panel_plan = (variant >= MI_SPMM_PANELS_2 && variant <= MI_SPMM_PANELS_8) ||
                          (variant >= MI_SPMM_GROUP_PANELS_2 && variant <= MI_SPMM_GROUP_PANELS_8);
  const double b_bytes_all = (double)K * (double)ldb * 4.0;
  if (panel_plan && batch == 1 && nnz > 0 && b_bytes_all < 768.0 * 1048576.0 && workspace != nullptr && workspace_bytes >= lw.bytes &&
      (reinterpret_cast<uintptr_t>(workspace) & 15u) == 0) {
    int* verdicts = reinterpret_cast<int*>(static_cast<char*>(workspace) + lw.adapt_off);
    const int st = launch_locality_probe(rowptr, col, M, ldb, b_bytes_all, verdicts, sched != nullptr ? sched->order : nullptr, s);
    if (st != MI_OK) return st;
    la.adapt = verdicts;
  }
  if (split) {
    if (workspace_bytes < lw.bytes) return MI_ENOMEM;
    if ((reinterpret_cast<uintptr_t>(workspace) & 15u) != 0) return MI_EINVAL;
    la.thresh = kLongRow;
    la.cap_e = (int)lw.cap_e, la.cap_s = (int)lw.cap_s, la.cap_p = (int)lw.cap_p;
    if (!prepared) {  // else: the list was built once by mi_spmm_long_rows_prepare
      // the four counters start from zero; entries and the slot → entry map are written before they are read.
      // MI_LONG_ROWS_AUTO_ZEROED: the caller's workspace enters with a zero header (and leaves with one)
      if (long_mode != MI_LONG_ROWS_AUTO_ZEROED) MI_HIP_TRY(hipMemsetAsync(ws, 0, 16, s));
      if (variant == MI_SPMM_SLAB || variant == MI_SPMM_LDS_B) {  // those kernels live in files of their own and only skip: list here
        LongArg fl = la;
        fl.ws = ws;
        const int st = launch_find_long_rows(rowptr, M, fl, s);
        if (st != MI_OK) return st;
      } else {
        la.ws = ws;  // the main kernel lists the rows it skips
      }
    }
  }
  int st = MI_OK;
  bool long_rows_done = false;
  if (sched == nullptr) {
    st = launch_variant(variant, sh, rowptr, col, val, nnz, batch, M, K, N, B, ldb, strideB, C, ldc, strideC, bias, la, s);
  } else {
    // A schedule: slots are rows by DESCENDING length class (longest first: no long row is left for the end of the grid;
    // rows that share a wave or a workgroup have about the same length).  Slots [0, heavy) — rows beyond the schedule's
    // heavy length, those the long-row kernel takes among them (skipped and listed as ever) — go in a launch of their own
    // with more gathers in flight per row, on the schedule's side stream BESIDE the launch(es) of the rest.  Per-row
    // arithmetic is that of the unscheduled plan: the same bits.
    const int heavy_slots = sched->heavy < 0 ? 0 : (sched->heavy > M ? M : sched->heavy);
    const int heavy = N >= 4 ? heavy_slots : 0;  // (narrower products: the narrow kernel, its own order, no heavy launch)
    LongArg lh = la, lr = la;
    lh.order = sched->order, lh.nslots = heavy;
    lh.adapt = nullptr;  // one pass, every column
    // No prepared list: a scan of rowptr lists the rows beyond the threshold FIRST (4 bytes per row: 3 – 5 µs), and the listed
    // rows are summed in the heavy rows' launch — beside the ordinary launch, not behind it (the plain entry points on their
    // automatic schedules: 185 long rows beside config C3's shape cost 0.6 ms behind it).  The scan reads EVERY row, not the
    // heavy slots alone: the schedule may be stale (the extension's automatic ones are keyed loosely — a permutation of the
    // rows that no longer knows where the long ones are), and a row that is skipped must be on the list.
    const bool early_list = split && !prepared && heavy_slots > 0 && la.ws != nullptr;
    const bool fork = sched->side != nullptr && sched->fork != nullptr && sched->join != nullptr &&
                      (heavy > 0 || (split && prepared) || early_list);
    // The few big launches — the listed long rows' and the heavy rows' kernel (8 waves + 130 KB of LDS per workgroup) — go FIRST
    // and on the caller's stream: a workgroup of that size only finds room on a CU before the
    // ordinary launch has filled every wave slot with its small ones.  The ordinary launch(es) follow on the schedule's side
    // stream and fill what is left; the caller's stream then waits for them.
    // (the scan goes BEFORE the fork: the ordinary launch must not get to the CUs ahead of the big workgroups)
    if (early_list) {
      st = launch_find_long_rows(rowptr, M, la, s);
      if (st != MI_OK) return st;
    }
    hipStream_t rest = s;
    if (fork) {
      MI_HIP_TRY(hipEventRecord(sched->fork, s));
      MI_HIP_TRY(hipStreamWaitEvent(sched->side, sched->fork, 0));
      rest = sched->side;
    }
    if (early_list) {
      lh.ws = nullptr;  // (listed just now: the heavy slots only skip them)
      // the counters are zeroed by a memset behind the launch, not by its last workgroup: counting 700 workgroups that have
      // nothing to sum through one atomic held the heavy slots' workgroups back until the ordinary launch had taken the CUs
      // (measured: 245 µs for this launch against 131 with a prepared list)
      st = launch_staged_rows(ws, lw, false, lh, rowptr, col, val, B, C, N, ldb, ldc, bias, s);
      if (st != MI_OK) return st;
      MI_HIP_TRY(hipMemsetAsync(ws, 0, 16, s));
      long_rows_done = true;
      lr.ws = nullptr;  // its rows beyond the threshold (a stale schedule may leave some there) are on the list already: skipped only
    // a PREPARED list of the rows beyond the long-row threshold: their kernel needs nothing from this product's other launches
    } else if (split && prepared && heavy > 0) {  // both in one launch — neither waits for the other
      st = launch_staged_rows(ws, lw, false, lh, rowptr, col, val, B, C, N, ldb, ldc, bias, s);
      if (st != MI_OK) return st;
      long_rows_done = true;
    } else if (split && prepared) {
      st = launch_long_rows(ws, lw, rowptr, col, val, B, C, N, ldb, ldc, bias, false, s);
      if (st != MI_OK) return st;
      long_rows_done = true;
    } else if (heavy > 0) {
      st = launch_heavy_rows(rowptr, col, val, M, N, B, ldb, C, ldc, bias, lh, s);
      if (st != MI_OK) return st;
    }
    lr.order = sched->order + heavy, lr.nslots = M - heavy;  // (with a pinned heavy length rows beyond the threshold may sit on either side of `heavy`: both launches list)
    st = launch_variant(variant, sh, rowptr, col, val, nnz, batch, M, K, N, B, ldb, strideB, C, ldc, strideC, bias, lr, rest);
    if (fork) {
      MI_HIP_TRY(hipEventRecord(sched->join, sched->side));
      MI_HIP_TRY(hipStreamWaitEvent(s, sched->join, 0));
    }
  }
  if (st != MI_OK || !split || long_rows_done) return st;
  // One follow-up launch: the listed rows, their combination (by the last workgroup of each row) and, for a list
  // built by this product, the reset of the counters.  With no long row every workgroup reads three zeros and exits.
  return launch_long_rows(ws, lw, rowptr, col, val, B, C, N, ldb, ldc, bias, !prepared, s);
}

}  // namespace mi

namespace {
using mi::spmm_dispatch;
}  // namespace

namespace mi {

// Row-major B [K, ldb], COLUMN-major C (Ccm [N, ldc]): the one-wave-per-row plan with the output transpose
// fused into its epilogue.  Returns MI_OK after launching, 1 when AUTO would not run the one-wave-per-row
// kernel on this problem (the caller then takes its two-transposes form), a negative status on error.
// Every row keeps the plain CSR-order chain (MI_LONG_ROWS_NONE).
int launch_spmm_wave_row_colmajor_out(const int32_t* rowptr, const int32_t* col, const float* val, int64_t nnz,
                                      int32_t M, int32_t K, int32_t N, const float* B, int64_t ldb, float* Ccm,
                                      int64_t ldc, bool launch, hipStream_t s) {
  // the plan of the row-major product this stands in for (C row-major with ldc = N, as the executor's Ct)
  const Shape sh = classify(N, ldb, N, 0, 0, B, B);
  const int variant = choose_variant(sh, nnz, 1, M, K, N, ldb);
  if (variant != MI_SPMM_WAVE_ROW_U8) return 1;
  if (!launch) return MI_OK;
  const long blocks = ((long)M + 15) / 16;
  if (blocks > 0x7fffffffL) return MI_ERANGE;
  const int vec_ok = (ldc % 4 == 0) && aligned16(Ccm);
  const size_t lds = (size_t)16 * (N + 4) * sizeof(float);
#define MI_CT(T_, U_)                                                                                           \
  do {                                                                                                          \
    auto k = spmm_wave_row_ct_kernel<T_, U_>;                                                                   \
    if (lds > 64 * 1024) MI_HIP_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(1024), lds, s, rowptr, col, val, B, Ccm, M, (long)ldb,    \
                       (long)ldc, vec_ok);                                                                      \
  } while (0)
  // T, U as launch_variant runs MI_SPMM_WAVE_ROW_U8
  if (N == 256) MI_CT(1, 8);
  else if (N == 512) MI_CT(2, 4);
  else MI_CT(4, 2);
#undef MI_CT
  return check_launch();
}

}  // namespace mi

extern "C" {

int mi_spmm_csr_f32(const int32_t* rowptr, const int32_t* col, const float* val, int64_t nnz,
                    int32_t M, int32_t K, int32_t N, const float* B, int64_t ldb, float* C,
                    int64_t ldc, mi_stream_t stream) {
  return spmm_dispatch(MI_SPMM_AUTO, rowptr, col, val, nnz, 1, M, K, N, B, ldb, 0, C, ldc, 0, nullptr, nullptr, 0,
                       static_cast<hipStream_t>(stream));
}

int mi_spmm_csr_ws_f32(const int32_t* rowptr, const int32_t* col, const float* val, int64_t nnz,
                       int32_t M, int32_t K, int32_t N, const float* B, int64_t ldb, const float* bias,
                       float* C, int64_t ldc, void* workspace, size_t workspace_bytes,
                       mi_stream_t stream) {
  return spmm_dispatch(MI_SPMM_AUTO, rowptr, col, val, nnz, 1, M, K, N, B, ldb, 0, C, ldc, 0, bias, workspace,
                       workspace_bytes, static_cast<hipStream_t>(stream));
}

int mi_spmm_csr_ex_f32(const int32_t* rowptr, const int32_t* col, const float* val, int64_t nnz,
                       int32_t M, int32_t K, int32_t N, const float* B, int64_t ldb, const float* bias,
                       float* C, int64_t ldc, int long_rows, void* workspace, size_t workspace_bytes,
                       mi_stream_t stream) {
  if (long_rows == MI_LONG_ROWS_SPLIT && workspace == nullptr && nnz > kLongRow) return MI_EINVAL;
  return spmm_dispatch(MI_SPMM_AUTO, rowptr, col, val, nnz, 1, M, K, N, B, ldb, 0, C, ldc, 0, bias, workspace,
                       workspace_bytes, static_cast<hipStream_t>(stream), long_rows);
}

int mi_spmm_csr_bias_f32(const int32_t* rowptr, const int32_t* col, const float* val, int64_t nnz,
                         int32_t M, int32_t K, int32_t N, const float* B, int64_t ldb,
                         const float* bias, float* C, int64_t ldc, mi_stream_t stream) {
  return spmm_dispatch(MI_SPMM_AUTO, rowptr, col, val, nnz, 1, M, K, N, B, ldb, 0, C, ldc, 0, bias, nullptr, 0,
                       static_cast<hipStream_t>(stream));
}

int mi_spmm_csr_f32_variant(int variant, const int32_t* rowptr, const int32_t* col,
                            const float* val, int64_t nnz, int32_t M, int32_t K, int32_t N,
                            const float* B, int64_t ldb, float* C, int64_t ldc,
                            mi_stream_t stream) {
  return spmm_dispatch(variant, rowptr, col, val, nnz, 1, M, K, N, B, ldb, 0, C, ldc, 0, nullptr, nullptr, 0,
                       static_cast<hipStream_t>(stream));
}

int mi_spmm_csr_ex_variant_f32(int variant, const int32_t* rowptr, const int32_t* col, const float* val, int64_t nnz,
                               int32_t M, int32_t K, int32_t N, const float* B, int64_t ldb, const float* bias, float* C,
                               int64_t ldc, int long_rows, void* workspace, size_t workspace_bytes, mi_stream_t stream) {
  if (long_rows == MI_LONG_ROWS_SPLIT && workspace == nullptr && nnz > kLongRow) return MI_EINVAL;
  return spmm_dispatch(variant, rowptr, col, val, nnz, 1, M, K, N, B, ldb, 0, C, ldc, 0, bias, workspace, workspace_bytes,
                       static_cast<hipStream_t>(stream), long_rows);
}

int mi_spmm_csr_batched_f32(const int32_t* rowptr, const int32_t* col, const float* val,
                            int64_t nnz_total, int32_t batch, int32_t M, int32_t K, int32_t N,
                            const float* B, int64_t ldb, int64_t strideB, float* C, int64_t ldc,
                            int64_t strideC, mi_stream_t stream) {
  if (strideB < 0 || strideC < 0) return MI_EINVAL;
  return spmm_dispatch(MI_SPMM_AUTO, rowptr, col, val, nnz_total, batch, M, K, N, B, ldb, strideB,
                       C, ldc, strideC, nullptr, nullptr, 0, static_cast<hipStream_t>(stream));
}

int mi_spmm_csr_batched_perm_f32(const int32_t* rowptr, const int32_t* col, const float* val, const int32_t* perm,
                                 int64_t nnz_total, int32_t batch, int32_t M, int32_t K, int32_t N, const float* B,
                                 int64_t ldb, int64_t strideB, float* C, int64_t ldc, int64_t strideC,
                                 mi_stream_t stream) {
  if (M < 0 || K < 0 || N < 0 || nnz_total < 0 || batch < 0 || strideB < 0 || strideC < 0) return MI_EINVAL;
  if (nnz_total > 0x7fffffffLL || batch > 65535) return MI_ERANGE;
  if (M == 0 || N == 0 || batch == 0) return MI_OK;
  if (!rowptr || !C || !perm) return MI_EINVAL;
  if (nnz_total > 0 && (!col || !val || !B)) return MI_EINVAL;
  if (ldb < N || ldc < N) return MI_EINVAL;
  const Shape sh = mi::classify(N, ldb, ldc, strideB, strideC, B, C);
  // only the LDS-resident-B kernel reads its values through a permutation; every row keeps the plain CSR-order chain
  // (what mi_spmm_csr_batched_f32 gives a batch: no long-row rule without a workspace)
  if (mi::choose_variant(sh, nnz_total, batch, M, K, N, ldb) != MI_SPMM_LDS_B || !sh.vec4_ok || !mi::spmm_ldsb_fits(K, N))
    return 1;
  // a B that goes in as column tiles reads a row's entries once per tile — and would gather every value through the
  // permutation once per tile (2048 tokens × 64 at 5 % kept: 0.214 ms against 0.068 for a gathered copy + the plain
  // product): not taken, the caller gathers once (mi_gather_f32)
  if (mi::spmm_ldsb_tiles(K, N) > 1) return 1;
  return mi::launch_spmm_ldsb(rowptr, col, val, B, C, batch, M, K, N, ldb, ldc, strideB, strideC, nullptr, 0x7fffffff,
                              static_cast<hipStream_t>(stream), perm, nnz_total);
}

int mi_spmm_csr_batched_variant_f32(int variant, const int32_t* rowptr, const int32_t* col, const float* val,
                                    int64_t nnz_total, int32_t batch, int32_t M, int32_t K, int32_t N,
                                    const float* B, int64_t ldb, int64_t strideB, float* C, int64_t ldc,
                                    int64_t strideC, mi_stream_t stream) {
  if (strideB < 0 || strideC < 0) return MI_EINVAL;
  return spmm_dispatch(variant, rowptr, col, val, nnz_total, batch, M, K, N, B, ldb, strideB, C, ldc, strideC, nullptr,
                       nullptr, 0, static_cast<hipStream_t>(stream));
}

}  // extern "C"
