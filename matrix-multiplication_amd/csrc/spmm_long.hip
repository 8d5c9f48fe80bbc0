// Rows beyond the long-row threshold ("skewed matrices"): listed by the main kernels on the way (spmm_device.h:
// long_list_append) or by find_long_rows_kernel, summed in a fixed order that the oracle restates (oracle_spmm_csr_long_f32) —
// here, by several 16-wave workgroups per row, for the shapes that cannot move float4s; float4 shapes go to the staged kernel of
// spmm_heavy.hip (the same order, other workgroup shapes).  Contract: include/mi_spmm.h (mi_spmm_csr_ws_f32, MI_LONG_ROWS_*).
// New relative to the reference, whose kernel walks every row with one warp (src/naive_sparse_mm.cu:60-92).
#include "spmm_device.h"
#include "spmm_internal.h"

namespace {

using mi::LongArg;

// ---------------------------------------------------------------------------
// Skewed matrices.  A row is owned by one wave, which sustains only a few GB/s of gathers, so
// a row with 10⁵–10⁶ non-zeros would be a serial tail of tens of milliseconds.  When the caller
// supplies a workspace (custom_mm always does), rows with more than kLongRow non-zeros are
// skipped AND listed by the kernels above (`la`: long_list_append; find_long_rows_kernel builds the same
// list for plans whose kernel lives elsewhere, and for prepared lists), and summed
// here by S = clamp(len / 32768, 1, 128) 16-wave workgroups: the row's 1024-non-zero chunks are
// dealt round-robin to 16·S chains (chain q runs the fmaf chain over chunks q, q+16S, q+32S, …
// in increasing position), workgroup g owns chains 16g … 16g+15 and adds them in that order,
// and the S workgroup sums are added in order g = 0 … S-1 — by the same workgroup when S = 1,
// else through the workspace by whichever of the S workgroups delivers its partial row LAST
// (an arrival counter per row; agent-scope release by every deliverer, acquire by the last: the sum
// itself is always taken in the order g = 0 … S-1, so it does not depend on who arrives when).
// That is a different — fixed, launch-independent, a function of the row length only — summation
// order for those rows; oracle_spmm_csr_long_f32 restates it, so results stay bit-identical to the oracle.
// ---------------------------------------------------------------------------

__global__ void find_long_rows_kernel(const int* __restrict__ rowptr, int M, LongArg la) {
  const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= M) return;
  const int len = rowptr[r + 1] - rowptr[r];
  if (len > kLongRow) long_list_append(la, (int)r, len);
}

// reset != 0: the list was built for this product only — the workgroup that finishes last zeroes the four
// counters, so a workspace that entered with a zero header leaves with one (MI_LONG_ROWS_AUTO_ZEROED: no memset
// per product).  reset == 0: a prepared list, used again by the next product (only the arrival counters are reset).
template <int VEC>
__global__ __launch_bounds__(kLongWaves * 64) void spmm_long_rows_kernel(
    int* __restrict__ ws, int cap_e, int cap_s, float* __restrict__ partial, const int* __restrict__ rowptr,
    const int* __restrict__ col, const float* __restrict__ val, const float* __restrict__ B, float* __restrict__ C,
    int N, long ldb, long ldc, const float* __restrict__ bias, int reset) {
  typedef Vec<VEC> V;
  typedef typename V::type vec_t;
  __shared__ vec_t part[kLongWaves][64];
  __shared__ int last_flag;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int listed = ws[0], handed = ws[1], handed_p = ws[2];
  if (listed == 0 && handed == 0 && handed_p == 0) return;  // no long row: nothing to sum, nothing to reset
  const int count = listed < cap_e ? listed : cap_e;
  const int slots = handed < cap_s ? handed : cap_s;
  const int* owner = ws + 4 + kLongEnt * (long)cap_e;
  for (int t = blockIdx.x; t < slots; t += gridDim.x) {
    const int e = owner[t];
    if ((unsigned)e >= (unsigned)count) continue;  // slot of a dropped entry
    int* ent = ws + 4 + kLongEnt * (long)e;
    const int row = ent[0], S = ent[2], pb = ent[3];
    const int g = t - ent[1];
    if ((unsigned)g >= (unsigned)S) continue;  // not a slot of that entry
    const long start = rowptr[row], end = rowptr[row + 1];
    const long stride = (long)kLongWaves * S * kLongChunk;
    for (int n0 = 0; n0 < N; n0 += 64 * VEC) {  // 64·VEC output columns per pass
      const int c0 = n0 + lane * VEC;
      const bool on = c0 < N;
      vec_t acc = V::zero();
      for (long cb = start + ((long)g * kLongWaves + wave) * kLongChunk; cb < end; cb += stride) {
        const long ce = cb + kLongChunk < end ? cb + kLongChunk : end;
        for (long p = cb; p < ce; p += 64) {
          const long idx = p + lane;
          const int myc = idx < ce ? col[idx] : 0;
          const float myv = idx < ce ? val[idx] : 0.f;
          const int cnt = ce - p < 64 ? (int)(ce - p) : 64;
          int i = 0;
          for (; i + 8 <= cnt; i += 8) {
            vec_t x[8];
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int c = __builtin_amdgcn_readlane(myc, i + u);
              v[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myv), i + u));
              if (on) x[u] = V::load(B + (long)c * ldb + c0);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
              if (on) acc = V::fma(v[u], x[u], acc);
          }
          for (; i < cnt; ++i) {
            const int c = __builtin_amdgcn_readlane(myc, i);
            const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myv), i));
            if (on) acc = V::fma(v, V::load(B + (long)c * ldb + c0), acc);
          }
        }
      }
      part[wave][lane] = acc;
      __syncthreads();
      if (wave == 0 && on) {
        vec_t tot = part[0][lane];
#pragma unroll
        for (int w = 1; w < kLongWaves; ++w) tot += part[w][lane];
        if (S == 1) {
          if (bias) tot += V::load(bias + c0);
          V::store(C + (long)row * ldc + c0, tot);
        } else {
          V::store(partial + (long)(pb + g) * N + c0, tot);  // N % VEC == 0 and 16-B base when VEC = 4
        }
      }
      __syncthreads();
    }
    if (S > 1) {
      // Deliver: wave 0 is the only wave that stored partial sums.  Its stores are drained, written back at
      // agent scope, and only then does one lane take an arrival ticket (cdna_hip_programming.md Guideline 16:
      // fence before the ticket, with the explicit wait hipcc may drop).  The workgroup that draws the last
      // ticket acquires and adds the S partial rows in order g = 0 … S-1, then the bias.
      if (wave == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
          const int ticket = __hip_atomic_fetch_add(&ent[4], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          last_flag = ticket == S - 1;
          if (ticket == S - 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
        }
      }
      __syncthreads();
      if (last_flag) {
        for (int c = threadIdx.x; c < N; c += blockDim.x) {
          // sc1 loads: served by L2 / memory, never by a line this CU cached before the other workgroups wrote
          float tot = __hip_atomic_load(partial + (long)pb * N + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          for (int gg = 1; gg < S; ++gg)
            tot += __hip_atomic_load(partial + (long)(pb + gg) * N + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (bias) tot += bias[c];
          C[(long)row * ldc + c] = tot;
        }
        if (threadIdx.x == 0) ent[4] = 0;  // a prepared list serves the next product too
      }
      __syncthreads();  // last_flag is rewritten by the next slot
    }
  }
  if (reset) {
    // every read of the counters by this workgroup is done (they were read into registers at the top)
    __syncthreads();
    if (threadIdx.x == 0) {
      const int done = __hip_atomic_fetch_add(&ws[3], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (done == (int)gridDim.x - 1) {
        __hip_atomic_store(&ws[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ws[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ws[2], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ws[3], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

}  // namespace

namespace mi {

LongWs long_ws_layout(int64_t nnz, int32_t N) {
  LongWs w;
  w.cap_e = nnz / kLongRow + 1;
  w.cap_p = nnz >> kLongSplitShift;
  w.cap_s = w.cap_e + w.cap_p;
  w.owner_off = 4 + (size_t)kLongEnt * (size_t)w.cap_e;
  const size_t ints = w.owner_off + (size_t)w.cap_s;
  w.partial_off = (ints * sizeof(int) + 15) / 16 * 16;
  w.adapt_off = (w.partial_off + (size_t)w.cap_p * (size_t)N * sizeof(float) + 15) / 16 * 16;
  w.bytes = w.adapt_off + kAdaptSlots * sizeof(int);
  return w;
}

int launch_find_long_rows(const int32_t* rowptr, int32_t M, const LongArg& la, hipStream_t s) {
  if (M <= 0) return MI_OK;
  hipLaunchKernelGGL(find_long_rows_kernel, dim3((unsigned)(((long)M + 255) / 256)), dim3(256), 0, s, rowptr, M, la);
  return check_launch();
}

int launch_long_rows(bool vec4, int* ws, const LongWs& lw, const int32_t* rowptr, const int32_t* col, const float* val,
                     const float* B, float* C, int32_t N, int64_t ldb, int64_t ldc, const float* bias, bool reset,
                     hipStream_t s) {
  // float4 shapes: a workgroup per group of chains and 64 columns, the chains staged through LDS (spmm_heavy.hip) — the same sums
  if (vec4) return launch_long_rows_staged(ws, lw, rowptr, col, val, B, C, N, ldb, ldc, bias, reset, s);
  float* partial = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + lw.partial_off);
  const unsigned grid = lw.cap_s < 256 ? (unsigned)lw.cap_s : 256u;  // one 16-wave workgroup per CU (grid-stride over the slots)
  hipLaunchKernelGGL(spmm_long_rows_kernel<1>, dim3(grid), dim3(kLongWaves * 64), 0, s, ws, (int)lw.cap_e,
                     (int)lw.cap_s, partial, rowptr, col, val, B, C, N, (long)ldb, (long)ldc, bias, reset ? 1 : 0);
  return check_launch();
}

}  // namespace mi

extern "C" {

size_t mi_spmm_csr_workspace_bytes(int64_t nnz, int32_t N) {
  return nnz > 0 && N > 0 ? mi::long_ws_layout(nnz, N).bytes : 0;
}

int mi_spmm_long_rows_prepare(const int32_t* rowptr, int32_t M, int64_t nnz, int32_t N, void* workspace,
                              size_t workspace_bytes, mi_stream_t stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (M < 0 || nnz < 0 || N < 0) return MI_EINVAL;
  if (nnz > 0x7fffffffLL) return MI_ERANGE;
  if (nnz <= kLongRow || M == 0 || N == 0) return MI_OK;  // no row can be long: the list is never read
  if (!rowptr || !workspace || (reinterpret_cast<uintptr_t>(workspace) & 15u) != 0) return MI_EINVAL;
  const mi::LongWs lw = mi::long_ws_layout(nnz, N);
  if (workspace_bytes < lw.bytes) return MI_ENOMEM;
  int* ws = static_cast<int*>(workspace);
  MI_HIP_TRY(hipMemsetAsync(ws, 0, 16, s));
  const mi::LongArg la = {kLongRow, (int)lw.cap_e, (int)lw.cap_s, (int)lw.cap_p, ws, nullptr, nullptr, 0};
  hipLaunchKernelGGL(find_long_rows_kernel, dim3((unsigned)(((long)M + 255) / 256)), dim3(256), 0, s, rowptr, M, la);
  return mi::check_launch();
}

int mi_spmm_long_row_threshold(void) { return kLongRow; }

}  // extern "C"
