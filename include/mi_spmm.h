/*
 * mi_spmm.h — C-ABI of the MI355X (gfx950) SpMM hot path.
 *
 * This is the drop-in boundary below the `custom_mm` pybind module: plain
 * pointers, sizes and a HIP stream handle; no torch types.  Every entry point
 * enqueues work on `stream` and returns without synchronising.  Return value:
 * 0 (MI_OK) or a negative MI_E* code; `mi_status_string` names it and
 * `mi_last_hip_error` gives the hipError_t behind an MI_EHIP.
 *
 * Each entry cites the reference interface (relative to the reference repo
 * smoorjani/matrix-multiplication) it replaces.  All device pointers must be
 * valid for the sizes stated; index arrays are int32, values are float32.
 */
#ifndef MI_SPMM_H_
#define MI_SPMM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_SPMM_ABI_VERSION 1

typedef void* mi_stream_t; /* a hipStream_t; NULL = the null stream */

enum {
  MI_OK = 0,
  MI_EINVAL = -1,   /* bad argument (null pointer, negative size, bad flag)   */
  MI_ERANGE = -2,   /* a size does not fit the 32-bit index math of the path  */
  MI_EHIP = -3,     /* a HIP runtime call / kernel launch failed              */
  MI_ENOMEM = -4,   /* workspace too small                                    */
  MI_EUNSORTED = -5 /* host inspector: COO rows not sorted (coo2csr contract) */
};

int mi_spmm_abi_version(void);
const char* mi_status_string(int status);
/* hipError_t (as int) recorded by the last MI_EHIP on this thread, else 0. */
int mi_last_hip_error(void);
const char* mi_last_hip_error_string(void);

/* ------------------------------------------------------------------------ *
 * K1 / B1 / B2 — C[M,N] = A_csr[M,K] · B[K,N], row-major B (ldb) and C (ldc).
 * Replaces  spmm_kernel<float,SUM,true> + naive_spmm_wrapper
 *           (src/naive_sparse_mm.cu:24-101, :104-136), reached from
 *           custom_mm.naive_spmm (src/custom_mm.cpp:166-179), and
 *           cusparse_mm_wrapper (src/baseline_mm.cu:167-216), reached from
 *           custom_mm.cusparse_mmul (src/custom_mm.cpp:203-217).
 * Per output element the products are accumulated with fused multiply-add in
 * CSR order (p = rowptr[r] … rowptr[r+1]-1), so the result does not depend on
 * the launch geometry.  Every element of C is written (zeros for empty rows).
 * Exception, N < 4 (SpMV-like): 64 lane-strided chains (lane l takes the row's
 * non-zeros l, l+64, …) combined by a xor-butterfly (32, 16, …, 1) — also a fixed,
 * launch-independent order, restated by the oracle.
 *
 * The `nnz` argument of every SpMM / SDDMM entry (nnz_total for the batched ones):
 *   rowptr's last entry  ≤  nnz  ≤  the number of entries `col` / `val` hold.
 * It picks the plan and sizes the long-row workspace; the kernels walk the rows through
 * rowptr and never read col / val at an index ≥ max(nnz, rowptr's last entry), so a CAPACITY (arrays
 * sized for every element of a dense operand, filled without a read-back) is a valid count
 * and gives the same bits as the exact one.  A count BELOW rowptr's last entry is outside
 * the contract wherever a long-row workspace is in use (its lists are sized from it); without
 * one (mi_spmm_csr_f32, MI_LONG_ROWS_NONE, the batched forms) it only steers the plan — the
 * 16-byte col / val loads of MI_SPMM_LDS_B take their clamp from max(nnz, rowptr's last entry),
 * read on the device.  It must never exceed the array length.
 * ------------------------------------------------------------------------ */
int mi_spmm_csr_f32(const int32_t* rowptr, const int32_t* col, const float* val,
                    int64_t nnz, int32_t M, int32_t K, int32_t N,
                    const float* B, int64_t ldb, float* C, int64_t ldc,
                    mi_stream_t stream);

/* Skew-robust form.  With a workspace of mi_spmm_csr_workspace_bytes(nnz, N) bytes (16-byte
 * aligned; ≈ nnz·N/8192 + nnz/400 bytes), rows with more than 8192 non-zeros are not left to a
 * single wave.  Such a row of `len` non-zeros is summed by S = clamp(len/32768, 1, 128) 16-wave
 * workgroups: its 1024-non-zero chunks are dealt round-robin to 16·S fmaf chains (chain q takes
 * chunks q, q+16S, …, in increasing position), workgroup g adds chains 16g … 16g+15 in that
 * order, and the S workgroup sums are added in order g = 0 … S-1 (then + bias).  The order
 * depends on the row length only.  Rows up to 8192 non-zeros keep the plain CSR-order chain, so
 * results equal mi_spmm_csr_f32 for them.  When the plan (mi_spmm_csr_f32_plan) is MI_SPMM_SLAB
 * or MI_SPMM_NARROW no row is split: every row keeps that plan's own order.  bias (N entries)
 * may be NULL.  This is what custom_mm.naive_spmm / cusparse_mmul call.
 * The workspace also lets the column-panel plans look at the MATRIX, not only at its shape (round 5): its last 64 bytes take
 * the verdicts of a probe launch — per window of 2048 rows, do the rows of B it gathers span ≤ 0.4 of B — that runs ahead of
 * the passes whenever B is below the Infinity-Cache regime (768 MiB); the panel kernels read them on the device, and on a
 * banded / block-diagonal matrix the first pass takes every column while the others return (one pass's chain: the same
 * bits; no read-back, capturable).  Without a workspace the passes always stay passes.  Reference: src/naive_sparse_mm.cu:24-136
 * is one kernel for any matrix; the plans and their adaptation are this library's. */
size_t mi_spmm_csr_workspace_bytes(int64_t nnz, int32_t N);
int mi_spmm_csr_ws_f32(const int32_t* rowptr, const int32_t* col, const float* val,
                       int64_t nnz, int32_t M, int32_t K, int32_t N, const float* B,
                       int64_t ldb, const float* bias, float* C, int64_t ldc,
                       void* workspace, size_t workspace_bytes, mi_stream_t stream);

/* The same product with the long-row rule pinned by the caller instead of following the plan:
 *   MI_LONG_ROWS_AUTO   what mi_spmm_csr_ws_f32 does (split unless the plan is SLAB / NARROW);
 *   MI_LONG_ROWS_NONE   every row keeps the plain CSR-order chain; workspace may be NULL — also the
 *                       cheap call when the caller knows no row exceeds mi_spmm_long_row_threshold()
 *                       non-zeros (one launch, no list building);
 *   MI_LONG_ROWS_SPLIT  rows beyond the threshold are always summed in the split order above, also
 *                       under the SLAB plan (workspace required).
 * A row shard of a larger matrix uses this to sum its rows exactly as the whole matrix would
 * (mi_spmm_auto_splits_long_rows on the WHOLE problem gives the rule), which keeps the
 * row-sharded multi-GPU result bit-identical to the single-GPU one (sharded.py).
 * No counterpart in the reference (single device, one wave per row: src/naive_sparse_mm.cu:24-101). */
/* MI_LONG_ROWS_PREPARED: as SPLIT, and the list of long rows in `workspace` was already built by
 * mi_spmm_long_rows_prepare for this matrix (same nnz and N): the per-call memset + list-building
 * launch are skipped — the inspector–executor form (custom_mm.cusparse_inspect / tiledspmm_inspect_*). */
/* MI_LONG_ROWS_AUTO_ZEROED: as AUTO, and the caller keeps the first 16 bytes of `workspace` ZERO between products:
 * they are zero on entry and zero again once the product's kernels have run (one workspace per stream, reused —
 * what custom_mm.naive_spmm / cusparse_mmul do).  Saves the per-product memset: a product is then exactly two
 * launches — the main kernel, which lists the rows it skips, and one follow-up that sums them (or finds none). */
enum { MI_LONG_ROWS_AUTO = -1, MI_LONG_ROWS_NONE = 0, MI_LONG_ROWS_SPLIT = 1, MI_LONG_ROWS_PREPARED = 2,
       MI_LONG_ROWS_AUTO_ZEROED = 3 };
/* Inspector step for MI_LONG_ROWS_PREPARED: lists the rows beyond the threshold once.  workspace ≥
 * mi_spmm_csr_workspace_bytes(nnz, N); it must stay untouched between the products that use it,
 * and products sharing one workspace must be ordered on one stream (the partial-row area is reused).
 * Replaces what TiledSpMM_inspect amortises in the reference (src/sparse_mm.cu:137-368). */
int mi_spmm_long_rows_prepare(const int32_t* rowptr, int32_t M, int64_t nnz, int32_t N,
                              void* workspace, size_t workspace_bytes, mi_stream_t stream);
int mi_spmm_csr_ex_f32(const int32_t* rowptr, const int32_t* col, const float* val,
                       int64_t nnz, int32_t M, int32_t K, int32_t N, const float* B,
                       int64_t ldb, const float* bias, float* C, int64_t ldc, int long_rows,
                       void* workspace, size_t workspace_bytes, mi_stream_t stream);
/* 1 when mi_spmm_csr_ws_f32 would split long rows for this problem, 0 when not (no GPU work). */
int mi_spmm_auto_splits_long_rows(int64_t nnz, int32_t M, int32_t K, int32_t N, const float* B,
                                  int64_t ldb, const float* C, int64_t ldc);
int mi_spmm_long_row_threshold(void);

/* As above with a kernel-variant override, for benchmarks and tests.
 * variant: MI_SPMM_AUTO or one of the MI_SPMM_* ids below; an id that cannot
 * handle the shape returns MI_EINVAL. */
enum {
  MI_SPMM_AUTO = 0,
  MI_SPMM_WAVE_ROW_U4 = 1,  /* one wave per row, float4 lanes, 4 B-rows in flight  */
  MI_SPMM_WAVE_ROW_U8 = 2,  /* … 8 in flight                                       */
  MI_SPMM_WAVE_ROW_U16 = 3, /* … 16 in flight                                      */
  MI_SPMM_GROUP_VEC4 = 4,   /* G = N/4 lanes per row, 64/G rows per wave            */
  MI_SPMM_GROUP_SCALAR = 5, /* any N / alignment: one float per lane                */
  MI_SPMM_WAVE_ROW_VL = 6,  /* wave per row, col/val via vector load + readlane     */
  MI_SPMM_PANELS_2 = 7,     /* N = 256: K cut into 2 column panels, one launch per   */
  MI_SPMM_PANELS_3 = 8,     /*   panel (Infinity-Cache blocking of B ≥ 768 MiB; L2   */
                            /*   blocking of 6 MiB < B ≤ 128 MiB with P ≈ |B| / 4 MiB); … 3 panels. */
                            /*   Rows whose columns descend somewhere are detected   */
                            /*   and summed in plain CSR order, so the result equals */
                            /*   the one-pass kernels' for every legal CSR input     */
  MI_SPMM_PANELS_4 = 9,
  MI_SPMM_PANELS_5 = 10,
  MI_SPMM_PANELS_6 = 11,
  MI_SPMM_PANELS_8 = 12,
  MI_SPMM_GROUP_VEC2 = 13,  /* G = N/2 lanes per row, 8 B per lane                    */
  MI_SPMM_COLTILE = 14,     /* wide N: XCD-aware column tiles so each XCD's L2 holds its B slice */
  MI_SPMM_COLTILE_PANELS = 15, /* wide N and tall K: column tiles × row panels of B, one launch per panel */
  MI_SPMM_NARROW = 16,      /* N < 4: wave per row, lanes over non-zeros, shuffle reduction (own order) */
  MI_SPMM_SLAB = 17,        /* moderate density, N ≥ 128: 128 rows × 256 columns per workgroup, B staged
                               through LDS in 64-row slabs, one ds_read_b128 per non-zero           */
  MI_SPMM_LDS_B = 18,       /* K·N·4 ≤ 128 KB (N ≤ 256, N % 4 == 0): an item's whole B copied into LDS, rows
                               gather from there — batched products of small matrices (pruned attention) */
  MI_SPMM_GROUP_PANELS_2 = 19, /* N ≤ 128 (float4 lanes), B beyond the Infinity Cache: the lane-group kernel in 2 column   */
  MI_SPMM_GROUP_PANELS_3 = 20, /*   panels (3, 4), one launch per panel, C carried; a pass takes the entries whose running */
  MI_SPMM_GROUP_PANELS_4 = 21, /*   maximum of the row's columns lies in its panel: CSR order kept for every legal input   */
  MI_SPMM_GROUP_PANELS_6 = 22,
  MI_SPMM_GROUP_PANELS_8 = 23,
  MI_SPMM_GROUP_VEC4U = 24, /* any N ≥ 4 at any 4-byte alignment: four floats per lane on dword-aligned 16-byte accesses, a
                               row's partial last quad shifted back onto its neighbour (same chain, same bits) */
  MI_SPMM_VARIANT_COUNT = 25
};
/* The two forms of MI_SPMM_LDS_B (same bits): 16 lanes per row (any tile width), or — tiles of 64 / 128 columns — a
 * quad per row with 16-byte loads of col / val (what BERT's head size runs).  form: -1 by rule (default), 0 the 16-lane
 * form only, 1 the quad form wherever it covers the shape.  Process-wide; a developer / test knob. */
int mi_spmm_ldsb_set_form(int form);
int mi_spmm_csr_f32_variant(int variant, const int32_t* rowptr, const int32_t* col,
                            const float* val, int64_t nnz, int32_t M, int32_t K,
                            int32_t N, const float* B, int64_t ldb, float* C,
                            int64_t ldc, mi_stream_t stream);
/* mi_spmm_csr_ex_f32 with the plan pinned (benchmarks and tests: bias and the long-row rule on a chosen kernel). */
int mi_spmm_csr_ex_variant_f32(int variant, const int32_t* rowptr, const int32_t* col, const float* val,
                               int64_t nnz, int32_t M, int32_t K, int32_t N, const float* B, int64_t ldb,
                               const float* bias, float* C, int64_t ldc, int long_rows, void* workspace,
                               size_t workspace_bytes, mi_stream_t stream);

/* What MI_SPMM_AUTO resolves to for this problem (no GPU work), how many kernel
 * launches a variant issues per product, and the kernel's name as profilers show
 * it — so a benchmark can attribute launch durations. */
int mi_spmm_csr_f32_plan(int64_t nnz, int32_t M, int32_t K, int32_t N, const float* B,
                         int64_t ldb, const float* C, int64_t ldc);
int mi_spmm_variant_launches(int variant);
const char* mi_spmm_variant_name(int variant);

/* ------------------------------------------------------------------------ *
 * Row schedules — the inspector's format for degree-skewed matrices (power-law row lengths: adjacency matrices).
 * Counterpart of what the reference's inspector builds once per matrix (TiledSpMM_inspect: footprint tiles + warp-sliced
 * ELL, src/sparse_mm.cu:137-368) and of the merge-spmm lineage of its kernel (src/naive_sparse_mm.cu:20-21); here the CSR
 * arrays stay as they are and the inspector builds the ORDER in which rows are handed to waves:
 *     order[slot] = row, rows by descending length class (exact below 32 entries, eight classes per octave above).
 * Longest rows first (no serial chain left for the end of the grid), rows sharing a wave / workgroup alike in length, and
 * the rows beyond a heavy length — chosen from nnz, see mi_spmm_schedule_info — in a launch of their own with more gathers in
 * flight per row on the schedule's side stream, beside the launch(es) of the rest.  Every row's fmaf chain is that of the
 * unscheduled product: mi_spmm_csr_scheduled_f32 returns the SAME BITS as mi_spmm_csr_ex_f32 for every plan and long-row
 * mode (plans that cannot take a row order — column tiles, MI_SPMM_SLAB, MI_SPMM_LDS_B, MI_SPMM_NARROW — run unscheduled).
 * LOCALITY the row order hides (the reference compacts each block's footprint of B through `mapindex`, src/sparse_mm.cu:62-68,
 * 259): given the columns, the inspector also tries the rows in the order of their median column inside a length class, MEASURES
 * what a window of 2048 consecutive slots then touches of B against the natural order (its FOOTPRINT: the 256ths of B that hold
 * three quarters of its gathers), and keeps that order only when the natural order is not local already (footprint > 40 % of
 * B) and the new one more than halves it — a banded or
 * community-structured matrix whose rows arrive shuffled then gathers like the unshuffled one.  Same bits, again.
 *   mi_spmm_schedule_create: `order` (device; M ints, 2·M when col is given) is caller-owned and must outlive the schedule; workspace ≥
 *     mi_spmm_schedule_workspace_bytes(M) is only used during the call.  col may be NULL (no locality pass).  Builds on
 *     `stream` and SYNCHRONISES it (reads ≈ 1.5 KiB back: the class table, the window statistics): inspection time, not
 *     capturable.  N: the dense width the schedule will mostly be used with.
 *   Products on one schedule must be ordered on one stream (they share its fork / join events).  The side stream itself is ONE per
 *   device for every schedule of the process (a stream per schedule ran out of hardware queues): unrelated products may see a
 *   false order between their ordinary launches, nothing else.
 *   mi_spmm_schedule_info: info[12] = {rows, heavy slots, heavy length, non-empty classes, lower bound of the longest row,
 *     flags (1: has a side stream; 2: ACTIVE — an order costs the locality of consecutive rows, so a matrix of short, alike
 *     rows keeps its products unscheduled: active with heavy rows, with a locality order, or mean ≥ 16 entries with ≥ 2 % of the
 *     entries in rows of ≥ 1.5 × the mean; 4: locality order), nnz, N, a window's footprint in natural order / in this order (‰ of B), mean row span (‰ of
 *     K) (-1: not measured), 0}.
 *   (Heavy slots: the rows of the length classes above the heavy length's own — by the rule at most 128 rows, the longest classes.)
 *   mi_spmm_schedule_set_heavy: another heavy length (0: every row; ≥ the longest: none) / every launch in line on the caller's
 *     stream instead of the rest beside the heavy launch — tests and A/B measurements; makes the schedule active.
 *   mi_spmm_csr_scheduled_f32: mi_spmm_csr_ex_variant_f32 (variant MI_SPMM_AUTO = by plan) on the schedule.
 * ------------------------------------------------------------------------ */
typedef struct mi_spmm_schedule mi_spmm_schedule_t;
size_t mi_spmm_schedule_workspace_bytes(int32_t M);
/* The build in two halves for callers that must not synchronise: _begin enqueues the build of both candidate orders (`order`:
 * by length class; `order_locality`, M ints, may be NULL: + by median column) and the copies of the class table and the window
 * statistics to host_out (MI_SCHEDULE_HOST_INTS ints; pinned memory keeps the copies asynchronous); once those have landed
 * (an event behind them), _finish reads them on the host, picks the order and creates the object.  mi_spmm_schedule_create =
 * _begin + a stream synchronise + _finish, with the locality order in the SECOND half of `order` (2·M ints when col != NULL). */
#define MI_SCHEDULE_HOST_INTS 512
int mi_spmm_schedule_begin(const int32_t* rowptr, const int32_t* col, int32_t M, int32_t K, int64_t nnz, int32_t N,
                           int32_t* order, int32_t* order_locality, void* workspace, size_t workspace_bytes,
                           int32_t* host_out, mi_stream_t stream);
int mi_spmm_schedule_finish(const int32_t* host_out, int32_t* order, int32_t* order_locality, int32_t M, int32_t K, int64_t nnz,
                            int32_t N, mi_spmm_schedule_t** out);
int mi_spmm_schedule_create(const int32_t* rowptr, const int32_t* col, int32_t M, int32_t K, int64_t nnz, int32_t N,
                            int32_t* order, void* workspace, size_t workspace_bytes, mi_stream_t stream,
                            mi_spmm_schedule_t** out);
int mi_spmm_schedule_destroy(mi_spmm_schedule_t* schedule);
int mi_spmm_schedule_info(const mi_spmm_schedule_t* schedule, int64_t* info);
int mi_spmm_schedule_set_heavy(mi_spmm_schedule_t* schedule, int32_t heavy_len, int use_side_stream);
int mi_spmm_csr_scheduled_f32(const mi_spmm_schedule_t* schedule, int variant, const int32_t* rowptr, const int32_t* col,
                              const float* val, int64_t nnz, int32_t M, int32_t K, int32_t N, const float* B, int64_t ldb,
                              const float* bias, float* C, int64_t ldc, int long_rows, void* workspace,
                              size_t workspace_bytes, mi_stream_t stream);
/* The column-major executor (mi_spmm_csr_colmajor_ex_f32, below: B3 / K2) on a schedule — what an inspector handle runs for a
 * degree-skewed matrix (custom_mm.cusparse_mmul_opt / tiledspmm_mm).  schedule may be NULL or inactive: then exactly
 * mi_spmm_csr_colmajor_ex_f32.  Same bits either way. */
int mi_spmm_csr_colmajor_sched_f32(const mi_spmm_schedule_t* schedule, const int32_t* rowptr, const int32_t* col,
                                   const float* val, int64_t nnz, int32_t M, int32_t K, int32_t N, const float* B,
                                   int64_t ldb, float* C, int64_t ldc, int long_rows, void* long_rows_workspace,
                                   size_t long_rows_workspace_bytes, void* workspace, size_t workspace_bytes,
                                   mi_stream_t stream);

/* ------------------------------------------------------------------------ *
 * Batched form — `batch` independent products in ONE launch:
 *   C[b] (M×N) = A[b] (M×K, CSR) · B[b] (K×N).
 * rowptr is [batch, M+1]; entry rowptr[b*(M+1)+r] indexes into col/val with
 * the batch item's base ALREADY included (a "rowptr of rowptrs": item b's
 * nonzeros are rowptr[b*(M+1)] … rowptr[b*(M+1)+M]-1).  strideB / strideC are
 * element strides between consecutive items (strideB = 0 broadcasts one B).
 * Replaces the Python recursion + torch.stack of naive_matmul
 * (matmuls.py:289-293) and the dead batch_idx feature of spmm_kernel
 * (src/naive_sparse_mm.cu:36,52-53,86).
 * ------------------------------------------------------------------------ */
int mi_spmm_csr_batched_f32(const int32_t* rowptr, const int32_t* col,
                            const float* val, int64_t nnz_total, int32_t batch,
                            int32_t M, int32_t K, int32_t N, const float* B,
                            int64_t ldb, int64_t strideB, float* C, int64_t ldc,
                            int64_t strideC, mi_stream_t stream);
/* … with a kernel-variant override, for benchmarks and tests (a variant that does not take
 * batches, or not this shape, returns MI_EINVAL), and what AUTO resolves to for a batch. */
/* The batched product with the values read THROUGH A PERMUTATION: entry p of the batched CSR (rowptr, col) has the
 * value val[perm[p]] — the transposed pattern of a batched CSR tensor with the permutation that carries the values
 * into it (matmuls caches both per tensor; the reference has no backward for this input, matmuls.py:245-256), so a
 * backward needs no gathered copy of the values.  Same bits as mi_spmm_csr_batched_f32 on the gathered values.
 * Returns MI_OK after launching, 1 (nothing launched) when AUTO's plan for the problem is not the LDS-resident-B
 * kernel, the only one that takes a permutation — the caller then gathers and calls mi_spmm_csr_batched_f32. */
int mi_spmm_csr_batched_perm_f32(const int32_t* rowptr, const int32_t* col, const float* val,
                                 const int32_t* perm, int64_t nnz_total, int32_t batch, int32_t M,
                                 int32_t K, int32_t N, const float* B, int64_t ldb, int64_t strideB,
                                 float* C, int64_t ldc, int64_t strideC, mi_stream_t stream);
int mi_spmm_csr_batched_variant_f32(int variant, const int32_t* rowptr, const int32_t* col,
                                    const float* val, int64_t nnz_total, int32_t batch,
                                    int32_t M, int32_t K, int32_t N, const float* B,
                                    int64_t ldb, int64_t strideB, float* C, int64_t ldc,
                                    int64_t strideC, mi_stream_t stream);
/* Y[i] (K × N) = A[i]ᵀ · X[i] for a batched CSR A (rowptr [batch, M + 1] with base offsets, as above), X [batch, M, N],
 * WITHOUT transposing A: the output tile is kept in registers, every wave walks the item's rows in ascending order and
 * picks the entries of its columns (csrc/spmm_at.hip).  Per output element the terms arrive by ascending row, entries of
 * one row in CSR order — the chain of csr_transpose + mi_spmm_csr_batched_f32, bit for bit.  N ≤ 64.  Returns 1 (nothing
 * launched) for shapes it does not cover.  The gradient of V in pruned attention (reference matmuls.py:245-256 has no
 * backward for a batched CSR operand; the forward's per-call conversion is :289-297). */
int mi_spmm_csr_batched_at_f32(const int32_t* rowptr, const int32_t* col, const float* val, int64_t nnz_total,
                               int32_t batch, int32_t M, int32_t K, int32_t N, const float* X, int64_t ldx,
                               int64_t strideX, float* Y, int64_t ldy, int64_t strideY, mi_stream_t stream);
/* torch's batched CSR indices (int64 crow [batch, M + 1] from 0 per item, int64 col [batch · per_item]) → int32 offsets
 * with the items' bases added and int32 columns, in one launch. */
int mi_batched_csr_narrow_i64(const int64_t* crow, const int64_t* col, int32_t batch, int32_t M, int64_t per_item,
                              int32_t* offsets, int32_t* columns, mi_stream_t stream);
int mi_spmm_csr_batched_f32_plan(int64_t nnz_total, int32_t batch, int32_t M, int32_t K, int32_t N,
                                 const float* B, int64_t ldb, int64_t strideB, const float* C,
                                 int64_t ldc, int64_t strideC);

/* ------------------------------------------------------------------------ *
 * B3 / K2 executor — column-major dense operands:
 *   C (M×N, column-major, ldc ≥ M) = A_csr (M×K) · B (K×N, column-major, ldb ≥ K)
 * i.e. the caller holds activations X = Bᵀ as row-major [N,K] and receives
 * Y = Cᵀ as row-major [N,M].
 * Replaces torch_cusparse_mm_wrapper (src/baseline_mm.cu:272-321) behind
 * custom_mm.cusparse_mmul_opt (src/custom_mm.cpp:259-270) and
 * kernel_TiledELL / TiledSpMM_multiply (src/sparse_mm.cu:39-99, :371-385)
 * behind custom_mm.tiledspmm_mm (src/custom_mm.cpp:337-348).
 * `workspace` must hold mi_spmm_colmajor_workspace_bytes(M,K,N) bytes.
 * ------------------------------------------------------------------------ */
size_t mi_spmm_colmajor_workspace_bytes(int32_t M, int32_t K, int32_t N);
int mi_spmm_csr_colmajor_f32(const int32_t* rowptr, const int32_t* col,
                             const float* val, int64_t nnz, int32_t M, int32_t K,
                             int32_t N, const float* B, int64_t ldb, float* C,
                             int64_t ldc, void* workspace, size_t workspace_bytes,
                             mi_stream_t stream);
/* The same executor with the long-row rule of mi_spmm_csr_ex_f32 (MI_LONG_ROWS_*) and its own
 * long-row workspace (mi_spmm_csr_workspace_bytes(nnz, N); may be NULL for MI_LONG_ROWS_NONE, which
 * is what the plain entry above uses): skewed weight matrices reach the split path through the
 * inspector handles, which prepare the list once (MI_LONG_ROWS_PREPARED). */
/* 1 when the executor runs its NATIVE form for this problem (MI_LONG_ROWS_NONE only): the LDS-slab
 * kernel reading B column-major and writing C column-major directly — transposing slab loads, transposed
 * tile store, no transposed copies; 0 when it transposes B in and C out around the row-major kernel.
 * Host-side decision (plan + a cost comparison), no GPU work; the result bits are the same either way. */
int mi_spmm_colmajor_native_form(int64_t nnz, int32_t M, int32_t K, int32_t N, const float* B,
                                 int64_t ldb, const float* C, int64_t ldc);
/* Which form the executor takes (MI_LONG_ROWS_NONE): 1 = native (above); 2 = B transposed in, then the
 * one-wave-per-row kernel writing C column-major from its epilogue (16 rows per workgroup meet in LDS and
 * leave as 64-byte pieces) — no transposed copy of C; 0 = B transposed in, row-major kernel, C transposed
 * out.  `workspace` is the executor's workspace (its address decides the vector width of the kernel). */
int mi_spmm_colmajor_form(int64_t nnz, int32_t M, int32_t K, int32_t N, const float* B, int64_t ldb,
                          const float* C, int64_t ldc, const void* workspace);
int mi_spmm_csr_colmajor_ex_f32(const int32_t* rowptr, const int32_t* col,
                                const float* val, int64_t nnz, int32_t M, int32_t K,
                                int32_t N, const float* B, int64_t ldb, float* C,
                                int64_t ldc, int long_rows, void* long_rows_workspace,
                                size_t long_rows_workspace_bytes, void* workspace,
                                size_t workspace_bytes, mi_stream_t stream);

/* ------------------------------------------------------------------------ *
 * K5 — dense fp32 product, row-major, optional batch and transposes:
 *   C[b] (m×n) = op(A[b]) · op(B[b]),   op(X) = X or Xᵀ
 * A[b] is stored (transa ? k×m : m×k) with leading dimension lda, B[b] is
 * stored (transb ? n×k : k×n) with ldb, C[b] m×n with ldc; stride* are element
 * strides between batch items (0 broadcasts).  alpha = 1, beta = 0.
 * Replaces cublas_mm_wrapper / cublas_bmm_wrapper
 * (src/baseline_mm.cu:52-102, :105-155) behind custom_mm.cublas_mmul /
 * cublas_bmm (src/custom_mm.cpp:104-164).  Products are accumulated over k in
 * increasing order with fused multiply-add (exact-fp32 MFMA), one rounding per
 * product.
 * ------------------------------------------------------------------------ */
int mi_gemm_f32(int transa, int transb, int32_t m, int32_t n, int32_t k,
                const float* A, int64_t lda, int64_t strideA, const float* B,
                int64_t ldb, int64_t strideB, float* C, int64_t ldc,
                int64_t strideC, int32_t batch, mi_stream_t stream);

/* Two products that share one large operand, in ONE launch that reads it once:
 *     C1[b] = A[b] · B1[b]   ([m, k]·[k, n])        C2[b] = A[b]ᵀ · B2[b]   ([k, m]·[m, n])
 * all operands contiguous row-major, `batch` items.  The backward of `scores = cublasTransbMM.apply(q, k)` (reference
 * README.md:69-77, matmuls.py:131-152): dQ = dS·K (A = dS, B1 = K) and dK = dSᵀ·Q (B2 = Q) — two launches of
 * mi_gemm_f32 read the 403 MB of dS twice.  Every output element is the same k-ordered fused-multiply-add chain as
 * mi_gemm_f32's: same bits.  Returns MI_OK after launching, 1 (nothing launched) for shapes the fused form does not
 * cover — it takes n = 64, k = 64, 128, 256 or 512, m a multiple of 64 (BERT-base / -large attention heads) — the
 * caller then runs the two plain products. */
int mi_gemm_pair_a_at_f32(const float* A, const float* B1, const float* B2, float* C1, float* C2,
                          int32_t batch, int32_t m, int32_t k, int32_t n, mi_stream_t stream);

/* Deterministic split-k (what custom_mm.cublas_mmul / cublas_bmm call): products with few output tiles and a long k — weight
 * gradients over thousands of tokens — leave most of the chip idle as one chain per element.  mi_gemm_split_count(m, n, k, batch)
 * is a function of the SHAPE alone: 1 (the plain chain) unless batch == 1, k ≥ 4096 and fewer than 128 output tiles of 128 × 128;
 * else S = the largest power of two ≤ min(320 / tiles, k / 1024) with k % (32·S) == 0.  With S > 1, k is cut into S equal ranges,
 * each the usual k-ordered fmaf chain from zero, and an element's S partial sums are added in index order ((p0 + p1) + p2) + …,
 * then the bias — a fixed order the oracle restates (oracle_gemm_f32).  The reference's criterion is torch.allclose at 1e-5 with
 * cuBLAS's unspecified order (tests/cublas_kernel_test.py:27-28, src/baseline_mm.cu:96-101).  workspace ≥
 * mi_gemm_workspace_bytes (S·m·n floats; 0 when S == 1), 16-byte aligned; too small a workspace is MI_ENOMEM, never a silent
 * plain chain.  mi_gemm_f32 / mi_gemm_bias_f32 (no workspace) always run the plain chain. */
int mi_gemm_split_count(int32_t m, int32_t n, int32_t k, int32_t batch);
size_t mi_gemm_workspace_bytes(int32_t m, int32_t n, int32_t k, int32_t batch);
int mi_gemm_ws_f32(int transa, int transb, int32_t m, int32_t n, int32_t k, const float* A, int64_t lda, int64_t strideA,
                   const float* B, int64_t ldb, int64_t strideB, const float* bias, float* C, int64_t ldc, int64_t strideC,
                   int32_t batch, void* workspace, size_t workspace_bytes, mi_stream_t stream);

/* Which kernel family mi_gemm_f32 / mi_gemm_bias_f32 take (process-wide; every
 * plan produces the same bits — tests pin one to compare it with another):
 * AUTO picks; TILES = one output tile (or a short chain) per 4-wave workgroup;
 * DUO = the persistent 8-wave kernel whose halves run in anti-phase (whole
 * 128-row tiles only: MI_EINVAL when pinned on a shape it cannot take). */
#define MI_GEMM_PLAN_AUTO 0
#define MI_GEMM_PLAN_TILES 1
#define MI_GEMM_PLAN_DUO 2
int mi_gemm_set_plan(int plan);

/* ------------------------------------------------------------------------ *
 * Fused "sparsify on the fly" form: A is given DENSE (batch × M×K, leading
 * dimension lda, item stride strideA) and its exact zeros are skipped inside the
 * kernel, in ascending column order — bit-identical to mi_dense_to_csr_* followed
 * by mi_spmm_csr_batched_f32, without materialising a CSR.  B[b] is K×N
 * (strideB = 0 shares one B); bias (N entries) may be NULL.
 * Replaces `a.to_sparse_csr()` + get_sparse_tensor_properties + spmm_kernel per
 * call / per slice (reference matmuls.py:289-297, :178-187).
 * Supported when mi_spmm_dense_skip_supported(...) != 0 (N ≤ 256, N % 4 == 0,
 * 16-byte aligned B and C rows); otherwise MI_EINVAL — use the CSR entry points.
 * ------------------------------------------------------------------------ */
int mi_spmm_dense_skip_supported(int32_t N, int64_t lda, int64_t ldb, int64_t ldc,
                                 const float* A, const float* B, const float* C);
int mi_spmm_dense_skip_f32(const float* A, int64_t lda, int64_t strideA, int32_t batch,
                           int32_t M, int32_t K, int32_t N, const float* B, int64_t ldb,
                           int64_t strideB, const float* bias, float* C, int64_t ldc,
                           int64_t strideC, mi_stream_t stream);

/* Gated form of the product above, for callers that must keep the zero-skipping
 * semantics of `a.to_sparse_csr()` (reference matmuls.py:295-296: a zero of A
 * never meets B) while normally taking the dense MFMA product (mi_gemm_f32), in
 * which 0·inf = nan: the launch does nothing unless *gate != 0 (device memory,
 * read when the kernel starts), so
 *     mi_gemm_f32(...);  mi_nonfinite_flag_f32(B, ..., gate);  this(..., gate)
 * leaves the dense product in C when B is finite (where both agree bit for bit)
 * and overwrites it with the zero-skipping product when B holds an inf / nan —
 * decided on the device, nothing read back, graph-capturable.  Any N % 4 == 0
 * (column tiles of 256 inside one launch); gate == NULL always runs.
 * mi_nonfinite_flag_f32 writes *flag = 1 if x (rows×cols, leading dimension ld)
 * holds an inf or nan, else 0. */
int mi_spmm_dense_skip_gated_f32(const float* A, int64_t lda, int64_t strideA, int32_t batch,
                                 int32_t M, int32_t K, int32_t N, const float* B, int64_t ldb,
                                 int64_t strideB, const float* bias, float* C, int64_t ldc,
                                 int64_t strideC, const int32_t* gate, mi_stream_t stream);
int mi_nonfinite_flag_f32(const float* x, int64_t rows, int64_t cols, int64_t ld, int32_t* flag,
                          mi_stream_t stream);

/* ------------------------------------------------------------------------ *
 * Fused FC-layer epilogues:  C = (product) + bias, bias[N] (resp. bias[n]) added
 * to every row AFTER the accumulation chain (one extra rounding, exactly what
 * `output = t.clone(); output += self.bias` computes).  bias may be NULL.
 * Replace the separate clone + add of cublasLinear / cusparseLinear.forward
 * (reference benchmarks/cublas_fc_layer.py:41-45, cusparse_fc_layer.py:41-45).
 * ------------------------------------------------------------------------ */
int mi_spmm_csr_bias_f32(const int32_t* rowptr, const int32_t* col, const float* val,
                         int64_t nnz, int32_t M, int32_t K, int32_t N, const float* B,
                         int64_t ldb, const float* bias, float* C, int64_t ldc,
                         mi_stream_t stream);
int mi_gemm_bias_f32(int transa, int transb, int32_t m, int32_t n, int32_t k,
                     const float* A, int64_t lda, int64_t strideA, const float* B,
                     int64_t ldb, int64_t strideB, const float* bias, float* C,
                     int64_t ldc, int64_t strideC, int32_t batch, mi_stream_t stream);

/* ------------------------------------------------------------------------ *
 * Dense → CSR on the device (exact-zero test, columns ascending within a row),
 * batched: `batch` matrices of rows×cols (leading dimension ld, item stride
 * `stride`).  Two steps so the caller can size col/val without a host sync
 * per item:
 *   1. mi_dense_to_csr_count: writes rowptr[batch*(rows+1)] as a "rowptr of
 *      rowptrs" (see mi_spmm_csr_batched_f32); rowptr[batch*(rows+1)-1] is the
 *      total nnz.  `workspace` ≥ mi_dense_to_csr_workspace_bytes(batch, rows).
 *   2. mi_dense_to_csr_fill: writes col / val (capacity ≥ total nnz).
 * Replaces dense_to_csr (src/baseline_mm.cu:218-264, cusparseDenseToSparse)
 * and the per-slice torch `to_sparse_csr()` of naive_matmul (matmuls.py:295-296).
 * ------------------------------------------------------------------------ */
size_t mi_dense_to_csr_workspace_bytes(int32_t batch, int32_t rows);
int mi_dense_to_csr_count(const float* dense, int32_t batch, int32_t rows,
                          int32_t cols, int64_t ld, int64_t stride,
                          int32_t* rowptr, void* workspace, size_t workspace_bytes,
                          mi_stream_t stream);
int mi_dense_to_csr_fill(const float* dense, int32_t batch, int32_t rows,
                         int32_t cols, int64_t ld, int64_t stride,
                         const int32_t* rowptr, int32_t* col, float* val,
                         mi_stream_t stream);

/* ------------------------------------------------------------------------ *
 * Device CSR transpose (A M×K → Aᵀ K×M, columns ascending within each row of
 * Aᵀ, stable for duplicates): used by the backward pass  grad_B = Aᵀ · dC.
 * Hand-written least-significant-digit counting passes over the column index (csr_transpose.hip):
 * two passes for K·batch ≤ 2²⁰, three beyond; no atomics, deterministic.
 * `workspace` ≥ mi_csr_transpose_workspace_bytes(M, K, nnz) (≈ 8–12 bytes per non-zero plus the
 * per-tile digit tables), 16-byte aligned.  No counterpart in the reference (its backward
 * re-sparsifies a strided view, matmuls.py:319-325, SURVEY.md §8a defect 1).
 * Batched form: `batch` matrices in the batched-CSR layout of mi_spmm_csr_batched_f32 (rowptr
 * [batch, M+1] with global offsets) → t_rowptr [batch, K+1] in the same layout, item b's
 * transposed entries at t_rowptr[b*(K+1)] … ; one set of launches for the whole batch.
 * ------------------------------------------------------------------------ */
size_t mi_csr_transpose_workspace_bytes(int32_t M, int32_t K, int64_t nnz);
int mi_csr_transpose_f32(const int32_t* rowptr, const int32_t* col, const float* val,
                         int64_t nnz, int32_t M, int32_t K, int32_t* t_rowptr,
                         int32_t* t_col, float* t_val, void* workspace,
                         size_t workspace_bytes, mi_stream_t stream);
/* Which plan transposes (process-wide; every plan gives the same bits — tests pin one to compare it with the other):
 *   TABLES    per-(tile, digit) count tables, one count launch + three scan launches ahead of every scatter pass;
 *   ONE_SWEEP the columns are read once for counting: the first pass's count launch also counts the last pass's
 *             digits per group of bins, and the last scatter launch finds its tiles' offsets by decoupled look-back
 *             inside those groups (status words published / polled with agent-scope accesses, tiles handed out by
 *             tickets so every wait is on a running workgroup) — no second count pass over the intermediate array,
 *             no scan of its table; one matrix, 2¹⁰ < K ≤ 2²⁰ (two passes), ≥ 4 M non-zeros, nnz < 2³⁰ —
 *             mi_csr_transpose_one_sweep_applies says whether a problem qualifies;
 *   AUTO      ONE_SWEEP where it applies and pays (≥ 33 M non-zeros), else TABLES.  Pinning ONE_SWEEP on a problem it
 *             does not cover: MI_EINVAL.
 * mi_csr_transpose_check (synchronises `stream`; a debugging aid) reads back the one-sweep plan's give-up flag from
 * the workspace of the last transpose: MI_EHIP if a look-back poll ran into its limit (never on a correct run). */
#define MI_TRANSPOSE_PLAN_AUTO 0
#define MI_TRANSPOSE_PLAN_TABLES 1
#define MI_TRANSPOSE_PLAN_ONE_SWEEP 2
int mi_csr_transpose_set_plan(int plan);
int mi_csr_transpose_one_sweep_applies(int32_t batch, int32_t M, int32_t K, int64_t nnz);
/* 1 when the next mi_csr_transpose*_f32 of this problem runs the one-sweep plan under the plan in force (the only plan
 * whose give-up flag mi_csr_transpose_check can report). */
int mi_csr_transpose_auto_takes_one_sweep(int32_t batch, int32_t M, int32_t K, int64_t nnz);
int mi_csr_transpose_check(const void* workspace, size_t workspace_bytes, int32_t batch, int32_t M,
                           int32_t K, int64_t nnz, mi_stream_t stream);
size_t mi_csr_transpose_batched_workspace_bytes(int32_t batch, int32_t M, int32_t K, int64_t nnz);
/* 1 when the batch is transposed by the one-workgroup-per-item plan (items small enough for a [waves][K] table in LDS,
 * ≤ 256 K entries per item, ≥ 64 items or ≤ 128 K entries in all; plan AUTO): one launch, no workspace used, ≈ 20× faster
 * than the general plan on pruned-attention batches — cheap enough to transpose the VALUES on every backward instead of
 * keeping a permutation (matmuls._batched_csr_backward).  Same stable order, same bits. */
int mi_csr_transpose_batched_in_lds(int64_t nnz, int32_t batch, int32_t M, int32_t K);
int mi_csr_transpose_batched_f32(const int32_t* rowptr, const int32_t* col, const float* val,
                                 int64_t nnz, int32_t batch, int32_t M, int32_t K,
                                 int32_t* t_rowptr, int32_t* t_col, float* t_val,
                                 void* workspace, size_t workspace_bytes, mi_stream_t stream);

/* ------------------------------------------------------------------------ *
 * SDDMM on A's pattern:  out[p] = Σ_j dC[row(p), j] · B[col[p], j]
 * = the gradient of C = A·B with respect to A's stored values.  Any N; fixed summation order
 * (lane l of 64 chains columns 256t + 4l + c, then a xor tree), restated by the oracle.
 * ------------------------------------------------------------------------ */
int mi_sddmm_csr_f32(const int32_t* rowptr, const int32_t* col, int64_t nnz,
                     int32_t M, int32_t K, int32_t N, const float* dC, int64_t lddc,
                     const float* B, int64_t ldb, float* out_val,
                     mi_stream_t stream);

/* SDDMM on a BATCHED CSR pattern (rowptr [batch, M+1] with global offsets, as mi_spmm_csr_batched_f32):
 *   out[p] = Σ_j dC[item][row(p), j] · B[item][col[p], j]      (strideB = 0: one B shared by every item)
 * = the gradient of the stored values of a batched CSR tensor (the reference has no backward for this input,
 * matmuls.py:245-256).  Same bits as mi_sddmm_csr_f32 on the block-diagonal matrix of the batch.  Returns MI_OK after
 * launching the LDS-resident form (an item's B staged in LDS once per workgroup, N ≤ 64, K·N·4 ≤ 128 KB, ≥ 16384 rows
 * of ≥ 4 non-zeros on average), 1 — nothing launched — for every other problem: the caller then runs mi_sddmm_csr_f32
 * on the block-diagonal form. */
int mi_sddmm_csr_batched_f32(const int32_t* rowptr, const int32_t* col, int64_t nnz_total, int32_t batch,
                             int32_t M, int32_t K, int32_t N, const float* dC, int64_t lddc,
                             int64_t strideDC, const float* B, int64_t ldb, int64_t strideB,
                             float* out_val, mi_stream_t stream);

/* dst[p] = src[perm[p]], p < n: the stored values of a CSR tensor carried into its cached transposed pattern by the
 * permutation mi_csr_transpose_* produced for the values 0, 1, 2, … (matmuls' backward; replaces the reference-side
 * `index_select`).  perm entries must lie in [0, length of src). */
int mi_gather_f32(const float* src, const int32_t* perm, int64_t n, float* dst, mi_stream_t stream);

/* Column sums dst[j] = Σ_r src[r, j] (src rows×n, leading dimension ld): the bias gradient of
 * the FC layers (autograd of `output += self.bias`, reference benchmarks/cublas_fc_layer.py:44-45).
 * Fixed summation order (per row chunk: 4 waves × 4 interleaved row chains, added in a fixed
 * order; chunks — 64 rows up to 64 Ki rows, larger beyond — added in order), no atomics.
 * workspace ≥ mi_colsum_workspace_bytes(rows, n). */
size_t mi_colsum_workspace_bytes(int32_t rows, int32_t n);
int mi_colsum_f32(const float* src, int32_t rows, int32_t n, int64_t ld, float* dst,
                  void* workspace, size_t workspace_bytes, mi_stream_t stream);

/* Dense 2-D transpose  dst[cols, rows] = src[rows, cols]ᵀ (row-major, ld's). */
int mi_transpose_f32(const float* src, int32_t rows, int32_t cols, int64_t ld_src,
                     float* dst, int64_t ld_dst, mi_stream_t stream);

/* ------------------------------------------------------------------------ *
 * HOST inspector step (no GPU): COO (sorted by row) → CSR, keeping the input
 * order inside a row.  Replaces TiledSpMM_coo2csr (src/sparse_mm.cu:110-134).
 * rowptr has M+1 entries; col_out / val_out have nnz entries.
 * ------------------------------------------------------------------------ */
int mi_coo_to_csr_host(int32_t M, int64_t nnz, const int32_t* coo_row,
                       const int32_t* coo_col, const float* coo_val,
                       int32_t* rowptr, int32_t* col_out, float* val_out);

/* ------------------------------------------------------------------------ *
 * Peer mappings (multi-GPU exchange) — NEW relative to the reference, which is
 * single-device (`cudaSetDevice(0)`, src/sparse_mm.cu:295; no collective, no peer access).
 * One process per GPU; rank A exports the allocation its C buffer lives in, the 64 handle bytes + the
 * buffer's offset travel to the peers by whatever channel the host side has (torch.distributed's
 * all_gather_object), every peer opens the handle and copies its row blocks straight into A's C
 * (device-to-device over xGMI).  Lifetimes are EXPLICIT: a mapping lives from mi_ipc_open to the matching
 * mi_ipc_close; opens of one handle in one process are counted (one hipIpcOpenMemHandle, closed with the
 * last mi_ipc_close).  The exporter keeps the allocation alive until every peer has closed (the host side
 * puts a barrier between the peers' closes and the free).  No stream argument: these are host calls.
 *   mi_ipc_export : handle_out[MI_IPC_HANDLE_BYTES], *offset_out = dev_ptr − base of its allocation,
 *                   *alloc_bytes_out (may be NULL) = size of that allocation
 *   mi_ipc_open   : *base_out = the peer allocation's base in this process (add the offset)
 *   mi_ipc_close  : drops one open of that handle; MI_EINVAL if it is not open here
 *   mi_ipc_open_count : opens not yet closed in this process (tests: nothing is left mapped)
 * ------------------------------------------------------------------------ */
#define MI_IPC_HANDLE_BYTES 64
int mi_ipc_export(const void* dev_ptr, void* handle_out, int64_t* offset_out, int64_t* alloc_bytes_out);
int mi_ipc_open(const void* handle, void** base_out);
int mi_ipc_close(const void* handle);
int mi_ipc_open_count(void);

/* Replaces dummy_kernel_launch (src/baseline_mm.cu:24-35): launches a 64×64
 * grid on the stream; each thread writes its global id into out[4096]
 * (the reference printf()s it). */
int mi_dummy_kernel(int32_t* out, mi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MI_SPMM_H_ */
