// Host-side interfaces between the SpMM translation units (not installed): the plan rules (spmm_plan.hip), the launchers
// of the column-panel passes (spmm_panels.hip) and of the long-row kernels (spmm_long.hip), used by the dispatcher in
// spmm_csr.hip.
#ifndef MI_SPMM_INTERNAL_H_
#define MI_SPMM_INTERNAL_H_

#include "mi_common.h"

namespace mi {

// ---- long rows (spmm_long.hip) -------------------------------------------------------------------------------------
constexpr int kLongRowThreshold = 8192;  // rows with more non-zeros are summed in the split order (with a workspace)
constexpr int kAdaptVerdicts = 16;       // ints the locality probe writes behind the long-row workspace

// What the kernels get: which rows to leave to the long-row kernel, where to list them, the locality verdicts, and —
// scheduled launches — the order in which row slots map to rows.
struct LongArg {
  int thresh;               // rows with more non-zeros are left to the listed-rows launch (spmm_heavy.hip)
  int cap_e, cap_s, cap_p;  // list capacities (hold for any rowptr consistent with nnz)
  int* ws;                  // the list; nullptr: skip only (the list was prepared beforehand) or nothing is skipped
  const int* adapt;         // kAdaptSlots verdicts of spmm_locality_probe_kernel (L2-level panel plans with a workspace), or nullptr
  const int* order;         // nullptr: slot i is row i.  Else slot i is row order[i], i < nslots (an inspector's schedule:
  int nslots;               //   rows by descending length class — spmm_sched.hip); arithmetic per row unchanged
};

struct LongWs {
  long cap_e, cap_s, cap_p;
  size_t owner_off, partial_off, bytes;  // offsets in ints / bytes
  size_t adapt_off;                      // bytes: the locality probe's verdicts, behind everything else
};
LongWs long_ws_layout(int64_t nnz, int32_t N);
// lists the rows beyond la.thresh in la.ws (plans whose kernel lives in a file of its own and only skips; prepared lists; scheduled
// products without a prepared list)
int launch_find_long_rows(const int32_t* rowptr, int32_t M, const LongArg& la, hipStream_t s);
// the follow-up launch of a product that splits its long rows: sums the listed rows, combines, resets the counters
int launch_long_rows(int* ws, const LongWs& lw, const int32_t* rowptr, const int32_t* col, const float* val, const float* B,
                     float* C, int32_t N, int64_t ldb, int64_t ldc, const float* bias, bool reset, hipStream_t s);
// (its body: one 8-wave workgroup per group of 16 chains and 64 columns, the chains staged through LDS at the CU's gather rate, or
// one wave per chain once the list is long — spmm_heavy.hip; any width ≥ 4, any 4-byte alignment)
int launch_long_rows_staged(int* ws, const LongWs& lw, const int32_t* rowptr, const int32_t* col, const float* val,
                            const float* B, float* C, int32_t N, int64_t ldb, int64_t ldc, const float* bias, bool reset,
                            hipStream_t s);

// ---- column-panel passes (spmm_panels.hip) -------------------------------------------------------------------------
int launch_locality_probe(const int32_t* rowptr, const int32_t* col, int32_t M, int64_t ldb, double b_bytes, int* verdicts,
                          const int32_t* order, hipStream_t s);  // order (may be null): a schedule's slot → row map
int launch_panels(int panels, const int* rowptr, const int* col, const float* val, const float* B, float* C, int M, int K,
                  int N, long ldb, long ldc, const float* bias, LongArg la, hipStream_t s);
// 256-column tiles dealt XCD-aware × row panels of B (MI_SPMM_COLTILE_PANELS)
int launch_coltile_panels(int panels, const int* rowptr, const int* col, const float* val, const float* B, float* C, int M,
                          int K, int N, long ldb, long ldc, const float* bias, LongArg la, hipStream_t s);
int group_panel_count(int variant);
int launch_group_panels(int panels, const int* rowptr, const int* col, const float* val, const float* B, float* C, int M,
                        int K, int N, long ldb, long ldc, const float* bias, LongArg la, hipStream_t s);

// ---- heavy rows of a schedule (spmm_heavy.hip) ----------------------------------------------------------------------
// One 8-wave workgroup per slot of la.order and 64 columns: the row's B rows gathered by the loader waves into LDS, up to 224
// entries at a time, and summed from there by the chain wave — ONE chain per output element in CSR order (the same bits as every
// other kernel).  N ≥ 4, any 4-byte alignment; rows beyond la.thresh are skipped and listed as everywhere else.
int launch_heavy_rows(const int32_t* rowptr, const int32_t* col, const float* val, int32_t M, int32_t N, const float* B,
                      int64_t ldb, float* C, int64_t ldc, const float* bias, LongArg la, hipStream_t s);
// both in ONE launch (a prepared list beside a schedule's heavy slots would otherwise wait for each other on the stream):
// ws == nullptr: no list; heavy.nslots == 0: no heavy slots
int launch_staged_rows(int* ws, const LongWs& lw, bool reset, const LongArg& heavy, const int32_t* rowptr, const int32_t* col,
                       const float* val, const float* B, float* C, int32_t N, int64_t ldb, int64_t ldc, const float* bias,
                       hipStream_t s);

// ---- plan rules (spmm_plan.hip; host only) --------------------------------------------------------------------------
struct Shape {
  bool vec4_ok, vec2_ok, wave_ok;
};
Shape classify(int32_t N, int64_t ldb, int64_t ldc, int64_t strideB, int64_t strideC, const float* B, const float* C);
int choose_variant(const Shape& sh, int64_t nnz, int32_t batch, int32_t M, int32_t K, int32_t N, int64_t ldb);
int coltile_width(int32_t M, int32_t K, int32_t N, int64_t ldb);
int coltile_panels(int32_t M, int32_t K, int32_t N, int64_t ldb, int64_t nnz);

// ---- the dispatcher (spmm_csr.hip), as the scheduled entry points call it (spmm_sched.hip) --------------------------------
// Host view of an inspector's row schedule (built once per matrix by spmm_sched.hip; the arithmetic per row is untouched).
struct RowSchedule {
  const int32_t* order;   // device, `rows` entries: slot → row, rows by descending length class
  int32_t rows;           // M
  int32_t heavy;          // slots [0, heavy): the rows longer than heavy_len — a launch of their own, more gathers in flight per row
  int32_t heavy_len;
  bool active;            // false: the matrix has no skew worth an indirection (short, alike rows) — products run unscheduled
  hipStream_t side;       // the launch(es) of the rest run on this stream beside the heavy launch (nullptr: in line, behind it)
  hipEvent_t fork, join;  // fork / join of `side` against the caller's stream
};
int spmm_dispatch(int variant, const int32_t* rowptr, const int32_t* col, const float* val, int64_t nnz, int32_t batch,
                  int32_t M, int32_t K, int32_t N, const float* B, int64_t ldb, int64_t strideB, float* C, int64_t ldc,
                  int64_t strideC, const float* bias, void* workspace, size_t workspace_bytes, hipStream_t s,
                  int long_mode = MI_LONG_ROWS_AUTO, const RowSchedule* sched = nullptr);

}  // namespace mi

#endif  // MI_SPMM_INTERNAL_H_
