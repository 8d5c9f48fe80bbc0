// Which kernel MI_SPMM_AUTO runs for a problem: the plan rules of the row-split SpMM (host code only; fitted on MI355X —
// tools/bench_plans.py, tools/plan_grid.py, tools/bench_hbm_regime.py; profiles/r0*_plan_*.log).  Contract: include/mi_spmm.h
// (mi_spmm_csr_f32_plan and friends).  The reference has one kernel for every shape (src/naive_sparse_mm.cu:104-136).
#include "spmm_internal.h"

namespace mi {

// Tile width (columns) for the XCD-aware column-tiled launch, or 0 when it does not apply.
// The K-row slice of B an XCD gathers from should fit its 4 MiB L2; the tiles must spread evenly
// over the 8 XCDs and each tile needs enough row blocks to occupy an XCD's 32 CUs.  Measured on
// MI355X (tools/bench_wide.py): 4096² 1 % 0.344 → 0.129 ms (21 TB/s of gathers served by L2),
// 65536×8192 × 1024 0.5 % 1.33 → 0.51 ms, 8192² 1 % 2.96 → 2.13 ms, 16384² 1 % 24.2 → 20.7 ms.
int coltile_width(int32_t M, int32_t K, int32_t N, int64_t ldb) {
  const long slice_budget = 4L << 20;
  if (N < 512 || M < 512 || (long)K * ldb * 4 <= (8L << 20)) return 0;  // narrow, short, or B small as it is
  for (int w : {256, 128, 64}) {
    if ((long)K * w * 4 > slice_budget || N < 8 * w) continue;
    const int tiles = (N + w - 1) / w, rounds = (tiles + 7) / 8;
    if (tiles * 5 >= rounds * 8 * 4) return w;  // at least 80 % of the XCD × round slots used
  }
  return 0;
}

// Wide N with K too tall for a K × 256 slice to fit an L2: block BOTH ways — 256-column tiles
// dealt XCD-aware and K cut into row panels of B (one launch per panel, C carried through memory
// like the two-panel path), so each XCD gathers from a (K/P) × 256 slice of ≈3 MiB.
// Returns the number of panels, or 0 when the plan does not apply.
int coltile_panels(int32_t M, int32_t K, int32_t N, int64_t ldb, int64_t nnz) {
  if (N % 256 != 0 || N < 2048 || M < 512 || ldb != N) return 0;
  if ((long)K * 1024 <= (4L << 20)) return 0;              // plain column tiling already fits
  const int panels = (int)(((long)K * 1024 + (3L << 20) - 1) / (3L << 20));
  if (panels > 16 || nnz < 8L * panels * M) return 0;       // too many C round trips for the work per pass
  const int tiles = N / 256, rounds = (tiles + 7) / 8;
  return tiles * 5 >= rounds * 8 * 4 ? panels : 0;
}

// L2-level panel blocking for N = 256 (one wave per row): when B is too large for the L2s (> 6 MiB) every
// gathered row comes from the Infinity Cache or HBM and the one-pass kernels drop from ≈13 to ≈5 TFLOP/s.
// Cutting K into P panels of ≈4 MiB (one launch per panel, all CUs on the same panel, C carried through
// memory) keeps the gathers in L2 — the same mechanism as the two Infinity-Cache panels of config C3, one
// level down.  Measured (tools/bench_plans.py, profiles/r02_plan_choice.log): 16384² × 256 at 10 %: 1.02 ms
// with 4 panels vs 1.77 one-pass (slab 1.46); 8192 × 32768 × 256 at 10 %: 1.14 vs 1.83; 8192 × 65536 × 256
// at 5 %: 1.50 (8 panels) vs 2.43; 8192 × 131072 × 256 at 1 %: 1.04 vs 1.27; at 0.5 % (82 non-zeros per row):
// 0.091 vs 0.141; B = 4 MiB: one pass stays ahead.  N = 512 is left to the XCD-aware column tiles, which need
// no second pass over C (16384² × 512 at 0.5 %: 0.150 ms vs 0.206 with panels) and to the slab plan.
// Returns the number of panels (2, 3, 4, 5, 6 or 8), or 0 when the plan does not apply.
int l2_panels(int32_t M, int32_t K, int32_t N, int64_t ldb, int64_t nnz) {
  // N = 512 / 1024 only where no column-tile width keeps an XCD's slice of B in its L2 (K too tall): 32768² × 512 at
  // 0.3 %: 0.64 ms with 8 panels vs 0.86 one-pass
  if (N != 256 && !((N == 512 || N == 1024) && coltile_width(M, K, N, ldb) == 0)) return 0;
  const double b_bytes = (double)K * (double)ldb * 4.0;
  // (from 32 Ki rows already at 4.5 MiB — B then fills the L2s to the brim and the many rows keep evicting it: 170752 × 2816 ×
  // 512 with 129 per row, 5.5 MiB: one pass 1.90 ms, two panels 1.50; 134912 × 5376 × 256, 5.2 MiB: 0.74 → 0.63)
  // (N = 512 / 1024 from 6 MiB like N = 256 — the column tiles only start beyond 8 MiB: 14592 × 4096 × 512 with 138 per row,
  // 8.0 MiB: one pass 0.249 ms, two panels 0.162)
  if (b_bytes <= (M >= 32768 ? 4.5 : 6.0) * 1024 * 1024 || b_bytes > 192.0 * 1024 * 1024) return 0;
  // panels turn re-gathers into L2 hits: below ≈ 32 gathers per row of B nothing is won (round 5, tools/plan_grid.py:
  // 1280 × 14592 × 256 with 38 per row — 3.3 gathers per row of B — one pass 0.008 ms, the four panels taken until then 0.019)
  // … and a launch of fewer waves than the chip holds is latency-bound: cutting it into passes multiplies that (1536 × 9728
  // × 256 with 499 per row: one pass 0.044 ms, three panels 0.080) — unless B is far beyond the L2s (3328 × 27904 × 512 with 240 per
  // row, 54 MiB: 0.170 → 0.129)
  // (with a B that half fits the L2s as it is, up to ≈ 10 Ki rows: 4096 × 13056 × 256 with 393 per row one pass 0.065 ms, two
  // panels 0.080 – 0.096; 8192 × 131072 × 256 with 1311 per row, 128 MiB: 1.27 → 0.58)
  if (nnz < 24L * K || (M < 10240 && b_bytes < 32.0 * 1024 * 1024)) return 0;
  int p = (int)((b_bytes + (4 << 20) - 1) / (4 << 20));
  p = p > 6 ? 8 : p;
  const int p_by_size = p;
  while (p >= 2 && nnz < 8L * p * M) p = p > 6 ? 6 : p - 1;  // each pass carries C once: it needs work to pay for that
  // … and panels that short rows leave too large for the L2s only add passes (162560 × 106240 × 256 with 45 per row, 104 MiB:
  // five panels of 21 MiB 1.11 ms, one pass 1.03; eight panels of 19 – 21 MiB for long rows pay: 19712 × 38400 × 1024 with 502
  // per row 5.18 → 4.36 ms)
  if (p >= 2 && p < p_by_size && b_bytes / p > 16.0 * 1024 * 1024) return 0;
  return p >= 2 ? p : 0;
}

// Infinity-Cache panels for B beyond the cache (≥ 768 MiB): how many column panels K is cut into, or 0 for one pass.
// Measured on MI355X over N 64 … 512, K = M 1 … 4 M, 20 / 100 / 400 non-zeros per row, every plan on the same operands and
// the same output buffer (tools/bench_hbm_regime.py --variants, profiles/r05_hbm_regime.log): panels pay when a panel's
// slice of B is ≈ 0.5–0.7 GiB — about twice the cache, so that about half of a pass's gathers hit it — i.e. P ≈ |B| / 683 MiB
// (1 GiB: 2 panels −6 %; 2 GiB: 3–4 panels −5 … −11 %; 4 GiB: 6 panels −9 … −11 %; fewer, larger panels than that bring
// nothing: 4 GiB in 2–3 panels +1 … +2 %), and when the rows are long enough to carry C through memory once more per
// panel (2 (P − 1) row passes against nnz/M gathered rows per row: 20 per row never pays, 100 per row pays up to 6
// panels).  Beyond ≈ 6 GiB no panel count helps (8 GiB in 8 panels: ± 1 %): one pass at the HBM random-row rate.
// The launch must also re-touch a panel often enough to keep it resident (gathered bytes ≥ 8 × |B|), which excludes the
// short row blocks of a sharded run.
int ic_panels(int64_t nnz, int32_t M, int32_t K, int32_t N, int64_t ldb) {
  const double b_bytes = (double)K * (double)ldb * 4.0;
  if (b_bytes < 768.0 * 1048576.0 || M <= 0) return 0;
  if ((double)nnz * (double)N < 8.0 * (double)K * (double)ldb) return 0;
  const double want = b_bytes / (683.0 * 1048576.0);
  if (want > 9.0) return 0;
  int best = 2;
  for (int p : {2, 3, 4, 6, 8})
    if ((p - want < 0 ? want - p : p - want) <= (best - want < 0 ? want - best : best - want)) best = p;  // nearest, ties up
  return nnz >= 16L * best * M ? best : 0;
}

// L2-level panels for the lane-group panel kernel: B beyond the L2s but inside the Infinity Cache (6 MiB < |B| ≤ 128 MiB).
// Fitted on tools/probes/l2_regime_shapes*.sh (17 shapes) and tools/plan_grid.py (120 random shapes, every plan pinned;
// profiles/r05_l2_regime_plans.log, r05_plan_grid.log):
//   * panels of ≈ 6 MiB are best or within a few per cent of it over B = 8 … 128 MiB at N = 192 … 768 (B = 12 MiB: 2 panels,
//     24 MiB: 3–4, 48 MiB: 8, 128 MiB: 8); every pass walks the row's columns up to its panel, which is why rows of thousands
//     of entries want half as many (8192 × 65536 × 256 at 5 %, 3277 per row: 4 panels 1.13 ms, 8 panels 1.19);
//   * a pass carries C once (two rows' worth of gathers) and costs a launch: it needs ≥ 8 entries per row, more when the
//     product is small (16384² × 256 with 82 per row: 2 panels 0.085 ms, 3 panels 0.096; 230656 × 9472 × 768 with 66 per row:
//     2 panels 4.70 ms, 4 panels 3.38, 6 panels 3.12; 22272 × 14080 × 384 with 41 per row: one pass 0.158, 3 panels 0.127) —
//     8 + 200 000 / M;
//   * panels turn RE-gathers into L2 hits, the first touch of a row of B comes from beyond either way: with fewer than ≈ 24
//     gathers per row of B nothing is won (1280 × 14592 × 256 with 38 per row, 3.3 gathers per row of B: one pass 0.008 ms,
//     four panels 0.019; 1536 × 39424 × 768 with 715 per row, 28 per row of B: two panels 0.296 against 0.344);
//   * up to 192 MiB of B (242176 × 188928 × 192, 138 MiB, 1010 per row: one pass 24.7 ms, 8 panels 19.4), up to 384 MiB for
//     rows of ≥ 256 entries (57344 × 326656 × 256, 319 MiB, 869 per row: 6.79 → 6.00; with 38 per row at 206 MiB panels lose
//     7 – 30 %); between that and the Infinity-Cache regime (768 MiB) panels move a product by ± 5 %: one pass;
//   * B barely beyond the L2s (< 12 MiB) needs ≥ 48 entries per row (95488 × 6144 × 320 with 21 per row, 7.5 MiB: one pass
//     0.187 ms, two panels 0.215; 309504 × 4864 × 384 with 65 per row, 7.1 MiB: 1.86 → 1.34);
//   * N ≤ 128 (rows of B of ≤ 512 bytes, several rows per wave): two panels (four for ≥ 500 K rows of ≥ 128 entries and
//     ≥ 32 MiB), only for launches that are throughput-bound — ≥ 28 K rows at N = 64 (four rows per wave), ≥ 60 K rows beyond
//     (two); ≥ 48 entries per row, ≥ 48 gathers per row of B, 8 … 128 MiB (129536 × 22784 × 128 1.53 → 1.24 ms, 961024 × 82176
//     × 128 10.2 → 8.0, 154368 × 278528 × 64 1.13 → 0.99, 55040 × 268800 × 64 0.63 → 0.50, 79872 × 34048 × 64 0.49 → 0.37,
//     67328 × 90624 × 100 2.01 → 1.45); with fewer rows a pass is latency-bound and splitting it only multiplies that (16384 ×
//     65536 × 128: one pass 0.31 ms, two panels 0.37; 12032 × 67328 × 64: 0.077 vs 0.153; 24832 × 25856 × 96: 0.094 vs 0.151);
//     N = 32 … 60 only for ≥ 96 Ki rows of ≥ 128 entries;
//   * fewer than ≈ 10 Ki rows: only with B far beyond the L2s (≥ 32 MiB) and ≥ 6e8 multiply-adds in the product (4096 × 13056 ×
//     256 with 393 per row, 12.8 MiB: one pass 0.065 ms, two panels 0.093; 4096 × 49920 × 192 with 478 per row: 0.092 vs
//     0.141; 2304 × 59904 × 384 with 847 per row, 88 MiB: 0.296 → 0.232), and in at most 3 (< 4 Ki rows) or 4 (< 8 Ki) passes.
// Returns 2, 3, 4, 6 or 8, or 0.
int l2_group_panels(int32_t M, int32_t K, int32_t N, int64_t ldb, int64_t nnz) {
  const double b_bytes = (double)K * (double)ldb * 4.0, mib = 1048576.0;
  if (b_bytes <= 6.0 * mib || b_bytes > (nnz >= 256L * M ? 384.0 : 192.0) * mib || M <= 0) return 0;
  if (nnz < 24L * K || (b_bytes < 12.0 * mib && nnz < 48L * M)) return 0;
  // (round 6, unseen grid: 3584 × 34048 × 384 with 456 per row — 6.3e8 multiply-adds, 50 MiB — one pass 0.234 ms, four panels 0.169)
  if (M < 10240 && (b_bytes < 32.0 * mib || (double)nnz * (double)N < 6e8)) return 0;
  const long per_pass = 8 + 200000L / M;
  if (N <= 128) {
    if (N < 64)  // N = 32 … 60 (16-lane groups, half of them idle at 32): many long rows only — 117248 × 295936 × 32 with 210 per row
                 // 0.50 → 0.41 ms, 290048 × 136192 × 32 with 343 per row 1.76 → 1.43; short rows lose (17 per row: 0.18 vs 0.34)
      return (N >= 32 && M >= 98304 && nnz >= 128L * M && nnz >= 48L * K && b_bytes >= 16.0 * mib && b_bytes <= 128.0 * mib) ? 2 : 0;
    if (!(nnz >= 48L * M && nnz >= 48L * K && b_bytes >= 8.0 * mib && b_bytes <= 128.0 * mib)) return 0;
    // (between 64 and 128 columns from 24 MiB only: below that uniform columns are level — 271104 × 29952 × 96, 11 MiB: 2.21 vs
    // 2.20 ms — and banded or power-law ones lose, 1.36 vs 1.58 / 1.49 vs 2.05)
    // (round 6, unseen grid: with ≥ 400 K long rows already from 12 MiB — 658688 × 37632 × 100 with 169 per row, 14.4 MiB: one pass
    // 5.25 ms, two panels 3.86)
    if (N > 64 && N < 128 && b_bytes < ((M >= 400000 && nnz >= 128L * M) ? 12.0 : 24.0) * mib) return 0;
    if (M < (N == 64 ? 28000 : 60000)) return 0;
    // rows long enough for two passes: 48 at N = 64, 96 at N = 128 (48 from 96 Ki rows; config C2 — 65536² × 128, 65 per row —
    // is level: 0.2516 one pass, 0.2476 in two panels, and stays one pass), 160 between (65536² × 96 with 100 per row: 0.249 vs 0.265)
    const long len_min = N == 64 ? 48 : (N == 128 ? (M >= 98304 ? 48 : 96) : 160);
    if (nnz < len_min * M) return 0;
    return (M >= 500000 && b_bytes >= 32.0 * mib && nnz >= 128L * M) ? 4 : 2;
  }
  // (panels of ≈ 4 MiB from 64 Ki rows: 316160 × 10240 × 384 with 299 per row, 15 MiB: 2 panels 9.35 ms, 4 panels 6.85; 349952 ×
  // 19456 × 256, 19 MiB: 3 panels 7.45, 4 panels 6.60; 259840 × 18176 × 384, 26.6 MiB: 4 panels 4.21, 6 panels 3.69)
  const double want = b_bytes / ((nnz >= 2048L * M ? 12.0 : (M >= 65536 ? 4.0 : 6.0)) * mib);
  int p = 2;
  for (int c : {2, 3, 4, 6, 8})
    if ((c < want ? want - c : c - want) < (p < want ? want - p : p - want)) p = c;
  static const int kLower[9] = {0, 0, 0, 2, 3, 0, 4, 0, 6};
  const int p_by_size = p;
  while (p >= 2 && nnz < per_pass * p * M) p = kLower[p];
  // panels that short rows leave too large for the L2s only add passes (as in l2_panels): 339200 × 115456 × 192 with 40 per row,
  // 85 MiB, in the four panels the rows pay for — 21 MiB each — leaves the L2s with 1.01 × its algorithmic bytes, one pass with
  // 0.96 × (profiles/r05_l2_panel_traffic.log): 1.44 vs 1.40 ms, and 1.21 vs 0.97 with power-law columns
  if (p >= 2 && p < p_by_size && b_bytes / p > 16.0 * mib) return 0;
  // a few thousand rows are fewer waves than the chip holds: every further pass is one more latency-bound launch (3072 × 20480 ×
  // 768 with 739 per row: 8 panels 0.365 ms, 3 panels 0.310; 2304 × 59904 × 384 with 847 per row: 0.278 vs 0.231; from 8 Ki rows
  // eight panels are the best again: 8192 × 131072 × 256 with 1311 per row 0.58 ms, four panels 0.74)
  const int p_cap = M < 4096 ? 3 : (M < 8192 ? 4 : 8);
  while (p > p_cap) p = kLower[p];
  return p >= 2 ? p : 0;
}

Shape classify(int32_t N, int64_t ldb, int64_t ldc, int64_t strideB, int64_t strideC, const float* B,
               const float* C) {
  Shape sh;
  sh.vec4_ok = (N % 4 == 0) && (ldb % 4 == 0) && (ldc % 4 == 0) && (strideB % 4 == 0) &&
               (strideC % 4 == 0) && mi::aligned16(B) && mi::aligned16(C);
  sh.vec2_ok = (N % 2 == 0) && (ldb % 2 == 0) && (ldc % 2 == 0) && (strideB % 2 == 0) &&
               (strideC % 2 == 0) &&
               ((reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(C)) & 7u) == 0;
  sh.wave_ok = sh.vec4_ok && (N == 256 || N == 512 || N == 1024);
  return sh;
}

#ifndef MI_SPMM_LDSB_MIN_ROW
#define MI_SPMM_LDSB_MIN_ROW 4L  // mean non-zeros per row from which MI_SPMM_LDS_B is AUTO's choice (tools/bench_attn_csr.py)
#endif
// The kernel AUTO resolves to.
int choose_variant(const Shape& sh, int64_t nnz, int32_t batch, int32_t M, int32_t K, int32_t N,
                   int64_t ldb) {
  // B beyond the 256 MiB Infinity Cache: K in column panels, one launch per panel (ic_panels: how many, fitted on
  // tools/bench_hbm_regime.py, profiles/r05_hbm_regime.log) — the one-wave-per-row panel kernel for N = 256 / 512 / 1024,
  // the lane-group panel kernel for N ≤ 128
  if (batch == 1 && sh.vec4_ok) {
    const int p = ic_panels(nnz, M, K, N, ldb);
    if (p > 0 && sh.wave_ok) {
      static const int kWave[9] = {0, 0, MI_SPMM_PANELS_2, MI_SPMM_PANELS_3, MI_SPMM_PANELS_4, 0, MI_SPMM_PANELS_6, 0, MI_SPMM_PANELS_8};
      return kWave[p];
    }
    if (p > 0 && N >= 36 && N <= 1024) {  // every other width, N % 4 == 0 (narrower rows: 16-lane groups would idle half their lanes)
      static const int kGroup[9] = {0, 0, MI_SPMM_GROUP_PANELS_2, MI_SPMM_GROUP_PANELS_3, MI_SPMM_GROUP_PANELS_4, 0,
                                    MI_SPMM_GROUP_PANELS_6, 0, MI_SPMM_GROUP_PANELS_8};
      return kGroup[p];
    }
  }
  if (N < 4) return MI_SPMM_NARROW;
  // Many small products (or one tall one) whose B fits a CU's LDS: gather from LDS instead of from the L2s
  // (spmm_ldsb.hip).  It pays once rows are long enough to amortise copying B per workgroup.
  // (tools/bench_attn_csr.py, profiles/r04_attention_csr.log: 384 × 512² × 64 at 10 % kept 0.121 → 0.050 ms, at
  // 1 % 0.038 → 0.026; N = 256 has the one-wave-per-row kernels, whose col / val travel through scalar registers:
  // 65536 × 128 × 256 at 5 % 0.020 ms there vs 0.026 here, at 25 % 0.069 (0.055 as the slab plan) vs 0.035)
  // With the quad form the plan pays from ≈4 non-zeros per row whatever the number of column tiles (tools/bench_plans.py,
  // one tall matrix: 65536 × 256 × 128 at 3 % — 7.7 per row, two tiles — 0.014 ms against 0.025 for the group kernel;
  // 65536 × 128 × 256 at 5 % — 6.4 per row, four tiles — 0.021 against 0.025 for the one-wave-per-row kernel and at 10 %
  // 0.025 against 0.040 for the slab plan; 32768 × 512 × 128 at 1 %: 0.010 against 0.014; 131072 × 512 × 64 at 0.5 % — 2.6
  // per row — level with the group kernel)
  // (rows: enough 256-row units for the persistent grid — or, from 8192 rows, enough non-zeros that the gathers decide:
  // 12 heads of 1024 tokens at 25 % kept 0.041 → 0.020 ms, 24 × 512² 0.024 → 0.017, tools/probes/small_batch_probe.py)
  if (sh.vec4_ok && mi::spmm_ldsb_fits(K, N) &&
      ((long)batch * M >= 16384 || ((long)batch * M >= 8192 && nnz >= 1500000)) &&
      nnz >= MI_SPMM_LDSB_MIN_ROW * (long)batch * M)
    return MI_SPMM_LDS_B;
  // N = 20 … 32, up to 16 Ki rows of ≥ 32 entries: 16 lanes per row (half of them idle) instead of 8 — the launch form of the
  // column tiles with ONE tile: twice the waves for a launch that has too few — was ahead on every such shape of
  // tools/plan_grid.py (2816 × 6400 × 32 with 506 per row 0.066 → 0.046 ms, 14336 × 90112 × 32 with 65 per row 0.021 → 0.018);
  // with many rows it loses (969984 × 2304 × 32 with 63 per row: 0.49 vs 0.72)
  if (batch == 1 && sh.vec4_ok && N > 16 && N <= 32 && nnz >= 32L * M && M <= 16384) return MI_SPMM_COLTILE;
  int lp = (sh.wave_ok && batch == 1) ? l2_panels(M, K, N, ldb, nnz) : 0;
  // L2-level panels on the lane-group panel kernel (round 5): for the widths the one-wave-per-row panel kernel does not take
  // (128 < N ≤ 1024 other than 256 / 512 / 1024) and, at N = 256, for long rows — it beats the wave-per-row panel kernel there
  // (its passes stop at the first chunk behind their panel; 8192 × 131072 × 256 at 1 %: 0.99 → 0.58 ms, 16384² × 256 at 10 %:
  // 1.04 → 0.96) and loses on short rows (65536 × 16384 at 0.3 %, 49 per row: 0.21 vs 0.24).  N ≤ 128: two panels in a narrow band of B only
  // (l2_group_panels).  tools/probes/l2_regime_shapes*.sh, tools/plan_grid.py; profiles/r05_l2_regime_plans.log, r05_plan_grid.log.
  int gp = 0;
  if (sh.vec4_ok && batch == 1 && N >= 32 && N <= 1024 && (N == 256 ? nnz >= 224L * M : !sh.wave_ok)) gp = l2_group_panels(M, K, N, ldb, nnz);
  if (gp > 0) lp = 0;
  // Moderate density: stage B through LDS (spmm_slab.hip) when its cost model beats the L2-blocked
  // row-split plans.  Fitted on MI355X (tools/bench_density.py, tools/bench_plans.py): a slab
  // workgroup (128 rows × 256 columns) spends ≈2.35 µs + 34 µs × density per 64-row slab of B, one
  // workgroup per CU at a time; the row-split plans sustain ≈13 TFLOP/s with L2 blocking (N ≥ 512)
  // and ≈5 without.  E.g. 8192² × 8192 at 10 %: 5.9 ms slab vs 9.4 ms; 4096² × 2048 at 20 %: 0.60 vs
  // 0.96 ms; 8192² × 8192 at 3 %: 3.3 vs 3.0 ms (row-split kept).
  if (sh.vec4_ok && batch == 1 && K >= 64 && N >= 128 && nnz > 0) {
    const double wgs = (double)(((long)M + 127) / 128) * (double)(((long)N + 255) / 256);
    const double density = (double)nnz / ((double)M * (double)K);
    // whole rounds of workgroups up to four of them (312 workgroups take two rounds, not 1.22: 13312 × 2304 × 768 at 7 % measured
    // 0.347 ms = 2 × 36 slabs × 4.8 µs; round 5, tools/plan_grid.py); beyond that the tail averages out
    const double rounds = wgs <= 256.0 ? 1.0 : (wgs < 1024.0 ? (double)(((long)wgs + 255) / 256) : wgs / 256.0);
    const double t_slab = rounds * (double)(((long)K + 63) / 64) * (2.35e-6 + 34e-6 * density);
    // … and as much for narrower N while B (≤ 8 MiB) stays in the L2s: 16384 × 4096 × 256 at 10 %: row-split
    // 0.25 ms (13.7 TFLOP/s) vs slab 0.38; 16384 × 768 × 128 at 30 %: 0.082 vs 0.164; the 5 TFLOP/s figure
    // holds once B streams from the Infinity Cache or HBM (profiles/r02_plan_choice.log)
    const bool b_in_l2 = (double)K * (double)ldb * 4.0 <= 8.0 * 1024 * 1024;
    // with L2 panels the row-split plan gathers at the L2 rate and carries C (2·lp − 1) times
    const int panels = lp > 0 ? lp : gp;
    // (round 5, fitted on the grids' slab-vs-rows misroutes: the panel plans at 15 TFLOP/s with C carried at 5 TB/s — 97792 ×
    // 3584 × 768 with 140 per row: three panels 1.55 ms, slabs 1.84; "B in every L2 at once" up to 3.5 MiB — 61184 × 3072 × 256:
    // one pass 0.327 ms = 15.9 TFLOP/s, slabs 0.407; and the lane-group kernel's idle lanes where N is not a whole number of
    // 256-column tiles — 15104 × 1536 × 320: 0.327 ms = 11 TFLOP/s, slabs 0.270; N ≥ 512 gathers at the L2 rate only where
    // a column-tile plan keeps an XCD's slice of B in its L2 — 25856 × 57344 × 1024 with 1003 per row, 224 MiB, K too tall for
    // any tile width: one pass 13.95 ms = 3.8 TFLOP/s, slabs 9.39)
    const double lanes_used = sh.wave_ok || N <= 256 ? 1.0 : (double)N / (256.0 * (double)((N + 255) / 256));
    // (… but N ≥ 512 in eight panels of 8 – 10 MiB, many rows, a few per cent dense: 7 TFLOP/s — 78848 × 19456 × 1024 at 3.7 %:
    // eight panels 18.1 ms, slabs 10.8; 25856 × 16640 × 1024 at 4.2 %: 4.9 vs 3.8; with 5120 rows the panels stay ahead, 0.94 vs 1.21)
    const double panel_rate = (lp > 0 && N >= 512 && M >= 16384 && density >= 0.03) ? 7e12 : 15e12;
    const double t_rows = (panels > 0 ? 2.0 * (double)nnz * (double)N / panel_rate + (2.0 * panels - 1.0) * (double)M * (double)N * 4.0 / 5e12
                                  : 2.0 * (double)nnz * (double)N /
                                        ((double)K * (double)ldb * 4.0 <= 3.5 * 1024 * 1024 ? 15e12  // B in every L2 at once
                                         : (b_in_l2 || (N >= 512 && (coltile_width(M, K, N, ldb) > 0 || coltile_panels(M, K, N, ldb, nnz) > 0))) ? 13e12
                                                                                                                                         : 5e12)) / lanes_used;
    // below ≈100 workgroups too few CUs have work for the model to hold
    if (wgs >= 96.0 && t_slab < t_rows) return MI_SPMM_SLAB;
  }
  if (gp > 0) {
    static const int kGroupOf[9] = {0, 0, MI_SPMM_GROUP_PANELS_2, MI_SPMM_GROUP_PANELS_3, MI_SPMM_GROUP_PANELS_4, 0,
                                    MI_SPMM_GROUP_PANELS_6, 0, MI_SPMM_GROUP_PANELS_8};
    return kGroupOf[gp];
  }
  if (lp > 0) {
    static const int kVariantOf[9] = {0, 0, MI_SPMM_PANELS_2, MI_SPMM_PANELS_3, MI_SPMM_PANELS_4, MI_SPMM_PANELS_5,
                                      MI_SPMM_PANELS_6, 0, MI_SPMM_PANELS_8};
    return kVariantOf[lp];
  }
  if (sh.vec4_ok && batch == 1 && coltile_panels(M, K, N, ldb, nnz) > 0) return MI_SPMM_COLTILE_PANELS;
  if (sh.vec4_ok && batch == 1 && coltile_width(M, K, N, ldb) > 0) return MI_SPMM_COLTILE;
  // N = 256, one pass, a few thousand rows, B beyond the L2s but small: the lane-group kernel's whole-wave form (its col / val
  // travel by vector load + ds_bpermute, the one-wave-per-row kernel's through the scalar unit) is ahead on such latency-bound
  // launches — 4096 × 16384 × 256 with 164 per row 0.052 → 0.035 ms, 4096 × 13056 with 393 per row 0.078 → 0.065, 8960 × 8704
  // with 213 per row 0.117 → 0.097, 16384² with 20 per row 0.047 → 0.040; with ≤ 3000 rows the other way round (1536 × 9728
  // with 499 per row: 0.044 vs 0.063); tools/plan_grid.py, tools/probes/l2_regime_shapes3.sh
  if (sh.wave_ok && N == 256 && batch == 1 && M >= 3500 && M <= 16384) {
    const double b_bytes = (double)K * (double)ldb * 4.0;
    if (b_bytes > 6.0 * 1048576.0 && b_bytes <= 64.0 * 1048576.0) return MI_SPMM_GROUP_VEC4;
  }
  if (sh.wave_ok) return MI_SPMM_WAVE_ROW_U8;
  // rows that do not start on 16 bytes (N % 4 != 0, odd leading dimensions, offset views): four floats per lane all the
  // same, on dword-aligned 16-byte accesses (2 M rows, 100 per row: N = 77 0.38 → of 8 TB/s with one float per lane, 130: 0.35, 250: 0.49)
  return sh.vec4_ok ? MI_SPMM_GROUP_VEC4 : (N >= 4 ? MI_SPMM_GROUP_VEC4U : MI_SPMM_GROUP_SCALAR);
}

}  // namespace mi

extern "C" {

int mi_spmm_auto_splits_long_rows(int64_t nnz, int32_t M, int32_t K, int32_t N, const float* B, int64_t ldb,
                                  const float* C, int64_t ldc) {
  const int v = mi_spmm_csr_f32_plan(nnz, M, K, N, B, ldb, C, ldc);
  if (v < 0) return v;
  return (nnz > mi::kLongRowThreshold && v != MI_SPMM_NARROW && v != MI_SPMM_SLAB) ? 1 : 0;
}

int mi_spmm_csr_batched_f32_plan(int64_t nnz_total, int32_t batch, int32_t M, int32_t K, int32_t N, const float* B,
                                 int64_t ldb, int64_t strideB, const float* C, int64_t ldc, int64_t strideC) {
  if (M < 0 || K < 0 || N < 0 || nnz_total < 0 || batch < 0 || strideB < 0 || strideC < 0) return MI_EINVAL;
  return mi::choose_variant(mi::classify(N, ldb, ldc, strideB, strideC, B, C), nnz_total, batch, M, K, N, ldb);
}

int mi_spmm_csr_f32_plan(int64_t nnz, int32_t M, int32_t K, int32_t N, const float* B, int64_t ldb,
                         const float* C, int64_t ldc) {
  if (M < 0 || K < 0 || N < 0 || nnz < 0) return MI_EINVAL;
  return mi::choose_variant(mi::classify(N, ldb, ldc, 0, 0, B, C), nnz, 1, M, K, N, ldb);
}

int mi_spmm_variant_launches(int variant) {
  static const int kPanels[] = {2, 3, 4, 5, 6, 8};
  if (variant >= MI_SPMM_PANELS_2 && variant <= MI_SPMM_PANELS_8) return kPanels[variant - MI_SPMM_PANELS_2];
  if (variant >= MI_SPMM_GROUP_PANELS_2 && variant <= MI_SPMM_GROUP_PANELS_8) return mi::group_panel_count(variant);
  if (variant == MI_SPMM_COLTILE_PANELS) return 0;  // one per row panel of B: depends on K
  return (variant > MI_SPMM_AUTO && variant < MI_SPMM_VARIANT_COUNT) ? 1 : MI_EINVAL;
}

const char* mi_spmm_variant_name(int variant) {
  switch (variant) {
    case MI_SPMM_AUTO: return "auto";
    case MI_SPMM_WAVE_ROW_U4: case MI_SPMM_WAVE_ROW_U8: case MI_SPMM_WAVE_ROW_U16: return "spmm_wave_row_kernel";
    case MI_SPMM_WAVE_ROW_VL: return "spmm_wave_row_vl_kernel";
    case MI_SPMM_GROUP_VEC4: case MI_SPMM_GROUP_VEC2: case MI_SPMM_GROUP_SCALAR: case MI_SPMM_COLTILE: case MI_SPMM_GROUP_VEC4U:
      return "spmm_group_kernel";
    case MI_SPMM_NARROW: return "spmm_narrow_kernel";
    case MI_SPMM_SLAB: return "spmm_slab_kernel";
    case MI_SPMM_LDS_B: return "spmm_ldsq_kernel";  // (its quad form; spmm_ldsb_kernel where only the 16-lane form covers the shape)
    case MI_SPMM_PANELS_2: case MI_SPMM_PANELS_3: case MI_SPMM_PANELS_4: case MI_SPMM_PANELS_5:
    case MI_SPMM_PANELS_6: case MI_SPMM_PANELS_8: case MI_SPMM_COLTILE_PANELS:
      return "spmm_wave_row_panel_kernel";
    case MI_SPMM_GROUP_PANELS_2: case MI_SPMM_GROUP_PANELS_3: case MI_SPMM_GROUP_PANELS_4: case MI_SPMM_GROUP_PANELS_6:
    case MI_SPMM_GROUP_PANELS_8: return "spmm_group_panel_kernel";
    default: return "unknown";
  }
}

}  // extern "C"
