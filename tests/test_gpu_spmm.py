"""CSR × dense SpMM: every plan, edge cases, long rows, panels, slab, bias epilogues.

Parity of the HIP path with the oracle — needs the MI355X (`-m gpu`).  Everything here calls the product path
(custom_mm → libmi_spmm.so → HIP kernels, or the C-ABI directly through ctypes) and compares with the CPU oracle on the
same seeded inputs: bit-exact where the oracle states the same summation order, rtol 1e-5 / atol 1e-8 (the reference
tests' torch.allclose defaults, tests/naive_kernel_test.py:36-37) against torch expectations and the golden fixtures.
"""
import ctypes
import sys

import numpy as np
import pytest
import torch

from gpu_helpers import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


def test_spmm_golden_bit_exact_vs_oracle(cmm, dev, golden, oracle_mod):
    for name in golden.cases("spmm"):
        c = golden.case(name)
        M, K = c["a"].shape
        expect = oracle_mod.spmm_csr(c["rowptr"], c["col"], c["val"], M, K, c["b"])
        for op in ("naive_spmm", "cusparse_mmul"):
            got = run_spmm(cmm, dev, c["rowptr"], c["col"], c["val"], M, K, c["b"], op)
            assert np.array_equal(got, expect), (name, op)
            assert np.allclose(got, c["c"], rtol=RTOL, atol=ATOL), (name, op)


@pytest.mark.parametrize("M,K,N,density", [
    (512, 1024, 256, 0.1),    # reference tests/cusparse_kernel_test.py:38
    (1024, 1024, 1024, 0.01), (1024, 2048, 512, 0.01), (2048, 1024, 512, 0.01),  # tiledsppm_kernel_test.py:34-39
    (333, 777, 256, 0.05), (333, 777, 512, 0.05), (65, 129, 1024, 0.2), (1000, 1000, 100, 0.02),
    (77, 300, 1, 0.1), (77, 300, 2, 0.1), (77, 300, 7, 0.1), (300, 77, 1031, 0.1), (5, 40, 2048, 0.5),
    (900, 3000, 64, 0.03), (900, 3000, 128, 0.03), (0 + 411, 2000, 96, 0.05),
])
def test_spmm_every_variant_is_bit_identical(capi, dev, oracle_mod, M, K, N, density):
    rowptr, col, val = oracle_mod.make_csr(M, K, density, seed=M + N)
    B = np.random.Generator(np.random.PCG64(N)).random((K, N), dtype=np.float32)
    expect = oracle_mod.spmm_csr(rowptr, col, val, M, K, B)
    d = [t(x, dev) for x in (rowptr, col, val, B)]
    ran = 0
    chain = oracle_mod.spmm_csr_chain(rowptr, col, val, M, K, B)  # explicit group variants ignore the N < 4 rule
    for variant in range(25):
        C = torch.full((M, N), float("nan"), device=dev)
        st = capi.mi_spmm_csr_f32_variant(variant, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), len(val), M, K,
                                          N, d[3].data_ptr(), N, C.data_ptr(), N,
                                          torch.cuda.current_stream().cuda_stream)
        if st == -1:  # variant does not cover this shape
            continue
        assert st == 0
        ran += 1
        assert np.array_equal(C.cpu().numpy(), chain if variant in (4, 5, 13) else expect), f"variant {variant}"
    assert ran >= 2  # AUTO plus the generic kernel at least


@pytest.mark.parametrize("M,K,N,density,kernel,panels", [
    (12288, 16384, 256, 0.004, "spmm_wave_row_panel_kernel", 4), (12288, 24576, 256, 0.003, "spmm_wave_row_panel_kernel", 6),
    (16384, 40960, 256, 0.002, "spmm_wave_row_panel_kernel", 8), (12288, 8192, 256, 0.005, "spmm_wave_row_panel_kernel", 2),
    # ≥ 224 per row at N = 256, and the other widths above 128: the lane-group panel kernel (round 5)
    (12288, 16384, 256, 0.02, "spmm_group_panel_kernel", 3), (32768, 24576, 192, 0.005, "spmm_group_panel_kernel", 3),
    (32768, 16384, 384, 0.005, "spmm_group_panel_kernel", 4), (16384, 16384, 640, 0.02, "spmm_group_panel_kernel", 6),
    (2304, 59904, 384, 0.014, "spmm_group_panel_kernel", 3),     # few rows, but B far beyond the L2s: at most three passes
    # N ≤ 128: two panels, for throughput-bound launches only (≥ 96 Ki rows)
    (131072, 28416, 128, 0.004, "spmm_group_panel_kernel", 2),
    # too few rows, or too few gathers per row of B, for passes to pay: one pass
    (1536, 9728, 256, 0.05, "spmm_wave_row_kernel", 1), (3000, 16384, 256, 0.004, "spmm_wave_row_kernel", 1),
    (4096, 13056, 256, 0.03, "spmm_group_kernel", 1)])
def test_spmm_auto_takes_l2_panels_for_mid_size_b(cmm, dev, oracle_mod, M, K, N, density, kernel, panels):
    """B beyond the L2s (> 6 MiB) but far from the Infinity-Cache regime: AUTO cuts K into panels of about
    4 – 6 MiB (one launch per panel, C carried) — still the CSR-order chain for every row, rows whose columns do not
    ascend included.  Which shapes get panels was re-fitted in round 5 on tools/plan_grid.py (profiles/r05_plan_grid.log):
    ≥ 24 gathers per row of B, enough rows for a pass to be throughput-bound, enough entries per row and pass."""
    g = np.random.Generator(np.random.PCG64(M + K))
    rowptr, col, val = oracle_mod.make_csr(M, K, density, seed=N)
    col, val = col.copy(), val.copy() - 0.5
    for r in (0, 9, M // 2, M - 1):
        s0, e0 = rowptr[r], rowptr[r + 1]
        perm = g.permutation(e0 - s0)
        col[s0:e0], val[s0:e0] = col[s0:e0][perm], val[s0:e0][perm]
    B = g.random((K, N), dtype=np.float32) - 0.5
    d_B = t(B, dev)
    C = torch.full((M, N), float("nan"), device=dev)
    variant, name, launches, splits = cmm.spmm_plan(len(val), M, K, d_B, C)
    assert name == kernel and (launches == panels or panels == 1), (variant, name, launches)
    cmm.naive_spmm(t(val, dev), t(col, dev), t(rowptr, dev), len(val), M, K, d_B, C)
    assert np.array_equal(C.cpu().numpy(), oracle_mod.spmm_csr(rowptr, col, val, M, K, B))


def test_spmm_edge_cases(cmm, dev, oracle_mod):
    # nnz = 0, empty rows at both ends, a row much longer than a wave, inf/nan must not leak from unused B rows
    M, K, N = 9, 70, 256
    A = np.zeros((M, K), np.float32)
    A[2, :] = np.random.Generator(np.random.PCG64(0)).random(K, dtype=np.float32)
    A[2, 5] = 0
    A[6, 3] = 2.0
    rowptr, col, val = oracle_mod.dense_to_csr(A)
    B = np.random.Generator(np.random.PCG64(1)).random((K, N), dtype=np.float32)
    B[5, :] = np.inf   # column 5 is referenced by no nonzero
    B[0, 0] = np.nan   # row 0 is referenced only by row 2
    got = run_spmm(cmm, dev, rowptr.reshape(-1), col, val, M, K, B)
    expect = oracle_mod.spmm_csr(rowptr.reshape(-1), col, val, M, K, B)
    assert np.array_equal(got, expect, equal_nan=True)
    assert np.all(got[[0, 1, 3, 4, 5, 7, 8]] == 0) and np.isfinite(got[6]).all() and np.isnan(got[2, 0])
    z = run_spmm(cmm, dev, np.zeros(M + 1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32), M, K, B)
    assert np.all(z == 0)
    # strided B (a column slice of a wider matrix) is honoured through ldb, not misread
    Bw = torch.rand(K, 2 * N, device=dev)
    C = torch.empty(M, N, device=dev)
    cmm.naive_spmm(t(val, dev), t(col, dev), t(rowptr.reshape(-1), dev), len(val), M, K, Bw[:, N:], C)
    assert np.array_equal(C.cpu().numpy(),
                          oracle_mod.spmm_csr(rowptr.reshape(-1), col, val, M, K, Bw[:, N:].cpu().numpy()))
    # a transposed view is copied, not read as if contiguous (reference defect 1)
    Bt = torch.rand(N, K, device=dev)
    cmm.naive_spmm(t(val, dev), t(col, dev), t(rowptr.reshape(-1), dev), len(val), M, K, Bt.t(), C)
    assert np.array_equal(C.cpu().numpy(),
                          oracle_mod.spmm_csr(rowptr.reshape(-1), col, val, M, K, Bt.t().contiguous().cpu().numpy()))


def test_spmm_argument_errors(cmm, dev):
    v, ci, rp = torch.rand(3, device=dev), torch.zeros(3, dtype=torch.int32, device=dev), \
        torch.tensor([0, 3], dtype=torch.int32, device=dev)
    B, C = torch.rand(4, 8, device=dev), torch.empty(1, 8, device=dev)
    with pytest.raises(RuntimeError, match="rows"):
        cmm.naive_spmm(v, ci, rp, 3, 1, 5, B, C)            # A_cols != B rows
    with pytest.raises(RuntimeError, match="int32"):
        cmm.naive_spmm(v, ci.long(), rp, 3, 1, 4, B, C)     # int64 indices (reference: data_ptr<int> dtype check)
    with pytest.raises(RuntimeError, match="float32"):
        cmm.naive_spmm(v.double(), ci, rp, 3, 1, 4, B, C)
    with pytest.raises(RuntimeError, match="C must be"):
        cmm.naive_spmm(v, ci, rp, 3, 1, 4, B, torch.empty(2, 8, device=dev))
    with pytest.raises(RuntimeError, match="device"):
        cmm.naive_spmm(v, ci, rp, 3, 1, 4, B.cpu(), C)


def test_spmm_config_c2_full_output_bit_exact(cmm, dev, oracle_mod):
    """BASELINE.json configs[1]: 64k×64k CSR at 0.1 % × 64k×128, pinned generator, full output."""
    M = K = 65536
    N = 128
    rowptr, col, val = oracle_mod.make_csr(M, K, 1e-3, seed=0)
    B = np.random.Generator(np.random.PCG64(1)).random((K, N), dtype=np.float32)
    got = run_spmm(cmm, dev, rowptr, col, val, M, K, B)
    assert np.array_equal(got, oracle_mod.spmm_csr_omp(rowptr, col, val, M, K, B))
    # last link of the chain GPU -> oracle -> reference expression, at full size
    assert_matches_reference_expression(got, torch_cpu_csr_matmul(rowptr, col, val, M, K, B))


def test_spmm_config_c3_full_size(cmm, dev, oracle_mod):
    """BASELINE.json configs[2]: 1M×1M CSR at 0.01 % × 1M×256 on one GPU.  Full output against
    the oracle (bit-exact) plus size-independent properties: row-shard equivalence and linearity."""
    M = K = 1 << 20
    N = 256
    rowptr, col, val = oracle_mod.make_csr(M, K, 1e-4, seed=0)
    assert rowptr[-1] == len(val) and 1.09e8 < len(val) < 1.11e8
    B = np.random.Generator(np.random.PCG64(1)).random((K, N), dtype=np.float32)
    d_rp, d_col, d_val, d_B = (t(x, dev) for x in (rowptr, col, val, B))
    C = torch.empty(M, N, device=dev)
    cmm.naive_spmm(d_val, d_col, d_rp, len(val), M, K, d_B, C)
    got = C.cpu().numpy()
    expect = oracle_mod.spmm_csr_omp(rowptr, col, val, M, K, B)
    assert np.array_equal(got, expect)
    del expect
    # the reference's own CPU expression on the whole 1M x 256 output (rows average 105 non-zeros), at its tests'
    # tolerance: closes GPU -> oracle -> torch at the size the metric is quoted on
    assert_matches_reference_expression(got, torch_cpu_csr_matmul(rowptr, col, val, M, K, B))
    # a row shard computed alone gives the same bits as the same rows of the full product
    r0, r1 = 300_000, 431_072
    rp_s = (d_rp[r0:r1 + 1] - d_rp[r0]).contiguous()
    p0, p1 = int(rowptr[r0]), int(rowptr[r1])
    Cs = torch.empty(r1 - r0, N, device=dev)
    cmm.naive_spmm(d_val[p0:p1], d_col[p0:p1], rp_s, p1 - p0, r1 - r0, K, d_B, Cs)
    assert torch.equal(Cs, C[r0:r1])
    # linearity in B: A·(2B) == 2·(A·B) exactly (power-of-two scaling commutes with every rounding)
    C2 = torch.empty(M, N, device=dev)
    cmm.naive_spmm(d_val, d_col, d_rp, len(val), M, K, d_B * 2, C2)
    assert torch.equal(C2, C * 2)


def test_batched_spmm_one_launch(cmm, dev, golden, oracle_mod):
    c = golden.case("batched/bert")
    a, b = c["a"].reshape(-1, 16, 16), c["b"].reshape(-1, 16, 8)
    vals, cols, offs = cmm.dense_to_csr(t(a, dev))
    o_rp, o_col, o_val = oracle_mod.dense_to_csr(a)
    assert np.array_equal(offs.cpu().numpy(), o_rp) and np.array_equal(cols.cpu().numpy(), o_col)
    assert np.array_equal(vals.cpu().numpy(), o_val)
    C = torch.empty(a.shape[0], 16, 8, device=dev)
    cmm.naive_spmm_batched(vals, cols, offs, vals.numel(), a.shape[0], 16, 16, t(b, dev), C)
    expect = oracle_mod.spmm_csr_batched(o_rp, o_col, o_val, a.shape[0], 16, 16, b)
    assert np.array_equal(C.cpu().numpy(), expect)
    assert np.allclose(C.cpu().numpy().reshape(c["c"].shape), c["c"], rtol=RTOL, atol=ATOL)
    # one B shared by every item
    cmm.naive_spmm_batched(vals, cols, offs, vals.numel(), a.shape[0], 16, 16, t(b[0], dev), C)
    assert np.array_equal(C.cpu().numpy(), oracle_mod.spmm_csr_batched(o_rp, o_col, o_val, a.shape[0], 16, 16, b[0]))


def test_dummy_kernel_and_streams(cmm, dev, oracle_mod, capfd):
    cmm.dummy_kernel()
    assert "0..4095 ok" in capfd.readouterr().out
    # work is enqueued on torch's CURRENT stream (the reference uses the legacy default stream)
    M, K, N = 2000, 3000, 256
    rowptr, col, val = oracle_mod.make_csr(M, K, 0.01, seed=9)
    B = np.random.Generator(np.random.PCG64(9)).random((K, N), dtype=np.float32)
    side = torch.cuda.Stream()
    d = [t(x, dev) for x in (val, col, rowptr, B)]
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        C = torch.empty(M, N, device=dev)
        cmm.naive_spmm(d[0], d[1], d[2], len(val), M, K, d[3], C)
        side.synchronize()
        got = C.cpu().numpy()
    assert np.array_equal(got, oracle_mod.spmm_csr(rowptr, col, val, M, K, B))


def test_spmm_addresses_beyond_2_31_elements(cmm, dev, oracle_mod):
    """B with more than 2^31 elements (K·N = 2.36e9): row offsets need 64-bit arithmetic
    (the reference multiplies `int` indices, src/naive_sparse_mm.cu:32-42,86)."""
    K, N, M = 2_300_000, 1024, 300
    g = torch.Generator(device=dev).manual_seed(5)
    B = torch.rand(K, N, device=dev, generator=g)
    rng = np.random.Generator(np.random.PCG64(5))
    rows = []
    for r in range(M):
        c = np.unique(rng.integers(0, K, size=40))
        c[-1] = K - 1 - r           # make sure the far end of B is touched
        rows.append(np.unique(c))
    col = np.concatenate(rows).astype(np.int32)
    rowptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    val = rng.random(len(col), dtype=np.float32)
    C = torch.empty(M, N, device=dev)
    cmm.naive_spmm(t(val, dev), t(col, dev), t(rowptr, dev), len(val), M, K, B, C)
    # oracle on the compacted problem: only the referenced rows of B travel to the host
    used, inv = np.unique(col, return_inverse=True)
    Bsmall = B[torch.from_numpy(used.astype(np.int64)).to(dev)].cpu().numpy()
    expect = oracle_mod.spmm_csr(rowptr, inv.astype(np.int32), val, M, len(used), Bsmall)
    assert np.array_equal(C.cpu().numpy(), expect)


@pytest.mark.parametrize("N", [256, 512, 100, 64, 7])
def test_fused_bias_epilogues_bit_exact(cmm, dev, oracle_mod, N):
    """C = A·B + bias and C = op(A)·op(B) + bias: bias added once, after the accumulation chain."""
    M, K = 130, 200
    rowptr, col, val = oracle_mod.make_csr(M, K, 0.1, seed=N)
    g = np.random.Generator(np.random.PCG64(N))
    B, bias = g.random((K, N), dtype=np.float32), g.random(N, dtype=np.float32)
    C = torch.empty(M, N, device=dev)
    cmm.naive_spmm_bias(t(val, dev), t(col, dev), t(rowptr, dev), len(val), M, K, t(B, dev), t(bias, dev), C)
    assert np.array_equal(C.cpu().numpy(), oracle_mod.spmm_csr(rowptr, col, val, M, K, B) + bias[None, :])
    a, w = g.random((M, K), dtype=np.float32), g.random((N, K), dtype=np.float32)
    cmm.cublas_mmul_bias(t(a, dev), t(w, dev), t(bias, dev), C, False, True)
    assert np.array_equal(C.cpu().numpy(), oracle_mod.gemm(a, w, False, True) + bias[None, :])


def test_fused_bias_on_the_panel_path(capi, dev, oracle_mod):
    """The two-panel large-B path adds the bias in its LAST pass only (forced through the variant id)."""
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    M, K, N = 700, 900, 256
    rowptr, col, val = oracle_mod.make_csr(M, K, 0.05, seed=2)
    g = np.random.Generator(np.random.PCG64(2))
    B = g.random((K, N), dtype=np.float32)
    d = [t(x, dev) for x in (rowptr, col, val, B)]
    expect = oracle_mod.spmm_csr(rowptr, col, val, M, K, B)
    for variant in (7, 8, 12):  # 2, 3 and 8 panels
        C = torch.full((M, N), float("nan"), device=dev)
        assert capi.mi_spmm_csr_f32_variant(variant, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), len(val), M,
                                            K, N, d[3].data_ptr(), N, C.data_ptr(), N,
                                            torch.cuda.current_stream().cuda_stream) == 0
        assert np.array_equal(C.cpu().numpy(), expect)
    # unsorted columns inside a row (legal CSR): every pass detects such rows on the col entries it
    # scans anyway, the first pass sums them in plain CSR order and the later passes leave them alone,
    # so the panel plans equal the CSR-order oracle bit for bit here too.  Mixed: most rows shuffled,
    # some left sorted, one with only its last two entries swapped, duplicates of a column.
    g = np.random.Generator(np.random.PCG64(9))
    colp, valp = col.copy(), val.copy()
    for r in range(M):
        s0, e0 = rowptr[r], rowptr[r + 1]
        if r % 3 != 0 and e0 - s0 > 1:
            perm = g.permutation(e0 - s0)
            colp[s0:e0], valp[s0:e0] = col[s0:e0][perm], val[s0:e0][perm]
    s0, e0 = rowptr[300], rowptr[301]
    colp[s0:e0], valp[s0:e0] = col[s0:e0], val[s0:e0]
    colp[[e0 - 2, e0 - 1]] = colp[[e0 - 1, e0 - 2]]
    colp[rowptr[3] + 1] = colp[rowptr[3]]  # a duplicate column in a sorted row
    expect_p = oracle_mod.spmm_csr(rowptr, colp, valp, M, K, B)
    for variant in (7, 8, 12, 15):  # 2, 3, 8 panels; 15 = column tiles x panels (one 256-column tile here)
        C = torch.full((M, N), float("nan"), device=dev)
        assert capi.mi_spmm_csr_f32_variant(variant, d[0].data_ptr(), t(colp, dev).data_ptr(), t(valp, dev).data_ptr(),
                                            len(val), M, K, N, d[3].data_ptr(), N, C.data_ptr(), N,
                                            torch.cuda.current_stream().cuda_stream) == 0
        torch.cuda.synchronize()
        assert np.array_equal(C.cpu().numpy(), expect_p), variant
    # two column tiles x panels, unsorted rows
    B2 = g.random((K, 512), dtype=np.float32)
    C = torch.full((M, 512), float("nan"), device=dev)
    assert capi.mi_spmm_csr_f32_variant(15, d[0].data_ptr(), t(colp, dev).data_ptr(), t(valp, dev).data_ptr(), len(val),
                                        M, K, 512, t(B2, dev).data_ptr(), 512, C.data_ptr(), 512,
                                        torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    assert np.array_equal(C.cpu().numpy(), oracle_mod.spmm_csr(rowptr, colp, valp, M, K, B2))


def test_panel_plans_detect_a_descent_across_the_64_entry_chunk_boundary(capi, dev, oracle_mod):
    """The panel kernels scan a row 64 entries at a time; the only out-of-order pair of a row may straddle
    two chunks (entries 63|64 or 127|128), or sit in the last, partial chunk: each must still send the row
    to the plain CSR-order chain.  Rows of 130–200 entries, exactly one descent each, at those places."""
    K, N = 5000, 256
    g = np.random.Generator(np.random.PCG64(64))
    spots = [63, 127, 0, 62, 64, 128, 129, 191]          # index i: entry i+1 < entry i
    cols, M = [], 0
    split = (K + 1) // 2                                  # boundary of the two-panel plan
    for spot in spots + [None, None]:                     # two fully sorted rows as well
        n = int(g.integers(max(131, (spot or 0) + 3), 200))
        below = (spot + 1) if spot is not None else n // 2   # entries left of the panel boundary
        c = np.concatenate([np.sort(g.choice(split, size=below, replace=False)),
                            split + np.sort(g.choice(K - split, size=n - below, replace=False))]).astype(np.int32)
        if spot is not None:
            c[spot], c[spot + 1] = c[spot + 1], c[spot]   # the single descent straddles the panel boundary
        cols.append(c)
        M += 1
    rowptr = np.concatenate([[0], np.cumsum([len(c) for c in cols])]).astype(np.int32)
    col = np.concatenate(cols)
    val = g.random(len(col), dtype=np.float32) - 0.5
    B = g.random((K, N), dtype=np.float32) - 0.5
    want = oracle_mod.spmm_csr(rowptr, col, val, M, K, B)
    d = [t(x, dev) for x in (rowptr, col, val, B)]
    for variant in (7, 8, 12):
        C = torch.full((M, N), float("nan"), device=dev)
        assert capi.mi_spmm_csr_f32_variant(variant, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), len(val), M, K, N,
                                            d[3].data_ptr(), N, C.data_ptr(), N, torch.cuda.current_stream().cuda_stream) == 0
        torch.cuda.synchronize()
        assert np.array_equal(C.cpu().numpy(), want), variant
    # sanity of the construction: consumed panel by panel WITHOUT the check, these rows would get other bits
    panel_order = oracle_mod.spmm_csr(rowptr, *_panel_sorted(rowptr, col, val, split), M, K, B)
    differing = [r for r in range(len(spots)) if not np.array_equal(panel_order[r], want[r])]
    assert len(differing) >= len(spots) - 1 and np.array_equal(panel_order[len(spots):], want[len(spots):])


def test_long_row_rule_pinned_by_the_caller(cmm, dev, oracle_mod):
    """custom_mm.naive_spmm_ex: under a SLAB plan (which by itself never splits) mode 1 sums the rows
    beyond 8192 non-zeros in the split order, modes 0 / -1 keep the CSR-order chain; rows up to the
    threshold are the same bits in every mode."""
    M, K, N = 4096, 12000, 1024
    hubs = [3, 2500, M - 1]
    rowptr, col, val = _moderately_dense_with_hub_rows(M, K, 0.5, hubs, 31)
    B = np.random.Generator(np.random.PCG64(32)).random((K, N), dtype=np.float32)
    d = [t(x, dev) for x in (val, col, rowptr)]
    d_B = t(B, dev)
    C = torch.empty(M, N, device=dev)
    variant, name, launches, splits = cmm.spmm_plan(len(val), M, K, d_B, C)
    assert name == "spmm_slab_kernel" and launches == 1 and splits is False
    assert cmm.long_row_threshold() == 8192
    sample = hubs + [0, 4, 1000, 4000]
    sp = _sub_csr(rowptr, col, val, sample)
    chain = oracle_mod.spmm_csr(*sp, len(sample), K, B)
    split = oracle_mod.spmm_csr_long(*sp, len(sample), K, B)
    assert not np.array_equal(chain[:3], split[:3]) and np.array_equal(chain[3:], split[3:])
    idx = torch.tensor(sample, device=dev)
    for mode, want in ((1, split), (0, chain), (-1, chain)):
        C.fill_(float("nan"))
        cmm.naive_spmm_ex(*d, len(val), M, K, d_B, C, mode)
        assert np.array_equal(C[idx].cpu().numpy(), want), mode
    full_auto = C.clone()
    cmm.naive_spmm(*d, len(val), M, K, d_B, C)
    assert torch.equal(C, full_auto)
    with pytest.raises(ValueError):
        cmm.naive_spmm_ex(*d, len(val), M, K, d_B, C, 2)


@pytest.mark.parametrize("shape_a,shape_b,density", [
    ((37, 29), (29, 64), 0.3), ((130, 257), (257, 256), 0.2), ((65, 300), (300, 128), 0.1), ((9, 70), (70, 4), 0.5),
    ((70, 1000), (1000, 100), 0.05), ((300, 64), (64, 36), 0.6),
    ((6, 512, 512), (6, 512, 64), 0.1),      # BERT-shaped: B[item] = 128 KiB → staged in LDS
    ((3, 100, 96), (96, 128), 0.2),          # one B shared by every item, LDS-staged
    ((2, 3, 64, 200), (2, 3, 200, 256), 1.0), ((4, 64, 0), (4, 0, 8), 0.5), ((1, 1), (1, 4), 1.0),
    ((90, 300), (300, 512), 0.2), ((3, 70, 128), (3, 128, 300), 0.3), ((33, 65), (65, 1024), 0.5),   # column tiles of 256
])
def test_fused_dense_skip_is_bit_identical_to_the_csr_route(cmm, dev, oracle_mod, shape_a, shape_b, density):
    g = np.random.Generator(np.random.PCG64(sum(shape_a) + sum(shape_b)))
    a = (g.random(shape_a, dtype=np.float32) * (g.random(shape_a) < density)).astype(np.float32)
    if a.size > 3:
        a.flat[1] = -0.0
    b = g.random(shape_b, dtype=np.float32)
    M, K, N = shape_a[-2], shape_a[-1], shape_b[-1]
    batch = int(np.prod(shape_a[:-2])) if len(shape_a) > 2 else 1
    C = torch.full(tuple(shape_a[:-1]) + (N,), float("nan"), device=dev)
    assert cmm.naive_spmm_dense(t(a, dev), t(b, dev), C) is True
    rp, col, val = oracle_mod.dense_to_csr(a)
    bb = b if b.ndim == 2 else b.reshape(batch, K, N)
    expect = oracle_mod.spmm_csr_batched(rp, col, val, batch, M, K, bb).reshape(C.shape)
    assert np.array_equal(C.cpu().numpy(), expect)
    # widths the fused kernel does not cover are declined, not mis-computed
    for n_bad in (7, 1028):
        C2 = torch.empty(tuple(shape_a[:-1]) + (n_bad,), device=dev)
        assert cmm.naive_spmm_dense(t(a, dev), t(g.random(shape_b[:-1] + (n_bad,), dtype=np.float32), dev), C2) is False


def test_entry_points_are_graph_capturable(cmm, dev, oracle_mod):
    """The C-ABI launches neither allocate nor synchronise, so a hipGraph can capture them
    (`custom_mm.naive_spmm`, `cublas_bmm`, the fused dense-input product) and replay on new data."""
    M, K, N = 3000, 2000, 256
    rowptr, col, val = oracle_mod.make_csr(M, K, 0.01, seed=12)
    d_rp, d_col, d_val = t(rowptr, dev), t(col, dev), t(val, dev)
    B = torch.zeros(K, N, device=dev)
    C = torch.empty(M, N, device=dev)
    q = torch.zeros(3, 2, 96, 64, device=dev)
    kk = torch.zeros(3, 2, 96, 64, device=dev)
    S = torch.empty(3, 2, 96, 96, device=dev)
    P = torch.empty(3, 2, 96, 64, device=dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):  # warm-up outside capture
        cmm.naive_spmm(d_val, d_col, d_rp, len(val), M, K, B, C)
        cmm.cublas_bmm(q, kk, S, 4, False, True)
        cmm.naive_spmm_dense(S, kk, P)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        cmm.naive_spmm(d_val, d_col, d_rp, len(val), M, K, B, C)
        cmm.cublas_bmm(q, kk, S, 4, False, True)
        assert cmm.naive_spmm_dense(S, kk, P) is True
    g = np.random.Generator(np.random.PCG64(12))
    for _ in range(2):  # replay on fresh contents of the same buffers
        Bh = g.random((K, N), dtype=np.float32)
        qh, kh = g.random(q.shape, dtype=np.float32), g.random(kk.shape, dtype=np.float32)
        B.copy_(torch.from_numpy(Bh))
        q.copy_(torch.from_numpy(qh))
        kk.copy_(torch.from_numpy(kh))
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(C.cpu().numpy(), oracle_mod.spmm_csr(rowptr, col, val, M, K, Bh))
        s_exp = oracle_mod.gemm(qh, kh, False, True)
        assert np.array_equal(S.cpu().numpy(), s_exp)
        rp2, c2, v2 = oracle_mod.dense_to_csr(s_exp)
        assert np.array_equal(P.cpu().numpy().reshape(6, 96, 64),
                              oracle_mod.spmm_csr_batched(rp2, c2, v2, 6, 96, 96, kh.reshape(6, 96, 64)))


def test_wide_n_column_tiled_launch_is_bit_exact(capi, cmm, dev, oracle_mod):
    """Wide N with a B that fits neither one L2 nor 8 MiB: AUTO takes the XCD-aware column-tiled launch
    (variant 14); every output element still sees its row's non-zeros in CSR order."""
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    capi.mi_spmm_csr_f32_plan.argtypes = [i64, i32, i32, i32, vp, i64, vp, i64]
    for (M, K, N) in [(2048, 2048, 2048), (2100, 1500, 2560), (4096, 3000, 1024)]:
        rowptr, col, val = oracle_mod.make_csr(M, K, 0.02, seed=N)
        B = np.random.Generator(np.random.PCG64(N)).random((K, N), dtype=np.float32)
        d_B = t(B, dev)
        C = torch.full((M, N), float("nan"), device=dev)
        assert capi.mi_spmm_csr_f32_plan(len(val), M, K, N, d_B.data_ptr(), N, C.data_ptr(), N) == 14
        cmm.naive_spmm(t(val, dev), t(col, dev), t(rowptr, dev), len(val), M, K, d_B, C)
        assert np.array_equal(C.cpu().numpy(), oracle_mod.spmm_csr_omp(rowptr, col, val, M, K, B))


@pytest.mark.parametrize("N", [256, 100, 64, 1024, 7])
def test_skewed_rows_long_row_kernel(cmm, dev, oracle_mod, N):
    """Rows far longer than the rest (one fully dense, some just over / at the 8192 threshold): the
    custom_mm path hands them to the 16-wave long-row kernel; result bit-identical to the oracle's
    statement of that order, and equal to torch's product at the reference tolerance."""
    M, K = 301, 30000
    g = np.random.Generator(np.random.PCG64(N))
    lens = g.integers(0, 200, size=M)
    lens[5], lens[17], lens[18], lens[150], lens[300] = K, 8193, 8192, 20011, 9000
    cols = [np.sort(g.choice(K, size=int(n), replace=False)).astype(np.int32) for n in lens]
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    col = np.concatenate(cols)
    val = g.random(len(col), dtype=np.float32)
    B = g.random((K, N), dtype=np.float32)
    for op in ("naive_spmm", "cusparse_mmul"):
        got = run_spmm(cmm, dev, rowptr, col, val, M, K, B, op)
        assert np.array_equal(got, oracle_mod.spmm_csr_long(rowptr, col, val, M, K, B)), op
    A = torch.sparse_csr_tensor(torch.from_numpy(rowptr.astype(np.int64)), torch.from_numpy(col.astype(np.int64)),
                                torch.from_numpy(val), (M, K))
    assert np.allclose((A @ torch.from_numpy(B)).numpy(), got, rtol=RTOL, atol=ATOL)
    # with a bias, through the fused epilogue of the long-row kernel as well
    bias = g.random(N, dtype=np.float32)
    C = torch.empty(M, N, device=dev)
    cmm.naive_spmm_bias(t(val, dev), t(col, dev), t(rowptr, dev), len(val), M, K, t(B, dev), t(bias, dev), C)
    assert np.array_equal(C.cpu().numpy(), oracle_mod.spmm_csr_long(rowptr, col, val, M, K, B) + bias[None, :])


@pytest.mark.parametrize("N", [64, 260, 512, 100])
def test_many_long_rows_take_the_bulk_form_and_few_the_staged_one(cmm, dev, oracle_mod, N):
    """How the listed rows are summed is decided on the device from their number (csrc/spmm_heavy.hip): a few — one 8-wave
    workgroup per group of chains and 64 columns, the chains staged through LDS; ≥ 128 units — one wave per chain over whole rows
    of B.  The arithmetic is the oracle's statement of the long-row order either way: 150 rows beyond the threshold (two of them
    split over S = 2 and S = 3 groups) beside short rows, then the same rows thinned to 3 long ones; with and without the bias;
    the scheduled product (a prepared list, summed in the heavy rows' launch) gives the same bits."""
    M, K = 400, 100_000
    g = np.random.Generator(np.random.PCG64(N + 7))
    lens = g.integers(0, 120, size=M)
    hubs = g.choice(M, size=150, replace=False)
    lens[hubs] = g.integers(8193, 8600, size=150)
    lens[hubs[0]], lens[hubs[1]], lens[hubs[2]] = 65536, 100_000, 8193
    for keep in (150, 3):
        l2 = lens.copy()
        l2[hubs[keep:]] = 50
        rowptr, col, val = _random_rows_csr(M, K, l2, seed=N + keep)
        B = g.random((K, N), dtype=np.float32) - 0.5
        bias = g.random(N, dtype=np.float32)
        want = oracle_mod.spmm_csr_long(rowptr, col, val, M, K, B)
        got = run_spmm(cmm, dev, rowptr, col, val, M, K, B, "naive_spmm")
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (N, keep)
        d_rp, d_col, d_val, d_B = t(rowptr, dev), t(col, dev), t(val, dev), t(B, dev)
        C = torch.full((M, N), float("nan"), device=dev)
        cmm.naive_spmm_bias(d_val, d_col, d_rp, len(val), M, K, d_B, t(bias, dev), C)
        assert np.array_equal(C.cpu().numpy().view(np.uint32), (want + bias[None, :]).view(np.uint32)), (N, keep)
        sched = cmm.spmm_schedule(d_rp, len(val), M, N)
        assert sched.info()["long_rows"] == keep
        sched.set_heavy(64, True)   # the short rows' longer half becomes heavy slots: list and slots share a launch
        C2 = torch.full((M, N), float("nan"), device=dev)
        cmm.naive_spmm_scheduled(sched, d_val, d_col, d_rp, len(val), M, K, d_B, C2)
        assert np.array_equal(C2.cpu().numpy().view(np.uint32), want.view(np.uint32)), (N, keep)


@pytest.mark.parametrize("N", [64, 256, 30])
def test_hub_rows_split_over_workgroups(cmm, dev, oracle_mod, N):
    """Rows of ≥ 65536 non-zeros are summed by S = len/32768 workgroups through partial rows in the
    workspace (S = 2, 3, 9 here, beside S = 1 long rows and ordinary ones); bit-identical to the
    oracle's statement of that order, with and without the fused bias."""
    M, K = 40, 300000
    g = np.random.Generator(np.random.PCG64(N + 1))
    lens = g.integers(0, 300, size=M)
    lens[0], lens[7], lens[8], lens[20], lens[39] = 65536, 65535, 100000, K, 9000
    cols = [np.sort(g.choice(K, size=int(n), replace=False)).astype(np.int32) for n in lens]
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    col = np.concatenate(cols)
    val = (g.random(len(col), dtype=np.float32) - 0.5)
    B = g.random((K, N), dtype=np.float32)
    want = oracle_mod.spmm_csr_long(rowptr, col, val, M, K, B)
    got = run_spmm(cmm, dev, rowptr, col, val, M, K, B, "naive_spmm")
    assert np.array_equal(got, want)
    ref = np.zeros((M, N), dtype=np.float64)
    for r in (0, 8, 20):
        sl = slice(rowptr[r], rowptr[r + 1])
        ref[r] = val[sl].astype(np.float64) @ B[col[sl]].astype(np.float64)
        assert np.allclose(got[r], ref[r], rtol=1e-4, atol=1e-2)
    bias = g.random(N, dtype=np.float32)
    C = torch.full((M, N), float("nan"), device=dev)
    cmm.naive_spmm_bias(t(val, dev), t(col, dev), t(rowptr, dev), len(val), M, K, t(B, dev), t(bias, dev), C)
    assert np.array_equal(C.cpu().numpy(), want + bias[None, :])


@pytest.mark.parametrize("M,K,N,density,bias", [
    (300, 500, 256, 0.2, False), (129, 64, 300, 0.5, True), (1000, 1000, 100, 0.08, False),
    (257, 70, 512, 1.0, True), (64, 2000, 260, 0.03, False), (5, 10, 4, 0.5, False),
])
def test_spmm_slab_kernel_shapes(capi, cmm, dev, oracle_mod, M, K, N, density, bias):
    """The LDS-slab kernel (variant 17) at ragged shapes — row blocks, k-slabs and column tiles all
    partial, rows from empty to fully dense, windows of more than 64 entries — bit-identical to the
    CSR-order oracle; with the fused bias through the bias entry point forced onto the same plan."""
    rowptr, col, val = oracle_mod.make_csr(M, K, density, seed=M + K + N)
    val = val - 0.5  # signed values: cancellations, exact zeros and the (-0.0)·(+0.0) padding product
    B = np.random.Generator(np.random.PCG64(N)).random((K, N), dtype=np.float32) - 0.25
    expect = oracle_mod.spmm_csr_chain(rowptr, col, val, M, K, B)
    d = [t(x, dev) for x in (rowptr, col, val, B)]
    C = torch.full((M, N), float("nan"), device=dev)
    assert capi.mi_spmm_csr_f32_variant(17, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), len(val), M, K, N,
                                        d[3].data_ptr(), N, C.data_ptr(), N,
                                        torch.cuda.current_stream().cuda_stream) == 0
    assert np.array_equal(C.cpu().numpy(), expect)
    if bias:
        # B with inf / nan in rows no non-zero refers to must not leak (the padding slot reads the zero row)
        used = np.zeros(K, bool)
        used[col] = True
        if (~used).any():
            B2 = B.copy()
            B2[~used] = np.inf
            C.fill_(float("nan"))
            assert capi.mi_spmm_csr_f32_variant(17, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), len(val), M, K,
                                                N, t(B2, dev).data_ptr(), N, C.data_ptr(), N,
                                                torch.cuda.current_stream().cuda_stream) == 0
            assert np.array_equal(C.cpu().numpy(), expect)


def test_spmm_slab_kernel_fuzz(capi, dev, oracle_mod):
    """Seeded random shapes / densities through the LDS-slab kernel (variant 17), a fraction of the
    rows shuffled out of column order, duplicate columns allowed — always the CSR-order chain."""
    rng = np.random.Generator(np.random.PCG64(4242))
    for case in range(30):
        M = int(rng.integers(1, 400))
        K = int(rng.integers(1, 700))
        N = 4 * int(rng.integers(1, 160))
        density = float(rng.choice([0.0, 0.01, 0.1, 0.5, 1.0]))
        lens = rng.binomial(K, density, size=M) if density < 1.0 else np.full(M, K)
        if case % 5 == 0:
            lens = (lens * rng.integers(0, 3, size=M)).clip(0, 3 * K)  # rows longer than K: duplicate columns
        cols = []
        for n in lens:
            c = rng.integers(0, K, size=int(n)) if n > K else rng.choice(K, size=int(n), replace=False)
            c = np.sort(c)
            if rng.random() < 0.2:
                c = rng.permutation(c)  # an unsorted row
            cols.append(c.astype(np.int32))
        rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        col = np.concatenate(cols) if len(cols) else np.zeros(0, np.int32)
        val = (rng.random(len(col), dtype=np.float32) - 0.5)
        B = rng.random((K, N), dtype=np.float32) - 0.5
        d = [t(x, dev) for x in (rowptr, col, val, B)]
        C = torch.full((M, N), float("nan"), device=dev)
        st = capi.mi_spmm_csr_f32_variant(17, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), len(val), M, K, N,
                                          d[3].data_ptr(), N, C.data_ptr(), N, torch.cuda.current_stream().cuda_stream)
        assert st == 0, (case, st)
        assert np.array_equal(C.cpu().numpy(), oracle_mod.spmm_csr_chain(rowptr, col, val, M, K, B)), (case, M, K, N, density)


def test_spmm_slab_kernel_unsorted_rows_and_auto_plan(capi, cmm, dev, oracle_mod):
    """AUTO picks the slab plan at moderate density on a large enough problem; rows whose columns do
    not ascend (legal CSR: the reference's COO→CSR keeps input order) are recomputed in CSR order
    inside the kernel, so the result still equals the oracle bit for bit (fused bias on top)."""
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    capi.mi_spmm_csr_f32_plan.argtypes = [i64, i32, i32, i32, vp, i64, vp, i64]
    M, K, N = 4096, 9000, 4096
    g = np.random.Generator(np.random.PCG64(17))
    rowptr, col, val = oracle_mod.make_csr(M, K, 0.08, seed=5)
    col = col.copy()
    val = val.copy()
    # shuffle the entries of some rows (first, last, one crossing a 64-entry window boundary late)
    for r in (0, 77, 1000, M - 1):
        s, e = rowptr[r], rowptr[r + 1]
        perm = g.permutation(e - s)
        col[s:e] = col[s:e][perm]
        val[s:e] = val[s:e][perm]
    r = 2000
    s, e = rowptr[r], rowptr[r + 1]
    col[[e - 2, e - 1]] = col[[e - 1, e - 2]]  # only the last two entries out of order
    B = g.random((K, N), dtype=np.float32)
    d_B = t(B, dev)
    C = torch.full((M, N), float("nan"), device=dev)
    assert capi.mi_spmm_csr_f32_plan(len(val), M, K, N, d_B.data_ptr(), N, C.data_ptr(), N) == 17
    bias = g.random(N, dtype=np.float32)
    cmm.naive_spmm_bias(t(val, dev), t(col, dev), t(rowptr, dev), len(val), M, K, d_B, t(bias, dev), C)
    want = oracle_mod.spmm_csr_omp(rowptr, col, val, M, K, B)
    assert np.array_equal(C.cpu().numpy(), want + bias[None, :])


def test_spmm_slab_plan_keeps_csr_order_for_long_rows(capi, cmm, dev, oracle_mod):
    """Under the slab plan rows of more than 8192 non-zeros are NOT handed to the long-row kernel
    (mi_spmm.h: no split under MI_SPMM_SLAB): custom_mm.naive_spmm equals the plain CSR-order chain."""
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    capi.mi_spmm_csr_f32_plan.argtypes = [i64, i32, i32, i32, vp, i64, vp, i64]
    M, K, N = 4096, 12000, 4096
    g = np.random.Generator(np.random.PCG64(23))
    mask = g.random((M, K), dtype=np.float32) < 0.08
    mask[[3, 2500, M - 1]] = True       # three fully dense rows (12000 non-zeros each)
    mask[7] = False                      # and an empty one
    rows, col = np.nonzero(mask)
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=M))]).astype(np.int32)
    col = col.astype(np.int32)
    val = g.random(len(col), dtype=np.float32)
    B = g.random((K, N), dtype=np.float32)
    d_B = t(B, dev)
    C = torch.full((M, N), float("nan"), device=dev)
    assert capi.mi_spmm_csr_f32_plan(len(val), M, K, N, d_B.data_ptr(), N, C.data_ptr(), N) == 17
    cmm.naive_spmm(t(val, dev), t(col, dev), t(rowptr, dev), len(val), M, K, d_B, C)
    assert np.array_equal(C.cpu().numpy(), oracle_mod.spmm_csr_omp(rowptr, col, val, M, K, B))


def test_spmm_shape_fuzz_against_oracle(cmm, capi, dev, oracle_mod):
    """Seeded random shapes through custom_mm.naive_spmm (AUTO dispatch incl. the column-tiled and
    panel plans, partial last tiles, odd widths) — every one bit-identical to the oracle."""
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    capi.mi_spmm_csr_f32_plan.argtypes = [i64, i32, i32, i32, vp, i64, vp, i64]
    rng = np.random.Generator(np.random.PCG64(2026))
    seen = set()
    cases = [(2100, 1100, 2100, 0.02), (2048, 9000, 2048, 0.02), (2300, 5000, 2304, 0.01), (2048, 1030, 4100, 0.002),
             (3000, 2500, 1028, 0.02)]
    for _ in range(40):
        cases.append((int(rng.integers(1, 600)), int(rng.integers(1, 900)), int(rng.choice([1, 2, 3, 4, 5, 8, 12, 31, 32, 33, 64,
                     96, 100, 128, 192, 255, 256, 257, 260, 384, 512, 516, 640, 1000, 1024, 1028, 2048])),
                      float(rng.choice([0.0, 0.002, 0.02, 0.2]))))
    for (M, K, N, density) in cases:
        rowptr, col, val = oracle_mod.make_csr(M, K, density, seed=M * 7 + N) if density > 0 else \
            (np.zeros(M + 1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32))
        B = rng.random((K, N), dtype=np.float32)
        d_B = t(B, dev)
        C = torch.full((M, N), float("nan"), device=dev)
        seen.add(capi.mi_spmm_csr_f32_plan(len(val), M, K, N, d_B.data_ptr(), N, C.data_ptr(), N))
        cmm.naive_spmm(t(val, dev), t(col, dev), t(rowptr, dev), len(val), M, K, d_B, C)
        assert np.array_equal(C.cpu().numpy(), oracle_mod.spmm_csr_omp(rowptr, col, val, M, K, B)), (M, K, N, density)
    assert {2, 4, 14, 15, 24} <= seen, seen  # wave-row, group vec4 / unaligned quads (odd widths), column-tiled, tiles × panels all exercised


@pytest.mark.gpu
def test_long_rows_listed_by_the_main_kernel_and_zero_header_contract(capi, cmm, dev, oracle_mod):
    """Round 3: the kernels list the rows they skip (no separate scan of rowptr), one follow-up launch sums them,
    combines the split ones (last workgroup of a row) and resets the counters.  MI_LONG_ROWS_AUTO_ZEROED (3): a
    workspace that enters with a zero 16-byte header leaves with one, product after product, and gives the bits of
    the memset-per-call mode (-1) and of the oracle; custom_mm.naive_spmm keeps such a workspace per stream, so
    matrices with and without hub rows can alternate on it.  Reference entry: src/custom_mm.cpp:166-179."""
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    capi.mi_spmm_csr_ex_f32.argtypes = [vp, vp, vp, i64, i32, i32, i32, vp, i64, vp, vp, i64, ctypes.c_int, vp,
                                        ctypes.c_size_t, vp]
    capi.mi_spmm_csr_workspace_bytes.restype = ctypes.c_size_t
    capi.mi_spmm_csr_workspace_bytes.argtypes = [i64, i32]
    g = np.random.Generator(np.random.PCG64(77))
    K = 200000

    def skewed(M, hubs, N):
        lens = g.integers(0, 200, size=M)
        for r, n in hubs:
            lens[r] = n
        cols = [np.sort(g.choice(K, size=int(n), replace=False)).astype(np.int32) for n in lens]
        rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        col = np.concatenate(cols)
        val = g.random(len(col), dtype=np.float32) - 0.5
        B = g.random((K, N), dtype=np.float32)
        return rowptr, col, val, B

    stream = torch.cuda.current_stream().cuda_stream
    for N in (256, 128, 36):  # one wave per row, lane groups, scalar lanes
        M = 64
        with_hubs = skewed(M, [(0, 70000), (5, 9000), (33, 140000), (63, 8193)], N)
        without = skewed(M, [], N)
        nnz_max = max(len(with_hubs[1]), len(without[1]))
        nbytes = capi.mi_spmm_csr_workspace_bytes(nnz_max, N)
        ws = torch.full((nbytes,), 0x5A, dtype=torch.uint8, device=dev)  # garbage beyond the header …
        ws[:16] = 0                                                        # … and the contract's zero header
        for rowptr, col, val, B in (with_hubs, without, with_hubs):
            want = oracle_mod.spmm_csr_long(rowptr, col, val, M, K, B)
            d = [t(x, dev) for x in (rowptr, col, val, B)]
            for mode in (3, -1):
                C = torch.full((M, N), float("nan"), device=dev)
                wsm = ws if mode == 3 else torch.full((nbytes,), 0xA5, dtype=torch.uint8, device=dev)
                assert capi.mi_spmm_csr_ex_f32(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), len(val), M, K, N,
                                               d[3].data_ptr(), N, None, C.data_ptr(), N, mode, wsm.data_ptr(), nbytes,
                                               stream) == 0
                assert np.array_equal(C.cpu().numpy(), want), (N, mode)
                if mode == 3:
                    assert int(wsm[:16].to(torch.int32).sum()) == 0, "the counters are zero again after the product"
            # the reference-named entry on its persistent per-stream workspace
            assert np.array_equal(run_spmm(cmm, dev, rowptr, col, val, M, K, B, "naive_spmm"), want)
            assert np.array_equal(run_spmm(cmm, dev, rowptr, col, val, M, K, B, "cusparse_mmul"), want)
    # a side stream gets a workspace of its own
    side = torch.cuda.Stream()
    rowptr, col, val, B = with_hubs
    want = oracle_mod.spmm_csr_long(rowptr, col, val, 64, K, B)
    with torch.cuda.stream(side):
        got = run_spmm(cmm, dev, rowptr, col, val, 64, K, B, "naive_spmm")
    side.synchronize()
    assert np.array_equal(got, want)


def test_long_row_machinery_fuzz_against_oracle(capi, cmm, dev, oracle_mod):
    """Random skewed matrices — mostly short rows plus 0–4 rows around and far beyond the 8192 threshold (lengths at
    the split points 65535 / 65536, up to a fully dense row), sorted or shuffled columns, widths for every kernel
    family, padded ldb / ldc, the fused bias — through every long-row mode of the C-ABI (`mi_spmm_csr_ex_f32`:
    memset-per-call AUTO, SPLIT, PREPARED after `mi_spmm_long_rows_prepare`, AUTO_ZEROED on one workspace reused
    across all cases) and through `custom_mm.naive_spmm` (reference entry src/custom_mm.cpp:166-179): bit-identical to
    the oracle's statement of the long-row order, padding of C untouched.  MI_FUZZ_CASES / MI_FUZZ_SEED as above."""
    import os
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    capi.mi_spmm_csr_ex_f32.argtypes = [vp, vp, vp, i64, i32, i32, i32, vp, i64, vp, vp, i64, ctypes.c_int, vp,
                                        ctypes.c_size_t, vp]
    capi.mi_spmm_long_rows_prepare.argtypes = [vp, i32, i64, i32, vp, ctypes.c_size_t, vp]
    capi.mi_spmm_csr_workspace_bytes.restype = ctypes.c_size_t
    capi.mi_spmm_csr_workspace_bytes.argtypes = [i64, i32]
    g = np.random.Generator(np.random.PCG64(int(os.environ.get("MI_FUZZ_SEED", "303"))))
    cases = int(os.environ.get("MI_FUZZ_CASES", "24"))
    stream = torch.cuda.current_stream().cuda_stream
    thr = capi.mi_spmm_long_row_threshold()
    assert thr == 8192
    zeroed_bytes = capi.mi_spmm_csr_workspace_bytes(4_000_000, 1024)
    zeroed = torch.full((zeroed_bytes,), 0x5A, dtype=torch.uint8, device=dev)
    zeroed[:16] = 0
    for case in range(cases):
        M = int(g.integers(1, 200))
        K = int(g.choice([20000, 70000, 140000, 300000]))
        N = int(g.choice([4, 8, 30, 36, 64, 100, 128, 192, 256, 260, 512]))
        lens = g.integers(0, 150, size=M)
        hubs = int(g.integers(0, 5))
        for _ in range(hubs):
            kind = int(g.integers(0, 5))
            n = (thr + int(g.integers(-2, 3)), int(g.integers(thr, 4 * thr)), int(g.choice([65535, 65536, 65537, 98304])),
                 int(g.integers(thr, K + 1)), K)[kind]
            lens[int(g.integers(0, M))] = min(n, K)
        while int(lens.sum()) * N > 150_000_000:  # the oracle stays around a second
            lens[int(np.argmax(lens))] //= 2
        shuffled = bool(g.integers(0, 2))
        cols = []
        for n in lens:
            c = g.choice(K, size=int(n), replace=False).astype(np.int32)
            cols.append(c if shuffled else np.sort(c))
        rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        col = np.concatenate(cols) if len(cols) else np.zeros(0, np.int32)
        nnz = len(col)
        val = g.random(nnz, dtype=np.float32) - 0.5
        ldb, ldc = N + int(g.choice([0, 0, 4, 7])), N + int(g.choice([0, 0, 4, 9]))
        Bp = g.random((K, ldb), dtype=np.float32)
        B = np.ascontiguousarray(Bp[:, :N])
        with_bias = bool(g.integers(0, 3) == 0)
        bias = g.random(N, dtype=np.float32) if with_bias else None
        want = oracle_mod.spmm_csr_long(rowptr, col, val, M, K, B)
        if with_bias:
            want = want + bias[None, :]
        d_rp, d_col, d_val, d_B = t(rowptr, dev), t(col, dev), t(val, dev), t(Bp, dev)
        d_bias = t(bias, dev) if with_bias else None
        nbytes = capi.mi_spmm_csr_workspace_bytes(nnz, N)
        assert nbytes <= zeroed_bytes
        what = (case, M, K, N, nnz, sorted(int(x) for x in lens if x > thr - 3), shuffled, ldb, ldc, with_bias)
        # AUTO follows the plan: SLAB / NARROW plans keep their own order — not reachable here (N ≥ 4, K ≥ 20000 at
        # < 1 % density), asserted rather than assumed
        assert capi.mi_spmm_auto_splits_long_rows(nnz, M, K, N, d_B.data_ptr(), ldb, None, ldc) == (1 if nnz > thr else 0), what
        for mode in (-1, 1, 2, 3):
            ws = zeroed if mode == 3 else torch.full((max(nbytes, 16),), 0xA5, dtype=torch.uint8, device=dev)
            if mode == 2:
                assert capi.mi_spmm_long_rows_prepare(d_rp.data_ptr(), M, nnz, N, ws.data_ptr(), ws.numel(), stream) == 0
            C = torch.full((M, ldc), float("nan"), device=dev)
            st = capi.mi_spmm_csr_ex_f32(d_rp.data_ptr(), d_col.data_ptr(), d_val.data_ptr(), nnz, M, K, N, d_B.data_ptr(),
                                         ldb, d_bias.data_ptr() if with_bias else None, C.data_ptr(), ldc, mode,
                                         ws.data_ptr(), ws.numel(), stream)
            assert st == 0, (mode,) + what
            got = C.cpu().numpy()
            assert np.array_equal(got[:, :N], want), (mode,) + what
            assert np.isnan(got[:, N:]).all(), (mode,) + what
            if mode == 3:
                assert int(zeroed[:16].to(torch.int32).sum()) == 0, what
        if not with_bias:
            assert np.array_equal(run_spmm(cmm, dev, rowptr, col, val, M, K, B, "naive_spmm"), want), what


def test_batched_spmm_variants_fuzz_against_oracle(capi, dev, oracle_mod):
    """Random batched products through `mi_spmm_csr_batched_variant_f32` — the LDS-resident-B plan (18), the group
    kernels with float4 / scalar lanes (4, 5: DPP row broadcasts for 8- and 16-lane groups, 32-entry ds_bpermute chunks
    for 32 lanes, readlane for 64) and AUTO (0): random widths (every group width, widths that are not powers of two),
    padded ldb / ldc and item strides, shared B, shuffled rows with duplicates, empty rows and items — each bit-identical
    to the oracle's batched product, padding of C untouched.  MI_FUZZ_CASES / MI_FUZZ_SEED as above."""
    import os
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    capi.mi_spmm_csr_batched_variant_f32.argtypes = [ctypes.c_int, vp, vp, vp, i64, i32, i32, i32, i32, vp, i64, i64, vp, i64, i64, vp]
    stream = torch.cuda.current_stream().cuda_stream
    g = np.random.Generator(np.random.PCG64(int(os.environ.get("MI_FUZZ_SEED", "505"))))
    cases = int(os.environ.get("MI_FUZZ_CASES", "40"))
    took = {}
    for case in range(cases):
        batch = int(g.choice([1, 1, 2, 3, 6]))
        M, K = int(g.integers(1, 500)), int(g.integers(1, 400))
        N = int(g.choice([4, 8, 12, 16, 20, 32, 36, 48, 64, 68, 96, 128, 132, 200, 256, 7, 30, 100]))
        mean = float(g.choice([0.5, 3, 12, 40, 90]))
        lens = g.poisson(mean, size=batch * M)
        lens[g.integers(0, batch * M, size=3)] = 0
        cols = []
        for i, n in enumerate(lens):
            c = g.integers(0, K, size=int(n)).astype(np.int32)
            cols.append(np.sort(c) if i % 4 else c)
        col = np.concatenate(cols) if len(cols) else np.zeros(0, np.int32)
        val = g.random(len(col), dtype=np.float32) - 0.5
        off = np.zeros((batch, M + 1), np.int64)
        off[:, 1:] = np.cumsum(lens).reshape(batch, M)
        off[1:, 0] = off[:-1, M]
        off = off.astype(np.int32)
        share_b = batch > 1 and bool(g.integers(0, 3) == 0)
        pad = int(g.choice([0, 0, 4, 8])) if N % 4 == 0 else int(g.choice([0, 1, 3]))
        ldb, ldc = N + pad, N + (pad if g.integers(0, 2) else 0)
        sb = 0 if share_b else K * ldb + (8 if pad else 0)
        sc = M * ldc + (12 if pad else 0)
        Bbuf = g.random((1 if share_b else batch) * max(sb, K * ldb) + 16, dtype=np.float32) - 0.5
        view = lambda buf, i, stride, rows, cols_, ld: np.lib.stride_tricks.as_strided(buf[i * stride:], shape=(rows, cols_), strides=(ld * 4, 4))
        Bs = np.stack([np.ascontiguousarray(view(Bbuf, 0 if share_b else i, sb, K, N, ldb)) for i in range(batch)])
        want = oracle_mod.spmm_csr_batched(off, col, val, batch, M, K, Bs)
        d_off, d_col, d_val, d_B = t(off, dev), t(col, dev), t(val, dev), t(Bbuf, dev)
        ran = 0
        for variant in (0, 4, 5, 18):
            C = torch.full((batch * sc + 16,), float("nan"), device=dev)
            st = capi.mi_spmm_csr_batched_variant_f32(variant, d_off.data_ptr(), d_col.data_ptr(), d_val.data_ptr(), len(col),
                                                      batch, M, K, N, d_B.data_ptr(), ldb, sb, C.data_ptr(), ldc, sc, stream)
            what = (case, variant, batch, M, K, N, ldb, ldc, sb, sc, len(col))
            if st == -1 and variant in (4, 18):   # not float4-able (N % 4, padding) or B does not fit LDS
                continue
            assert st == 0, what
            ran += 1
            took[variant] = took.get(variant, 0) + 1
            got = C.cpu().numpy()
            written = np.zeros(got.shape, dtype=bool)
            for i in range(batch):
                assert np.array_equal(view(got, i, sc, M, N, ldc), want[i]), what
                written[(i * sc + np.arange(M)[:, None] * ldc + np.arange(N)[None, :]).ravel()] = True
            assert np.isnan(got[~written]).all(), what
        assert ran >= 2
    assert took.get(18, 0) >= cases // 4 and took.get(4, 0) >= cases // 4 and took[0] == took[5] == cases, took


@pytest.mark.parametrize("N", [64, 128, 36, 100, 192, 320, 700, 1024])
def test_group_panel_plans_keep_csr_order_for_any_row(capi, cmm, dev, oracle_mod, N):
    """Round 5: the lane-group kernel in column panels (MI_SPMM_GROUP_PANELS_2/3/4/6/8 = 19 … 23; N ≤ 128 in lane groups, wider N as a whole wave per row with up to four column
    tiles per lane; B beyond the
    Infinity Cache).  A pass takes the entries whose running maximum of the row's columns lies in its panel, which cuts
    every row into contiguous ranges in CSR order — so sorted rows, shuffled rows, rows with duplicate columns, rows
    longer than a chunk, empty rows and a descent exactly at a chunk boundary all give the one-pass chain, bit for bit;
    bias in the last pass only; rows beyond the long-row threshold skipped in every pass and summed by the follow-up
    launch.  Reference: src/naive_sparse_mm.cu:39,116 (any N through one kernel), :60-92 (CSR-order accumulation)."""
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    capi.mi_spmm_csr_ex_variant_f32.argtypes = [ctypes.c_int, vp, vp, vp, i64, i32, i32, i32, vp, i64, vp, vp, i64, ctypes.c_int,
                                                vp, ctypes.c_size_t, vp]
    capi.mi_spmm_csr_workspace_bytes.restype = ctypes.c_size_t
    capi.mi_spmm_csr_workspace_bytes.argtypes = [i64, i32]
    stream = torch.cuda.current_stream().cuda_stream
    g = np.random.Generator(np.random.PCG64(500 + N))
    M, K = 1300, 5000
    lens = g.integers(0, 90, size=M)
    lens[g.integers(0, M, size=20)] = 0
    lens[5], lens[6], lens[7] = 16, 32, 33          # whole chunks of a 16- / 32-lane group, one entry over
    lens[100] = 9000                                  # beyond the long-row threshold (duplicates: K = 5000)
    cols = []
    for r, n in enumerate(lens):
        c = g.integers(0, K, size=int(n))
        cols.append(np.sort(c) if r % 3 else c)      # two thirds sorted, one third in random order (with duplicates)
    cols[6] = np.sort(cols[6])
    cols[6][[15, 16]] = cols[6][[16, 15]]             # a descent exactly at the 16-entry chunk boundary
    col = np.concatenate(cols).astype(np.int32)
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    val = g.random(len(col), dtype=np.float32) - 0.5
    B, bias = g.random((K, N), dtype=np.float32) - 0.5, g.random(N, dtype=np.float32)
    chain = oracle_mod.spmm_csr(rowptr, col, val, M, K, B)
    split = oracle_mod.spmm_csr_long(rowptr, col, val, M, K, B)
    d = [t(x, dev) for x in (rowptr, col, val, B, bias)]
    ws_bytes = capi.mi_spmm_csr_workspace_bytes(len(col), N)
    ws = torch.zeros(ws_bytes + 16, dtype=torch.uint8, device=dev)
    for variant in (19, 20, 21, 22, 23):
        for with_bias in (False, True):
            for rule, want in ((0, chain), (1, split)):     # MI_LONG_ROWS_NONE / _SPLIT
                C = torch.full((M, N), float("nan"), device=dev)
                st = capi.mi_spmm_csr_ex_variant_f32(variant, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), len(col), M, K, N,
                                                     d[3].data_ptr(), N, d[4].data_ptr() if with_bias else None, C.data_ptr(), N,
                                                     rule, ws.data_ptr(), ws_bytes, stream)
                assert st == 0, (variant, with_bias, rule)
                expect = want + bias[None, :] if with_bias else want
                assert np.array_equal(C.cpu().numpy().view(np.int32), expect.view(np.int32)), (variant, with_bias, rule)


@pytest.mark.parametrize("N", [4, 5, 7, 30, 77, 130, 250, 257, 1031])
def test_rows_off_16_byte_boundaries_take_unaligned_quads_bit_exact(capi, cmm, dev, oracle_mod, N):
    """Round 5: N % 4 != 0 (or an offset view of B) used to fall to one float per lane (0.35 – 0.49 of the roofline at 2 M
    rows); MI_SPMM_GROUP_VEC4U = 24 keeps four floats per lane on dword-aligned 16-byte accesses, the row's partial last quad
    shifted back onto its neighbour — the shared columns are computed twice from the same chain.  Bit-exact against the
    oracle for contiguous B and for a column-offset view with a padded leading dimension; AUTO picks it.
    Reference: src/naive_sparse_mm.cu:39,116 (any N through one kernel)."""
    M, K = 700, 900
    rowptr, col, val = oracle_mod.make_csr(M, K, 0.03, seed=N)
    g = np.random.Generator(np.random.PCG64(N))
    Bpad = g.random((K, N + 7), dtype=np.float32) - 0.5
    d_pad = t(Bpad, dev)
    for B_host, d_B in ((np.ascontiguousarray(Bpad[:, 3:3 + N]), d_pad[:, 3:3 + N].contiguous()), (Bpad[:, 3:3 + N], d_pad[:, 3:3 + N])):
        want = oracle_mod.spmm_csr_chain(rowptr, col, val, M, K, np.ascontiguousarray(B_host))
        C = torch.full((M, N), float("nan"), device=dev)
        plan = cmm.spmm_plan(len(val), M, K, d_B, C)
        if N % 4 != 0 or not d_B.is_contiguous():
            assert plan[0] == 24, plan
        cmm.naive_spmm_ex(t(val, dev), t(col, dev), t(rowptr, dev), len(val), M, K, d_B, C, 0)
        assert np.array_equal(C.cpu().numpy().view(np.int32), want.view(np.int32)), (N, d_B.is_contiguous())


@pytest.mark.parametrize("N,plan_id", [(128, 19), (320, 21)])
def test_hbm_regime_panel_plans_at_full_size(capi, cmm, dev, oracle_mod, N, plan_id):
    """Round 5, at the sizes of tools/bench_hbm_regime.py (2 M × 2 M, 100 per row: B = 1 / 2.5 GiB, beyond the Infinity
    Cache): AUTO takes the lane-group panel kernel; the whole output equals the pinned one-pass kernel's bit for bit,
    sampled rows equal the oracle bit for bit (the rows' sub-matrix with columns renumbered in order — the chain is
    unchanged — times the gathered rows of B), and the product is exactly linear in B (×2 is exact in fp32).
    Size-independent properties where the oracle cannot run the whole problem in seconds."""
    M = K = 1 << 21
    g = torch.Generator(device=dev).manual_seed(N)
    keys = torch.unique(torch.randint(0, M * K, (M * 100,), device=dev, generator=g, dtype=torch.int64))
    col = (keys % K).to(torch.int32)
    rowptr = torch.zeros(M + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(torch.bincount(keys // K, minlength=M), 0)
    rowptr = rowptr.to(torch.int32)
    del keys
    nnz = col.numel()
    val = torch.rand(nnz, device=dev, generator=g) - 0.5
    B = torch.rand(K, N, device=dev, generator=g) - 0.5
    C = torch.full((M, N), float("nan"), device=dev)
    assert cmm.spmm_plan(nnz, M, K, B, C)[0] == plan_id
    cmm.naive_spmm(val, col, rowptr, nnz, M, K, B, C)
    # (a) the pinned one-pass lane-group kernel: same bits on the whole output
    C1 = torch.full((M, N), float("nan"), device=dev)
    assert capi.mi_spmm_csr_f32_variant(4, rowptr.data_ptr(), col.data_ptr(), val.data_ptr(), nnz, M, K, N, B.data_ptr(), N,
                                        C1.data_ptr(), N, torch.cuda.current_stream().cuda_stream) == 0
    assert torch.equal(C.view(torch.int32), C1.view(torch.int32))
    # (b) linear in B, exactly
    cmm.naive_spmm(val, col, rowptr, nnz, M, K, B * 2, C1)
    assert torch.equal(C1, C * 2)
    del C1
    # (c) sampled rows against the oracle
    rs = np.unique(np.concatenate([[0, M - 1], np.random.default_rng(N).integers(0, M, 160)]))
    rp = rowptr.cpu().numpy().astype(np.int64)
    segs = [np.arange(rp[r], rp[r + 1]) for r in rs]
    idx = torch.from_numpy(np.concatenate(segs)).to(dev)
    cs, vs = col[idx].cpu().numpy(), val[idx].cpu().numpy()
    sub_rp = np.concatenate([[0], np.cumsum([len(s) for s in segs])]).astype(np.int32)
    uniq, inv = np.unique(cs, return_inverse=True)
    Bs = B[torch.from_numpy(uniq.astype(np.int64)).to(dev)].cpu().numpy()
    want = oracle_mod.spmm_csr(sub_rp, inv.astype(np.int32), vs, len(rs), len(uniq), Bs)
    got = C[torch.from_numpy(rs).to(dev)].cpu().numpy()
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32))


@pytest.mark.parametrize("N,variants", [(256, (7, 9, 12, 19, 21)), (192, (19, 20, 23)), (512, (8, 11, 22)), (128, (19, 21))])
def test_panel_passes_adapt_to_banded_structure_on_the_device(capi, cmm, dev, oracle_mod, N, variants):
    """Round 5: the L2-level panel plans are chosen from the shape alone; with a workspace at hand a probe launch ahead of the
    passes writes, per window of 2048 rows, whether the rows of B it touches span ≤ 0.4 of B (and ≤ 128 MiB), and every
    workgroup of the panel kernels reads those verdicts: on a banded matrix the FIRST pass takes every column and the bias, the
    other passes return — one pass's chain, decided on the device (no read-back).  Checked here: the verdicts (all local for
    a band of ± 600 columns, none for uniform columns, mixed for a half-and-half matrix), and bit-identical results to the
    oracle in all three cases for every pinned panel plan — rows out of column order, empty rows, a row beyond the long-row
    threshold, with and without the bias — and through custom_mm.naive_spmm (AUTO + the extension's own workspace).
    Reference: src/naive_sparse_mm.cu:60-92 (one CSR-order chain per output element, whatever the matrix looks like)."""
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    capi.mi_spmm_csr_ex_variant_f32.argtypes = [ctypes.c_int, vp, vp, vp, i64, i32, i32, i32, vp, i64, vp, vp, i64, ctypes.c_int,
                                                vp, ctypes.c_size_t, vp]
    capi.mi_spmm_csr_workspace_bytes.restype = ctypes.c_size_t
    capi.mi_spmm_csr_workspace_bytes.argtypes = [i64, i32]
    stream = torch.cuda.current_stream().cuda_stream
    M, K = 20000, 24000
    g = np.random.Generator(np.random.PCG64(900 + N))

    def matrix(kind):
        lens = g.integers(0, 40, size=M)
        lens[g.integers(0, M, size=30)] = 0
        lens[4000] = 9001                                   # beyond the long-row threshold
        cols = []
        for r, n in enumerate(lens):
            banded = kind == "banded" or (kind == "mixed" and r < M // 2)
            if banded:
                centre = r * K // M
                lo, hi = max(0, centre - 600), min(K, centre + 600)
            else:
                lo, hi = 0, K
            c = g.integers(lo, hi, size=int(n))
            cols.append(np.sort(c) if r % 4 else c)          # a quarter of the rows out of column order (duplicates allowed)
        col = np.concatenate(cols).astype(np.int32)
        rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        val = g.random(len(col), dtype=np.float32) - 0.5
        return rowptr, col, val

    B, bias = g.random((K, N), dtype=np.float32) - 0.5, g.random(N, dtype=np.float32)
    d_B, d_bias = t(B, dev), t(bias, dev)
    for kind, expect_local in (("banded", 16), ("uniform", 0), ("mixed", None)):
        rowptr, col, val = matrix(kind)
        chain = oracle_mod.spmm_csr(rowptr, col, val, M, K, B)
        split = oracle_mod.spmm_csr_long(rowptr, col, val, M, K, B)
        d = [t(x, dev) for x in (rowptr, col, val)]
        ws_bytes = capi.mi_spmm_csr_workspace_bytes(len(col), N)
        ws = torch.zeros(ws_bytes + 16, dtype=torch.uint8, device=dev)
        for variant in variants:
            for with_bias in (False, True):
                for rule, want in ((0, chain), (1, split)):     # MI_LONG_ROWS_NONE / _SPLIT
                    C = torch.full((M, N), float("nan"), device=dev)
                    st = capi.mi_spmm_csr_ex_variant_f32(variant, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), len(col), M, K, N,
                                                         d_B.data_ptr(), N, d_bias.data_ptr() if with_bias else None, C.data_ptr(), N,
                                                         rule, ws.data_ptr(), ws_bytes, stream)
                    assert st == 0, (kind, variant, with_bias, rule)
                    expect = want + bias[None, :] if with_bias else want
                    assert np.array_equal(C.cpu().numpy().view(np.int32), expect.view(np.int32)), (kind, variant, with_bias, rule)
            verdicts = ws[ws_bytes - 64:ws_bytes].view(torch.int32).cpu().numpy()
            assert set(verdicts.tolist()) <= {0, 1}
            if expect_local is not None:
                assert int(verdicts.sum()) == expect_local, (kind, variant, verdicts)
            else:
                assert 0 < int(verdicts.sum()) < 14, (kind, verdicts)   # below 7/8: the passes stay passes
        got = run_spmm(cmm, dev, rowptr, col, val, M, K, B)              # AUTO through the extension (its own workspace)
        assert np.array_equal(got.view(np.int32), split.view(np.int32)) or np.array_equal(got.view(np.int32), chain.view(np.int32)), kind


def test_adaptive_panel_passes_fuzz_against_oracle(cmm, dev, oracle_mod):
    """Random structured matrices on shapes where AUTO takes an L2-level panel plan, through custom_mm.naive_spmm (the
    extension's workspace: the locality probe decides on the device whether the passes stay passes) — bands of random
    width around a diagonal of random slope, block-diagonal matrices, uniform columns and mixtures by row range; rows
    partly out of column order; a bias on every other case.  Bit-identical to the oracle whatever the probe decided.
    MI_FUZZ_CASES (default 10) / MI_FUZZ_SEED set the number of cases and the seed."""
    import os
    g = np.random.Generator(np.random.PCG64(int(os.environ.get("MI_FUZZ_SEED", "77"))))
    cases = int(os.environ.get("MI_FUZZ_CASES", "10"))
    panel_plans = 0
    for case in range(cases):
        N = int(g.choice([128, 192, 256, 320, 512]))
        M = int(g.integers(10, 40)) * 1024 if N != 128 else int(g.integers(100, 140)) * 1024
        K = int(g.integers(8, 48)) * 1024
        d = int(g.integers(24, 90)) if N != 128 else int(g.integers(100, 130))
        kind = str(g.choice(["band", "blocks", "uniform", "mixed"]))
        half = int(g.integers(50, 3000))
        rows = np.repeat(np.arange(M, dtype=np.int64), d)
        u = g.integers(0, 1 << 30, size=M * d, dtype=np.int64)
        centre = rows * K // M
        if kind == "band" or kind == "mixed":
            c_band = np.clip(centre + (u % (2 * half + 1)) - half, 0, K - 1)
        if kind == "blocks":
            nb = int(g.integers(4, 40))
            blk = rows * nb // M
            c_band = blk * (K // nb) + u % (K // nb)
        c_uni = u % K
        if kind == "uniform":
            cols = c_uni
        elif kind == "mixed":
            cut = int(g.integers(M // 4, 3 * M // 4))
            cols = np.where(rows < cut, c_band, c_uni)
        else:
            cols = c_band
        # rows sorted by column, except every fifth row (left in generation order: duplicates and descents are legal CSR)
        order = np.lexsort((np.where(rows % 5 == 0, 0, cols), rows))
        col = cols[order].astype(np.int32)
        rowptr = (np.arange(M + 1, dtype=np.int64) * d).astype(np.int32)
        val = g.random(M * d, dtype=np.float32) - 0.5
        B = g.random((K, N), dtype=np.float32) - 0.5
        d_B = t(B, dev)
        C = torch.full((M, N), float("nan"), device=dev)
        plan = cmm.spmm_plan(len(val), M, K, d_B, C)
        panel_plans += plan[1] in ("spmm_wave_row_panel_kernel", "spmm_group_panel_kernel")
        want = oracle_mod.spmm_csr(rowptr, col, val, M, K, B)
        if case % 2:
            bias = g.random(N, dtype=np.float32)
            cmm.naive_spmm_bias(t(val, dev), t(col, dev), t(rowptr, dev), len(val), M, K, d_B, t(bias, dev), C)
            want = want + bias[None, :]
        else:
            cmm.naive_spmm(t(val, dev), t(col, dev), t(rowptr, dev), len(val), M, K, d_B, C)
        assert np.array_equal(C.cpu().numpy().view(np.int32), want.view(np.int32)), (case, kind, M, K, N, d, half, plan)
    assert panel_plans >= cases // 2, panel_plans   # the fuzz is about the panel plans: most cases must reach them


def test_adaptive_panel_product_is_graph_capturable_and_decides_per_replay(cmm, dev, oracle_mod):
    """The locality probe and the passes that read its verdicts are plain launches on the caller's stream: a hipGraph
    captures a panel-plan product once, and every REPLAY decides anew on the device — the same graph is replayed on a banded
    matrix (the first pass takes everything) and, after the CSR arrays were overwritten in place, on a uniform one (the
    passes stay passes); both results bit-identical to the oracle.  Reference: one kernel for any matrix,
    src/naive_sparse_mm.cu:24-136."""
    M, K, N, d = 12288, 16384, 256, 64
    g = np.random.Generator(np.random.PCG64(31))
    rows = np.repeat(np.arange(M, dtype=np.int64), d)

    def matrix(banded):
        u = g.integers(0, 1 << 30, size=M * d, dtype=np.int64)
        cols = np.clip(rows * K // M + (u % 801) - 400, 0, K - 1) if banded else u % K
        order = np.lexsort((cols, rows))
        return cols[order].astype(np.int32), (g.random(M * d, dtype=np.float32) - 0.5)

    rowptr = (np.arange(M + 1, dtype=np.int64) * d).astype(np.int32)
    col_b, val_b = matrix(True)
    col_u, val_u = matrix(False)
    B = g.random((K, N), dtype=np.float32) - 0.5
    d_rp, d_col, d_val, d_B = t(rowptr, dev), t(col_b, dev), t(val_b, dev), t(B, dev)
    C = torch.full((M, N), float("nan"), device=dev)
    assert cmm.spmm_plan(M * d, M, K, d_B, C)[1] == "spmm_wave_row_panel_kernel"
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):  # warm-up outside capture (sizes the extension's workspace)
        cmm.naive_spmm(d_val, d_col, d_rp, M * d, M, K, d_B, C)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        cmm.naive_spmm(d_val, d_col, d_rp, M * d, M, K, d_B, C)
    for col, val in ((col_b, val_b), (col_u, val_u), (col_b, val_b)):
        d_col.copy_(torch.from_numpy(col))
        d_val.copy_(torch.from_numpy(val))
        C.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(C.cpu().numpy().view(np.int32), oracle_mod.spmm_csr(rowptr, col, val, M, K, B).view(np.int32))
