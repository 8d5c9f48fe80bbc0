"""Inspector–executor handles and the column-major executor.

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


def test_cusparse_inspect_and_mmul_opt_column_major(cmm, dev, golden, oracle_mod):
    c = golden.case("colmajor/fc")
    M, K = c["a"].shape
    N = c["x"].shape[0]
    cmm.cusparse_inspect(t(c["rowptr"], dev), t(c["col"], dev), t(c["val"], dev), len(c["val"]), M, N, K, "fc1")
    x = t(c["x"], dev)                       # activations [N, K] row-major == B column-major K×N
    y = torch.full((N, M), float("nan"), device=dev)
    out = cmm.cusparse_mmul_opt(x, y, "fc1")
    assert out.data_ptr() == y.data_ptr()
    expect = oracle_mod.spmm_csr_colmajor(c["rowptr"], c["col"], c["val"], M, K, N, c["x"]).reshape(N, M)
    assert np.array_equal(y.cpu().numpy(), expect)
    assert np.allclose(y.cpu().numpy(), c["y"], rtol=RTOL, atol=ATOL)
    cmm.cusparse_clean()
    with pytest.raises(RuntimeError, match="Invalid handle_id"):
        cmm.cusparse_mmul_opt(x, y, "fc1")


@pytest.mark.parametrize("n", [128, 1024])
def test_tiledspmm_inspect_and_multiply(cmm, dev, oracle_mod, n):
    """reference tests/tiledsppm_kernel_test.py:34-39 shapes: C[M×K] = A[M×N]·B[N×K], column-major B and C."""
    for idx, (M, N, K) in enumerate([(n, n, n), (n, 2 * n, n), (n, n, n // 2), (2 * n, n, n // 2)]):
        rowptr, col, val = oracle_mod.make_csr(M, N, 0.01, seed=idx)
        g = np.random.Generator(np.random.PCG64(idx))
        Bcm = g.random((K, N), dtype=np.float32)  # flat column-major N×K buffer
        expect = oracle_mod.spmm_csr_colmajor(rowptr, col, val, M, N, K, Bcm)
        # CSR entry: int64 host tensors (reference custom_mm.cpp:321-326)
        cmm.tiledspmm_inspect_csr(M, N, K, torch.from_numpy(rowptr.astype(np.int64)), torch.from_numpy(col.astype(np.int64)),
                                  torch.from_numpy(val), f"csr{idx}")
        # COO entry: int32 host tensors sorted by row (reference custom_mm.cpp:293-298)
        rows = np.repeat(np.arange(M, dtype=np.int32), np.diff(rowptr))
        cmm.tiledspmm_inspect_coo(M, N, K, len(val), torch.from_numpy(rows), torch.from_numpy(col), torch.from_numpy(val),
                                  f"coo{idx}")
        for layer in (f"csr{idx}", f"coo{idx}"):
            C = torch.zeros(K, M, device=dev)  # callers pre-zero (reference kernel skips empty row blocks)
            cmm.tiledspmm_mm(t(Bcm, dev), C, layer)
            assert np.array_equal(C.cpu().numpy().reshape(-1), expect), (n, idx, layer)
    cmm.tiledspmm_clean()
    with pytest.raises(RuntimeError, match="Invalid handle_id"):
        cmm.tiledspmm_mm(torch.zeros(1, device=dev), torch.zeros(1, device=dev), "csr0")
    with pytest.raises(RuntimeError, match="sorted"):
        cmm.tiledspmm_inspect_coo(4, 4, 4, 2, torch.tensor([3, 1], dtype=torch.int32), torch.tensor([0, 0], dtype=torch.int32),
                                  torch.ones(2), "bad")


def test_inspect_handles_amortise_transpose_and_long_rows(cmm, dev, oracle_mod):
    """What `cusparse_inspect` / `tiledspmm_inspect_*` keep: the validated CSR, Aᵀ,
    and the prepared long-row lists.  A weight matrix with hub rows AND a hub column: the executor
    sums long rows in the split order (same bits as cusparse_mmul on the same data, restated by
    oracle.spmm_csr_long), `*_mmul_opt_t` runs the product with the cached Aᵀ, repeated products on
    the handle give the same bits, and malformed CSR is refused at inspect time."""
    M, K, N = 300, 20000, 64
    g = np.random.Generator(np.random.PCG64(77))
    lens = g.integers(0, 120, size=M)
    lens[4], lens[100], lens[299] = 20000, 9000, 8193           # hub rows (> 8192 non-zeros)
    cols = [np.sort(g.choice(K, size=int(n), replace=False)).astype(np.int32) for n in lens]
    for c in cols[150:]:                                         # a hub column: every later row holds column 7
        if len(c) and c[0] != 7:
            c[0] = 7
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    col = np.concatenate(cols)
    val = g.random(len(col), dtype=np.float32) - 0.5
    nnz = len(val)
    x = g.random((N, K), dtype=np.float32)                       # activations [N, K] == B col-major K×N
    cmm.cusparse_inspect(t(rowptr, dev), t(col, dev), t(val, dev), nnz, M, N, K, "hub")
    info = cmm.inspect_info("hub", False)
    assert info["max_row"] == 20000 and info["long_rows_prepared"] is True
    assert info["max_row_transposed"] <= M and info["long_rows_prepared_transposed"] is False
    e_rp, e_col, e_val = oracle_mod.csr_transpose(rowptr, col, val, M, K)
    t_val, t_col, t_rp = info["transpose"]
    assert np.array_equal(t_rp.cpu().numpy(), e_rp) and np.array_equal(t_col.cpu().numpy(), e_col)
    assert np.array_equal(t_val.cpu().numpy(), e_val)
    y = torch.full((N, M), float("nan"), device=dev)
    cmm.cusparse_mmul_opt(t(x, dev), y, "hub")
    want = oracle_mod.spmm_csr_long(rowptr, col, val, M, K, np.ascontiguousarray(x.T)).T   # [N, M]
    assert np.array_equal(y.cpu().numpy(), want)
    plain = torch.empty(M, N, device=dev)                         # the non-inspect path on the same data
    cmm.cusparse_mmul(t(val, dev), t(col, dev), t(rowptr, dev), nnz, M, K, t(np.ascontiguousarray(x.T), dev), plain)
    assert np.array_equal(plain.cpu().numpy().T, y.cpu().numpy())
    y2 = torch.full((N, M), float("nan"), device=dev)
    cmm.cusparse_mmul_opt(t(x, dev), y2, "hub")                   # the prepared list is not consumed
    assert torch.equal(y, y2)
    # transposed product: dX [N, K] from dY [N, M]  (dXᵀ = Aᵀ·dYᵀ)
    dy = g.random((N, M), dtype=np.float32)
    dx = torch.full((N, K), float("nan"), device=dev)
    cmm.cusparse_mmul_opt_t(t(dy, dev), dx, "hub")
    want_t = oracle_mod.spmm_csr(e_rp, e_col, e_val, K, M, np.ascontiguousarray(dy.T)).T
    assert np.array_equal(dx.cpu().numpy(), want_t)
    assert np.allclose(dx.cpu().numpy(), dy.astype(np.float64) @ _dense_of(rowptr, col, val, M, K), rtol=1e-4, atol=1e-4)
    # a hub COLUMN long enough to make a long row of Aᵀ: the transposed side prepares its own list
    M2, K2 = 9000, 50
    lens2 = np.full(M2, 3)
    cols2 = [np.array([0, 1 + r % 20, 30 + r % 20], dtype=np.int32) for r in range(M2)]   # column 0 in every row
    rowptr2 = np.concatenate([[0], np.cumsum(lens2)]).astype(np.int32)
    col2, val2 = np.concatenate(cols2), g.random(3 * M2, dtype=np.float32)
    cmm.tiledspmm_inspect_csr(M2, K2, 8, torch.from_numpy(rowptr2.astype(np.int64)), torch.from_numpy(col2.astype(np.int64)),
                              torch.from_numpy(val2), "hubcol")
    info2 = cmm.inspect_info("hubcol", True)
    assert info2["max_row"] == 3 and info2["max_row_transposed"] == M2 and info2["long_rows_prepared_transposed"] is True
    dy2 = g.random((8, M2), dtype=np.float32)
    dx2 = torch.full((8, K2), float("nan"), device=dev)
    cmm.tiledspmm_mm_t(t(dy2, dev), dx2, "hubcol")
    t2 = oracle_mod.csr_transpose(rowptr2, col2, val2, M2, K2)
    assert np.array_equal(dx2.cpu().numpy(), oracle_mod.spmm_csr_long(*t2, K2, M2, np.ascontiguousarray(dy2.T)).T)
    # malformed CSR is refused once, at inspect time
    bad_col = col.copy()
    bad_col[5] = K
    with pytest.raises(RuntimeError, match="column index out of range"):
        cmm.cusparse_inspect(t(rowptr, dev), t(bad_col, dev), t(val, dev), nnz, M, N, K, "bad")
    bad_rp = rowptr.copy()
    bad_rp[10] = bad_rp[11] + 1
    with pytest.raises(RuntimeError, match="must not decrease"):
        cmm.cusparse_inspect(t(bad_rp, dev), t(col, dev), t(val, dev), nnz, M, N, K, "bad")
    with pytest.raises(RuntimeError, match="Invalid handle_id"):
        cmm.inspect_info("bad", False)
    cmm.cusparse_clean()
    cmm.tiledspmm_clean()


@pytest.mark.parametrize("M,K,N,density,native", [(1024, 256, 4096, 0.5, True), (1100, 300, 4100, 0.6, True),
                                                  (20011, 1030, 516, 0.25, False)])
def test_column_major_executor_native_slab_form(cmm, capi, dev, oracle_mod, M, K, N, density, native):
    """Where the LDS-slab plan serves the product, the column-major executor reads the activations
    X = Bᵀ [N, K] and writes Y = Cᵀ [N, M] directly (transposing slab loads, transposed tile store, no
    transposed copies): bit-identical to the CSR-order oracle, ragged edges, empty rows and rows whose
    columns do not ascend included; the transposed product with the cached Aᵀ takes whichever form its own
    shape selects and is checked the same way."""
    g = np.random.Generator(np.random.PCG64(M + N))
    rowptr, col, val = oracle_mod.make_csr(M, K, density, seed=M % 97)
    keep = np.ones(len(col), bool)
    for r in (3, 130, M - 2):                            # three empty rows
        keep[rowptr[r]:rowptr[r + 1]] = False
    lens = np.diff(rowptr)
    lens[[3, 130, M - 2]] = 0
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    col, val = col[keep].copy(), val[keep].copy() - 0.5
    for r in (0, 5, M // 2, M - 1):                      # a few rows out of column order
        s0, e0 = rowptr[r], rowptr[r + 1]
        perm = g.permutation(e0 - s0)
        col[s0:e0], val[s0:e0] = col[s0:e0][perm], val[s0:e0][perm]
    x = g.random((N, K), dtype=np.float32) - 0.5
    probe_b, probe_c = torch.empty(K, N, device=dev), torch.empty(M, N, device=dev)
    assert cmm.spmm_plan(len(val), M, K, probe_b, probe_c)[1] == "spmm_slab_kernel"
    cmm.cusparse_inspect(t(rowptr, dev), t(col, dev), t(val, dev), len(val), M, N, K, "slabcm")
    y = torch.full((N, M), float("nan"), device=dev)
    d_x = t(x, dev)
    capi.mi_spmm_colmajor_native_form.argtypes = [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                  ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]
    assert capi.mi_spmm_colmajor_native_form(len(val), M, K, N, d_x.data_ptr(), K, y.data_ptr(), M) == int(native)
    cmm.cusparse_mmul_opt(t(x, dev), y, "slabcm")
    want = oracle_mod.spmm_csr_colmajor(rowptr, col, val, M, K, N, x).reshape(N, M)
    assert np.array_equal(y.cpu().numpy(), want)
    dy = g.random((N, M), dtype=np.float32) - 0.5
    dx = torch.full((N, K), float("nan"), device=dev)
    cmm.cusparse_mmul_opt_t(t(dy, dev), dx, "slabcm")
    t_rp, t_col, t_val = oracle_mod.csr_transpose(rowptr, col, val, M, K)
    assert np.array_equal(dx.cpu().numpy(), oracle_mod.spmm_csr_colmajor(t_rp, t_col, t_val, K, M, N, dy).reshape(N, K))
    cmm.cusparse_clean()


@pytest.mark.parametrize("M,K,N,density", [(4096, 4096, 512, 0.05), (1003, 700, 256, 0.02), (530, 1200, 1024, 0.01),
                                           (37, 64, 512, 0.3)])
def test_column_major_executor_fused_output_form(cmm, capi, dev, oracle_mod, M, K, N, density):
    """Where the one-wave-per-row plan serves the product, the executor transposes the activations in and the
    kernel writes Y = Cᵀ [N, M] from its epilogue (16 rows per workgroup meet in LDS, 64-byte pieces out):
    bit-identical to the CSR-order oracle — ragged M (scalar tail, M % 4 ≠ 0), empty rows, rows out of column
    order; the transposed product with the cached Aᵀ is checked the same way."""
    g = np.random.Generator(np.random.PCG64(M + N))
    rowptr, col, val = oracle_mod.make_csr(M, K, density, seed=M % 89)
    keep = np.ones(len(col), bool)
    empties = [2, M // 3, M - 1]
    for r in empties:
        keep[rowptr[r]:rowptr[r + 1]] = False
    lens = np.diff(rowptr)
    lens[empties] = 0
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    col, val = col[keep].copy(), val[keep].copy() - 0.5
    for r in (0, 7, M // 2):                             # a few rows out of column order
        s0, e0 = rowptr[r], rowptr[r + 1]
        perm = g.permutation(e0 - s0)
        col[s0:e0], val[s0:e0] = col[s0:e0][perm], val[s0:e0][perm]
    x = g.random((N, K), dtype=np.float32) - 0.5
    cmm.cusparse_inspect(t(rowptr, dev), t(col, dev), t(val, dev), len(val), M, N, K, "fusedcm")
    y = torch.full((N, M), float("nan"), device=dev)
    d_x = t(x, dev)
    probe_b, probe_c = torch.empty(K, N, device=dev), torch.empty(M, N, device=dev)
    assert cmm.spmm_plan(len(val), M, K, probe_b, probe_c)[1] == "spmm_wave_row_kernel"
    capi.mi_spmm_colmajor_form.argtypes = [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
                                           ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    assert capi.mi_spmm_colmajor_form(len(val), M, K, N, d_x.data_ptr(), K, y.data_ptr(), M, probe_b.data_ptr()) == 2
    cmm.cusparse_mmul_opt(d_x, y, "fusedcm")
    want = oracle_mod.spmm_csr_colmajor(rowptr, col, val, M, K, N, x).reshape(N, M)
    assert np.array_equal(y.cpu().numpy(), want)
    dy = g.random((N, M), dtype=np.float32) - 0.5
    dx = torch.full((N, K), float("nan"), device=dev)
    cmm.cusparse_mmul_opt_t(t(dy, dev), dx, "fusedcm")
    t_rp, t_col, t_val = oracle_mod.csr_transpose(rowptr, col, val, M, K)
    assert np.array_equal(dx.cpu().numpy(), oracle_mod.spmm_csr_colmajor(t_rp, t_col, t_val, K, M, N, dy).reshape(N, K))
    cmm.cusparse_clean()


def test_inspector_products_and_round2_entries_are_graph_capturable(cmm, dev, oracle_mod):
    """An inspector handle owns its buffers, so `cusparse_mmul_opt` / `_opt_t` neither allocate nor
    synchronise and can be captured in a hipGraph — every form of the executor (transposes around the
    row-split kernel; native LDS-slab; transpose in + column-major output fused into the kernel) — as can `naive_spmm_ex` (one launch, no workspace) and the chained
    short-k GEMM; the graph replays on new activations."""
    g = np.random.Generator(np.random.PCG64(21))
    cases = []
    for tag, (M, K, N, density) in {"rows": (900, 700, 64, 0.02), "slab": (1024, 256, 4096, 0.5),
                                    "fused-out": (700, 900, 1024, 0.02)}.items():  # 1024 columns: > 64 KiB of LDS per workgroup
        rowptr, col, val = oracle_mod.make_csr(M, K, density, seed=len(tag))
        cmm.cusparse_inspect(t(rowptr, dev), t(col, dev), t(val, dev), len(val), M, N, K, tag)
        cases.append((tag, M, K, N, rowptr, col, val, torch.zeros(N, K, device=dev), torch.empty(N, M, device=dev),
                      torch.zeros(N, M, device=dev), torch.empty(N, K, device=dev)))
    M0, K0 = 2000, 1500
    rp0, c0, v0 = oracle_mod.make_csr(M0, K0, 0.01, seed=3)
    d0 = [t(v0, dev), t(c0, dev), t(rp0, dev)]
    B0, C0 = torch.zeros(K0, 256, device=dev), torch.empty(M0, 256, device=dev)
    q, kk, S = torch.zeros(4, 256, 64, device=dev), torch.zeros(4, 256, 64, device=dev), torch.empty(4, 256, 256, device=dev)

    def run():
        for tag, M, K, N, *_rest, x, y, dy, dx in cases:
            cmm.cusparse_mmul_opt(x, y, tag)
            cmm.cusparse_mmul_opt_t(dy, dx, tag)
        cmm.naive_spmm_ex(*d0, len(v0), M0, K0, B0, C0, 0)
        cmm.cublas_bmm(q, kk, S, 3, False, True)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run()  # warm-up outside capture
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        run()
    for _ in range(2):
        fresh = []
        for tag, M, K, N, rowptr, col, val, x, y, dy, dx in cases:
            xh, dyh = g.random((N, K), dtype=np.float32), g.random((N, M), dtype=np.float32)
            x.copy_(torch.from_numpy(xh))
            dy.copy_(torch.from_numpy(dyh))
            fresh.append((xh, dyh))
        Bh, qh, kh = g.random((K0, 256), dtype=np.float32), g.random(q.shape, dtype=np.float32), g.random(kk.shape, dtype=np.float32)
        B0.copy_(torch.from_numpy(Bh))
        q.copy_(torch.from_numpy(qh))
        kk.copy_(torch.from_numpy(kh))
        graph.replay()
        torch.cuda.synchronize()
        for (tag, M, K, N, rowptr, col, val, x, y, dy, dx), (xh, dyh) in zip(cases, fresh):
            assert np.array_equal(y.cpu().numpy(), oracle_mod.spmm_csr_colmajor(rowptr, col, val, M, K, N, xh).reshape(N, M)), tag
            t_rp, t_col, t_val = oracle_mod.csr_transpose(rowptr, col, val, M, K)
            assert np.array_equal(dx.cpu().numpy(), oracle_mod.spmm_csr_colmajor(t_rp, t_col, t_val, K, M, N, dyh).reshape(N, K)), tag
        assert np.array_equal(C0.cpu().numpy(), oracle_mod.spmm_csr(rp0, c0, v0, M0, K0, Bh))
        assert np.array_equal(S.cpu().numpy(), oracle_mod.gemm(qh, kh, False, True))
    cmm.cusparse_clean()


@pytest.mark.parametrize("K", [301, 302, 303])
def test_column_major_native_form_with_a_padded_leading_dimension(capi, cmm, dev, oracle_mod, K):
    """Advisor (round 2, medium): through the C-ABI the activations may come with ldb > K; with K % 4 != 0 the
    native LDS-slab form read its partial last k-quad at the wrong rows.  K % 4 = 1, 2, 3 with ldb = K rounded up,
    on a slab-plan shape: bit-identical to the oracle (executor of reference src/baseline_mm.cu:272-321)."""
    M, N, density = 1100, 4100, 0.6
    ldb = (K + 3) // 4 * 4
    g = np.random.Generator(np.random.PCG64(K))
    rowptr, col, val = oracle_mod.make_csr(M, K, density, seed=K)
    val = val - 0.5
    x = g.random((N, K), dtype=np.float32) - 0.5
    xp = np.full((N, ldb), np.float32(777.0))   # the padding must never reach a result
    xp[:, :K] = x
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    capi.mi_spmm_colmajor_native_form.argtypes = [i64, i32, i32, i32, vp, i64, vp, i64]
    capi.mi_spmm_colmajor_workspace_bytes.restype = ctypes.c_size_t
    capi.mi_spmm_colmajor_workspace_bytes.argtypes = [i32, i32, i32]
    capi.mi_spmm_csr_colmajor_f32.argtypes = [vp, vp, vp, i64, i32, i32, i32, vp, i64, vp, i64, vp, ctypes.c_size_t, vp]
    d = [t(a, dev) for a in (rowptr, col, val, xp)]
    y = torch.full((N, M), float("nan"), device=dev)
    assert capi.mi_spmm_colmajor_native_form(len(val), M, K, N, d[3].data_ptr(), ldb, y.data_ptr(), M) == 1
    nbytes = capi.mi_spmm_colmajor_workspace_bytes(M, K, N)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    assert capi.mi_spmm_csr_colmajor_f32(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), len(val), M, K, N,
                                         d[3].data_ptr(), ldb, y.data_ptr(), M, ws.data_ptr(), nbytes,
                                         torch.cuda.current_stream().cuda_stream) == 0
    want = oracle_mod.spmm_csr_colmajor(rowptr, col, val, M, K, N, x).reshape(N, M)
    assert np.array_equal(y.cpu().numpy(), want)


def test_validate_csr_rejects_bad_contents_for_every_plan(cmm, dev, oracle_mod):
    """The per-product entry points trust CSR contents as the reference does (src/naive_sparse_mm.cu:60-92) and an
    out-of-range column fails differently under different plans; custom_mm.validate_csr is the opt-in check — it
    must catch an out-of-range column, a negative column and non-monotone offsets on inputs that would take the
    row-split, the two L2-panel plans and the LDS-slab plan alike (the check does not depend on the plan: asserted per shape)."""
    shapes = [("spmm_wave_row_kernel", 2000, 3000, 256, 0.01), ("spmm_wave_row_panel_kernel", 16384, 16384, 256, 0.005),
              ("spmm_group_panel_kernel", 16384, 16384, 256, 0.02),
              ("spmm_slab_kernel", 4096, 4096, 2048, 0.2), ("spmm_group_kernel", 500, 700, 64, 0.05)]
    for plan, M, K, N, density in shapes:
        rowptr, col, val = oracle_mod.make_csr(M, K, density, seed=M)
        nnz = len(val)
        b, c = torch.empty(K, N, device=dev), torch.empty(M, N, device=dev)
        assert cmm.spmm_plan(nnz, M, K, b, c)[1] == plan, (plan, cmm.spmm_plan(nnz, M, K, b, c))
        d_val = t(val, dev)
        cmm.validate_csr(d_val, t(col, dev), t(rowptr, dev), nnz, M, K)  # the good matrix passes
        bad = col.copy(); bad[nnz // 2] = K
        with pytest.raises(RuntimeError):
            cmm.validate_csr(d_val, t(bad, dev), t(rowptr, dev), nnz, M, K)
        bad = col.copy(); bad[7] = -1
        with pytest.raises(RuntimeError):
            cmm.validate_csr(d_val, t(bad, dev), t(rowptr, dev), nnz, M, K)
        rp = rowptr.copy(); rp[M // 2] = rp[M // 2 + 1] + 1
        with pytest.raises(RuntimeError):
            cmm.validate_csr(d_val, t(col, dev), t(rp, dev), nnz, M, K)
        rp = rowptr.copy(); rp[-1] = nnz - 1
        with pytest.raises(RuntimeError):
            cmm.validate_csr(d_val, t(col, dev), t(rp, dev), nnz, M, K)
