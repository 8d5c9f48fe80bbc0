"""The drop-in boundary without a GPU: libmi_spmm.so loads and exports every
symbol include/mi_spmm.h declares; argument validation that returns before any
HIP call; the host inspector; the custom_mm module surface and its error
behaviour on CPU tensors (no CPU fallback)."""
import ctypes
import re
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

REPO = Path(__file__).resolve().parent.parent
HEADER = REPO / "include" / "mi_spmm.h"

REFERENCE_NAMES = [  # reference src/custom_mm.cpp:393-416
    "init_cublas", "destroy_cublas", "init_cusparse", "destroy_cusparse", "cublas_mmul", "cublas_bmm",
    "cusparse_mmul", "dummy_kernel", "naive_spmm", "tiledspmm_inspect_csr", "tiledspmm_inspect_coo",
    "tiledspmm_mm", "tiledspmm_clean", "cusparse_inspect", "cusparse_mmul_opt", "cusparse_clean"]


def declared_functions():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib(built):
    import torch  # noqa: F401  (torch's HIP runtime first, as in the product)
    return ctypes.CDLL(str(built / "libmi_spmm.so"))


def test_header_declares_the_expected_entry_points():
    names = declared_functions()
    assert len(names) >= 38, names
    for must in ("mi_csr_transpose_batched_f32", "mi_csr_transpose_batched_workspace_bytes", "mi_spmm_long_rows_prepare",
                 "mi_spmm_csr_colmajor_ex_f32", "mi_spmm_colmajor_form", "mi_spmm_csr_ex_f32", "mi_spmm_auto_splits_long_rows", "mi_spmm_long_row_threshold", "mi_spmm_csr_f32", "mi_spmm_csr_batched_f32", "mi_spmm_csr_colmajor_f32", "mi_gemm_f32",
                 "mi_dense_to_csr_count", "mi_dense_to_csr_fill", "mi_csr_transpose_f32", "mi_sddmm_csr_f32",
                 "mi_coo_to_csr_host", "mi_dummy_kernel"):
        assert must in names


def test_library_exports_every_declared_symbol(lib):
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in include/mi_spmm.h but not exported"
    assert lib.mi_spmm_abi_version() == 1


def test_status_strings(lib):
    lib.mi_status_string.restype = ctypes.c_char_p
    assert lib.mi_status_string(0) == b"ok"
    seen = {lib.mi_status_string(-i) for i in range(0, 6)}
    assert len(seen) == 6 and b"unknown status" not in seen
    assert lib.mi_status_string(-99) == b"unknown status"


def test_argument_validation_needs_no_gpu(lib):
    """Bad arguments and empty problems return before any HIP call."""
    i64, i32, vp = ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p
    lib.mi_spmm_csr_f32.argtypes = [vp, vp, vp, i64, i32, i32, i32, vp, i64, vp, i64, vp]
    assert lib.mi_spmm_csr_f32(None, None, None, 0, -1, 4, 4, None, 4, None, 4, None) == -1   # MI_EINVAL
    assert lib.mi_spmm_csr_f32(None, None, None, 0, 0, 4, 4, None, 4, None, 4, None) == 0     # M == 0: nothing to do
    assert lib.mi_spmm_csr_f32(None, None, None, 2 ** 31, 4, 4, 4, None, 4, None, 4, None) == -2  # MI_ERANGE
    assert lib.mi_spmm_csr_f32(None, None, None, 0, 4, 4, 4, None, 4, None, 4, None) == -1   # null rowptr / C
    lib.mi_gemm_f32.argtypes = [ctypes.c_int, ctypes.c_int, i32, i32, i32, vp, i64, i64, vp, i64, i64, vp, i64, i64,
                                i32, vp]
    assert lib.mi_gemm_f32(0, 0, -1, 2, 2, None, 2, 0, None, 2, 0, None, 2, 0, 1, None) == -1
    assert lib.mi_gemm_f32(0, 0, 0, 2, 2, None, 2, 0, None, 2, 0, None, 2, 0, 1, None) == 0
    assert lib.mi_gemm_f32(0, 0, 2, 2, 0, None, 2, 0, None, 2, 0, None, 2, 0, 70000, None) == -1  # null C
    assert lib.mi_gemm_f32(1, 1, 2, 2, 2, None, 2, 0, None, 2, 0, None, 1, 0, 1, None) == -1       # ldc < n
    lib.mi_spmm_colmajor_workspace_bytes.restype = ctypes.c_size_t
    lib.mi_spmm_colmajor_workspace_bytes.argtypes = [i32, i32, i32]
    assert lib.mi_spmm_colmajor_workspace_bytes(10, 20, 5) >= 4 * (20 * 5 + 10 * 5)
    lib.mi_dense_to_csr_workspace_bytes.restype = ctypes.c_size_t
    lib.mi_dense_to_csr_workspace_bytes.argtypes = [i32, i32]
    assert lib.mi_dense_to_csr_workspace_bytes(3, 7) >= 4 * 21


def test_round2_entry_points_validate_without_a_gpu(lib):
    """The entries added in round 2 — pinned long-row rule, prepared lists, batched transpose, the
    column-major executor with its long-row workspace — refuse bad arguments before any HIP call, and
    the workspace queries are plain host functions."""
    i64, i32, vp, sz = ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_size_t
    lib.mi_spmm_csr_ex_f32.argtypes = [vp, vp, vp, i64, i32, i32, i32, vp, i64, vp, vp, i64, ctypes.c_int, vp, sz, vp]
    ex = lib.mi_spmm_csr_ex_f32
    assert ex(None, None, None, 0, 4, 4, 4, None, 4, None, None, 4, 3, None, 0, None) == -1    # unknown mode
    assert ex(None, None, None, 0, 4, 4, 4, None, 4, None, None, 4, -2, None, 0, None) == -1
    assert ex(None, None, None, 10000, 4, 4, 4, None, 4, None, None, 4, 1, None, 0, None) == -1  # SPLIT needs a workspace
    assert ex(None, None, None, 0, 0, 4, 4, None, 4, None, None, 4, 0, None, 0, None) == 0      # M == 0
    assert lib.mi_spmm_long_row_threshold() == 8192
    lib.mi_spmm_auto_splits_long_rows.argtypes = [i64, i32, i32, i32, vp, i64, vp, i64]
    assert lib.mi_spmm_auto_splits_long_rows(100, 10, 10, 64, None, 64, None, 64) == 0        # nnz <= threshold
    assert lib.mi_spmm_auto_splits_long_rows(10 ** 6, 1000, 30000, 64, None, 64, None, 64) == 1
    assert lib.mi_spmm_auto_splits_long_rows(10 ** 6, 1000, 30000, 2, None, 2, None, 2) == 0  # narrow kernel: own order
    lib.mi_spmm_long_rows_prepare.argtypes = [vp, i32, i64, i32, vp, sz, vp]
    assert lib.mi_spmm_long_rows_prepare(None, 10, 100, 8, None, 0, None) == 0                # no row can be long
    assert lib.mi_spmm_long_rows_prepare(None, 10, 100000, 8, None, 0, None) == -1            # null rowptr / workspace
    assert lib.mi_spmm_long_rows_prepare(None, -1, 100, 8, None, 0, None) == -1
    lib.mi_csr_transpose_batched_workspace_bytes.restype = sz
    lib.mi_csr_transpose_batched_workspace_bytes.argtypes = [i32, i32, i32, i64]
    lib.mi_csr_transpose_workspace_bytes.restype = sz
    lib.mi_csr_transpose_workspace_bytes.argtypes = [i32, i32, i64]
    one = lib.mi_csr_transpose_workspace_bytes(1 << 20, 1 << 20, 110_000_000)
    assert 8 * 110_000_000 <= one < 12 * 110_000_000       # the 8-byte intermediate + tables, not 20 B per entry
    assert lib.mi_csr_transpose_batched_workspace_bytes(1, 1 << 20, 1 << 20, 110_000_000) == one
    assert lib.mi_csr_transpose_batched_workspace_bytes(0, 5, 5, 10) == 0
    lib.mi_csr_transpose_batched_f32.argtypes = [vp, vp, vp, i64, i32, i32, i32, vp, vp, vp, vp, sz, vp]
    tr = lib.mi_csr_transpose_batched_f32
    assert tr(None, None, None, 0, 0, 4, 4, None, None, None, None, 0, None) == 0             # batch == 0
    assert tr(None, None, None, 5, -1, 4, 4, None, None, None, None, 0, None) == -1
    assert tr(None, None, None, 5, 2, 4, 4, None, None, None, None, 0, None) == -1            # null t_rowptr
    assert tr(None, None, None, 5, 70000, 70000, 4, None, None, None, None, 0, None) == -2   # batch*(M+1) beyond int32
    lib.mi_spmm_csr_colmajor_ex_f32.argtypes = [vp, vp, vp, i64, i32, i32, i32, vp, i64, vp, i64, ctypes.c_int, vp, sz, vp, sz, vp]
    cm = lib.mi_spmm_csr_colmajor_ex_f32
    assert cm(None, None, None, 0, 0, 4, 4, None, 4, None, 4, 0, None, 0, None, 0, None) == 0  # M == 0
    assert cm(None, None, None, 0, 4, 4, 4, None, 4, None, 4, 0, None, 0, None, 0, None) == -1  # null rowptr / C


def test_every_variant_id_has_a_kernel_name_and_the_plan_query_needs_no_gpu(lib):
    """The MI_SPMM_* enum of mi_spmm.h, the name table and the AUTO planner agree (host-only calls)."""
    import re
    hdr = (Path(__file__).resolve().parent.parent / "include" / "mi_spmm.h").read_text()
    count = int(re.search(r"MI_SPMM_VARIANT_COUNT\s*=\s*(\d+)", hdr).group(1))
    ids = {int(v) for v in re.findall(r"MI_SPMM_[A-Z0-9_]+\s*=\s*(\d+)", hdr)} - {count}
    assert ids == set(range(count))
    lib.mi_spmm_variant_name.restype = ctypes.c_char_p
    for v in range(1, count):
        assert lib.mi_spmm_variant_name(v) != b"unknown", v
    assert lib.mi_spmm_variant_name(count) == b"unknown"
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    lib.mi_spmm_csr_f32_plan.argtypes = [i64, i32, i32, i32, vp, i64, vp, i64]
    al = 4096  # an aligned fake address: the planner only looks at alignment
    plan = lambda nnz, M, K, N: lib.mi_spmm_csr_f32_plan(nnz, M, K, N, al, N, al, N)  # noqa: E731
    assert plan(109945643, 1 << 20, 1 << 20, 256) == 7      # C3: two Infinity-Cache panels
    assert plan(4292815, 65536, 65536, 128) == 4            # C2: group kernel, 16 B per lane
    assert plan(100, 64, 64, 1) == 16                       # SpMV-like: narrow kernel
    assert plan(int(0.1 * 8192 * 8192), 8192, 8192, 8192) == 17   # pruned-weight density: LDS slabs
    assert plan(int(0.01 * 8192 * 8192), 8192, 8192, 8192) == 15  # 1 %: column tiles x row panels
    assert plan(int(0.1 * 8192 * 4096), 8192, 4096, 256) == 2     # too few 128 x 256 blocks for the slab kernel; B (4 MiB) in L2
    assert plan(int(0.1 * 16384 * 8192), 16384, 8192, 256) == 19  # B = 8 MiB, long rows: two L2 panels on the lane-group panel kernel
    assert plan(int(0.1 * 16384 * 16384), 16384, 16384, 256) == 20  # B = 16 MiB: three of them beat the slab plan
    assert plan(int(0.05 * 8192 * 65536), 8192, 65536, 256) == 22   # 64 MiB, 3277 per row: six (12 MiB panels for very long rows)
    assert plan(int(0.01 * 32768 * 32768), 32768, 32768, 192) == 21 and plan(int(0.01 * 16384 * 32768), 16384, 32768, 384) == 23
    assert plan(40 * 16384, 16384, 16384, 256) == 9 and plan(20 * 16384, 16384, 16384, 256) == 4  # < 24 gathers per row of B: one pass (few rows at N = 256: the lane-group kernel)
    assert plan(499 * 1536, 1536, 9728, 256) == 2                  # fewer rows than the chip holds waves: passes only multiply latency
    assert plan(45 * 162560, 162560, 106240, 256) == 2            # 104 MiB with short rows: the panels they can pay for miss the L2s
    assert plan(66 * 230656, 230656, 9472, 768) == 22             # many rows: panels of ≈ 4 MiB, 8 + 200 000 / M entries per row and pass are enough
    assert plan(333 * 129536, 129536, 22784, 128) == 19 and plan(655 * 16384, 16384, 65536, 128) == 4   # N ≤ 128: throughput-bound launches only
    assert plan(341 * 55040, 55040, 268800, 64) == 19 and plan(172 * 24832, 24832, 25856, 96) == 4  # … ≥ 28 K rows at N = 64, ≥ 60 K beyond
    assert plan(506 * 2816, 2816, 6400, 32) == 14 and plan(63 * 969984, 969984, 2304, 32) == 4      # N = 32, few long rows: 16 lanes per row
    assert plan(478 * 4096, 4096, 49920, 192) == 4                # few rows and a small product: one pass
    assert plan(210 * 117248, 117248, 295936, 32) == 19 and plan(17 * 645120, 645120, 117248, 32) == 4   # N = 32: many LONG rows only
    assert plan(260 * 271104, 271104, 29952, 96) == 4 and plan(624 * 67328, 67328, 90624, 100) == 19      # 64 < N < 128: from 24 MiB of B
    assert plan(1010 * 242176, 242176, 188928, 192) == 23         # up to 192 MiB of B
    assert plan(168 * 961024, 961024, 82176, 128) == 21 and plan(502 * 19712, 19712, 38400, 1024) == 12
    assert plan(393 * 4096, 4096, 13056, 256) == 4 and plan(38 * 1280, 1280, 14592, 256) == 2 and plan(847 * 2304, 2304, 59904, 384) == 20   # few rows: only where B is far beyond the L2s, and in three passes at most
    assert plan(8 * 16384, 16384, 16384, 256) == 4                 # too short to carry C at all
    # B beyond the Infinity Cache (round 5, fitted on tools/bench_hbm_regime.py): P ≈ |B| / 683 MiB panels when the rows
    # are long enough (≥ 16 P non-zeros per row), the lane-group panel kernel for N ≤ 128; none beyond ≈ 6 GiB
    m2, m4 = 1 << 21, 1 << 22
    assert plan(100 * m2, m2, m2, 256) == 8                        # 2 GiB: three panels
    assert plan(100 * m2, m2, m2, 512) == 11                       # 4 GiB: six panels
    assert plan(20 * m2, m2, m2, 256) == 2                         # 20 per row: one pass
    assert plan(100 * m2, m2, m2, 128) == 19                       # 1 GiB at N = 128: two group panels
    assert plan(100 * m4, m4, m4, 128) == 20                       # 2 GiB at N = 128: three
    assert plan(100 * m4, m4, m4, 64) == 19 and plan(20 * m4, m4, m4, 64) == 4   # N = 64: two panels / one pass for short rows
    assert plan(100 * m4, m4, m4, 512) == 2                        # 8 GiB: no panel count pays
    assert plan(100 * m2, m2, m2, 192) == 19 and plan(100 * m2, m2, m2, 320) == 21   # other widths: lane-group panels (whole wave, column tiles)
    assert plan(100 * m2, m2, m2, 768) == 4 and plan(400 * m2, m2, m2, 768) == 23    # 6 GiB: eight panels need long rows


def test_host_inspector_coo_to_csr(lib, golden, oracle_mod):
    c = golden.case("coo")
    M, nnz = c["a"].shape[0], len(c["val"])
    ip = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
    fp = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
    lib.mi_coo_to_csr_host.argtypes = [ctypes.c_int32, ctypes.c_int64, ip, ip, fp, ip, ip, fp]
    rp, col, val = np.zeros(M + 1, np.int32), np.zeros(nnz, np.int32), np.zeros(nnz, np.float32)
    assert lib.mi_coo_to_csr_host(M, nnz, c["row"], c["col"], c["val"], rp, col, val) == 0
    assert np.array_equal(rp, c["rowptr"]) and np.array_equal(col, c["csr_col"]) and np.array_equal(val, c["csr_val"])
    o_rp, o_col, o_val = oracle_mod.coo_to_csr(M, c["row"], c["col"], c["val"])
    assert np.array_equal(rp, o_rp) and np.array_equal(col, o_col) and np.array_equal(val, o_val)
    # unsorted rows are rejected (the reference silently builds a wrong CSR, src/sparse_mm.cu:110-134)
    assert lib.mi_coo_to_csr_host(M, nnz, c["row"][::-1].copy(), c["col"], c["val"], rp, col, val) == -5
    bad = c["row"].copy()
    bad[0] = M
    assert lib.mi_coo_to_csr_host(M, nnz, bad, c["col"], c["val"], rp, col, val) == -1


@pytest.fixture(scope="module")
def custom_mm(built):
    sys.modules.pop("custom_mm", None)
    import custom_mm
    assert Path(custom_mm.__file__).parent == built
    return custom_mm


def test_custom_mm_surface(custom_mm):
    for name in REFERENCE_NAMES:
        assert callable(getattr(custom_mm, name)), name
    for extra in ("dense_to_csr", "naive_spmm_batched", "csr_transpose", "sddmm", "naive_spmm_dense",
                  "naive_spmm_dense_bias", "cublas_mmul_bias", "naive_spmm_bias", "naive_spmm_bias_ex", "column_sums"):
        assert callable(getattr(custom_mm, extra)), extra
    # positional-only, like the reference's m.def without py::arg
    with pytest.raises(TypeError):
        custom_mm.cublas_mmul(A=torch.zeros(1, 1))
    # handle init / destroy are cheap, idempotent no-throw calls (reference custom_mm.cpp:361-391)
    for _ in range(2):
        custom_mm.init_cublas()
        custom_mm.init_cusparse()
        custom_mm.destroy_cusparse()
        custom_mm.destroy_cublas()
    custom_mm.cusparse_clean()
    custom_mm.tiledspmm_clean()


def test_custom_mm_has_no_cpu_path(custom_mm):
    a, b, c = torch.rand(2, 3), torch.rand(3, 4), torch.zeros(2, 4)
    with pytest.raises(RuntimeError, match="device"):
        custom_mm.cublas_mmul(a, b, c, False, False)
    with pytest.raises(RuntimeError, match="device"):
        custom_mm.cublas_bmm(a[None], b[None], c[None], 3, False, False)
    csr = a.to_sparse_csr()
    args = (csr.values(), csr.col_indices().int(), csr.crow_indices().int(), 6, 2, 3, b, c)
    with pytest.raises(RuntimeError, match="device"):
        custom_mm.naive_spmm(*args)
    with pytest.raises(RuntimeError, match="device"):
        custom_mm.cusparse_mmul(*args)
    with pytest.raises(RuntimeError, match="device"):
        custom_mm.dense_to_csr(a)


def test_custom_mm_error_convention(custom_mm):
    a = torch.rand(2, 2)
    with pytest.raises(ValueError, match="Invalid dim"):  # std::invalid_argument, reference custom_mm.cpp:162
        custom_mm.cublas_bmm(a, a, a, 5, False, False)
    with pytest.raises(RuntimeError, match="Invalid handle_id"):  # reference custom_mm.cpp:262,340
        custom_mm.cusparse_mmul_opt(a, a, "no-such-layer")
    with pytest.raises(RuntimeError, match="Invalid handle_id"):
        custom_mm.tiledspmm_mm(a, a, "no-such-layer")
    with pytest.raises(RuntimeError):  # host inspector inputs must be CPU int64 / float32
        custom_mm.tiledspmm_inspect_csr(2, 2, 2, torch.zeros(3, dtype=torch.int32), torch.zeros(0, dtype=torch.int64),
                                        torch.zeros(0), "x")


def test_config_c1_dense_host_operands_through_the_wrappers(built):
    """BASELINE.json configs[0], literally: "torch.mm dense 8×64 @ 64×8 on CPU via the matmuls.py wrapper".  With the REAL
    extension imported, two dense host tensors take the reference's own CPU expression `a @ b` (reference matmuls.py:39-41,
    279, 302) — torch, never the oracle — forward and backward, through every class; the README shapes (reference
    README.md:24-30)."""
    for k in ("custom_mm", "matmuls"):
        sys.modules.pop(k, None)
    import matmuls
    assert "fake" not in matmuls.custom_mm.__name__ and matmuls._REAL_EXTENSION
    torch.manual_seed(0)
    a0, b0 = torch.rand(8, 64), torch.rand(64, 8)
    dc = torch.rand(8, 8)
    for cls, fa, fb in ((matmuls.cublasMM, lambda x: x, lambda x: x), (matmuls.naiveSpMM, lambda x: x, lambda x: x),
                        (matmuls.cusparseMM, lambda x: x, lambda x: x), (matmuls.cublasTransbMM, lambda x: x, lambda x: x.t().contiguous()),
                        (matmuls.cublasTransaMM, lambda x: x.t().contiguous(), lambda x: x),
                        (matmuls.cublasTransabMM, lambda x: x.t().contiguous(), lambda x: x.t().contiguous())):
        a1, b1 = fa(a0).clone().requires_grad_(True), fb(b0).clone().requires_grad_(True)
        a2, b2 = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        out = cls.apply(a1, b1)
        ref = torch.mm(a2, b2)
        assert not out.is_cuda and torch.equal(out, ref), cls.__name__
        out.backward(dc)
        ref.backward(dc)
        assert torch.allclose(fa(a2.grad), a1.grad, rtol=1e-5, atol=1e-8) and torch.allclose(fb(b2.grad), b1.grad, rtol=1e-5, atol=1e-8), cls.__name__
    # batched and matrix-vector forms follow torch.matmul
    x, w = torch.rand(3, 5, 16), torch.rand(16, 4)
    assert torch.equal(matmuls.cublasMM.apply(x, w), x @ w) and torch.equal(matmuls.naiveSpMM.apply(x, w), x @ w)
    assert torch.equal(matmuls.cublasMM.apply(x[0], w[:, 0]), x[0] @ w[:, 0])


def test_matmuls_product_path_fails_loudly_on_cpu(built):
    """Everything else on the host raises: a CSR operand in host memory has no kernel to go to (custom_mm takes device tensors
    only), and the entry points themselves refuse host tensors — nothing falls back to the oracle or to a CPU kernel of this
    package.  (A device operand paired with a host one raises too: tests/test_gpu_matmuls.py.)"""
    for k in ("custom_mm", "matmuls"):
        sys.modules.pop(k, None)
    import matmuls
    assert "fake" not in matmuls.custom_mm.__name__
    a, b = torch.rand(8, 64), torch.rand(64, 8)
    if not torch.cuda.is_available():
        for cls in (matmuls.naiveSpMM, matmuls.cusparseMM):
            with pytest.raises((RuntimeError, AssertionError)):
                cls.apply(a.to_sparse_csr(), b)
    c = torch.empty(8, 8)
    with pytest.raises(RuntimeError, match="device"):
        matmuls.custom_mm.cublas_mmul(a, b, c, False, False)
    with pytest.raises(RuntimeError, match="device"):
        matmuls.custom_mm.cublas_bmm(a[None], b[None], c[None], 3, False, False)


def test_gemm_split_rule_is_a_function_of_the_shape_and_matches_the_oracles_restatement(lib, oracle_mod):
    """mi_gemm_split_count (include/mi_spmm.h "Deterministic split-k") against the oracle's own restatement of the rule, on a
    grid of shapes: both sides must cut k alike or no parity check of a split product could be bit for bit."""
    import ctypes
    lib.mi_gemm_split_count.argtypes = [ctypes.c_int32] * 4
    lib.mi_gemm_workspace_bytes.argtypes = [ctypes.c_int32] * 4
    lib.mi_gemm_workspace_bytes.restype = ctypes.c_size_t
    seen = set()
    for m in (1, 64, 128, 129, 768, 1152, 2048, 4096):
        for n in (1, 64, 256, 768, 1024, 4097):
            for k in (0, 32, 4095, 4096, 4160, 5000, 8192, 16384, 65536, 100_000):
                for batch in (1, 2):
                    S = lib.mi_gemm_split_count(m, n, k, batch)
                    assert S == oracle_mod.gemm_split_count(m, n, k, batch), (m, n, k, batch)
                    assert S >= 1 and (S == 1 or (batch == 1 and k >= 4096 and k % (32 * S) == 0 and S & (S - 1) == 0))
                    assert lib.mi_gemm_workspace_bytes(m, n, k, batch) == (4 * S * m * n if S > 1 else 0)
                    seen.add(S)
    assert {1, 2, 4, 8, 64} <= seen
    assert lib.mi_gemm_split_count(768, 768, 16384, 1) == 8 and lib.mi_gemm_split_count(256, 256, 65536, 1) == 64
    assert lib.mi_gemm_split_count(4096, 1024, 16384, 1) == 1   # 256 tiles: the chip is busy as it is
