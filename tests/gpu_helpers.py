"""Shared helpers of the GPU parity modules (tests/test_gpu_*.py): input builders, reference expressions, C-ABI shims.
Fixtures (dev, cmm, mm, capi) live in conftest.py."""
import ctypes
import sys

import numpy as np
import pytest
import torch

RTOL, ATOL = 1e-5, 1e-8

__all__ = ['RTOL', 'ATOL', 't', 'torch_cpu_csr_matmul', 'assert_matches_reference_expression', 'run_spmm', '_random_rows_csr', '_transpose_through_the_c_abi', 'gemm_ref', '_dense_of', 'fwd_bwd_device', '_panel_sorted', '_sub_csr', '_moderately_dense_with_hub_rows']


def t(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def torch_cpu_csr_matmul(rowptr, col, val, M, K, B):
    """The reference's own CPU expression `a @ b` (reference matmuls.py:41,71,210,234,279,302) with A as a torch
    CSR tensor, evaluated by torch-CPU: the expectation of the reference's tests (tests/naive_kernel_test.py:30)."""
    a_csr = torch.sparse_csr_tensor(torch.from_numpy(rowptr.astype(np.int64)), torch.from_numpy(col.astype(np.int64)),
                                    torch.from_numpy(val), (M, K))
    return (a_csr @ torch.from_numpy(B)).numpy()


def assert_matches_reference_expression(got, ref):
    """tests/naive_kernel_test.py:36-37: shape equality + torch.allclose at its defaults, on the FULL output."""
    assert got.shape == ref.shape
    assert torch.allclose(torch.from_numpy(got), torch.from_numpy(ref), rtol=RTOL, atol=ATOL), \
        f"max rel err vs torch-CPU A_csr @ B: {np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-30))}"


def run_spmm(cmm, dev, rowptr, col, val, M, K, B, op="naive_spmm"):
    C = torch.full((M, B.shape[1]), float("nan"), device=dev)
    out = getattr(cmm, op)(t(val, dev), t(col, dev), t(rowptr, dev), len(val), M, K, t(B, dev), C)
    assert out is C or out.data_ptr() == C.data_ptr()  # same tensor returned (reference custom_mm.cpp:178,216)
    return C.cpu().numpy()


def _random_rows_csr(M, K, lens, seed, shuffle=0.0, duplicates=False):
    g = np.random.Generator(np.random.PCG64(seed))
    cols = []
    lens = np.asarray(lens)
    for r in np.nonzero(lens)[0]:
        n = int(lens[r])
        c = g.integers(0, K, size=n) if (duplicates or n > K) else g.choice(K, size=n, replace=False)
        c = np.sort(c)
        if shuffle and g.random() < shuffle:
            c = g.permutation(c)
        cols.append(c.astype(np.int32))
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    col = np.concatenate(cols) if cols else np.zeros(0, np.int32)
    return rowptr, col, (g.random(len(col), dtype=np.float32) - 0.5)


def _transpose_through_the_c_abi(capi, dev, rowptr, col, val, M, K, plan):
    """mi_csr_transpose_f32 with the plan pinned (1 tables, 2 one-sweep); returns (t_rowptr, t_col, t_val) on the host and
    checks the one-sweep plan's give-up flag."""
    vp, i64, i32, sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_size_t
    capi.mi_csr_transpose_workspace_bytes.restype = sz
    capi.mi_csr_transpose_workspace_bytes.argtypes = [i32, i32, i64]
    capi.mi_csr_transpose_f32.argtypes = [vp, vp, vp, i64, i32, i32, vp, vp, vp, vp, sz, vp]
    capi.mi_csr_transpose_check.argtypes = [vp, sz, i32, i32, i32, i64, vp]
    nnz = len(val)
    d_rp, d_col, d_val = t(rowptr, dev), t(col, dev), t(val, dev)
    ws_bytes = capi.mi_csr_transpose_workspace_bytes(M, K, nnz)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    t_rp = torch.full((K + 1,), -1, dtype=torch.int32, device=dev)
    t_col = torch.full((nnz,), -1, dtype=torch.int32, device=dev)
    t_val = torch.full((nnz,), -1.0, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    assert capi.mi_csr_transpose_set_plan(plan) == 0
    try:
        st = capi.mi_csr_transpose_f32(d_rp.data_ptr(), d_col.data_ptr(), d_val.data_ptr(), nnz, M, K, t_rp.data_ptr(),
                                       t_col.data_ptr(), t_val.data_ptr(), ws.data_ptr(), ws_bytes, stream)
    finally:
        capi.mi_csr_transpose_set_plan(0)
    assert st == 0, st
    assert capi.mi_csr_transpose_check(ws.data_ptr(), ws_bytes, 1, M, K, nnz, stream) == 0, "a look-back poll gave up"
    return t_rp.cpu().numpy(), t_col.cpu().numpy(), t_val.cpu().numpy()


def gemm_ref(oracle_mod, a, b, ta, tb):
    return oracle_mod.gemm(a, b, ta, tb)


def _dense_of(rowptr, col, val, M, K):
    A = np.zeros((M, K), np.float64)
    rows = np.repeat(np.arange(M), np.diff(rowptr))
    np.add.at(A, (rows, col), val)
    return A


def fwd_bwd_device(fn, ref_fn, a, b, dev):
    a1, b1 = a.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    a2, b2 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    out, exp = fn(a1, b1), ref_fn(a2, b2)
    assert out.is_cuda and out.shape == exp.shape
    assert torch.allclose(exp, out.cpu(), rtol=RTOL, atol=ATOL)
    dc = torch.rand(exp.shape, generator=torch.Generator().manual_seed(3))
    out.backward(dc.to(dev))
    exp.backward(dc)
    assert torch.allclose(a2.grad, a1.grad.cpu(), rtol=RTOL, atol=ATOL)
    assert torch.allclose(b2.grad, b1.grad.cpu(), rtol=RTOL, atol=ATOL)


def _panel_sorted(rowptr, col, val, split):
    """Entries of each row reordered the way two unchecked panel passes would consume them."""
    c2, v2 = col.copy(), val.copy()
    for r in range(len(rowptr) - 1):
        s0, e0 = rowptr[r], rowptr[r + 1]
        order = np.argsort(col[s0:e0] >= split, kind="stable")
        c2[s0:e0], v2[s0:e0] = col[s0:e0][order], val[s0:e0][order]
    return c2, v2


def _sub_csr(rowptr, col, val, rows):
    """CSR of the selected rows (rows of a product are independent: the oracle on this equals the
    oracle on the whole matrix restricted to these rows)."""
    lens = [int(rowptr[r + 1] - rowptr[r]) for r in rows]
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    idx = np.concatenate([np.arange(rowptr[r], rowptr[r + 1]) for r in rows]) if rows else np.zeros(0, np.int64)
    return rp, col[idx], val[idx]


def _moderately_dense_with_hub_rows(M, K, density, hubs, seed):
    g = np.random.Generator(np.random.PCG64(seed))
    mask = g.random((M, K), dtype=np.float32) < density
    mask[hubs] = True
    rows, col = np.nonzero(mask)
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=M))]).astype(np.int32)
    return rowptr, col.astype(np.int32), g.random(len(col), dtype=np.float32) - 0.5
