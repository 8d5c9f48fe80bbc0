"""A stand-in for the `custom_mm` extension built on the CPU oracle — TEST ONLY.

Lets the `-m "not gpu"` tests exercise matmuls.py's host logic (rank dispatch,
output allocation, autograd formulas, broadcasting) on CPU tensors: the test
puts this module into sys.modules['custom_mm'] before importing matmuls, so the
default `mm_op` / `bmm_op` arguments bind to these functions.  Same names and
positional signatures as the real module (reference src/custom_mm.cpp:393-416).
"""
import numpy as np
import torch

import oracle

calls = []  # (name, extra) per kernel invocation, for dispatch assertions


def _np(t):
    return t.detach().cpu().numpy()


def _write(C, arr):
    C.copy_(torch.from_numpy(np.ascontiguousarray(arr)).reshape(C.shape))
    return C


def cublas_mmul(A, B, C, transa, transb):
    calls.append(("cublas_mmul", (transa, transb)))
    return _write(C, oracle.gemm(_np(A), _np(B), transa, transb))


def cublas_bmm(A, B, C, dim, transa, transb):
    calls.append(("cublas_bmm", (dim, transa, transb)))
    if dim not in (2, 3, 4):
        raise ValueError("Invalid dim argument.")
    assert A.dim() == dim and B.dim() == dim
    return _write(C, oracle.gemm(np.ascontiguousarray(_np(A)), np.ascontiguousarray(_np(B)), transa, transb))


def _spmm(name, vals, cols, offs, nnz, rows, kcols, B, C):
    calls.append((name, (rows, kcols)))
    assert offs.dtype == torch.int32 and cols.dtype == torch.int32 and vals.dtype == torch.float32
    assert offs.numel() == rows + 1 and B.shape[0] == kcols
    # `nnz` may be a bound (or, without a long-row workspace, an estimate): include/mi_spmm.h — the rows are walked through offs
    true_nnz = int(offs[-1]) if offs.numel() else 0
    assert vals.numel() >= max(nnz, true_nnz) and cols.numel() >= max(nnz, true_nnz)
    return _write(C, oracle.spmm_csr(_np(offs), _np(cols)[:true_nnz], _np(vals)[:true_nnz], rows, kcols, _np(B)))


def naive_spmm(vals, cols, offs, nnz, rows, kcols, B, C):
    return _spmm("naive_spmm", vals, cols, offs, nnz, rows, kcols, B, C)


def naive_spmm_ex(vals, cols, offs, nnz, rows, kcols, B, C, long_rows):
    return _spmm("naive_spmm_ex", vals, cols, offs, nnz, rows, kcols, B, C)


def long_row_threshold():
    return 8192


def cusparse_mmul(vals, cols, offs, nnz, rows, kcols, B, C):
    return _spmm("cusparse_mmul", vals, cols, offs, nnz, rows, kcols, B, C)


def dense_row_offsets(dense):
    calls.append(("dense_row_offsets", tuple(dense.shape)))
    rp, _, _ = oracle.dense_to_csr(_np(dense))
    return torch.from_numpy(rp.copy())


def dense_to_csr_fill(dense, offsets, nnz):
    calls.append(("dense_to_csr_fill", tuple(dense.shape)))
    rp, c, v = oracle.dense_to_csr(_np(dense))
    assert np.array_equal(rp, offsets.numpy()) and len(v) <= nnz  # nnz: the arrays' capacity (≥ the true count)
    vals, cols = np.full(nnz, np.nan, np.float32), np.full(nnz, -1, np.int32)
    vals[:len(v)], cols[:len(v)] = v, c
    return torch.from_numpy(vals), torch.from_numpy(cols)


def dense_to_csr(dense):
    calls.append(("dense_to_csr", tuple(dense.shape)))
    rp, c, v = oracle.dense_to_csr(_np(dense))
    return torch.from_numpy(v.copy()), torch.from_numpy(c.copy()), torch.from_numpy(rp.copy())


def naive_spmm_batched(vals, cols, offs, nnz, batch, rows, kcols, B, C):
    calls.append(("naive_spmm_batched", (batch, rows, kcols)))
    return _write(C, oracle.spmm_csr_batched(_np(offs), _np(cols)[:nnz], _np(vals)[:nnz], batch, rows, kcols,
                                             np.ascontiguousarray(_np(B))))


perm_plan = True  # tests flip this: the plan for a problem may or may not take a permutation


def naive_spmm_batched_perm(vals, perm, cols, offs, nnz, batch, rows, kcols, B, C):
    calls.append(("naive_spmm_batched_perm", (batch, rows, kcols)))
    if not perm_plan:
        return False
    gathered = _np(vals)[_np(perm)[:nnz].astype(np.int64)]
    _write(C, oracle.spmm_csr_batched(_np(offs), _np(cols)[:nnz], gathered, batch, rows, kcols,
                                      np.ascontiguousarray(_np(B))))
    return True


def csr_transpose(vals, cols, offs, nnz, rows, kcols):
    calls.append(("csr_transpose", (rows, kcols)))
    rp, c, v = oracle.csr_transpose(_np(offs), _np(cols)[:nnz], _np(vals)[:nnz], rows, kcols)
    return torch.from_numpy(v.copy()), torch.from_numpy(c.copy()), torch.from_numpy(rp.copy())


def csr_transpose_batched(vals, cols, offs, nnz, batch, rows, kcols):
    """offs: [batch, rows + 1] with every item's base included; returns offsets [batch, kcols + 1] in the same form."""
    calls.append(("csr_transpose_batched", (batch, rows, kcols)))
    o, c, v = _np(offs).reshape(batch, rows + 1), _np(cols)[:nnz], _np(vals)[:nnz]
    t_val, t_col = np.empty(nnz, v.dtype), np.empty(nnz, np.int32)
    t_off = np.empty((batch, kcols + 1), np.int32)
    for b in range(batch):
        s0, s1 = int(o[b, 0]), int(o[b, rows])
        rp, cc, vv = oracle.csr_transpose((o[b] - s0).astype(np.int32), c[s0:s1], v[s0:s1], rows, kcols)
        t_val[s0:s1], t_col[s0:s1], t_off[b] = vv, cc, rp + s0
    return torch.from_numpy(t_val), torch.from_numpy(t_col), torch.from_numpy(t_off)


def sddmm(cols, offs, nnz, rows, kcols, dC, B):
    calls.append(("sddmm", (rows, kcols)))
    return torch.from_numpy(oracle.sddmm(_np(offs), _np(cols)[:nnz], rows, _np(dC), _np(B)).copy())


batched_sddmm = True  # tests flip this: the LDS-resident batched form may or may not take a problem


def sddmm_batched(cols, offs, nnz, batch, rows, kcols, dC, B, out):
    calls.append(("sddmm_batched", (batch, rows, kcols)))
    if not batched_sddmm:
        return False
    o, c = _np(offs).reshape(batch, rows + 1), _np(cols)[:nnz]
    res = np.empty(nnz, np.float32)
    for b in range(batch):
        s0, s1 = int(o[b, 0]), int(o[b, rows])
        Bb = _np(B) if B.dim() == 2 else _np(B)[b]
        res[s0:s1] = oracle.sddmm((o[b] - s0).astype(np.int32), c[s0:s1], rows, _np(dC)[b], Bb)
    out[:nnz] = torch.from_numpy(res)
    return True


fused_dense = True  # tests flip this to exercise the CSR route as well


def naive_spmm_dense(A, B, C):
    calls.append(("naive_spmm_dense", tuple(A.shape)))
    if not fused_dense or C.shape[-1] % 4 != 0:
        return False
    a = _np(A)
    rows, kcols = a.shape[-2:]
    batch = int(np.prod(a.shape[:-2])) if a.ndim > 2 else 1
    rp, c, v = oracle.dense_to_csr(a)
    _write(C, oracle.spmm_csr_batched(rp, c, v, batch, rows, kcols, np.ascontiguousarray(_np(B))))
    return True


def cublas_mmul_bias(A, B, bias, C, transa, transb):
    calls.append(("cublas_mmul_bias", (transa, transb)))
    return _write(C, oracle.gemm(_np(A), _np(B), transa, transb) + _np(bias)[None, :])


def naive_spmm_bias(vals, cols, offs, nnz, rows, kcols, B, bias, C):
    calls.append(("naive_spmm_bias", (rows, kcols)))
    n = int(offs[-1]) if offs.numel() else 0  # (`nnz` may be a bound or an estimate: see _spmm)
    return _write(C, oracle.spmm_csr(_np(offs), _np(cols)[:n], _np(vals)[:n], rows, kcols, _np(B)) + _np(bias)[None, :])


def naive_spmm_bias_ex(vals, cols, offs, nnz, rows, kcols, B, bias, C, long_rows):
    calls.append(("naive_spmm_bias_ex", (rows, kcols)))
    _spmm_into = naive_spmm_bias(vals, cols, offs, nnz, rows, kcols, B, bias, C)
    calls.pop()   # the inner naive_spmm_bias record
    return _spmm_into


def column_sums(src):
    calls.append(("column_sums", tuple(src.shape)))
    return torch.from_numpy(_np(src).astype(np.float64).sum(axis=0).astype(np.float32))


def init_cublas():
    pass


def destroy_cublas():
    pass


def init_cusparse():
    pass


def destroy_cusparse():
    pass
