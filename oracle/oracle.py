"""ctypes/numpy front end of oracle/spmm_oracle.c (test infrastructure).

Every function takes and returns numpy arrays (float32 / int32, C-contiguous)
and runs on the CPU.  `build()` compiles the C file with gcc via the Makefile.
"""
from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

import numpy as np

__all__ = ["build", "lib", "gemm_split_count", "spmm_csr", "spmm_csr_omp", "spmm_csr_long", "spmm_csr_chain", "spmm_csr_batched", "spmm_csr_colmajor", "gemm",
           "coo_to_csr", "dense_to_csr", "csr_transpose", "sddmm", "make_csr"]

_DIR = Path(__file__).resolve().parent
_SO = _DIR / "_build" / "liboracle.so"
_lib = None

_i32 = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f32 = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_c32, _c64 = ctypes.c_int32, ctypes.c_int64


def build(force: bool = False) -> Path:
    src = _DIR / "spmm_oracle.c"
    if force or not _SO.exists() or _SO.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_DIR), "-s"] + (["-B"] if force else []), check=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(str(_SO))
        spmm_args = [_i32, _i32, _f32, _c32, _c32, _c32, _f32, _c64, _f32, _c64]
        for name in ("oracle_spmm_csr_f32", "oracle_spmm_csr_f32_omp", "oracle_spmm_csr_colmajor_f32",
                     "oracle_spmm_csr_long_f32", "oracle_spmm_csr_chain_f32"):
            getattr(L, name).argtypes = spmm_args
            getattr(L, name).restype = None
        L.oracle_spmm_csr_batched_f32.argtypes = [_i32, _i32, _f32, _c32, _c32, _c32, _c32, _f32, _c64, _c64,
                                                  _f32, _c64, _c64]
        L.oracle_spmm_csr_batched_f32.restype = None
        L.oracle_gemm_f32.argtypes = [ctypes.c_int, ctypes.c_int, _c32, _c32, _c32, _f32, _c64, _c64, _f32, _c64,
                                      _c64, _f32, _c64, _c64, _c32]
        L.oracle_gemm_f32.restype = None
        L.oracle_gemm_ex_f32.argtypes = list(L.oracle_gemm_f32.argtypes) + [ctypes.c_int]
        L.oracle_gemm_ex_f32.restype = None
        L.oracle_gemm_split_count.argtypes = [_c32, _c32, _c32, _c32]
        L.oracle_gemm_split_count.restype = _c32
        L.oracle_coo_to_csr.argtypes = [_c32, _c64, _i32, _i32, _f32, _i32, _i32, _f32]
        L.oracle_coo_to_csr.restype = ctypes.c_int
        L.oracle_dense_to_csr.argtypes = [_f32, _c32, _c32, _c32, _c64, _c64, _i32, ctypes.c_void_p, ctypes.c_void_p]
        L.oracle_dense_to_csr.restype = _c64
        L.oracle_csr_transpose.argtypes = [_i32, _i32, _f32, _c32, _c32, _i32, _i32, _f32]
        L.oracle_csr_transpose.restype = None
        L.oracle_sddmm_csr_f32.argtypes = [_i32, _i32, _c32, _c32, _f32, _c64, _f32, _c64, _f32]
        L.oracle_sddmm_csr_f32.restype = None
        _lib = L
    return _lib


def _f(x):
    return np.ascontiguousarray(x, dtype=np.float32)


def _i(x):
    return np.ascontiguousarray(x, dtype=np.int32)


def _pad1(x, dtype):
    """ctypes ndpointer rejects nothing about size, but keep empty arrays addressable."""
    x = np.ascontiguousarray(x, dtype=dtype)
    return x if x.size else np.zeros(1, dtype=dtype)


def spmm_csr(rowptr, col, val, M, K, B, omp=False):
    """C[M,N] = A_csr · B (row-major), fused multiply-add in CSR order per element."""
    B = _f(B)
    N = B.shape[1]
    C = np.empty((M, N), dtype=np.float32)
    fn = lib().oracle_spmm_csr_f32_omp if omp else lib().oracle_spmm_csr_f32
    fn(_i(rowptr), _pad1(col, np.int32), _pad1(val, np.float32), M, K, N, _pad1(B, np.float32), max(N, 1),
       C if C.size else np.zeros(1, np.float32), max(N, 1))
    return C


def spmm_csr_omp(rowptr, col, val, M, K, B):
    return spmm_csr(rowptr, col, val, M, K, B, omp=True)


def spmm_csr_chain(rowptr, col, val, M, K, B):
    """The pure CSR-order fmaf chain for every N (no narrow-N rule): what the explicit group-kernel
    variants compute when forced onto N < 4."""
    B = _f(B)
    N = B.shape[1]
    C = np.empty((M, N), dtype=np.float32)
    lib().oracle_spmm_csr_chain_f32(_i(rowptr), _pad1(col, np.int32), _pad1(val, np.float32), M, K, N,
                                    _pad1(B, np.float32), max(N, 1), C if C.size else np.zeros(1, np.float32), max(N, 1))
    return C


def spmm_csr_long(rowptr, col, val, M, K, B):
    """As spmm_csr, with the long-row rule of mi_spmm_csr_ws_f32 (rows > 8192 non-zeros: 16
    interleaved chains added in order) — what custom_mm.naive_spmm / cusparse_mmul compute."""
    B = _f(B)
    N = B.shape[1]
    C = np.empty((M, N), dtype=np.float32)
    lib().oracle_spmm_csr_long_f32(_i(rowptr), _pad1(col, np.int32), _pad1(val, np.float32), M, K, N,
                                   _pad1(B, np.float32), max(N, 1), C if C.size else np.zeros(1, np.float32), max(N, 1))
    return C


def spmm_csr_batched(rowptr, col, val, batch, M, K, B):
    """rowptr [batch, M+1] with global offsets; B [batch, K, N] or [K, N] (broadcast)."""
    B = _f(B)
    N = B.shape[-1]
    strideB = K * N if B.ndim == 3 else 0
    C = np.empty((batch, M, N), dtype=np.float32)
    lib().oracle_spmm_csr_batched_f32(_i(rowptr).reshape(-1), _pad1(col, np.int32), _pad1(val, np.float32), batch, M,
                                      K, N, _pad1(B, np.float32), max(N, 1), strideB,
                                      C if C.size else np.zeros(1, np.float32), max(N, 1), M * N)
    return C


def spmm_csr_colmajor(rowptr, col, val, M, K, N, B_colmajor):
    """B given as the flat column-major K×N buffer (== row-major [N,K]); returns the flat
    column-major M×N buffer (== row-major [N,M])."""
    Bf = _f(B_colmajor).reshape(-1)
    C = np.empty(M * N, dtype=np.float32)
    lib().oracle_spmm_csr_colmajor_f32(_i(rowptr), _pad1(col, np.int32), _pad1(val, np.float32), M, K, N,
                                       _pad1(Bf, np.float32), max(K, 1), C if C.size else np.zeros(1, np.float32),
                                       max(M, 1))
    return C


def gemm_split_count(m, n, k, batch=1):
    """The product path's split rule (include/mi_spmm.h, "Deterministic split-k"), restated: 1 = one chain per element."""
    return int(lib().oracle_gemm_split_count(int(m), int(n), int(k), int(batch)))


def gemm(A, B, transa=False, transb=False, split=True):
    """C = op(A)·op(B) over the last two dims; leading dims (equal on both) are the batch.  split=True: the order of
    custom_mm.cublas_mmul / cublas_bmm (few output tiles and k ≥ 4096: S equal k-ranges added in index order); split=False: one
    k-ordered fmaf chain per element whatever the shape (the raw C-ABI entries mi_gemm_f32 / mi_gemm_bias_f32)."""
    A, B = _f(A), _f(B)
    batch_shape = A.shape[:-2]
    assert batch_shape == B.shape[:-2]
    batch = int(np.prod(batch_shape)) if batch_shape else 1
    ar, ac = A.shape[-2:]
    br, bc = B.shape[-2:]
    m, k = (ac, ar) if transa else (ar, ac)
    k2, n = (bc, br) if transb else (br, bc)
    assert k == k2, (A.shape, B.shape, transa, transb)
    C = np.empty(batch_shape + (m, n), dtype=np.float32)
    lib().oracle_gemm_ex_f32(int(transa), int(transb), m, n, k, _pad1(A, np.float32), max(ac, 1), ar * ac,
                             _pad1(B, np.float32), max(bc, 1), br * bc, C if C.size else np.zeros(1, np.float32),
                             max(n, 1), m * n, batch, 1 if split else 0)
    return C


def coo_to_csr(M, coo_row, coo_col, coo_val):
    nnz = len(coo_val)
    rowptr = np.zeros(M + 1, np.int32)
    col = np.zeros(max(nnz, 1), np.int32)
    val = np.zeros(max(nnz, 1), np.float32)
    st = lib().oracle_coo_to_csr(M, nnz, _pad1(coo_row, np.int32), _pad1(coo_col, np.int32),
                                 _pad1(coo_val, np.float32), rowptr, col, val)
    if st != 0:
        raise ValueError(f"oracle_coo_to_csr status {st}")
    return rowptr, col[:nnz], val[:nnz]


def dense_to_csr(dense):
    """dense [..., rows, cols] → (rowptr [batch, rows+1] global offsets, col, val)."""
    d = _f(dense)
    rows, cols = d.shape[-2:]
    batch = int(np.prod(d.shape[:-2])) if d.ndim > 2 else 1
    rowptr = np.zeros((batch, rows + 1), np.int32)
    flat = _pad1(d, np.float32)
    nnz = lib().oracle_dense_to_csr(flat, batch, rows, cols, max(cols, 1), rows * cols, rowptr.reshape(-1), None,
                                    None)
    col = np.zeros(max(nnz, 1), np.int32)
    val = np.zeros(max(nnz, 1), np.float32)
    lib().oracle_dense_to_csr(flat, batch, rows, cols, max(cols, 1), rows * cols, rowptr.reshape(-1),
                              col.ctypes.data_as(ctypes.c_void_p), val.ctypes.data_as(ctypes.c_void_p))
    return rowptr, col[:nnz], val[:nnz]


def csr_transpose(rowptr, col, val, M, K):
    nnz = len(val)
    t_rowptr = np.zeros(K + 1, np.int32)
    t_col = np.zeros(max(nnz, 1), np.int32)
    t_val = np.zeros(max(nnz, 1), np.float32)
    lib().oracle_csr_transpose(_i(rowptr), _pad1(col, np.int32), _pad1(val, np.float32), M, K, t_rowptr, t_col, t_val)
    return t_rowptr, t_col[:nnz], t_val[:nnz]


def sddmm(rowptr, col, M, dC, B):
    dC, B = _f(dC), _f(B)
    N = B.shape[1]
    out = np.zeros(max(len(col), 1), np.float32)
    lib().oracle_sddmm_csr_f32(_i(rowptr), _pad1(col, np.int32), M, N, _pad1(dC, np.float32), max(N, 1),
                               _pad1(B, np.float32), max(N, 1), out)
    return out[:len(col)]


def make_csr(M, K, density, seed):
    """The pinned synthetic generator of SURVEY.md §8(d): numpy PCG64, unique uniform
    keys in [0, M*K), values U[0,1) float32.  Returns (rowptr i32, col i32, val f32).
    (Test-side twin of matrix-multiplication_amd/synthetic.py: tests/test_oracle.py checks the two agree.)"""
    rng = np.random.Generator(np.random.PCG64(seed))
    keys = np.unique(rng.integers(0, M * K, size=round(M * K * density), dtype=np.int64))
    row = keys // K
    col = (keys % K).astype(np.int32)
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(row, minlength=M))]).astype(np.int32)
    val = rng.random(len(keys), dtype=np.float32)
    return rowptr, col, val
