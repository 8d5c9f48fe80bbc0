'''
synthetic — the pinned synthetic workload generator.

numpy only; used by bench.py and the benchmarks to build the CSR A and dense B of the
BASELINE.json configurations byte-for-byte reproducibly:

    rngA = Generator(PCG64(seedA)); keys = unique(rngA.integers(0, M*K, size=round(M*K*density)))
    row = keys // K; col = keys % K; rowptr = [0, cumsum(bincount(row, minlength=M))]
    val = rngA.random(nnz, float32)                       # U[0,1), like the reference tests' torch.rand
    B   = Generator(PCG64(seedB)).random((K, N), float32)

Uniform-random pattern, columns sorted and unique within a row.
'''
import numpy as np


def make_csr(M: int, K: int, density: float, seed: int = 0):
    '''(rowptr int32[M+1], col int32[nnz], val float32[nnz]).'''
    rng = np.random.Generator(np.random.PCG64(seed))
    keys = np.unique(rng.integers(0, M * K, size=round(M * K * density), dtype=np.int64))
    row = keys // K
    col = (keys % K).astype(np.int32)
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(row, minlength=M))]).astype(np.int32)
    val = rng.random(len(keys), dtype=np.float32)
    return rowptr, col, val


def make_dense(K: int, N: int, seed: int = 1):
    return np.random.Generator(np.random.PCG64(seed)).random((K, N), dtype=np.float32)
