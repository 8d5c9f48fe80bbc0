"""Developer probe: the backward-side kernels of the CSR path at the C3 shape (1M x 1M CSR at 0.01 %, N = 256):
forward SpMM, SDDMM (gradient of A's values), CSR transpose, and the transposed SpMM (gradient of B)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "matrix-multiplication_amd"))
import custom_mm  # noqa: E402
import synthetic  # noqa: E402
dev = torch.device("cuda")
M = K = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
density = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-4
N = 256
rowptr, col, val = (torch.from_numpy(x).to(dev) for x in synthetic.make_csr(M, K, density, seed=0))
nnz = val.numel()
B = torch.from_numpy(synthetic.make_dense(K, N, seed=1)).to(dev)
dC = torch.rand(M, N, device=dev)
C = torch.empty(M, N, device=dev)


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


gb = nnz * (4 * N + 8) / 1e9
t = timeit(lambda: custom_mm.naive_spmm(val, col, rowptr, nnz, M, K, B, C))
print(f"forward spmm        {t:8.3f} ms  ({(gb + 4e-9 * M * N) / t * 1e3:7.0f} GB/s algorithmic)")
t = timeit(lambda: custom_mm.sddmm(col, rowptr, nnz, M, K, dC, B))
print(f"sddmm (grad values) {t:8.3f} ms  ({(gb + 4e-9 * M * N) / t * 1e3:7.0f} GB/s: B-row gather + dC read + 4 B out per non-zero)")
t = timeit(lambda: custom_mm.csr_transpose(val, col, rowptr, nnz, M, K), iters=5)
print(f"csr_transpose       {t:8.3f} ms  ({nnz * 16 / t / 1e6:7.0f} GB/s of the 16 B per non-zero a transpose must move)")
tv, tc, to = custom_mm.csr_transpose(val, col, rowptr, nnz, M, K)
t = timeit(lambda: custom_mm.naive_spmm(tv, tc, to.view(-1), nnz, K, M, dC, C))
print(f"At x dC (grad B)    {t:8.3f} ms")
