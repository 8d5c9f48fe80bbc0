"""Developer probe: effect of the long-row kernel on a skewed matrix (a few rows with 10^5-10^6 non-zeros)."""
import ctypes
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "matrix-multiplication_amd"))
import custom_mm  # noqa: E402
lib = ctypes.CDLL(str(Path(custom_mm.__file__).parent / "libmi_spmm.so"))
vp = ctypes.c_void_p
lib.mi_spmm_csr_f32.argtypes = [vp, vp, vp, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, vp,
                                ctypes.c_int64, vp, ctypes.c_int64, vp]
dev = torch.device("cuda")
M, K, N = 100_000, 1_000_000, 256
g = torch.Generator(device=dev).manual_seed(0)
lens = torch.full((M,), 100, dtype=torch.int64)
lens[7], lens[5000], lens[99_999] = 1_000_000, 100_000, 300_000
rowptr = torch.cat([torch.zeros(1, dtype=torch.int64), lens.cumsum(0)]).to(torch.int32).to(dev)
nnz = int(lens.sum())
col = torch.randint(0, K, (nnz,), device=dev, dtype=torch.int32, generator=g)
val = torch.rand(nnz, device=dev, generator=g)
B = torch.rand(K, N, device=dev, generator=g)
C1, C2 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)


def timeit(fn, iters=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


t_plain = timeit(lambda: lib.mi_spmm_csr_f32(rowptr.data_ptr(), col.data_ptr(), val.data_ptr(), nnz, M, K, N, B.data_ptr(), N,
                                             C1.data_ptr(), N, torch.cuda.current_stream().cuda_stream))
t_ws = timeit(lambda: custom_mm.naive_spmm(val, col, rowptr, nnz, M, K, B, C2))
print(f"nnz {nnz}: one wave per row {t_plain:.2f} ms; with the long-row kernel {t_ws:.2f} ms; "
      f"allclose {torch.allclose(C1, C2, rtol=1e-5, atol=1e-8)}")
for r in (7, 5000, 99_999, 3):
    s, e = int(rowptr[r]), int(rowptr[r + 1])
    ref = (val[s:e].double()[:, None] * B[col[s:e].long()].double()).sum(0)
    e1 = ((C1[r].double() - ref).abs() / ref.abs()).max().item()
    e2 = ((C2[r].double() - ref).abs() / ref.abs()).max().item()
    print(f"row {r} ({e - s} nnz): max rel err vs fp64 — single chain {e1:.2e}, 16·S chains {e2:.2e}")
