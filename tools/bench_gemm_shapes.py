"""Developer probe: the exact-fp32 product (custom_mm.cublas_mmul / cublas_bmm) beside torch (rocBLAS) over a set of shapes
and transposition flags — looks for shapes where the tile choice leaves the chip idle."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "matrix-multiplication_amd"))
import custom_mm  # noqa: E402
dev = torch.device("cuda")


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


g = torch.Generator(device=dev).manual_seed(0)
SHAPES = [(16384, 3072, 768, 0, 1), (16384, 768, 3072, 0, 1), (16384, 768, 3072, 0, 0), (3072, 768, 16384, 1, 0), (768, 3072, 16384, 1, 0),
          (4096, 4096, 4096, 0, 0), (4096, 4096, 4096, 0, 1), (4096, 4096, 4096, 1, 0), (1024, 1024, 1024, 0, 1), (2048, 512, 8192, 1, 0),
          (768, 768, 16384, 1, 0), (1536, 768, 8192, 1, 0), (1152, 1152, 8192, 1, 0), (256, 256, 65536, 1, 0), (8192, 128, 8192, 0, 0),
          (128, 8192, 8192, 0, 0), (16384, 64, 768, 0, 1), (1000, 1000, 1000, 0, 0), (4096, 1024, 16384, 1, 0), (1024, 4096, 16384, 1, 0), (2048, 2048, 8192, 1, 0), (1024, 1024, 8192, 1, 0), (16384, 256, 3072, 0, 1), (2048, 2048, 256, 0, 1)]
for (m, n, k, ta, tb) in SHAPES:
    a = torch.rand((k, m) if ta else (m, k), device=dev, generator=g)
    b = torch.rand((n, k) if tb else (k, n), device=dev, generator=g)
    c = torch.empty(m, n, device=dev)
    t = timeit(lambda: custom_mm.cublas_mmul(a, b, c, bool(ta), bool(tb)))
    aa, bb = (a.t() if ta else a), (b.t() if tb else b)
    tt = timeit(lambda: torch.matmul(aa, bb, out=c))
    tf = 2.0 * m * n * k / t / 1e9
    print(f"m {m:6d} n {n:6d} k {k:6d} {'T' if ta else 'N'}{'T' if tb else 'N'}: ours {t:.3f} ms ({tf:6.1f} TFLOP/s)  torch {tt:.3f} ms  ratio {t / tt:.2f}", flush=True)
