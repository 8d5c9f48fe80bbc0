"""Developer probe: square fp32 products through custom_mm.cublas_mmul and torch.matmul (rocBLAS), timed
with events; also the target of rocprofv3 --pmc runs.   python tools/gemm_square_probe.py 1024 4096"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "matrix-multiplication_amd"))
import custom_mm  # noqa: E402
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)


def timeit(fn, iters=20):
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


for d in [int(x) for x in sys.argv[1:]] or [1024, 2048, 4096, 8192]:
    a = torch.rand(d, d, device=dev, generator=g)
    b = torch.rand(d, d, device=dev, generator=g)
    c = torch.empty(d, d, device=dev)
    fl = 2.0 * d ** 3
    for ta, tb in ((False, False), (False, True), (True, False)):
        t = timeit(lambda: custom_mm.cublas_mmul(a, b, c, ta, tb))
        aa = a.t() if ta else a
        bb = b.t() if tb else b
        t2 = timeit(lambda: torch.matmul(aa, bb, out=c))
        print(f"{d:6d} ta={int(ta)} tb={int(tb)}  ours {t:8.4f} ms {fl / t / 1e9:7.1f} TFLOP/s   rocBLAS {t2:8.4f} ms {fl / t2 / 1e9:7.1f} TFLOP/s",
              flush=True)
