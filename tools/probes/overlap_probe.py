"""Developer probe: do the two gradient products of one attention matmul run faster side by side (two streams) than
one after the other?  BERT-base shapes (384 items, S = 512, D = 64)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "matrix-multiplication_amd"))
import custom_mm  # noqa: E402
dev = torch.device("cuda")
items, S, D = 384, 512, 64
q, k, v, dc = (torch.rand(items, S, D, device=dev) for _ in range(4))
p, ds = torch.rand(items, S, S, device=dev), torch.rand(items, S, S, device=dev)
dp = torch.empty(items, S, S, device=dev)
dv, dq, dk = (torch.empty(items, S, D, device=dev) for _ in range(3))
side = torch.cuda.Stream()


def pair(fa, fb, concurrent):
    if not concurrent:
        fa(); fb()
        return
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        fb()
    fa()
    main.wait_stream(side)


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best


pairs = {
    "bwd of probs.V : dP = dC.VT (writes 403 MB) | dV = PT.dC (reads 403 MB)":
        (lambda: custom_mm.cublas_bmm(dc, v, dp, 3, False, True), lambda: custom_mm.cublas_bmm(p, dc, dv, 3, True, False)),
    "bwd of q.kT    : dQ = dS.K (reads dS)       | dK = dST.Q (reads dS)":
        (lambda: custom_mm.cublas_bmm(ds, k, dq, 3, False, False), lambda: custom_mm.cublas_bmm(ds, q, dk, 3, True, False)),
}
for name, (fa, fb) in pairs.items():
    ta, tb = timeit(fa), timeit(fb)
    seq = timeit(lambda: pair(fa, fb, False))
    con = timeit(lambda: pair(fa, fb, True))
    print(f"{name}\n   alone {ta:.4f} + {tb:.4f} = {ta + tb:.4f} ms   back to back {seq:.4f}   two streams {con:.4f}", flush=True)
