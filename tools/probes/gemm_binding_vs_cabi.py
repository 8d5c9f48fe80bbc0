"""The same dense product through custom_mm.cublas_bmm (pybind) and through the C-ABI (ctypes), taking turns: do they cost the same?"""
import ctypes
import sys
from pathlib import Path
import numpy as np
import torch
root = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(root / "matrix-multiplication_amd"))
import custom_mm  # noqa: E402
lib = ctypes.CDLL(str(root / "matrix-multiplication_amd" / "libmi_spmm.so"))
vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
lib.mi_gemm_f32.argtypes = [ctypes.c_int, ctypes.c_int, i32, i32, i32, vp, i64, i64, vp, i64, i64, vp, i64, i64, i32, vp]
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
items, S, D = (int(x) for x in sys.argv[1:4])
q, kk = torch.rand(items, S, D, device=dev), torch.rand(items, S, D, device=dev)
sc, sc2 = torch.empty(items, S, S, device=dev), torch.empty(items, S, S, device=dev)
entries = {
    "pybind cublas_bmm": lambda: custom_mm.cublas_bmm(q, kk, sc, 3, False, True),
    "C-ABI mi_gemm_f32": lambda: lib.mi_gemm_f32(0, 1, S, S, D, q.data_ptr(), D, S * D, kk.data_ptr(), D, S * D, sc2.data_ptr(), S, S * S, items, st),
    "torch.matmul": lambda: torch.matmul(q, kk.transpose(-1, -2), out=sc),
}


def block(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for fn in entries.values():
    for _ in range(3):
        fn()
torch.cuda.synchronize()
print("same bits:", torch.equal(sc, sc2) or "n/a (sc overwritten by torch)")
t = {n: [] for n in entries}
for _ in range(5):
    for n, fn in entries.items():
        t[n].append(block(fn, 40))
for n in entries:
    print(f"{items} x {S} x {D} q.kT  {n:<20} {float(np.median(t[n])):.4f} ms   blocks {[round(x, 4) for x in t[n]]}")
import time
t0 = time.perf_counter()
for _ in range(200):
    entries["pybind cublas_bmm"]()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host time per pybind call (200 calls issued, not synchronised): {(t1 - t0) / 200 * 1e6:.1f} us")
