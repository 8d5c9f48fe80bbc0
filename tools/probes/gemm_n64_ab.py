"""A/B builds of the dense kernels on probs.V / PT.dC at n = 64 (developer probe; the probe macro was removed once its verdict went into pick_tile — history: profiles/r05_gemm_n64_ab.log):
    MI_HIPCC_FLAGS=-DMI_GEMM_N64_PROBE=1|2 builds of libmi_spmm.so copied to tools/probes/gemm_n64_probe{1,2}.so
    python tools/probes/gemm_n64_ab.py"""
import ctypes
import sys
from pathlib import Path
import torch
here = Path(__file__).resolve().parent
vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
libs = {"default": here.parent.parent / "matrix-multiplication_amd" / "libmi_spmm.so", "wn1 (4x1 waves)": here / "gemm_n64_probe1.so",
        "n16 (16x16 blocks)": here / "gemm_n64_probe2.so"}
L = {}
for k, p in libs.items():
    if p.exists():
        lib = ctypes.CDLL(str(p))
        lib.mi_gemm_f32.argtypes = [ctypes.c_int, ctypes.c_int, i32, i32, i32, vp, i64, i64, vp, i64, i64, vp, i64, i64, i32, vp]
        L[k] = lib
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best


for items, S, D in ((48, 2048, 64), (96, 1024, 64), (384, 512, 64)):
    p = torch.rand(items, S, S, device=dev)
    v = torch.rand(items, S, D, device=dev)
    ref = None
    for ta in (0, 1):
        outs = {}
        for name, lib in L.items():
            c = torch.empty(items, S, D, device=dev)
            call = lambda: lib.mi_gemm_f32(ta, 0, S, D, S, p.data_ptr(), S, S * S, v.data_ptr(), D, S * D, c.data_ptr(), D, S * D, items, st)
            assert call() == 0
            outs[name] = (timeit(call), c)
        t_torch = timeit(lambda: torch.matmul(p.transpose(-1, -2) if ta else p, v))
        base = outs["default"][1]
        print(f"{items} x {S} x {D} {'PT.dC' if ta else 'P.V  '}: torch {t_torch:.4f}  " +
              "  ".join(f"{k} {t:.4f}{'' if torch.equal(c, base) else ' (DIFFERS)'}" for k, (t, c) in outs.items()), flush=True)
