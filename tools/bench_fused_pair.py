"""Developer probe: the fused pair of products (dQ = dS·K, dK = dSᵀ·Q in one launch, csrc/gemm_f32_fused.hip) at the C5
shape against the two plain products; optional ablation libraries (tools/probes/fused_probe_abl*.so, timing only).

    python tools/bench_fused_pair.py [lib.so …]
"""
import ctypes
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "matrix-multiplication_amd"))
import custom_mm  # noqa: E402
import matmuls  # noqa: E402

dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
items, S, D = 384, 512, 64
dS = torch.rand(items, S, S, device=dev, generator=g)
q = torch.rand(items, S, D, device=dev, generator=g)
k = torch.rand(items, S, D, device=dev, generator=g)
dq, dk = torch.empty_like(q), torch.empty_like(k)


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


flops = 2 * 2.0 * items * S * S * D
t = timeit(lambda: (matmuls.custom_matmul(dS, k), matmuls.custom_matmul(dS, q, transa=True)))
print(f"two plain products      {t:.4f} ms  ({flops / t / 1e9:6.1f} TFLOP/s)")
t = timeit(lambda: custom_mm.cublas_bmm_pair(dS, k, q, dq, dk))
print(f"fused (libmi_spmm.so)   {t:.4f} ms  ({flops / t / 1e9:6.1f} TFLOP/s)")
for path in sys.argv[1:]:
    lib = ctypes.CDLL(path)
    vp, i32 = ctypes.c_void_p, ctypes.c_int32
    lib.mi_gemm_pair_a_at_f32.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    st = torch.cuda.current_stream().cuda_stream
    t = timeit(lambda: lib.mi_gemm_pair_a_at_f32(dS.data_ptr(), k.data_ptr(), q.data_ptr(), dq.data_ptr(), dk.data_ptr(),
                                                 items, S, S, D, st))
    print(f"{Path(path).name:24s}{t:.4f} ms")
