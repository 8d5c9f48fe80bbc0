"""Developer probe: wide-N SpMM (the reference's random-tensor sweep regime) — group kernel vs the
XCD-aware column-tiled launch, through the C-ABI variant override."""
import ctypes
import sys
from pathlib import Path
import torch

LIB = Path(__file__).resolve().parent.parent / "matrix-multiplication_amd" / "libmi_spmm.so"
lib = ctypes.CDLL(str(LIB))
vp = ctypes.c_void_p
lib.mi_spmm_csr_f32_variant.argtypes = [ctypes.c_int, vp, vp, vp, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                        ctypes.c_int32, vp, ctypes.c_int64, vp, ctypes.c_int64, vp]
dev = torch.device("cuda")


def run(variant, rp, ci, v, M, K, B, C):
    st = lib.mi_spmm_csr_f32_variant(variant, rp.data_ptr(), ci.data_ptr(), v.data_ptr(), v.numel(), M, K, B.shape[1],
                                     B.data_ptr(), B.stride(0), C.data_ptr(), C.stride(0),
                                     torch.cuda.current_stream().cuda_stream)
    assert st == 0, st


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for (M, K, N, dens) in [(4096, 4096, 4096, 0.01), (4096, 4096, 4096, 0.1), (8192, 8192, 8192, 0.01), (8192, 8192, 8192, 0.1),
                        (12288, 12288, 12288, 0.01), (16384, 16384, 16384, 0.01), (65536, 8192, 1024, 0.005),
                        (4096, 4096, 768, 0.05)]:
    g = torch.Generator(device=dev).manual_seed(M + N)
    a = torch.rand(M, K, device=dev, generator=g)
    a = a * (torch.rand(M, K, device=dev, generator=g) < dens)
    csr = a.to_sparse_csr()
    rp, ci, v = csr.crow_indices().int(), csr.col_indices().int(), csr.values()
    del a, csr
    B = torch.rand(K, N, device=dev, generator=g)
    ref = torch.empty(M, N, device=dev)
    run(5, rp, ci, v, M, K, B, ref)  # scalar group kernel as the reference result
    line = f"M={M} K={K} N={N} dens={dens} nnz={v.numel()}:"
    for variant, name in [(0, "auto"), (4, "group"), (14, "coltile"), (15, "coltile+panels")]:
        C = torch.empty(M, N, device=dev)
        try:
            run(variant, rp, ci, v, M, K, B, C)
        except AssertionError:
            continue
        same = torch.equal(C, ref)
        t = timeit(lambda: run(variant, rp, ci, v, M, K, B, C))
        line += f"  {name} {t:.3f} ms ({v.numel() * 4.0 * N / t / 1e9:.1f} TB/s gather){'' if same else ' MISMATCH'}"
    print(line, flush=True)
