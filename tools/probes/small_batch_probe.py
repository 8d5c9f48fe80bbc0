import ctypes, sys
from pathlib import Path
import torch
PKG = Path("/root/repo/matrix-multiplication_amd")
sys.path.insert(0, str(PKG))
import custom_mm
lib = ctypes.CDLL(str(PKG / "libmi_spmm.so"))
vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
lib.mi_spmm_csr_batched_variant_f32.argtypes = [ctypes.c_int, vp, vp, vp, i64, i32, i32, i32, i32, vp, i64, i64, vp, i64, i64, vp]
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best
for items, M, K, N in ((4,512,512,64),(8,512,512,64),(12,512,512,64),(16,512,512,64),(24,512,512,64),(32,512,512,64),(48,512,512,64),(12,1024,1024,64),(12,128,128,64),(96,128,128,64)):
    v = torch.rand(items, K, N, device=dev, generator=g)
    c = torch.empty(items, M, N, device=dev)
    for kept in (0.25, 0.05):
        probs = torch.rand(items, M, K, device=dev, generator=g) * (torch.rand(items, M, K, device=dev, generator=g) < kept)
        val, col, off = custom_mm.dense_to_csr(probs)
        nnz = val.numel()
        def run(variant):
            return lib.mi_spmm_csr_batched_variant_f32(variant, off.data_ptr(), col.data_ptr(), val.data_ptr(), nnz, items, M, K, N, v.data_ptr(), N, K*N, c.data_ptr(), N, M*N, torch.cuda.current_stream().cuda_stream)
        tg = timeit(lambda: run(4)); tl = timeit(lambda: run(18)); ta = timeit(lambda: run(0))
        print(f"{items:4d} x {M} x {K} x {N} kept {kept}: group {tg:.4f}  lds {tl:.4f}  auto {ta:.4f}", flush=True)
