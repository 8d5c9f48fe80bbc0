"""Developer probe: the kernel variants that can take config C2's shape (64k x 64k at 0.1 % x 128), timed."""
import ctypes, sys
from pathlib import Path
import numpy as np, torch
PKG = Path(__file__).resolve().parents[2] / "matrix-multiplication_amd"
lib = ctypes.CDLL(str(PKG / "libmi_spmm.so"))
vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
lib.mi_spmm_csr_f32_variant.argtypes = [ctypes.c_int, vp, vp, vp, i64, i32, i32, i32, vp, i64, vp, i64, vp]
dev = torch.device("cuda")
M = K = 65536
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
g = torch.Generator(device=dev).manual_seed(0)
lens = torch.poisson(torch.full((M,), 65.5, device=dev), generator=g).to(torch.int64)
rp = torch.zeros(M + 1, dtype=torch.int64, device=dev); rp[1:] = lens.cumsum(0)
nnz = int(rp[-1])
col = torch.randint(0, K, (nnz,), device=dev, generator=g, dtype=torch.int32)
rows = torch.repeat_interleave(torch.arange(M, device=dev), lens)
order = torch.argsort(rows * K + col.to(torch.int64)); col = col[order].contiguous()
val = torch.rand(nnz, device=dev, generator=g)
rp32 = rp.to(torch.int32)
B = torch.rand(K, N, device=dev, generator=g); C = torch.empty(M, N, device=dev); ref = None
st = torch.cuda.current_stream().cuda_stream
for variant in (0, 4, 13, 5, 14):
    def run():
        return lib.mi_spmm_csr_f32_variant(variant, rp32.data_ptr(), col.data_ptr(), val.data_ptr(), nnz, M, K, N, B.data_ptr(), N, C.data_ptr(), N, st)
    if run() != 0:
        print(variant, "n/a"); continue
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    if ref is None: ref = C.clone()
    print(f"variant {variant:2d}: {best:.4f} ms  same bits as AUTO: {bool(torch.equal(ref, C))}", flush=True)
