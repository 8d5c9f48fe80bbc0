"""Developer probe: every applicable SpMM plan on one shape.   python tools/bench_plans.py M K N density"""
import ctypes
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "matrix-multiplication_amd"))
import custom_mm  # noqa: E402
lib = ctypes.CDLL(str(Path(custom_mm.__file__).parent / "libmi_spmm.so"))
vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
lib.mi_spmm_csr_f32_variant.argtypes = [ctypes.c_int, vp, vp, vp, i64, i32, i32, i32, vp, i64, vp, i64, vp]
lib.mi_spmm_csr_f32_plan.argtypes = [i64, i32, i32, i32, vp, i64, vp, i64]
lib.mi_spmm_variant_name.restype = ctypes.c_char_p
dev = torch.device("cuda")
M, K, N = (int(x) for x in sys.argv[1:4])
density = float(sys.argv[4])
g = torch.Generator(device=dev).manual_seed(0)
a = torch.rand(M, K, device=dev, generator=g) * (torch.rand(M, K, device=dev, generator=g) < density)
val, col, rp = custom_mm.dense_to_csr(a)
del a
rp = rp.view(-1)
nnz = val.numel()
B = torch.rand(K, N, device=dev, generator=g)
C = torch.empty(M, N, device=dev)
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


plan = lib.mi_spmm_csr_f32_plan(nnz, M, K, N, B.data_ptr(), N, C.data_ptr(), N)
out = []
for v in (2, 4, 7, 8, 9, 10, 11, 12, 14, 15, 17, 18, 19, 20, 21, 22, 23, 24):
    if lib.mi_spmm_csr_f32_variant(v, rp.data_ptr(), col.data_ptr(), val.data_ptr(), nnz, M, K, N, B.data_ptr(), N,
                                   C.data_ptr(), N, st) != 0:
        continue
    t = timeit(lambda: lib.mi_spmm_csr_f32_variant(v, rp.data_ptr(), col.data_ptr(), val.data_ptr(), nnz, M, K, N,
                                                   B.data_ptr(), N, C.data_ptr(), N, st))
    out.append(f"{v}{'*' if v == plan else ''}: {t:.3f}")
print(f"M {M} K {K} N {N} density {density}: AUTO = {plan};  ms by variant  " + "  ".join(out), flush=True)
