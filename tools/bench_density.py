"""Developer probe: square A (dim×dim) at several densities × dense B (dim×N): the row-split AUTO plan, the LDS-slab
kernel (variant 17) and the dense MFMA product.   python tools/bench_density.py [dim] [N] [densities…]"""
import ctypes
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "matrix-multiplication_amd"))
import custom_mm  # noqa: E402
lib = ctypes.CDLL(str(Path(custom_mm.__file__).parent / "libmi_spmm.so"))
vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
lib.mi_spmm_csr_f32_variant.argtypes = [ctypes.c_int, vp, vp, vp, i64, i32, i32, i32, vp, i64, vp, i64, vp]
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
N = int(sys.argv[2]) if len(sys.argv) > 2 else dim


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


B = torch.rand(dim, N, device=dev, generator=g)
C0, C1 = torch.empty(dim, N, device=dev), torch.empty(dim, N, device=dev)
DENS = [float(x) for x in sys.argv[3:]] or [0.5, 0.25, 0.1, 0.05, 0.03, 0.02, 0.01, 0.003]
for density in DENS:
    a = torch.rand(dim, dim, device=dev, generator=g)
    a = a * (torch.rand(dim, dim, device=dev, generator=g) < density)
    val, col, rp = custom_mm.dense_to_csr(a)
    rp = rp.view(-1)
    nnz = val.numel()
    st = torch.cuda.current_stream().cuda_stream

    def run(variant, C):
        s = lib.mi_spmm_csr_f32_variant(variant, rp.data_ptr(), col.data_ptr(), val.data_ptr(), nnz, dim, dim, N,
                                        B.data_ptr(), N, C.data_ptr(), N, st)
        assert s == 0, s
    t_auto = timeit(lambda: run(0, C0))
    t_slab = timeit(lambda: run(17, C1))
    same = torch.equal(C0, C1)
    t_dense = timeit(lambda: custom_mm.cublas_mmul(a, B, C0, False, False))
    fl = 2.0 * nnz * N
    print(f"dim {dim} N {N} density {density:5.3f}: auto {t_auto:8.3f} ms  slab {t_slab:8.3f} ms ({fl / t_slab / 1e9:6.1f} TFLOP/s)  "
          f"dense {t_dense:8.3f} ms  bit-identical {same}", flush=True)
