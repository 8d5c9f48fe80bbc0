"""Developer probe: BERT-base probs·V (384 items of 512×512 · 512×64) with the probabilities ALREADY in batched CSR form
(no dense read), per kept fraction: the row-split group kernel (gathers from the L2s), the LDS-resident-B kernel
(MI_SPMM_LDS_B), AUTO's choice through custom_mm.naive_spmm_batched, and the dense MFMA product of the same operands;
bit-equality of the two CSR kernels asserted.  Then other batched / tall shapes whose B fits LDS.
-> profiles/r03_attention_csr.log"""
import ctypes
import sys
from pathlib import Path
import torch
PKG = Path(__file__).resolve().parent.parent / "matrix-multiplication_amd"
sys.path.insert(0, str(PKG))
import custom_mm  # noqa: E402
lib = ctypes.CDLL(str(PKG / "libmi_spmm.so"))
vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
lib.mi_spmm_csr_batched_variant_f32.argtypes = [ctypes.c_int, vp, vp, vp, i64, i32, i32, i32, i32, vp, i64, i64, vp, i64, i64, vp]
lib.mi_spmm_csr_batched_f32_plan.argtypes = [i64, i32, i32, i32, i32, vp, i64, i64, vp, i64, i64]
dev = torch.device("cuda")
GROUP, LDSB = 4, 18


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


def run(variant, off, col, val, nnz, items, M, K, N, b, c):
    st = lib.mi_spmm_csr_batched_variant_f32(variant, off.data_ptr(), col.data_ptr(), val.data_ptr(), nnz, items, M, K, N,
                                             b.data_ptr(), N, K * N, c.data_ptr(), N, M * N,
                                             torch.cuda.current_stream().cuda_stream)
    if st == -1 and variant == LDSB:
        return False  # the plan does not take this shape
    assert st == 0, (variant, st)
    return True


g = torch.Generator(device=dev).manual_seed(0)
print("# tools/bench_attn_csr.py on MI355X (ms; best of 3 blocks of 20)")
print("# items x M x K x N   kept     nnz        group(L2)  LDS-B    AUTO(plan)      dense MFMA")
shapes = [(384, 512, 512, 64, (1.0, 0.5, 0.25, 0.1, 0.05, 0.02, 0.01)),
          (96, 1024, 1024, 64, (0.25, 0.1, 0.05, 0.02)), (48, 2048, 2048, 64, (0.1, 0.05, 0.02)), (96, 1024, 1024, 128, (0.1,)),
          (384, 128, 128, 64, (0.5, 0.1)), (384, 256, 256, 64, (0.5, 0.1)), (96, 1024, 512, 64, (0.1, 0.02)),
          (192, 512, 256, 128, (0.25, 0.05)), (768, 512, 1024, 32, (0.1, 0.02)), (1, 131072, 512, 64, (0.1, 0.02)),
          (1, 65536, 128, 256, (0.25, 0.05)), (4096, 64, 64, 64, (0.5,))]
for items, M, K, N, kepts in shapes:
    v = torch.rand(items, K, N, device=dev, generator=g)
    c1, c2, c3, c4 = (torch.empty(items, M, N, device=dev) for _ in range(4))
    for kept in kepts:
        probs = torch.rand(items, M, K, device=dev, generator=g)
        if kept < 1:
            probs = probs * (torch.rand(items, M, K, device=dev, generator=g) < kept)
        val, col, off = custom_mm.dense_to_csr(probs)
        nnz = val.numel()
        t_grp = timeit(lambda: run(GROUP, off, col, val, nnz, items, M, K, N, v, c1))
        fits = run(LDSB, off, col, val, nnz, items, M, K, N, v, c2)
        t_lds = timeit(lambda: run(LDSB, off, col, val, nnz, items, M, K, N, v, c2)) if fits else float("nan")
        assert not fits or torch.equal(c1, c2), "the two CSR kernels must agree bit for bit"
        t_auto = timeit(lambda: custom_mm.naive_spmm_batched(val, col, off, nnz, items, M, K, v, c3))
        assert torch.equal(c1, c3)
        plan = lib.mi_spmm_csr_batched_f32_plan(nnz, items, M, K, N, v.data_ptr(), N, K * N, c3.data_ptr(), N, M * N)
        t_dense = timeit(lambda: custom_mm.cublas_bmm(probs, v, c4, 3, False, False))
        print(f"{items:5d} x {M:6d} x {K:4d} x {N:3d}  {kept:5.2f} {nnz:11d}   {t_grp:8.4f}  {t_lds:8.4f}  {t_auto:8.4f} ({plan:2d})   {t_dense:8.4f}",
              flush=True)
        del probs, val, col, off
