"""Developer probe: dense input with zeros × dense B at widths 256 < N ≤ 1024 — the fused zero-skipping kernel run
as column tiles of 256 (no CSR, no host read-back) beside the CSR route (dense_to_csr + naive_spmm[_batched])."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "matrix-multiplication_amd"))
import custom_mm  # noqa: E402
dev = torch.device("cuda")


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


g = torch.Generator(device=dev).manual_seed(0)
for (shape_a, shape_b, kept) in [((512, 512), (512, 64), 0.1), ((1024, 1024), (1024, 256), 0.1), ((1024, 1024), (1024, 1024), 1.0), ((512, 512), (512, 256), 1.0), ((2048, 2048), (2048, 256), 0.1),
                                 ((2048, 2048), (2048, 64), 0.5), ((4096, 4096), (4096, 256), 0.1), ((4096, 4096), (4096, 64), 0.1),
                                 ((16384, 768), (768, 256), 0.5), ((16384, 768), (768, 256), 0.1), ((16384, 768), (768, 768), 0.1),
                                 ((16384, 3072), (3072, 512), 0.1), ((4096, 4096), (4096, 1024), 0.01),
                                 ((2, 512, 512), (2, 512, 64), 0.1), ((8, 512, 512), (8, 512, 64), 0.1), ((384, 512, 512), (384, 512, 64), 0.1),
                                 ((384, 512, 512), (384, 512, 256), 0.1), ((384, 512, 512), (384, 512, 512), 0.1),
                                 ((64, 1024, 1024), (64, 1024, 64), 0.1), ((16, 2048, 2048), (16, 2048, 128), 0.1),
                                 ((32, 128, 3072), (3072, 256), 0.2), ((4, 4096, 4096), (4, 4096, 256), 0.1)]:
    a = torch.rand(*shape_a, device=dev, generator=g) * (torch.rand(*shape_a, device=dev, generator=g) < kept)
    b = torch.rand(*shape_b, device=dev, generator=g)
    c = torch.empty(*shape_a[:-1], shape_b[-1], device=dev)
    assert custom_mm.naive_spmm_dense(a, b, c)
    t_fused = timeit(lambda: custom_mm.naive_spmm_dense(a, b, c))

    def csr_route():
        v, ci, off = custom_mm.dense_to_csr(a)
        if a.dim() == 2:
            custom_mm.naive_spmm(v, ci, off.view(-1), v.numel(), a.shape[0], a.shape[1], b, c)
        else:
            custom_mm.naive_spmm_batched(v, ci, off, v.numel(), a.shape[0], a.shape[1], a.shape[2], b, c)
    c1 = c.clone()
    csr_route()
    assert torch.equal(c, c1)
    t_csr = timeit(csr_route)
    import matmuls  # noqa: E402
    items = 1 if a.dim() == 2 else a.shape[0]
    pick = "in-kernel" if matmuls.fused_skip_pays(items, a.shape[-2], a.shape[-1], b.shape[-1]) else "CSR"
    print(f"A {tuple(shape_a)} kept {kept} x B {tuple(shape_b)}: zeros skipped in the kernel {t_fused:.3f} ms; dense->CSR + CSR kernels "
          f"{t_csr:.3f} ms; matmuls takes the {pick} route", flush=True)
