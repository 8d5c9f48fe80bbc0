"""Developer probe (round 3): a DENSE tensor with zeros handed to naiveSpMM / cusparseMM — the exact-fp32 MFMA product
beside the zero-skipping routes (in-kernel, dense->CSR + CSR kernels), what `matmuls.naiveSpMM.apply` takes (sampled
density + one read-back included in its time) and what the cost model `matmuls.dense_route_pays` says for the true
density.  -> profiles/r03_dense_input_routing.log"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "matrix-multiplication_amd"))
import custom_mm  # noqa: E402
import matmuls  # noqa: E402
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
cases = []
for kept in (1.0, 0.5, 0.1, 0.05, 0.02, 0.005):
    cases.append(((384, 512, 512), (384, 512, 64), kept))          # BERT-base probs.V
for kept in (1.0, 0.1):
    cases.append(((256, 512, 512), (256, 512, 64), kept))          # reference naive test shape, batch 4096 cut to 256
for kept in (1.0, 0.3, 0.1, 0.02):
    cases.append(((16384, 768), (768, 3072), kept))                # FC layer call shape (flattened tokens x W)
for kept in (1.0, 0.25, 0.1, 0.03, 0.01):
    cases.append(((4096, 4096), (4096, 4096), kept))               # the reference's random-tensor sweep
for kept in (0.5, 0.1, 0.01):
    cases.append(((4096, 4096), (4096, 256), kept))
cases += [((1024, 1024), (1024, 1024), 1.0), ((512, 512), (512, 64), 0.1), ((64, 1024, 1024), (64, 1024, 64), 0.1)]
print("# tools/bench_dense_routing.py on MI355X, round 3 (ms; '-' = route not applicable to the shape)")
for shape_a, shape_b, kept in cases:
    a = torch.rand(*shape_a, device=dev, generator=g)
    if kept < 1.0:
        a = a * (torch.rand(*shape_a, device=dev, generator=g) < kept)
    b = torch.rand(*shape_b, device=dev, generator=g)
    c = torch.empty(*shape_a[:-1], shape_b[-1], device=dev)
    items = 1 if a.dim() == 2 else a.shape[0]
    t_dense = timeit(lambda: matmuls.custom_matmul(a, b))
    t_skip = timeit(lambda: custom_mm.naive_spmm_dense(a, b, c)) if custom_mm.naive_spmm_dense(a, b, c) else None

    def csr_route():
        v, ci, off = custom_mm.dense_to_csr(a)
        if a.dim() == 2:
            custom_mm.naive_spmm(v, ci, off.view(-1), v.numel(), a.shape[0], a.shape[1], b, c)
        else:
            custom_mm.naive_spmm_batched(v, ci, off, v.numel(), a.shape[0], a.shape[1], a.shape[2], b, c)
    t_csr = timeit(csr_route)
    out = matmuls.naiveSpMM.apply(a, b)
    csr_route()
    assert torch.equal(out, c), "the route must not change a bit"
    t_pick = timeit(lambda: matmuls.naiveSpMM.apply(a, b))
    model = "dense" if matmuls.dense_route_pays(kept, items, a.shape[-2], a.shape[-1], b.shape[-1]) else "skip"
    best = min(x for x in (t_dense, t_skip, t_csr) if x is not None)
    print(f"A {tuple(shape_a)} kept {kept:<5} x B {tuple(shape_b)}: dense MFMA {t_dense:.3f} | in-kernel skip "
          f"{'-' if t_skip is None else f'{t_skip:.3f}'} | dense->CSR + CSR {t_csr:.3f} | naiveSpMM.apply {t_pick:.3f} "
          f"(model: {model}; best route {best:.3f})", flush=True)
