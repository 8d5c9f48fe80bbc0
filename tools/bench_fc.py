"""Developer probe: the FC-layer modules (fc_layers.cublasLinear / cusparseLinear) vs torch.nn.Linear,
forward + backward, at BERT-base FFN shapes with ReLU-sparse activations."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "matrix-multiplication_amd"))
import fc_layers  # noqa: E402

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


SHAPES = [(32 * 512, 768, 3072), (32 * 512, 3072, 768), (4096, 4096, 4096), (32 * 512, 3072, 256)]
for (tokens, fin, fout, zero_frac) in [(t, i, o, z) for (t, i, o) in SHAPES for z in (0.5, 0.75, 0.9, 0.99)]:
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.rand(tokens, fin, device=dev, generator=g)
    x = x * (torch.rand(tokens, fin, device=dev, generator=g) >= zero_frac)
    dy = torch.rand(tokens, fout, device=dev, generator=g)
    line = f"tokens={tokens} {fin}->{fout} zeros={zero_frac}:"
    for name, layer in [("nn.Linear", torch.nn.Linear(fin, fout).to(dev)), ("cublasLinear", fc_layers.cublasLinear(fin, fout).to(dev)),
                        ("cusparseLinear", fc_layers.cusparseLinear(fin, fout).to(dev))]:
        def step():
            xx = x.clone().requires_grad_(True)
            layer.zero_grad(set_to_none=True)
            layer(xx).backward(dy)
        def fwd():
            with torch.no_grad():
                layer(x)
        line += f"  {name} fwd {timeit(fwd):.3f} / fwd+bwd {timeit(step):.3f} ms"
    print(line, flush=True)

# components of the dense backward at tokens=16384, 768 -> 3072
import custom_mm  # noqa: E402
tokens, fin, fout = 32 * 512, 768, 3072
g = torch.Generator(device=dev).manual_seed(1)
x = torch.rand(tokens, fin, device=dev, generator=g)
w = torch.rand(fout, fin, device=dev, generator=g)
dy = torch.rand(tokens, fout, device=dev, generator=g)
y = torch.empty(tokens, fout, device=dev)
gi = torch.empty(tokens, fin, device=dev)
gw = torch.empty(fout, fin, device=dev)
ones = torch.ones(1, tokens, device=dev)
gb = torch.empty(1, fout, device=dev)
print(f"fwd x.WT        {timeit(lambda: custom_mm.cublas_mmul(x, w, y, False, True)):.3f} ms (torch {timeit(lambda: torch.matmul(x, w.t(), out=y)):.3f})")
print(f"grad_inp dY.W   {timeit(lambda: custom_mm.cublas_mmul(dy, w, gi, False, False)):.3f} ms (torch {timeit(lambda: torch.matmul(dy, w, out=gi)):.3f})")
print(f"grad_w dYT.x    {timeit(lambda: custom_mm.cublas_mmul(dy, x, gw, True, False)):.3f} ms (torch {timeit(lambda: torch.matmul(dy.t(), x, out=gw)):.3f})")
print(f"grad_b colsums  {timeit(lambda: custom_mm.column_sums(dy)):.3f} ms (torch sum {timeit(lambda: dy.sum(0)):.3f}; as a 1 x tokens GEMM "
      f"{timeit(lambda: custom_mm.cublas_mmul(ones, dy, gb, False, False)):.3f})")
