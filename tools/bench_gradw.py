import sys, torch
sys.path.insert(0, "matrix-multiplication_amd")
import custom_mm
import numpy as np
dev = torch.device("cuda")
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
g = torch.Generator(device=dev).manual_seed(1)
for (tokens, fin, fout) in [(16384, 768, 3072), (16384, 3072, 768), (4096, 4096, 4096), (8192, 1536, 768)]:
    x = torch.rand(tokens, fin, device=dev, generator=g); dy = torch.rand(tokens, fout, device=dev, generator=g)
    gw = torch.empty(fout, fin, device=dev)
    t = timeit(lambda: custom_mm.cublas_mmul(dy, x, gw, True, False))
    tt = timeit(lambda: torch.matmul(dy.t(), x, out=gw))
    custom_mm.cublas_mmul(dy, x, gw, True, False)
    ref = (dy.double().t() @ x.double()).float()
    print(f"grad_w dYT.x  tokens={tokens} {fout}x{fin}: ours {t:.3f} ms  torch {tt:.3f} ms  maxrel {((gw-ref).abs()/ref.abs()).max().item():.2e}")
