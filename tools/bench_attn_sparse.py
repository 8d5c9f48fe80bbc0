"""Developer probe: BERT-base probs·V (B 32, H 12, S 512, D 64) with pruned probabilities — naiveSpMM.apply(probs_pruned, v)
forward and forward+backward beside the dense cublasMM.apply, per kept fraction."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "matrix-multiplication_amd"))
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
v = torch.rand(32, 12, 512, 64, device=dev, generator=g)
dctx = torch.rand(32, 12, 512, 64, device=dev, generator=g)
for kept in (1.0, 0.5, 0.1, 0.02):
    probs = torch.rand(32, 12, 512, 512, device=dev, generator=g)
    if kept < 1:
        probs = probs * (torch.rand(32, 12, 512, 512, device=dev, generator=g) < kept)
    line = f"kept {kept}:"
    for name, cls in (("cublasMM", matmuls.cublasMM), ("naiveSpMM", matmuls.naiveSpMM)):
        def fwd():
            with torch.no_grad():
                cls.apply(probs, v)

        def fwdbwd():
            p = probs.clone().requires_grad_(True)
            vv = v.clone().requires_grad_(True)
            cls.apply(p, vv).backward(dctx)
        line += f"  {name} fwd {timeit(fwd):.3f} / fwd+bwd {timeit(fwdbwd):.3f} ms"
    print(line, flush=True)
