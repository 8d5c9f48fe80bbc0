"""Developer probe: BERT q·kᵀ (and probs·v) through custom_mm.cublas_bmm a few times, for rocprofv3 --pmc runs."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "matrix-multiplication_amd"))
import custom_mm  # noqa: E402
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
q = torch.rand(32, 12, 512, 64, device=dev, generator=g)
k = torch.rand(32, 12, 512, 64, device=dev, generator=g)
s = torch.empty(32, 12, 512, 512, device=dev)
c = torch.empty(32, 12, 512, 64, device=dev)
for _ in range(5):
    custom_mm.cublas_bmm(q, k, s, 4, False, True)
for _ in range(5):
    custom_mm.cublas_bmm(s, k, c, 4, False, False)
torch.cuda.synchronize()
