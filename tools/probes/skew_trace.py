"""Developer probe: one degree-skewed shape through the scheduled product, a few times — a target for rocprofv3 --kernel-trace
(which kernels run, how long, do the heavy and the ordinary launch overlap).   python tools/probes/skew_trace.py arxiv 8000 [heavy_len [inline]]"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "matrix-multiplication_amd"))
sys.path.insert(0, str(ROOT / "tools"))
import custom_mm  # noqa: E402
from bench_degree_skew import csr_from_lengths, pareto_lengths  # noqa: E402

SHAPES = {"arxiv": (170_000, 128, 14), "reddit": (233_000, 602, 490), "products": (2_400_000, 100, 50), "c3skew": (1 << 20, 256, 105)}
name, clip = sys.argv[1], int(sys.argv[2])
M, N, mean = SHAPES[name]
N = int(__import__('os').environ.get('MI_SKEW_N', N))  # (another dense width on the same matrix)
dev = torch.device("cuda")
lens = pareto_lengths(M, mean, clip, M, seed=3)
rowptr, col, val = csr_from_lengths(lens, M, seed=4)
nnz = col.numel()
B = torch.rand(M, N, device=dev)
C = torch.empty(M, N, device=dev)
sched = custom_mm.spmm_schedule(rowptr, nnz, M, N)
if len(sys.argv) > 3:
    sched.set_heavy(int(sys.argv[3]), not (len(sys.argv) > 4 and sys.argv[4] == "inline"))
print(sched.info())
for _ in range(3):
    custom_mm.naive_spmm(val, col, rowptr, nnz, M, M, B, C)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for label, fn in (("plain", lambda: custom_mm.naive_spmm(val, col, rowptr, nnz, M, M, B, C)),
                  ("scheduled", lambda: custom_mm.naive_spmm_scheduled(sched, val, col, rowptr, nnz, M, M, B, C))):
    fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(label, e0.elapsed_time(e1) / 10, "ms")
