"""Developer probe: the arxiv-like matrix with one row beyond the long-row threshold through a scheduled product with the
prepared list (long_rows = -1) and without it (long_rows = 1: the list is built by a scan of the heavy slots) — a target for
rocprofv3 --kernel-trace.   python tools/probes/early_list.py"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "matrix-multiplication_amd"))
sys.path.insert(0, str(ROOT / "tools"))
import custom_mm  # noqa: E402
from bench_degree_skew import csr_from_lengths, pareto_lengths  # noqa: E402

M, N, mean = 170_000, 128, 14
dev = torch.device("cuda")
lens = pareto_lengths(M, mean, 0, M, seed=3)
rowptr, col, val = csr_from_lengths(lens, M, seed=4)
nnz = col.numel()
B = torch.rand(M, N, device=dev)
C = torch.empty(M, N, device=dev)
sched = custom_mm.spmm_schedule(rowptr, nnz, M, N)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for label, lr in (("prepared", -1), ("scanned", 1), ("prepared", -1), ("scanned", 1)):
    fn = lambda: custom_mm.naive_spmm_scheduled(sched, val, col, rowptr, nnz, M, M, B, C, None, lr)
    fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(label, e0.elapsed_time(e1) / 10, "ms")
