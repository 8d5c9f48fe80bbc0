"""Developer probe: what the follow-up launch of a product WITHOUT long rows costs (the listed-rows kernel finds three zeros and
exits) — small uniform products through custom_mm.naive_spmm; a target for rocprofv3 --kernel-trace --stats (tools/kstats.sh)."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "matrix-multiplication_amd"))
sys.path.insert(0, str(ROOT / "tools"))
import custom_mm  # noqa: E402
from bench_hbm_regime import make_csr  # noqa: E402

dev = torch.device("cuda")
for (M, per, N) in ((16384, 64, 128), (65536, 32, 64), (262144, 16, 256), (32768, 100, 512)):
    rowptr, col, val = make_csr(M, M, per, "uniform", 1)
    nnz = col.numel()
    B = torch.rand(M, N, device=dev)
    C = torch.empty(M, N, device=dev)
    for _ in range(3):
        custom_mm.naive_spmm(val, col, rowptr, nnz, M, M, B, C)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        custom_mm.naive_spmm(val, col, rowptr, nnz, M, M, B, C)
    e1.record()
    torch.cuda.synchronize()
    print(f"M {M} per row {per} N {N} nnz {nnz}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per product")
