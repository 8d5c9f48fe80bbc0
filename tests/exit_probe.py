"""Helper of tests/test_gpu_schedule.py::test_process_exit_with_live_schedules_and_handles (not collected): leaves an automatic
schedule with a side stream and an inspector handle alive at interpreter exit; the parent checks the exit code."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "matrix-multiplication_amd"))
import custom_mm  # noqa: E402

dev = torch.device("cuda")
g = np.random.Generator(np.random.PCG64(5))
M, K, N = 4000, 20000, 64
lens = g.integers(1, 40, size=M)
lens[11], lens[2000] = 6000, 3000
rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
col = np.concatenate([np.sort(g.choice(K, size=int(n), replace=False)) for n in lens]).astype(np.int32)
val = g.random(len(col), dtype=np.float32)
d_rp, d_col, d_val = (torch.from_numpy(x).to(dev) for x in (rowptr, col, val))
B = torch.rand(K, N, device=dev)
C = torch.empty(M, N, device=dev)
for _ in range(4):
    custom_mm.naive_spmm(d_val, d_col, d_rp, len(val), M, K, B, C)
    torch.cuda.synchronize()
st = custom_mm.auto_schedule_stats()
assert st["active"] == 1, st
custom_mm.cusparse_inspect(d_rp, d_col, d_val, len(val), M, N, K, "left-alive")
custom_mm.cusparse_mmul_opt(B.t().contiguous(), torch.empty(N, M, device=dev), "left-alive")
torch.cuda.synchronize()
print("done", flush=True)
