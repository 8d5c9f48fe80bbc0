"""Why did `custom_mm.naive_spmm` (AUTO + long-row workspace + follow-up launch) time 10-15 % above the same plan pinned
through mi_spmm_csr_f32_variant on the HBM-regime sweep's large short-row shapes?  Alternates the entries on one shape.
    python tools/probes/auto_vs_pinned.py [M=K] [N] [per_row]"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench_hbm_regime as h  # noqa: E402

mk = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 21
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
d = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rowptr, col, val = h.make_csr(mk, mk, d, "uniform")
nnz = col.numel()
B = torch.rand(mk, N, device=h.dev)
C = torch.empty(mk, N, device=h.dev)
C2 = torch.empty(mk, N, device=h.dev)
st = torch.cuda.current_stream().cuda_stream
plan = h.custom_mm.spmm_plan(nnz, mk, mk, B, C)
print("plan", plan, flush=True)
pin = lambda out: h.lib.mi_spmm_csr_f32_variant(plan[0], rowptr.data_ptr(), col.data_ptr(), val.data_ptr(), nnz, mk, mk, N,
                                                  B.data_ptr(), N, out.data_ptr(), N, st)
entries = {
    "naive_spmm(C)": lambda: h.custom_mm.naive_spmm(val, col, rowptr, nnz, mk, mk, B, C),
    "pinned(C2)": lambda: pin(C2),
    "pinned(C)": lambda: pin(C),
    "naive_spmm(C2)": lambda: h.custom_mm.naive_spmm(val, col, rowptr, nnz, mk, mk, B, C2),
    "naive_spmm_ex rule0 (C)": lambda: h.custom_mm.naive_spmm_ex(val, col, rowptr, nnz, mk, mk, B, C, 0),
}
for rnd in range(3):
    for name, fn in entries.items():
        print(f"round {rnd} {name:<26} {h.timeit(fn, budget_ms=400.0):8.3f} ms", flush=True)
