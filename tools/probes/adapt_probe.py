"""What the device-side locality probe buys: custom_mm.naive_spmm (AUTO with the extension's workspace: the probe runs) beside AUTO
without a workspace (plain C-ABI entry: the passes always stay passes) and the one-pass plans, on one shape and pattern.
    python tools/probes/adapt_probe.py M K N per_row pattern"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench_hbm_regime as h  # noqa: E402
import custom_mm  # noqa: E402
M, K, N, d = (int(x) for x in sys.argv[1:5])
pattern = sys.argv[5]
lib, dev = h.lib, h.dev
lib.mi_spmm_csr_f32_plan.argtypes = [h.i64, h.i32, h.i32, h.i32, h.vp, h.i64, h.vp, h.i64]
st = torch.cuda.current_stream().cuda_stream
rowptr, col, val = h.make_csr(M, K, d, pattern)
nnz = col.numel()
B = torch.rand(K, N, device=dev)
C = torch.empty(M, N, device=dev)
plan = lib.mi_spmm_csr_f32_plan(nnz, M, K, N, B.data_ptr(), N, C.data_ptr(), N)
entries = {"naive_spmm (probe)": lambda: custom_mm.naive_spmm(val, col, rowptr, nnz, M, K, B, C)}
entries["naive_spmm (probe)"]()
ref = C.clone()
same = {}
for v in (0, 2, 4):
    args = (v, rowptr.data_ptr(), col.data_ptr(), val.data_ptr(), nnz, M, K, N, B.data_ptr(), N, C.data_ptr(), N, st)
    C.fill_(float("nan"))
    if lib.mi_spmm_csr_f32_variant(*args) != 0:
        continue
    name = {0: f"AUTO without workspace (plan {plan})", 2: "one pass: wave per row", 4: "one pass: lane groups"}[v]
    same[name] = torch.equal(C.view(torch.int32), ref.view(torch.int32))
    entries[name] = (lambda a: (lambda: lib.mi_spmm_csr_f32_variant(*a)))(args)
ms = h.time_interleaved(entries, rounds=5, budget_ms=150.0)
print(f"M {M} K {K} N {N} per-row {nnz / M:.1f} |B| {K * N * 4 / 2**20:.1f} MiB {pattern}: " +
      "   ".join(f"{k}: {t:.4f}{'' if same.get(k, True) else ' BITS DIFFER'}" for k, t in ms.items()), flush=True)
