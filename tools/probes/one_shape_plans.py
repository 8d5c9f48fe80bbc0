"""Every plan on ONE shape (sparse generator, same output buffer, interleaved):  python tools/probes/one_shape_plans.py M K N per_row [pattern]"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench_hbm_regime as h  # noqa: E402
M, K, N, d = (int(x) for x in sys.argv[1:5])
pattern = sys.argv[5] if len(sys.argv) > 5 else "uniform"
lib, dev = h.lib, h.dev
lib.mi_spmm_csr_f32_plan.argtypes = [h.i64, h.i32, h.i32, h.i32, h.vp, h.i64, h.vp, h.i64]
st = torch.cuda.current_stream().cuda_stream
rowptr, col, val = h.make_csr(M, K, d, pattern)
nnz = col.numel()
B = torch.rand(K, N, device=dev)
C = torch.empty(M, N, device=dev)
plan = lib.mi_spmm_csr_f32_plan(nnz, M, K, N, B.data_ptr(), N, C.data_ptr(), N)
entries, ref, same = {}, None, {}
for v in (2, 4, 7, 8, 9, 10, 11, 12, 14, 15, 17, 18, 19, 20, 21, 22, 23, 24):
    args = (v, rowptr.data_ptr(), col.data_ptr(), val.data_ptr(), nnz, M, K, N, B.data_ptr(), N, C.data_ptr(), N, st)
    C.fill_(float("nan"))
    if lib.mi_spmm_csr_f32_variant(*args) != 0:
        continue
    if ref is None:
        ref = C.clone()
    same[v] = torch.equal(C.view(torch.int32), ref.view(torch.int32))
    entries[v] = (lambda a: (lambda: lib.mi_spmm_csr_f32_variant(*a)))(args)
ms = h.time_interleaved(entries, rounds=5, budget_ms=200.0)
print(f"M {M} K {K} N {N} per-row {nnz / M:.1f} |B| {K * N * 4 / 2**20:.1f} MiB {pattern}: AUTO = {plan};  " +
      "  ".join(f"{v}{'*' if v == plan else ''}: {t:.4f}{'' if same[v] else '!'}" for v, t in ms.items()), flush=True)
