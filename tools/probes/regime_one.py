"""One HBM-regime shape through custom_mm.naive_spmm, a few products — the program for rocprofv3 counter passes:
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir> -- python3 tools/probes/regime_one.py 2097152 128 100"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench_hbm_regime as h  # noqa: E402
mk, N, d = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rowptr, col, val = h.make_csr(mk, mk, d, "uniform")
nnz = col.numel()
B = torch.rand(mk, N, device=h.dev)
C = torch.empty(mk, N, device=h.dev)
for _ in range(4):
    h.custom_mm.naive_spmm(val, col, rowptr, nnz, mk, mk, B, C)
torch.cuda.synchronize()
print(f"nnz {nnz} alg_bytes {nnz * (4 * N + 8) + 4 * (mk + 1) + 4 * mk * N} plan {h.custom_mm.spmm_plan(nnz, mk, mk, B, C)}")
