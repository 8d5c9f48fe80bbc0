"""One L2-regime shape, one pinned plan, four products — the program for rocprofv3 counter passes (tools/probes/l2_pmc.sh):
    python3 tools/probes/l2_pmc_one.py M K N per_row variant [pattern]"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench_hbm_regime as h  # noqa: E402
M, K, N, d, v = (int(x) for x in sys.argv[1:6])
rowptr, col, val = h.make_csr(M, K, d, sys.argv[6] if len(sys.argv) > 6 else "uniform")
nnz = col.numel()
B = torch.rand(K, N, device=h.dev)
C = torch.empty(M, N, device=h.dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(4):
    assert h.lib.mi_spmm_csr_f32_variant(v, rowptr.data_ptr(), col.data_ptr(), val.data_ptr(), nnz, M, K, N, B.data_ptr(), N, C.data_ptr(), N, st) == 0
torch.cuda.synchronize()
print(f"nnz {nnz} alg_bytes {nnz * (4 * N + 8) + 4 * (M + 1) + 4 * M * N}")
