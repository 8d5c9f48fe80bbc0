"""SDDMM (grad of the CSR values: out[p] = <dC[row(p)], B[col(p)]>) in the HBM regime, beside the forward product on the same
operands.   python tools/probes/sddmm_regime.py"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench_hbm_regime as h  # noqa: E402

for mk, N, d in ((1 << 20, 256, 105), (1 << 21, 256, 100), (1 << 21, 128, 100), (1 << 22, 256, 100), (1 << 22, 64, 100), (1 << 21, 256, 20)):
    rowptr, col, val = h.make_csr(mk, mk, d, "uniform")
    nnz = col.numel()
    B = torch.rand(mk, N, device=h.dev)
    dC = torch.rand(mk, N, device=h.dev)
    C = torch.empty(mk, N, device=h.dev)
    t_s = h.timeit(lambda: h.custom_mm.sddmm(col, rowptr, nnz, mk, mk, dC, B), budget_ms=800.0)
    t_f = h.timeit(lambda: h.custom_mm.naive_spmm(val, col, rowptr, nnz, mk, mk, B, C), budget_ms=800.0)
    out = h.custom_mm.sddmm(col, rowptr, nnz, mk, mk, dC, B)
    # sampled entries against fp64
    idx = torch.randint(0, nnz, (4096,), device=h.dev)
    rows = torch.searchsorted(rowptr.long(), idx, right=True) - 1
    want = (dC[rows].double() * B[col[idx].long()].double()).sum(1)
    err = ((out[idx].double() - want).abs() / want.abs().clamp_min(1e-30)).max().item()
    alg = nnz * (4 * N + 8) + 4 * (mk + 1) + 4 * mk * N
    print(f"M=K={mk} N={N} nnz/row={nnz / mk:.0f} |B|={mk * N * 4 >> 20} MiB: sddmm {t_s:8.3f} ms ({alg / t_s / 8e9:.3f} of 8 TB/s on the forward's byte count)  "
          f"forward {t_f:8.3f} ms  max rel err vs fp64 on 4096 entries {err:.2e}", flush=True)
    del rowptr, col, val, B, dC, C, out
    torch.cuda.empty_cache()
