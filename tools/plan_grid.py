"""Dispatcher audit on a seeded random grid of shapes: AUTO's plan against every plan pinned through the C-ABI.

    python tools/plan_grid.py [--cases 120] [--seed 5] [--log FILE]

Shapes: M, K log-uniform in [1 Ki, 1 Mi] (multiples of 256), N from {32 … 1024, odd widths included}, 4 … 1024 non-zeros per
row (uniform columns), B and C ≤ 4 GiB, nnz ≤ 2.5e8.  Every plan that accepts the shape runs on the SAME operands and the
SAME output buffer, taking turns (tools/bench_hbm_regime.time_interleaved); its output is compared bit for bit with AUTO's.
Per shape: AUTO's plan and time, the best pinned plan and time, AUTO / best.  The summary lists the shapes where AUTO is
more than 10 % behind.  Reference: src/naive_sparse_mm.cu:24-136 is ONE kernel for every shape; the plans here are this
repository's own, which is why they are audited.
"""
import argparse
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent))
import bench_hbm_regime as h  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=120)
ap.add_argument("--seed", type=int, default=5)
ap.add_argument("--log", default="")
ap.add_argument("--variants", default="2,4,7,8,9,10,11,12,14,15,17,18,19,20,21,22,23,24")
ap.add_argument("--pattern", default="uniform", help="uniform | banded (± 32 K of the diagonal) | band1k (± 1 K) | powerlaw (column popularity) | degskew (power-law row lengths)")
ap.add_argument("--extension", action="store_true", help="AUTO = custom_mm.naive_spmm (the extension's workspace: long-row kernels and the "
                "device-side locality probe are active) instead of the plain C-ABI entry")
a = ap.parse_args()
variants = [int(x) for x in a.variants.split(",")]
lib, dev = h.lib, h.dev
lib.mi_spmm_csr_f32_plan.argtypes = [h.i64, h.i32, h.i32, h.i32, h.vp, h.i64, h.vp, h.i64]
g = np.random.Generator(np.random.PCG64(a.seed))
st = torch.cuda.current_stream().cuda_stream
rows_out = []
print(f"# device {torch.cuda.get_device_name(0)}; tools/plan_grid.py --cases {a.cases} --seed {a.seed} --pattern {a.pattern}{' --extension' if a.extension else ''}; ms per product", flush=True)
case = 0
while case < a.cases:
    M = int(2 ** g.uniform(10, 20)) // 256 * 256
    K = int(2 ** g.uniform(10, 20)) // 256 * 256
    N = int(g.choice([32, 64, 96, 100, 128, 192, 256, 256, 320, 384, 512, 768, 1024]))
    d = int(2 ** g.uniform(2, 10))
    d = max(1, min(d, K // 2))
    if K * N * 4 > (4 << 30) or M * N * 4 > (4 << 30) or M * d > 2.5e8 or M * d * N > 6e10:
        continue
    case += 1
    rowptr, col, val = h.make_csr(M, K, d, a.pattern, seed=case)
    nnz = col.numel()
    B = torch.rand(K, N, device=dev)
    C = torch.empty(M, N, device=dev)
    plan = lib.mi_spmm_csr_f32_plan(nnz, M, K, N, B.data_ptr(), N, C.data_ptr(), N)
    entries, same, ref = {}, {}, None
    for v in [0] + variants:
        args = (v, rowptr.data_ptr(), col.data_ptr(), val.data_ptr(), nnz, M, K, N, B.data_ptr(), N, C.data_ptr(), N, st)
        C.fill_(float("nan"))
        if lib.mi_spmm_csr_f32_variant(*args) != 0:
            continue
        if ref is None:
            ref = C.clone()
        same[v] = torch.equal(C.view(torch.int32), ref.view(torch.int32))
        entries[v] = (lambda ar: (lambda: lib.mi_spmm_csr_f32_variant(*ar)))(args)
    if a.extension:
        import custom_mm
        C.fill_(float("nan"))
        custom_mm.naive_spmm(val, col, rowptr, nnz, M, K, B, C)
        same[0] = torch.equal(C.view(torch.int32), ref.view(torch.int32))
        entries[0] = lambda: custom_mm.naive_spmm(val, col, rowptr, nnz, M, K, B, C)
    del ref
    ms = h.time_interleaved(entries, rounds=3, budget_ms=60.0)
    auto = ms[0]
    best_v = min((v for v in ms if v != 0), key=lambda v: ms[v])
    line = (f"M {M:>8} K {K:>8} N {N:>4} per-row {nnz / M:7.1f} |B| {K * N * 4 / 2**20:7.1f} MiB  AUTO {plan:>2} {auto:9.4f}  best {best_v:>2} "
            f"{ms[best_v]:9.4f}  ratio {auto / ms[best_v]:5.2f}  bits {'ok' if all(same.values()) else 'DIFFER ' + str([v for v, s in same.items() if not s])}   "
            + " ".join(f"{v}:{t:.4f}" for v, t in ms.items() if v != 0))
    print(line, flush=True)
    rows_out.append((auto / ms[best_v], line))
    del rowptr, col, val, B, C
    torch.cuda.empty_cache()
rows_out.sort(key=lambda r: -r[0])
ratios = np.array([r[0] for r in rows_out])
summary = [f"# {len(rows_out)} shapes: AUTO / best pinned — median {np.median(ratios):.3f}, 90th percentile {np.percentile(ratios, 90):.3f}, "
           f"max {ratios.max():.3f}; more than 10 % behind: {(ratios > 1.10).sum()}"]
summary += ["# " + r[1][:150] for r in rows_out if r[0] > 1.10]
print("\n".join(summary), flush=True)
if a.log:
    Path(a.log).parent.mkdir(parents=True, exist_ok=True)
    Path(a.log).write_text("\n".join([r[1] for r in rows_out] + summary) + "\n")
