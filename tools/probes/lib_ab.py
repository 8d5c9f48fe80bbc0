"""A/B of builds of the SpMM kernels in ONE process (box-to-box and buffer-placement differences hide a few per cent):
    python tools/probes/lib_ab.py M=K N per_row variant[,variant…] lib_a.so lib_b.so …
Every (library, variant) pair takes turns on the same operands and the same output buffer; medians of 3 rounds."""
import ctypes
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench_hbm_regime as h  # noqa: E402

mk, N, d = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
variants = [int(x) for x in sys.argv[4].split(",")]
libs = {}
for p in sys.argv[5:]:
    L = ctypes.CDLL(str(Path(p).resolve()))
    L.mi_spmm_csr_f32_variant.argtypes = h.lib.mi_spmm_csr_f32_variant.argtypes
    libs[Path(p).name] = L
rowptr, col, val = h.make_csr(mk, mk, d, "uniform")
nnz = col.numel()
B = torch.rand(mk, N, device=h.dev)
C = torch.empty(mk, N, device=h.dev)
st = torch.cuda.current_stream().cuda_stream
entries, ref = {}, None
for name, L in libs.items():
    for v in variants:
        args = (v, rowptr.data_ptr(), col.data_ptr(), val.data_ptr(), nnz, mk, mk, N, B.data_ptr(), N, C.data_ptr(), N, st)
        if L.mi_spmm_csr_f32_variant(*args) != 0:
            continue
        if ref is None:
            ref = C.clone()
        assert torch.equal(C.view(torch.int32), ref.view(torch.int32)), (name, v)
        entries[f"{name}:{v}"] = (lambda L_, a: (lambda: L_.mi_spmm_csr_f32_variant(*a)))(L, args)
alg = nnz * (4 * N + 8) + 4 * (mk + 1) + 4 * mk * N
for k, t in h.time_interleaved(entries, rounds=3, budget_ms=400.0).items():
    print(f"M=K={mk} N={N} nnz/row={nnz / mk:.1f}  {k:<40} {t:9.3f} ms  frac {alg / t / 8e9:.3f}", flush=True)
