"""A/B of builds of the dense fp32 product in ONE process, beside torch.matmul, taking turns block by block:
    python tools/probes/gemm_ab.py items m n k ta tb [name=lib.so …]
`default` is the tree's libmi_spmm.so; a probe library is a single-unit build of csrc/gemm_f32.hip, e.g.
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Imatrix-multiplication_amd/csrc -DMI_GEMM_SINGLE_TU \
          -DMI_GEMM_FORCE_TILE=4 -shared matrix-multiplication_amd/csrc/{gemm_f32,gemm_f32_duo,mi_status}.hip -o tools/probes/gemm_t64.so
Outputs of every library are compared bit for bit with the default's."""
import ctypes
import sys
from pathlib import Path
import numpy as np
import torch
here = Path(__file__).resolve().parent
items, m, n, k, ta, tb = (int(x) for x in sys.argv[1:7])
libs = {"default": here.parent.parent / "matrix-multiplication_amd" / "libmi_spmm.so"}
for spec in sys.argv[7:]:
    name, path = spec.split("=")
    libs[name] = Path(path).resolve()
vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
a = torch.rand((items, k, m) if ta else (items, m, k), device=dev)
b = torch.rand((items, n, k) if tb else (items, k, n), device=dev)
lda, ldb = a.shape[2], b.shape[2]
entries, same = {}, {}
# ONE output buffer for every entry: where a 255 MB output lies moved the same kernel by 25 % (577-token q.kT: 0.122 vs 0.159 ms
# into two buffers of one process, tools/probes/gemm_binding_vs_cabi.py)
c = torch.empty(items, m, n, device=dev)
ref = None
for name, p in libs.items():
    lib = ctypes.CDLL(str(p))
    lib.mi_gemm_f32.argtypes = [ctypes.c_int, ctypes.c_int, i32, i32, i32, vp, i64, i64, vp, i64, i64, vp, i64, i64, i32, vp]
    args = (ta, tb, m, n, k, a.data_ptr(), lda, a.shape[1] * lda, b.data_ptr(), ldb, b.shape[1] * ldb, c.data_ptr(), n, m * n, items, st)
    c.fill_(float("nan"))
    assert lib.mi_gemm_f32(*args) == 0, name
    if ref is None:
        ref = c.clone()
    same[name] = torch.equal(c.view(torch.int32), ref.view(torch.int32))
    entries[name] = (lambda L, ar: (lambda: L.mi_gemm_f32(*ar)))(lib, args)
del ref
at, bt = (a.transpose(-1, -2) if ta else a), (b.transpose(-1, -2) if tb else b)
entries["torch"] = lambda: torch.matmul(at, bt, out=c)
torch.cuda.synchronize()


def block(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for fn in entries.values():
    for _ in range(3):
        fn()
iters = int(max(20, min(400, 4.0 / max(block(entries["default"], 5), 1e-3))))
t = {name: [] for name in entries}
for _ in range(5):
    for name, fn in entries.items():
        t[name].append(block(fn, iters))
flops = 2.0 * items * m * n * k
tt = float(np.median(t["torch"]))
print(f"{items} x ({m} x {k}) . ({k} x {n})  ta={ta} tb={tb}")
for name in entries:
    x = float(np.median(t[name]))
    print(f"  {name:<22} {x:8.4f} ms  {flops / x / 1e9:6.1f} TFLOP/s  / torch {x / tt:5.2f}" + ("" if name == "torch" else f"   bits equal default: {same[name]}"))
