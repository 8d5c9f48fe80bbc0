"""Developer probe: A/B builds of csrc/spmm_ldsb.hip alone (hipcc -shared -DMI_LDSB_PROBE [-D…] spmm_ldsb.hip mi_status.hip
-o tools/probes/ldsb_ab_<tag>.so) timed against each other in ONE process on the pruned-attention shape — box-to-box
differences (±5 %) otherwise hide a few per cent.   python tools/probes/ldsb_ab.py tools/probes/ldsb_ab_*.so"""
import ctypes
import sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(REPO / "matrix-multiplication_amd"))
import custom_mm  # noqa: E402
vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
libs = []
for path in sys.argv[1:]:
    lib = ctypes.CDLL(path)
    lib.mi_ldsb_probe.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i64, vp]
    libs.append((Path(path).stem, lib))
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best


for items, M, K, N in ((384, 512, 512, 64), (96, 1024, 1024, 64), (48, 2048, 2048, 64)):
    v = torch.rand(items, K, N, device=dev, generator=g)
    c = torch.empty(items, M, N, device=dev)
    ref = None
    for kept in (1.0, 0.25, 0.1, 0.02):
        probs = torch.rand(items, M, K, device=dev, generator=g)
        if kept < 1:
            probs = probs * (torch.rand(items, M, K, device=dev, generator=g) < kept)
        val, col, off = custom_mm.dense_to_csr(probs)
        line = f"{items:4d} x {M} x {K} x {N} kept {kept:4.2f}:"
        ref = None
        for rounds in range(2):  # twice, so that drift over the run shows
            for name, lib in libs:
                t = timeit(lambda: lib.mi_ldsb_probe(off.data_ptr(), col.data_ptr(), val.data_ptr(), v.data_ptr(), c.data_ptr(),
                                                     items, M, K, N, val.numel(), None))
                if ref is None:
                    ref = c.clone()
                assert torch.equal(ref, c), name
                line += f"  {name} {t:.4f}"
        print(line, flush=True)
        del probs, val, col, off
