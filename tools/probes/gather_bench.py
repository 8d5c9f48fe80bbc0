import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path("/root/repo/matrix-multiplication_amd")))
import custom_mm
dev = torch.device("cuda")
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for n in (10_000_000, 110_000_000):
    v = torch.rand(n, device=dev)
    # a transpose-like permutation: stride pattern with locality similar to a CSR transpose (sorted by column)
    perm = torch.randperm(n, device=dev).to(torch.int32)
    t1 = timeit(lambda: custom_mm.gather_perm(v, perm))
    pl = perm.long()
    t2 = timeit(lambda: v.index_select(0, perm))
    print(f"n {n}: gather_perm {t1:.3f} ms  index_select(int32) {t2:.3f} ms")
