"""Developer probe: pruned BERT-base attention probs·V with the probabilities as a batched CSR TENSOR through
matmuls.cusparseMM.apply — forward and forward + backward (both gradients), beside the dense cublasMM.apply."""
import os
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "matrix-multiplication_amd"))
import matmuls  # noqa: E402
dev = torch.device("cuda")
if "MI_LDSB_FORM" in os.environ:  # developer A/B: pin the 16-lane (0) / quad (1) form of the LDS-resident kernels
    import ctypes
    _lib = ctypes.CDLL(str(Path(__file__).resolve().parent.parent / "matrix-multiplication_amd" / "libmi_spmm.so"))
    _lib.mi_spmm_ldsb_set_form.argtypes = [ctypes.c_int]
    _lib.mi_spmm_ldsb_set_form(int(os.environ["MI_LDSB_FORM"]))


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


g = torch.Generator(device=dev).manual_seed(0)
S = int(os.environ.get('MI_SEQ', '512'))
items, D = 384 * 512 * 512 // (S * S), 64  # (the same number of score entries at every length)
v = torch.rand(items, S, D, device=dev, generator=g, requires_grad=True)
dctx = torch.rand(items, S, D, device=dev, generator=g)
print(f"# tools/bench_attn_csr_bwd.py on MI355X: {items} x ({S}x{S} . {S}x{D}), ms")
for kept in tuple(float(x) for x in os.environ.get('MI_KEPT','0.25,0.1,0.05').split(',')):
    per_item = int(S * S * kept)
    # equal non-zero counts per item (torch's batched CSR layout)
    idx = torch.rand(items, S * S, device=dev, generator=g).topk(per_item, dim=1).indices
    mask = torch.zeros(items, S * S, device=dev).scatter_(1, idx, 1.0)
    probs = ((torch.rand(items, S * S, device=dev, generator=g) * 0.9 + 0.1) * mask).reshape(items, S, S)
    a = probs.to_sparse_csr().requires_grad_(True)
    pd = probs.clone().requires_grad_(True)

    def fb(cls, x):
        x.grad = None
        v.grad = None
        cls.apply(x, v).backward(dctx)
    t_f = timeit(lambda: matmuls.cusparseMM.apply(a, v))
    t_fb = timeit(lambda: fb(matmuls.cusparseMM, a))
    d_f = timeit(lambda: matmuls.cublasMM.apply(pd, v))
    d_fb = timeit(lambda: fb(matmuls.cublasMM, pd))
    print(f"kept {kept:4.2f}: CSR tensor fwd {t_f:.3f} fwd+bwd {t_fb:.3f}   dense cublasMM fwd {d_f:.3f} fwd+bwd {d_fb:.3f}", flush=True)
