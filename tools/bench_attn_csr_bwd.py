"""Developer probe: pruned BERT-base attention probs·V with the probabilities as a batched CSR TENSOR through
matmuls.cusparseMM.apply — forward and forward + backward (both gradients), beside the dense cublasMM.apply.

    python tools/bench_attn_csr_bwd.py [--fresh]

--fresh: a NEW CSR tensor (new top-k pattern) on every iteration, as attention probabilities are — whatever matmuls
keeps on the tensor object between calls (narrowed indices, the transposed pattern and its permutation) is then paid
inside the timing, as the reference pays its conversion on every call (matmuls.py:289-297, backward :245-256).  The
tensors are built beforehand (top-k + to_sparse_csr are the caller's work in both libraries)."""
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


FRESH = "--fresh" in sys.argv
# --nonleaf (with --fresh): the CSR tensor is the OUTPUT of an upstream op — torch.sparse_csr_tensor over a dense leaf of values —
# as attention probabilities are (softmax → top-k → CSR); its gradient then flows on as values instead of being accumulated
# on a sparse leaf, which autograd does by CLONING the whole CSR gradient (≈ 53 µs of the leaf figure; a dense leaf's
# gradient is taken over without a copy, so the dense classes never pay that).  Measured: NOT a way out — torch's own backward of
# the sparse_csr_tensor constructor takes ≈ 3 ms at this size (fwd + bwd 3.48 ms at 10 % kept against 0.35 with a leaf): the leaf
# figure is the one quoted.
NONLEAF = "--nonleaf" in sys.argv
WARM, ITERS = 3, 10


def timeit(fn, iters=ITERS):
    """fn(i): i counts every call, warm-up included (a --fresh run hands call i its own tensor)."""
    for i in range(WARM):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(WARM + i)
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

    def make():
        # equal non-zero counts per item (torch's batched CSR layout)
        idx = torch.rand(items, S * S, device=dev, generator=g).topk(per_item, dim=1).indices
        mask = torch.zeros(items, S * S, device=dev).scatter_(1, idx, 1.0)
        probs = ((torch.rand(items, S * S, device=dev, generator=g) * 0.9 + 0.1) * mask).reshape(items, S, S)
        return probs

    probs = make()
    pd = probs.clone().requires_grad_(True)
    def csr_of(dense):
        c = dense.to_sparse_csr()
        if not NONLEAF:
            return c.requires_grad_(True)
        vals = c.values().detach().clone().requires_grad_(True)
        return torch.sparse_csr_tensor(c.crow_indices(), c.col_indices(), vals, size=c.shape)

    if FRESH:  # one tensor per call of each timed loop (forward alone, then forward + backward)
        pool_f = [csr_of(make()) for _ in range(WARM + ITERS)]
        pool_fb = [csr_of(make()) for _ in range(WARM + ITERS)]
    else:
        a = probs.to_sparse_csr().requires_grad_(True)
        pool_f = pool_fb = [a] * (WARM + ITERS)
    del probs

    def fb(cls, x):
        if x.is_leaf:
            x.grad = None
        v.grad = None
        cls.apply(x, v).backward(dctx)
    t_f = timeit(lambda i: matmuls.cusparseMM.apply(pool_f[i], v))
    t_fb = timeit(lambda i: fb(matmuls.cusparseMM, pool_fb[i]))
    d_f = timeit(lambda i: matmuls.cublasMM.apply(pd, v))
    d_fb = timeit(lambda i: fb(matmuls.cublasMM, pd))
    print(f"kept {kept:4.2f}{' fresh pattern per call' if FRESH else ''}{' (CSR tensor = output of an upstream op)' if NONLEAF else ''}: CSR tensor fwd {t_f:.3f} fwd+bwd {t_fb:.3f}   "
          f"dense cublasMM fwd {d_f:.3f} fwd+bwd {d_fb:.3f}", flush=True)
    del pool_f, pool_fb
