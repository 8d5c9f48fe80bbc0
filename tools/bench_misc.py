"""Developer timing probe for the non-headline entry points (BERT config C5, conversions,
column-major executor).  Prints one line per measurement; torch (rocBLAS / hipSPARSE) timings are
shown beside ours for orientation only.

    python tools/bench_misc.py
"""
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "matrix-multiplication_amd"))
import custom_mm  # noqa: E402
import matmuls  # noqa: E402


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(0)
    Bz, H, S, D = 32, 12, 512, 64
    q = torch.rand(Bz, H, S, D, device=dev, generator=g)
    k = torch.rand(Bz, H, S, D, device=dev, generator=g)
    v = torch.rand(Bz, H, S, D, device=dev, generator=g)
    scores = torch.empty(Bz, H, S, S, device=dev)
    ctx = torch.empty(Bz, H, S, D, device=dev)
    fl_qk = 2.0 * Bz * H * S * S * D

    t = timeit(lambda: custom_mm.cublas_bmm(q, k, scores, 4, False, True))
    print(f"gemm q.kT  ours   {t:8.3f} ms  {fl_qk / t / 1e9:8.1f} TFLOP/s  out-write {scores.numel() * 4 / t / 1e6:7.0f} GB/s")
    t = timeit(lambda: torch.matmul(q, k.transpose(-1, -2), out=scores))
    print(f"gemm q.kT  torch  {t:8.3f} ms  {fl_qk / t / 1e9:8.1f} TFLOP/s")
    probs = torch.softmax(scores / 8, dim=-1)
    t = timeit(lambda: custom_mm.cublas_bmm(probs, v, ctx, 4, False, False))
    print(f"gemm p.v   ours   {t:8.3f} ms  {fl_qk / t / 1e9:8.1f} TFLOP/s  in-read {probs.numel() * 4 / t / 1e6:7.0f} GB/s")
    t = timeit(lambda: torch.matmul(probs, v, out=ctx))
    print(f"gemm p.v   torch  {t:8.3f} ms  {fl_qk / t / 1e9:8.1f} TFLOP/s")
    # backward-shaped products: dQ = dS·K (nn), dK = dSᵀ·Q (tn)
    dq = torch.empty_like(q)
    t = timeit(lambda: custom_mm.cublas_bmm(scores, k, dq, 4, False, False))
    print(f"gemm dS.K  ours   {t:8.3f} ms  {fl_qk / t / 1e9:8.1f} TFLOP/s")
    t = timeit(lambda: custom_mm.cublas_bmm(scores, q, dq, 4, True, False))
    print(f"gemm dST.Q ours   {t:8.3f} ms  {fl_qk / t / 1e9:8.1f} TFLOP/s")
    t = timeit(lambda: torch.matmul(scores.transpose(-1, -2), q, out=dq))
    print(f"gemm dST.Q torch  {t:8.3f} ms  {fl_qk / t / 1e9:8.1f} TFLOP/s")

    # whole fwd+bwd through the autograd wrapper
    def fwd_bwd(fn):
        qq, kk = q.detach().requires_grad_(True), k.detach().requires_grad_(True)
        fn(qq, kk).backward(scores)
    t = timeit(lambda: fwd_bwd(matmuls.cublasTransbMM.apply), iters=10)
    print(f"cublasTransbMM fwd+bwd ours  {t:8.3f} ms")
    t = timeit(lambda: fwd_bwd(lambda a, b: torch.matmul(a, b.transpose(-1, -2))), iters=10)
    print(f"matmul         fwd+bwd torch {t:8.3f} ms")

    # pruned attention probabilities through the sparse path (top 10 % per row)
    for keep in (1.0, 0.1):
        kth = max(1, int(S * keep))
        thresh = probs.topk(kth, dim=-1).values[..., -1:]
        pp = torch.where(probs >= thresh, probs, torch.zeros_like(probs))
        a3 = pp.reshape(-1, S, S)
        t = timeit(lambda: custom_mm.dense_to_csr(a3), iters=5)
        vals, cols, offs = custom_mm.dense_to_csr(a3)
        print(f"dense_to_csr keep={keep}: {t:8.3f} ms  ({a3.numel() * 4 / t / 1e6:7.0f} GB/s read), nnz {vals.numel()}")
        c3 = torch.empty(Bz * H, S, D, device=dev)
        v3 = v.reshape(-1, S, D)
        t = timeit(lambda: custom_mm.naive_spmm_batched(vals, cols, offs, vals.numel(), Bz * H, S, S, v3, c3))
        print(f"batched spmm keep={keep}: {t:8.3f} ms  {2.0 * vals.numel() * D / t / 1e9:8.1f} TFLOP/s")
        t = timeit(lambda: custom_mm.naive_spmm_dense(pp, v, c3.view(Bz, H, S, D)))
        print(f"fused dense-skip keep={keep}: {t:8.3f} ms  (A read {pp.numel() * 4 / t / 1e6:7.0f} GB/s)")
        t = timeit(lambda: matmuls.naiveSpMM.apply(pp, v), iters=5)
        print(f"naiveSpMM.apply keep={keep}: {t:8.3f} ms (whole call)")

    # column-major executor (FC layer: y[N,M] = x[N,K] · Wᵀ, sparse W M×K)
    M, K, N = 4096, 4096, 512
    w = torch.rand(M, K, device=dev, generator=g) * (torch.rand(M, K, device=dev, generator=g) < 0.05)
    vals, cols, offs = custom_mm.dense_to_csr(w)
    custom_mm.cusparse_inspect(offs.view(-1), cols, vals, vals.numel(), M, N, K, "w")
    x = torch.rand(N, K, device=dev, generator=g)
    y = torch.empty(N, M, device=dev)
    t = timeit(lambda: custom_mm.cusparse_mmul_opt(x, y, "w"))
    print(f"cusparse_mmul_opt {M}x{K} 5% x {N}: {t:8.3f} ms  {2.0 * vals.numel() * N / t / 1e9:8.2f} TFLOP/s")
    t = timeit(lambda: custom_mm.naive_spmm(vals, cols, offs.view(-1), vals.numel(), M, K, x.t().contiguous(),
                                            torch.empty(M, N, device=dev)))
    print(f"  (row-major spmm incl. x.t() copy: {t:8.3f} ms)")
    custom_mm.cusparse_clean()

    # CSR transpose + SDDMM at a mid size
    vt = timeit(lambda: custom_mm.csr_transpose(vals, cols, offs.view(-1), vals.numel(), M, K), iters=5)
    print(f"csr_transpose nnz {vals.numel()}: {vt:8.3f} ms")
    dC = torch.rand(M, 256, device=dev, generator=g)
    Bm = torch.rand(K, 256, device=dev, generator=g)
    st = timeit(lambda: custom_mm.sddmm(cols, offs.view(-1), vals.numel(), M, K, dC, Bm), iters=5)
    print(f"sddmm N=256 nnz {vals.numel()}: {st:8.3f} ms  {vals.numel() * 1024 / st / 1e6:7.0f} GB/s gather")


if __name__ == "__main__":
    main()
