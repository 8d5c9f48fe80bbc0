"""Exact-fp32 MFMA products: cublas_mmul / cublas_bmm, the fused pair, BASELINE config C5.

Parity of the HIP path with the oracle — needs the MI355X (`-m gpu`).  Everything here calls the product path
(custom_mm → libmi_spmm.so → HIP kernels, or the C-ABI directly through ctypes) and compares with the CPU oracle on the
same seeded inputs: bit-exact where the oracle states the same summation order, rtol 1e-5 / atol 1e-8 (the reference
tests' torch.allclose defaults, tests/naive_kernel_test.py:36-37) against torch expectations and the golden fixtures.
"""
import ctypes
import sys

import numpy as np
import pytest
import torch

from gpu_helpers import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ta", [False, True])
@pytest.mark.parametrize("tb", [False, True])
@pytest.mark.parametrize("m,n,k", [(1, 1, 1), (37, 45, 53), (64, 64, 32), (65, 129, 33), (128, 128, 64), (200, 70, 130),
                                   (130, 260, 7), (3, 300, 257),
                                   (37, 45, 1030), (256, 256, 4099), (16, 16, 1024), (100, 7, 3000), (1, 300, 2048), (512, 512, 1025),
                                   (256, 256, 4096), (64, 96, 2048)])   # tiny aligned outputs, long k: 32×32 tiles under Aᵀ·B
def test_gemm_bit_exact_vs_oracle(cmm, dev, oracle_mod, ta, tb, m, n, k):
    g = np.random.Generator(np.random.PCG64(m * 7 + n * 3 + k))
    a = g.random((k, m) if ta else (m, k), dtype=np.float32)
    b = g.random((n, k) if tb else (k, n), dtype=np.float32)
    C = torch.full((m, n), float("nan"), device=dev)
    out = cmm.cublas_mmul(t(a, dev), t(b, dev), C, ta, tb)
    assert out.data_ptr() == C.data_ptr()
    assert np.array_equal(C.cpu().numpy(), gemm_ref(oracle_mod, a, b, ta, tb))


def test_gemm_few_tiles_long_k_96_tile_kernel_bit_exact(cmm, dev, oracle_mod):
    """Aᵀ·B with few output tiles and a long k (an FC layer's weight gradient dYᵀ·x) runs on 96×96 tiles of
    16×16 MFMA blocks — 256 workgroups at m = 3072, n = 768 instead of 144 — and stays the k-ordered chain:
    bit-identical to the oracle, plain and batched (strided items)."""
    g = np.random.Generator(np.random.PCG64(96))
    dy = g.random((512, 3072), dtype=np.float32) - 0.5   # [k, m]
    x = g.random((512, 768), dtype=np.float32) - 0.5     # [k, n]
    C = torch.full((3072, 768), float("nan"), device=dev)
    cmm.cublas_mmul(t(dy, dev), t(x, dev), C, True, False)
    assert np.array_equal(C.cpu().numpy(), gemm_ref(oracle_mod, dy, x, True, False))
    C2 = torch.full((768, 3072), float("nan"), device=dev)
    cmm.cublas_mmul(t(x, dev), t(dy, dev), C2, True, False)
    assert np.array_equal(C2.cpu().numpy(), gemm_ref(oracle_mod, x, dy, True, False))
    # the same kernel on 64×64 tiles (1024² outputs: 256 workgroups)
    for mn in (1024,):
        a2, b2 = g.random((512, mn), dtype=np.float32) - 0.5, g.random((512, mn), dtype=np.float32) - 0.5
        C3 = torch.full((mn, mn), float("nan"), device=dev)
        cmm.cublas_mmul(t(a2, dev), t(b2, dev), C3, True, False)
        assert np.array_equal(C3.cpu().numpy(), gemm_ref(oracle_mod, a2, b2, True, False)), mn
    a = g.random((4, 576, 768), dtype=np.float32) - 0.5  # four items, k = 576 = 18 chunks
    b = g.random((4, 576, 768), dtype=np.float32) - 0.5
    Cb = torch.full((4, 768, 768), float("nan"), device=dev)
    cmm.cublas_bmm(t(a, dev), t(b, dev), Cb, 3, True, False)
    assert np.array_equal(Cb.cpu().numpy(), gemm_ref(oracle_mod, a, b, True, False))


def test_gemm_tiny_output_long_k_batched(cmm, dev, oracle_mod):
    """Aᵀ·B with a tiny output and a long k runs on 32×32 tiles of 16×16 MFMA blocks (one block per wave) — batched."""
    g = np.random.Generator(np.random.PCG64(16))
    a = g.random((3, 2176, 64), dtype=np.float32) - 0.5    # [k, m] per item, k = 34 chunks
    b = g.random((3, 2176, 96), dtype=np.float32) - 0.5    # [k, n]
    C = torch.full((3, 64, 96), float("nan"), device=dev)
    cmm.cublas_bmm(t(a, dev), t(b, dev), C, 3, True, False)
    assert np.array_equal(C.cpu().numpy(), gemm_ref(oracle_mod, a, b, True, False))


def test_gemm_golden_and_batched(cmm, dev, golden, oracle_mod):
    for name in golden.cases("gemm"):
        c = golden.case(name)
        ta, tb = (bool(x) for x in c["flags"])
        a, b = c["a"], c["b"]
        if a.ndim != b.ndim:
            continue  # mixed ranks go through matmuls (test_matmuls_on_device)
        C = torch.empty(c["c"].shape, device=dev)
        if a.ndim == 2:
            cmm.cublas_mmul(t(a, dev), t(b, dev), C, ta, tb)
        else:
            cmm.cublas_bmm(t(a, dev), t(b, dev), C, a.ndim, ta, tb)
        assert np.array_equal(C.cpu().numpy(), gemm_ref(oracle_mod, a, b, ta, tb)), name
        assert np.allclose(C.cpu().numpy(), c["c"], rtol=RTOL, atol=ATOL), name


def test_gemm_views_are_honoured(cmm, dev, oracle_mod):
    g = torch.Generator(device="cpu").manual_seed(0)
    a, b = torch.rand(40, 50, generator=g), torch.rand(60, 50, generator=g)
    C = torch.empty(40, 60, device=dev)
    # b.t() is a transposed view: must give a @ b.t(), not a misread of b's memory
    cmm.cublas_mmul(a.to(dev), b.to(dev).t(), C, False, False)
    assert np.array_equal(C.cpu().numpy(), oracle_mod.gemm(a.numpy(), b.numpy(), False, True))
    # row slices with a leading dimension, batch-broadcast (stride-0) operand
    wide = torch.rand(3, 40, 100, generator=g)
    bb = torch.rand(50, 20, generator=g)
    C3 = torch.empty(3, 40, 20, device=dev)
    cmm.cublas_bmm(wide.to(dev)[:, :, 25:75], bb.to(dev).expand(3, 50, 20), C3, 3, False, False)
    expect = oracle_mod.gemm(wide[:, :, 25:75].contiguous().numpy(), bb.expand(3, 50, 20).contiguous().numpy())
    assert np.array_equal(C3.cpu().numpy(), expect)
    with pytest.raises(RuntimeError, match="inner dimensions"):
        cmm.cublas_mmul(a.to(dev), b.to(dev), C, False, False)
    with pytest.raises(ValueError, match="Invalid dim"):
        cmm.cublas_bmm(a.to(dev), b.to(dev), C, 7, False, False)


def test_gemm_bert_base_attention_shapes(cmm, dev, oracle_mod):
    """BASELINE.json configs[4]: B=32, H=12, S=512, D=64 — q·kᵀ and probs·v at full size."""
    g = torch.Generator(device="cpu").manual_seed(0)
    q, k, v = (torch.rand(32, 12, 512, 64, generator=g) for _ in range(3))
    qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
    scores = torch.empty(32, 12, 512, 512, device=dev)
    cmm.cublas_bmm(qd, kd, scores, 4, False, True)
    assert torch.allclose(torch.matmul(qd, kd.transpose(-1, -2)), scores, rtol=RTOL, atol=ATOL)
    for (bi, hi) in [(0, 0), (17, 5), (31, 11)]:  # oracle on three heads, bit-exact
        assert np.array_equal(scores[bi, hi].cpu().numpy(), oracle_mod.gemm(q[bi, hi].numpy(), k[bi, hi].numpy(), False, True))
    probs = torch.softmax(scores / 8.0, dim=-1)
    ctxt = torch.empty(32, 12, 512, 64, device=dev)
    cmm.cublas_bmm(probs, vd, ctxt, 4, False, False)
    assert torch.allclose(torch.matmul(probs, vd), ctxt, rtol=RTOL, atol=ATOL)
    p = probs[3, 7].cpu().numpy()
    assert np.array_equal(ctxt[3, 7].cpu().numpy(), oracle_mod.gemm(p, v[3, 7].numpy()))


@pytest.mark.parametrize("items,S,D", [(96, 512, 80), (96, 512, 96), (24, 197, 64), (12, 577, 64), (6, 2048, 64), (24, 300, 72),
                                       (3, 130, 96), (600, 256, 88), (24, 2048, 64)])
def test_gemm_attention_products_at_other_head_sizes_and_lengths_bit_exact(cmm, dev, oracle_mod, items, S, D):
    """Round 5: the four attention products (q·kᵀ, probs·V and the two transposed products of their backward) at head
    sizes 80 / 96 / 72 / 88 — probs·V and Pᵀ·dC there run ONE 96-column tile per 128 rows (4 × 1 wave layout of the
    pipelined kernel) instead of a 128-column one — and at ragged lengths (ViT's 197 / 577 tokens, 300, 130): bit-exact
    against the oracle's k-ordered chain on sampled items, allclose against torch on all.  Reference: README.md:69-77,
    tests/cublas_kernel_test.py:68-69."""
    g = torch.Generator(device="cpu").manual_seed(S + D)
    q, kk, v, dc = (torch.rand(items, S, D, generator=g) - 0.5 for _ in range(4))
    p = torch.rand(items, S, S, generator=g) - 0.5
    d = {n: x.to(dev) for n, x in (("q", q), ("k", kk), ("v", v), ("dc", dc), ("p", p))}
    cases = [("q.kT", d["q"], d["k"], (S, S), False, True, q, kk), ("P.V", d["p"], d["v"], (S, D), False, False, p, v),
             ("PT.dC", d["p"], d["dc"], (S, D), True, False, p, dc), ("dC.VT", d["dc"], d["v"], (S, S), False, True, dc, v)]
    for name, a, b, shape, ta, tb, ah, bh in cases:
        out = torch.full((items,) + shape, float("nan"), device=dev)
        cmm.cublas_bmm(a, b, out, 3, ta, tb)
        ref = torch.matmul(a.transpose(-1, -2) if ta else a, b.transpose(-1, -2) if tb else b)
        assert torch.allclose(ref, out, rtol=1e-4, atol=1e-5), name   # (signed operands: sums cancel)
        for i in sorted({0, items // 2, items - 1}):
            assert np.array_equal(out[i].cpu().numpy(), oracle_mod.gemm(ah[i].numpy(), bh[i].numpy(), ta, tb)), (name, i)


@pytest.mark.parametrize("batch,m,k", [((3, 2), 512, 512), ((5,), 256, 256), ((2, 3), 128, 512), ((9,), 512, 64), ((1,), 64, 128)])
def test_fused_pair_of_products_sharing_an_operand_bit_exact(cmm, mm, dev, oracle_mod, batch, m, k):
    """Round 4: custom_mm.cublas_bmm_pair — dA = dC·B and dB = dCᵀ·A (the backward of cublasTransbMM, reference
    matmuls.py:131-152 / README.md:69-77) in ONE launch that reads dC once — against the oracle's k-ordered chain and, bit
    for bit, against the two plain products it replaces; item counts that are no multiple of 8 (the XCD mapping leaves
    blocks without work), rectangular dC, the smallest and an odd multiple of the key tile; shapes it does not cover
    report False and write nothing; matmuls takes it in cublasTransbMM's backward when both gradients are wanted."""
    n = 64
    g = torch.Generator(device=dev).manual_seed(m + k)
    dC = torch.rand(*batch, m, k, device=dev, generator=g) - 0.5
    B = torch.rand(*batch, k, n, device=dev, generator=g) - 0.5
    A = torch.rand(*batch, m, n, device=dev, generator=g) - 0.5
    dA = torch.full((*batch, m, n), float("nan"), device=dev)
    dB = torch.full((*batch, k, n), float("nan"), device=dev)
    assert cmm.cublas_bmm_pair(dC, B, A, dA, dB) is True
    ref_a = mm.custom_matmul(dC, B)
    ref_b = mm.custom_matmul(dC, A, transa=True)
    assert torch.equal(dA.view(torch.int32), ref_a.view(torch.int32))
    assert torch.equal(dB.view(torch.int32), ref_b.view(torch.int32))
    first = (0,) * len(batch)
    last = tuple(x - 1 for x in batch)
    for it in {first, last}:
        assert np.array_equal(dA[it].cpu().numpy(), oracle_mod.gemm(dC[it].cpu().numpy(), B[it].cpu().numpy()))
        assert np.array_equal(dB[it].cpu().numpy(), oracle_mod.gemm(dC[it].cpu().numpy(), A[it].cpu().numpy(), True, False))
    # through autograd: the drop-in's backward
    q = A.clone().requires_grad_(True)
    kk = B.clone().requires_grad_(True)
    mm.cublasTransbMM.apply(q, kk).backward(dC)
    assert torch.equal(q.grad, dA) and torch.equal(kk.grad, dB)
    # not covered: head dim 32, k beyond 512, strided operands
    d32 = torch.full((2, 64, 32), -7.0, device=dev)
    assert cmm.cublas_bmm_pair(torch.rand(2, 64, 64, device=dev), torch.rand(2, 64, 32, device=dev),
                               torch.rand(2, 64, 32, device=dev), d32, d32.clone()) is False and bool((d32 == -7.0).all())
    assert cmm.cublas_bmm_pair(torch.rand(1, 32, 1024, device=dev), torch.rand(1, 1024, 64, device=dev),
                               torch.rand(1, 32, 64, device=dev), torch.empty(1, 32, 64, device=dev),
                               torch.empty(1, 1024, 64, device=dev)) is False


def test_config_c5_bert_base_attention_full_size_forward_and_backward(mm, dev, oracle_mod):
    """BASELINE config C5 at full size (B 32, H 12, S 512, D 64) through the drop-in wrappers, forward
    AND backward: scores = cublasTransbMM.apply(q, k), ctx = cublasMM.apply(probs, v) (reference
    README.md:69-77) against torch autograd of torch.matmul at rtol 1e-5 (north_star's tolerance), and
    bit-exact against the sequential-k oracle on sampled heads — forward outputs and every gradient."""
    Bz, H, S, D = 32, 12, 512, 64
    g = torch.Generator(device=dev).manual_seed(0)
    q, k, v = (torch.rand(Bz, H, S, D, device=dev, generator=g) for _ in range(3))
    probs = torch.softmax(torch.rand(Bz, H, S, S, device=dev, generator=g) * 4, dim=-1)
    # positive upstream gradients, like the reference tests' torch.rand operands: no cancellation, so a
    # RELATIVE tolerance is well-posed for every element
    d_scores = torch.rand(Bz, H, S, S, device=dev, generator=g)
    d_ctx = torch.rand(Bz, H, S, D, device=dev, generator=g)
    heads = [(0, 0), (17, 5), (31, 11)]

    def leaf(*ts):
        return [x.clone().requires_grad_(True) for x in ts]

    # scores = q.kT
    q1, k1 = leaf(q, k)
    scores = mm.cublasTransbMM.apply(q1, k1)
    scores.backward(d_scores)
    q2, k2 = leaf(q, k)
    ref = torch.matmul(q2, k2.transpose(-1, -2))
    ref.backward(d_scores)
    assert scores.shape == (Bz, H, S, S)
    for got, want in ((scores, ref), (q1.grad, q2.grad), (k1.grad, k2.grad)):
        assert torch.allclose(want, got, rtol=1e-5, atol=1e-8)
    for (b, h) in heads:
        qh, kh, dsh = (x[b, h].cpu().numpy() for x in (q, k, d_scores))
        assert np.array_equal(scores[b, h].detach().cpu().numpy(), oracle_mod.gemm(qh, kh, False, True))
        assert np.array_equal(q1.grad[b, h].cpu().numpy(), oracle_mod.gemm(dsh, kh))               # dQ = dS.K
        assert np.array_equal(k1.grad[b, h].cpu().numpy(), oracle_mod.gemm(dsh, qh, True, False))  # dK = dST.Q
    del scores, ref, q1, k1, q2, k2

    # ctx = probs.v
    p1, v1 = leaf(probs, v)
    ctx = mm.cublasMM.apply(p1, v1)
    ctx.backward(d_ctx)
    p2, v2 = leaf(probs, v)
    ref = torch.matmul(p2, v2)
    ref.backward(d_ctx)
    for got, want in ((ctx, ref), (p1.grad, p2.grad), (v1.grad, v2.grad)):
        assert torch.allclose(want, got, rtol=1e-5, atol=1e-8)
    for (b, h) in heads:
        ph, vh, dch = (x[b, h].cpu().numpy() for x in (probs, v, d_ctx))
        assert np.array_equal(ctx[b, h].detach().cpu().numpy(), oracle_mod.gemm(ph, vh))
        assert np.array_equal(p1.grad[b, h].cpu().numpy(), oracle_mod.gemm(dch, vh, False, True))  # dP = dC.VT
        assert np.array_equal(v1.grad[b, h].cpu().numpy(), oracle_mod.gemm(ph, dch, True, False))  # dV = PT.dC


@pytest.mark.parametrize("rows,n", [(1, 1), (5, 7), (1024, 64), (1025, 65), (16384, 3072), (3, 1000)])
def test_column_sums(cmm, dev, rows, n):
    g = np.random.Generator(np.random.PCG64(rows + n))
    x = g.random((rows, n), dtype=np.float32)
    got = cmm.column_sums(t(x, dev)).cpu().numpy()
    assert got.shape == (n,) and np.allclose(got, x.astype(np.float64).sum(0), rtol=1e-5, atol=1e-6)
    wide = torch.rand(rows, 2 * n + 3, device=dev)
    assert torch.allclose(cmm.column_sums(wide[:, 1:n + 1]), wide[:, 1:n + 1].double().sum(0).float(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("ta,tb", [(False, True), (False, False), (True, False), (True, True)])
def test_gemm_duo_plan_bit_exact_vs_oracle_and_tiles(capi, cmm, dev, oracle_mod, ta, tb):
    """The persistent two-halves kernel (gemm_f32_duo.hip, pinned with mi_gemm_set_plan(2); AUTO takes the tile
    kernels) is the same k-ordered chain: bit-identical to the oracle and to the tile kernels for every transposition,
    128- and 64-column tiles, several k-tiles, an odd number of tiles per workgroup, a batch, and the fused bias;
    a shape that is not made of whole tiles is refused when pinned (reference entry: src/custom_mm.cpp:104-164)."""
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    capi.mi_gemm_bias_f32.argtypes = [ctypes.c_int, ctypes.c_int, i32, i32, i32, vp, i64, i64, vp, i64, i64, vp, vp, i64,
                                      i64, i32, vp]
    stream = torch.cuda.current_stream().cuda_stream
    g = np.random.Generator(np.random.PCG64(int(ta) * 2 + int(tb)))
    try:
        for batch, m, n, k, with_bias in ((1, 384, 256, 192, False), (3, 128, 192, 64, True), (5, 256, 64, 320, False),
                                          (1, 1152, 128, 128, True)):
            a = g.random((batch, k, m) if ta else (batch, m, k), dtype=np.float32) - 0.5
            b = g.random((batch, n, k) if tb else (batch, k, n), dtype=np.float32) - 0.5
            bias = g.random(n, dtype=np.float32) if with_bias else None
            want = gemm_ref(oracle_mod, a, b, ta, tb)
            if with_bias:
                want = want + bias[None, None, :]
            d_a, d_b = t(a, dev), t(b, dev)
            d_bias = t(bias, dev) if with_bias else None
            outs = {}
            for plan in (1, 2):
                assert capi.mi_gemm_set_plan(plan) == 0
                C = torch.full((batch, m, n), float("nan"), device=dev)
                st = capi.mi_gemm_bias_f32(int(ta), int(tb), m, n, k, d_a.data_ptr(), m if ta else k, m * k, d_b.data_ptr(),
                                           k if tb else n, n * k, d_bias.data_ptr() if with_bias else None, C.data_ptr(), n,
                                           m * n, batch, stream)
                assert st == 0, (plan, batch, m, n, k)
                outs[plan] = C.cpu().numpy()
            assert np.array_equal(outs[2], want), (batch, m, n, k)
            assert np.array_equal(outs[1], outs[2])
        # pinned, but 100 rows are not whole 128-row tiles
        assert capi.mi_gemm_set_plan(2) == 0
        a, b, C = torch.rand(100, 64, device=dev), torch.rand(64, 64, device=dev), torch.empty(100, 64, device=dev)
        assert capi.mi_gemm_bias_f32(0, 0, 100, 64, 64, a.data_ptr(), 64, 0, b.data_ptr(), 64, 0, None, C.data_ptr(), 64, 0, 1,
                                     stream) == -1
        assert capi.mi_gemm_set_plan(7) == -1
    finally:
        capi.mi_gemm_set_plan(0)


def test_gemm_shape_and_stride_fuzz_against_oracle(capi, dev, oracle_mod):
    """Random products through the C-ABI (`mi_gemm_bias_f32`, the entry behind `cublas_mmul` / `cublas_bmm`,
    reference src/custom_mm.cpp:104-164): extents around the tile sizes the dispatcher chooses between (32 / 64 / 96 /
    128, ± a few), every transposition, padded leading dimensions, batches with padded item strides, a broadcast
    (stride-0) operand, the fused bias.  Each case: bit-identical to the oracle, and not one element of the padding
    of C is written.  MI_FUZZ_CASES (default 80) / MI_FUZZ_SEED set the number of cases and the seed."""
    import os
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    capi.mi_gemm_bias_f32.argtypes = [ctypes.c_int, ctypes.c_int, i32, i32, i32, vp, i64, i64, vp, i64, i64, vp, vp, i64,
                                      i64, i32, vp]
    stream = torch.cuda.current_stream().cuda_stream
    g = np.random.Generator(np.random.PCG64(int(os.environ.get("MI_FUZZ_SEED", "2026"))))
    cases = int(os.environ.get("MI_FUZZ_CASES", "80"))

    def extent(limit):
        base = int(g.choice([0, 32, 64, 96, 128, 192, 256, 384, 512]))
        return int(min(limit, max(1, base + int(g.integers(-3, 4)) * int(g.integers(0, 2)) + (int(g.integers(1, 40)) if base == 0 else 0))))

    for case in range(cases):
        ta, tb = bool(g.integers(0, 2)), bool(g.integers(0, 2))
        batch = int(g.choice([1, 1, 2, 3, 5]))
        m, n = extent(520), extent(520)
        k = int(g.choice([1, 7, 32, 64, 96, 128, 256, 320, 1024, 2048])) + int(g.integers(0, 3)) * int(g.integers(0, 2))
        if g.integers(0, 4) == 0:  # whole tiles only: the chained short-k kernel, the 16×16-block kernel
            m, n, k = int(g.choice([128, 256, 384])), int(g.choice([64, 128, 256, 512])), int(g.choice([32, 64, 128, 512, 2048]))
        while batch * m * n * k > 40_000_000:  # the oracle's scalar chain stays under a second
            k = max(1, k // 2)
        pad = lambda: int(g.choice([0, 0, 1, 4, 12]))
        rows_a, cols_a = (k, m) if ta else (m, k)
        rows_b, cols_b = (n, k) if tb else (k, n)
        lda, ldb, ldc = cols_a + pad(), cols_b + pad(), n + pad()
        share_b = batch > 1 and g.integers(0, 4) == 0
        sa = rows_a * lda + pad()
        sb = 0 if share_b else rows_b * ldb + pad()
        sc = m * ldc + pad()
        a_buf = g.random(batch * sa + 16, dtype=np.float32) - 0.5
        b_buf = g.random((1 if share_b else batch) * max(sb, rows_b * ldb) + 16, dtype=np.float32) - 0.5
        with_bias = bool(g.integers(0, 3) == 0)
        bias = (g.random(n, dtype=np.float32) - 0.5) if with_bias else None
        view = lambda buf, i, stride, rows, cols, ld: np.lib.stride_tricks.as_strided(
            buf[i * stride:], shape=(rows, cols), strides=(ld * 4, 4))
        want = np.empty((batch, m, n), dtype=np.float32)
        for i in range(batch):
            w = gemm_ref(oracle_mod, np.ascontiguousarray(view(a_buf, i, sa, rows_a, cols_a, lda)),
                         np.ascontiguousarray(view(b_buf, i, sb, rows_b, cols_b, ldb)), ta, tb)
            want[i] = w + bias[None, :] if with_bias else w
        d_a, d_b = t(a_buf, dev), t(b_buf, dev)
        d_bias = t(bias, dev) if with_bias else None
        C = torch.full((batch * sc + 16,), float("nan"), device=dev)
        st = capi.mi_gemm_bias_f32(int(ta), int(tb), m, n, k, d_a.data_ptr(), lda, sa, d_b.data_ptr(), ldb, sb,
                                   d_bias.data_ptr() if with_bias else None, C.data_ptr(), ldc, sc, batch, stream)
        what = (case, ta, tb, batch, m, n, k, lda, ldb, ldc, sa, sb, sc, with_bias)
        assert st == 0, what
        got = C.cpu().numpy()
        written = np.zeros(got.shape, dtype=bool)
        for i in range(batch):
            assert np.array_equal(view(got, i, sc, m, n, ldc), want[i]), what
            written[(i * sc + np.arange(m)[:, None] * ldc + np.arange(n)[None, :]).ravel()] = True
        assert np.isnan(got[~written]).all(), what


@pytest.mark.parametrize("k", [32, 64, 128])
def test_gemm_short_k_chains_of_two_three_and_four_tiles(cmm, dev, oracle_mod, k):
    """Short-k products on whole 128×128 tiles run a CHAIN of output tiles per workgroup (each tile's epilogue inside
    the next tile's MFMAs): four for tile rows of 4·j tiles, two for other even counts, three for 3, 9, 15 … (384
    tokens).  Every chain length, every transposition, batched: bit-identical to the oracle
    (reference entry: cublas_bmm, src/custom_mm.cpp:104-164)."""
    g = np.random.Generator(np.random.PCG64(k))
    for n in (256, 384, 512, 768, 1152):       # 2, 3, 4, 6 (→ 2), 9 (→ 3) tiles per tile row
        for ta, tb in ((False, True), (False, False), (True, False), (True, True)):
            m, batch = 256, 2
            a = g.random((batch, k, m) if ta else (batch, m, k), dtype=np.float32) - 0.5
            b = g.random((batch, n, k) if tb else (batch, k, n), dtype=np.float32) - 0.5
            C = torch.full((batch, m, n), float("nan"), device=dev)
            cmm.cublas_bmm(t(a, dev), t(b, dev), C, 3, ta, tb)
            assert np.array_equal(C.cpu().numpy(), gemm_ref(oracle_mod, a, b, ta, tb)), (n, ta, tb)


@pytest.mark.parametrize("ta,tb", [(False, False), (True, False), (False, True), (True, True)])
@pytest.mark.parametrize("m,n,k", [(197, 64, 197), (577, 64, 300), (130, 70, 333), (69, 197, 64), (5, 3, 40), (260, 129, 1031)])
def test_gemm_rows_off_16_byte_boundaries_take_unconditional_loads_bit_exact(capi, dev, oracle_mod, ta, tb, m, n, k):
    """Round 5: operands whose rows are not 16-byte aligned (ViT's 197 / 577 tokens: probs is 197 × 197; an odd leading
    dimension, a base one float off) take the unconditional buffer loads too — dword-aligned 16-byte loads along k, clamped
    dword loads along m / n — and a long ragged or unaligned k runs in the single-buffer kernel.  Every float around the
    operands (row padding, the floats before and after) is NaN: a load that strayed outside an operand's extent, or a clamp
    that duplicated the wrong element into a stored row / column, would show.  Bit-identical to the oracle, nothing but
    C's m × n written.  Reference: src/custom_mm.cpp:104-164 (cublas_mmul / cublas_bmm take any shape)."""
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    capi.mi_gemm_f32.argtypes = [ctypes.c_int, ctypes.c_int, i32, i32, i32, vp, i64, i64, vp, i64, i64, vp, i64, i64, i32, vp]
    g = np.random.Generator(np.random.PCG64(m * 31 + n * 7 + k + 2 * ta + tb))
    batch = 3
    rows_a, cols_a = (k, m) if ta else (m, k)
    rows_b, cols_b = (n, k) if tb else (k, n)
    for pad_a, pad_b, pad_c, lead in ((0, 0, 0, 0), (1, 3, 1, 1), (2, 0, 5, 3)):
        lda, ldb, ldc = cols_a + pad_a, cols_b + pad_b, n + pad_c
        sa, sb, sc = rows_a * lda + pad_a, rows_b * ldb + pad_b, m * ldc + pad_c
        a_buf = np.full(lead + batch * sa + 8, np.nan, dtype=np.float32)
        b_buf = np.full(lead + batch * sb + 8, np.nan, dtype=np.float32)
        view = lambda buf, i, stride, rows, cols, ld: np.lib.stride_tricks.as_strided(  # noqa: E731
            buf[lead + i * stride:], shape=(rows, cols), strides=(ld * 4, 4))
        want = np.empty((batch, m, n), dtype=np.float32)
        for i in range(batch):
            view(a_buf, i, sa, rows_a, cols_a, lda)[:] = g.random((rows_a, cols_a), dtype=np.float32) - 0.5
            view(b_buf, i, sb, rows_b, cols_b, ldb)[:] = g.random((rows_b, cols_b), dtype=np.float32) - 0.5
            want[i] = gemm_ref(oracle_mod, np.ascontiguousarray(view(a_buf, i, sa, rows_a, cols_a, lda)),
                               np.ascontiguousarray(view(b_buf, i, sb, rows_b, cols_b, ldb)), ta, tb)
        d_a, d_b = t(a_buf, dev), t(b_buf, dev)
        C = torch.full((lead + batch * sc + 8,), float("nan"), device=dev)
        st = capi.mi_gemm_f32(int(ta), int(tb), m, n, k, d_a.data_ptr() + 4 * lead, lda, sa, d_b.data_ptr() + 4 * lead, ldb, sb,
                              C.data_ptr() + 4 * lead, ldc, sc, batch, torch.cuda.current_stream().cuda_stream)
        assert st == 0
        got = C.cpu().numpy()
        written = np.zeros(got.shape, dtype=bool)
        for i in range(batch):
            assert np.array_equal(view(got, i, sc, m, n, ldc), want[i]), (pad_a, pad_b, pad_c, lead, i)
            written[(lead + i * sc + np.arange(m)[:, None] * ldc + np.arange(n)[None, :]).ravel()] = True
        assert np.isnan(got[~written]).all(), (pad_a, pad_b, pad_c, lead)


@pytest.mark.parametrize("m,n,k,ta,tb,bias", [(256, 256, 8192, True, False, False), (384, 130, 4096, False, False, True),
                                              (100, 520, 6144, False, True, False), (768, 64, 16384, True, True, True),
                                              (128, 128, 4160, True, False, False)])
def test_gemm_deterministic_split_k_bit_exact_vs_oracle(cmm, capi, dev, oracle_mod, m, n, k, ta, tb, bias):
    """Round 6: few output tiles and a long k (weight gradients over thousands of tokens) — custom_mm.cublas_mmul cuts k into S
    equal ranges, S a function of the shape alone, each range the k-ordered chain from zero, the partial sums added in index
    order (include/mi_spmm.h "Deterministic split-k"); the oracle restates the rule, so parity stays bit for bit, and the
    result stays within the reference tests' criterion of torch (tests/cublas_kernel_test.py:27-28: allclose at 1e-5).  The
    raw C-ABI entry without a workspace keeps the plain chain; a workspace that is too small is refused, never ignored."""
    g = np.random.Generator(np.random.PCG64(m + n + k))
    a = g.random((k, m) if ta else (m, k), dtype=np.float32) - 0.5
    b = g.random((n, k) if tb else (k, n), dtype=np.float32) - 0.5
    bv = (g.random(n, dtype=np.float32) - 0.5) if bias else None
    S = oracle_mod.gemm_split_count(m, n, k)
    capi.mi_gemm_split_count.argtypes = [ctypes.c_int32] * 4
    assert capi.mi_gemm_split_count(m, n, k, 1) == S and (S > 1) == (k % 64 == 0)
    want = oracle_mod.gemm(a, b, ta, tb)
    chain = oracle_mod.gemm(a, b, ta, tb, split=False)
    if bias:
        want, chain = want + bv[None, :], chain + bv[None, :]
    d_a, d_b = t(a, dev), t(b, dev)
    C = torch.full((m, n), float("nan"), device=dev)
    if bias:
        cmm.cublas_mmul_bias(d_a, d_b, t(bv, dev), C, ta, tb)
    else:
        cmm.cublas_mmul(d_a, d_b, C, ta, tb)
    got = C.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (m, n, k, S)
    if S > 1:
        assert not np.array_equal(want, chain)      # the split order is a different (fixed) order …
    ref = (torch.from_numpy(a).t() if ta else torch.from_numpy(a)).double() @ (torch.from_numpy(b).t() if tb else torch.from_numpy(b)).double()
    if bias:
        ref = ref + torch.from_numpy(bv).double()[None, :]
    scale = (torch.from_numpy(np.abs(a)).t() if ta else torch.from_numpy(np.abs(a))).double() @ \
        (torch.from_numpy(np.abs(b)).t() if tb else torch.from_numpy(np.abs(b))).double()
    assert float(((torch.from_numpy(got).double() - ref).abs() / scale).max()) < 1e-5     # … as close to the exact product
    # the raw entry: plain chain; the workspace form with too little room: refused
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    capi.mi_gemm_bias_f32.argtypes = [ctypes.c_int, ctypes.c_int, i32, i32, i32, vp, i64, i64, vp, i64, i64, vp, vp, i64, i64, i32, vp]
    capi.mi_gemm_ws_f32.argtypes = [ctypes.c_int, ctypes.c_int, i32, i32, i32, vp, i64, i64, vp, i64, i64, vp, vp, i64, i64, i32, vp,
                                    ctypes.c_size_t, vp]
    stream = torch.cuda.current_stream().cuda_stream
    d_bias = t(bv, dev) if bias else None
    args = (int(ta), int(tb), m, n, k, d_a.data_ptr(), m if ta else k, 0, d_b.data_ptr(), k if tb else n, 0,
            d_bias.data_ptr() if bias else None, C.data_ptr(), n, 0, 1)
    assert capi.mi_gemm_bias_f32(*args, stream) == 0
    assert np.array_equal(C.cpu().numpy().view(np.uint32), chain.view(np.uint32))
    if S > 1:
        small = torch.empty(1024, dtype=torch.uint8, device=dev)
        assert capi.mi_gemm_ws_f32(*args, small.data_ptr(), 1024, stream) == -4
        assert capi.mi_gemm_ws_f32(*args, None, 0, stream) == -1
    # a batch is never split
    assert oracle_mod.gemm_split_count(m, n, k, 2) == 1


def test_gemm_split_k_golden_v2(cmm, dev, oracle_mod):
    """The v2 golden fixtures (tests/golden/make_golden_v2.py: torch-CPU products of shapes the split-k rule cuts) through
    custom_mm.cublas_mmul: bit-identical to the oracle's restatement of the rule, within the reference tests' criterion of
    torch's fp32 product (tests/cublas_kernel_test.py:27-28)."""
    from pathlib import Path
    data = np.load(Path(__file__).resolve().parent / "golden" / "golden_v2.npz")
    for name in (str(n) for n in data["__names__"]):
        a, b = data[name + "/a_u8"].astype(np.float32) / 256.0, data[name + "/b_u8"].astype(np.float32) / 256.0
        ta, tb = bool(data[name + "/transa"]), bool(data[name + "/transb"])
        m = a.shape[1] if ta else a.shape[0]
        n = b.shape[0] if tb else b.shape[1]
        C = torch.full((m, n), float("nan"), device=dev)
        cmm.cublas_mmul(t(a, dev), t(b, dev), C, ta, tb)
        got = C.cpu().numpy()
        assert np.array_equal(got.view(np.uint32), oracle_mod.gemm(a, b, ta, tb).view(np.uint32)), name
        assert np.allclose(got, data[name + "/c"], rtol=1e-5, atol=1e-8), name
