"""Host logic of matmuls.py on CPU tensors — runs without a GPU.

matmuls.py is imported against tests/fake_custom_mm.py (the oracle behind the
custom_mm names), so rank dispatch, output shapes, broadcasting and the
autograd formulas are checked against torch.matmul / torch autograd — the
expectation of the reference's tests (tests/naive_kernel_test.py:30,36-37) and,
for gradients, SURVEY.md §8a defect 2's "correct gradients" list.
Covers BASELINE.json configs[0] (dense 8×64 @ 64×8 through the wrappers, fwd+bwd).
"""
import importlib
import sys

import numpy as np
import pytest
import torch

RTOL, ATOL = 1e-5, 1e-8


@pytest.fixture()
def mm(oracle_mod):
    """(matmuls bound to the fake custom_mm, the fake module)."""
    import fake_custom_mm
    saved = {k: sys.modules.get(k) for k in ("custom_mm", "matmuls")}
    sys.modules["custom_mm"] = fake_custom_mm
    sys.modules.pop("matmuls", None)
    matmuls = importlib.import_module("matmuls")
    fake_custom_mm.calls.clear()
    yield matmuls, fake_custom_mm
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


def rand(g, *shape):
    return torch.rand(*shape, generator=g)


def check_fwd_bwd(fn, ref_fn, a, b, grad_a=True, grad_b=True):
    a1, b1 = a.clone().requires_grad_(grad_a), b.clone().requires_grad_(grad_b)
    a2, b2 = a.clone().requires_grad_(grad_a), b.clone().requires_grad_(grad_b)
    out, exp = fn(a1, b1), ref_fn(a2, b2)
    assert out.shape == exp.shape
    assert torch.allclose(exp, out, rtol=RTOL, atol=ATOL)
    dc = torch.rand(exp.shape, generator=torch.Generator().manual_seed(3))
    out.backward(dc)
    exp.backward(dc)
    if grad_a:
        assert a1.grad.shape == a.shape and torch.allclose(a2.grad, a1.grad, rtol=RTOL, atol=ATOL)
    else:
        assert a1.grad is None
    if grad_b:
        assert b1.grad.shape == b.shape and torch.allclose(b2.grad, b1.grad, rtol=RTOL, atol=ATOL)
    else:
        assert b1.grad is None


def test_c1_plumbing_config(mm, monkeypatch):
    """BASELINE.json configs[0]: dense 8×64 @ 64×8 via the wrappers, fwd + bwd, equals torch.mm."""
    matmuls, fake = mm
    torch.manual_seed(0)
    a, b = torch.rand(8, 64), torch.rand(64, 8)
    check_fwd_bwd(matmuls.cublasMM.apply, torch.mm, a, b)
    check_fwd_bwd(matmuls.naiveSpMM.apply, torch.mm, a, b)
    check_fwd_bwd(matmuls.cusparseMM.apply, torch.mm, a, b)
    names = [c[0] for c in fake.calls]
    assert "cublas_mmul" in names and "naive_spmm_dense" in names
    fake.fused_dense = False  # and through the CSR kernels proper
    try:
        check_fwd_bwd(matmuls.naiveSpMM.apply, torch.mm, a, b)
        check_fwd_bwd(matmuls.cusparseMM.apply, torch.mm, a, b)
    finally:
        fake.fused_dense = True
    names = [c[0] for c in fake.calls]
    assert "naive_spmm_ex" in names   # the stock kernels' one-launch entry: a dense 8 × 64 matrix has no over-long row
    fake.calls.clear()
    monkeypatch.delattr(fake, "naive_spmm_ex")   # a custom_mm with the reference's names only
    check_fwd_bwd(matmuls.naiveSpMM.apply, torch.mm, sparsify(torch.Generator().manual_seed(1), 30, 40), torch.rand(40, 8))
    fake.fused_dense = False
    try:
        check_fwd_bwd(matmuls.naiveSpMM.apply, torch.mm, a, b)
        check_fwd_bwd(matmuls.cusparseMM.apply, torch.mm, a, b)
    finally:
        fake.fused_dense = True
    names = [c[0] for c in fake.calls]
    assert "naive_spmm" in names and "cusparse_mmul" in names


@pytest.mark.parametrize("cls,ta,tb", [("cublasMM", False, False), ("cublasTransaMM", True, False),
                                       ("cublasTransbMM", False, True), ("cublasTransabMM", True, True)])
@pytest.mark.parametrize("batch", [(), (3,), (2, 3), (2, 1, 3)])
def test_dense_classes_all_ranks(mm, cls, ta, tb, batch):
    matmuls, _ = mm
    g = torch.Generator().manual_seed(11)
    m, n, k = 5, 7, 6
    a = rand(g, *batch, *((k, m) if ta else (m, k)))
    b = rand(g, *batch, *((n, k) if tb else (k, n)))

    def ref(x, y):
        return torch.matmul(x.transpose(-1, -2) if ta else x, y.transpose(-1, -2) if tb else y)
    check_fwd_bwd(getattr(matmuls, cls).apply, ref, a, b)


def test_dense_broadcast_and_mixed_ranks(mm):
    matmuls, fake = mm
    g = torch.Generator().manual_seed(12)
    # 3-d × 2-d (FC layer call shape, reference benchmarks/cublas_fc_layer.py:41) incl. a .t() view
    w = rand(g, 9, 6)
    check_fwd_bwd(matmuls.cublasMM.apply, torch.matmul, rand(g, 2, 4, 6), w.t())
    assert ("cublas_mmul", (False, False)) in fake.calls  # flattened to one 2-d product
    # 2-d × 3-d (reference matmuls.py:48-52 is wrong here; target = torch.matmul)
    check_fwd_bwd(matmuls.cublasMM.apply, torch.matmul, rand(g, 4, 6), rand(g, 3, 6, 5))
    # broadcast batch dims
    check_fwd_bwd(matmuls.cublasMM.apply, torch.matmul, rand(g, 2, 1, 4, 6), rand(g, 3, 6, 5))
    check_fwd_bwd(matmuls.cublasTransbMM.apply, lambda x, y: x @ y.transpose(-1, -2), rand(g, 3, 4, 6), rand(g, 5, 6))
    # matrix-vector forms (the reference prints and falls back to a @ b, matmuls.py:39-41)
    check_fwd_bwd(matmuls.cublasMM.apply, torch.matmul, rand(g, 6), rand(g, 6, 5))
    check_fwd_bwd(matmuls.cublasMM.apply, torch.matmul, rand(g, 4, 6), rand(g, 6))
    check_fwd_bwd(matmuls.cublasMM.apply, torch.matmul, rand(g, 2, 4, 6), rand(g, 6))


def test_needs_input_grad_gating(mm):
    matmuls, fake = mm
    g = torch.Generator().manual_seed(13)
    a, b = rand(g, 4, 6), rand(g, 6, 5)
    fake.calls.clear()
    check_fwd_bwd(matmuls.cublasMM.apply, torch.matmul, a, b, grad_a=True, grad_b=False)
    assert len([c for c in fake.calls if c[0] == "cublas_mmul"]) == 2  # forward + one gradient only
    check_fwd_bwd(matmuls.naiveSpMM.apply, torch.matmul, a, b, grad_a=False, grad_b=True)


def sparsify(g, *shape, density=0.3):
    return torch.rand(*shape, generator=g) * (torch.rand(*shape, generator=g) < density)


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("cls", ["naiveSpMM", "cusparseMM"])
def test_sparse_classes_all_ranks(mm, cls, fused, monkeypatch):
    """Dense inputs take the fused skip-zeros kernel when it covers the shape (fused=True) and the
    dense→CSR + batched-CSR route otherwise (fused=False: the stand-in declines, as the real module
    does for unsupported widths)."""
    matmuls, fake = mm
    monkeypatch.setattr(fake, "fused_dense", fused)
    g = torch.Generator().manual_seed(14)
    apply = getattr(matmuls, cls).apply
    check_fwd_bwd(apply, torch.matmul, sparsify(g, 7, 9), rand(g, 9, 5))                    # 2-d × 2-d
    check_fwd_bwd(apply, torch.matmul, sparsify(g, 7, 9), rand(g, 3, 9, 5))                 # 2-d × 3-d
    check_fwd_bwd(apply, torch.matmul, sparsify(g, 2, 4, 9), rand(g, 9, 5))                 # 3-d × 2-d (FC layer)
    fake.calls.clear()
    check_fwd_bwd(apply, torch.matmul, sparsify(g, 2, 3, 8, 8), rand(g, 2, 3, 8, 4))        # BERT-shaped batch
    if fused:
        assert [c[0] for c in fake.calls if c[0].startswith("naive_spmm") or c[0] == "dense_to_csr"] == ["naive_spmm_dense"]
    else:
        assert [c for c in fake.calls if c[0] == "naive_spmm_batched"] == [("naive_spmm_batched", (6, 8, 8))]
        assert len([c for c in fake.calls if c[0] == "dense_to_csr"]) == 1                   # one conversion
    check_fwd_bwd(apply, torch.matmul, sparsify(g, 3, 8, 8), rand(g, 1, 8, 4))              # broadcast batch
    check_fwd_bwd(apply, torch.matmul, sparsify(g, 9), rand(g, 9, 5))                       # vector × matrix
    check_fwd_bwd(apply, torch.matmul, sparsify(g, 7, 9), rand(g, 9))                       # matrix × vector


def test_sparse_csr_tensor_input_and_pattern_gradient(mm):
    """reference tests/cusparse_kernel_test.py:55-56: cusparseMM.apply(a.to_sparse_csr(), b)."""
    matmuls, fake = mm
    g = torch.Generator().manual_seed(15)
    a, b = sparsify(g, 10, 20, density=0.1), rand(g, 20, 10)
    out = matmuls.cusparseMM.apply(a.to_sparse_csr(), b)
    assert torch.allclose(a @ b, out, rtol=RTOL, atol=ATOL)
    # backward: grad wrt B through the CSR transpose; grad wrt A sampled on A's pattern
    a_csr = a.to_sparse_csr().requires_grad_(True)
    b1 = b.clone().requires_grad_(True)
    dc = rand(g, 10, 10)
    matmuls.naiveSpMM.apply(a_csr, b1).backward(dc)
    a2, b2 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    (a2 @ b2).backward(dc)
    assert torch.allclose(b2.grad, b1.grad, rtol=RTOL, atol=ATOL)
    assert a_csr.grad.is_sparse_csr
    assert torch.equal(a_csr.grad.crow_indices(), a_csr.crow_indices())
    assert torch.allclose((a2.grad * (a != 0)), a_csr.grad.to_dense(), rtol=RTOL, atol=ATOL)
    assert ("csr_transpose", (10, 20)) in fake.calls and ("sddmm", (10, 20)) in fake.calls


def test_csr_times_batched_operand_backward_and_transpose_cache(mm):
    """A CSR m1 against a batched m2 ([..., K, N]; reference call shape matmuls.py:245-256): forward
    and both gradients equal torch autograd of the dense product; the device transpose of A is built
    once per CSR tensor and reused by later backward passes until the values change in place."""
    matmuls, fake = mm
    g = torch.Generator().manual_seed(21)
    a = sparsify(g, 9, 14, density=0.3)
    for bshape in ((3, 14, 5), (2, 3, 14, 4)):
        b = rand(g, *bshape)
        a_csr = a.to_sparse_csr().requires_grad_(True)
        b1 = b.clone().requires_grad_(True)
        out = matmuls.naiveSpMM.apply(a_csr, b1)
        a2, b2 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
        ref = torch.matmul(a2, b2)
        assert out.shape == ref.shape and torch.allclose(ref, out, rtol=RTOL, atol=ATOL)
        dc = rand(g, *ref.shape)
        out.backward(dc)
        ref.backward(dc)
        assert torch.allclose(b2.grad, b1.grad, rtol=RTOL, atol=1e-6)
        assert torch.allclose(a2.grad * (a != 0), a_csr.grad.to_dense(), rtol=RTOL, atol=1e-6)
    # the transpose cache: second backward on the same CSR tensor does not transpose again
    a_csr = a.to_sparse_csr().requires_grad_(True)
    b = rand(g, 14, 6)
    n0 = sum(c[0] == "csr_transpose" for c in fake.calls)
    for _ in range(3):
        matmuls.cusparseMM.apply(a_csr, b.clone().requires_grad_(True)).sum().backward()
    assert sum(c[0] == "csr_transpose" for c in fake.calls) == n0 + 1
    with torch.no_grad():
        torch.Tensor.values(a_csr).mul_(2.0)  # in-place update of the values: the entry is rebuilt
    b3 = b.clone().requires_grad_(True)
    matmuls.cusparseMM.apply(a_csr, b3).sum().backward()
    assert sum(c[0] == "csr_transpose" for c in fake.calls) == n0 + 2
    assert torch.allclose(b3.grad, (2 * a).t() @ torch.ones(9, 6), rtol=RTOL, atol=1e-6)


def test_batched_csr_operand_backward_and_pattern_cache(mm):
    """A batched CSR tensor as the sparse operand (reference matmuls.py:289-293 recurses; no backward there): both
    gradients equal torch autograd of the dense product — per-item and shared dense operand — the gradient of the
    sparse operand comes back as a batched CSR tensor on its pattern, and the int32 pattern + the batched transpose are
    built once per tensor: a second backward transposes nothing."""
    matmuls, fake = mm
    g = torch.Generator().manual_seed(33)
    nb, M, K = 4, 7, 9
    keep = torch.zeros(nb, M * K, dtype=torch.bool)
    for i in range(nb):  # torch's batched CSR wants the same number of non-zeros in every item
        keep[i, torch.randperm(M * K, generator=g)[:20]] = True
    dense = (rand(g, nb, M, K) + 0.1) * keep.reshape(nb, M, K)
    # (the 1-d operand: batched CSR × vector — the forward promotes it to a one-column matrix, the backward must too)
    for b in (rand(g, nb, K, 5), rand(g, K, 6), rand(g, 2, 2, K, 3), rand(g, K)):
        shape = (2, 2, M, K) if b.dim() == 4 else (nb, M, K)
        a = dense.reshape(shape).to_sparse_csr().requires_grad_(True)
        n0 = sum(c[0] == "csr_transpose_batched" for c in fake.calls)
        for rep in range(2):
            a.grad = None
            b1 = b.clone().requires_grad_(True)
            out = matmuls.cusparseMM.apply(a, b1)
            a2, b2 = dense.reshape(shape).clone().requires_grad_(True), b.clone().requires_grad_(True)
            ref = torch.matmul(a2, b2)
            assert out.shape == ref.shape and torch.allclose(ref, out, rtol=RTOL, atol=1e-6)
            dc = rand(g, *ref.shape)
            out.backward(dc)
            ref.backward(dc)
            assert torch.allclose(b2.grad, b1.grad, rtol=RTOL, atol=1e-6)
            assert a.grad.is_sparse_csr and a.grad.shape == a.shape
            assert torch.allclose(a2.grad * keep.reshape(shape), a.grad.to_dense(), rtol=RTOL, atol=1e-6)
        assert sum(c[0] == "csr_transpose_batched" for c in fake.calls) == n0 + 1
    # the values reach the transposed product through the cached permutation inside the kernel where the plan takes one
    # (no gathered copy); where it does not, one index_select + the plain batched product — same gradients either way
    a = dense.to_sparse_csr().requires_grad_(True)
    grads = []
    for take in (True, False):
        fake.perm_plan = take
        try:
            del fake.calls[:]
            a.grad = None
            b1 = rand(torch.Generator().manual_seed(5), nb, K, 5).requires_grad_(True)
            matmuls.cusparseMM.apply(a, b1).backward(torch.ones(nb, M, 5))
            names = [c[0] for c in fake.calls]
            assert "naive_spmm_batched_perm" in names
            assert (names.count("naive_spmm_batched") == 1) == take  # forward only / forward + the gathered fallback
            grads.append(b1.grad.clone())
        finally:
            fake.perm_plan = True
    assert torch.equal(grads[0], grads[1])
    # the gradient of the stored values: the batched SDDMM where it takes the problem, else the block-diagonal one
    grads = []
    for take in (True, False):
        fake.batched_sddmm = take
        try:
            del fake.calls[:]
            a.grad = None
            b1 = rand(torch.Generator().manual_seed(5), nb, K, 5).requires_grad_(True)
            matmuls.cusparseMM.apply(a, b1).backward(torch.ones(nb, M, 5))
            names = [c[0] for c in fake.calls]
            assert "sddmm_batched" in names and ("sddmm" in names) == (not take)
            grads.append(a.grad.values().clone())
        finally:
            fake.batched_sddmm = True
    assert torch.equal(grads[0], grads[1])


def test_get_sparse_tensor_properties_contract(mm):
    """Argument order = naive_spmm / cusparse_mmul signature (reference matmuls.py:178-187)."""
    matmuls, _ = mm
    a = sparsify(torch.Generator().manual_seed(16), 6, 8)
    vals, cols, offs, nnz, rows, kcols = matmuls.get_sparse_tensor_properties(a.to_sparse_csr())
    assert vals.dtype == torch.float32 and cols.dtype == torch.int32 and offs.dtype == torch.int32
    assert (nnz, rows, kcols) == (int((a != 0).sum()), 6, 8) and offs.numel() == 7
    with pytest.raises(AssertionError):
        matmuls.get_sparse_tensor_properties(a)  # dense input: `assert a.is_sparse_csr`


def test_custom_mm_op_injection(mm):
    """mm_op / bmm_op remain injection points (reference matmuls.py:15-16,191,260)."""
    matmuls, _ = mm
    seen = []

    def my_mm(A, B, C, ta, tb):
        seen.append("mm")
        return C.copy_(A @ B)

    def my_spmm(vals, cols, offs, nnz, rows, kcols, B, C):
        seen.append("spmm")
        A = torch.sparse_csr_tensor(offs.long(), cols.long(), vals, (rows, kcols))
        return C.copy_(A @ B)

    g = torch.Generator().manual_seed(17)
    a, b = sparsify(g, 2, 5, 6), rand(g, 2, 6, 3)
    assert torch.allclose(matmuls.custom_matmul(a[0], b[0], mm_op=my_mm), a[0] @ b[0])
    assert torch.allclose(matmuls.naive_matmul(a, b, mm_op=my_spmm), a @ b, rtol=RTOL, atol=ATOL)
    assert seen == ["mm", "spmm", "spmm"]  # a caller's 2-d kernel is applied per slice


def test_golden_through_wrappers(mm, golden):
    matmuls, _ = mm
    for name in golden.cases("gemm"):
        c = golden.case(name)
        ta, tb = (bool(x) for x in c["flags"])
        cls = {(False, False): matmuls.cublasMM, (True, False): matmuls.cublasTransaMM,
               (False, True): matmuls.cublasTransbMM, (True, True): matmuls.cublasTransabMM}[(ta, tb)]
        a = torch.from_numpy(c["a"]).requires_grad_(True)
        b = torch.from_numpy(c["b"]).requires_grad_(True)
        out = cls.apply(a, b)
        assert np.allclose(out.detach().numpy(), c["c"], rtol=RTOL, atol=ATOL), name
        out.backward(torch.from_numpy(c["dc"]))
        assert np.allclose(a.grad.numpy(), c["grad_a"], rtol=RTOL, atol=ATOL), name
        assert np.allclose(b.grad.numpy(), c["grad_b"], rtol=RTOL, atol=ATOL), name
    c = golden.case("batched/bert")
    a = torch.from_numpy(c["a"]).requires_grad_(True)
    b = torch.from_numpy(c["b"]).requires_grad_(True)
    out = matmuls.naiveSpMM.apply(a, b)
    assert np.allclose(out.detach().numpy(), c["c"], rtol=RTOL, atol=ATOL)
    out.backward(torch.from_numpy(c["dc"]))
    assert np.allclose(a.grad.numpy(), c["grad_a"], rtol=RTOL, atol=ATOL)
    assert np.allclose(b.grad.numpy(), c["grad_b"], rtol=RTOL, atol=ATOL)


def test_fc_layer_modules_host_logic(mm):
    """cublasLinear / cusparseLinear (reference benchmarks/*_fc_layer.py) against nn.Linear with the
    same parameters: forward with the fused bias epilogue, backward for input, weight and bias."""
    matmuls, fake = mm
    sys.modules.pop("fc_layers", None)
    import fc_layers
    _model = fc_layers.sparse_forward_pays
    g = torch.Generator().manual_seed(31)
    # the cost model of cusparseLinear: ReLU-sparse wide layers go sparse, dense-ish or narrow ones do not
    assert fc_layers.sparse_forward_pays(int(0.01 * 16384 * 3072), 16384, 3072, 768)
    assert not fc_layers.sparse_forward_pays(int(0.5 * 16384 * 3072), 16384, 3072, 768)
    assert not fc_layers.sparse_forward_pays(int(0.1 * 16384 * 3072), 16384, 3072, 768)
    assert fc_layers.sparse_forward_pays(int(0.05 * 4096 * 4096), 4096, 4096, 4096)
    assert not fc_layers.sparse_forward_pays(10, 15, 12, 7)
    for cls, force_sparse in ((fc_layers.cublasLinear, False), (fc_layers.cusparseLinear, True),
                              (fc_layers.cusparseLinear, False)):
        fc_layers.sparse_forward_pays = (lambda *a: True) if force_sparse else _model
        fc_layers.worth_sampling = lambda *a: True  # tiny test layers: take the decision path anyway
        for bias in (True, False):
            layer = cls(12, 7, bias=bias)
            ref = torch.nn.Linear(12, 7, bias=bias)
            ref.load_state_dict(layer.state_dict())
            x = torch.rand(3, 5, 12, generator=g) * (torch.rand(3, 5, 12, generator=g) < 0.6)
            x1, x2 = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
            y, yr = layer(x1), ref(x2)
            assert y.shape == yr.shape and torch.allclose(yr, y, rtol=RTOL, atol=1e-6)
            dy = torch.rand(yr.shape, generator=g)
            y.backward(dy)
            yr.backward(dy)
            assert torch.allclose(x2.grad, x1.grad, rtol=RTOL, atol=1e-6)
            assert torch.allclose(ref.weight.grad, layer.weight.grad, rtol=RTOL, atol=1e-6)
            if bias:
                assert torch.allclose(ref.bias.grad, layer.bias.grad, rtol=RTOL, atol=1e-6)
        assert "in_features=12, out_features=7" in repr(layer)
        assert layer(torch.rand(2, 5)) == 0  # wrong width: prints and returns 0, like the reference
    fc_layers.sparse_forward_pays = _model
    names = [c[0] for c in fake.calls]
    assert "cublas_mmul_bias" in names and "naive_spmm_bias_ex" in names and "dense_to_csr_fill" in names


def test_dense_inputs_are_routed_by_shape(mm, monkeypatch):
    """A dense A with zeros goes to the in-kernel zero-skipping product only where it beats dense→CSR + the CSR
    kernels (one small matrix, batches of small / medium matrices) and to the CSR route for one large matrix
    (the FC-layer call shape flattens to that) and for very long rows — tools/bench_skipwide.py."""
    matmuls, fake = mm
    assert matmuls.fused_skip_pays(1, 512, 512, 64) and matmuls.fused_skip_pays(1, 512, 512, 256)
    assert not matmuls.fused_skip_pays(1, 1024, 1024, 1024) and not matmuls.fused_skip_pays(1, 512, 512, 512)
    assert not matmuls.fused_skip_pays(1, 2048, 2048) and not matmuls.fused_skip_pays(1, 16384, 768)
    assert matmuls.fused_skip_pays(384, 512, 512) and matmuls.fused_skip_pays(16, 2048, 2048)
    assert not matmuls.fused_skip_pays(4, 4096, 4096)
    # dense-with-zeros → the MFMA product above the crossover (profiles/r03_dense_input_routing.log): BERT-base
    # probs·V pays at 100 % and 10 % kept, not at 1 %; the FC call shape pays at 100 % and 10 %, not at 2 %
    assert matmuls.dense_route_pays(1.0, 384, 512, 512, 64) and matmuls.dense_route_pays(0.1, 384, 512, 512, 64)
    assert not matmuls.dense_route_pays(0.01, 384, 512, 512, 64)
    assert matmuls.dense_route_pays(1.0, 1, 16384, 768, 3072) and matmuls.dense_route_pays(0.1, 1, 16384, 768, 3072)
    assert not matmuls.dense_route_pays(0.02, 1, 16384, 768, 3072)
    assert not matmuls.dense_route_pays(0.0, 1, 4096, 4096, 4096)
    monkeypatch.setattr(fake, "fused_dense", True)
    g = torch.Generator().manual_seed(3)

    def route(a, b):
        fake.calls.clear()
        out = matmuls.naiveSpMM.apply(a, b)
        assert torch.allclose(out, torch.matmul(a, b), rtol=1e-5, atol=1e-6)
        return [c[0] for c in fake.calls if c[0].startswith("naive_spmm") or c[0] == "dense_to_csr"]

    assert route(sparsify(g, 40, 30), rand(g, 30, 8)) == ["naive_spmm_dense"]                # one small matrix
    monkeypatch.setattr(matmuls, "fused_skip_pays", lambda items, rows, cols, width=256: items > 1 and cols <= 16)
    assert route(sparsify(g, 40, 30), rand(g, 30, 8)) == ["dense_to_csr", "naive_spmm_ex"]      # "large": CSR route
    assert route(sparsify(g, 3, 20, 30), rand(g, 30, 8)) == ["dense_to_csr", "naive_spmm_ex"]   # FC call shape: flattened
    assert route(sparsify(g, 3, 8, 12), rand(g, 3, 12, 4)) == ["naive_spmm_dense"]           # batch of small matrices
    assert route(sparsify(g, 3, 8, 20), rand(g, 3, 20, 4)) == ["dense_to_csr", "naive_spmm_batched"]  # rows "too long"


def test_csr_operand_is_narrowed_once_and_skips_the_long_row_machinery(mm, monkeypatch):
    """A CSR tensor's int32 arrays and its longest row are kept on the tensor between products (rebuilt when the
    values change in place); without over-long rows the stock kernel is called as naive_spmm_ex(rule 0): one
    launch, no workspace."""
    matmuls, fake = mm
    g = torch.Generator().manual_seed(8)
    a = sparsify(g, 12, 20).to_sparse_csr()
    b = rand(g, 20, 6)
    seen = []
    real = matmuls.get_sparse_tensor_properties
    monkeypatch.setattr(matmuls, "get_sparse_tensor_properties", lambda t: (seen.append(1), real(t))[1])
    fake.calls.clear()
    out1 = matmuls.naiveSpMM.apply(a, b)
    out2 = matmuls.cusparseMM.apply(a, b)
    assert len(seen) == 1                                     # narrowed once
    assert [c[0] for c in fake.calls if c[0].startswith("naive_spmm") or c[0] == "cusparse_mmul"] == ["naive_spmm_ex", "naive_spmm_ex"]
    assert torch.allclose(out1, a.to_dense() @ b, rtol=1e-5, atol=1e-6) and torch.equal(out1, out2)
    torch.Tensor.values(a).mul_(2.0)                          # in-place update: the entry is rebuilt
    out3 = matmuls.naiveSpMM.apply(a, b)
    assert len(seen) == 2 and torch.allclose(out3, 2 * out1, rtol=1e-5, atol=1e-6)
    monkeypatch.setattr(fake, "long_row_threshold", lambda: 3)   # "long" rows present: the stock entry with its workspace
    fake.calls.clear()
    matmuls.naiveSpMM.apply(a, b)
    assert [c[0] for c in fake.calls if c[0].startswith("naive_spmm")] == ["naive_spmm"]
