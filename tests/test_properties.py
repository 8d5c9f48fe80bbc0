"""Property tests (hypothesis) on the CPU side: the oracle against torch.matmul over random shapes
and sparsity, and matmuls.py's dispatch (through the oracle-backed stand-in) against torch for
arbitrary broadcastable ranks — the shapes a drop-in has to survive, beyond the fixed lists."""
import importlib
import sys

import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings, strategies as st

RTOL, ATOL = 1e-5, 1e-8
SETTINGS = dict(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])


def _rand(shape, seed, density=1.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(shape, generator=g)
    if density < 1.0:
        x = x * (torch.rand(shape, generator=g) < density)
    return x


@settings(**SETTINGS)
@given(m=st.integers(0, 40), k=st.integers(0, 70), n=st.integers(0, 300), density=st.sampled_from([0.0, 0.05, 0.3, 1.0]),
       seed=st.integers(0, 2 ** 16))
def test_oracle_spmm_equals_torch(oracle_mod, m, k, n, density, seed):
    a, b = _rand((m, k), seed, density), _rand((k, n), seed + 1)
    rp, col, val = oracle_mod.dense_to_csr(a.numpy())
    got = oracle_mod.spmm_csr(rp.reshape(-1), col, val, m, k, b.numpy())
    exp = torch.matmul(a, b).numpy()
    assert got.shape == exp.shape and np.allclose(got, exp, rtol=RTOL, atol=ATOL)
    # transpose round trip and the column-major form are consistent with it
    t_rp, t_col, t_val = oracle_mod.csr_transpose(rp.reshape(-1), col, val, m, k)
    rp2, col2, val2 = oracle_mod.csr_transpose(t_rp, t_col, t_val, k, m)
    assert np.array_equal(rp2, rp.reshape(-1)) and np.array_equal(col2, col) and np.array_equal(val2, val)
    if m and n:
        cm = oracle_mod.spmm_csr_colmajor(rp.reshape(-1), col, val, m, k, n, np.ascontiguousarray(b.numpy().T))
        assert np.array_equal(cm.reshape(n, m), got.T)


@settings(**SETTINGS)
@given(m=st.integers(1, 33), k=st.integers(1, 40), n=st.integers(1, 33), ta=st.booleans(), tb=st.booleans(),
       batch=st.sampled_from([(), (2,), (2, 3)]), seed=st.integers(0, 2 ** 16))
def test_oracle_gemm_equals_torch(oracle_mod, m, k, n, ta, tb, batch, seed):
    a = _rand(batch + ((k, m) if ta else (m, k)), seed)
    b = _rand(batch + ((n, k) if tb else (k, n)), seed + 1)
    exp = torch.matmul(a.transpose(-1, -2) if ta else a, b.transpose(-1, -2) if tb else b).numpy()
    assert np.allclose(oracle_mod.gemm(a.numpy(), b.numpy(), ta, tb), exp, rtol=RTOL, atol=ATOL)


@pytest.fixture()
def mm(oracle_mod):
    import fake_custom_mm
    saved = {k: sys.modules.get(k) for k in ("custom_mm", "matmuls")}
    sys.modules["custom_mm"] = fake_custom_mm
    sys.modules.pop("matmuls", None)
    matmuls = importlib.import_module("matmuls")
    yield matmuls
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


shapes = st.sampled_from([
    # (a_shape, b_shape): every rank combination torch.matmul accepts, with broadcasting
    ((5, 4), (4, 3)), ((4,), (4, 3)), ((5, 4), (4,)), ((4,), (4,)), ((2, 5, 4), (4, 3)), ((5, 4), (2, 4, 3)),
    ((2, 5, 4), (2, 4, 3)), ((1, 5, 4), (3, 4, 3)), ((2, 3, 5, 4), (2, 3, 4, 3)), ((2, 1, 5, 4), (3, 4, 3)),
    ((2, 2, 2, 5, 4), (2, 2, 2, 4, 3)), ((3, 5, 4), (4,)), ((4,), (3, 4, 2)),
])


@settings(**SETTINGS)
@given(pair=shapes, which=st.sampled_from(["cublasMM", "naiveSpMM", "cusparseMM"]), density=st.sampled_from([0.3, 1.0]),
       seed=st.integers(0, 2 ** 16), fused=st.booleans())
def test_wrappers_follow_torch_matmul_for_every_rank(mm, pair, which, density, seed, fused):
    import fake_custom_mm
    fake_custom_mm.fused_dense = fused
    try:
        a_shape, b_shape = pair
        a, b = _rand(a_shape, seed, density), _rand(b_shape, seed + 1)
        a1, b1 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
        a2, b2 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
        out, exp = getattr(mm, which).apply(a1, b1), torch.matmul(a2, b2)
        assert out.shape == exp.shape and torch.allclose(exp, out, rtol=RTOL, atol=ATOL)
        dc = _rand(tuple(exp.shape), seed + 2) if exp.dim() else torch.tensor(0.7)
        out.backward(dc)
        exp.backward(dc)
        assert torch.allclose(a2.grad, a1.grad, rtol=RTOL, atol=ATOL) and a1.grad.shape == a.shape
        assert torch.allclose(b2.grad, b1.grad, rtol=RTOL, atol=ATOL) and b1.grad.shape == b.shape
    finally:
        fake_custom_mm.fused_dense = True
