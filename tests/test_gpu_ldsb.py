"""MI_SPMM_LDS_B and the LDS-resident SDDMM (pruned attention in batched CSR form), gather_perm.

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


@pytest.mark.parametrize("N", [4, 8, 16, 32, 64, 96, 128, 256])
def test_spmm_lds_resident_b_plan_bit_exact(capi, cmm, dev, oracle_mod, N):
    """MI_SPMM_LDS_B (spmm_ldsb.hip): an item's whole B copied into LDS, rows gather from there — every group width
    (DPP row broadcasts for 8 / 16 / 32 lanes, readlane for 64, ds_bpermute below), batched with per-item and shared
    B, shuffled columns, duplicates, empty rows, a row count that is not a multiple of anything: bit-identical to the
    oracle's batched product and to the row-split group kernel.  Reference: the per-slice recursion of naive_matmul,
    matmuls.py:282-297."""
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    capi.mi_spmm_csr_batched_variant_f32.argtypes = [ctypes.c_int, vp, vp, vp, i64, i32, i32, i32, i32, vp, i64, i64, vp, i64, i64, vp]
    stream = torch.cuda.current_stream().cuda_stream
    g = np.random.Generator(np.random.PCG64(N))
    K = min(300, (128 * 1024) // (4 * N))
    if N in (64, 128, 256):
        K = 2 * (128 * 1024) // (4 * N) - 3   # B no longer fits as a whole: two column tiles
    for batch, M, share_b in ((1, 777, False), (5, 333, False), (7, 130, True)):
        lens = g.integers(0, 70, size=batch * M)
        lens[g.integers(0, batch * M, size=5)] = 0
        lens[3] = 2 * K + 5                                     # longer than K: duplicate columns
        cols = [g.integers(0, K, size=int(n)).astype(np.int32) for n in lens]
        cols = [c if i % 3 else np.sort(c) for i, c in enumerate(cols)]
        col = np.concatenate(cols)
        val = g.random(len(col), dtype=np.float32) - 0.5
        off = np.zeros((batch, M + 1), np.int64)
        off[:, 1:] = np.cumsum(lens).reshape(batch, M)
        off[1:, 0] = off[:-1, M]
        off = off.astype(np.int32)
        B = g.random((K, N) if share_b else (batch, K, N), dtype=np.float32) - 0.5
        want = oracle_mod.spmm_csr_batched(off, col, val, batch, M, K, B)
        d_off, d_col, d_val, d_B = t(off, dev), t(col, dev), t(val, dev), t(B, dev)
        outs = {}
        for variant in (18, 5):
            C = torch.full((batch, M, N), float("nan"), device=dev)
            st = capi.mi_spmm_csr_batched_variant_f32(variant, d_off.data_ptr(), d_col.data_ptr(), d_val.data_ptr(), len(col),
                                                      batch, M, K, N, d_B.data_ptr(), N, 0 if share_b else K * N, C.data_ptr(),
                                                      N, M * N, stream)
            assert st == 0, (variant, batch, M, K, N)
            outs[variant] = C.cpu().numpy()
        assert np.array_equal(outs[18], want), (batch, M, K, N)
        assert np.array_equal(outs[18], outs[5])


@pytest.mark.parametrize("N", [16, 32, 64, 128, 256])
def test_spmm_lds_resident_b_quad_form_bit_exact(capi, cmm, dev, oracle_mod, N):
    """The quad form of MI_SPMM_LDS_B (spmm_ldsq_kernel: four lanes per row, 16-byte loads of col / val clamped to the
    arrays' last 16 bytes, LDS-DMA staging, 256-row steps) pinned through mi_spmm_ldsb_set_form, beside the 16-lane
    form and the oracle: column tiles of 64 (N = 128, 256), K from a few rows to the 512 the image holds, row counts
    below / across / beyond a 256-row step, empty rows, duplicate and unsorted columns, a last row of 1 … 3 entries
    (the lane whose clamped load starts early), per-item and shared B; the values through a permutation; bias and the
    long-row rule on one item.  Column tiles of 64, 32 or 16 by the height of B (up to 2048 rows at 16 columns:
    attention over 2048 tokens, which only this form keeps in LDS — pinned off, the product falls back to the L2
    gathers with the same bits).  Reference: the per-slice recursion of naive_matmul, matmuls.py:282-297."""
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    capi.mi_spmm_csr_batched_variant_f32.argtypes = [ctypes.c_int, vp, vp, vp, i64, i32, i32, i32, i32, vp, i64, i64, vp, i64, i64, vp]
    capi.mi_spmm_ldsb_set_form.argtypes = [ctypes.c_int]
    stream = torch.cuda.current_stream().cuda_stream
    g = np.random.Generator(np.random.PCG64(1000 + N))
    try:
        k_tall = {16: 2000, 32: 2000, 64: 2000, 128: 1000, 256: 509}[N]   # narrower tiles: 16 / 16 / 16 / 32 / 64 columns
        for case, (batch, M, K, share_b, tail) in enumerate(((1, 777, 512, False, 1), (5, 333, 300, False, 2), (7, 130, 17, True, 3),
                                                             (3, 9, 512, False, 1), (2, 256, 64, False, 0), (40, 512, 128, False, 2),
                                                             (3, 300, k_tall, False, 3), (2, 515, 1024 if N <= 64 else 700, True, 1))):
            lens = g.integers(0, 70, size=batch * M)
            lens[g.integers(0, batch * M, size=5)] = 0
            lens[3] = 2 * K + 5                                     # longer than K: duplicate columns
            lens[-1] = tail                                          # the arrays end 0 … 3 entries into a 16-byte load
            cols = [g.integers(0, K, size=int(n)).astype(np.int32) for n in lens]
            cols = [c if i % 3 else np.sort(c) for i, c in enumerate(cols)]
            col = np.concatenate(cols)
            val = g.random(len(col), dtype=np.float32) - 0.5
            off = np.zeros((batch, M + 1), np.int64)
            off[:, 1:] = np.cumsum(lens).reshape(batch, M)
            off[1:, 0] = off[:-1, M]
            off = off.astype(np.int32)
            B = g.random((K, N) if share_b else (batch, K, N), dtype=np.float32) - 0.5
            want = oracle_mod.spmm_csr_batched(off, col, val, batch, M, K, B)
            d_off, d_col, d_val, d_B = t(off, dev), t(col, dev), t(val, dev), t(B, dev)
            for form in (1, 0, -1):
                assert capi.mi_spmm_ldsb_set_form(form) == 0
                C = torch.full((batch, M, N), float("nan"), device=dev)
                st = capi.mi_spmm_csr_batched_variant_f32(18, d_off.data_ptr(), d_col.data_ptr(), d_val.data_ptr(), len(col),
                                                          batch, M, K, N, d_B.data_ptr(), N, 0 if share_b else K * N,
                                                          C.data_ptr(), N, M * N, stream)
                assert st == 0, (form, batch, M, K, N)
                assert np.array_equal(C.cpu().numpy().view(np.int32), want.view(np.int32)), (form, case, batch, M, K, N)
            capi.mi_spmm_csr_batched_f32_plan.argtypes = [i64, i32, i32, i32, i32, vp, i64, i64, vp, i64, i64]
            if not share_b and capi.mi_spmm_csr_batched_f32_plan(len(col), batch, M, K, N, None, N, K * N, None, N, M * N) == 18:
                # (what AUTO hands the permuted entry point)
                capi.mi_spmm_ldsb_set_form(1)
                shuffle = g.permutation(len(val))
                stored = np.empty_like(val)
                stored[shuffle] = val
                C = torch.full((batch, M, N), float("nan"), device=dev)
                took = cmm.naive_spmm_batched_perm(t(stored, dev), t(shuffle.astype(np.int32), dev), d_col, d_off, len(col),
                                                   batch, M, K, d_B, C)
                # (taken where B goes in as ONE tile; a column-tiled B would gather every value once per tile: declined,
                # matmuls then gathers once — custom_mm.gather_perm — and runs the plain product)
                assert took is (N <= 64 and K <= 512)
                if took:
                    assert np.array_equal(C.cpu().numpy().view(np.int32), want.view(np.int32))
                else:
                    gathered = cmm.gather_perm(t(stored, dev), t(shuffle.astype(np.int32), dev))
                    assert torch.equal(gathered, d_val)
        # one item, a row beyond the long-row threshold (skipped by the kernel, summed by the follow-up), fused bias
        capi.mi_spmm_ldsb_set_form(1)
        M, K = 20000, 256
        lens = g.integers(10, 40, size=M)   # (long enough for AUTO to keep the LDS plan at every N here: bias + column tiles)
        lens[77], lens[19999] = 9000, 3
        col = np.concatenate([g.integers(0, K, size=int(n)).astype(np.int32) for n in lens])
        rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        val = g.random(len(col), dtype=np.float32) - 0.5
        B, bias = g.random((K, N), dtype=np.float32) - 0.5, g.random(N, dtype=np.float32)
        want = oracle_mod.spmm_csr_long(rowptr, col, val, M, K, B)
        capi.mi_spmm_csr_f32_plan.argtypes = [i64, i32, i32, i32, vp, i64, vp, i64]
        assert capi.mi_spmm_csr_f32_plan(len(col), M, K, N, None, N, None, N) == 18
        C = torch.full((M, N), float("nan"), device=dev)
        cmm.naive_spmm_bias(t(val, dev), t(col, dev), t(rowptr, dev), len(col), M, K, t(B, dev), t(bias, dev), C)
        assert np.array_equal(C.cpu().numpy(), want + bias[None, :])
    finally:
        capi.mi_spmm_ldsb_set_form(-1)


def test_lds_resident_b_keeps_negative_zero_and_reads_values_through_a_permutation(capi, cmm, dev, oracle_mod):
    """Round 4.  (1) Advisor: a row whose products all underflow negatively ends as −0 in the oracle and in every plan;
    the LDS-resident-B kernel pads a row's last four-entry step — with value −0 on an all-zero row, which leaves every
    accumulator's bits (padding with +0 turned −0 into +0).  Compared as raw bits.  (2) The batched product with the
    values read through a permutation (custom_mm.naive_spmm_batched_perm: what the backward of a batched CSR tensor
    uses instead of a gathered copy of the values) is the plain product bit for bit, and reports False — launching
    nothing — on a problem whose plan takes no permutation."""
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    capi.mi_spmm_csr_batched_variant_f32.argtypes = [ctypes.c_int, vp, vp, vp, i64, i32, i32, i32, i32, vp, i64, i64, vp, i64, i64, vp]
    stream = torch.cuda.current_stream().cuda_stream
    g = np.random.Generator(np.random.PCG64(44))
    batch, M, K, N = 64, 256, 256, 64
    lens = g.integers(1, 60, size=batch * M)
    col = np.concatenate([np.sort(g.choice(K, int(n), replace=False)) for n in lens]).astype(np.int32)
    val = g.random(len(col), dtype=np.float32) - 0.5
    off = np.zeros((batch, M + 1), np.int64)
    off[:, 1:] = np.cumsum(lens).reshape(batch, M)
    off[1:, 0] = off[:-1, M]
    off = off.astype(np.int32)
    B = g.random((batch, K, N), dtype=np.float32) + 0.5
    # rows 0, 5 and 9 of every item: tiny negative values against tiny positive entries of B → every product underflows
    # to −0 (row lengths 1 … 59: every tail length of the four-entry steps occurs)
    rows = np.repeat(np.arange(batch * M), lens)
    tiny = np.isin(rows % M, (0, 5, 9))
    val[tiny] = -1e-30
    B[:, :, 7] = 1e-30                      # column 7: −1e-30 · 1e-30 underflows for the tiny rows …
    want = oracle_mod.spmm_csr_batched(off, col, val, batch, M, K, B)
    assert np.signbit(want[0, 0, 7]) and want[0, 0, 7] == 0.0   # … to −0 in the oracle's chain
    d_off, d_col, d_val, d_B = t(off, dev), t(col, dev), t(val, dev), t(B, dev)
    for variant in (18, 5, 0):
        C = torch.full((batch, M, N), float("nan"), device=dev)
        st = capi.mi_spmm_csr_batched_variant_f32(variant, d_off.data_ptr(), d_col.data_ptr(), d_val.data_ptr(), len(col),
                                                  batch, M, K, N, d_B.data_ptr(), N, K * N, C.data_ptr(), N, M * N, stream)
        assert st == 0
        assert np.array_equal(C.cpu().numpy().view(np.int32), want.view(np.int32)), f"variant {variant}: bits differ (−0?)"
    # (2) the same product with the values stored in another order and a permutation leading to them
    shuffle = g.permutation(len(val))
    stored = np.empty_like(val)
    stored[shuffle] = val                    # stored[shuffle[p]] = val[p]
    C = torch.full((batch, M, N), float("nan"), device=dev)
    took = cmm.naive_spmm_batched_perm(t(stored, dev), t(shuffle.astype(np.int32), dev), d_col, d_off, len(col), batch, M, K,
                                       d_B, C)
    assert took is True and np.array_equal(C.cpu().numpy().view(np.int32), want.view(np.int32))
    small = torch.full((2, 3, 8), -7.0, device=dev)
    took = cmm.naive_spmm_batched_perm(torch.ones(4, device=dev), torch.arange(4, dtype=torch.int32, device=dev),
                                       torch.zeros(4, dtype=torch.int32, device=dev),
                                       torch.tensor([[0, 1, 2, 2], [2, 3, 4, 4]], dtype=torch.int32, device=dev), 4, 2, 3, 5,
                                       torch.rand(5, 8, device=dev), small)
    assert took is False and bool((small == -7.0).all())


def test_gather_perm_is_index_select(capi, cmm, dev):
    """custom_mm.gather_perm / mi_gather_f32: values[perm] — what carries a CSR tensor's values into its cached transposed
    pattern in matmuls' backward (the reference has no such step: its backward densifies, matmuls.py:245-256).  Any
    length (the last lanes take single entries), any alignment of the operands, −0 / inf / nan moved as bits."""
    g = torch.Generator(device=dev).manual_seed(5)
    for n in (1, 3, 4, 5, 1023, 1024, 100_003):
        src = torch.randn(n + 7, device=dev, generator=g)
        src[0], src[-1] = float("inf"), -0.0
        if n > 4:
            src[2] = float("nan")
        perm = torch.randint(0, n + 7, (n,), device=dev, generator=g, dtype=torch.int32)
        got = cmm.gather_perm(src, perm)
        assert torch.equal(got.view(torch.int32), src.index_select(0, perm.long()).view(torch.int32))
        # through the C-ABI with operands that are not 16-byte aligned (the scalar form of the kernel)
        vp, i64 = ctypes.c_void_p, ctypes.c_int64
        capi.mi_gather_f32.argtypes = [vp, vp, i64, vp, vp]
        buf_p = torch.empty(n + 1, dtype=torch.int32, device=dev)
        buf_p[1:] = perm
        out = torch.full((n + 1,), 7.0, device=dev)
        st = capi.mi_gather_f32(src.data_ptr(), buf_p.data_ptr() + 4, n, out.data_ptr() + 4,
                                torch.cuda.current_stream().cuda_stream)
        assert st == 0 and float(out[0]) == 7.0
        assert torch.equal(out[1:].view(torch.int32), got.view(torch.int32))
    assert capi.mi_gather_f32(None, None, 0, None, None) == 0 and capi.mi_gather_f32(None, None, 5, None, None) < 0


@pytest.mark.parametrize("N,shared", [(64, False), (64, True), (32, False), (48, False), (16, False), (8, True), (4, False)])
def test_sddmm_batched_lds_resident_form_bit_exact(capi, cmm, dev, oracle_mod, N, shared):
    """Round 4: custom_mm.sddmm_batched — the gradient of a batched CSR tensor's stored values with the item's dense
    operand resident in LDS — against the oracle's SDDMM per item and, bit for bit, against custom_mm.sddmm on the
    block-diagonal matrix of the batch (what matmuls ran before and still runs where the form does not apply): rows of
    0 … 150 entries (every tail length of the G-entry chunks), shuffled columns, duplicates, empty rows, an item count
    that leaves the last workgroups short; a problem the form does not take reports False and writes nothing."""
    g = np.random.Generator(np.random.PCG64(100 + N))
    batch, M, K = 70, 260, 300
    lens = g.integers(0, 150, size=batch * M)
    lens[g.integers(0, batch * M, size=50)] = 0
    cols = [g.integers(0, K, size=int(n)).astype(np.int32) for n in lens]
    cols = [c if i % 3 else np.sort(c) for i, c in enumerate(cols)]
    col = np.concatenate(cols)
    off = np.zeros((batch, M + 1), np.int64)
    off[:, 1:] = np.cumsum(lens).reshape(batch, M)
    off[1:, 0] = off[:-1, M]
    off = off.astype(np.int32)
    dC = g.random((batch, M, N), dtype=np.float32) - 0.5
    B = g.random((K, N) if shared else (batch, K, N), dtype=np.float32) - 0.5
    out = torch.full((len(col),), float("nan"), device=dev)
    took = cmm.sddmm_batched(t(col, dev), t(off, dev), len(col), batch, M, K, t(dC, dev), t(B, dev), out)
    assert took is True
    got = out.cpu().numpy()
    for b in (0, 1, batch // 2, batch - 1):
        s0, s1 = int(off[b, 0]), int(off[b, M])
        want = oracle_mod.sddmm((off[b] - s0).astype(np.int32), col[s0:s1], M, dC[b], B if shared else B[b])
        assert np.array_equal(got[s0:s1].view(np.int32), want.view(np.int32)), (N, b)
    # the block-diagonal form on the whole batch
    flat_off = np.concatenate([off[:, :-1].reshape(-1), off[-1:, -1]]).astype(np.int32)
    diag_col = (col.astype(np.int64) + np.repeat(np.arange(batch), np.diff(off, axis=1).sum(1)) * K).astype(np.int32)
    b_stack = np.ascontiguousarray(np.broadcast_to(B, (batch, K, N)).reshape(batch * K, N))
    ref = cmm.sddmm(t(diag_col, dev), t(flat_off, dev), len(col), batch * M, batch * K, t(dC.reshape(batch * M, N), dev),
                    t(b_stack, dev))
    assert torch.equal(out.view(torch.int32), ref.view(torch.int32))
    if N == 64:
        # the default above was the quad form (sddmm_ldsq_kernel: N = 64, K ≤ 512); the 16-lane form pinned beside it,
        # and a batch whose last row holds the arrays' last 1 … 3 entries (the lane whose 16-byte load is clamped)
        capi.mi_spmm_ldsb_set_form.argtypes = [ctypes.c_int]
        try:
            capi.mi_spmm_ldsb_set_form(0)
            out16 = torch.full((len(col),), float("nan"), device=dev)
            assert cmm.sddmm_batched(t(col, dev), t(off, dev), len(col), batch, M, K, t(dC, dev), t(B, dev), out16) is True
            assert torch.equal(out.view(torch.int32), out16.view(torch.int32))
            for tail in (1, 2, 3):
                cut = len(col) - int(lens[-1]) + tail if lens[-1] >= tail else None
                if cut is None:
                    continue
                off2 = off.copy()
                off2[-1, -1] = cut
                capi.mi_spmm_ldsb_set_form(1)
                o1 = torch.full((cut,), float("nan"), device=dev)
                assert cmm.sddmm_batched(t(col[:cut], dev), t(off2, dev), cut, batch, M, K, t(dC, dev), t(B, dev), o1) is True
                assert torch.equal(o1.view(torch.int32), out[:cut].view(torch.int32)), tail
        finally:
            capi.mi_spmm_ldsb_set_form(-1)
        # B beyond the 512 rows the image holds (attention over 1024 / 2048 tokens): 2 / 4 / 8 row tiles of B, a pass each —
        # an entry needs one row of B, so this is exact for any order of columns inside a row (sorted and shuffled rows below)
        for K2 in (1000, 2049, 3000):
            b2, M2 = 66, 257
            lens2 = g.integers(0, 120, size=b2 * M2)
            cols2 = [g.integers(0, K2, size=int(n)).astype(np.int32) for n in lens2]
            cols2 = [c if i % 4 == 0 else np.sort(c) for i, c in enumerate(cols2)]
            col2 = np.concatenate(cols2)
            off2 = np.zeros((b2, M2 + 1), np.int64)
            off2[:, 1:] = np.cumsum(lens2).reshape(b2, M2)
            off2[1:, 0] = off2[:-1, M2]
            off2 = off2.astype(np.int32)
            dC2 = g.random((b2, M2, N), dtype=np.float32) - 0.5
            B2 = g.random((K2, N) if shared else (b2, K2, N), dtype=np.float32) - 0.5
            out2 = torch.full((len(col2),), float("nan"), device=dev)
            assert cmm.sddmm_batched(t(col2, dev), t(off2, dev), len(col2), b2, M2, K2, t(dC2, dev), t(B2, dev), out2) is True
            got2 = out2.cpu().numpy()
            for b in (0, b2 - 1):
                a0, a1 = int(off2[b, 0]), int(off2[b, M2])
                want = oracle_mod.sddmm((off2[b] - a0).astype(np.int32), col2[a0:a1], M2, dC2[b], B2 if shared else B2[b])
                assert np.array_equal(got2[a0:a1].view(np.int32), want.view(np.int32)), (K2, b)
            flat2 = np.concatenate([off2[:, :-1].reshape(-1), off2[-1:, -1]]).astype(np.int32)
            diag2 = (col2.astype(np.int64) + np.repeat(np.arange(b2), np.diff(off2, axis=1).sum(1)) * K2).astype(np.int32)
            stack2 = np.ascontiguousarray(np.broadcast_to(B2, (b2, K2, N)).reshape(b2 * K2, N))
            ref2 = cmm.sddmm(t(diag2, dev), t(flat2, dev), len(col2), b2 * M2, b2 * K2, t(dC2.reshape(b2 * M2, N), dev), t(stack2, dev))
            assert torch.equal(out2.view(torch.int32), ref2.view(torch.int32)), K2
    # not taken: too few rows / an operand beyond the LDS image
    small = torch.full((6,), -7.0, device=dev)
    assert cmm.sddmm_batched(torch.zeros(6, dtype=torch.int32, device=dev),
                             torch.tensor([[0, 3], [3, 6]], dtype=torch.int32, device=dev), 6, 2, 1, 5,
                             torch.rand(2, 1, 8, device=dev), torch.rand(5, 8, device=dev), small) is False
    assert bool((small == -7.0).all())


def test_spmm_lds_resident_b_is_autos_choice_for_pruned_attention_and_keeps_the_long_row_rule(capi, cmm, dev, oracle_mod):
    """AUTO resolves BERT-base's pruned probs·V (384 items of 512×512 · 512×64 in batched CSR form, BASELINE.json
    configs[4]) to MI_SPMM_LDS_B; a shape the plan does not fit (K·N·4 > 128 KB) stays on the row-split kernels.
    With one item (a tall matrix on a small B) the long-row rule still holds under this plan: a row beyond 8192
    non-zeros (duplicate columns) is skipped, listed and summed by the follow-up kernel in the split order, with the
    fused bias — the oracle's statement of mi_spmm_csr_ws_f32 (reference entry src/custom_mm.cpp:166-179)."""
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    capi.mi_spmm_csr_batched_f32_plan.argtypes = [i64, i32, i32, i32, i32, vp, i64, i64, vp, i64, i64]
    capi.mi_spmm_csr_f32_plan.argtypes = [i64, i32, i32, i32, vp, i64, vp, i64]
    assert capi.mi_spmm_csr_batched_f32_plan(10_000_000, 384, 512, 512, 64, None, 64, 512 * 64, None, 64, 512 * 64) == 18
    # 1024 tokens: B (256 KB) goes in as two column tiles of 32, 2048 tokens as four of 16 (the quad form); 4096: row-split
    assert capi.mi_spmm_csr_batched_f32_plan(10_000_000, 96, 1024, 1024, 64, None, 64, 1024 * 64, None, 64, 1024 * 64) == 18
    assert capi.mi_spmm_csr_batched_f32_plan(10_000_000, 48, 2048, 2048, 64, None, 64, 2048 * 64, None, 64, 2048 * 64) == 18
    assert capi.mi_spmm_csr_batched_f32_plan(10_000_000, 24, 4096, 4096, 64, None, 64, 4096 * 64, None, 64, 4096 * 64) != 18
    # from 4 non-zeros per row whatever the number of column tiles (tools/bench_plans.py: 65536 × 128 × 256 — four tiles of
    # 64 — at 25 / 10 / 5 % kept 0.039 / 0.025 / 0.021 ms here against 0.058 / 0.040 / 0.025 for the plans AUTO took before)
    assert capi.mi_spmm_csr_f32_plan(2_097_040, 65536, 128, 256, None, 256, None, 256) == 18
    assert capi.mi_spmm_csr_f32_plan(418_690, 65536, 128, 256, None, 256, None, 256) == 18
    assert capi.mi_spmm_csr_f32_plan(3 * 65536, 65536, 128, 256, None, 256, None, 256) != 18
    g = np.random.Generator(np.random.PCG64(18))
    M, K, N = 20000, 256, 64
    lens = g.integers(2, 12, size=M)
    lens[77], lens[19999] = 9000, 70000
    col = np.concatenate([g.integers(0, K, size=int(n)).astype(np.int32) for n in lens])
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    val = g.random(len(col), dtype=np.float32) - 0.5
    B, bias = g.random((K, N), dtype=np.float32) - 0.5, g.random(N, dtype=np.float32)
    assert capi.mi_spmm_csr_f32_plan(len(col), M, K, N, None, N, None, N) == 18
    want = oracle_mod.spmm_csr_long(rowptr, col, val, M, K, B)
    assert np.array_equal(run_spmm(cmm, dev, rowptr, col, val, M, K, B, "naive_spmm"), want)
    C = torch.full((M, N), float("nan"), device=dev)
    cmm.naive_spmm_bias(t(val, dev), t(col, dev), t(rowptr, dev), len(col), M, K, t(B, dev), t(bias, dev), C)
    assert np.array_equal(C.cpu().numpy(), want + bias[None, :])


def test_lds_resident_b_product_is_graph_capturable(capi, cmm, dev, oracle_mod):
    """`custom_mm.naive_spmm_batched` on a batch AUTO resolves to MI_SPMM_LDS_B (function attribute for > 64 KB of LDS,
    device query for the grid) captured into a hipGraph and replayed on new values and a new B."""
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    capi.mi_spmm_csr_batched_f32_plan.argtypes = [i64, i32, i32, i32, i32, vp, i64, i64, vp, i64, i64]
    batch, M, K, N = 80, 256, 300, 64
    g = np.random.Generator(np.random.PCG64(31))
    lens = g.integers(4, 30, size=batch * M)
    col = np.concatenate([np.sort(g.choice(K, size=int(n), replace=False)).astype(np.int32) for n in lens])
    off = np.zeros((batch, M + 1), np.int64)
    off[:, 1:] = np.cumsum(lens).reshape(batch, M)
    off[1:, 0] = off[:-1, M]
    off = off.astype(np.int32)
    assert capi.mi_spmm_csr_batched_f32_plan(len(col), batch, M, K, N, None, N, K * N, None, N, M * N) == 18
    d_off, d_col = t(off, dev), t(col, dev)
    d_val = torch.zeros(len(col), device=dev)
    B = torch.zeros(batch, K, N, device=dev)
    C = torch.empty(batch, M, N, device=dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        cmm.naive_spmm_batched(d_val, d_col, d_off, len(col), batch, M, K, B, C)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        cmm.naive_spmm_batched(d_val, d_col, d_off, len(col), batch, M, K, B, C)
    for _ in range(2):
        val = g.random(len(col), dtype=np.float32) - 0.5
        Bh = g.random((batch, K, N), dtype=np.float32) - 0.5
        d_val.copy_(torch.from_numpy(val))
        B.copy_(torch.from_numpy(Bh))
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(C.cpu().numpy(), oracle_mod.spmm_csr_batched(off, col, val, batch, M, K, Bh))


@pytest.mark.parametrize("N", [64, 256])
def test_spmm_count_as_bound_or_estimate_same_bits(capi, cmm, dev, oracle_mod, N):
    """Round 5 (advisor, high / medium).  include/mi_spmm.h: rowptr's last entry ≤ nnz ≤ array length; without a long-row
    workspace a count BELOW the true one only steers the plan.  The quad form of MI_SPMM_LDS_B clamps its 16-byte col / val
    loads: the clamp is max(count, rowptr's last entry) read on the device, so the same bits come out for (a) the exact
    count, (b) a capacity (arrays padded with garbage behind the data), (c) an underestimate — both forms, one matrix
    (mi_spmm_csr_f32_variant) and a batch; and custom_mm.naive_spmm_ex (rule 0, what fc_layers calls) with an estimate."""
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    capi.mi_spmm_csr_batched_variant_f32.argtypes = [ctypes.c_int, vp, vp, vp, i64, i32, i32, i32, i32, vp, i64, i64, vp, i64, i64, vp]
    capi.mi_spmm_ldsb_set_form.argtypes = [ctypes.c_int]
    stream = torch.cuda.current_stream().cuda_stream
    g = np.random.Generator(np.random.PCG64(77 + N))
    try:
        for batch, M, K in ((1, 20000, 128), (6, 700, 256 if N == 64 else 100)):
            lens = g.integers(0, 30, size=batch * M)
            lens[-1] = 3  # the arrays end inside a 16-byte load
            col = np.concatenate([np.sort(g.integers(0, K, size=int(n))).astype(np.int32) for n in lens])
            val = g.random(len(col), dtype=np.float32) - 0.5
            off = np.zeros((batch, M + 1), np.int64)
            off[:, 1:] = np.cumsum(lens).reshape(batch, M)
            off[1:, 0] = off[:-1, M]
            off = off.astype(np.int32)
            B = g.random((batch, K, N), dtype=np.float32) - 0.5
            want = oracle_mod.spmm_csr_batched(off, col, val, batch, M, K, B)
            true = len(col)
            pad = 1000
            # garbage behind the data: a kernel that read it as an entry would show it (huge values, out-of-range columns)
            col_p = np.concatenate([col, np.full(pad, 2**30, np.int32)])
            val_p = np.concatenate([val, np.full(pad, 1e30, np.float32)])
            d_off, d_col, d_val, d_B = t(off, dev), t(col_p, dev), t(val_p, dev), t(B, dev)
            for form in (1, 0):
                assert capi.mi_spmm_ldsb_set_form(form) == 0
                for count in (true, true + pad, true + 5, max(4, true // 2), max(4, true - 2), 4):
                    C = torch.full((batch, M, N), float("nan"), device=dev)
                    st = capi.mi_spmm_csr_batched_variant_f32(18, d_off.data_ptr(), d_col.data_ptr(), d_val.data_ptr(), count, batch,
                                                              M, K, N, d_B.data_ptr(), N, K * N, C.data_ptr(), N, M * N, stream)
                    assert st == 0
                    assert np.array_equal(C.cpu().numpy().view(np.int32), want.view(np.int32)), (form, batch, count, true)
            if batch == 1:
                capi.mi_spmm_ldsb_set_form(-1)
                assert cmm.spmm_plan(true // 2, M, K, d_B[0], torch.empty(M, N, device=dev))[0] == 18
                C = torch.full((M, N), float("nan"), device=dev)
                cmm.naive_spmm_ex(d_val, d_col, d_off.view(-1), true // 2, M, K, d_B[0], C, 0)
                assert np.array_equal(C.cpu().numpy().view(np.int32), want[0].view(np.int32))
    finally:
        capi.mi_spmm_ldsb_set_form(-1)


@pytest.mark.parametrize("batch,M,K,N", [(5, 333, 300, 64), (3, 700, 1100, 48), (2, 9, 17, 7), (40, 512, 512, 64), (1, 1, 1, 1),
                                         (7, 130, 33, 20), (2, 1200, 40, 64), (300, 64, 64, 32)])
def test_transpose_free_at_product_bit_exact(cmm, dev, oracle_mod, batch, M, K, N):
    """Round 5: Y[i] = A[i]ᵀ·X[i] for batched CSR A WITHOUT a transpose (csrc/spmm_at.hip: output tile in registers, every
    wave walks the item's rows in ascending order and picks the entries of its own columns) — the gradient of V in pruned
    attention with a new pattern per step.  Bit-exact against oracle.csr_transpose + oracle.spmm_csr per item: rows longer
    than a 64-entry chunk, duplicate and unsorted columns, empty rows and columns, items taller than one 512-row tile of
    the staged operand, more columns than one workgroup owns, N not a multiple of 4 (scalar staging).  The index narrowing
    of torch's int64 batched CSR (one launch) is checked on the way.  Reference: matmuls.py:245-256 (no backward for a
    batched CSR operand), :289-297 (per-call conversion)."""
    g = np.random.Generator(np.random.PCG64(batch * 1000 + M + K + N))
    lens = g.integers(0, min(2 * K + 3, 150), size=batch * M)
    lens[g.integers(0, batch * M, size=max(1, batch * M // 20))] = 0
    if M > 3:
        lens[2] = min(2 * K + 5, 300)          # longer than K: duplicate columns; longer than a chunk
    cols = [g.integers(0, K, size=int(n)).astype(np.int32) for n in lens]
    cols = [c if i % 3 == 0 else np.sort(c) for i, c in enumerate(cols)]
    col = np.concatenate(cols) if len(cols) else np.zeros(0, np.int32)
    val = (g.random(len(col), dtype=np.float32) - 0.5).astype(np.float32)
    off = np.zeros((batch, M + 1), np.int64)
    off[:, 1:] = np.cumsum(lens).reshape(batch, M)
    off[1:, 0] = off[:-1, M]
    off = off.astype(np.int32)
    X = g.random((batch, M, N), dtype=np.float32) - 0.5
    want = np.zeros((batch, K, N), np.float32)
    for i in range(batch):
        p0, p1 = off[i, 0], off[i, M]
        t_rp, t_col, t_val = oracle_mod.csr_transpose(off[i] - p0, col[p0:p1], val[p0:p1], M, K)
        want[i] = oracle_mod.spmm_csr(t_rp, t_col, t_val, K, M, X[i])
    Y = torch.full((batch, K, N), float("nan"), device=dev)
    took = cmm.naive_spmm_batched_at(t(val, dev), t(col, dev), t(off, dev), len(col), batch, M, K, t(X, dev), Y)
    assert took
    assert np.array_equal(Y.cpu().numpy().view(np.int32), want.view(np.int32))
    # wider than 64 columns: declined, nothing written
    Yw = torch.full((batch, K, 68), float("nan"), device=dev)
    assert not cmm.naive_spmm_batched_at(t(val, dev), t(col, dev), t(off, dev), len(col), batch, M, K,
                                         torch.zeros(batch, M, 68, device=dev), Yw)
    assert bool(torch.isnan(Yw).all())
    # the one-launch narrowing of torch's batched CSR indices (equal counts per item)
    per_item = 11
    crow = torch.arange(0, M + 1, dtype=torch.int64).clamp(max=per_item).repeat(batch, 1).to(dev)
    ccol = torch.from_numpy(g.integers(0, K, size=(batch, per_item))).to(dev)
    off32, col32 = cmm.batched_csr_narrow(crow, ccol)
    base = torch.arange(batch, device=dev, dtype=torch.int64).unsqueeze(1) * per_item
    assert off32.dtype == torch.int32 and torch.equal(off32.long(), crow + base)
    assert col32.dtype == torch.int32 and torch.equal(col32.long(), ccol.reshape(-1))
