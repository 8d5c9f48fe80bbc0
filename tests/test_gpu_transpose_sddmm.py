"""Device conversions behind the sparse backward: dense→CSR, CSR transpose, SDDMM.

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


@pytest.mark.parametrize("shape,density", [((1, 1), 1.0), ((7, 5), 0.5), ((3, 64, 64), 0.3), ((2, 3, 100, 257), 0.05),
                                           ((513, 1000), 0.01), ((4, 0, 8), 0.5), ((1, 5, 0), 0.5)])
def test_dense_to_csr_bit_exact(cmm, dev, oracle_mod, shape, density):
    g = np.random.Generator(np.random.PCG64(sum(shape)))
    a = (g.random(shape, dtype=np.float32) * (g.random(shape) < density)).astype(np.float32)
    if a.size:
        a.flat[0] = -0.0  # negative zero is a zero (x != 0 test, like torch.to_sparse_csr)
    vals, cols, offs = cmm.dense_to_csr(t(a, dev))
    rp, col, val = oracle_mod.dense_to_csr(a)
    assert offs.dtype == torch.int32 and cols.dtype == torch.int32
    assert np.array_equal(offs.cpu().numpy(), rp)
    assert np.array_equal(cols.cpu().numpy(), col) and np.array_equal(vals.cpu().numpy(), val)


@pytest.mark.parametrize("M,K,density", [(300, 170, 0.05), (50, 4000, 0.01), (4000, 50, 0.3), (1, 1, 1.0), (64, 64, 0.0),
                                          (7, 100000, 0.00002), (3000, 3000, 0.02)])
def test_csr_transpose_bit_exact(cmm, dev, oracle_mod, M, K, density):
    rowptr, col, val = oracle_mod.make_csr(M, K, density, seed=3) if density > 0 else \
        (np.zeros(M + 1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32))
    t_val, t_col, t_off = cmm.csr_transpose(t(val, dev), t(col, dev), t(rowptr, dev), len(val), M, K)
    e_rp, e_col, e_val = oracle_mod.csr_transpose(rowptr, col, val, M, K)
    assert np.array_equal(t_off.cpu().numpy(), e_rp) and np.array_equal(t_col.cpu().numpy(), e_col)
    assert np.array_equal(t_val.cpu().numpy(), e_val)


@pytest.mark.parametrize("M,K,nnz_per_row,what", [
    (500, 1000, 40, "one pass (K <= 1024), several tiles"),
    (20000, 70000, 30, "two passes, ragged digit split (17 bits), many tiles"),
    (3000, 1 << 20, 300, "two passes, 10 + 10 bits"),
    (2000, 3_000_000, 25, "three passes, full keys + boundary pass"),
    (500, 3_000_000, 60, "three passes with 8-byte entries: staged passes, then the register-scatter last pass"),
    (5_000_000, 600_000, 0, "two passes, rows too wide for the 8-byte entry (register scatter path)"),
    (40, 1_200_000_000, 50, "31-bit keys: three passes with an 11-bit digit"),
])
def test_csr_transpose_pass_structures(cmm, dev, oracle_mod, M, K, nnz_per_row, what):
    """Every plan of csr_transpose.hip — 1 / 2 / 3 counting passes, LDS-staged and register-scatter
    entries — against the oracle, with a fifth of the rows shuffled out of column order and duplicate
    columns allowed (the transpose must be stable: equal columns keep their row / position order)."""
    g = np.random.Generator(np.random.PCG64(M + K))
    if nnz_per_row:
        lens = g.integers(0, 2 * nnz_per_row, size=M)
    else:  # a few thousand short rows scattered over five million
        lens = np.zeros(M, np.int64)
        lens[g.choice(M, size=4000, replace=False)] = g.integers(1, 30, size=4000)
    rowptr, col, val = _random_rows_csr(M, K, lens, seed=K % 1000, shuffle=0.2, duplicates=True)
    t_val, t_col, t_off = cmm.csr_transpose(t(val, dev), t(col, dev), t(rowptr, dev), len(val), M, K)
    if K > 10_000_000:
        # the oracle's dense row-offset array would be 8 GB: check through the sorted entries instead
        order = np.argsort(col, kind="stable")
        rows = np.repeat(np.arange(M, dtype=np.int32), np.diff(rowptr))
        assert np.array_equal(t_col.cpu().numpy(), rows[order]) and np.array_equal(t_val.cpu().numpy(), val[order])
        off = t_off.cpu().numpy()
        assert off[0] == 0 and off[-1] == len(val) and np.all(np.diff(off[::4096]) >= 0)
        used = np.unique(col)
        assert np.array_equal(off[used], np.searchsorted(col[order], used, side="left"))
        assert np.array_equal(off[used + 1], np.searchsorted(col[order], used, side="right"))
        return
    e_rp, e_col, e_val = oracle_mod.csr_transpose(rowptr, col, val, M, K)
    assert np.array_equal(t_off.cpu().numpy(), e_rp), what
    assert np.array_equal(t_col.cpu().numpy(), e_col) and np.array_equal(t_val.cpu().numpy(), e_val), what


@pytest.mark.parametrize("batch,M,K,density", [(6, 50, 70, 0.3), (384, 64, 512, 0.1), (3, 700, 5000, 0.01), (5, 1, 9, 0.5),
                                               (4, 300, 300000, 0.001)])
def test_csr_transpose_batched_bit_exact(cmm, dev, oracle_mod, batch, M, K, density):
    """The batched CSR ("rowptr of rowptrs") of dense_to_csr transposed in one set of launches equals
    the per-item oracle transposes laid out the same way."""
    g = np.random.Generator(np.random.PCG64(batch * M + K))
    dense = (g.random((batch, M, K), dtype=np.float32) - 0.5) * (g.random((batch, M, K), dtype=np.float32) < density)
    dense[batch // 2] = 0  # an empty item
    values, columns, offsets = cmm.dense_to_csr(t(dense, dev))
    nnz = values.numel()
    t_val, t_col, t_off = cmm.csr_transpose_batched(values, columns, offsets, nnz, batch, M, K)
    assert t_off.shape == (batch, K + 1)
    off = offsets.cpu().numpy()
    col, val = columns.cpu().numpy(), values.cpu().numpy()
    want_off, want_col, want_val = [], [], []
    for b in range(batch):
        s0, s1 = off[b, 0], off[b, M]
        rp, c, v = oracle_mod.csr_transpose((off[b] - s0).astype(np.int32), col[s0:s1], val[s0:s1], M, K)
        want_off.append(rp.astype(np.int64) + s0)
        want_col.append(c)
        want_val.append(v)
    assert np.array_equal(t_off.cpu().numpy(), np.stack(want_off).astype(np.int32))
    assert np.array_equal(t_col.cpu().numpy(), np.concatenate(want_col))
    assert np.array_equal(t_val.cpu().numpy(), np.concatenate(want_val))
    # and it feeds the batched kernel: C[b] = A[b]ᵀ · G[b]
    G = g.random((batch, M, 8), dtype=np.float32)
    C = torch.empty(batch, K, 8, device=dev)
    cmm.naive_spmm_batched(t_val, t_col, t_off, nnz, batch, K, M, t(G, dev), C)
    assert np.allclose(C.cpu().numpy(), np.einsum("bmk,bmn->bkn", dense.astype(np.float64), G.astype(np.float64)),
                       rtol=1e-4, atol=1e-5)


def test_csr_transpose_shape_fuzz(cmm, dev, oracle_mod):
    """Seeded random shapes around the plan boundaries of csr_transpose.hip (K·batch at 2^10 and 2^20 ± 1,
    tiles of exactly / just over 8192 entries, rows spanning many tiles, runs of empty rows, duplicates,
    unsorted rows), single and batched: always the oracle's stable transpose."""
    rng = np.random.Generator(np.random.PCG64(20260))
    shapes = [(1, 1, 1), (1, 3, 1024), (1, 3, 1025), (1, 700, 1023), (1, 9000, 1024), (1, 300, (1 << 20) - 1),
              (1, 300, 1 << 20), (1, 300, (1 << 20) + 1), (1, 1, 50000), (1, 50000, 1), (3, 64, 341), (3, 64, 342),
              (7, 100, 149797), (7, 100, 149798), (2, 5000, 600), (1, 8192, 2), (1, 8193, 2)]
    for case, (batch, M, K) in enumerate(shapes):
        for density in (0.0, 0.002, 0.05, 0.6):
            target = int(min(batch * M * K * density, 300_000))
            if density > 0 and target == 0:
                target = min(batch * M * K, 5)
            lens = rng.multinomial(target, rng.dirichlet(np.full(batch * M, 0.3))) if target else np.zeros(batch * M, np.int64)
            lens = np.minimum(lens, 4 * K)                      # rows longer than K: duplicate columns
            cols = []
            for n in lens:
                c = np.sort(rng.integers(0, K, size=int(n)))
                if n > 1 and rng.random() < 0.3:
                    c = rng.permutation(c)
                cols.append(c.astype(np.int32))
            col = np.concatenate(cols) if len(cols) else np.zeros(0, np.int32)
            val = rng.random(len(col), dtype=np.float32) - 0.5
            off = np.zeros((batch, M + 1), np.int64)
            off[:, 1:] = np.cumsum(lens).reshape(batch, M)
            off[1:, 0] = off[:-1, M]
            off = off.astype(np.int32)
            tag = (case, batch, M, K, density, len(col))
            if batch == 1:
                t_val, t_col, t_off = cmm.csr_transpose(t(val, dev), t(col, dev), t(off[0], dev), len(col), M, K)
                t_off = t_off.view(1, -1)
            else:
                t_val, t_col, t_off = cmm.csr_transpose_batched(t(val, dev), t(col, dev), t(off, dev), len(col), batch, M, K)
            got_off, got_col, got_val = t_off.cpu().numpy(), t_col.cpu().numpy(), t_val.cpu().numpy()
            for b in range(batch):
                s0, s1 = off[b, 0], off[b, M]
                rp, c, v = oracle_mod.csr_transpose((off[b] - s0).astype(np.int32), col[s0:s1], val[s0:s1], M, K)
                assert np.array_equal(got_off[b], rp.astype(np.int64) + s0), tag
                assert np.array_equal(got_col[s0:s1], c) and np.array_equal(got_val[s0:s1], v), tag


def test_sddmm_config_c3_two_panel_launches_bit_exact(cmm, dev, oracle_mod):
    """SDDMM at BASELINE config C3's shape (B = 1 GiB, beyond the Infinity Cache) runs as two column-panel
    launches; every value still comes from the same per-value chain + tree: bit-exact against the oracle
    on sampled rows (first, last, a few thousand in between), and every non-zero is written exactly once."""
    import synthetic
    M = K = 1 << 20
    N = 256
    rowptr, col, val = synthetic.make_csr(M, K, 1e-4, seed=0)
    B = synthetic.make_dense(K, N, seed=1)
    g = torch.Generator(device=dev).manual_seed(3)
    dC = torch.rand(M, N, device=dev, generator=g)
    out = cmm.sddmm(t(col, dev), t(rowptr, dev), len(val), M, K, dC, t(B, dev))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert got.shape == (len(val),) and np.isfinite(got).all() and (got > 0).all()   # positive operands: no slot left unwritten
    rows = np.unique(np.concatenate([[0, 1, M - 1], np.random.Generator(np.random.PCG64(5)).integers(0, M, 3000)]))
    sub_rp, sub_col, _ = _sub_csr(rowptr, col, val, list(rows))
    want = oracle_mod.sddmm(sub_rp, sub_col, len(rows), dC[torch.from_numpy(rows).to(dev)].cpu().numpy(), B)
    idx = np.concatenate([np.arange(rowptr[r], rowptr[r + 1]) for r in rows])
    assert np.array_equal(got[idx], want)


@pytest.mark.parametrize("batch,M,K,nnz,skew", [(1, 200000, 300000, 6_000_000, 0.3), (1, 50000, 900, 5_000_000, 1.0),
                                                 (1, 3000, 3_000_000, 4_000_000, 0.5), (6, 20000, 70000, 5_000_000, 0.2)])
def test_csr_transpose_many_tiles_per_workgroup_skewed(cmm, dev, batch, M, K, nnz, skew):
    """Several tiles per persistent workgroup (the software-pipelined scatter: previous tile streaming out, next
    tile loading while one is ranked) on SKEWED data — Dirichlet row lengths with runs of empty rows, a hub
    column holding 5 % of the entries (every lane of a ranking step with the same digit), duplicates, some rows
    out of column order — in the one-, two- and three-pass plans and the batched form: offsets, row indices and
    values equal numpy's stable sort of the keys."""
    rng = np.random.Generator(np.random.PCG64(M + K))
    lens = rng.multinomial(nnz, rng.dirichlet(np.full(batch * M, skew)))
    col = rng.integers(0, K, size=nnz).astype(np.int32)
    col[rng.random(nnz) < 0.05] = K // 3                       # the hub column
    off = np.zeros(batch * M + 1, np.int64)
    off[1:] = np.cumsum(lens)
    rows_flat = np.repeat(np.arange(batch * M, dtype=np.int64), lens)
    # ascending columns inside every row … except in every 7th row
    order = np.lexsort((col, rows_flat))
    unsorted = (rows_flat % 7) == 3
    col = np.where(unsorted, col, col[order]).astype(np.int32)
    val = rng.random(nnz, dtype=np.float32) - 0.5
    offs = np.zeros((batch, M + 1), np.int64)
    offs[:, 1:] = off[1:].reshape(batch, M)
    offs[1:, 0] = offs[:-1, M]
    offs = offs.astype(np.int32)
    if batch == 1:
        t_val, t_col, t_off = cmm.csr_transpose(t(val, dev), t(col, dev), t(offs[0], dev), nnz, M, K)
        t_off = t_off.view(1, -1)
    else:
        t_val, t_col, t_off = cmm.csr_transpose_batched(t(val, dev), t(col, dev), t(offs, dev), nnz, batch, M, K)
    torch.cuda.synchronize()
    item = rows_flat // M
    key = item * K + col
    order = np.argsort(key, kind="stable")
    assert np.array_equal(t_col.cpu().numpy(), (rows_flat - item * M).astype(np.int32)[order])
    assert np.array_equal(t_val.cpu().numpy(), val[order])
    counts = np.bincount(key, minlength=batch * K).reshape(batch, K)
    want_off = np.zeros((batch, K + 1), np.int64)
    want_off[:, 1:] = np.cumsum(counts, axis=1)
    want_off += np.concatenate([[0], np.cumsum(counts.sum(1))[:-1]])[:, None]
    assert np.array_equal(t_off.cpu().numpy().astype(np.int64), want_off)


@pytest.mark.parametrize("M,K,nnz,skew,holes", [(200_000, 300_000, 6_000_000, 0.3, False),   # digits of 10 + 9 bits
                                                (150_000, 1 << 20, 7_000_000, 1.0, False),   # 10 + 10 bits
                                                (90_000, 1 << 20, 5_000_000, 0.05, True)])   # most low-digit bins empty
def test_csr_transpose_one_sweep_plan_bit_exact(capi, dev, M, K, nnz, skew, holes):
    """Round 4: the one-sweep plan (the first pass's count launch also counts the last pass's digits per group of bins;
    the last scatter launch finds its tiles' offsets by decoupled look-back inside those groups, tiles handed out by
    tickets) against numpy's stable sort of the columns AND bit for bit against the table plan — on skewed data:
    Dirichlet row lengths with runs of empty rows, a hub column (every lane of a ranking step with the same digit: one
    tile's whole count in one status word), duplicates, rows out of column order, bins without a single entry (their row
    offsets come from the next bin that has one).  The plan's give-up flag (a poll that ran into its limit) must stay 0."""
    rng = np.random.Generator(np.random.PCG64(M + K))
    lens = rng.multinomial(nnz, rng.dirichlet(np.full(M, skew)))
    col = rng.integers(0, K, size=nnz).astype(np.int32)
    if holes:
        col &= ~np.int32(0x3F8)                                # low digits 0..7 only (+ the hub's)
    col[rng.random(nnz) < 0.05] = K // 3                       # the hub column
    rowptr = np.zeros(M + 1, np.int64)
    rowptr[1:] = np.cumsum(lens)
    rows = np.repeat(np.arange(M, dtype=np.int64), lens)
    order = np.lexsort((col, rows))
    unsorted = (rows % 7) == 3                                 # ascending columns … except in every 7th row
    col = np.where(unsorted, col, col[order]).astype(np.int32)
    val = rng.random(nnz, dtype=np.float32) - 0.5
    rowptr = rowptr.astype(np.int32)
    assert capi.mi_csr_transpose_one_sweep_applies(1, M, K, nnz) == 1
    got = _transpose_through_the_c_abi(capi, dev, rowptr, col, val, M, K, plan=2)
    ref = _transpose_through_the_c_abi(capi, dev, rowptr, col, val, M, K, plan=1)
    for x, y, what in zip(got, ref, ("t_rowptr", "t_col", "t_val")):
        assert np.array_equal(x, y), what
    order = np.argsort(col, kind="stable")
    assert np.array_equal(got[1], rows.astype(np.int32)[order])
    assert np.array_equal(got[2], val[order])
    want_off = np.concatenate([[0], np.cumsum(np.bincount(col, minlength=K))]).astype(np.int32)
    assert np.array_equal(got[0], want_off)
    # a problem the plan does not cover cannot be pinned onto it
    assert capi.mi_csr_transpose_one_sweep_applies(1, 1000, 1000, 50_000) == 0
    assert capi.mi_csr_transpose_one_sweep_applies(4, M, K // 4, nnz) == 0


def test_csr_transpose_config_c3_shape(cmm, dev, oracle_mod):
    """The transpose at BASELINE config C3's matrix (1M x 1M, 110 M non-zeros): integer artefacts
    (offsets, row indices) and values bit-exact against numpy's stable argsort of the columns.  (At this size AUTO
    takes the one-sweep plan, round 4.)"""
    import synthetic
    M = K = 1 << 20
    rowptr, col, val = synthetic.make_csr(M, K, 1e-4, seed=0)
    t_val, t_col, t_off = cmm.csr_transpose(t(val, dev), t(col, dev), t(rowptr, dev), len(val), M, K)
    torch.cuda.synchronize()
    order = np.argsort(col, kind="stable")
    rows = np.repeat(np.arange(M, dtype=np.int32), np.diff(rowptr))
    assert np.array_equal(t_col.cpu().numpy(), rows[order])
    assert np.array_equal(t_val.cpu().numpy(), val[order])
    want_off = np.concatenate([[0], np.cumsum(np.bincount(col, minlength=K))]).astype(np.int32)
    assert np.array_equal(t_off.cpu().numpy(), want_off)


@pytest.mark.parametrize("N", [1, 4, 8, 12, 16, 32, 48, 64, 68, 100, 128, 256, 300, 512, 777, 1024, 1500, 4100])
def test_sddmm_bit_exact(cmm, dev, oracle_mod, N):
    """Rows of 0 … 400 pattern entries (batches of 64 with a ragged tail), every register-pass count
    of the dC row, odd widths (scalar loads) and rows wider than one register chunk; signed data."""
    M, K = 120, 400
    rowptr, col, val = oracle_mod.make_csr(M, K, 0.3, seed=N)
    g = np.random.Generator(np.random.PCG64(N))
    dC, B = g.random((M, N), dtype=np.float32) - 0.5, g.random((K, N), dtype=np.float32) - 0.5
    got = cmm.sddmm(t(col, dev), t(rowptr, dev), len(val), M, K, t(dC, dev), t(B, dev))
    assert np.array_equal(got.cpu().numpy(), oracle_mod.sddmm(rowptr, col, M, dC, B))
    # strided operands (column slices of wider tensors) go through lddc / ldb
    if N % 4 == 0 and N <= 512:
        dCw, Bw = torch.rand(M, N + 8, device=dev) - 0.5, torch.rand(K, 2 * N, device=dev) - 0.5
        got = cmm.sddmm(t(col, dev), t(rowptr, dev), len(val), M, K, dCw[:, 4:4 + N], Bw[:, N:])
        assert np.array_equal(got.cpu().numpy(), oracle_mod.sddmm(rowptr, col, M, dCw[:, 4:4 + N].cpu().numpy(),
                                                                   Bw[:, N:].cpu().numpy()))


@pytest.mark.parametrize("M,K,N,density", [(1500, 16384, 256, 0.02), (900, 40000, 256, 0.01), (700, 9000, 256, 0.05)])
def test_sddmm_mid_size_b_runs_in_l2_panels_bit_exact(cmm, dev, oracle_mod, M, K, N, density):
    """B beyond the L2s (6 MiB < |B| ≤ 128 MiB): SDDMM runs as up to 8 column-panel launches, every pattern
    entry computed in exactly one of them by the same dot + tree — bit-exact against the oracle, rows out of
    column order and empty rows included, every entry written."""
    g = np.random.Generator(np.random.PCG64(M + N))
    rowptr, col, val = oracle_mod.make_csr(M, K, density, seed=K % 97)
    col = col.copy()
    for r in (0, 5, M - 1):
        s0, e0 = rowptr[r], rowptr[r + 1]
        col[s0:e0] = g.permutation(col[s0:e0])
    dC, B = g.random((M, N), dtype=np.float32) - 0.5, g.random((K, N), dtype=np.float32) - 0.5
    got = cmm.sddmm(t(col, dev), t(rowptr, dev), len(col), M, K, t(dC, dev), t(B, dev))
    assert np.array_equal(got.cpu().numpy(), oracle_mod.sddmm(rowptr, col, M, dC, B))


def test_csr_transpose_small_items_lds_plan_skewed_batch_bit_exact(cmm, dev, oracle_mod):
    """Round 5: the one-workgroup-per-item LDS transpose (tr_item_lds_kernel) on a batch it was NOT sized for: one item with
    25 × the average number of entries (it needs many more column passes than the launch's workgroups per item: the last
    workgroup of an item takes every pass behind its own), a column longer than the staging area (its entries go out
    directly), rows longer than a 64-entry chunk, unsorted rows and duplicate columns (positions taken lane by lane),
    empty items and empty rows.  Bit-exact against the oracle's stable transpose, item by item."""
    g = np.random.Generator(np.random.PCG64(123))
    batch, M, K = 96, 3000, 700
    assert cmm.csr_transpose_in_lds(batch * 2000, batch, M, K)
    lens = g.integers(0, 2, size=(batch, M))                      # ≈ 1.5 K entries per item on average
    lens[0] = g.integers(8, 18, size=M)                            # item 0: ≈ 37 K entries
    lens[0, 5] = 200                                               # a row of several chunks
    lens[7] = 0                                                    # an empty item
    lens[batch - 2:] = 0                                           # … and the LAST items empty: their base offset is nnz itself, one
                                                                   # past the arrays (round-5 advisor: the clamped loads must stay inside)
    cols = []
    for i in range(batch):
        for r in range(M):
            n = int(lens[i, r])
            c = g.integers(0, K, size=n)
            if i == 0:
                c[:1] = 5 if n else c[:1]                          # column 5 in every row of item 0: 3000 entries > the staging area
            cols.append(c if (i + r) % 4 == 0 else np.sort(c))     # a quarter of the rows unsorted (duplicates possible anywhere)
    col = np.concatenate(cols).astype(np.int32)
    val = g.random(len(col), dtype=np.float32) - 0.5
    off = np.zeros((batch, M + 1), np.int64)
    off[:, 1:] = np.cumsum(lens.reshape(-1)).reshape(batch, M)
    off[1:, 0] = off[:-1, M]
    off = off.astype(np.int32)
    assert cmm.csr_transpose_in_lds(len(col), batch, M, K)
    t_val, t_col, t_off = cmm.csr_transpose_batched(t(val, dev), t(col, dev), t(off, dev), len(col), batch, M, K)
    t_val, t_col, t_off = t_val.cpu().numpy(), t_col.cpu().numpy(), t_off.cpu().numpy()
    for i in range(batch):
        p0, p1 = off[i, 0], off[i, M]
        w_rp, w_col, w_val = oracle_mod.csr_transpose(off[i] - p0, col[p0:p1], val[p0:p1], M, K)
        assert np.array_equal(t_off[i] - p0, w_rp), i
        assert np.array_equal(t_col[p0:p1], w_col), i
        assert np.array_equal(t_val[p0:p1].view(np.int32), w_val.view(np.int32)), i
