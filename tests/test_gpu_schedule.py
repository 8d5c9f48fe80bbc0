"""Row schedules (include/mi_spmm.h "Row schedules"; csrc/spmm_sched.hip): the inspector's format for degree-skewed matrices
— rows handed to waves longest first, like lengths together, the heavy rows in a launch of their own (SURVEY.md §8f-3;
reference: the inspector that builds a format once, src/sparse_mm.cu:137-368, and the merge-spmm lineage of the kernel,
src/naive_sparse_mm.cu:20-21).  A schedule changes WHO computes a row, never how: every check here is bit for bit —
against the unscheduled product, and against the CPU oracle on the same seeded inputs.  Needs the MI355X (`-m gpu`).
"""
import numpy as np
import pytest
import torch

from gpu_helpers import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


def _length_class(n):
    n = np.asarray(n, dtype=np.int64)
    e = np.floor(np.log2(np.maximum(n, 1))).astype(np.int64)
    return np.where(n < 32, n, 32 + (e - 5) * 8 + ((n >> np.maximum(e - 3, 0)) & 7))


def _pareto_lens(M, mean, top, seed, empty=0.0):
    g = np.random.Generator(np.random.PCG64(seed))
    w = g.random(M) ** (-1.0 / 1.6)
    lens = np.clip(np.round(w * mean / w.mean() * 0.6), 1, top).astype(np.int64)
    if empty:
        lens[g.random(M) < empty] = 0
    return lens


def _product(cmm, dev, sched, rowptr, col, val, M, K, B, **kw):
    C = torch.full((M, B.shape[1]), float("nan"), device=dev)
    if sched is None:
        long_rows = kw.pop("long_rows", -1)
        bias = kw.pop("bias", None)
        assert not kw
        if bias is None:
            cmm.naive_spmm_ex(val, col, rowptr, col.numel(), M, K, B, C, long_rows)
        else:
            cmm.naive_spmm_bias_ex(val, col, rowptr, col.numel(), M, K, B, bias, C, long_rows)
    else:
        out = cmm.naive_spmm_scheduled(sched, val, col, rowptr, col.numel(), M, K, B, C, **kw)
        assert out.data_ptr() == C.data_ptr()
    return C


def test_schedule_is_a_permutation_by_descending_length_class(cmm, dev):
    """order[] holds every row once; length classes (exact below 32 entries, eight per octave above) never increase along
    it; the heavy slots are exactly the rows of the classes above the heavy length's class; info() agrees."""
    M = 50_000
    lens = _pareto_lens(M, 20, 20_000, seed=1, empty=0.05)
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    nnz = int(rowptr[-1])
    sched = cmm.spmm_schedule(t(rowptr, dev), nnz, M, 128)
    order = sched.order.cpu().numpy()
    assert order.dtype == np.int32 and np.array_equal(np.sort(order), np.arange(M))
    cls = _length_class(lens[order])
    assert np.all(np.diff(cls) <= 0)
    info = sched.info()
    assert info["rows"] == M and info["nnz"] == nnz and info["width"] == 128 and info["side_stream"] is True
    assert info["classes"] == len(np.unique(cls))
    # the rule: rows beyond nnz / 4096 entries and four times the mean (128 … the long-row threshold), and where more than 128 rows are, the longest classes only
    rule = min(max(nnz // 4096, 4 * nnz // M, 128), cmm.long_row_threshold())
    cls_all = _length_class(lens)
    c = int(_length_class(rule))
    while (cls_all > c).sum() > 128 and c < int(_length_class(cmm.long_row_threshold())):
        c += 1
    assert (info["heavy_length"] == rule) if c == int(_length_class(rule)) else (int(_length_class(info["heavy_length"])) == c and int(_length_class(info["heavy_length"] + 1)) == c + 1)
    assert info["heavy_rows"] <= 128 or c == int(_length_class(cmm.long_row_threshold()))
    assert info["heavy_rows"] == int((_length_class(lens) > _length_class(info["heavy_length"])).sum())
    assert np.all(lens[order[:info["heavy_rows"]]] > info["heavy_length"])
    assert info["longest_at_least"] <= lens.max() < info["longest_at_least"] * 1.13 + 1
    sched.set_heavy(1000, True)
    assert sched.info()["heavy_rows"] == int((_length_class(lens) > _length_class(1000)).sum())
    # a schedule of an empty matrix, and of one row
    for m in (0, 1):
        rp = np.zeros(m + 1, np.int32)
        s0 = cmm.spmm_schedule(t(rp, dev), 0, m, 64)
        assert s0.order.numel() == m and s0.info()["heavy_rows"] == 0
    with pytest.raises(RuntimeError):
        cmm.spmm_schedule(t(rowptr, dev)[:-1], nnz, M, 128)


@pytest.mark.parametrize("N", [256, 128, 100, 64, 602, 512, 40, 33, 2, 8, 16, 320])
def test_scheduled_product_equals_the_plain_product_and_the_oracle(cmm, dev, oracle_mod, N):
    """Pareto row lengths with empty rows and rows beyond the long-row threshold: the scheduled product has the bits of the
    unscheduled one under every long-row rule and with a bias, with the heavy launch on the side stream, in line, for every
    row (heavy length 0) and for none; and both equal the oracle (split order for the rows beyond the threshold)."""
    M, K = 6000, 40_000
    lens = _pareto_lens(M, 30, 12_000, seed=N, empty=0.03)
    lens[7], lens[4321] = 9000, 20_000      # beyond the threshold whatever the draw
    rowptr, col, val = _random_rows_csr(M, K, lens, seed=N + 1)
    g = np.random.Generator(np.random.PCG64(N + 2))
    B = g.random((K, N), dtype=np.float32) - 0.5
    bias = g.random(N, dtype=np.float32)
    d = [t(x, dev) for x in (rowptr, col, val, B, bias)]
    d_rp, d_col, d_val, d_B, d_bias = d
    sched = cmm.spmm_schedule(d_rp, len(val), M, N)
    assert sched.info()["heavy_rows"] > 0
    want_split = oracle_mod.spmm_csr_long(rowptr, col, val, M, K, B) if N >= 4 else oracle_mod.spmm_csr(rowptr, col, val, M, K, B)
    want_plain = oracle_mod.spmm_csr(rowptr, col, val, M, K, B)
    for long_rows, want in ((-1, want_split), (1, want_split), (0, want_plain)):
        plain = _product(cmm, dev, None, d_rp, d_col, d_val, M, K, d_B, long_rows=long_rows)
        assert np.array_equal(plain.cpu().numpy().view(np.uint32), want.view(np.uint32)), (N, long_rows)
        for heavy, side in ((None, True), (None, False), (0, True), (1 << 30, True), (64, False)):
            s2 = cmm.spmm_schedule(d_rp, len(val), M, N) if heavy is not None or not side else sched
            if heavy is not None or not side:
                s2.set_heavy(sched.info()["heavy_length"] if heavy is None else heavy, side)
            got = _product(cmm, dev, s2, d_rp, d_col, d_val, M, K, d_B, long_rows=long_rows)
            assert torch.equal(got.view(torch.int32), plain.view(torch.int32)), (N, long_rows, heavy, side)
    with_bias = _product(cmm, dev, None, d_rp, d_col, d_val, M, K, d_B, bias=d_bias)
    got = _product(cmm, dev, sched, d_rp, d_col, d_val, M, K, d_B, bias=d_bias)
    assert torch.equal(got.view(torch.int32), with_bias.view(torch.int32))
    # a schedule built for another matrix is refused; unknown long-row rules too
    other = cmm.spmm_schedule(d_rp[:101].contiguous(), int(rowptr[100]), 100, N)
    with pytest.raises(RuntimeError):
        _product(cmm, dev, other, d_rp, d_col, d_val, M, K, d_B)
    with pytest.raises(ValueError):
        _product(cmm, dev, sched, d_rp, d_col, d_val, M, K, d_B, long_rows=2)


@pytest.mark.parametrize("N,variants", [(256, (1, 2, 3, 4, 5, 7, 9, 13, 19, 21, 24, 6, 14, 17)), (128, (4, 5, 13, 19, 20, 24, 14, 18)),
                                        (512, (2, 4, 7, 8, 19, 14))])
def test_every_plan_gives_the_same_bits_on_a_schedule(cmm, dev, oracle_mod, N, variants):
    """The schedule under every pinned plan: the plans that walk rows wave by wave or group by group take the order (one-pass
    and column-panel kernels alike, rows out of column order included), the others (column tiles 14, slabs 17, LDS-resident
    18, the vector-load form 6) run unscheduled — all the same bits as the plain CSR-order oracle."""
    M, K = 5000, 9000
    lens = _pareto_lens(M, 40, 3000, seed=N + 5, empty=0.02)
    rowptr, col, val = _random_rows_csr(M, K, lens, seed=N + 6, shuffle=0.2)
    B = np.random.Generator(np.random.PCG64(N + 7)).random((K, N), dtype=np.float32) - 0.5
    d_rp, d_col, d_val, d_B = (t(x, dev) for x in (rowptr, col, val, B))
    want = oracle_mod.spmm_csr(rowptr, col, val, M, K, B)
    sched = cmm.spmm_schedule(d_rp, len(val), M, N)
    sched.set_heavy(200, True)
    assert 0 < sched.info()["heavy_rows"] < M
    ran = 0
    for v in variants:
        try:
            got = _product(cmm, dev, sched, d_rp, d_col, d_val, M, K, d_B, long_rows=0, variant=v)
        except ValueError:
            continue  # the variant does not cover this shape (invalid argument)
        ran += 1
        assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32)), (N, v)
    assert ran >= len(variants) - 3


def test_scheduled_product_is_graph_capturable(cmm, dev, oracle_mod):
    """The heavy launch forks onto the schedule's side stream and joins again: under capture that is a fork / join inside
    the graph; replays on new contents of the same buffers give the right bits."""
    M, K, N = 4000, 6000, 128
    lens = _pareto_lens(M, 25, 4000, seed=3)
    rowptr, col, val = _random_rows_csr(M, K, lens, seed=4)
    g = np.random.Generator(np.random.PCG64(5))
    B1, B2 = g.random((K, N), dtype=np.float32), g.random((K, N), dtype=np.float32)
    d_rp, d_col, d_val = (t(x, dev) for x in (rowptr, col, val))
    d_B = t(B1, dev)
    C = torch.empty(M, N, device=dev)
    sched = cmm.spmm_schedule(d_rp, len(val), M, N)
    sched.set_heavy(150, True)
    assert sched.info()["heavy_rows"] > 0
    cmm.naive_spmm_scheduled(sched, d_val, d_col, d_rp, len(val), M, K, d_B, C, None, 0)  # warm-up outside the capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            cmm.naive_spmm_scheduled(sched, d_val, d_col, d_rp, len(val), M, K, d_B, C, None, 0)
    for Bh in (B1, B2):
        d_B.copy_(t(Bh, dev))
        C.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(C.cpu().numpy().view(np.uint32), oracle_mod.spmm_csr(rowptr, col, val, M, K, Bh).view(np.uint32))


def test_inspector_handles_run_skewed_matrices_on_their_schedule(cmm, dev, oracle_mod):
    """cusparse_inspect / tiledspmm_inspect_csr build the row schedule of A and of Aᵀ (what an inspector is for: reference
    src/sparse_mm.cu:137-368 restructures A once); cusparse_mmul_opt / _t and tiledspmm_mm then run the column-major
    executor on it — the same bits as the oracle's column-major product (split order for the rows beyond the threshold),
    for a skewed matrix (schedule active, long rows prepared) and for a uniform one (schedule inactive)."""
    K, N = 30_000, 128
    for tag, lens in (("skewed", _pareto_lens(5000, 30, 12_000, seed=11, empty=0.02)), ("uniform", np.full(5000, 24))):
        M = len(lens)
        if tag == "skewed":
            lens[17] = 9500
        rowptr, col, val = _random_rows_csr(M, K, lens, seed=12)
        nnz = len(val)
        g = np.random.Generator(np.random.PCG64(13))
        Bt = g.random((N, K), dtype=np.float32) - 0.5      # column-major K x N  ==  row-major [N, K]
        d_rp, d_col, d_val = (t(x, dev) for x in (rowptr, col, val))
        cmm.cusparse_inspect(d_rp, d_col, d_val, nnz, M, N, K, "sk")
        info = cmm.inspect_info("sk", False)
        assert info["schedule_active"] is (tag == "skewed"), info
        Ct = torch.full((N, M), float("nan"), device=dev)
        cmm.cusparse_mmul_opt(t(Bt, dev), Ct, "sk")
        B = np.ascontiguousarray(Bt.T)
        want = (oracle_mod.spmm_csr_long if tag == "skewed" else oracle_mod.spmm_csr)(rowptr, col, val, M, K, B)
        assert np.array_equal(Ct.cpu().numpy().view(np.uint32), np.ascontiguousarray(want.T).view(np.uint32)), tag
        # the transposed product with the cached Aᵀ (its own schedule)
        Gt = g.random((N, M), dtype=np.float32) - 0.5
        Dt = torch.full((N, K), float("nan"), device=dev)
        cmm.cusparse_mmul_opt_t(t(Gt, dev), Dt, "sk")
        t_rp, t_col, t_val = oracle_mod.csr_transpose(rowptr, col, val, M, K)
        want_t = oracle_mod.spmm_csr_long(t_rp, t_col, t_val, K, M, np.ascontiguousarray(Gt.T))
        assert np.array_equal(Dt.cpu().numpy().view(np.uint32), np.ascontiguousarray(want_t.T).view(np.uint32)), tag
        cmm.cusparse_clean()
        # the tiled inspector (host CSR, int64 indices) ends on the same handle
        cmm.tiledspmm_inspect_csr(M, K, N, torch.from_numpy(rowptr.astype(np.int64)), torch.from_numpy(col.astype(np.int64)),
                                  torch.from_numpy(val), "tl")
        Ct.fill_(float("nan"))
        cmm.tiledspmm_mm(t(Bt, dev), Ct, "tl")
        assert np.array_equal(Ct.cpu().numpy().view(np.uint32), np.ascontiguousarray(want.T).view(np.uint32)), tag
        cmm.tiledspmm_clean()


def test_matmuls_keeps_a_schedule_on_a_csr_tensor_that_is_used_again(mm, cmm, dev, oracle_mod):
    """cusparseMM.apply(A_csr, b): the first product of a pattern runs plain and leaves a mark, the second builds the row
    schedule and keeps it on the tensor (per dense width), later ones reuse it; the backward's Aᵀ·dC does the same on the
    cached transposed pattern.  Forward values equal the oracle bit for bit every time, gradients equal torch autograd's."""
    M, K, N = 6000, 8000, 64
    lens = _pareto_lens(M, 20, 4000, seed=21, empty=0.02)
    rowptr, col, val = _random_rows_csr(M, K, lens, seed=22)
    g = np.random.Generator(np.random.PCG64(23))
    B = g.random((K, N), dtype=np.float32) - 0.5
    a = torch.sparse_csr_tensor(torch.from_numpy(rowptr.astype(np.int64)), torch.from_numpy(col.astype(np.int64)),
                                torch.from_numpy(val), (M, K)).to(dev).requires_grad_(True)
    b = t(B, dev).requires_grad_(True)
    want = oracle_mod.spmm_csr(rowptr, col, val, M, K, B)
    dC = g.random((M, N), dtype=np.float32)
    for i in range(3):
        b.grad = None
        out = mm.cusparseMM.apply(a, b)
        assert np.array_equal(out.detach().cpu().numpy().view(np.uint32), want.view(np.uint32)), i
        out.backward(t(dC, dev))
        book = a._mi_csr_sched[1]
        assert (book[N] == 'seen') if i == 0 else (book[N].info()["active"] and book[N].info()["heavy_rows"] > 0), (i, book)
        book_t = a._mi_csr_sched_t[1]
        assert (book_t[N] == 'seen') if i == 0 else (book_t[N].info()["rows"] == K)
        t_rp, t_col, t_val = oracle_mod.csr_transpose(rowptr, col, val, M, K)
        assert np.array_equal(b.grad.cpu().numpy().view(np.uint32), oracle_mod.spmm_csr(t_rp, t_col, t_val, K, M, dC).view(np.uint32)), i
    # another width: its own entry
    b2 = t(g.random((K, 32), dtype=np.float32), dev)
    mm.cusparseMM.apply(a, b2)
    assert a._mi_csr_sched[1][32] == 'seen' and a._mi_csr_sched[1][N] != 'seen'


def test_locality_order_is_taken_where_it_measurably_helps_and_changes_no_bit(cmm, dev, oracle_mod):
    """Given the columns, the inspector also tries the rows in the order of their median column inside a length class and
    MEASURES a window's footprint in B against the natural order (reference: the inspector that compacts each block's
    footprint of B, src/sparse_mm.cu:62-68,259).  A banded matrix whose rows arrive shuffled: taken (footprint more than
    halved), rows that run together centre on neighbouring columns; the same matrix unshuffled: the natural order is local
    already — declined; uniform columns: nothing to recover — declined.  The product has the oracle's bits either way."""
    M = K = 40_000
    N = 64
    g = np.random.Generator(np.random.PCG64(31))
    per = 24
    rows = np.repeat(np.arange(M), per)
    band_cols = np.clip(rows + g.integers(-300, 301, size=len(rows)), 0, K - 1)
    perm = g.permutation(M)

    def csr_of(r, c):
        keys = np.unique(r.astype(np.int64) * K + c)
        rr, cc = keys // K, (keys % K).astype(np.int32)
        rowptr = np.concatenate([[0], np.cumsum(np.bincount(rr, minlength=M))]).astype(np.int32)
        return rowptr, cc, g.random(len(cc), dtype=np.float32) - 0.5

    B = g.random((K, N), dtype=np.float32) - 0.5
    d_B = t(B, dev)
    for tag, (rowptr, col, val), expect in (("shuffled band", csr_of(perm[rows], band_cols), True),
                                            ("band", csr_of(rows, band_cols), False),
                                            ("uniform", csr_of(rows, g.integers(0, K, size=len(rows))), False)):
        d_rp, d_col, d_val = (t(x, dev) for x in (rowptr, col, val))
        sched = cmm.spmm_schedule(d_rp, len(val), M, N, d_col, K)
        info = sched.info()
        assert info["locality_order"] is expect, (tag, info)
        assert info["window_span_natural"] >= 0 and info["window_span_scheduled"] >= 0
        if expect:
            assert info["active"] and 2 * info["window_span_scheduled"] < info["window_span_natural"], info
            order = sched.order.cpu().numpy()
            med = col[(rowptr[:-1] + (rowptr[1:] - rowptr[:-1]) // 2).clip(max=len(col) - 1)][order].astype(np.int64)
            lens = np.diff(rowptr)[order]
            same = lens[1:] == lens[:-1]   # inside a length class the median columns ascend (up to the 2048-slot tiles' inner order)
            assert np.mean(np.abs(np.diff(med))[same]) < 0.02 * K
        elif tag == "band":
            assert info["window_span_natural"] <= 0.4
        sched.set_heavy(info["heavy_length"], True)   # run scheduled whatever the activity rule says
        got = _product(cmm, dev, sched, d_rp, d_col, d_val, M, K, d_B)
        assert np.array_equal(got.cpu().numpy().view(np.uint32), oracle_mod.spmm_csr(rowptr, col, val, M, K, B).view(np.uint32)), tag
    # without the columns the pass is not tried
    assert cmm.spmm_schedule(d_rp, len(val), M, N).info()["window_span_natural"] == -1.0


def test_plain_entry_points_build_their_own_schedule_without_synchronising(cmm, dev, oracle_mod):
    """custom_mm.naive_spmm has no inspector in the reference's API, and the reference's tests and benchmarks call it in loops on
    one matrix: the extension remembers the CSR arrays it has seen, enqueues the schedule's build behind the SECOND product of
    the same arrays (device kernels + copies to pinned memory + an event), and a later product that finds the event done runs on
    the schedule — with torch's synchronisation debug mode on "error" throughout.  Every product has the bits of the first;
    a stale key can cost time, never bits (a schedule is a permutation of the rows): shown by rewriting the arrays in place."""
    M, K, N = 20_000, 30_000, 64
    lens = _pareto_lens(M, 120, 6000, seed=41, empty=0.02)
    rowptr, col, val = _random_rows_csr(M, K, lens, seed=42)
    assert len(val) >= (1 << 20)
    g = np.random.Generator(np.random.PCG64(43))
    B = g.random((K, N), dtype=np.float32) - 0.5
    d_rp, d_col, d_val, d_B = (t(x, dev) for x in (rowptr, col, val, B))
    want = oracle_mod.spmm_csr(rowptr, col, val, M, K, B)
    cmm.auto_schedule_clear()
    assert cmm.auto_schedule_stats() == {"enabled": True, "entries": 0, "pending": 0, "active": 0, "inactive": 0, "built": cmm.auto_schedule_stats()["built"]}
    built0 = cmm.auto_schedule_stats()["built"]
    C = torch.empty(M, N, device=dev)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        states = []
        for i in range(5):
            C.fill_(float("nan"))
            cmm.naive_spmm(d_val, d_col, d_rp, len(val), M, K, d_B, C)
            states.append(dict(cmm.auto_schedule_stats()))
            torch.cuda.set_sync_debug_mode("default")
            torch.cuda.synchronize()     # (the test's own: lets the build's event complete between products)
            assert np.array_equal(C.cpu().numpy().view(np.uint32), want.view(np.uint32)), i
            torch.cuda.set_sync_debug_mode("error")
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert states[0]["entries"] == 1 and states[0]["pending"] == 0            # first product: seen, nothing built
    assert states[1]["pending"] == 1                                          # second: the build is enqueued behind it
    assert states[2]["pending"] == 0 and states[2]["active"] == 1 and states[2]["built"] == built0 + 1   # third: finished and in use
    assert states[4] == states[2]
    # the arrays rewritten IN PLACE (same pointers, same sizes, another matrix): the stale schedule is still a permutation
    # (with a row BEYOND the long-row threshold where the old matrix had a short one: the stale schedule does not know it, the
    # product must still list and split it — the scan ahead of the scheduled launches reads every row)
    lens2 = _pareto_lens(M, 100, 6000, seed=51, empty=0.02)
    lens2[int(np.argmin(lens[:100]))] = 9000   # (a short row of the old matrix, early enough to survive the cut below)
    rp2, col2, val2 = _random_rows_csr(M, K, lens2, seed=52)
    n2 = min(len(val2), len(val))
    keep = np.searchsorted(rp2, n2, side="right") - 1          # whole rows that fit the old arrays
    rp2 = np.minimum(rp2, rp2[keep]).astype(np.int32)
    col2, val2 = col2[:rp2[-1]], val2[:rp2[-1]]
    d_rp.copy_(t(rp2, dev))
    d_col[:len(col2)].copy_(t(col2, dev))
    d_val[:len(val2)].copy_(t(val2, dev))
    C.fill_(float("nan"))
    cmm.naive_spmm(d_val, d_col, d_rp, len(val), M, K, d_B, C)     # (the count is now a capacity: a valid count)
    assert int(np.diff(rp2).max()) == 9000
    assert np.array_equal(C.cpu().numpy().view(np.uint32), oracle_mod.spmm_csr_long(rp2, col2, val2, M, K, B).view(np.uint32))
    # a uniform matrix: its schedule is built once and found inactive
    rpu, colu, valu = _random_rows_csr(M, K, np.full(M, 60), seed=61)
    du = [t(x, dev) for x in (rpu, colu, valu)]
    for i in range(4):
        cmm.naive_spmm(du[2], du[1], du[0], len(valu), M, K, d_B, C)
        torch.cuda.synchronize()
    st = cmm.auto_schedule_stats()
    assert st["inactive"] == 1 and st["active"] == 1 and st["pending"] == 0
    assert np.array_equal(C.cpu().numpy().view(np.uint32), oracle_mod.spmm_csr(rpu, colu, valu, M, K, B).view(np.uint32))
    cmm.auto_schedule_clear()


def test_scheduled_products_fuzz_against_the_plain_product_and_the_oracle(cmm, dev, oracle_mod):
    """Random skewed matrices through the scheduled product — random widths (every column-part count of the staged kernel, widths
    that cannot move float4s), a random pinned heavy length (0: every row a heavy slot … none), rows around and beyond the long-row
    threshold (S = 1 and S > 1), the bias, prepared and scanned lists (long_rows −1 / 1) and the plain chain (0): the bits of the
    unscheduled product every time, and the oracle's on every third case.  MI_FUZZ_CASES / MI_FUZZ_SEED as the other fuzz tests."""
    import os
    g = np.random.Generator(np.random.PCG64(int(os.environ.get("MI_FUZZ_SEED", "909"))))
    cases = int(os.environ.get("MI_FUZZ_CASES", "10"))
    thr = cmm.long_row_threshold()
    for case in range(cases):
        M = int(g.integers(150, 2500))
        K = int(g.choice([12_000, 40_000, 70_000]))
        N = int(g.choice([4, 8, 16, 20, 36, 64, 68, 100, 128, 132, 192, 256, 260, 320, 512, 700, 33, 602]))
        lens = np.minimum(_pareto_lens(M, int(g.integers(4, 60)), K, seed=int(g.integers(1 << 30))), K)
        for _ in range(int(g.integers(0, 4))):
            lens[int(g.integers(0, M))] = min(K, int(g.choice([thr, thr + 1, thr + 700, 3 * thr, 65536, 66000])))
        while int(lens.sum()) * N > 120_000_000:
            lens[int(np.argmax(lens))] //= 2
        rowptr, col, val = _random_rows_csr(M, K, lens, seed=int(g.integers(1 << 30)))
        B = g.random((K, N), dtype=np.float32) - 0.5
        bias = g.random(N, dtype=np.float32) if g.integers(0, 3) == 0 else None
        d_rp, d_col, d_val, d_B = (t(x, dev) for x in (rowptr, col, val, B))
        d_bias = None if bias is None else t(bias, dev)
        sched = cmm.spmm_schedule(d_rp, len(val), M, N)
        heavy = int(g.choice([-1, 0, 8, 40, 300, 3000, 1 << 30]))
        if heavy >= 0:
            sched.set_heavy(heavy, bool(g.integers(0, 2)))
        what = (case, M, K, N, len(val), sorted(int(x) for x in lens if x >= thr), heavy, bias is not None)
        for long_rows in (-1, 1, 0):
            kw = {} if bias is None else {"bias": d_bias}
            plain = _product(cmm, dev, None, d_rp, d_col, d_val, M, K, d_B, long_rows=long_rows, **kw)
            got = _product(cmm, dev, sched, d_rp, d_col, d_val, M, K, d_B, long_rows=long_rows, **kw)
            assert torch.equal(got.view(torch.int32), plain.view(torch.int32)), (long_rows,) + what
            if case % 3 == 0 and long_rows != 1:
                want = oracle_mod.spmm_csr(rowptr, col, val, M, K, B) if (long_rows == 0 or N < 4) else oracle_mod.spmm_csr_long(rowptr, col, val, M, K, B)
                if bias is not None:
                    want = want + bias[None, :]
                assert np.array_equal(plain.cpu().numpy().view(np.uint32), want.view(np.uint32)), (long_rows,) + what


def test_process_exit_with_live_schedules_and_handles():
    """A program that never calls cusparse_clean / auto_schedule_clear must still END cleanly: handles and automatic schedules own
    HIP streams and events, which the extension releases from an atexit hook (left to static destructors they were destroyed
    after the HIP runtime: exit code 139 after the last line of output).  tests/exit_probe.py in a child process."""
    import subprocess
    import sys
    from pathlib import Path
    r = subprocess.run([sys.executable, str(Path(__file__).resolve().parent / "exit_probe.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("done"), (r.returncode, r.stdout[-500:], r.stderr[-2000:])
