"""Row-sharded SpMM over a 2-rank gloo group on the CPU (no GPU): partitioning,
rowptr rebasing (bit-exact integers), the in-place block-cyclic all-gather, and
equality of the gathered C with the single-rank result.  The local 2-d kernel is
the oracle here (test stand-in for custom_mm.naive_spmm)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_mm_op(vals, cols, offs, nnz, rows, kcols, B, C):
    import oracle
    C.copy_(torch.from_numpy(oracle.spmm_csr(offs.numpy(), cols.numpy()[:nnz], vals.numpy()[:nnz], rows, kcols,
                                             B.numpy())))
    return C


def _skewed_csr(M, K):
    """Row r has about K/(r+1) non-zeros: a few heavy rows at the top, a long light tail."""
    rng = np.random.Generator(np.random.PCG64(7))
    lens = np.maximum(1, K // (np.arange(M) + 1))
    lens[M // 2::5] = 0  # some empty rows, too
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    col = np.concatenate([np.sort(rng.choice(K, n, replace=False)) for n in lens]).astype(np.int32)
    val = rng.random(len(col), dtype=np.float32)
    return rowptr, col, val


def _emulated_all_to_all(outs, ins, group=None, async_op=False):
    """gloo has no list all_to_all: the same semantics (entry r of `ins` goes to rank r, entry r of `outs` comes from
    rank r, empty entries are skipped) out of isend / irecv, so the operator's index arithmetic can be tested on CPU."""
    rank = dist.get_rank(group)
    ops = []
    for r, (o, i) in enumerate(zip(outs, ins)):
        if r == rank:
            if i.numel():
                o.copy_(i)
            continue
        if o.numel():
            ops.append(dist.P2POp(dist.irecv, o, r, group=group))
        if i.numel():
            ops.append(dist.P2POp(dist.isend, i, r, group=group))
    works = dist.batch_isend_irecv(ops) if ops else []

    class _All:
        def wait(self):
            for w in works:
                w.wait()
    if async_op:
        return _All()
    _All().wait()


def _worker(rank, world, port, M, K, N, chunks, out_dir, split="rows", skew=False, exchange="allgather",
            refuse_in_place=False, emulate_all_to_all=False, refuse_on_rank=None, expect_exchange=None):
    for p in (str(REPO), str(REPO / "matrix-multiplication_amd")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        import sharded
        if refuse_in_place:
            # a torch / RCCL build that refuses the aliased in-place form: an argument check, raised on every rank
            # alike before anything is sent
            real = dist.all_gather_into_tensor

            def picky(output, input, *a, **k):
                lo, hi = output.data_ptr(), output.data_ptr() + output.numel() * output.element_size()
                if lo <= input.data_ptr() < hi:
                    raise RuntimeError("all_gather_into_tensor: input aliases output (refused by this build)")
                return real(output, input, *a, **k)
            dist.all_gather_into_tensor = picky
        if refuse_on_rank is not None and rank == refuse_on_rank:
            # ONE rank's build refuses the in-place form at its argument check (nothing enqueued there)
            real_ag = dist.all_gather_into_tensor

            def picky_here(output, input, *a, **k):
                lo, hi = output.data_ptr(), output.data_ptr() + output.numel() * output.element_size()
                if lo <= input.data_ptr() < hi:
                    raise RuntimeError("all_gather_into_tensor: input aliases output (refused on this rank)")
                return real_ag(output, input, *a, **k)
            dist.all_gather_into_tensor = picky_here
        if emulate_all_to_all:
            dist.all_to_all = _emulated_all_to_all
        rowptr, col, val = _skewed_csr(M, K) if skew else oracle.make_csr(M, K, 0.05, seed=0)
        B = torch.from_numpy(np.random.Generator(np.random.PCG64(1)).random((K, N), dtype=np.float32))
        op = sharded.ShardedSpMM(torch.from_numpy(rowptr), torch.from_numpy(col), torch.from_numpy(val), M, K, "cpu",
                                 chunks=chunks, mm_op=_oracle_mm_op, split=split, exchange=exchange)
        if expect_exchange is not None:
            assert op.exchange == expect_exchange, (op.exchange, op.fallbacks)
        # integer artefacts: block ownership and rebased rowptrs
        assert [b[0] for b in op.blocks] == [j * world + rank for j in range(chunks)]
        for blk, rp, ci, v, nnz, rows, has_long in op.blocks:
            r0, r1 = min(int(op.bounds[blk]), M), min(int(op.bounds[blk + 1]), M)
            expect = (rowptr[r0:r1 + 1].astype(np.int64) - rowptr[r0]).astype(np.int32)
            got = rp.numpy()
            assert rows == int(op.bounds[blk + 1] - op.bounds[blk]) and not has_long
            assert got.dtype == np.int32 and got[0] == 0 and len(got) == rows + 1
            assert np.array_equal(got[:r1 - r0 + 1], expect) and np.all(got[r1 - r0:] == expect[-1])
            assert nnz == rowptr[r1] - rowptr[r0] and np.array_equal(ci.numpy(), col[rowptr[r0]:rowptr[r1]])
        total = torch.tensor([op.local_nnz])
        dist.all_reduce(total)
        assert int(total) == len(val)
        C = op.forward(B)
        np.save(os.path.join(out_dir, f"c_{rank}.npy"), C.numpy())
        if refuse_in_place:
            assert op.exchange == "allgather_copy" and len(op.fallbacks) == 1 and "refused" in op.fallbacks[0]
            C2 = op.forward(B)  # stays on the fallback without a second refusal
            assert len(op.fallbacks) == 1 and torch.equal(C, C2)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("M,chunks", [(96, 2), (101, 3), (7, 4)])
def test_two_rank_gather_equals_single_rank(tmp_path, oracle_mod, M, chunks):
    K, N, world = 64, 24, 2
    mp.spawn(_worker, args=(world, _free_port(), M, K, N, chunks, str(tmp_path)), nprocs=world, join=True)
    rowptr, col, val = oracle_mod.make_csr(M, K, 0.05, seed=0)
    B = np.random.Generator(np.random.PCG64(1)).random((K, N), dtype=np.float32)
    single = oracle_mod.spmm_csr(rowptr, col, val, M, K, B)
    for r in range(world):
        got = np.load(tmp_path / f"c_{r}.npy")
        assert got.shape == (M, N)
        assert np.array_equal(got, single), f"rank {r}: gathered C differs from the single-rank result"


@pytest.mark.parametrize("M,chunks,skew", [(96, 2, True), (101, 3, True), (40, 4, False)])
def test_two_rank_nnz_balanced_split_equals_single_rank(tmp_path, oracle_mod, M, chunks, skew):
    """split="nnz": blocks of different heights, exchanged with in-place broadcasts."""
    K, N, world = 64, 24, 2
    mp.spawn(_worker, args=(world, _free_port(), M, K, N, chunks, str(tmp_path), "nnz", skew), nprocs=world, join=True)
    rowptr, col, val = _skewed_csr(M, K) if skew else oracle_mod.make_csr(M, K, 0.05, seed=0)
    B = np.random.Generator(np.random.PCG64(1)).random((K, N), dtype=np.float32)
    single = oracle_mod.spmm_csr(rowptr, col, val, M, K, B)
    for r in range(world):
        got = np.load(tmp_path / f"c_{r}.npy")
        assert got.shape == (M, N) and np.array_equal(got, single), f"rank {r}"


@pytest.mark.parametrize("M,chunks,split,skew,world", [(96, 2, "rows", False, 2), (101, 3, "nnz", True, 2), (7, 4, "rows", False, 2),
                                                       (90, 2, "nnz", True, 3)])
def test_direct_p2p_exchange_equals_single_rank(tmp_path, oracle_mod, M, chunks, split, skew, world):
    """exchange="p2p": every rank sends its block straight to every peer and receives the peers' blocks
    in place (one grouped batch of isend / irecv per step) — equal-row and nnz-balanced blocks, ragged tail,
    empty blocks, two and three ranks: the assembled C equals the single-rank result on every rank."""
    K, N = 64, 24
    mp.spawn(_worker, args=(world, _free_port(), M, K, N, chunks, str(tmp_path), split, skew, "p2p"), nprocs=world, join=True)
    rowptr, col, val = _skewed_csr(M, K) if skew else oracle_mod.make_csr(M, K, 0.05, seed=0)
    B = np.random.Generator(np.random.PCG64(1)).random((K, N), dtype=np.float32)
    single = oracle_mod.spmm_csr(rowptr, col, val, M, K, B)
    for r in range(world):
        got = np.load(tmp_path / f"c_{r}.npy")
        assert got.shape == (M, N) and np.array_equal(got, single), f"rank {r}"


@pytest.mark.parametrize("M,chunks,refuse", [(96, 2, True), (101, 3, True), (101, 3, False)])
def test_refused_in_place_gather_falls_back_to_gather_plus_copy(tmp_path, oracle_mod, M, chunks, refuse):
    """Round 3 (first contact with a multi-GPU node): if the in-place all_gather_into_tensor is what a torch / RCCL
    build refuses, the operator switches — on every rank alike, nothing having been sent — to the out-of-place
    collective + one copy per step and still returns the single-rank bits; exchange="allgather_copy" pins that form."""
    K, N, world = 64, 24, 2
    mp.spawn(_worker, args=(world, _free_port(), M, K, N, chunks, str(tmp_path), "rows", False,
                            "allgather" if refuse else "allgather_copy", refuse), nprocs=world, join=True)
    rowptr, col, val = oracle_mod.make_csr(M, K, 0.05, seed=0)
    B = np.random.Generator(np.random.PCG64(1)).random((K, N), dtype=np.float32)
    single = oracle_mod.spmm_csr(rowptr, col, val, M, K, B)
    for r in range(world):
        got = np.load(tmp_path / f"c_{r}.npy")
        assert got.shape == (M, N) and np.array_equal(got, single), f"rank {r}"


@pytest.mark.parametrize("M,chunks,split,skew,world", [(96, 2, "rows", False, 2), (101, 3, "nnz", True, 2), (7, 4, "rows", False, 2),
                                                       (90, 2, "nnz", True, 3)])
def test_all_to_all_exchange_equals_single_rank(tmp_path, oracle_mod, M, chunks, split, skew, world):
    """exchange="alltoall" (round 4): one list-form all_to_all per step — every rank's block to every peer, straight
    into its final position, own entry empty — on equal-row and nnz-balanced blocks, ragged tails, empty blocks, two
    and three ranks.  gloo has no list all_to_all, so the collective is emulated here out of isend / irecv (same
    semantics); what is tested is the operator's index arithmetic and the construction-time probe."""
    K, N = 64, 24
    mp.spawn(_worker, args=(world, _free_port(), M, K, N, chunks, str(tmp_path), split, skew, "alltoall", False, True, None,
                            "alltoall"), nprocs=world, join=True)
    rowptr, col, val = _skewed_csr(M, K) if skew else oracle_mod.make_csr(M, K, 0.05, seed=0)
    B = np.random.Generator(np.random.PCG64(1)).random((K, N), dtype=np.float32)
    single = oracle_mod.spmm_csr(rowptr, col, val, M, K, B)
    for r in range(world):
        got = np.load(tmp_path / f"c_{r}.npy")
        assert got.shape == (M, N) and np.array_equal(got, single), f"rank {r}"


def test_all_to_all_refused_by_the_backend_falls_back_to_the_all_gather_by_agreement(tmp_path, oracle_mod):
    """A backend without the list form (gloo: "does not support alltoall", raised on every rank at call time): the
    construction-time probe catches it, the ranks agree (all-reduce MIN) and every rank runs the in-place all-gather."""
    M, K, N, world, chunks = 101, 64, 24, 2, 3
    mp.spawn(_worker, args=(world, _free_port(), M, K, N, chunks, str(tmp_path), "rows", False, "alltoall", False, False, None,
                            "allgather"), nprocs=world, join=True)
    rowptr, col, val = oracle_mod.make_csr(M, K, 0.05, seed=0)
    B = np.random.Generator(np.random.PCG64(1)).random((K, N), dtype=np.float32)
    single = oracle_mod.spmm_csr(rowptr, col, val, M, K, B)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"c_{r}.npy"), single), f"rank {r}"


def test_push_exchange_without_device_buffers_falls_back_by_agreement(tmp_path, oracle_mod):
    """exchange="push" (round 5: peers' C buffers mapped through CUDA IPC, blocks copied straight into them) needs device
    buffers: on CPU tensors the construction-time probe refuses on every rank, the ranks agree and move to the list
    all_to_all — which gloo refuses in turn — and end on the in-place all-gather, with the single-rank bits."""
    M, K, N, world, chunks = 101, 64, 24, 2, 3
    mp.spawn(_worker, args=(world, _free_port(), M, K, N, chunks, str(tmp_path), "rows", False, "push", False, False, None,
                            "allgather"), nprocs=world, join=True)
    rowptr, col, val = oracle_mod.make_csr(M, K, 0.05, seed=0)
    B = np.random.Generator(np.random.PCG64(1)).random((K, N), dtype=np.float32)
    single = oracle_mod.spmm_csr(rowptr, col, val, M, K, B)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"c_{r}.npy"), single), f"rank {r}"


def test_probe_refused_on_one_rank_only_moves_every_rank_to_the_fallback(tmp_path, oracle_mod):
    """The probe's outcome is AGREED, not assumed identical on every rank: here rank 1 alone judges its probe of the
    in-place all_gather_into_tensor a failure (the collective went through, so nothing is left pending on rank 0); the
    all-reduce carries that verdict to rank 0, both ranks end on allgather_copy and return the single-rank bits."""
    M, K, N, world, chunks = 96, 64, 24, 2, 2
    mp.spawn(_worker_disagree, args=(world, _free_port(), M, K, N, chunks, str(tmp_path)), nprocs=world, join=True)
    rowptr, col, val = oracle_mod.make_csr(M, K, 0.05, seed=0)
    B = np.random.Generator(np.random.PCG64(1)).random((K, N), dtype=np.float32)
    single = oracle_mod.spmm_csr(rowptr, col, val, M, K, B)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"c_{r}.npy"), single), f"rank {r}"


def _worker_disagree(rank, world, port, M, K, N, chunks, out_dir):
    for p in (str(REPO), str(REPO / "matrix-multiplication_amd")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        import sharded
        real = sharded.ShardedSpMM._agreed
        seen = []

        def agreed(self, ok):
            # rank 1 reports that its probe of the in-place form failed although the collective itself went through
            # (e.g. it delivered the wrong blocks there): the all-reduce must carry that verdict to rank 0
            mine = ok and not (rank == 1 and self.exchange == "allgather")
            out = real(self, mine)
            seen.append((self.exchange, mine, out))
            return out
        sharded.ShardedSpMM._agreed = agreed
        rowptr, col, val = oracle.make_csr(M, K, 0.05, seed=0)
        B = torch.from_numpy(np.random.Generator(np.random.PCG64(1)).random((K, N), dtype=np.float32))
        op = sharded.ShardedSpMM(torch.from_numpy(rowptr), torch.from_numpy(col), torch.from_numpy(val), M, K, "cpu",
                                 chunks=chunks, mm_op=_oracle_mm_op, exchange="allgather")
        assert seen == [("allgather", rank == 0, False)], seen
        assert op.exchange == "allgather_copy" and len(op.fallbacks) == 1
        np.save(os.path.join(out_dir, f"c_{rank}.npy"), op.forward(B).numpy())
    finally:
        dist.destroy_process_group()


def _worker_p2p_probe(rank, world, port, fail_rank, out_dir):
    for p in (str(REPO), str(REPO / "matrix-multiplication_amd")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sharded
        ok = sharded.probe_p2p("cpu", timeout_s=3.0, _fail_here=(rank == fail_rank))
        # the default group is still usable afterwards, whatever the probe's group went through
        t = torch.tensor([rank + 1.0])
        dist.all_reduce(t)
        assert float(t) == world * (world + 1) / 2
        with open(os.path.join(out_dir, f"p2p_{rank}.txt"), "w") as f:
            f.write(str(ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,fail_rank", [(2, None), (2, 1), (3, 0)])
def test_p2p_probe_is_agreed_and_a_failure_on_one_rank_keeps_everyone_on_the_collective(tmp_path, world, fail_rank):
    """bench.py --exchange try-p2p probes direct sends BEFORE its trial, in the probe's own group with a short
    timeout: all ranks fine → True everywhere; one rank fails before posting its sends (its peers' receives time out in
    the probe group) → False everywhere, no exception, and the default group still works."""
    mp.spawn(_worker_p2p_probe, args=(world, _free_port(), fail_rank, str(tmp_path)), nprocs=world, join=True)
    got = [(tmp_path / f"p2p_{r}.txt").read_text() for r in range(world)]
    assert got == [str(fail_rank is None)] * world, got


def test_balanced_boundaries_are_exact_lower_bounds():
    """Integer artefact of the nnz-balanced split: boundary i is the FIRST row whose offset reaches
    i*nnz // nblocks (lower_bound), checked against a plain Python scan."""
    sys.path.insert(0, str(REPO / "matrix-multiplication_amd"))
    import sharded
    rng = np.random.Generator(np.random.PCG64(3))
    for M, nblocks in [(1, 1), (5, 8), (64, 4), (1000, 16), (257, 6)]:
        lens = rng.integers(0, 50, size=M) * (rng.random(M) < 0.7)
        lens[rng.integers(0, M)] += 5000  # a hub row
        rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        b = sharded.balanced_boundaries(rowptr, nblocks)
        nnz = int(rowptr[-1])
        assert b.dtype == np.int64 and len(b) == nblocks + 1 and b[0] == 0 and b[-1] == M
        assert np.all(np.diff(b) >= 0)
        for i in range(1, nblocks):
            target = i * nnz // nblocks
            first = next(r for r in range(M + 1) if rowptr[r] >= target)
            assert b[i] == first, (M, nblocks, i)
        # balance: no block exceeds its share by more than one row's worth of non-zeros
        per = np.diff(rowptr[b].astype(np.int64))
        assert per.sum() == nnz and per.max() <= -(-nnz // nblocks) + lens.max()
    uniform = np.arange(0, 1001, 10, dtype=np.int32)  # 100 rows of 10
    assert sharded.balanced_boundaries(uniform, 4).tolist() == [0, 25, 50, 75, 100]


def test_single_process_layout(oracle_mod):
    sys.path.insert(0, str(REPO / "matrix-multiplication_amd"))
    import sharded
    assert sharded.block_layout(1 << 20, 8, 4) == (32768, 1 << 20)
    assert sharded.block_layout(101, 2, 3) == (17, 102)
    assert sharded.owned_blocks(1, 2, 3) == [1, 3, 5]
    rp = torch.tensor([0, 2, 2, 5, 9], dtype=torch.int32)
    assert sharded.shard_rowptr(rp, 1, 3, 4).tolist() == [0, 0, 3]
    assert sharded.shard_rowptr(rp, 3, 6, 4).tolist() == [0, 4, 4, 4]  # padded tail rows are empty


def _shared_buffer_stand_ins(sharded, out_dir, rank, log=None):
    """CPU stand-ins for the three mapping calls of the push exchange (hipIpc on the GPU): output buffers are float32
    tensors over shared files, `export` hands out the file's path, `open` maps the same file in the peer.  Everything else
    — registration, agreements, fences, the copies' index arithmetic, release — is the product code."""
    paths, counter = {}, [0]

    def new_buffer(self, rows, cols):
        counter[0] += 1
        path = os.path.join(out_dir, f"buf_r{rank}_{counter[0]}.bin")
        t = torch.from_file(path, shared=True, size=max(rows * cols, 1), dtype=torch.float32)[:rows * cols].view(rows, cols)
        paths[t.data_ptr()] = path
        return t

    def export(self, out):
        return ("file", paths[out.data_ptr()])

    def open_peer(self, payload, shape):
        n = int(np.prod(shape))
        if log is not None:
            log.append(("open", payload[1]))
        return torch.from_file(payload[1], shared=True, size=max(n, 1), dtype=torch.float32)[:n].view(*shape)

    def close_peer(self, payload):
        if log is not None:
            log.append(("close", payload[1]))

    sharded.ShardedSpMM._new_buffer = new_buffer
    sharded.ShardedSpMM._export_buffer = export
    sharded.ShardedSpMM._open_peer = open_peer
    sharded.ShardedSpMM._close_peer = close_peer


def _worker_push(rank, world, port, M, K, N, chunks, out_dir, split):
    for p in (str(REPO), str(REPO / "matrix-multiplication_amd")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        import sharded
        log = []
        _shared_buffer_stand_ins(sharded, out_dir, rank, log)
        real_all_reduce, real_barrier, real_push = dist.all_reduce, dist.barrier, sharded.ShardedSpMM._push_block

        def all_reduce(t, *a, **k):
            log.append(("all_reduce", tuple(t.shape)))
            return real_all_reduce(t, *a, **k)

        def barrier(*a, **k):
            log.append(("barrier", k.get("group")))
            return real_barrier(*a, **k)

        def push(self, dst, src):
            log.append(("push", tuple(src.shape)))
            return real_push(self, dst, src)
        dist.all_reduce, dist.barrier, sharded.ShardedSpMM._push_block = all_reduce, barrier, push

        rowptr, col, val = _skewed_csr(M, K) if split == "nnz" else oracle.make_csr(M, K, 0.05, seed=0)
        B1 = torch.from_numpy(np.random.Generator(np.random.PCG64(1)).random((K, N), dtype=np.float32))
        B2 = torch.from_numpy(np.random.Generator(np.random.PCG64(2)).random((K, N), dtype=np.float32))
        op = sharded.ShardedSpMM(torch.from_numpy(rowptr), torch.from_numpy(col), torch.from_numpy(val), M, K, "cpu",
                                 chunks=chunks, mm_op=_oracle_mm_op, split=split, exchange="push")   # group=None: the default group
        assert op.exchange == "push" and op.fallbacks == [], op.fallbacks
        # the probe registered a tiny buffer, pushed, released: its mappings are closed, one barrier behind the closes
        assert [e[0] for e in log].count("open") == world - 1 and [e[0] for e in log].count("close") == world - 1
        assert ("barrier", None) in log and op._registered == {}
        del log[:]
        C = op.forward(B1)                       # the operator's own persistent registered buffer
        kinds = [e[0] for e in log]
        # entry fence (2 ints) BEFORE the first copy, exit fence (1 int) behind the last one
        assert ("all_reduce", (2,)) in log and ("all_reduce", (1,)) in log
        pushes = [i for i, k in enumerate(kinds) if k == "push"]
        if pushes:
            assert log.index(("all_reduce", (2,))) < pushes[0] and pushes[-1] < log.index(("all_reduce", (1,)))
        np.save(os.path.join(out_dir, f"c_{rank}.npy"), C.numpy().copy())
        opens_after_first = kinds.count("open")
        assert opens_after_first == world - 1    # the buffer was mapped once …
        del log[:]
        C2 = op.forward(B2)                      # … and a second product maps nothing, pushes into the same registration
        assert C2.data_ptr() == C.data_ptr() and "open" not in [e[0] for e in log] and "barrier" not in [e[0] for e in log]
        np.save(os.path.join(out_dir, f"c2_{rank}.npy"), C2.numpy().copy())
        # a caller's own buffer: an unregistered one is refused before any collective, a registered one is pushed into
        mine = op._new_buffer(op.padded_rows, N)
        del log[:]
        with pytest.raises(ValueError, match="REGISTERED"):
            op.forward(B1, out=mine)
        assert log == []
        assert op.register_output(mine) and op.register_output(mine)   # (the second call: already registered, no collective)
        C3 = op.forward(B1, out=mine)
        np.save(os.path.join(out_dir, f"c3_{rank}.npy"), C3.numpy().copy())
        # ranks that hand in buffers of DIFFERENT registrations are told so (one call later, or at release)
        other = op._new_buffer(op.padded_rows, N)
        assert op.register_output(other)
        op.forward(B1, out=mine if rank == 0 else other)
        with pytest.raises(RuntimeError, match="different registrations"):
            op.release_peers()
        del log[:]
        op.release_peers()      # (the registrations are still there: the check raised before anything was dropped)
        kinds = [e[0] for e in log]
        # release on the DEFAULT group: every mapping closed (3 buffers x the peers), then exactly one barrier
        assert kinds.count("close") == 3 * (world - 1) and kinds.count("barrier") == 1 and kinds[-1] == "barrier", log
        assert op._registered == {} and op._own_out == {}
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("M,chunks,split,world", [(96, 2, "rows", 2), (101, 3, "nnz", 2), (90, 2, "rows", 3), (7, 4, "rows", 2)])
def test_push_exchange_end_to_end_on_shared_memory_stand_ins(tmp_path, oracle_mod, M, chunks, split, world):
    """Round 6: the whole push exchange over gloo with the hipIpc calls replaced by shared files — construction probe,
    collective registration (the default group: group=None), entry fence before the first copy, exit fence behind the
    last, a second product into the same registration without a new mapping, an unregistered `out` refused before any
    collective, buffers of different registrations detected, and release_peers() closing every mapping behind ONE
    barrier on the default group (round 5's release returned early there and issued none)."""
    K, N = 64, 24
    mp.spawn(_worker_push, args=(world, _free_port(), M, K, N, chunks, str(tmp_path), split), nprocs=world, join=True)
    rowptr, col, val = _skewed_csr(M, K) if split == "nnz" else oracle_mod.make_csr(M, K, 0.05, seed=0)
    for name, seed in (("c", 1), ("c2", 2), ("c3", 1)):
        B = np.random.Generator(np.random.PCG64(seed)).random((K, N), dtype=np.float32)
        single = oracle_mod.spmm_csr(rowptr, col, val, M, K, B)
        for r in range(world):
            assert np.array_equal(np.load(tmp_path / f"{name}_{r}.npy"), single), f"{name}, rank {r}"


@pytest.mark.parametrize("fail_at", ["probe", "first_product"])
def test_push_mapping_failure_on_one_rank_only_keeps_every_rank_in_step(tmp_path, oracle_mod, fail_at):
    """The IPC push exchange maps every peer's buffer (a collective hand-over of handles, then a per-rank open that can
    fail on one rank alone — IPC limits, a refused mapping).  Here rank 1's open fails AFTER the hand-over while ranks 0
    and 2 succeed: the registration's agreement must carry that to everybody WITH rank 1's full message, every rank
    closes what it had opened and ends on the same collective form, nobody is left waiting in a collective its peer never
    enters, and the product is the single-rank product bit for bit.  `first_product`: the probe's tiny buffer maps
    everywhere (push is agreed) and the mapping of the real output buffer fails on rank 1 — alloc_output agrees on that
    too and every rank moves to the collective forms before anything was exchanged."""
    M, K, N, world, chunks = 120, 64, 24, 3, 2
    mp.spawn(_worker_partial_push, args=(world, _free_port(), M, K, N, chunks, str(tmp_path), fail_at), nprocs=world, join=True)
    rowptr, col, val = oracle_mod.make_csr(M, K, 0.05, seed=0)
    B = np.random.Generator(np.random.PCG64(1)).random((K, N), dtype=np.float32)
    single = oracle_mod.spmm_csr(rowptr, col, val, M, K, B)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"c_{r}.npy"), single), f"rank {r}"
        assert (tmp_path / f"form_{r}.txt").read_text() == (tmp_path / "form_0.txt").read_text() == "allgather"
        text = (tmp_path / f"fallbacks_{r}.txt").read_text()
        assert ("push refused" if fail_at == "probe" else "not mappable") in text
        # every rank knows WHO failed doing WHAT, and the message is whole (round 5 cut it at 120 characters)
        assert "rank 1 mapping rank 0's buffer: RuntimeError: hipIpcOpenMemHandle: simulated refusal" in text
        assert "x" * 200 in text


def _worker_partial_push(rank, world, port, M, K, N, chunks, out_dir, fail_at):
    for p in (str(REPO), str(REPO / "matrix-multiplication_amd")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        import sharded
        log = []
        _shared_buffer_stand_ins(sharded, out_dir, rank, log)
        real_open = sharded.ShardedSpMM._open_peer

        def open_peer(self, payload, shape):
            # the per-rank open: fails on rank 1 only (the probe's buffer has `world` rows)
            if self.rank == 1 and (fail_at == "probe" or shape[0] > self.world):
                raise RuntimeError("hipIpcOpenMemHandle: simulated refusal on this rank " + "x" * 200)
            return real_open(self, payload, shape)
        sharded.ShardedSpMM._open_peer = open_peer
        rowptr, col, val = oracle.make_csr(M, K, 0.05, seed=0)
        B = np.random.Generator(np.random.PCG64(1)).random((K, N), dtype=np.float32)
        op = sharded.ShardedSpMM(torch.from_numpy(rowptr), torch.from_numpy(col), torch.from_numpy(val), M, K,
                                 torch.device("cpu"), chunks=chunks, exchange="push", mm_op=_oracle_mm_op)
        assert op.exchange == ("allgather" if fail_at == "probe" else "push")
        C = op.forward(torch.from_numpy(B))
        # whatever a rank had opened before the refusal was agreed is closed again
        assert [e[0] for e in log].count("open") == [e[0] for e in log].count("close")
        op.release_peers()
        np.save(Path(out_dir) / f"c_{rank}.npy", C[:M].numpy())
        (Path(out_dir) / f"form_{rank}.txt").write_text(op.exchange)
        (Path(out_dir) / f"fallbacks_{rank}.txt").write_text(" | ".join(op.fallbacks))
    finally:
        dist.destroy_process_group()
