"""Row-sharded product on one GPU / a one-rank RCCL group, and the bench.py --gpus N rehearsal.

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


def test_sharded_layout_on_one_gpu(cmm, dev, oracle_mod):
    """The row-sharded driver with world = 1: block-cyclic chunks computed in place give the
    same bits as one launch (the N>1 collective path is covered by tests/test_sharded_cpu.py)."""
    import sharded
    M, K, N = 10_001, 8_000, 256
    rowptr, col, val = oracle_mod.make_csr(M, K, 0.005, seed=4)
    B = t(np.random.Generator(np.random.PCG64(4)).random((K, N), dtype=np.float32), dev)
    op = sharded.ShardedSpMM(torch.from_numpy(rowptr), torch.from_numpy(col), torch.from_numpy(val), M, K, dev, chunks=4)
    C = op.forward(B)
    single = torch.empty(M, N, device=dev)
    cmm.naive_spmm(t(val, dev), t(col, dev), t(rowptr, dev), len(val), M, K, B, single)
    assert C.shape == (M, N) and torch.equal(C, single)


def test_sharded_collective_on_a_one_rank_rccl_group(cmm, dev, oracle_mod):
    """The RCCL leg of sharded.ShardedSpMM (in-place all_gather_into_tensor on RCCL's stream beside
    the next chunk's kernel) under a real NCCL(=RCCL) process group of one rank."""
    import os
    import torch.distributed as dist
    import sharded
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    opts = dist.ProcessGroupNCCL.Options()
    opts.is_high_priority_stream = True   # as bench.py creates it: the gather must not queue behind the SpMM
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, pg_options=opts)
    try:
        M, K, N = 4099, 3000, 256
        rowptr, col, val = oracle_mod.make_csr(M, K, 0.01, seed=6)
        B = t(np.random.Generator(np.random.PCG64(6)).random((K, N), dtype=np.float32), dev)
        op = sharded.ShardedSpMM(torch.from_numpy(rowptr), torch.from_numpy(col), torch.from_numpy(val), M, K, dev,
                                 chunks=3)
        C = op.forward(B, force_collective=True)
        torch.cuda.synchronize()
        want = oracle_mod.spmm_csr(rowptr, col, val, M, K, B.cpu().numpy())
        assert np.array_equal(C.cpu().numpy(), want)
        # the nnz-balanced split exchanges with in-place RCCL broadcasts (blocks of different heights)
        op2 = sharded.ShardedSpMM(torch.from_numpy(rowptr), torch.from_numpy(col), torch.from_numpy(val), M, K, dev,
                                  chunks=5, split="nnz")
        C2 = op2.forward(B, force_collective=True)
        torch.cuda.synchronize()
        assert np.array_equal(C2.cpu().numpy(), want)
        # round 4: the list-form all_to_all exchange on RCCL (own entry empty on both sides), its construction-time
        # probe and the agreement all-reduce, the gather-only leg, and the direct-send probe in its own group
        for split in ("rows", "nnz"):
            op3 = sharded.ShardedSpMM(torch.from_numpy(rowptr), torch.from_numpy(col), torch.from_numpy(val), M, K, dev,
                                      chunks=4, split=split, exchange="alltoall")
            op3._probe_exchange()   # (a one-rank group skips it at construction)
            assert op3.exchange == "alltoall" and op3.fallbacks == [], op3.fallbacks
            out3 = op3.alloc_output(N)
            C3 = op3.forward(B, out=out3, force_collective=True)
            op3.forward(B, out=out3, force_collective=True, compute=False)   # exchanges only: C unchanged
            torch.cuda.synchronize()
            assert np.array_equal(C3.cpu().numpy(), want), split
        op._probe_exchange()
        assert op.exchange == "allgather" and op.fallbacks == []
        src, dst = torch.arange(8, device=dev, dtype=torch.float32), torch.zeros(8, device=dev)
        dist.all_to_all([dst], [src])   # RCCL's list form with a non-empty entry
        assert torch.equal(src, dst)
        assert sharded.probe_p2p(dev, timeout_s=20.0) is True
    finally:
        dist.destroy_process_group()


def test_sharded_hub_rows_follow_the_whole_problems_rule(cmm, dev, oracle_mod):
    """Row shards pick their own kernels (a 512-row shard of a SLAB-plan matrix runs a row-split plan),
    but rows beyond 8192 non-zeros are summed the way the WHOLE matrix's plan sums them, so the
    sharded result is bit-identical to the single-GPU one in both regimes."""
    import sharded
    # (a) whole problem: SLAB plan (no split) — shards: row-split plans, must not split either
    M, K, N = 4096, 12000, 1024
    rowptr, col, val = _moderately_dense_with_hub_rows(M, K, 0.5, [3, 2500, M - 1], 41)
    d_B = t(np.random.Generator(np.random.PCG64(42)).random((K, N), dtype=np.float32), dev)
    single = torch.empty(M, N, device=dev)
    assert cmm.spmm_plan(len(val), M, K, d_B, single)[3] is False
    cmm.naive_spmm(t(val, dev), t(col, dev), t(rowptr, dev), len(val), M, K, d_B, single)
    for split in ("rows", "nnz"):
        op = sharded.ShardedSpMM(torch.from_numpy(rowptr), torch.from_numpy(col), torch.from_numpy(val), M, K, dev,
                                 chunks=8, split=split)
        assert sum(b[6] for b in op.blocks) == 3  # three blocks hold a hub row
        assert cmm.spmm_plan(op.blocks[0][4], op.blocks[0][5], K, d_B, single[:op.blocks[0][5]])[1] != "spmm_slab_kernel"
        assert torch.equal(op.forward(d_B), single), split
    # (b) whole problem: row-split plan that splits its long rows — shards must split them too
    M, K, N = 301, 30000, 256
    g = np.random.Generator(np.random.PCG64(43))
    lens = g.integers(0, 200, size=M)
    lens[5], lens[17], lens[18], lens[150], lens[300] = K, 8193, 8192, 20011, 9000
    cols = [np.sort(g.choice(K, size=int(n), replace=False)).astype(np.int32) for n in lens]
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    col, val = np.concatenate(cols), g.random(int(lens.sum()), dtype=np.float32)
    B = g.random((K, N), dtype=np.float32)
    d_B = t(B, dev)
    single = torch.empty(M, N, device=dev)
    assert cmm.spmm_plan(len(val), M, K, d_B, single)[3] is True
    cmm.naive_spmm(t(val, dev), t(col, dev), t(rowptr, dev), len(val), M, K, d_B, single)
    assert np.array_equal(single.cpu().numpy(), oracle_mod.spmm_csr_long(rowptr, col, val, M, K, B))
    for split in ("rows", "nnz"):
        op = sharded.ShardedSpMM(torch.from_numpy(rowptr), torch.from_numpy(col), torch.from_numpy(val), M, K, dev,
                                 chunks=4, split=split)
        assert torch.equal(op.forward(d_B), single), split


def test_bench_multi_gpu_path_rehearsal(dev):
    """`python bench.py --gpus 2` started plainly: the parent spawns two child ranks itself (here both on
    the one GPU over gloo, MI_BENCH_REHEARSE=1 — a functional rehearsal, labelled as such, never a
    measurement), the ranks shard A's rows, exchange C block-cyclically, check a peer's block bit for
    bit, pick a chunk count from the pre-timing trial, and rank 0 prints ONE JSON line."""
    import json
    import os
    import subprocess
    from pathlib import Path
    repo = Path(__file__).resolve().parent.parent
    env = dict(os.environ, MI_BENCH_REHEARSE="1")
    proc = subprocess.run([sys.executable, str(repo / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                           "--workload", "c2"], capture_output=True, text=True, timeout=600, env=env)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong" and "REHEARSAL" in rec["data"]
    cfg = rec["config"]
    # the default trial times the two COLLECTIVE forms (in-place all-gather, list all_to_all); gloo has no list
    # all_to_all: the construction-time probe sees the refusal, the ranks agree and that candidate folds into the
    # all-gather (recorded in exchange_fallbacks).  Direct sends can half-fail: they are tried on request only.
    # (the IPC push exchange joins the trial on request only: --exchange try-push, below)
    assert cfg["rccl_ranks"] == 2 and cfg["chunks"] in (2, 4) and cfg["exchange"] == "allgather"
    assert set(cfg["chunk_trials_ms_per_step"]) == {"allgather/2", "allgather/4"}
    assert len(cfg["exchange_fallbacks"]) == 1 and "alltoall refused" in cfg["exchange_fallbacks"][0]
    assert [d["rank"] for d in cfg["rank_devices"]] == [0, 1] and all("name" in d and "pid" in d for d in cfg["rank_devices"])
    assert cfg["compute_only_ms_per_step"] > 0 and rec["value"] > 0 and "cpu_baseline" not in rec
    # the gather-only leg: its time and the implied per-rank receive rate are in the line
    assert cfg["gather_only_ms_per_step"] > 0 and cfg["gather_receive_GBps_per_rank"] > 0 and cfg["p2p_probe_ok"] is None
    proc = subprocess.run([sys.executable, str(repo / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                           "--workload", "c2", "--exchange", "try-p2p"], capture_output=True, text=True, timeout=600, env=env)
    assert proc.returncode == 0, proc.stderr[-3000:]
    cfg = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith('{"metric"')][0])["config"]
    assert cfg["p2p_probe_ok"] is True  # probed in its own group before the trial
    assert set(cfg["chunk_trials_ms_per_step"]) == {"allgather/2", "allgather/4", "p2p/2", "p2p/4"}
    assert cfg["exchange"] in ("allgather", "p2p")
    # the IPC push exchange in the trial (round 5): CUDA IPC works between the two processes on this GPU
    proc = subprocess.run([sys.executable, str(repo / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                           "--workload", "c2", "--exchange", "try-push"], capture_output=True, text=True, timeout=600, env=env)
    assert proc.returncode == 0, proc.stderr[-3000:]
    cfg = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith('{"metric"')][0])["config"]
    assert set(cfg["chunk_trials_ms_per_step"]) == {"allgather/2", "allgather/4", "push/2", "push/4"}
    assert cfg["exchange"] in ("allgather", "push")
    # and the nnz-balanced split (in-place broadcasts) through the same driver
    proc = subprocess.run([sys.executable, str(repo / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                           "--workload", "c2", "--split", "nnz", "--chunks", "3", "--exchange", "allgather"],
                          capture_output=True, text=True, timeout=600, env=env)
    assert proc.returncode == 0, proc.stderr[-3000:]
    rec = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith('{"metric"')][0])
    assert rec["config"]["chunks"] == 3 and "nnz-balanced" in rec["config"]["parallelism"]
    assert rec["config"]["exchange"] == "allgather" and rec["config"]["chunk_trials_ms_per_step"] is None
    # … and with direct sends to every peer
    proc = subprocess.run([sys.executable, str(repo / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                           "--workload", "c2", "--split", "nnz", "--chunks", "2", "--exchange", "p2p"],
                          capture_output=True, text=True, timeout=600, env=env)
    assert proc.returncode == 0, proc.stderr[-3000:]
    rec = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith('{"metric"')][0])
    assert rec["config"]["exchange"] == "p2p" and "direct RCCL sends" in rec["config"]["parallelism"]
    # … and with the IPC push exchange (round 5): both ranks map each other's C (two processes on this one GPU) and copy
    # their blocks straight into it; bench.py's own checks (a peer's block recomputed locally, the gathered C against rank
    # 0's single-GPU product) hold bit for bit
    proc = subprocess.run([sys.executable, str(repo / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                           "--workload", "c2", "--chunks", "2", "--exchange", "push"],
                          capture_output=True, text=True, timeout=600, env=env)
    assert proc.returncode == 0, proc.stderr[-3000:]
    rec = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith('{"metric"')][0])
    assert rec["config"]["exchange"] == "push" and rec["config"]["exchange_fallbacks"] == [], rec["config"]
    assert rec["config"]["gathered_C_equals_single_gpu_product"] is True and rec["config"]["speedup_vs_single_gpu_on_rank0"] > 0


@pytest.mark.parametrize("world", [2, 3])
def test_push_exchange_between_processes_on_one_gpu_with_a_reader_between_products(dev, tmp_path, world):
    """Round 6 (review: the push exchange's cross-rank write-after-read hazard, its torch-IPC abort, its per-call cache):
    `world` processes on this one GPU map each other's C through mi_ipc_* (hipIpc handles, explicit lifetimes) and push
    their blocks.  tests/push_rehearsal.py checks on every rank: the gathered product equals the single-GPU product bit
    for bit; a reader of C that is still pending when the next product is issued sees the OLD product everywhere (the
    entry fence orders the peers' pushes behind it); a second product opens no new mapping; a caller's registered buffer
    works and an unregistered one is refused before any collective; after release_peers() nothing is left mapped."""
    import json
    import os
    import socket
    import subprocess
    from pathlib import Path
    repo = Path(__file__).resolve().parent.parent
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = tmp_path / "push.json"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    proc = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                           "--master-addr", "127.0.0.1", "--master-port", str(port),
                           str(repo / "tests" / "push_rehearsal.py"), str(out)],
                          capture_output=True, text=True, timeout=600, env=env)
    assert proc.returncode == 0, proc.stderr[-4000:]
    recs = json.loads(out.read_text())
    assert len(recs) == world
    for rec in recs:
        assert rec["exchange"] == "push" and rec["fallbacks"] == [] and rec["fallbacks_end"] == [], rec
        assert rec["first_product_bit_exact"] and rec["second_product_bit_exact"] and rec["callers_buffer_bit_exact"], rec
        assert rec["reader_saw_the_next_product"] == [False, False, False], rec
        assert rec["unregistered_refused"] is True
        # the probe's mappings were closed; one buffer = world-1 mappings, more products add none; release closes all
        assert rec["open_after_probe"] == 0 and rec["open_after_first_product"] == world - 1
        assert rec["open_after_more_products"] == world - 1 and rec["open_after_release"] == 0, rec
