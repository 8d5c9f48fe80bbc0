"""The bench.py output contract, checked on the committed recordings of real runs (profiles/):
every key the driver and the judge read is there with the right type, and the derived figures
agree with each other.  (bench.py itself needs the MI355X; this guards the record format.)"""
import json
import os
import sys
from pathlib import Path

import pytest

PROFILES = Path(__file__).resolve().parent.parent / "profiles"
REQUIRED = {"metric": str, "value": (int, float), "unit": str, "n_gpus": int, "steps": int, "warmup": int,
            "ms_per_step": (int, float), "higher_is_better": bool, "scaling": str, "dtype": str, "data": str,
            "config": dict, "roofline": dict}


@pytest.mark.parametrize("name", ["r01_bench_c3.json", "r01_bench_c2.json", "r01_bench_c5.json",
                                  "r02_bench_c3.json", "r02_bench_c2.json", "r02_bench_c5.json",
                                  "r03_bench_c3.json", "r03_bench_c2.json", "r03_bench_c5.json",
                                  "r06_bench_c3.json", "r06_bench_c2.json"])
def test_recorded_bench_lines_follow_the_contract(name):
    rec = json.loads((PROFILES / name).read_text())
    for key, typ in REQUIRED.items():
        assert key in rec and isinstance(rec[key], typ), (name, key)
    assert "vs_baseline" in rec and rec["vs_baseline"] is None  # BASELINE.md publishes no number for this metric
    assert rec["n_gpus"] == 1 and rec["higher_is_better"] is True and rec["dtype"] == "f32"
    assert "workload" in rec["config"] and "model" not in rec["config"]
    roof = rec["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, (name, key)
    assert roof["bound"] in ("hbm", "mfma", "cache") and roof["unit"] in ("GB/s", "TFLOP/s")
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    if name[:3] in ("r03", "r06"):
        # median and min per step beside the mean
        assert roof["kernel_ms_per_step_min"] <= roof["kernel_ms_per_step_median"] and roof["kernel_ms_per_step"] > 0
    if name[:3] in ("r02", "r03", "r06"):
        # a cache-resident B (C2) is priced against the cache-gather figure, never as an HBM fraction > 1
        # (that figure is the guide's MEASURED gather rate from a cache-resident table, 8.6 TB/s — a reference, not a
        # hardware limit: C2 passed it by 2 % once its kernel stopped paying for ds_bpermute in round 3; the HBM and
        # MFMA peaks are limits)
        assert roof["frac"] <= (1.05 if roof["bound"] == "cache" else 1.0)
        assert (roof["bound"] == "cache") == ("64k" in rec["config"]["workload"])
    if roof["bound"] in ("hbm", "cache"):
        if roof["bound"] == "hbm":
            assert roof["peak"] == 8000.0
        elif roof.get("beyond_l2_share") is None:
            assert roof["peak"] == 8600.0
        else:
            # round 4 on: the L2-aware bound — the harmonic mix of the cache-gather figure (8.6 TB/s) for the share of the bytes that
            # leaves the L2s and the aggregate L2 rate (34.5 TB/s) for the share that does not
            share = roof["beyond_l2_share"]
            assert abs(roof["peak"] - 1.0 / (share / 8600.0 + (1.0 - share) / 34500.0)) < 1.0
        cpu = rec["cpu_baseline"]
        if name[:3] in ("r02", "r03", "r06"):
            # every core the process may use, the whole matrix, 1 warm-up + best of 3
            assert cpu["cores"] == cpu["usable_cpus"] <= cpu["host_logical_cpus"] and "whole matrix" in cpu["sample"]
            assert "torch_cpu_csr_matmul_gflops" in cpu and "host_cpu" in cpu
        for key in ("value", "unit", "cores", "kind", "sample"):
            assert key in cpu, (name, key)
        assert cpu["kind"] in ("port", "reference") and cpu["cores"] >= 1
        assert cpu.get("gpu_matches_oracle_on_sample") == "bit-exact"
        # value = flops / time; achieved = algorithmic bytes / kernel time
        flops, ms = rec["config"]["flops_per_step"], rec["ms_per_step"]
        assert abs(rec["value"] - flops / (ms * 1e-3) / 1e9) / rec["value"] < 0.01
        alg = rec["config"]["algorithmic_bytes_per_step"]
        assert abs(roof["achieved"] - alg / (roof["kernel_ms_per_step"] * 1e-3) / 1e9) / roof["achieved"] < 0.01


@pytest.mark.parametrize("rnd", ["r01", "r02", "r03", "r04", "r05", "r06"])
def test_c3_kernel_stats_agree_with_the_bench_line(rnd):
    """rocprofv3's average launch durations (same command) add up to bench.py's HIP-event time per product."""
    import csv
    rec = json.loads((PROFILES / f"{rnd}_bench_c3.json").read_text())
    rows = list(csv.DictReader(open(PROFILES / f"{rnd}_bench_c3_kernel_stats.csv")))
    main = [r for r in rows if "spmm_wave_row_panel_kernel" in r["Name"]]
    assert len(main) == rec["roofline"]["launches_per_step"] == 2
    total_ms = sum(float(r["AverageNs"]) for r in main) / 1e6
    assert abs(total_ms - rec["roofline"]["kernel_ms_per_step"]) / total_ms < 0.03
    traffic = json.loads((PROFILES / "pmc_traffic.json").read_text())["c3"]
    assert traffic["round"] in ("r03", "r04", "r05", "r06")
    if rnd == traffic["round"]:
        # the bench line quotes the committed PMC record only while it was taken with the very sources that print it
        assert len(traffic["source_fingerprint"]) == 16
        if rec["roofline"]["traffic"] is None:
            pytest.skip("the committed bench line was printed by sources newer than the PMC record: re-collect both")
        # (a line that measured its traffic itself — two rocprofv3 --pmc child passes of that run — agrees with the committed
        # passes to within a run-to-run per cent; a line that quotes the record carries it exactly)
        live = str(rec["roofline"].get("traffic_source") or "").startswith("live")
        assert abs(traffic["hbm_bytes_per_product"] - rec["roofline"]["traffic"]) < (2e-2 if live else 1e-3) * rec["roofline"]["traffic"]
        assert 1.0 <= traffic["hbm_bytes_per_product"] / rec["config"]["algorithmic_bytes_per_step"] < 1.05
        assert rec["roofline"]["traffic_dram"] is None and "TARGET" in traffic["dram_note"]
        # two dispatches per product at C2: the main kernel and the (empty) follow-up — no memset, no scan of rowptr
        c2 = [r["Name"] for r in csv.DictReader(open(PROFILES / f"{rnd}_bench_c2_kernel_stats.csv"))]
        spmm = [n for n in c2 if "spmm_" in n or "find_long" in n or "combine_long" in n]
        assert len(spmm) == 2 and any("spmm_group_kernel" in n for n in spmm) and any("spmm_long_rows_kernel" in n or "spmm_staged_rows_kernel" in n for n in spmm)


def test_committed_traffic_is_tied_to_the_sources(tmp_path, monkeypatch):
    """bench.committed_traffic returns the committed PMC bytes only for the fingerprint (bench.py + csrc/spmm_*.hip) and
    the kernel they were taken with; otherwise (None, None) — never a stale figure beside a fresh `achieved`."""
    bench = _load_bench()
    fp = bench.source_fingerprint()
    assert len(fp) == 16 and fp == bench.source_fingerprint()
    rec = json.loads((PROFILES / "pmc_traffic.json").read_text())["c3"]
    got = bench.committed_traffic("c3", "spmm_wave_row_panel_kernel")
    if rec.get("source_fingerprint") == fp:
        assert got[0] == rec["hbm_bytes_per_product"]
        assert bench.committed_traffic("c3", "some_other_kernel") == (None, None)
    else:
        assert got == (None, None)
    assert bench.committed_traffic("no-such-workload") == (None, None)


def test_live_traffic_sums_one_launch_of_each_instantiation_and_doubles_fetch(monkeypatch):
    """bench.live_traffic: two child passes (FETCH_SIZE, WRITE_SIZE), per kernel instantiation the mean per launch, one launch
    of each per product, 2 x FETCH + WRITE in KB (the guide's gfx950 correction) — shown on counter files written by a stand-in
    for the child process; a failing pass leaves (None, reason) and never raises."""
    bench = _load_bench()
    if not os.path.exists("/opt/rocm/bin/rocprofv3"):
        pytest.skip("no rocprofv3 in this image")
    calls = []

    def fake_run(cmd, cwd, env, timeout_s):
        counter = cmd[cmd.index("--pmc") + 1]
        out = Path(cmd[cmd.index("-d") + 1]) / "host" / "1"
        out.mkdir(parents=True)
        assert cmd[cmd.index("--") + 1] == sys.executable and "--no-live-pmc" in cmd and "--no-cpu-baseline" in cmd and cwd == "/tmp"
        k1 = "void (anonymous namespace)::spmm_wave_row_panel_kernel<false, 1, 8, false>(int const*, int const*)"
        k2 = "void (anonymous namespace)::spmm_wave_row_panel_kernel<true, 1, 8, false>(int const*, int const*)"
        k3 = "void (anonymous namespace)::spmm_staged_rows_kernel<8>(int const*, int const*)"
        rows = [(k1, 100.0), (k1, 102.0), (k2, 50.0), (k2, 50.0), (k3, 1.0), ("at::native::fill", 999.0)] if counter == "FETCH_SIZE" \
            else [(k1, 10.0), (k2, 10.0), (k3, 0.0)]
        with open(out / "1_counter_collection.csv", "w") as f:
            f.write('"Kernel_Name","Counter_Name","Counter_Value"\n')
            for k, v in rows:
                f.write(f'"{k}","{counter}",{v}\n')
        calls.append(counter)
        return 0

    monkeypatch.setattr(bench, "run_group", fake_run)
    got, detail = bench.live_traffic("c3")
    assert calls == ["FETCH_SIZE", "WRITE_SIZE"]
    assert got == (2.0 * (101.0 + 50.0 + 1.0) + 20.0) * 1024.0
    assert set(detail) == {"FETCH_SIZE", "WRITE_SIZE"} and len(detail["FETCH_SIZE"]) == 3
    monkeypatch.setattr(bench, "run_group", lambda cmd, cwd, env, timeout_s: 3)
    assert bench.live_traffic("c3") == (None, "the FETCH_SIZE pass ended with code 3")
    monkeypatch.setattr(bench, "run_group", lambda cmd, cwd, env, timeout_s: None)
    assert bench.live_traffic("c3", timeout_s=5) == (None, "the FETCH_SIZE pass took longer than 5 s")
    monkeypatch.undo()
    # the real runner: exit codes pass through; a program that outlives the limit is killed with its whole process group
    assert bench.run_group([sys.executable, "-c", "import sys; sys.exit(7)"], cwd="/tmp", env=dict(os.environ), timeout_s=60) == 7
    t0 = __import__("time").time()
    assert bench.run_group([sys.executable, "-c", "import subprocess, sys, time; subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(60)']); time.sleep(60)"],
                           cwd="/tmp", env=dict(os.environ), timeout_s=2) is None
    assert __import__("time").time() - t0 < 30
    monkeypatch.setenv("ROCP_TOOL_LIBRARIES", "x")
    assert bench.profiler_attached()


# --- `python bench.py --gpus N` started plainly: the parent spawns the ranks itself ------------------

def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", PROFILES.parent / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_self_launch_builds_a_torchrun_child_and_relays_rank0(monkeypatch, capsys):
    """The parent path: no torch import, a child `python -m torch.distributed.run --nproc-per-node N
    bench.py …` on 127.0.0.1, rank 0's JSON line relayed, the child's return code returned."""
    import argparse
    import subprocess
    import sys
    bench = _load_bench()
    seen = {}

    def fake_run(cmd, env=None, stdout=None, text=None):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 0, stdout='noise\n{"metric": "x", "n_gpus": 4}\n')

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.delitem(sys.modules, "torch", raising=False)
    args = argparse.Namespace(gpus=4, steps=7, warmup=2, workload="c3", chunks=0, split="rows", exchange="auto",
                              no_cpu_baseline=True)
    assert bench.self_launch(args) == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    tail = cmd[cmd.index(str(PROFILES.parent / "bench.py")) + 1:]
    assert tail[:6] == ["--gpus", "4", "--steps", "7", "--warmup", "2"] and "--no-cpu-baseline" in tail
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert capsys.readouterr().out.strip() == '{"metric": "x", "n_gpus": 4}'

    def failing_run(cmd, env=None, stdout=None, text=None):
        return subprocess.CompletedProcess(cmd, 3, stdout="")
    monkeypatch.setattr(bench.subprocess, "run", failing_run)
    assert bench.self_launch(args) == 3


def test_plain_multi_gpu_invocation_spawns_ranks_without_touching_hip():
    """End to end in this GPU-less container: `python bench.py --gpus 2` must get as far as its child
    ranks, which refuse loudly because there is no MI355X here (no CPU fallback), and hand their
    failure back as a non-zero exit code; the parent itself never imports torch."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, MI_BENCH_TRACE_PARENT="1")
    proc = subprocess.run([sys.executable, str(PROFILES.parent / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                           "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env)
    assert proc.returncode != 0
    # (torch.distributed.run stops the other ranks as soon as one has failed: a sibling that was still importing torch
    # may never get to print its own refusal — one is the guaranteed count, two the usual one)
    assert proc.stderr.count("bench.py needs an MI355X") >= 1, proc.stderr[-2000:]
    assert "parent: torch imported = False" in proc.stderr
    assert not [ln for ln in proc.stdout.splitlines() if ln.startswith('{"metric"')]


# --- N ranks share rank 0's generated inputs through files: where they go depends on the free space found -------------

def test_shared_dir_needs_room_for_the_inputs(monkeypatch, tmp_path):
    """A container's default /dev/shm is 64 MB and C3's arrays are 1.9 GB: the directory is chosen by free space (with
    headroom), $TMPDIR is the second choice, and None — every rank generates its own inputs — the last."""
    import collections
    import shutil
    bench = _load_bench()
    usage = collections.namedtuple("usage", "total used free")
    free = {"/dev/shm": 64 << 20, str(tmp_path): 10 << 30}
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    monkeypatch.setattr(shutil, "disk_usage", lambda p: usage(0, 0, free[str(p)]))
    assert bench.shared_dir_for(8 << 20) == Path("/dev/shm")       # small inputs fit a 64 MB /dev/shm
    assert bench.shared_dir_for(1900 << 20) == tmp_path            # C3's do not: $TMPDIR
    free[str(tmp_path)] = 1 << 30
    assert bench.shared_dir_for(1900 << 20) is None                # no room anywhere


def _load_inputs_worker(rank, world, port, out_dir, no_room):
    import os
    import sys
    import numpy as np
    import torch.distributed as dist
    repo = PROFILES.parent
    for p in (str(repo), str(repo / "matrix-multiplication_amd")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bench = _load_bench()
        if no_room:
            bench.shared_dir_for = lambda nbytes: None
        rowptr, col, val, B = bench.load_inputs(300, 200, 0.05, 8, rank, world, dist)
        kind = "generated" if isinstance(B, np.ndarray) and not isinstance(B, np.memmap) else "mapped"
        np.savez(os.path.join(out_dir, f"in_{rank}.npz"), rowptr=rowptr, col=col, val=val, B=B, kind=kind)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("no_room", [False, True])
def test_ranks_get_rank0s_inputs_with_or_without_room_for_the_files(tmp_path, no_room):
    import socket
    import sys
    import numpy as np
    import torch.multiprocessing as mp
    sys.path.insert(0, str(PROFILES.parent / "matrix-multiplication_amd"))
    import synthetic
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_load_inputs_worker, args=(2, port, str(tmp_path), no_room), nprocs=2, join=True)
    rowptr, col, val = synthetic.make_csr(300, 200, 0.05, seed=0)
    B = synthetic.make_dense(200, 8, seed=1)
    for r in range(2):
        got = np.load(tmp_path / f"in_{r}.npz")
        assert all(np.array_equal(got[k], w) for k, w in (("rowptr", rowptr), ("col", col), ("val", val), ("B", B)))
        assert str(got["kind"]) == ("generated" if (r == 0 or no_room) else "mapped")
