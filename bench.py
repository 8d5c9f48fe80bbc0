"""bench.py — the SpMM hot path on MI355X, BASELINE.json's metric.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2]

A step is one pass of the hot path over the whole synthetic matrix:
C = A_csr · B through `custom_mm.naive_spmm` (row-split HIP kernel).  N = 1 runs
BASELINE.json configs[2] (1M×1M CSR at 0.01 % × 1M×256 dense fp32 — the
configuration the metric is quoted on); N > 1 (launched by torch.distributed.run,
one rank per GPU) runs configs[3]: the same matrix row-sharded over the ranks
with an RCCL all-gather of C, strong scaling, `value` = whole-job GFLOP/s
including the gather.  Inputs are generated with the pinned generator of
SURVEY.md §8(d) and are resident in HBM before the timed region.

`--gpus N` with N > 1 works both ways: under a launcher (RANK / WORLD_SIZE set by
torch.distributed.run) this process is one rank; started plainly
(`python bench.py --gpus N`) it is only a parent that — before anything touches the
GPU — starts `python -m torch.distributed.run --nproc-per-node N bench.py …` as a
CHILD process, relays rank 0's JSON line and exits with the child's return code.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (C3:
spmm_wave_row_panel_kernel, two launches per product), from HIP events on the launch
stream inside the timed region; `cpu_baseline` times the oracle (a CPU port of the
reference's algorithm — test infrastructure) on the host cores after the timed region.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
for _p in (str(REPO), str(REPO / "matrix-multiplication_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md §Chip-level parameters
# gather of uniformly random rows from a table that fits the 256 MiB Infinity Cache (same guide,
# §Indexed rows: 38 MB table 8.6 TB/s): the bound of a product whose B is cache-resident (C2)
CACHE_GATHER_PEAK_GBS = 8600.0
L2_PEAK_GBS = 34500.0  # aggregate L2 bandwidth (same guide, § L2 (per XCD)): what the L2-hit share of a gather is priced at

WORKLOADS = {
    # name: (M, K, density, N, description)
    "c3": (1 << 20, 1 << 20, 1e-4, 256, "CSR 1M x 1M at 0.01% nnz x dense 1M x 256 fp32 (BASELINE.json configs[2])"),
    "c2": (1 << 16, 1 << 16, 1e-3, 128, "CSR 64k x 64k at 0.1% nnz x dense 64k x 128 fp32 (BASELINE.json configs[1])"),
}


def algorithmic_bytes(nnz, M, N):
    """per nonzero one B-row gather + col + val; per row rowptr + one C-row write."""
    return nnz * (4 * N + 8) + 4 * (M + 1) + 4 * M * N


def source_fingerprint():
    """sha256 (16 hex digits) of bench.py and of the SpMM kernel sources (csrc/spmm_*.hip and spmm_*.h): what a committed PMC figure is tied to."""
    h = hashlib.sha256()
    for f in [REPO / "bench.py"] + sorted((REPO / "matrix-multiplication_amd" / "csrc").glob("spmm_*.h*")):  # the units and their shared headers
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def committed_traffic(workload, kernel_name=None):
    """HBM-side bytes per product from the committed PMC passes of this same command (profiles/pmc_traffic.json,
    written by tools/collect_profiles.py) as (fabric_bytes, dram_bytes_or_None) — or (None, None) when the record
    was taken with other sources (bench.py or the SpMM kernels changed since: fingerprint) or another kernel."""
    p = REPO / "profiles" / "pmc_traffic.json"
    if not p.exists():
        return None, None
    try:
        rec = json.loads(p.read_text()).get(workload, {})
    except (ValueError, OSError):
        return None, None
    if rec.get("source_fingerprint") != source_fingerprint():
        return None, None
    if kernel_name is not None and kernel_name not in " ".join(rec.get("kernels", {}).get("FETCH_SIZE", {})):
        return None, None
    return rec.get("hbm_bytes_per_product"), rec.get("dram_bytes_per_product")


def profiler_attached():
    """True when this process already runs under a ROCm profiler (tools/profile_bench.sh, the driver's own rocprofv3): no
    second profiler is started from inside it."""
    env = os.environ
    return any(k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCP_")) for k in env) or "rocprofiler" in env.get("LD_PRELOAD", "")


def run_group(cmd, cwd, env, timeout_s):
    """Runs `cmd` in a process group of its own, output discarded; its exit code, or None after `timeout_s` — the whole group
    (the profiler AND the program it started) is then killed: nothing of a pass that hung may stay on the GPU."""
    import signal
    p = subprocess.Popen(cmd, cwd=cwd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
    try:
        return p.wait(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except (ProcessLookupError, PermissionError):
            pass
        p.wait()
        return None


def live_traffic(workload, timeout_s=150):
    """HBM-side bytes per product MEASURED IN THIS RUN: two child processes, one per counter (the guide's HBM / rocprofv3 section:
    separate --pmc passes), each `rocprofv3 --kernel-trace --pmc <counter> -- python3 bench.py --workload W --steps 2 …` on the
    same seeded inputs; per kernel instantiation the mean counter value per launch, summed over the instantiations of one
    product (one launch of each), FETCH_SIZE (KB) doubled — on gfx950 it counts half the bytes of 16-byte-per-lane reads,
    MI355X_MICROARCH.md HBM section — and WRITE_SIZE (KB) as it is.  Returns (bytes, {counter: {kernel: KB}}) or (None, reason):
    any failure (no rocprofv3, a time-out, a refused counter) leaves the bench line to the committed record."""
    import csv
    import glob
    import shutil
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="mi_bench_pmc_", dir="/tmp")
    per = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [rocprof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--",
                   sys.executable, str(REPO / "bench.py"), "--workload", workload, "--steps", "2", "--warmup", "1",
                   "--no-cpu-baseline", "--no-live-pmc"]
            env = dict(os.environ, TMPDIR="/tmp")
            rc = run_group(cmd, cwd="/tmp", env=env, timeout_s=timeout_s)
            if rc is None:
                return None, f"the {counter} pass took longer than {timeout_s} s"
            if rc != 0:
                return None, f"the {counter} pass ended with code {rc}"
            files = glob.glob(os.path.join(out, "**", "*_counter_collection.csv"), recursive=True)
            if not files:
                return None, f"the {counter} pass wrote no counter file"
            by = {}
            for row in csv.DictReader(open(max(files, key=os.path.getmtime))):
                if "spmm" in row["Kernel_Name"] and row.get("Counter_Name", counter) == counter:
                    k = row["Kernel_Name"].split("(int const")[0].replace("void (anonymous namespace)::", "")
                    by.setdefault(k, []).append(float(row["Counter_Value"]))
            if not by:
                return None, f"the {counter} pass saw no SpMM kernel"
            per[counter] = {k: sum(v) / len(v) for k, v in by.items()}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    fetch_kb, write_kb = sum(per["FETCH_SIZE"].values()), sum(per["WRITE_SIZE"].values())
    return (2.0 * fetch_kb + write_kb) * 1024.0, per


def committed_beyond_l2_share(workload):
    """Share of a product's algorithmic bytes that left the L2s in the committed PMC passes (profiles/pmc_traffic.json:
    fabric-side bytes / algorithmic bytes, capped at 1) — a property of the matrix and the plan (which gathers meet in an
    L2), so it is read whatever the record's source fingerprint; None without a record."""
    p = REPO / "profiles" / "pmc_traffic.json"
    try:
        rec = json.loads(p.read_text()).get(workload, {})
        return min(1.0, float(rec["hbm_bytes_per_product"]) / float(rec["algorithmic_bytes_per_product"]))
    except (OSError, ValueError, KeyError, TypeError, ZeroDivisionError):
        return None


def spmm_plan(nnz, M, K, B, C):
    """Which kernel custom_mm.naive_spmm's AUTO dispatch runs for this problem and how many
    launches it issues per product (host-side plan query of the C-ABI, no GPU work)."""
    import custom_mm
    variant, name, launches, _ = custom_mm.spmm_plan(nnz, M, K, B, C)
    assert variant > 0, variant
    return name, launches


def usable_cpus():
    """Host cores this process may actually run on: the affinity mask, capped by a cgroup CPU quota
    (a GPU box hands a 1-GPU job a share of a 256-thread host)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def host_cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            return next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "unknown")
    except OSError:
        return "unknown"


def cpu_baseline(rowptr, col, val, M, K, B, N, gpu_C=None):
    """SURVEY.md §8(d) / BASELINE.md §3: the oracle's OpenMP row-split SpMM (CPU port of reference
    src/naive_sparse_mm.cu:24-101) and the reference's own CPU expression `a_csr @ b` in torch-CPU
    (matmuls.py:41,71,210,234,279,302), both on the WHOLE matrix with every host core this process
    may use, 1 warm-up + best of 3.  This leg is the only place bench.py touches oracle/: it times
    it, and uses its output as the checker for the GPU result."""
    import ctypes
    import oracle
    import torch
    nnz = int(rowptr[-1])
    logical, usable = os.cpu_count() or 1, usable_cpus()
    omp = None
    for name in ("libgomp.so.1", "libomp.so"):
        try:
            omp = ctypes.CDLL(name)
            break
        except OSError:
            pass

    def time_port(threads):
        if omp is not None:
            omp.omp_set_num_threads(threads)
        best, out = float("inf"), None
        for i in range(4):  # 1 warm-up + 3 timed
            t0 = time.perf_counter()
            out = oracle.spmm_csr_omp(rowptr, col, val, M, K, B)
            dt = time.perf_counter() - t0
            if i > 0:
                best = min(best, dt)
        return best, out

    runs = {}
    best, out = time_port(usable)
    runs[usable] = best
    if logical != usable and omp is not None:  # the whole host, should the share be only nominal
        runs[logical], _ = time_port(logical)
    threads = min(runs, key=runs.get)
    best = runs[threads]
    rec = {"value": round(2.0 * nnz * N / best / 1e9, 3), "unit": "GFLOP/s", "cores": threads, "kind": "port",
           "host_cpu": host_cpu_model(), "host_logical_cpus": logical, "usable_cpus": usable,
           "port_seconds_by_threads": {str(t): round(v, 3) for t, v in runs.items()},
           "sample": f"the whole matrix ({M} rows, {nnz} nnz) x the same B, oracle OpenMP row-split SpMM, "
                     f"1 warm-up + best of 3, {best:.2f} s on {threads} threads"}
    torch.set_num_threads(threads)
    a_csr = torch.sparse_csr_tensor(torch.from_numpy(rowptr.astype(np.int64)), torch.from_numpy(col.astype(np.int64)),
                                    torch.from_numpy(val), (M, K))
    b_t = torch.from_numpy(B)
    t_best, ref = float("inf"), None
    for i in range(4):
        t0 = time.perf_counter()
        ref = a_csr @ b_t
        dt = time.perf_counter() - t0
        if i > 0:
            t_best = min(t_best, dt)
    rec["torch_cpu_csr_matmul_gflops"] = round(2.0 * nnz * N / t_best / 1e9, 3)
    rec["torch_cpu_threads"] = threads
    if gpu_C is not None:
        rec["gpu_matches_oracle_on_sample"] = "bit-exact" if np.array_equal(gpu_C, out) else "MISMATCH"
        assert rec["gpu_matches_oracle_on_sample"] == "bit-exact", "GPU result differs from the oracle"
        # the reference's own CPU expression (`a @ b`, matmuls.py:41,...) at its tests' criterion
        # (tests/naive_kernel_test.py:30,36-37: torch.allclose defaults) on the FULL output
        got = torch.from_numpy(gpu_C)
        denom = ref.abs().clamp_min(1e-30)
        rec["max_rel_err_vs_torch_cpu"] = float(((got - ref).abs() / denom).max())
        rec["allclose_vs_torch_cpu_rtol1e-5_atol1e-8"] = bool(torch.allclose(got, ref, rtol=1e-5, atol=1e-8))
        assert rec["allclose_vs_torch_cpu_rtol1e-5_atol1e-8"], \
            f"GPU result differs from torch-CPU A_csr @ B beyond 1e-5 (max rel {rec['max_rel_err_vs_torch_cpu']})"
    return rec


def shared_dir_for(nbytes):
    """A directory every rank of this node can memory-map `nbytes` of input files from: /dev/shm when it has the room (a
    container's default /dev/shm is 64 MB — C3's arrays are 1.9 GB), else $TMPDIR or /tmp, else None (every rank then
    generates the inputs itself).  Free space is asked of the file system, with a tenth of headroom."""
    import shutil
    import tempfile
    for cand in (Path("/dev/shm"), Path(os.environ.get("TMPDIR", tempfile.gettempdir()))):
        try:
            if cand.is_dir() and os.access(cand, os.W_OK) and shutil.disk_usage(cand).free >= 1.1 * nbytes + (16 << 20):
                return cand
        except OSError:
            continue
    return None


def load_inputs(M, K, density, N, rank, world, dist):
    """The pinned synthetic inputs.  One rank: generated in place.  N ranks of one node: rank 0
    generates ONCE and publishes the arrays as files the others memory-map (each then copies only its own row blocks to
    its GPU) — instead of N ranks each spending 2 s and 2 GB of host memory on the same matrix on shared cores.  Where the
    files go is decided by rank 0 from the free space it finds (shared_dir_for) and told to the others; with no room
    anywhere every rank generates its own copy (the generator is deterministic).  Rank 0 removes the files when it exits."""
    import synthetic
    if world == 1:
        rowptr, col, val = synthetic.make_csr(M, K, density, seed=0)
        return rowptr, col, val, synthetic.make_dense(K, N, seed=1)
    import atexit
    import warnings
    where = [None]
    if rank == 0:
        rowptr, col, val = synthetic.make_csr(M, K, density, seed=0)
        B = synthetic.make_dense(K, N, seed=1)
        arrays = {"rowptr": rowptr, "col": col, "val": val, "B": B}
        shm = shared_dir_for(sum(a.nbytes for a in arrays.values()))
        if shm is not None:
            tag = f"mi_bench_{os.environ.get('MASTER_PORT', '0')}_{os.getpid()}_{M}_{K}_{N}"
            names = {k: shm / f"{tag}_{k}.npy" for k in arrays}

            def cleanup():
                for f in names.values():
                    try:
                        f.unlink()
                    except OSError:
                        pass
            atexit.register(cleanup)
            try:
                for k, arr in arrays.items():
                    np.save(names[k], arr)
                where = [{k: str(v) for k, v in names.items()}]
            except OSError:  # the room was gone after all: the others generate their own
                cleanup()
    dist.broadcast_object_list(where, src=0)  # (also the barrier behind rank 0's writes)
    if rank == 0:
        return rowptr, col, val, B
    if where[0] is None:
        rowptr, col, val = synthetic.make_csr(M, K, density, seed=0)
        return rowptr, col, val, synthetic.make_dense(K, N, seed=1)
    warnings.filterwarnings("ignore", message="The given NumPy array is not writable")
    return tuple(np.load(where[0][k], mmap_mode="r") for k in ("rowptr", "col", "val", "B"))


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: start the N ranks as a CHILD
    torch.distributed.run job — this parent never imports torch.cuda or touches HIP (a process that
    has initialised the GPU must not exec or fork GPU work on this pool) — relay rank 0's JSON
    line and return the child's exit code."""
    assert "torch" not in sys.modules, "the launching parent must stay free of torch / HIP"
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()),
           "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--workload", args.workload, "--chunks", str(args.chunks), "--split", args.split,
           "--exchange", args.exchange]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    if os.environ.get("MI_BENCH_TRACE_PARENT") == "1":
        print(f"parent: torch imported = {'torch' in sys.modules}", file=sys.stderr)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith('{"metric"')]
    for ln in proc.stdout.splitlines():
        if ln not in lines:
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    return proc.returncode if proc.returncode != 0 or lines else 1


def bench_c5(args):
    """BASELINE.json configs[4]: BERT-base attention (B 32, H 12, S 512, D 64) through the drop-in
    wrappers, forward + backward: scores = cublasTransbMM.apply(q, k); ctx = cublasMM.apply(probs, v).
    One step = both products forward and backward (6 dense fp32 products, 12.9 GFLOP each)."""
    import torch
    import custom_mm
    import matmuls
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    Bz, H, S, D = 32, 12, 512, 64
    q, k, v = (torch.rand(Bz, H, S, D, device=dev, generator=g).requires_grad_(True) for _ in range(3))
    probs = torch.softmax(torch.rand(Bz, H, S, S, device=dev, generator=g), dim=-1).requires_grad_(True)
    d_scores = torch.rand(Bz, H, S, S, device=dev, generator=g)
    d_ctx = torch.rand(Bz, H, S, D, device=dev, generator=g)
    custom_mm.init_cublas()

    def step():
        for t in (q, k, v, probs):
            t.grad = None
        matmuls.cublasTransbMM.apply(q, k).backward(d_scores)
        matmuls.cublasMM.apply(probs, v).backward(d_ctx)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    flops = 6 * 2.0 * Bz * H * S * S * D
    step_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]
    kern_ms = ev[0].elapsed_time(ev[-1]) / args.steps
    # light parity check against torch autograd of matmul on one head
    qq, kk2 = q.detach()[:1, :1].clone().requires_grad_(True), k.detach()[:1, :1].clone().requires_grad_(True)
    torch.matmul(qq, kk2.transpose(-1, -2)).backward(d_scores[:1, :1])
    assert torch.allclose(qq.grad, q.grad[:1, :1], rtol=1e-5, atol=1e-6)
    print(json.dumps({
        "metric": "BERT-base attention matmuls fwd+bwd (q.kT and probs.v), dense fp32", "value": round(flops * args.steps / elapsed / 1e12, 2),
        "unit": "TFLOP/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BERT-base attention B=32 H=12 S=512 D=64 via cublasTransbMM / cublasMM .apply, fwd+bwd "
                               "(BASELINE.json configs[4])", "flops_per_step": flops},
        "roofline": {"bound": "mfma",
                     "kernel": "per step: gemm_pair_a_at_kernel<8> ×1 (dS·K and dSᵀ·Q in one launch), gemm_f32_pipe_kernel<128,64,false,false> "
                               "×1 (probs·V), gemm_f32_pipe_kernel<128,64,true,false> ×1 (Pᵀ·dC), k = 512; "
                               "gemm_f32_pair_kernel<128,128,false,true,2,4> ×2 (q·kᵀ, dC·Vᵀ), k = 64",
                     "achieved": round(flops / kern_ms / 1e9, 2),
                     "peak": 157.3, "unit": "TFLOP/s", "frac": round(flops / kern_ms / 1e9 / 157.3, 4), "traffic": None,
                     "kernel_ms_per_step": round(kern_ms, 4), "kernel_ms_per_step_median": round(float(np.median(step_ms)), 4),
                     "kernel_ms_per_step_min": round(float(np.min(step_ms)), 4),
                     "note": "fp32-input MFMA peak (MI355X_MICROARCH.md); the q.kT product is also bound by writing "
                             "403 MB of scores"},
    }), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS) + ["c5"], default="c3")
    ap.add_argument("--chunks", type=int, default=0,
                    help="block-cyclic chunks per rank for N > 1 (0 = 4 up to 4 GPUs, 8 beyond: the gather is the "
                         "longer leg at 8 GPUs, so finer chunks expose less of the first chunk's compute)")
    ap.add_argument("--split", choices=["rows", "nnz"], default="rows",
                    help="N > 1: equal-row blocks (one all_gather_into_tensor per step) or nnz-balanced split "
                         "points (one in-place broadcast per owner and step)")
    ap.add_argument("--exchange", choices=["auto", "allgather", "allgather_copy", "alltoall", "p2p", "try-p2p", "push", "try-push"], default="auto",
                    help="N > 1: how a step's blocks reach the other ranks.  allgather: one in-place RCCL all-gather per "
                         "step (gather + copy if the build refuses the in-place form: allgather_copy pins that); alltoall: one "
                         "list-form all_to_all per step (every block straight to every peer: one xGMI link each, still a "
                         "collective); auto (default): both are timed before the timed region and the faster one runs; p2p: "
                         "independent direct sends to every peer; try-p2p: as auto, plus p2p if a probe of direct sends in "
                         "its own process group (short timeout) succeeds on every rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="N = 1: do not measure roofline.traffic in this run (two rocprofv3 --pmc child passes behind the timed "
                         "region, ~20 s each); the committed record of the same sources is quoted instead, or null")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        if args.workload == "c5":
            raise SystemExit("--workload c5 is a single-GPU measurement")
        sys.exit(self_launch(args))

    if args.workload == "c5":
        if args.gpus != 1:
            raise SystemExit("--workload c5 is a single-GPU measurement")
        return bench_c5(args)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # MI_BENCH_REHEARSE=1: run the N > 1 path with every rank on GPU 0 over gloo — a functional rehearsal of
    # the sharded driver on a one-GPU box (RCCL refuses two ranks on one device); never a measurement.
    rehearse = os.environ.get("MI_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            # "nccl" is RCCL on ROCm.  Its kernels go on a HIGH-PRIORITY stream: the all-gather of block j runs
            # beside the SpMM of block j+1, which fills every CU — at normal priority the collective's
            # workgroups would queue behind it and the overlap the layout is built for would be lost
            # (NCCL_DEBUG is left as the environment has it: RCCL's own VERSION line would go to the ranks' stdout,
            # beside the one JSON line the driver reads; the RCCL version is in config.collective_backend instead)
            opts = dist.ProcessGroupNCCL.Options()
            opts.is_high_priority_stream = True
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, pg_options=opts)
    rank_devices = None
    if world > 1:
        mine = {"rank": rank, "local_rank": local_rank, "device": torch.cuda.current_device(),
                "name": torch.cuda.get_device_name(dev), "pid": os.getpid()}
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, mine)

    import custom_mm
    import sharded
    import synthetic

    auto_chunks = args.chunks <= 0
    if auto_chunks:
        args.chunks = 4 if world <= 4 else 8
    chunk_trials, trial_fallbacks, p2p_probe = None, [], None
    M, K, density, N, desc = WORKLOADS[args.workload]
    t0 = time.perf_counter()
    rowptr, col, val, B_host = load_inputs(M, K, density, N, rank, world, dist)
    nnz = int(rowptr[-1])
    gen_s = time.perf_counter() - t0
    flops = 2.0 * nnz * N
    bytes_alg = algorithmic_bytes(nnz, M, N)

    B = torch.from_numpy(B_host).to(dev)
    custom_mm.init_cusparse()
    if world == 1:
        d_rp, d_col, d_val = (torch.from_numpy(x).to(dev) for x in (rowptr, col, val))
        C = torch.empty(M, N, device=dev)

        def step():
            custom_mm.naive_spmm(d_val, d_col, d_rp, nnz, M, K, B, C)
        kernel_name, launches_per_step = spmm_plan(nnz, M, K, B, C)
        local_bytes_alg = bytes_alg
    else:
        # col / val go to the device ONCE (840 MB at C3: nothing on 288 GB): every ShardedSpMM below — up to six trial
        # operators and the measured one — then cuts its blocks with device-side slices instead of uploading its share
        # again (round-4 review); rowptr stays on the host, where the constructor computes the block boundaries
        rp_t = torch.from_numpy(np.asarray(rowptr))
        col_t, val_t = torch.from_numpy(col).to(dev), torch.from_numpy(val).to(dev)
        # Which exchange?  The in-place all-gather (RCCL picks rings / trees) and the list-form all_to_all (every block
        # straight to every peer: one xGMI link each) are both COLLECTIVES: every rank calls them alike, a build that
        # refuses one does so at an argument check on every rank, which ShardedSpMM probes at construction and settles
        # by agreement — so both are part of the default trial.  Independent direct sends (p2p) can fail on some ranks
        # only; they join the trial only on request AND after sharded.probe_p2p (own process group, short timeout,
        # outcome agreed over the main group) succeeded everywhere.  A failed probe costs nothing but the option.
        # The IPC push exchange (peers' C mapped through hipIpc handles with explicit lifetimes, blocks copied straight into
        # them on a side stream between an entry and an exit fence — no RCCL data movement) is probed and agreed at
        # construction like the others, but joins the trial only on request (--exchange try-push / push): it has run on ONE
        # GPU only (two and four processes), never across xGMI, and the driver's 8-GPU run is not the place for a first.
        exch_cands = {"auto": ("allgather", "alltoall"), "try-p2p": ("allgather", "alltoall"),
                      "try-push": ("allgather", "alltoall", "push")}.get(args.exchange, (args.exchange,))
        if args.exchange == "try-p2p":
            p2p_probe = sharded.probe_p2p(dev, timeout_s=20.0)
            if p2p_probe:
                exch_cands += ("p2p",)
        if auto_chunks or len(exch_cands) > 1:
            # Before anything is timed: how finely to cut a rank's rows is a trade between starting the first
            # exchange early (many chunks) and RCCL's efficiency on larger messages (few chunks); try the candidates
            # for a few steps each and keep the fastest (max over ranks, so every rank decides alike).  Setup, like the
            # warm-up: not in the timed region.
            chunk_trials = {}
            chunk_cands = ((2, 4) if world <= 2 else (2, 4, 8)) if auto_chunks else (args.chunks,)

            for exch in exch_cands:
                for cand in chunk_cands:
                    trial = sharded.ShardedSpMM(rp_t, col_t, val_t, M, K, dev, chunks=cand, split=args.split, exchange=exch)
                    trial_fallbacks.extend(x for x in trial.fallbacks if x not in trial_fallbacks)
                    if (trial.exchange, cand) in chunk_trials:
                        continue  # a refused form fell back to one that is tried under its own name
                    Ct = trial.alloc_output(N)
                    for _ in range(2):
                        trial.forward(B, out=Ct)
                    dist.barrier()
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(3):
                        trial.forward(B, out=Ct)
                    torch.cuda.synchronize()
                    tt = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    chunk_trials[(trial.exchange, cand)] = float(tt[0]) / 3 * 1e3
                    trial.release_peers()  # (collective; a no-op unless the trial pushed through IPC mappings)
                    del trial, Ct
            args.exchange, args.chunks = min(chunk_trials, key=chunk_trials.get)
            torch.cuda.empty_cache()
        else:
            args.exchange = exch_cands[0]
        op = sharded.ShardedSpMM(rp_t, col_t, val_t, M, K, dev, chunks=args.chunks, split=args.split, exchange=args.exchange)
        C = op.alloc_output(N)

        def step():
            op.forward(B, out=C)
        kernel_name, per_block = spmm_plan(op.blocks[0][4], op.blocks[0][5], K, B, C[:max(op.blocks[0][5], 1)])
        launches_per_step = op.chunks * per_block
        local_bytes_alg = algorithmic_bytes(op.local_nnz, op.local_rows, N)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        step()
        ev[i + 1].record()  # same (current) stream the kernels are launched on
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax)
    compute_only_ms = gather_only_ms = None
    if world > 1:
        # the same steps without the all-gather: what the row shards alone cost (reported beside the
        # end-to-end figure; `value` stays end-to-end)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            op.forward(B, out=C, gather=False)
        barrier()
        tc = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
        dist.all_reduce(tc, op=dist.ReduceOp.MAX)
        compute_only_ms = float(tc) / args.steps * 1e3
        op.forward(B, out=C)  # leave the gathered result in C
        barrier()
        # the exchanges alone (what is in C travels again): with compute-only and end-to-end this says whether the
        # gather or the kernels bound the step, and the implied per-rank receive rate says which algorithm RCCL ran
        # (one ring: per-link bound; multi-link / direct: several links' worth)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            op.forward(B, out=C, compute=False)
        barrier()
        tg = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        gather_only_ms = float(tg) / args.steps * 1e3
        # placement self-check of the all-gather (no oracle involved): recompute a block that ANOTHER rank
        # owns with the same kernel and compare it bit-for-bit with what arrived in its slot of C
        peer = (rank + 1) % world
        r0 = min(int(op.bounds[peer]), M)
        r1 = min(int(op.bounds[peer + 1]), M)
        if r1 > r0:
            rp_t = torch.from_numpy(np.asarray(rowptr))
            p0, p1 = int(rowptr[r0]), int(rowptr[r1])
            chk = torch.empty(r1 - r0, N, device=dev)
            custom_mm.naive_spmm(torch.from_numpy(np.array(val[p0:p1])).to(dev), torch.from_numpy(np.array(col[p0:p1])).to(dev),
                                 (rp_t[r0:r1 + 1] - rp_t[r0]).to(torch.int32).to(dev), p1 - p0, r1 - r0, K, B, chk)
            assert torch.equal(chk, C[r0:r1]), f"rank {rank}: gathered block of rank {peer} differs from a local recompute"
    single_ms = None
    if world > 1:
        # The whole matrix on rank 0's GPU alone (after every timed leg; the other ranks wait at the barrier): the
        # N-GPU line then carries its own single-GPU time, speed-up and the exchange rate a 6x speed-up needs —
        # self-judging, whatever box the driver's separate N = 1 run landed on.
        if rank == 0:
            s_rp, s_col, s_val = (torch.from_numpy(np.asarray(x)).to(dev) for x in (rowptr, col, val))
            s_C = torch.empty(M, N, device=dev)
            for _ in range(2):
                custom_mm.naive_spmm(s_val, s_col, s_rp, nnz, M, K, B, s_C)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(max(3, min(args.steps, 10))):
                custom_mm.naive_spmm(s_val, s_col, s_rp, nnz, M, K, B, s_C)
            torch.cuda.synchronize()
            single_ms = (time.perf_counter() - t1) / max(3, min(args.steps, 10)) * 1e3
            assert torch.equal(s_C, C[:M]), "the gathered C differs from rank 0's single-GPU product"
            del s_rp, s_col, s_val, s_C
        barrier()
    step_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]
    # the step's launches run back to back on this stream: their durations sum to the step's event time
    kernels_ms_per_step = float(np.mean(step_ms))

    if rank == 0:
        if compute_only_ms is not None:  # N > 1: the event interval also spans the gather wait; use the compute-only loop
            kernels_ms_per_step = compute_only_ms
        achieved = local_bytes_alg / (kernels_ms_per_step * 1e-3) / 1e9
        traffic, traffic_dram = committed_traffic(args.workload, kernel_name) if world == 1 else (None, None)
        traffic_source = None if traffic is None else "profiles/pmc_traffic.json (committed passes of this command at these very sources)"
        live_detail = None
        if world == 1 and not args.no_live_pmc and os.environ.get("MI_BENCH_LIVE_PMC", "1") != "0" and not profiler_attached():
            try:
                live, live_detail = live_traffic(args.workload)
            except Exception as e:  # noqa: BLE001  (nothing here may cost the bench line)
                live, live_detail = None, f"{type(e).__name__}: {e}"
            if live is not None and any(kernel_name in k for k in live_detail["FETCH_SIZE"]):
                traffic = live
                traffic_source = "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this run, behind the timed region"
            elif live is not None:
                live_detail = f"the passes did not see {kernel_name}"
        cache_resident = 4 * K * N <= (256 << 20)
        peak = CACHE_GATHER_PEAK_GBS if cache_resident else HBM_PEAK_GBS
        beyond_l2 = committed_beyond_l2_share(args.workload) if (cache_resident and world == 1) else None
        if beyond_l2 is not None and beyond_l2 < 1.0:
            # a cache-resident B: part of the gathered bytes never leave the L2s (the counters say how many), and those
            # are not bound by the Infinity-Cache gather rate: the bound on the ALGORITHMIC bytes is the harmonic mix
            # of the two rates (a fraction above 1 would only say that the denominator ignored the L2 hits)
            peak = 1.0 / (beyond_l2 / CACHE_GATHER_PEAK_GBS + (1.0 - beyond_l2) / L2_PEAK_GBS)
        rec = {
            "metric": "SpMM GFLOP/s, CSR(1M,0.01%) x dense(256)" if args.workload == "c3"
                      else "SpMM GFLOP/s, CSR(64k,0.1%) x dense(128)",
            "value": round(flops * args.steps / elapsed / 1e9, 2),
            "unit": "GFLOP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if not rehearse else "synthetic (REHEARSAL: all ranks on one GPU over gloo, not a measurement)",
            "config": {
                "workload": desc, "M": M, "K": K, "N": N, "nnz": nnz,
                "generator": "numpy PCG64 seedA=0 seedB=1 (SURVEY.md 8d)",
                "sha256_rowptr_col_val": hashlib.sha256(rowptr.tobytes() + col.tobytes() + val.tobytes()).hexdigest()[:16],
                "parallelism": "single GPU" if world == 1 else
                               f"A row-sharded over {world} GPUs ({args.split}-balanced blocks), block-cyclic "
                               f"x{args.chunks}, " + {"allgather": "RCCL all-gather of C, in place",
                                                      "allgather_copy": "RCCL all-gather of C into a scratch span + copy",
                                                      "alltoall": "C exchanged by one list-form RCCL all_to_all per step (every block "
                                                                  "straight to every peer, in place)",
                                                      "p2p": "C exchanged by direct RCCL sends to every peer",
                                                      "push": "C blocks copied straight into every peer's C through hipIpc mappings "
                                                              "(side-stream device copies between an entry and an exit fence: two tiny all-reduces)"}[op.exchange if world > 1 else "allgather"],
                "rccl_ranks": world if world > 1 else None,
                "chunks": None if world == 1 else args.chunks,
                "exchange": None if world == 1 else op.exchange,
                "exchange_fallbacks": None if world == 1 else (op.fallbacks + [x for x in trial_fallbacks if x not in op.fallbacks]),
                "rank_devices": rank_devices,
                "nccl_debug_env": os.environ.get("NCCL_DEBUG"),
                "chunk_trials_ms_per_step": None if not chunk_trials else {f"{e}/{c}": round(v, 4) for (e, c), v in chunk_trials.items()},
                "collective_backend": None if world == 1 else ("gloo (rehearsal)" if rehearse else
                                                                "rccl " + ".".join(str(x) for x in torch.cuda.nccl.version())),
                "flops_per_step": flops, "algorithmic_bytes_per_step": bytes_alg,
                "effective_GBps_whole_job": round(bytes_alg * args.steps / elapsed / 1e9, 1),
                "input_generation_s": round(gen_s, 1),
                "compute_only_ms_per_step": None if compute_only_ms is None else round(compute_only_ms, 4),
                "gather_only_ms_per_step": None if gather_only_ms is None else round(gather_only_ms, 4),
                # bytes every rank RECEIVES per step (the other ranks' rows of C) / the gather-only time
                "gather_receive_GBps_per_rank": None if gather_only_ms is None else
                                                round(4.0 * N * (op.padded_rows - op.local_rows) / (gather_only_ms * 1e-3) / 1e9, 1),
                # self-judging fields for BASELINE's ">= 6x at 8 GPUs": rank 0's single-GPU time of the whole product
                # (measured in this run, same box), the speed-up of this line over it, and the receive rate per rank the
                # exchange needs for 6x even with perfect overlap (bytes a rank receives per step / (single / 6))
                "single_gpu_ms_on_rank0": None if single_ms is None else round(single_ms, 4),
                "speedup_vs_single_gpu_on_rank0": None if single_ms is None else round(single_ms / (elapsed / args.steps * 1e3), 3),
                "gathered_C_equals_single_gpu_product": None if single_ms is None else True,
                "gather_GBps_needed_for_6x": None if single_ms is None else
                                             round(4.0 * N * (op.padded_rows - op.local_rows) / (single_ms / 6.0 * 1e-3) / 1e9, 1),
                "gather_note": None if gather_only_ms is None else
                               "xGMI: 7 links x ~153 GB/s per GPU; a single ring is bound by ONE link, direct / multi-link "
                               "exchanges by several: compare gather_receive_GBps_per_rank with those",
                "p2p_probe_ok": p2p_probe,
            },
            "roofline": {
                # C2's B (32 MiB) sits in the Infinity Cache: its rate is a cache-gather rate and is priced
                # against the guide's cache-resident gather figure, never against HBM
                "bound": "cache" if cache_resident else "hbm", "kernel": kernel_name,
                "launches_per_step": launches_per_step,
                "achieved": round(achieved, 1), "peak": round(peak, 1), "unit": "GB/s",
                "beyond_l2_share": None if beyond_l2 is None else round(beyond_l2, 4),
                "frac": round(achieved / peak, 4),
                "bound_detail": ("B fits the 256 MiB Infinity Cache: gathers are cache hits; peak = random-row gather "
                                 "from a cache-resident table (MI355X_MICROARCH.md, Indexed rows: 8.6 TB/s) for the share "
                                 "of the algorithmic bytes that leaves the L2s (beyond_l2_share, from the committed PMC "
                                 "passes) and the aggregate L2 rate (34.5 TB/s) for the share that hits in an L2")
                                if cache_resident else
                                ("fabric-side gather rate: ~25-50 % of the B-row gathers hit the Infinity Cache "
                                 "(FETCH_SIZE counts those hits), priced against the 8 TB/s HBM spec peak; the guide's "
                                 "pure-HBM random-row gather ceiling is 5.5-5.8 TB/s, streaming 6.29 TB/s"),
                "traffic": traffic, "traffic_dram": traffic_dram,
                "traffic_source": traffic_source,
                "traffic_live_detail": live_detail,
                "frac_dram_only": None if not traffic_dram else round(traffic_dram / (kernels_ms_per_step * 1e-3) / 1e9 / peak, 4),
                "kernel_ms_per_step": round(kernels_ms_per_step, 4),
                "kernel_ms_per_step_median": round(float(np.median(step_ms)), 4),
                "kernel_ms_per_step_min": round(float(np.min(step_ms)), 4),
                "avg_launch_ms": round(kernels_ms_per_step / launches_per_step, 4),
                "algorithmic_bytes_per_step": local_bytes_alg,
                "note": "achieved = algorithmic bytes of one product / summed duration of its "
                        f"{launches_per_step} back-to-back launch(es) (HIP events on the launch stream); "
                        "traffic = fabric-side PMC bytes per product (Infinity-Cache hits included; 2 x FETCH_SIZE + "
                        "WRITE_SIZE, KB), measured in this run by two rocprofv3 --pmc child passes where traffic_source "
                        "says live, else from profiles/pmc_traffic.json — null unless that record was taken with these "
                        "very sources (fingerprint of bench.py + csrc/spmm_* sources) and this kernel; traffic_dram = the "
                        "DRAM-side share (not observable from rocprofv3 on this part: null); frac_dram_only = traffic_dram / time / peak",
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(rowptr, col, val, M, K, B_host, N, C.cpu().numpy())
        print(json.dumps(rec), flush=True)
    if world > 1:
        op.release_peers()  # (collective; peers' buffers mapped by the push exchange are let go one rank at a time)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
