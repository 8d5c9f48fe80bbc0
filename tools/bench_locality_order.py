"""Locality the inspector creates: a banded / community-structured matrix whose ROWS ARRIVE SHUFFLED gathers like a uniformly
random one; the row schedule (custom_mm.spmm_schedule with the columns) puts the rows back in the order of their median
column and the waves that run together gather from one neighbourhood of B again — the same bits.

    python tools/bench_locality_order.py [--only TAG] [--log FILE]
    python tools/bench_locality_order.py --one TAG --mode plain|scheduled     (a target for rocprofv3 --pmc FETCH_SIZE)

Cases (seeded, no downloads):  band32k / band1k — columns within ± 32 K / ± 1 K of the row's position, rows shuffled;
community — 256 communities, 90 % of a row's entries inside its own, rows shuffled; at 1 M x 1 M x N 256 (config C3's size, B
beyond the Infinity Cache) and at 128 K x 128 K x N 256 (B = 128 MiB: inside it, beyond the L2s).  Per case: the unshuffled
matrix (what the structure is worth), the shuffled one plain, the shuffled one scheduled; the schedule's measured window
spans; sampled rows bit-exact against the oracle.  Reference: the inspector that compacts each block's footprint of B,
src/sparse_mm.cu:62-68,259.
"""
import argparse
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "matrix-multiplication_amd"))
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import custom_mm  # noqa: E402
from bench_hbm_regime import check_rows, time_interleaved  # noqa: E402

dev = torch.device("cuda")


def structured_csr(M, K, per_row, kind, seed, shuffle):
    """(rowptr, col, val) on the device; `shuffle`: the rows in a random order (the pattern of row r moves to row perm[r])."""
    g = torch.Generator(device=dev).manual_seed(seed)
    n = M * per_row
    r = torch.arange(M, device=dev, dtype=torch.int64).repeat_interleave(per_row)
    pos = r * K // M
    if kind in ("band32k", "band1k"):
        half = 32768 if kind == "band32k" else 1024
        c = pos + torch.randint(-half, half + 1, (n,), device=dev, generator=g, dtype=torch.int64)
    elif kind == "community":
        size = K // 256
        inside = torch.rand(n, device=dev, generator=g) < 0.9
        c_in = (pos // size) * size + torch.randint(0, size, (n,), device=dev, generator=g, dtype=torch.int64)
        c_out = torch.randint(0, K, (n,), device=dev, generator=g, dtype=torch.int64)
        c = torch.where(inside, c_in, c_out)
    else:
        raise ValueError(kind)
    c = c.clamp_(0, K - 1)
    if shuffle:
        perm = torch.randperm(M, device=dev, generator=g)
        r = perm[r]
    keys = torch.unique(r * K + c)
    del r, c
    col = (keys % K).to(torch.int32)
    cnt = torch.bincount(keys // K, minlength=M)
    del keys
    rowptr = torch.zeros(M + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(cnt, 0)
    val = torch.rand(col.numel(), device=dev, generator=g)
    return rowptr.to(torch.int32), col, val


CASES = [(f"{kind}/{tag}", kind, M, per) for tag, M, per in (("1M", 1 << 20, 100), ("128K", 1 << 17, 100))
         for kind in ("band32k", "band1k", "community")]


def run(tag, kind, M, per_row, N, out):
    t0 = time.time()
    K = M
    B = torch.rand(K, N, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    C = torch.empty(M, N, device=dev)
    res = {}
    for label, shuffle in (("natural", False), ("shuffled", True)):
        rowptr, col, val = structured_csr(M, K, per_row, kind, 7, shuffle)
        nnz = col.numel()
        entries = {label: (lambda rp=rowptr, c=col, v=val, z=nnz: custom_mm.naive_spmm(v, c, rp, z, M, K, B, C))}
        info = None
        if shuffle:
            sched = custom_mm.spmm_schedule(rowptr, nnz, M, N, col, K)
            info = sched.info()
            entries["shuffled+schedule"] = (lambda rp=rowptr, c=col, v=val, z=nnz, s=sched:
                                            custom_mm.naive_spmm_scheduled(s, v, c, rp, z, M, K, B, C))
            Cp = torch.empty_like(C)
            custom_mm.naive_spmm(val, col, rowptr, nnz, M, K, B, Cp)
            custom_mm.naive_spmm_scheduled(sched, val, col, rowptr, nnz, M, K, B, C)
            assert torch.equal(C.view(torch.int32), Cp.view(torch.int32)), "the scheduled product differs from the plain one"
            del Cp
        ms = time_interleaved(entries)
        custom_mm.naive_spmm(val, col, rowptr, nnz, M, K, B, C)
        ok, nr = check_rows(rowptr, col, val, B, C, K)
        assert ok, "sampled rows differ from the oracle"
        alg = nnz * (4 * N + 8) + 4 * (M + 1) + 4 * M * N
        for k, v in ms.items():
            res[k] = (v, alg / v / 8e9, custom_mm.spmm_plan(nnz, M, K, B, C)[1])
        if info is not None:
            res["info"] = info
        del rowptr, col, val
        torch.cuda.empty_cache()
    i = res["info"]
    line = (f"{tag:<16} N={N} |B|={K * N * 4 / 2**20:5.0f} MiB  natural {res['natural'][0]:8.3f} ms ({res['natural'][1]:.3f})  "
            f"shuffled {res['shuffled'][0]:8.3f} ms ({res['shuffled'][1]:.3f})  shuffled+schedule {res['shuffled+schedule'][0]:8.3f} ms "
            f"({res['shuffled+schedule'][1]:.3f})  gain {res['shuffled'][0] / res['shuffled+schedule'][0]:4.2f}x  plan {res['shuffled'][2]}  "
            f"locality_order {i['locality_order']} window span natural {i['window_span_natural']:.3f} -> {i['window_span_scheduled']:.3f} "
            f"row span {i['row_span']:.3f}  rows bit-exact   [{time.time() - t0:.0f} s]")
    print(line, flush=True)
    out.append(line)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--log", default="")
    ap.add_argument("--one", default="", help="one case, one mode, a few products: a target for rocprofv3 --pmc")
    ap.add_argument("--mode", default="plain", choices=["plain", "scheduled", "natural"])
    a = ap.parse_args()
    import oracle
    oracle.build()
    N = 256
    if a.one:
        tag, kind, M, per = next(c for c in CASES if c[0] == a.one)
        rowptr, col, val = structured_csr(M, M, per, kind, 7, a.mode != "natural")
        nnz = col.numel()
        B = torch.rand(M, N, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
        C = torch.empty(M, N, device=dev)
        sched = custom_mm.spmm_schedule(rowptr, nnz, M, N, col, M) if a.mode == "scheduled" else None
        for _ in range(5):
            if sched is None:
                custom_mm.naive_spmm(val, col, rowptr, nnz, M, M, B, C)
            else:
                custom_mm.naive_spmm_scheduled(sched, val, col, rowptr, nnz, M, M, B, C)
        torch.cuda.synchronize()
        print(f"{tag} {a.mode}: nnz {nnz}, algorithmic bytes per product {nnz * (4 * N + 8) + 4 * (M + 1) + 4 * M * N}")
        return
    print(f"# device {torch.cuda.get_device_name(0)}; ms per product (fraction of 8 TB/s on the algorithmic bytes)", flush=True)
    out = []
    for tag, kind, M, per in CASES:
        if a.only and a.only not in tag:
            continue
        run(tag, kind, M, per, N, out)
    if a.log:
        Path(a.log).parent.mkdir(parents=True, exist_ok=True)
        Path(a.log).write_text("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
