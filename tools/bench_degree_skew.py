"""The SpMM hot path on DEGREE-SKEWED matrices (power-law row lengths), beside a uniform-length twin of the same nnz.

    python tools/bench_degree_skew.py [--only TAG] [--log FILE] [--handle] [--quick]

The reference's kernel descends from merge-spmm ("Design Principles for Sparse Matrix Multiplication on the GPU",
reference src/naive_sparse_mm.cu:20-21) and was lifted from a GNN library: its home ground is adjacency matrices with
power-law degrees, not the Binomial row lengths of the pinned generator.  Every shape here is GNN-like (seeded, no
downloads):

  arxiv     170 K x 170 K, mean 14 per row,  N = 128
  reddit    233 K x 233 K, mean 490 per row, N = 602   (N % 4 != 0)
  products  2.4 M x 2.4 M, mean 50 per row,  N = 100
  c3skew    1 M x 1 M,     mean 105 per row, N = 256   (config C3's nnz with skewed lengths)
  tiny      4 M x 4 M,     1 … 8 per row,    N = 64 and 256

Row lengths: Pareto (shape 1.6 — heavy tail; a few rows hold a large share of the entries), scaled to the target mean
after clipping at 100 / 1 000 / 8 000 / no clip; columns uniform, unique and ascending within a row.  Per row of the
log: ms through `custom_mm.naive_spmm` (and, with --handle, through an inspector handle: `cusparse_inspect` +
`cusparse_mmul_opt`), algorithmic GB/s and its fraction of 8 TB/s (SURVEY.md 8d's byte count), the longest row, the
share of the entries in the longest 1 % of the rows, the SAME nnz with uniform lengths beside it and the ratio of the
two times; sampled rows (the longest ones included) bit-exact against the oracle.
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "matrix-multiplication_amd"))
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import custom_mm  # noqa: E402
from bench_hbm_regime import time_interleaved  # noqa: E402  (round-robin timing: the entries share clock / power state)

dev = torch.device("cuda")


def pareto_lengths(M, mean, clip, K, seed, shape=1.6, lo=None):
    """int64[M] row lengths on the device: Pareto(shape) scaled so that the CLIPPED lengths average `mean` (bisection on
    the scale), every row at least `lo` (default 1) and at most min(clip, K)."""
    g = torch.Generator(device=dev).manual_seed(seed)
    u = torch.rand(M, device=dev, generator=g, dtype=torch.float64).clamp_(min=1e-12)
    w = u.pow(-1.0 / shape)
    top = float(min(clip if clip else K, K))
    lo = 1.0 if lo is None else float(lo)
    a, b = 1e-6, float(mean) * 4
    for _ in range(60):
        s = 0.5 * (a + b)
        m = float((w * s).round_().clamp_(lo, top).mean())
        if m < mean:
            a = s
        else:
            b = s
    return (w * (0.5 * (a + b))).round_().clamp_(lo, top).to(torch.int64)


def csr_from_lengths(lens, K, seed):
    """(rowptr i32[M+1], col i32[nnz], val f32[nnz]) with about lens[r] entries in row r: uniform columns, duplicates
    removed (a long row of a narrow matrix loses a few), ascending within a row."""
    g = torch.Generator(device=dev).manual_seed(seed)
    M = lens.numel()
    cols, counts = [], []
    cum = torch.cumsum(lens, 0)
    r0 = 0
    budget = 1 << 27  # keys per block of rows
    while r0 < M:
        base = int(cum[r0 - 1]) if r0 else 0
        r1 = int(torch.searchsorted(cum, torch.tensor([base + budget], device=dev), right=True)[0])
        r1 = min(M, max(r1, r0 + 1))
        blk = lens[r0:r1]
        rows = torch.repeat_interleave(torch.arange(r1 - r0, device=dev, dtype=torch.int64), blk)
        keys = rows * K + torch.randint(0, K, (rows.numel(),), device=dev, generator=g, dtype=torch.int64)
        del rows
        keys = torch.unique(keys)
        cols.append((keys % K).to(torch.int32))
        counts.append(torch.bincount(keys // K, minlength=r1 - r0))
        del keys
        r0 = r1
    col = torch.cat(cols)
    rowptr = torch.zeros(M + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(torch.cat(counts), 0)
    assert int(rowptr[-1]) == col.numel() < 2**31
    val = torch.rand(col.numel(), device=dev, generator=g)
    return rowptr.to(torch.int32), col, val


def check_rows(rowptr, col, val, B, C, split_long, n_rows=160, seed=0):
    """Sampled rows of C — the longest rows among them — bit-exact against the oracle (`spmm_csr_long` where the plan
    sums rows beyond the long-row threshold in the split order, else the plain CSR-order chain)."""
    import oracle
    M = rowptr.numel() - 1
    lens = (rowptr[1:] - rowptr[:-1])
    longest = torch.topk(lens, min(8, M)).indices.cpu().numpy()
    rs = np.unique(np.concatenate([[0, M - 1], longest, np.random.default_rng(seed).integers(0, M, n_rows)]))
    rp = rowptr.cpu().numpy().astype(np.int64)
    segs = [np.arange(rp[r], rp[r + 1]) for r in rs]
    idx = torch.from_numpy(np.concatenate(segs)).to(dev)
    c, v = col[idx].cpu().numpy(), val[idx].cpu().numpy()
    sub_rp = np.concatenate([[0], np.cumsum([len(s) for s in segs])]).astype(np.int32)
    uniq, inv = np.unique(c, return_inverse=True)
    Bs = B[torch.from_numpy(uniq.astype(np.int64)).to(dev)].cpu().numpy()
    fn = oracle.spmm_csr_long if split_long else oracle.spmm_csr
    want = fn(sub_rp, inv.astype(np.int32), v, len(rs), len(uniq), Bs)
    got = C[torch.from_numpy(rs).to(dev)].cpu().numpy()
    return np.array_equal(want.view(np.uint32), got.view(np.uint32)), len(rs)


def product_entries(rowptr, col, val, M, K, B, C, handle, layer):
    nnz = col.numel()
    entries = {"naive_spmm": lambda: custom_mm.naive_spmm(val, col, rowptr, nnz, M, K, B, C)}
    # the inspector's row schedule (built once per matrix, outside the timed region): the same product, rows handed to
    # waves longest first / like lengths together / heavy rows apart — the same bits
    sched = custom_mm.spmm_schedule(rowptr, nnz, M, B.shape[1], col, K)  # (with the columns: the locality pass is tried, and must decline on uniform columns)
    Cs = torch.empty_like(C)
    custom_mm.naive_spmm(val, col, rowptr, nnz, M, K, B, C)
    custom_mm.naive_spmm_scheduled(sched, val, col, rowptr, nnz, M, K, B, Cs)
    assert torch.equal(C.view(torch.int32), Cs.view(torch.int32)), "the scheduled product differs from the plain one"
    del Cs
    entries["scheduled"] = lambda: custom_mm.naive_spmm_scheduled(sched, val, col, rowptr, nnz, M, K, B, C)
    entries["scheduled"].info = sched.info()
    if handle:
        # the inspector handle's executor takes COLUMN-major operands (reference src/baseline_mm.cu:272-321): Bt [N, K],
        # Ct [N, M] row-major
        N = B.shape[1]
        custom_mm.cusparse_inspect(rowptr, col, val, nnz, M, N, K, layer)
        Bt = B.t().contiguous()
        Ct = torch.empty(N, M, device=dev)
        entries["handle"] = lambda: custom_mm.cusparse_mmul_opt(Bt, Ct, layer)
    return entries


def run(tag, M, K, N, mean, clip, out, handle, lo=None, hi=None):
    t0 = time.time()
    if hi is not None:  # 'tiny': lengths uniform in lo … hi
        g = torch.Generator(device=dev).manual_seed(3)
        lens = torch.randint(lo, hi + 1, (M,), device=dev, generator=g, dtype=torch.int64)
    else:
        lens = pareto_lengths(M, mean, clip, K, seed=3)
    rowptr, col, val = csr_from_lengths(lens, K, seed=4)
    del lens
    nnz = col.numel()
    B = torch.rand(K, N, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    C = torch.empty(M, N, device=dev)
    plan = custom_mm.spmm_plan(nnz, M, K, B, C)
    rl = (rowptr[1:] - rowptr[:-1]).to(torch.int64)
    longest = int(rl.max())
    top1 = float(torch.topk(rl, max(1, M // 100)).values.sum()) / max(nnz, 1)
    ents = product_entries(rowptr, col, val, M, K, B, C, handle, "skew")
    info = ents["scheduled"].info
    ms = time_interleaved(ents)
    del ents
    custom_mm.naive_spmm(val, col, rowptr, nnz, M, K, B, C)
    ok, nr = check_rows(rowptr, col, val, B, C, split_long=bool(plan[3]))
    del rowptr, col, val
    torch.cuda.empty_cache()
    # the twin: the same number of entries, uniform lengths (unique uniform keys: the pinned generator's distribution)
    g = torch.Generator(device=dev).manual_seed(5)
    per = max(1, round(nnz / M))
    lens_u = torch.full((M,), per, device=dev, dtype=torch.int64)
    extra = nnz - per * M
    if extra > 0:
        lens_u[torch.randperm(M, device=dev, generator=g)[:extra]] += 1
    elif extra < 0:
        lens_u[torch.randperm(M, device=dev, generator=g)[:-extra]] -= 1
    rp_u, col_u, val_u = csr_from_lengths(lens_u, K, seed=6)
    nnz_u = col_u.numel()
    plan_u = custom_mm.spmm_plan(nnz_u, M, K, B, C)
    ents = product_entries(rp_u, col_u, val_u, M, K, B, C, handle, "twin")
    ms_u = time_interleaved(ents)
    del ents
    alg = nnz * (4 * N + 8) + 4 * (M + 1) + 4 * M * N
    alg_u = nnz_u * (4 * N + 8) + 4 * (M + 1) + 4 * M * N
    for name in ms:
        t, tu = ms[name], ms_u[name]
        # every entry against the twin's BEST entry: what the same entries cost with uniform lengths
        tu = min(ms_u[name], ms_u["naive_spmm"])
        line = (f"{tag:<22} {name:<10} M={M:>8} N={N:>4} nnz={nnz:>10} mean {nnz / M:6.1f} longest {longest:>8} top1% {top1:5.1%}  "
                f"plan {plan[0]:>2} {plan[1]:<30} x{plan[2]}  {t:8.3f} ms {alg / t / 1e6:6.0f} GB/s frac {alg / t / 8e9:.3f}  |  "
                f"uniform twin plan {plan_u[0]:>2} {tu:8.3f} ms frac {alg_u / tu / 8e9:.3f}  skewed/uniform {t / tu * nnz_u / nnz:5.2f}  "
                f"rows {'bit-exact' if ok else 'MISMATCH'} ({nr})" + (f"  heavy {info['heavy_rows']} rows > {info['heavy_length']} locality_order {info['locality_order']}" if name == "scheduled" else "") + f"   [{time.time() - t0:.0f} s]")
        print(line, flush=True)
        out.append(line)
    assert ok, "sampled rows differ from the oracle"
    if handle:
        custom_mm.cusparse_clean()
    del rp_u, col_u, val_u, B, C
    torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--log", default="")
    ap.add_argument("--handle", action="store_true", help="also through an inspector handle (cusparse_inspect + cusparse_mmul_opt)")
    ap.add_argument("--quick", action="store_true", help="clip 1000 and no clip only")
    a = ap.parse_args()
    import oracle
    oracle.build()
    custom_mm.init_cusparse()
    print(f"# device {torch.cuda.get_device_name(0)}; ms per product; frac = algorithmic GB/s / 8000; skewed/uniform = time per "
          f"entry relative to the uniform-length twin", flush=True)
    shapes = [("arxiv", 170_000, 128, 14), ("reddit", 233_000, 602, 490), ("products", 2_400_000, 100, 50),
              ("c3skew", 1 << 20, 256, 105)]
    clips = (1000, 0) if a.quick else (100, 1000, 8000, 0)
    out = []
    for name, M, N, mean in shapes:
        for clip in clips:
            if clip and clip <= mean:
                continue
            tag = f"{name}/clip{clip or 'none'}"
            if a.only and a.only not in tag:
                continue
            run(tag, M, M, N, mean, clip, out, a.handle)
    for N in (64, 256):
        tag = f"tiny/1-8/N{N}"
        if a.only and a.only not in tag:
            continue
        run(tag, 1 << 22, 1 << 22, N, 4.5, 0, out, a.handle, lo=1, hi=8)
    if a.log:
        Path(a.log).parent.mkdir(parents=True, exist_ok=True)
        Path(a.log).write_text("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
