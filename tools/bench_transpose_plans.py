"""CSR transpose: the table plan against the one-sweep (decoupled look-back) plan through the C-ABI — same bits,
the one-sweep plan's give-up flag, and the time of each (HIP events, best of `reps`).

    python tools/bench_transpose_plans.py [--shape c3|mid|skew] [--reps 5]
"""
import argparse
import ctypes
import sys
import time
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
for p in (str(REPO), str(REPO / "matrix-multiplication_amd")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="c3")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import synthetic
    lib = ctypes.CDLL(str(REPO / "matrix-multiplication_amd" / "libmi_spmm.so"))
    vp, i64, i32, sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_size_t
    lib.mi_csr_transpose_workspace_bytes.restype = sz
    lib.mi_csr_transpose_workspace_bytes.argtypes = [i32, i32, i64]
    lib.mi_csr_transpose_f32.argtypes = [vp, vp, vp, i64, i32, i32, vp, vp, vp, vp, sz, vp]
    lib.mi_csr_transpose_check.argtypes = [vp, sz, i32, i32, i32, i64, vp]
    lib.mi_csr_transpose_one_sweep_applies.argtypes = [i32, i32, i32, i64]
    dev = torch.device("cuda", 0)
    if args.shape == "c3":
        M = K = 1 << 20
        rowptr, col, val = synthetic.make_csr(M, K, 1e-4, seed=0)
    elif args.shape == "mid":
        M, K = 200_000, 300_000
        rowptr, col, val = synthetic.make_csr(M, K, 1e-4, seed=3)
    else:  # skewed: a few very long rows and a hub column
        M, K = 300_000, 1 << 20
        g = np.random.Generator(np.random.PCG64(5))
        lens = np.minimum((g.pareto(1.2, M) * 8).astype(np.int64) + 1, 200_000)
        rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        col = np.concatenate([np.sort(g.choice(K, n, replace=False)) for n in lens]).astype(np.int32)
        col[rowptr[:-1][lens > 0]] = 7  # hub column (first entry of every row; 7 may repeat inside a row: stable order)
        val = g.random(len(col), dtype=np.float32)
    nnz = len(val)
    print(f"shape {args.shape}: M {M} K {K} nnz {nnz}; one-sweep applies: {lib.mi_csr_transpose_one_sweep_applies(1, M, K, nnz)}",
          flush=True)
    d_rp, d_col, d_val = (torch.from_numpy(x).to(dev) for x in (rowptr, col, val))
    ws_bytes = lib.mi_csr_transpose_workspace_bytes(M, K, nnz)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    out = {}
    for plan, name in ((1, "tables"), (2, "one_sweep")):
        assert lib.mi_csr_transpose_set_plan(plan) == 0
        t_rp = torch.full((K + 1,), -1, dtype=torch.int32, device=dev)
        t_col = torch.full((nnz,), -1, dtype=torch.int32, device=dev)
        t_val = torch.full((nnz,), -1.0, device=dev)

        def run():
            st = lib.mi_csr_transpose_f32(d_rp.data_ptr(), d_col.data_ptr(), d_val.data_ptr(), nnz, M, K, t_rp.data_ptr(),
                                          t_col.data_ptr(), t_val.data_ptr(), ws.data_ptr(), ws_bytes, stream)
            assert st == 0, st
        run()
        torch.cuda.synchronize()
        assert lib.mi_csr_transpose_check(ws.data_ptr(), ws_bytes, 1, M, K, nnz, stream) == 0, "look-back gave up"
        best = 1e9
        for _ in range(args.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        assert lib.mi_csr_transpose_check(ws.data_ptr(), ws_bytes, 1, M, K, nnz, stream) == 0, "look-back gave up"
        print(f"{name:10s} {best:8.3f} ms  ({16.0 * nnz / best / 1e6:7.0f} GB/s of the 16 B per non-zero)", flush=True)
        out[name] = (t_rp.clone(), t_col.clone(), t_val.clone())
    lib.mi_csr_transpose_set_plan(0)
    for a, b, what in zip(out["tables"], out["one_sweep"], ("t_rowptr", "t_col", "t_val")):
        same = torch.equal(a, b)
        print(f"{what}: {'identical' if same else 'DIFFERENT'}", flush=True)
        if not same:
            bad = (a != b).nonzero().flatten()
            print("  first differences at", bad[:10].tolist(), "count", bad.numel())
    # and against scipy on the host
    import scipy.sparse as sp
    ref = sp.csr_matrix((val, col, rowptr), shape=(M, K))
    if args.shape != "skew":
        refT = ref.T.tocsr()
        refT.sort_indices()
        assert np.array_equal(out["one_sweep"][0].cpu().numpy(), refT.indptr.astype(np.int32)), "t_rowptr vs scipy"
        assert np.array_equal(out["one_sweep"][1].cpu().numpy(), refT.indices.astype(np.int32)), "t_col vs scipy"
        assert np.array_equal(out["one_sweep"][2].cpu().numpy(), refT.data), "t_val vs scipy"
        print("one_sweep equals scipy's transpose", flush=True)


if __name__ == "__main__":
    main()
