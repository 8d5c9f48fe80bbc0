"""Where a fresh-pattern backward of the pruned-attention leg spends its time: each piece of matmuls._batched_csr_pattern /
_batched_csr_backward timed alone (384 x 512 x 512 at MI_KEPT, D = 64).   python tools/probes/attn_fresh_pieces.py"""
import os
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent / "matrix-multiplication_amd"))
import custom_mm  # noqa: E402
dev = torch.device("cuda")


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


S = int(os.environ.get("MI_SEQ", "512"))
items, D = 384 * 512 * 512 // (S * S), 64
g = torch.Generator(device=dev).manual_seed(0)
for kept in (0.25, 0.10, 0.05):
    per_item = int(S * S * kept)
    idx = torch.rand(items, S * S, device=dev, generator=g).topk(per_item, dim=1).indices
    mask = torch.zeros(items, S * S, device=dev).scatter_(1, idx, 1.0)
    probs = ((torch.rand(items, S * S, device=dev, generator=g) * 0.9 + 0.1) * mask).reshape(items, S, S)
    a = probs.to_sparse_csr()
    crow, col, val = a.crow_indices(), a.col_indices(), a.values()
    nb, total = items, items * per_item
    dc = torch.rand(items, S, D, device=dev, generator=g)
    v = torch.rand(items, S, D, device=dev, generator=g)
    out = torch.empty(items, S, D, device=dev)
    t = {}
    t["narrow (one launch)"] = timeit(lambda: custom_mm.batched_csr_narrow(crow.contiguous(), col.reshape(nb, per_item).contiguous()))
    base = torch.arange(nb, device=dev, dtype=torch.int64).unsqueeze(1) * per_item
    t["narrow (torch ops)"] = timeit(lambda: ((crow + base).to(torch.int32).contiguous(), col.reshape(-1).to(torch.int32).contiguous()))
    off32, col32 = custom_mm.batched_csr_narrow(crow.contiguous(), col.reshape(nb, per_item).contiguous())
    fval = val.reshape(-1).contiguous()
    t["flat_off cat"] = timeit(lambda: torch.cat([off32[:, :-1].reshape(-1), off32[-1:, -1]]).contiguous())
    t["shift repeat_interleave + add"] = timeit(lambda: col32 + (torch.arange(nb, device=dev, dtype=torch.int32) * S).repeat_interleave(per_item))
    t["iota"] = timeit(lambda: torch.arange(total, device=dev, dtype=torch.int32).view(torch.float32))
    iota = torch.arange(total, device=dev, dtype=torch.int32).view(torch.float32)
    t["csr_transpose_batched(iota)"] = timeit(lambda: custom_mm.csr_transpose_batched(iota, col32, off32, total, nb, S, S))
    t["csr_transpose_batched(values)"] = timeit(lambda: custom_mm.csr_transpose_batched(fval, col32, off32, total, nb, S, S))
    t_val, t_col, t_off = custom_mm.csr_transpose_batched(fval, col32, off32, total, nb, S, S)
    t_perm = custom_mm.csr_transpose_batched(iota, col32, off32, total, nb, S, S)[0].view(torch.int32)
    t["forward spmm_batched"] = timeit(lambda: custom_mm.naive_spmm_batched(fval, col32, off32, total, nb, S, S, v, out))
    t["A^T dC: plain product on transposed values"] = timeit(lambda: custom_mm.naive_spmm_batched(t_val, t_col, t_off, total, nb, S, S, dc, out))
    t["A^T dC: values through the permutation"] = timeit(lambda: custom_mm.naive_spmm_batched_perm(fval, t_perm, t_col, t_off, total, nb, S, S, dc, out))
    t["csr_transpose_batched(values) again"] = timeit(lambda: custom_mm.csr_transpose_batched(fval, col32, off32, total, nb, S, S))
    t["A^T dC: transpose-free (spmm_at)"] = timeit(lambda: custom_mm.naive_spmm_batched_at(fval, col32, off32, total, nb, S, S, dc, out))
    gv = torch.empty(total, device=dev)
    t["sddmm_batched"] = timeit(lambda: custom_mm.sddmm_batched(col32, off32, total, nb, S, S, dc, v, gv))
    t["sparse_csr_tensor(grad) construction"] = timeit(lambda: torch.sparse_csr_tensor(crow, col, gv.reshape(val.shape), size=a.shape))
    print(f"# {items} x {S}x{S} kept {kept}: us per piece")
    for k, x in t.items():
        print(f"  {k:<46} {x:9.1f}", flush=True)
