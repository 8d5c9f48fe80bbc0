"""Developer probe: the pieces of the batched-CSR backward at BERT size, 10 % kept."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "matrix-multiplication_amd"))
import custom_mm, matmuls  # noqa: E402
dev = torch.device("cuda")


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


items, S, D, kept = 384, 512, 64, 0.10
per_item = int(S * S * kept)
idx = torch.rand(items, S * S, device=dev).topk(per_item, dim=1).indices
mask = torch.zeros(items, S * S, device=dev).scatter_(1, idx, 1.0)
probs = ((torch.rand(items, S * S, device=dev) * 0.9 + 0.1) * mask).reshape(items, S, S)
a = probs.to_sparse_csr()
v = torch.rand(items, S, D, device=dev)
g = torch.rand(items, S, D, device=dev)
offsets, columns, (flat_off, diag_columns, t_perm, t_col, t_off) = matmuls._batched_csr_pattern(a, dev, transposed=True)
val = a.values().reshape(-1)
total = val.numel()
gb = torch.empty(items, S, D, device=dev)
print("sddmm block-diagonal     ", timeit(lambda: custom_mm.sddmm(diag_columns, flat_off, total, items * S, items * S, g.reshape(-1, D), v.reshape(-1, D))))
print("index_select values      ", timeit(lambda: val.index_select(0, t_perm)))
t_val = val.index_select(0, t_perm)
print("batched spmm (transposed)", timeit(lambda: custom_mm.naive_spmm_batched(t_val, t_col, t_off, total, items, S, S, g, gb)))
gv = custom_mm.sddmm(diag_columns, flat_off, total, items * S, items * S, g.reshape(-1, D), v.reshape(-1, D))
print("sparse_csr_tensor ctor   ", timeit(lambda: torch.sparse_csr_tensor(a.crow_indices(), a.col_indices(), gv.reshape(a.values().shape), size=a.shape)))
print("forward product          ", timeit(lambda: custom_mm.naive_spmm_batched(val, columns, offsets, total, items, S, S, v, gb)))
