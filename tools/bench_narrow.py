import sys, torch
sys.path.insert(0, "matrix-multiplication_amd")
import custom_mm
dev = torch.device("cuda")
M = K = 1 << 20
deg = 105
col = torch.randint(0, K, (M, deg), device=dev, dtype=torch.int32).sort(dim=1).values.reshape(-1).contiguous()
val = torch.rand(M * deg, device=dev)
rowptr = (torch.arange(M + 1, device=dev, dtype=torch.int64) * deg).to(torch.int32)
for N in (1, 2, 4, 8, 16, 32):
    B = torch.rand(K, N, device=dev); C = torch.empty(M, N, device=dev)
    for _ in range(2): custom_mm.naive_spmm(val, col, rowptr, M * deg, M, K, B, C)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): custom_mm.naive_spmm(val, col, rowptr, M * deg, M, K, B, C)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5
    A = torch.sparse_csr_tensor(rowptr.long(), col.long(), val, (M, K))
    for _ in range(2): A @ B
    torch.cuda.synchronize(); e0.record()
    for _ in range(5): A @ B
    e1.record(); torch.cuda.synchronize()
    print(f"N={N}: ours {t:.3f} ms ({M*deg*(8+4*N)/t/1e6:.0f} GB/s alg)  torch/hipSPARSE {e0.elapsed_time(e1)/5:.3f} ms")
