"""Developer probe (not part of the product path): loads libmi_spmm.so through
ctypes next to torch, checks small SpMM cases against torch and times the
kernel variants on a C3-like shape.  Run on the GPU box:

    python tools/quick_check.py [--big]
"""
import ctypes
import sys
import time
from pathlib import Path

import torch

LIB = Path(__file__).resolve().parent.parent / "matrix-multiplication_amd" / "libmi_spmm.so"
lib = ctypes.CDLL(str(LIB))
i32p, f32p, vp = ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p
lib.mi_spmm_csr_f32_variant.argtypes = [ctypes.c_int, vp, vp, vp, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                        ctypes.c_int32, vp, ctypes.c_int64, vp, ctypes.c_int64, vp]
lib.mi_spmm_csr_f32_variant.restype = ctypes.c_int
lib.mi_status_string.restype = ctypes.c_char_p
lib.mi_last_hip_error_string.restype = ctypes.c_char_p


def spmm(variant, rowptr, col, val, M, K, B, C):
    N = B.shape[1]
    st = lib.mi_spmm_csr_f32_variant(variant, rowptr.data_ptr(), col.data_ptr(), val.data_ptr(), val.numel(), M, K, N,
                                     B.data_ptr(), B.stride(0), C.data_ptr(), C.stride(0),
                                     torch.cuda.current_stream().cuda_stream)
    if st != 0:
        raise RuntimeError(f"variant {variant}: {lib.mi_status_string(st)} / {lib.mi_last_hip_error_string()}")
    return C


def rand_csr(M, K, density, dev, gen):
    a = torch.rand(M, K, generator=gen)
    a = a * (torch.rand(M, K, generator=gen) < density)
    csr = a.to_sparse_csr()
    return a, csr.crow_indices().to(torch.int32).to(dev), csr.col_indices().to(torch.int32).to(dev), csr.values().to(dev)


def main():
    dev = torch.device("cuda")
    print("device:", torch.cuda.get_device_name(0), "torch", torch.__version__, "hip", torch.version.hip, flush=True)
    gen = torch.Generator().manual_seed(0)
    ok = True
    for (M, K, N) in [(4, 2, 3), (10, 20, 10), (33, 47, 64), (64, 64, 256), (130, 257, 512), (65, 300, 1024),
                      (100, 100, 128), (257, 129, 33), (50, 60, 1), (512, 1024, 256)]:
        a, rp, ci, v = rand_csr(M, K, 0.3, dev, gen)
        B = torch.rand(K, N, generator=gen)
        exp = a @ B
        Bd = B.to(dev)
        for variant in range(0, 17):
            C = torch.full((M, N), float("nan"), device=dev)
            try:
                spmm(variant, rp, ci, v, M, K, Bd, C)
            except RuntimeError as e:
                if "invalid argument" in str(e):
                    continue
                raise
            torch.cuda.synchronize()
            good = torch.allclose(exp, C.cpu(), rtol=1e-5, atol=1e-8)
            ok &= good
            print(f"  {M}x{K}x{N} variant {variant}: {'ok' if good else 'MISMATCH'} maxerr "
                  f"{(exp - C.cpu()).abs().max().item():.3e}", flush=True)
    print("small cases:", "PASS" if ok else "FAIL", flush=True)
    if not ok:
        sys.exit(1)

    if "--big" in sys.argv:
        for (M, K, N, deg) in [(1 << 20, 1 << 20, 256, 105), (1 << 16, 1 << 16, 128, 66), (1 << 18, 1 << 16, 64, 64), (1 << 17, 1 << 19, 512, 64)]:
            col = torch.randint(0, K, (M, deg), device=dev, dtype=torch.int32).sort(dim=1).values.reshape(-1).contiguous()
            val = torch.rand(M * deg, device=dev)
            rowptr = (torch.arange(M + 1, device=dev, dtype=torch.int64) * deg).to(torch.int32)
            B = torch.rand(K, N, device=dev)
            C = torch.empty(M, N, device=dev)
            nnz = M * deg
            bytes_alg = nnz * (4 * N + 8) + 4 * (M + 1) + 4 * M * N
            ref = None
            for variant in range(0, 17):
                try:
                    spmm(variant, rowptr, col, val, M, K, B, C)
                except RuntimeError as e:
                    if "invalid argument" in str(e):
                        continue
                    raise
                torch.cuda.synchronize()
                if ref is None:
                    ref = C.clone()
                    A = torch.sparse_csr_tensor(rowptr.long(), col.long(), val, (M, K))
                    t0 = time.time()
                    exp = A @ B
                    torch.cuda.synchronize()
                    print(f"  torch(hipSPARSE) csr@dense: {time.time() - t0:.3f}s; allclose(ours)="
                          f"{torch.allclose(exp, C, rtol=1e-5, atol=1e-8)}", flush=True)
                    del exp, A
                same = torch.equal(ref, C)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ts = []
                for _ in range(6):
                    e0.record()
                    spmm(variant, rowptr, col, val, M, K, B, C)
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1))
                t = min(ts[1:])
                print(f"  M={M} N={N} variant {variant}: {t:.3f} ms  {bytes_alg / t / 1e6:.0f} GB/s "
                      f"({bytes_alg / t / 1e6 / 8000:.3f} of 8 TB/s)  {2 * nnz * N / t / 1e6:.0f} GFLOP/s "
                      f"bit-equal-to-variant0={same}", flush=True)


if __name__ == "__main__":
    main()
