'''
random_tensor_benchmark — dense vs sparse product over square sizes and sparsities.

Working counterpart of the reference's benchmarks/random_tensor_benchmark.py, whose intended
sweep is dims {1024, 4096, 8192, 12288, 16384}² × sparsity {0, .25, .5, .75, .9, .99}
(reference :70-73) but which cannot run as written (it imports `cublas_matmul` /
`cusparse_matmul`, names matmuls.py does not define, reference :11-14).  This one is seeded,
synchronises the device around the timed region (the reference's tests time without a sync,
tests/naive_kernel_test.py:18-20) and writes one JSON record per configuration.

    python matrix-multiplication_amd/benchmarks/random_tensor_benchmark.py [--dims 1024 4096] [--out results.jsonl]
'''
import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import custom_mm  # noqa: E402
import matmuls  # noqa: E402


def time_ms(fn, iters, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dims", type=int, nargs="+", default=[1024, 4096, 8192, 12288, 16384])
    ap.add_argument("--sparsity", type=float, nargs="+", default=[0.0, 0.25, 0.5, 0.75, 0.9, 0.99])
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda")
    custom_mm.init_cublas()
    custom_mm.init_cusparse()
    out = open(args.out, "w") if args.out else None
    for n in args.dims:
        g = torch.Generator(device=dev).manual_seed(n)
        b = torch.rand(n, n, device=dev, generator=g)
        for sp in args.sparsity:
            a = torch.rand(n, n, device=dev, generator=g)
            if sp > 0:
                a = a * (torch.rand(n, n, device=dev, generator=g) >= sp)
            a_csr = a.to_sparse_csr()
            props = matmuls.get_sparse_tensor_properties(a_csr)
            c = torch.empty(n, n, device=dev)
            rec = {
                "dim": n, "sparsity": sp, "nnz": props[3],
                "dense_cublasMM_ms": time_ms(lambda: matmuls.cublasMM.apply(a, b), args.iters),
                "sparse_kernel_only_ms": time_ms(lambda: custom_mm.naive_spmm(*props, b, c), args.iters),
                "sparse_cusparseMM_csr_input_ms": time_ms(lambda: matmuls.cusparseMM.apply(a_csr, b), args.iters),
                "sparse_naiveSpMM_dense_input_ms": time_ms(lambda: matmuls.naiveSpMM.apply(a, b), args.iters),
                "torch_matmul_ms": time_ms(lambda: torch.matmul(a, b), args.iters),
            }
            line = json.dumps(rec)
            print(line, flush=True)
            if out:
                out.write(line + "\n")
            del a, a_csr, props
    if out:
        out.close()
    custom_mm.destroy_cusparse()
    custom_mm.destroy_cublas()


if __name__ == "__main__":
    main()
