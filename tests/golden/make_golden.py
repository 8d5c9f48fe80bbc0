"""Generates tests/golden/golden_v1.npz from torch-CPU.

The reference stores no golden vectors; what its tests pin is
`out ≈ torch.matmul(a, b)` under torch.allclose defaults on the shape list of
tests/naive_kernel_test.py:62-68, tests/cublas_kernel_test.py:68-69,
tests/cusparse_kernel_test.py:32-38 and tests/tiledsppm_kernel_test.py:34-39.
This script evaluates exactly that expectation (torch.matmul, and torch
autograd of torch.matmul for the gradients) on those shapes — scaled down where
the original would not be a small fixture — plus the edge cases the kernels
must survive, and stores inputs + expected outputs.  Nothing from the
reference is imported or executed; torch is the reference's own dependency.

    python tests/golden/make_golden.py        # rewrites golden_v1.npz

Inputs are U[0,1) like the reference tests' torch.rand / random.random(), so a
relative tolerance of 1e-5 is well-posed for every output element.
"""
from pathlib import Path

import numpy as np
import torch

OUT = Path(__file__).resolve().parent / "golden_v1.npz"


def csr_arrays(a: torch.Tensor):
    """CSR of a dense 2-d tensor exactly as matmuls.get_sparse_tensor_properties
    extracts it (reference matmuls.py:178-187): to_sparse_csr() → values,
    col_indices → int32, crow_indices → int32."""
    s = a.to_sparse_csr()
    return (s.crow_indices().to(torch.int32).numpy(), s.col_indices().to(torch.int32).numpy(),
            s.values().numpy())


def sparse_rand(g, rows, cols, density):
    a = torch.rand(rows, cols, generator=g)
    keep = torch.rand(rows, cols, generator=g) < density
    return a * keep


def main():
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(1234)
    out = {}
    names = []

    def add_spmm(name, a, b, with_grads=True):
        rp, ci, v = csr_arrays(a)
        a_ = a.clone().requires_grad_(True)
        b_ = b.clone().requires_grad_(True)
        c = torch.matmul(a_, b_)
        out[f"spmm/{name}/a"] = a.numpy()
        out[f"spmm/{name}/rowptr"], out[f"spmm/{name}/col"], out[f"spmm/{name}/val"] = rp, ci, v
        out[f"spmm/{name}/b"] = b.numpy()
        out[f"spmm/{name}/c"] = c.detach().numpy()
        if with_grads:
            dc = torch.rand(c.shape, generator=g)
            ga, gb = torch.autograd.grad(c, (a_, b_), dc)
            out[f"spmm/{name}/dc"] = dc.numpy()
            out[f"spmm/{name}/grad_a"] = ga.numpy()
            out[f"spmm/{name}/grad_b"] = gb.numpy()
        names.append(f"spmm/{name}")

    # reference tests/naive_kernel_test.py:62 — torch.rand has no zeros: 100 % dense CSR
    add_spmm("naive_4x2_2x3", torch.rand(4, 2, generator=g), torch.rand(2, 3, generator=g))
    # reference tests/cusparse_kernel_test.py:32-38, ~10 % dense A (last shape scaled 512x1024x256 → 64x128x32)
    for (m, k, n) in [(10, 10, 10), (10, 20, 10), (10, 10, 5), (20, 10, 5), (64, 128, 32)]:
        add_spmm(f"cusparse_{m}x{k}_{k}x{n}", sparse_rand(g, m, k, 0.1), torch.rand(k, n, generator=g))
    # dense widths around the lane-group boundaries of the HIP kernels
    for n in [1, 3, 4, 32, 33, 64, 128, 256, 257]:
        add_spmm(f"width_{n}", sparse_rand(g, 37, 29, 0.3), torch.rand(29, n, generator=g), with_grads=False)
    # empty rows, one row longer than a wave (nnz > 64), M not a multiple of the rows per workgroup
    a = sparse_rand(g, 13, 200, 0.05)
    a[3] = 0
    a[7] = 0
    a[5] = torch.rand(200, generator=g)  # 200 nonzeros
    add_spmm("ragged_13x200", a, torch.rand(200, 16, generator=g))
    add_spmm("all_zero_a", torch.zeros(6, 9), torch.rand(9, 8, generator=g), with_grads=False)
    add_spmm("single_row", sparse_rand(g, 1, 50, 0.5), torch.rand(50, 12, generator=g), with_grads=False)

    def add_gemm(name, a, b, transa, transb):
        a_ = a.clone().requires_grad_(True)
        b_ = b.clone().requires_grad_(True)
        opa = a_.transpose(-1, -2) if transa else a_
        opb = b_.transpose(-1, -2) if transb else b_
        c = torch.matmul(opa, opb)
        dc = torch.rand(c.shape, generator=g)
        ga, gb = torch.autograd.grad(c, (a_, b_), dc)
        for k_, v_ in dict(a=a, b=b, c=c.detach(), dc=dc, grad_a=ga, grad_b=gb).items():
            out[f"gemm/{name}/{k_}"] = v_.numpy()
        out[f"gemm/{name}/flags"] = np.array([transa, transb])
        names.append(f"gemm/{name}")

    # C1 plumbing config (BASELINE.json configs[0]; README.md:24-30 shapes)
    add_gemm("c1_8x64_64x8", torch.rand(8, 64, generator=g), torch.rand(64, 8, generator=g), False, False)
    # reference tests/naive_kernel_test.py:63-64 small batched cases
    add_gemm("b3_2x4x2_2x2x3", torch.rand(2, 4, 2, generator=g), torch.rand(2, 2, 3, generator=g), False, False)
    add_gemm("b3_2x4x2_2x4x2_tb", torch.rand(2, 4, 2, generator=g), torch.rand(2, 4, 2, generator=g), False, True)
    # all four transpose combinations, sizes off every tile boundary
    for ta in (False, True):
        for tb in (False, True):
            m, n, k = 37, 45, 53
            a = torch.rand((k, m) if ta else (m, k), generator=g)
            b = torch.rand((n, k) if tb else (k, n), generator=g)
            add_gemm(f"t{int(ta)}{int(tb)}_37x45x53", a, b, ta, tb)
    # BERT attention shapes (reference tests/cublas_kernel_test.py:68-69, README.md:69-77)
    # scaled (256,16,512,64) → (2,3,32,16): q·kᵀ and probs·v
    add_gemm("bert_qk_2x3x32x16", torch.rand(2, 3, 32, 16, generator=g), torch.rand(2, 3, 32, 16, generator=g),
             False, True)
    add_gemm("bert_pv_2x3x32x32_16", torch.rand(2, 3, 32, 32, generator=g), torch.rand(2, 3, 32, 16, generator=g),
             False, False)
    # FC-layer call shape (reference benchmarks/cublas_fc_layer.py:41): 3-d activations × weight.t()
    w = torch.rand(24, 40, generator=g)
    add_gemm("fc_3d_x_wt", torch.rand(2, 5, 40, generator=g), w, False, True)

    # column-major executor (reference tests/tiledsppm_kernel_test.py:34-39 scaled; src/baseline_mm.cu:272-321)
    a = sparse_rand(g, 19, 23, 0.2)
    rp, ci, v = csr_arrays(a)
    x = torch.rand(11, 23, generator=g)  # activations [N, K] row-major == B column-major K×N
    y = torch.matmul(x, a.t())           # [N, M] row-major == C column-major M×N
    out["colmajor/fc/rowptr"], out["colmajor/fc/col"], out["colmajor/fc/val"] = rp, ci, v
    out["colmajor/fc/x"], out["colmajor/fc/y"], out["colmajor/fc/a"] = x.numpy(), y.numpy(), a.numpy()
    names.append("colmajor/fc")

    # COO sorted by row → CSR (reference src/sparse_mm.cu:110-134)
    a = sparse_rand(g, 9, 14, 0.3)
    coo = a.to_sparse_coo().coalesce()
    out["coo/a"] = a.numpy()
    out["coo/row"] = coo.indices()[0].to(torch.int32).numpy()
    out["coo/col"] = coo.indices()[1].to(torch.int32).numpy()
    out["coo/val"] = coo.values().numpy()
    rp, ci, v = csr_arrays(a)
    out["coo/rowptr"], out["coo/csr_col"], out["coo/csr_val"] = rp, ci, v
    names.append("coo")

    # batched sparse × dense (reference tests/naive_kernel_test.py:67, scaled (256,16,512,512)x(…,512,64))
    a = sparse_rand(g, 2 * 3 * 16 * 16, 1, 0.25).reshape(2, 3, 16, 16)
    b = torch.rand(2, 3, 16, 8, generator=g)
    a_ = a.clone().requires_grad_(True)
    b_ = b.clone().requires_grad_(True)
    c = torch.matmul(a_, b_)
    dc = torch.rand(c.shape, generator=g)
    ga, gb = torch.autograd.grad(c, (a_, b_), dc)
    for k_, v_ in dict(a=a, b=b, c=c.detach(), dc=dc, grad_a=ga, grad_b=gb).items():
        out[f"batched/bert/{k_}"] = v_.numpy()
    names.append("batched/bert")

    out["__names__"] = np.array(names)
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT} ({OUT.stat().st_size / 1024:.1f} KiB, {len(names)} cases)")


if __name__ == "__main__":
    main()
