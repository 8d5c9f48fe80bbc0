"""Generates tests/golden/golden_v2.npz from torch-CPU: fixtures for what round 6 added to the product path's arithmetic —
the dense product's split-k order (include/mi_spmm.h "Deterministic split-k"): shapes with few output tiles and k >= 4096.

The expectation is the reference's own: `torch.matmul(a, b)` in fp32 under torch.allclose defaults
(reference tests/cublas_kernel_test.py:19-28), stored beside the fp64 product rounded to fp32.  Nothing from the reference
is imported or executed; torch is the reference's own dependency.  Inputs are U[0,1) like the reference tests' torch.rand.

    python tests/golden/make_golden_v2.py        # rewrites golden_v2.npz
"""
from pathlib import Path

import numpy as np
import torch

OUT = Path(__file__).resolve().parent / "golden_v2.npz"


def main():
    g = torch.Generator().manual_seed(4321)
    out, names = {}, []
    # inputs are multiples of 1/256 in [0, 1) stored as uint8 (a = a_u8 / 256): exactly representable in fp32, and a fixture of
    # random fp32 words of these k would be megabytes
    for name, m, n, k, ta, tb in (("tn_16x8x4096", 16, 8, 4096, True, False), ("nn_8x16x8192", 8, 16, 8192, False, False),
                                  ("nt_9x5x4160", 9, 5, 4160, False, True), ("tt_4x4x16384", 4, 4, 16384, True, True)):
        a8 = torch.randint(0, 256, (k, m) if ta else (m, k), generator=g, dtype=torch.uint8)
        b8 = torch.randint(0, 256, (n, k) if tb else (k, n), generator=g, dtype=torch.uint8)
        a, b = a8.float() / 256.0, b8.float() / 256.0
        aa, bb = (a.t() if ta else a), (b.t() if tb else b)
        out[f"gemm/{name}/a_u8"], out[f"gemm/{name}/b_u8"] = a8.numpy(), b8.numpy()
        out[f"gemm/{name}/transa"], out[f"gemm/{name}/transb"] = np.array(ta), np.array(tb)
        out[f"gemm/{name}/c"] = torch.matmul(aa, bb).numpy()
        out[f"gemm/{name}/c_fp64"] = torch.matmul(aa.double(), bb.double()).float().numpy()
        names.append(f"gemm/{name}")
    out["__names__"] = np.array(names)
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT} ({OUT.stat().st_size} bytes, {len(names)} cases)")


if __name__ == "__main__":
    main()
