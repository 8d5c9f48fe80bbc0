"""Developer probe for `rocprofv3 --pmc`: a few launches of MI_SPMM_LDS_B (AUTO: the quad form) and of its 16-lane form
on the pruned-attention shape at 100 % and 10 % kept.
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS -d out -- python3 tools/probes/ldsb_pmc.py"""
import ctypes
import sys
from pathlib import Path
import torch
PKG = Path(__file__).resolve().parent.parent.parent / "matrix-multiplication_amd"
sys.path.insert(0, str(PKG))
import custom_mm  # noqa: E402
lib = ctypes.CDLL(str(PKG / "libmi_spmm.so"))
lib.mi_spmm_ldsb_set_form.argtypes = [ctypes.c_int]
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
items, M, K, N = 384, 512, 512, 64
v = torch.rand(items, K, N, device=dev, generator=g)
c = torch.empty(items, M, N, device=dev)
for kept in (1.0, 0.1):
    probs = torch.rand(items, M, K, device=dev, generator=g)
    if kept < 1:
        probs = probs * (torch.rand(items, M, K, device=dev, generator=g) < kept)
    val, col, off = custom_mm.dense_to_csr(probs)
    for form in (-1, 0):
        lib.mi_spmm_ldsb_set_form(form)
        for _ in range(3):
            custom_mm.naive_spmm_batched(val, col, off, val.numel(), items, M, K, v, c)
    torch.cuda.synchronize()
lib.mi_spmm_ldsb_set_form(-1)
