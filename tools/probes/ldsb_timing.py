"""Developer probe: phase stamps of the quad form of MI_SPMM_LDS_B (csrc/spmm_ldsb.hip built with -DMI_LDSB_TIMING into
tools/probes/ldsb_timing_probe.so): per workgroup, microseconds from the launch's earliest stamp to entry, the end of
each staging and the end of each unit.

  hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -DMI_LDSB_TIMING -Iinclude -Imatrix-multiplication_amd/csrc \
      matrix-multiplication_amd/csrc/spmm_ldsb.hip matrix-multiplication_amd/csrc/mi_status.hip -o tools/probes/ldsb_timing_probe.so
"""
import ctypes
import sys
from pathlib import Path
import numpy as np
import torch
REPO = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(REPO / "matrix-multiplication_amd"))
import custom_mm  # noqa: E402
lib = ctypes.CDLL(str(Path(__file__).resolve().parent / "ldsb_timing_probe.so"))
vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
lib.mi_ldsb_probe.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i64, vp]
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
items, M, K, N = 384, 512, 512, 64
v = torch.rand(items, K, N, device=dev, generator=g)
c = torch.empty(items, M, N, device=dev)
for kept in (0.01, 0.1, 1.0):
    probs = torch.rand(items, M, K, device=dev, generator=g)
    if kept < 1:
        probs = probs * (torch.rand(items, M, K, device=dev, generator=g) < kept)
    val, col, off = custom_mm.dense_to_csr(probs)
    stamps = np.zeros(512 * 16 + 512 * 16 * 4, dtype=np.uint64)
    for _ in range(3):  # the last of three launches is the one read
        st = lib.mi_ldsb_probe(off.data_ptr(), col.data_ptr(), val.data_ptr(), v.data_ptr(), c.data_ptr(), items, M, K, N,
                               val.numel(), stamps.ctypes.data)
        assert st == 0, st
    wave_end = stamps[512 * 16:].reshape(512, 16, 4).astype(np.float64)
    stamps = stamps[:512 * 16].reshape(512, 16)
    s = stamps[:256, :8].astype(np.float64)
    t0 = s[:, 0].min()
    us = np.where(s > 0, (s - t0) / 100.0, np.nan)  # 100 MHz
    names = ["entry", "staged1", "unit1", "unit2|staged2", "staged2|unit2", "unit3", "", ""]
    print(f"kept {kept}: nnz {val.numel()}")
    for wg in (0, 1, 2, 3, 128, 255):
        print(f"  wg {wg:3d}: " + "  ".join(f"{x:7.2f}" for x in us[wg, :7]))
    print("  mean   : " + "  ".join(f"{x:7.2f}" for x in np.nanmean(us[:, :7], axis=0)))
    print("  max    : " + "  ".join(f"{x:7.2f}" for x in np.nanmax(us[:, :7], axis=0)))
    for wg in (0, 1):
        for unit in range(3):
            print(f"  wg {wg} unit {unit}: waves leave at " + " ".join(f"{(x - t0) / 100.0:6.1f}" for x in wave_end[wg, :, unit]))
    print("  even wgs: entry, staged, unit, unit, staged, unit | odd wgs: entry, staged, unit, staged, unit, unit")
