"""Phase stamps of the one-workgroup-per-item LDS transpose (csrc/csr_transpose.hip built with -DMI_TR_ITEM_TIMING):
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Imatrix-multiplication_amd/csrc -DMI_TR_ITEM_TIMING -shared \
          matrix-multiplication_amd/csrc/csr_transpose.hip matrix-multiplication_amd/csrc/mi_status.hip -o tools/probes/tr_item_timing_probe.so
    python tools/probes/tr_item_timing.py
Prints, per density, the kernel's time and the mean cycles a workgroup's thread 0 spent in each phase."""
import ctypes
import sys
from pathlib import Path
import torch
here = Path(__file__).resolve().parent
lib = ctypes.CDLL(str(here / "tr_item_timing_probe.so"))
vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
lib.mi_csr_transpose_batched_f32.argtypes = [vp, vp, vp, i64, i32, i32, i32, vp, vp, vp, vp, ctypes.c_size_t, vp]
lib.mi_tr_item_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
dev = torch.device("cuda")
S, items = 512, 384
g = torch.Generator(device=dev).manual_seed(0)
for kept in (0.15, 0.10, 0.05):  # (beyond 4 column passes the LDS plan declines: 25 % kept goes to the general plan)
    per_item = int(S * S * kept)
    idx = torch.rand(items, S * S, device=dev, generator=g).topk(per_item, dim=1).indices.sort(dim=1).values
    col = (idx % S).to(torch.int32).reshape(-1).contiguous()
    row = idx // S
    counts = torch.zeros(items, S, dtype=torch.int64, device=dev).scatter_add_(1, row, torch.ones_like(row))
    off = torch.zeros(items, S + 1, dtype=torch.int64, device=dev)
    off[:, 1:] = counts.cumsum(1)
    off += (torch.arange(items, device=dev) * per_item).unsqueeze(1)
    off = off.to(torch.int32).contiguous()
    val = torch.rand(items * per_item, device=dev)
    t_off = torch.empty(items, S + 1, dtype=torch.int32, device=dev)
    t_col = torch.empty(items * per_item, dtype=torch.int32, device=dev)
    t_val = torch.empty(items * per_item, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    call = lambda: lib.mi_csr_transpose_batched_f32(off.data_ptr(), col.data_ptr(), val.data_ptr(), items * per_item, items, S, S,
                                                    t_off.data_ptr(), t_col.data_ptr(), t_val.data_ptr(), None, 0, st)
    assert call() == 0
    buf = (ctypes.c_ulonglong * 8)()
    lib.mi_tr_item_stamps(buf)
    iters = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    lib.mi_tr_item_stamps(buf)
    wgs = None
    names = ["bounds+zero", "count", "cursors", "place", "copy out"]
    print(f"kept {kept}: {e0.elapsed_time(e1) / iters * 1e3:.1f} us per transpose; cycle sums per launch (thread 0 of every workgroup): "
          + "  ".join(f"{n} {buf[i] / iters / 1e3:.0f}k" for i, n in enumerate(names)), flush=True)
