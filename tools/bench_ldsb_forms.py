"""Developer probe: the two forms of MI_SPMM_LDS_B (16 lanes per row / a quad per row, csrc/spmm_ldsb.hip) on batched CSR x
dense shapes whose B fits LDS: time of each, bit-equality, and a fuzz over ragged shapes (empty rows, rows ending at the
arrays' last entries, M not a multiple of the 256-row step) against the group kernel.
-> profiles/r04_ldsb_forms.log"""
import ctypes
import sys
from pathlib import Path
import torch
PKG = Path(__file__).resolve().parent.parent / "matrix-multiplication_amd"
sys.path.insert(0, str(PKG))
import custom_mm  # noqa: E402
lib = ctypes.CDLL(str(PKG / "libmi_spmm.so"))
vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
lib.mi_spmm_csr_batched_variant_f32.argtypes = [ctypes.c_int, vp, vp, vp, i64, i32, i32, i32, i32, vp, i64, i64, vp, i64, i64, vp]
lib.mi_spmm_ldsb_set_form.argtypes = [ctypes.c_int]
dev = torch.device("cuda")
GROUP, LDSB = 4, 18


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best


def run(variant, off, col, val, nnz, items, M, K, N, b, c):
    st = lib.mi_spmm_csr_batched_variant_f32(variant, off.data_ptr(), col.data_ptr(), val.data_ptr(), nnz, items, M, K, N,
                                             b.data_ptr(), N, K * N, c.data_ptr(), N, M * N,
                                             torch.cuda.current_stream().cuda_stream)
    if st == -1 and variant == LDSB:
        return False
    assert st == 0, (variant, st)
    return True


g = torch.Generator(device=dev).manual_seed(0)
quick = "--quick" in sys.argv
print("# tools/bench_ldsb_forms.py on MI355X (ms; best of 3 blocks of 20)")
# ---- fuzz first: a wrong kernel should not get to print timings
cpu = torch.Generator().manual_seed(1)
cases = 0
n_fuzz = 60 if not quick else 20
if "--fuzz" in sys.argv:
    n_fuzz = int(sys.argv[sys.argv.index("--fuzz") + 1])
for trial in range(n_fuzz):
    items = int(torch.randint(1, 40, (1,), generator=cpu))
    M = int(torch.randint(1, 700, (1,), generator=cpu))
    if trial % 5 == 1:
        items, M = int(torch.randint(30, 90, (1,), generator=cpu)), int(torch.randint(250, 600, (1,), generator=cpu))
    K = int(torch.randint(1, 513, (1,), generator=cpu))
    N = (64, 128, 32, 16, 64, 256)[trial % 6]
    if trial % 7 == 0:
        K = int(torch.randint(513, 2049, (1,), generator=cpu))  # narrower tiles
    if (K + 1) * 16 * 4 > 132 * 1024 or N // 16 > 8 and K > 512:
        K = 512
    kept = float(torch.rand(1, generator=cpu)) ** 2
    probs = torch.rand(items, M, K, device=dev, generator=g) * (torch.rand(items, M, K, device=dev, generator=g) < kept)
    if trial % 4 == 0:
        probs[:, ::3] = 0  # empty rows
    if trial % 5 == 0:
        probs[-1, -1, :] = 1.0  # the last row reaches the arrays' end
    val, col, off = custom_mm.dense_to_csr(probs)
    nnz = val.numel()
    if nnz < 4:
        continue
    v = torch.rand(items, K, N, device=dev, generator=g) - 0.5
    c1 = torch.full((items, M, N), 7.0, device=dev)
    c2 = torch.full((items, M, N), 8.0, device=dev)
    c3 = torch.full((items, M, N), 9.0, device=dev)
    run(GROUP, off, col, val, nnz, items, M, K, N, v, c1)
    lib.mi_spmm_ldsb_set_form(1)
    ok = run(LDSB, off, col, val, nnz, items, M, K, N, v, c2)
    lib.mi_spmm_ldsb_set_form(0)
    ok0 = run(LDSB, off, col, val, nnz, items, M, K, N, v, c3)
    lib.mi_spmm_ldsb_set_form(-1)
    if ok:
        assert torch.equal(c1.view(torch.int32), c2.view(torch.int32)), ("quad form differs", items, M, K, N, kept, nnz)
        cases += 1
    if ok0:
        assert torch.equal(c1.view(torch.int32), c3.view(torch.int32)), ("16-lane form differs", items, M, K, N, kept)
    if N == 64 and items * M >= 16384 and nnz >= 4 * items * M:
        # the batched SDDMM (quad form, row tiles of B beyond 512 rows) against the block-diagonal kernel
        dc = torch.rand(items, M, N, device=dev, generator=g) - 0.5
        o1 = torch.full((nnz,), 7.0, device=dev)
        if custom_mm.sddmm_batched(col, off, nnz, items, M, K, dc, v, o1):
            flat = torch.cat([off[:, :-1].reshape(-1), off[-1:, -1]])
            counts = (off[:, -1] - off[:, 0]).long()
            diag = (col.long() + torch.repeat_interleave(torch.arange(items, device=dev), counts) * K).to(torch.int32)
            ref = custom_mm.sddmm(diag, flat, nnz, items * M, items * K, dc.reshape(items * M, N), v.reshape(items * K, N))
            assert torch.equal(o1.view(torch.int32), ref.view(torch.int32)), ("sddmm quad form differs", items, M, K, kept)
            sddmm_cases = globals().get("sddmm_cases", 0) + 1
    del probs, val, col, off
print(f"# fuzz: {cases} ragged cases, quad form == group kernel bit for bit ({globals().get('sddmm_cases', 0)} of them also through the batched SDDMM)")
if "--fuzz" in sys.argv:
    sys.exit(0)

print("# items x M x K x N   kept     nnz        group(L2)  16-lane   quad     by rule")
shapes = [(384, 512, 512, 64, (1.0, 0.5, 0.25, 0.1, 0.05, 0.02, 0.01)),
          (384, 128, 128, 64, (0.5, 0.1)), (384, 256, 256, 64, (0.5, 0.1)), (96, 1024, 512, 64, (0.1, 0.02)),
          (192, 512, 256, 128, (0.25, 0.05)), (96, 512, 512, 128, (0.1,)), (1, 131072, 512, 64, (0.1, 0.02)),
          (4096, 64, 64, 64, (0.5,)), (1536, 197, 197, 64, (0.25,)), (96, 1024, 1024, 64, (0.25, 0.1, 0.02)),
          (48, 2048, 2048, 64, (0.1, 0.05, 0.02)), (96, 1024, 1024, 128, (0.1,)), (768, 512, 1024, 32, (0.1, 0.02))]
if quick:
    shapes = shapes[:1]
for items, M, K, N, kepts in shapes:
    v = torch.rand(items, K, N, device=dev, generator=g)
    c1, c2, c3, c4 = (torch.empty(items, M, N, device=dev) for _ in range(4))
    for kept in kepts:
        probs = torch.rand(items, M, K, device=dev, generator=g)
        if kept < 1:
            probs = probs * (torch.rand(items, M, K, device=dev, generator=g) < kept)
        val, col, off = custom_mm.dense_to_csr(probs)
        nnz = val.numel()
        t_grp = timeit(lambda: run(GROUP, off, col, val, nnz, items, M, K, N, v, c1))
        lib.mi_spmm_ldsb_set_form(0)
        t16 = timeit(lambda: run(LDSB, off, col, val, nnz, items, M, K, N, v, c2))  # (a shape of the quad form only: the group kernel again)
        lib.mi_spmm_ldsb_set_form(1)
        tq = timeit(lambda: run(LDSB, off, col, val, nnz, items, M, K, N, v, c3))
        lib.mi_spmm_ldsb_set_form(-1)
        tr = timeit(lambda: run(LDSB, off, col, val, nnz, items, M, K, N, v, c4))
        assert torch.equal(c1, c2) and torch.equal(c1, c3) and torch.equal(c1, c4)
        print(f"{items:5d} x {M:6d} x {K:4d} x {N:3d}  {kept:5.2f} {nnz:11d}   {t_grp:8.4f}  {t16:8.4f}  {tq:8.4f}  {tr:8.4f}", flush=True)
        del probs, val, col, off
