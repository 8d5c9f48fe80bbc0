"""C5 (BERT-base attention, dense fp32, fwd + bwd through the drop-in classes): where the step's time is.

    python tools/probes/c5_graph.py [--steps 100]

Three timings of the SAME six products on the same operands, taking turns (round-robin, so clock / power state is shared):
  eager   — bench.py's step: autograd Functions, six launches per step issued from Python;
  graph   — the same step captured once in a hipGraph (torch.cuda.CUDAGraph) and replayed: no Python, no autograd,
            no launch gaps — what is left is the kernels back to back;
  alone   — every kernel of the step timed by itself (its own loop, nothing between its launches), summed.
eager − graph = launch / autograd overhead that shows in the step; graph − alone = what a kernel loses by running behind
the others (the 403 MB of scores / dP written just before it, cold L2s), which no launch mechanism removes.
Reference: the BERT snippet, README.md:62-78 (cublasTransbMM / cublasMM .apply in the attention block).
"""
import argparse
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "matrix-multiplication_amd"))
import custom_mm  # noqa: E402
import matmuls  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--rounds", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
Bz, H, S, D = 32, 12, 512, 64
q, k, v = (torch.rand(Bz, H, S, D, device=dev, generator=g).requires_grad_(True) for _ in range(3))
probs = torch.softmax(torch.rand(Bz, H, S, S, device=dev, generator=g), dim=-1).requires_grad_(True)
d_scores = torch.rand(Bz, H, S, S, device=dev, generator=g)
d_ctx = torch.rand(Bz, H, S, D, device=dev, generator=g)
custom_mm.init_cublas()


def step():
    for t in (q, k, v, probs):
        t.grad = None
    matmuls.cublasTransbMM.apply(q, k).backward(d_scores)
    matmuls.cublasMM.apply(probs, v).backward(d_ctx)


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for _ in range(5):
    step()
torch.cuda.synchronize()
ref = [t.grad.clone() for t in (q, k, v, probs)]

# capture on a side stream, as torch asks
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    step()
graph.replay()
torch.cuda.synchronize()
same = all(torch.equal(t.grad, r) for t, r in zip((q, k, v, probs), ref))

# the six products by themselves, through the same bindings the classes call (forward + the backward products)
q3, k3, v3 = (t.detach().reshape(Bz * H, S, D) for t in (q, k, v))
p3, ds3, dc3 = probs.detach().reshape(Bz * H, S, S), d_scores.reshape(Bz * H, S, S), d_ctx.reshape(Bz * H, S, D)
out_ss, out_sd = torch.empty_like(p3), torch.empty_like(q3)
alone = {}

# matmuls exposes the dense product as cublasMM / cublasTransbMM / cublasTransaMM; time their forwards without autograd
with torch.no_grad():
    alone["q.kT    (k = 64)"] = lambda: matmuls.cublasTransbMM.apply(q3, k3)
    alone["dS.K    (k = 512)"] = lambda: matmuls.cublasMM.apply(ds3, k3)
    alone["dST.Q   (k = 512)"] = lambda: matmuls.cublasTransaMM.apply(ds3, q3)
    alone["P.V     (k = 512)"] = lambda: matmuls.cublasMM.apply(p3, v3)
    alone["dC.VT   (k = 64)"] = lambda: matmuls.cublasTransbMM.apply(dc3, v3)
    alone["PT.dC   (k = 512)"] = lambda: matmuls.cublasTransaMM.apply(p3, dc3)
    for fn in alone.values():
        fn()
torch.cuda.synchronize()

res = {"eager": [], "graph": [], **{n: [] for n in alone}}
for _ in range(a.rounds):
    res["eager"].append(timed(step, a.steps))
    res["graph"].append(timed(graph.replay, a.steps))
    with torch.no_grad():
        for n, fn in alone.items():
            res[n].append(timed(fn, a.steps))
med = {n: float(np.median(x)) for n, x in res.items()}
flops = 6 * 2.0 * Bz * H * S * S * D
print(f"# C5 step on {torch.cuda.get_device_name(0)}: median of {a.rounds} round-robin blocks of {a.steps} steps, ms")
print(f"eager step (bench.py's)            {med['eager']:.4f}   {flops / med['eager'] / 1e9:6.1f} TFLOP/s   {flops / med['eager'] / 1e9 / 157.3:.3f} of the fp32 MFMA peak")
print(f"captured hipGraph, replayed        {med['graph']:.4f}   {flops / med['graph'] / 1e9:6.1f} TFLOP/s   {flops / med['graph'] / 1e9 / 157.3:.3f}   gradients equal the eager step's: {same}")
tot = 0.0
for n in alone:
    print(f"  alone: {n:<22}     {med[n]:.4f}")
    tot += med[n]
print(f"sum of the six products alone      {tot:.4f}   {flops / tot / 1e9:6.1f} TFLOP/s   {flops / tot / 1e9 / 157.3:.3f}   (dS.K and dST.Q are ONE fused launch inside the step)")
print(f"eager - graph = {1e3 * (med['eager'] - med['graph']):.1f} us of launch / autograd overhead in the step; "
      f"graph - alone = {1e3 * (med['graph'] - tot):.1f} us lost by running behind the other products")
