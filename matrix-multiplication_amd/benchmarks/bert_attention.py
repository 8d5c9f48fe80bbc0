'''
bert_attention — the reference README's BERT drop-in (README.md:62-78) as a runnable benchmark.

In a BERT self-attention block the reference replaces
    attention_scores = torch.matmul(query_layer, key_layer.transpose(-1, -2))
by
    attention_scores = cublasTransbMM.apply(query_layer, key_layer)
and the context product by cublasMM.apply(attention_probs, value_layer) (or, with pruned
probabilities, naiveSpMM.apply).  This script times the whole core (scores → softmax → context),
forward + backward, with the drop-ins and with torch.matmul, at BERT-base shapes, and checks that
outputs and gradients agree (rtol 1e-5 like the reference's tests).

    python matrix-multiplication_amd/benchmarks/bert_attention.py [--batch 32] [--keep 0.1]
'''
import argparse
import json
import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import custom_mm  # noqa: E402
import matmuls  # noqa: E402


def attention(q, k, v, scores_mm, context_mm, mask):
    scores = scores_mm(q, k) / math.sqrt(q.shape[-1])
    probs = torch.softmax(scores, dim=-1)
    if mask is not None:  # magnitude pruning of the attention probabilities
        probs = probs * mask
    return context_mm(probs, v)


def topk_mask(q, k, keep):
    """Top-k-per-row pruning pattern, fixed once (from torch's scores) so that every variant prunes the
    same entries — otherwise last-bit differences in the scores flip entries at the threshold."""
    probs = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(q.shape[-1]), dim=-1)
    kth = max(1, int(probs.shape[-1] * keep))
    return (probs >= probs.topk(kth, dim=-1).values[..., -1:]).to(probs.dtype)


def time_ms(fn, iters=10, warmup=3):
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
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--heads", type=int, default=12)
    ap.add_argument("--seq", type=int, default=512)
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--keep", type=float, nargs="+", default=[1.0, 0.1])
    args = ap.parse_args()
    dev = torch.device("cuda")
    custom_mm.init_cublas()
    custom_mm.init_cusparse()
    g = torch.Generator(device=dev).manual_seed(0)
    shape = (args.batch, args.heads, args.seq, args.dim)
    q0, k0, v0 = (torch.rand(shape, device=dev, generator=g) for _ in range(3))
    dout = torch.rand(shape, device=dev, generator=g)

    def run(scores_mm, context_mm, mask):
        q, k, v = (t.clone().requires_grad_(True) for t in (q0, k0, v0))
        out = attention(q, k, v, scores_mm, context_mm, mask)
        out.backward(dout)
        return out, q.grad, k.grad, v.grad

    torch_scores = lambda a, b: torch.matmul(a, b.transpose(-1, -2))  # noqa: E731
    for keep in args.keep:
        variants = {"torch.matmul": (torch_scores, torch.matmul),
                    "cublasTransbMM + cublasMM": (matmuls.cublasTransbMM.apply, matmuls.cublasMM.apply)}
        if keep < 1.0:
            variants["cublasTransbMM + naiveSpMM"] = (matmuls.cublasTransbMM.apply, matmuls.naiveSpMM.apply)
        mask = topk_mask(q0, k0, keep) if keep < 1.0 else None
        ref = run(torch_scores, torch.matmul, mask)
        rec = {"batch": args.batch, "heads": args.heads, "seq": args.seq, "dim": args.dim, "keep": keep}
        for name, (smm, cmm_) in variants.items():
            got = run(smm, cmm_, mask)
            ok = all(torch.allclose(r, x, rtol=1e-5, atol=1e-6) for r, x in zip(ref, got))
            rec[name + " fwd+bwd ms"] = round(time_ms(lambda: run(smm, cmm_, mask)), 4)
            rec[name + " matches torch"] = bool(ok)
        print(json.dumps(rec), flush=True)
    custom_mm.destroy_cusparse()
    custom_mm.destroy_cublas()


if __name__ == "__main__":
    main()
