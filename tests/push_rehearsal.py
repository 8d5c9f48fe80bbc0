"""Helper of tests/test_gpu_sharded.py (not a test module): ONE rank of a multi-process run of the IPC push exchange with
every rank on GPU 0 and gloo as the control plane (RCCL refuses two ranks on one device) — a functional rehearsal of
sharded.ShardedSpMM(exchange="push") on the one-GPU box: real hipIpc mappings between real processes, real side streams.

    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tests/push_rehearsal.py OUT.json
        [--no-entry-fence]     (developer switch: drops the entry fence to show that the reader check below sees the hazard)

What it checks (every rank; rank 0 writes OUT.json):
  * the product gathered by pushes equals the single-GPU product of the whole matrix bit for bit;
  * a READER of C enqueued between two products (behind a long spin kernel on the rank's stream, so it is still pending
    when the next forward() is called) sees product 1 everywhere — the cross-rank write-after-read hazard the entry fence
    closes (round-5 review, weak 3);
  * a second product maps nothing new; a caller's registered buffer works; an unregistered one is refused;
  * after release_peers() no peer mapping is left open in any process.
"""
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

REPO = Path(__file__).resolve().parent.parent
for _p in (str(REPO), str(REPO / "matrix-multiplication_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def main():
    out_path = sys.argv[1]
    no_entry_fence = "--no-entry-fence" in sys.argv
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import custom_mm
    import sharded
    import synthetic

    if no_entry_fence:
        real = dist.all_reduce

        def all_reduce(t, *a, **k):
            if t.numel() == 2 and t.dtype == torch.int32:  # the entry fence's (serial, -serial)
                return None
            return real(t, *a, **k)
        dist.all_reduce = all_reduce

    M, K, N = 40_000, 30_000, 256
    rowptr, col, val = synthetic.make_csr(M, K, 2e-3, seed=5)
    nnz = len(val)
    d_rp, d_col, d_val = (torch.from_numpy(x).to(dev) for x in (rowptr, col, val))
    B1 = torch.from_numpy(synthetic.make_dense(K, N, seed=6)).to(dev)
    B2 = torch.from_numpy(synthetic.make_dense(K, N, seed=7)).to(dev)
    single1, single2 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    custom_mm.naive_spmm(d_val, d_col, d_rp, nnz, M, K, B1, single1)
    custom_mm.naive_spmm(d_val, d_col, d_rp, nnz, M, K, B2, single2)

    res = {"world": world, "no_entry_fence": no_entry_fence}
    op = sharded.ShardedSpMM(torch.from_numpy(rowptr), d_col, d_val, M, K, dev, chunks=2, exchange="push")
    res["exchange"], res["fallbacks"] = op.exchange, list(op.fallbacks)
    assert op.exchange == "push", op.fallbacks
    res["open_after_probe"] = custom_mm.ipc_open_count()

    C = op.forward(B1)
    torch.cuda.synchronize()
    res["first_product_bit_exact"] = bool(torch.equal(C, single1))
    res["open_after_first_product"] = custom_mm.ipc_open_count()

    # the reader between two products: still pending (behind tens of milliseconds of spinning) when the next forward() is issued
    hazard_seen = []
    for trial in range(3):
        op.forward(B1)
        torch.cuda._sleep(40_000_000 if rank % 2 == 0 else 2_000_000)   # uneven: the peers are ready long before
        snap = C.clone()                                                   # the reader of product 1
        C2 = op.forward(B2)                                                # product 2 into the same registered buffer
        torch.cuda.synchronize()
        hazard_seen.append(not bool(torch.equal(snap, single1)))
        assert C2.data_ptr() == C.data_ptr()
        res["second_product_bit_exact"] = bool(torch.equal(C2, single2))
    res["reader_saw_the_next_product"] = hazard_seen
    res["open_after_more_products"] = custom_mm.ipc_open_count()

    mine = op.alloc_output(N)    # collective: registered with the peers
    res["callers_buffer_bit_exact"] = bool(torch.equal(op.forward(B1, out=mine), single1))
    try:
        op.forward(B1, out=torch.empty_like(mine))
        res["unregistered_refused"] = False
    except ValueError:
        res["unregistered_refused"] = True
    res["fallbacks_end"] = list(op.fallbacks)
    op.release_peers()
    res["open_after_release"] = custom_mm.ipc_open_count()
    del mine, C
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    if rank == 0:
        Path(out_path).write_text(json.dumps(gathered))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
