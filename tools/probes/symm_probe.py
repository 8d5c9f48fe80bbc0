"""Developer probe: does torch's symmetric memory (peer-mapped buffers across processes) work on this
ROCm build?  Run with: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/probes/symm_probe.py
(both ranks on GPU 0 of a one-GPU box: a functional check of the rendezvous / peer-pointer path only)."""
import os
import sys
import torch
import torch.distributed as dist

rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", int(os.environ.get("PROBE_DEVICE", os.environ.get("LOCAL_RANK", "0")) if os.environ.get("PROBE_SPREAD") else 0))
torch.cuda.set_device(dev)
dist.init_process_group("gloo", rank=rank, world_size=world)
try:
    import torch.distributed._symmetric_memory as symm
    t = symm.empty(1 << 20, dtype=torch.float32, device=dev)
    t.fill_(float(rank + 1))
    hdl = symm.rendezvous(t, dist.group.WORLD.group_name)
    print(f"rank {rank}: rendezvous ok, world {hdl.world_size}, rank {hdl.rank}", flush=True)
    torch.cuda.synchronize()
    dist.barrier()
    peer = (rank + 1) % world
    pb = hdl.get_buffer(peer, (1 << 20,), torch.float32)
    print(f"rank {rank}: peer buffer first value {float(pb[0])} (expect {peer + 1})", flush=True)
    pb[1000:2000].fill_(100.0 + rank)  # write into the peer's memory
    torch.cuda.synchronize()
    dist.barrier()
    print(f"rank {rank}: my buffer after peer write: {float(t[1500])} (expect {100.0 + (rank - 1) % world})", flush=True)
except Exception as e:  # noqa: BLE001
    print(f"rank {rank}: symmetric memory FAILED: {type(e).__name__}: {e}", flush=True)
finally:
    dist.destroy_process_group()
