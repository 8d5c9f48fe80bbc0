'''
sharded — row-sharded SpMM across the GPUs of one node (one process per GPU).

New functionality relative to the reference, which is single-device
(`cudaSetDevice(0)`, reference src/sparse_mm.cu:295; no collective anywhere,
SURVEY.md §2.1): C = A·B with A's rows partitioned over the ranks, B replicated,
and C assembled on every rank with an RCCL all-gather over xGMI
(`torch.distributed`, backend "nccl" = RCCL on ROCm).

Rows of C depend only on the same rows of A, so the shards are independent and
each rank runs the same row-split kernel on its rows; per-row arithmetic is
identical to the single-GPU run, so the gathered C is bit-identical to it.

Layout (block-cyclic, so the gather overlaps the compute): the M rows are cut
into `chunks * world` blocks of `block_rows` rows (the tail is padded with empty
rows); rank r owns blocks j*world + r for j = 0 … chunks-1.  Step j computes the
rank's block straight into its final position inside the full C buffer and then
all-gathers blocks [j*world, (j+1)*world) IN PLACE (each rank's input is already
the right slice of the output), on RCCL's stream, while step j+1 computes.
xGMI is point-to-point (7 links per GPU): one shard per peer link per step, no
repacking, `chunks` collectives of M*N*4/chunks bytes each.
'''

import torch
import torch.distributed as dist


def block_layout(M: int, world: int, chunks: int):
    '''(block_rows, padded_rows) of the block-cyclic row layout.'''
    nblocks = world * chunks
    block_rows = max(1, -(-M // nblocks))
    return block_rows, block_rows * nblocks


def owned_blocks(rank: int, world: int, chunks: int):
    return [j * world + rank for j in range(chunks)]


def shard_rowptr(rowptr: torch.Tensor, r0: int, r1: int, M: int) -> torch.Tensor:
    '''Row offsets of rows [r0, r1) rebased to start at 0 (int32, exact integer
    arithmetic); rows at or beyond M are empty (the padded tail).'''
    idx = torch.arange(r0, r1 + 1, device=rowptr.device).clamp_(max=M)
    rp = rowptr.index_select(0, idx).to(torch.int64)
    return (rp - rp[0]).to(torch.int32)


class ShardedSpMM:
    '''C = A·B with A row-sharded over the process group.

    :param rowptr, col, val: the FULL CSR of A (int32 / int32 / float32) on any
        device; only this rank's row blocks are kept (on `device`).
    :param mm_op: 2-d kernel, signature of ``custom_mm.naive_spmm``.
    '''

    def __init__(self, rowptr, col, val, M, K, device, group=None, chunks=4, mm_op=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.M, self.K = int(M), int(K)
        self.chunks = max(1, int(chunks))
        self.device = torch.device(device)
        if mm_op is None:
            import custom_mm
            mm_op = custom_mm.naive_spmm
        self.mm_op = mm_op
        self.block_rows, self.padded_rows = block_layout(self.M, self.world, self.chunks)
        self.blocks = []  # per owned block: (block id, rowptr_local, col, val, nnz)
        rowptr = rowptr.to(torch.int32)
        for blk in owned_blocks(self.rank, self.world, self.chunks):
            r0, r1 = blk * self.block_rows, (blk + 1) * self.block_rows
            p0 = int(rowptr[min(r0, self.M)])
            p1 = int(rowptr[min(r1, self.M)])
            self.blocks.append((blk,
                                shard_rowptr(rowptr, r0, r1, self.M).to(self.device),
                                col[p0:p1].to(torch.int32).to(self.device).contiguous(),
                                val[p0:p1].to(torch.float32).to(self.device).contiguous(),
                                p1 - p0))
        self.local_nnz = sum(b[4] for b in self.blocks)

    def alloc_output(self, N: int) -> torch.Tensor:
        return torch.empty((self.padded_rows, N), device=self.device, dtype=torch.float32)

    def forward(self, B: torch.Tensor, out: torch.Tensor = None, gather: bool = True,
                force_collective: bool = False) -> torch.Tensor:
        '''Returns C [M, N] (a view of the padded buffer), complete on every rank
        when `gather` is true; with gather=False only this rank's blocks are valid.
        `force_collective` issues the all-gather even in a one-rank group (tests).'''
        N = B.shape[1]
        if out is None:
            out = self.alloc_output(N)
        assert out.shape == (self.padded_rows, N) and out.is_contiguous()
        works = []
        br = self.block_rows
        for j, (blk, rp, ci, v, nnz) in enumerate(self.blocks):
            mine = out[blk * br:(blk + 1) * br]
            self.mm_op(v, ci, rp, nnz, br, self.K, B, mine)
            if gather and (self.world > 1 or force_collective):
                span = out[j * self.world * br:(j + 1) * self.world * br]
                # in place: `mine` is span[rank*br : (rank+1)*br]; the collective is
                # ordered after the kernel above and runs beside the next step's kernel
                works.append(dist.all_gather_into_tensor(span, mine, group=self.group, async_op=True))
        for w in works:
            w.wait()
        return out[:self.M]
