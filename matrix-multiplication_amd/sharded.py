'''
sharded — row-sharded SpMM across the GPUs of one node (one process per GPU).

New functionality relative to the reference, which is single-device
(`cudaSetDevice(0)`, reference src/sparse_mm.cu:295; no collective anywhere,
SURVEY.md §2.1): C = A·B with A's rows partitioned over the ranks, B replicated,
and C assembled on every rank with an RCCL all-gather over xGMI
(`torch.distributed`, backend "nccl" = RCCL on ROCm).

Rows of C depend only on the same rows of A, so the shards are independent and
each rank runs the same row-split kernel on its rows.  Per-row arithmetic is made
identical to the single-GPU run by pinning the one thing that depends on the
launch shape — whether rows beyond `custom_mm.long_row_threshold()` non-zeros are
summed in the split order — to what the WHOLE problem's plan does
(`custom_mm.spmm_plan`), not to what a shard's own plan would do; so the gathered
C is bit-identical to the single-GPU result, hub rows included.

Layout (block-cyclic, so the gather overlaps the compute): the M rows are cut
into `chunks * world` blocks; rank r owns blocks j*world + r for
j = 0 … chunks-1.  Step j computes the rank's block straight into its final
position inside the full C buffer and then gathers blocks
[j*world, (j+1)*world) IN PLACE (each rank's input is already the right slice of
the output) on RCCL's stream, while step j+1 computes.  xGMI is point-to-point
(7 links per GPU): one shard per peer link per step, no repacking, `chunks`
collectives of M*N*4/chunks bytes each.

Two ways to cut the rows:
  * split="rows": equal row counts (the tail padded with empty rows) — right for
    uniform patterns; one `all_gather_into_tensor` per step.
  * split="nnz": nnz-balanced split points, block i = rows
    [lower_bound(rowptr, i*nnz/nblocks), lower_bound(rowptr, (i+1)*nnz/nblocks)) —
    for skewed matrices; blocks have different heights, so a step's exchange is one
    in-place broadcast per owner (RCCL runs them back to back on its stream).

Ways to exchange a step's blocks:
  * exchange="allgather": the collective above (RCCL picks its rings / trees), IN PLACE.  Should a torch / RCCL
    build refuse the aliased in-place form — an argument check, raised on every rank alike before anything is
    sent — the operator uses exchange="allgather_copy": the same collective into a scratch span followed by one
    device copy into C (an extra M·N·4/chunks bytes of traffic per step; same bits).  The form is PROBED ONCE at
    construction on a tiny tensor and the outcome agreed over the group with an all-reduce (never caught inside
    forward(): a communicator / transport failure — torch.distributed.DistBackendError — is re-raised, not mistaken
    for a refusal);
  * exchange="alltoall": ONE collective per step as well, but the list form of all_to_all — every rank hands its block
    to every peer and receives each peer's block straight into its final position; RCCL runs it as one group of
    send / recv pairs, i.e. the one-shard-per-link pattern below without leaving collective semantics (every rank
    calls it alike; nothing can half-fail the way independent sends can).  Probed and agreed like the in-place form;
    a build that refuses it (gloo has no list all_to_all) falls back to "allgather";
  * exchange="push": no RCCL data movement at all — every rank maps every peer's C buffer into its own address
    space once (hipIpcGetMemHandle / hipIpcOpenMemHandle behind `custom_mm.ipc_export / ipc_open / ipc_close`, explicit
    lifetimes; the 64 handle bytes travel by `all_gather_object`) and, when a block is computed, COPIES it into its final
    position in every peer's C on a side stream (device-to-device copies over xGMI, one shard per peer link, no
    collective kernel taking CUs from the product).  Two tiny all-reduces on that side stream order it against the
    peers: an ENTRY fence before the first copy (it starts behind this rank's current stream and ends only once every
    rank's has started — so every rank's readers of the PREVIOUS product, and whatever else it had enqueued on C, are
    done before anybody overwrites a peer's rows: no cross-rank write-after-read) and an EXIT fence behind the last
    copy (it ends once every rank's copies are done: all blocks have landed everywhere).
    Only REGISTERED buffers are pushed into: `alloc_output()` / `register_output()` are COLLECTIVE under this exchange
    (every rank calls them at the same point, like a constructor) and keep the buffer alive until `release_peers()`;
    `forward()` without `out` uses one persistent registered buffer per width (the next product overwrites it), and an
    unregistered `out` raises before any collective — there is no per-call, per-rank cache lookup that could send
    some ranks into a mapping collective and others not.  Probed at construction on a tiny buffer and agreed like the
    other forms; refused (CPU tensors, a build without IPC, a mapping some rank cannot open) → "alltoall";
  * exchange="p2p": every rank sends its block straight to every peer and receives each peer's block
    straight into its final position (one grouped batch of isend / irecv per step).  On a fully connected
    xGMI node that uses each of the 7 peer links for exactly one shard at a time — the pattern SURVEY.md
    §8e prices at a third of a ring's time; which one is faster on a given node is decided by timing
    (bench.py tries both before its timed region).  Either split works: blocks of different heights
    are just messages of different sizes.
'''

import numpy as np
import torch
import torch.distributed as dist


def block_layout(M: int, world: int, chunks: int):
    '''(block_rows, padded_rows) of the equal-rows block-cyclic layout.'''
    nblocks = world * chunks
    block_rows = max(1, -(-M // nblocks))
    return block_rows, block_rows * nblocks


def owned_blocks(rank: int, world: int, chunks: int):
    return [j * world + rank for j in range(chunks)]


def balanced_boundaries(rowptr, nblocks: int):
    '''nnz-balanced split points (exact integer arithmetic): boundary i is the first row r
    with rowptr[r] >= i*nnz // nblocks (lower_bound), boundary nblocks is M.  Returns int64[nblocks+1],
    non-decreasing, block i = rows [b[i], b[i+1]).'''
    rp = np.asarray(rowptr.cpu() if isinstance(rowptr, torch.Tensor) else rowptr).astype(np.int64)
    M = len(rp) - 1
    nnz = int(rp[-1])
    targets = (np.arange(nblocks + 1, dtype=np.int64) * nnz) // nblocks
    b = np.searchsorted(rp, targets, side="left").astype(np.int64)
    b[0] = 0
    b[-1] = M
    return np.minimum(np.maximum.accumulate(b), M)


def shard_rowptr(rowptr: torch.Tensor, r0: int, r1: int, M: int) -> torch.Tensor:
    '''Row offsets of rows [r0, r1) rebased to start at 0 (int32, exact integer
    arithmetic); rows at or beyond M are empty (the padded tail).'''
    idx = torch.arange(r0, r1 + 1, device=rowptr.device).clamp_(max=M)
    rp = rowptr.index_select(0, idx).to(torch.int64)
    return (rp - rp[0]).to(torch.int32)


_P2P_PROBED = {}  # (world size, device) -> outcome of probe_p2p in this process


def probe_p2p(device, timeout_s: float = 20.0, _fail_here: bool = False) -> bool:
    '''Can every rank exchange a tiny message with every peer by direct sends (the pattern of exchange="p2p")?
    Tried in its OWN process group with a short timeout, so a send that never completes cannot leave anything queued
    on the group the measured exchange runs on; the outcome is agreed on the default group (all-reduce MIN): True only
    if every rank saw every peer's message.  Never raises for a failed probe — direct sends are optional.
    (`_fail_here`: tests — this rank fails before posting anything.)'''
    import datetime
    world, rank = dist.get_world_size(), dist.get_rank()
    device = torch.device(device)
    key = (world, str(device))
    if key in _P2P_PROBED and not _fail_here:  # one probe (and one extra communicator) per process, not one per operator
        return _P2P_PROBED[key]
    ok = True
    pg = None
    try:
        pg = dist.new_group(timeout=datetime.timedelta(seconds=timeout_s))  # collective: every rank creates it
        if _fail_here:
            raise RuntimeError("probe_p2p: forced failure")
        send = torch.full((4,), float(rank + 1), device=device)
        recv = torch.zeros((world, 4), device=device)
        ops = []
        for r in range(world):
            if r != rank:
                ops.append(dist.P2POp(dist.irecv, recv[r], r, group=pg))
                ops.append(dist.P2POp(dist.isend, send, r, group=pg))
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        if device.type == "cuda":
            torch.cuda.synchronize(device)
        recv[rank] = float(rank + 1)
        expect = torch.arange(1, world + 1, device=device, dtype=torch.float32).unsqueeze(1).expand(world, 4)
        ok = bool(torch.equal(recv, expect))
    except Exception:  # noqa: BLE001 — whatever went wrong, direct sends are off the table
        ok = False
    flag = torch.tensor([1 if ok else 0], device=device, dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if pg is not None:
        try:
            dist.destroy_process_group(pg)  # the probe's communicator is not kept (advisor, round 4)
        except Exception:  # noqa: BLE001 — a group whose exchange hung may refuse; the probe's verdict stands
            pass
    _P2P_PROBED[key] = bool(int(flag[0]))
    return _P2P_PROBED[key]


class ShardedSpMM:
    '''C = A·B with A row-sharded over the process group.  CONSTRUCTION IS COLLECTIVE when the group has more than one
    rank: the chosen exchange is probed on a tiny tensor and the outcome agreed by all-reduce — every rank must build
    the operator, with the same arguments, at the same point.

    :param rowptr, col, val: the FULL CSR of A (int32 / int32 / float32) on any
        device; only this rank's row blocks are kept (on `device`).
    :param split: "rows" (equal row counts) or "nnz" (nnz-balanced split points).
    :param exchange: "allgather" (one collective per step) or "p2p" (direct sends to every peer).
    :param mm_op: 2-d kernel with the signature of ``custom_mm.naive_spmm`` (tests);
        default: ``custom_mm.naive_spmm_ex`` with the long-row rule of the whole problem.
    '''

    def __init__(self, rowptr, col, val, M, K, device, group=None, chunks=4, mm_op=None, split="rows",
                 layout=None, exchange="allgather"):
        if split not in ("rows", "nnz"):
            raise ValueError("split must be 'rows' or 'nnz'")
        if exchange not in ("allgather", "allgather_copy", "alltoall", "p2p", "push"):
            raise ValueError("exchange must be 'allgather', 'allgather_copy', 'alltoall', 'p2p' or 'push'")
        self.exchange = exchange
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if layout is not None:
            # (rank, world) of a rank that is only being MODELLED: its share of the rows without a
            # process group (tools/c4_model.py times one rank's blocks on one GPU); gather must stay off
            self.rank, self.world = int(layout[0]), int(layout[1])
        self.modelled = layout is not None
        self.M, self.K = int(M), int(K)
        self.chunks = max(1, int(chunks))
        self.device = torch.device(device)
        self.split = split
        self.mm_op = mm_op
        self.long_threshold = 8192
        if mm_op is None:
            import custom_mm
            self.long_threshold = custom_mm.long_row_threshold()
        rowptr = rowptr.to(torch.int32).cpu()
        self.nnz = int(rowptr[-1])
        nblocks = self.world * self.chunks
        if split == "rows":
            self.block_rows, self.padded_rows = block_layout(self.M, self.world, self.chunks)
            self.bounds = np.minimum(np.arange(nblocks + 1, dtype=np.int64) * self.block_rows, self.padded_rows)
        else:
            self.bounds = balanced_boundaries(rowptr, nblocks)
            self.block_rows, self.padded_rows = None, self.M
        lens = (rowptr[1:] - rowptr[:-1]) if self.M > 0 else rowptr[:0]
        self.blocks = []  # per owned block: (block id, rowptr_local, col, val, nnz, rows, has_long_rows)
        for blk in owned_blocks(self.rank, self.world, self.chunks):
            r0, r1 = int(self.bounds[blk]), int(self.bounds[blk + 1])
            c0, c1 = min(r0, self.M), min(r1, self.M)
            p0, p1 = int(rowptr[c0]), int(rowptr[c1])
            has_long = bool(c1 > c0 and int(lens[c0:c1].max()) > self.long_threshold)
            self.blocks.append((blk,
                                shard_rowptr(rowptr, r0, r1, self.M).to(self.device),
                                col[p0:p1].to(torch.int32).to(self.device).contiguous(),
                                val[p0:p1].to(torch.float32).to(self.device).contiguous(),
                                p1 - p0, r1 - r0, has_long))
        self.local_nnz = sum(b[4] for b in self.blocks)
        self.local_rows = sum(b[5] for b in self.blocks)
        self._rule = {}  # N -> does the whole problem split long rows
        self._two_streams = {}  # N -> alternate the blocks over two streams?
        self._side = None
        self._scratch = {}     # (step, shape) -> out-of-place gather target (exchange="allgather_copy")
        self._registered = {}  # (data_ptr, shape) of a REGISTERED output buffer -> its registration (exchange="push"); the
                               # registration holds the buffer, so its address cannot be handed to another tensor meanwhile
        self._own_out = {}     # N -> this operator's persistent registered output (forward() without `out` under "push")
        self._serial = 0       # registrations so far (the same on every rank: registering is collective)
        self._push_stream = None
        self._fence = None
        self._tag_check = None  # (pinned [max serial, max -serial] of the last entry fence, event): checked one call later
        self.fallbacks = []    # what was refused and replaced: reported by bench.py
        self._last_refusal = ''
        if self.world > 1 and not self.modelled and dist.is_initialized():
            self._probe_exchange()

    def _agreed(self, ok: bool) -> bool:
        '''True iff EVERY rank of the group reports ok (all-reduce MIN on the group the exchange runs on).'''
        flag = torch.tensor([1 if ok else 0], device=self.device, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(flag[0]))

    def _probe_exchange(self):
        '''Try the chosen collective form once, on a tiny tensor, and agree on the outcome over the group.  A refusal
        (argument validation: RuntimeError / ValueError / TypeError / NotImplementedError raised before anything is
        enqueued, on every rank alike) moves every rank to the next form: alltoall → allgather → allgather_copy.  A
        DistBackendError (communicator, transport) is not a refusal: it is re-raised.'''
        refusals = (RuntimeError, ValueError, TypeError, NotImplementedError)
        if self.exchange == "push":
            # Every rank issues the SAME collectives whatever goes wrong on it (the registration's hand-over and agreement,
            # two more agreements, the release's barrier): a rank that failed early never leaves its peers in a
            # collective it does not enter.
            probe = self._new_buffer(self.world, 4).zero_()
            if self.device.type == "cuda":
                # the zeros must be IN PLACE before any peer may push into this buffer: zero_() is only enqueued, and a peer's
                # copy that lands first would be wiped by it (met once in a 4-rank rehearsal: "pushed probe rows did not
                # arrive").  The registration's hand-over below is the barrier behind every rank's synchronise; forward() has
                # its entry fence for the same purpose.
                torch.cuda.synchronize(self.device)
            ok, why = self.register_output(probe, _quiet=True), ""
            if ok:
                try:
                    reg = self._registered[self._key(probe)]
                    for r in range(self.world):
                        if r != self.rank:
                            self._push_block(reg["views"][r][self.rank], torch.full((4,), float(self.rank + 1), device=self.device))
                    probe[self.rank] = float(self.rank + 1)
                    if self.device.type == "cuda":
                        torch.cuda.synchronize(self.device)
                except Exception as err:  # noqa: BLE001 — IPC is optional: whatever went wrong, the collectives remain
                    if isinstance(err, getattr(dist, "DistBackendError", ())):
                        raise
                    ok, why = False, self._describe(err, "pushing the probe rows")
                ok = self._agreed(ok)  # also the barrier behind every rank's pushes
                if ok:
                    try:
                        expect = torch.arange(1, self.world + 1, device=self.device, dtype=torch.float32)
                        if not torch.equal(probe, expect.unsqueeze(1).expand(self.world, 4)):
                            ok, why = False, f"rank {self.rank}: pushed probe rows did not arrive"
                    except Exception as err:  # noqa: BLE001
                        ok, why = False, self._describe(err, "reading the probe rows")
                    ok = self._agreed(ok)
                self.release_peers()  # peers' mappings closed, a barrier, then the probe buffer may go
                if not ok:
                    self.fallbacks.append(f"push refused ({why or 'on a peer'}): alltoall from now on")
            else:
                self.fallbacks.append(f"push refused ({self._last_refusal}): alltoall from now on")
            del probe
            if not ok:
                self.exchange = "alltoall"
        while self.exchange == "alltoall" or (self.exchange == "allgather" and self.split == "rows"):
            ok, why = True, ""
            try:
                if self.exchange == "alltoall":
                    buf = torch.zeros((self.world, 4), device=self.device, dtype=torch.float32)
                    buf[self.rank] = float(self.rank + 1)
                    empty = buf[self.rank][:0]
                    ins = [buf[self.rank] if r != self.rank else empty for r in range(self.world)]
                    outs = [buf[r] if r != self.rank else empty for r in range(self.world)]
                    dist.all_to_all(outs, ins, group=self.group)
                    expect = torch.arange(1, self.world + 1, device=self.device, dtype=torch.float32)
                    if not torch.equal(buf, expect.unsqueeze(1).expand(self.world, 4)):
                        ok, why = False, "all_to_all delivered the wrong blocks"
                else:
                    # the shapes forward() uses: a [world·rows, N] span and this rank's [rows, N] slice of it
                    span = torch.zeros((self.world * 2, 4), device=self.device, dtype=torch.float32)
                    span[2 * self.rank:2 * self.rank + 2] = float(self.rank + 1)
                    dist.all_gather_into_tensor(span, span[2 * self.rank:2 * self.rank + 2], group=self.group)
                    expect = torch.arange(1, self.world + 1, device=self.device, dtype=torch.float32)
                    if not torch.equal(span, expect.repeat_interleave(2).unsqueeze(1).expand(2 * self.world, 4)):
                        ok, why = False, "in-place all_gather_into_tensor delivered the wrong blocks"
            except refusals as err:
                if isinstance(err, getattr(dist, "DistBackendError", ())):
                    raise
                ok, why = False, self._describe(err, f"probing {self.exchange}")
            if self._agreed(ok):
                return
            nxt = "allgather" if self.exchange == "alltoall" else "allgather_copy"
            self.fallbacks.append(f"{self.exchange} refused on some rank ({why or 'on a peer'}): {nxt} from now on")
            self.exchange = nxt

    def _describe(self, err, doing: str) -> str:
        '''A failure as it goes into `fallbacks`: which rank, doing what, the WHOLE message (a HIP error text was once
        cut at 120 characters and could no longer be told apart from its neighbours).'''
        return f"rank {self.rank} {doing}: {type(err).__name__}: {err}"

    @staticmethod
    def _key(out: torch.Tensor):
        return (out.data_ptr(), tuple(out.shape))

    # -- the three calls the push exchange makes to map a peer's buffer (tests replace them on the CPU) ---------------
    def _export_buffer(self, out: torch.Tensor):
        '''What a peer needs to map `out`: (handle bytes, offset in the allocation); picklable.'''
        if not out.is_cuda:
            raise RuntimeError("push exchange needs device buffers (hipIpc mappings)")
        import custom_mm
        handle, offset, _ = custom_mm.ipc_export(out)
        return (handle, int(offset))

    def _open_peer(self, payload, shape):
        import custom_mm
        return custom_mm.ipc_open(payload[0], payload[1], list(shape), self.device.index or 0)

    def _close_peer(self, payload):
        import custom_mm
        custom_mm.ipc_close(payload[0])

    def _new_buffer(self, rows: int, cols: int) -> torch.Tensor:
        '''Uninitialised [rows, cols] float32 storage for an output (tests hand out shared host memory here).'''
        return torch.empty((rows, cols), device=self.device, dtype=torch.float32)

    def _push_block(self, dst: torch.Tensor, src: torch.Tensor):
        '''One block into its place in one peer's C (a device-to-device copy on the current stream).'''
        dst.copy_(src, non_blocking=True)

    def register_output(self, out: torch.Tensor, _quiet: bool = False) -> bool:
        '''COLLECTIVE (exchange="push"): every rank hands in ITS buffer of the same shape at the same point; each exports its
        buffer, the handles travel by all_gather_object, each maps every peer's buffer, and the outcome is agreed (every
        rank learns every rank's failure text).  True: `out` is registered — forward(out=out) pushes into the peers'
        buffers of THIS registration — and is kept alive by the operator until release_peers().  False (some rank could
        not export or map): every rank has closed what it opened, nothing is registered, the reasons are in
        `fallbacks`.  A communicator failure (DistBackendError) is not a refusal: it propagates.'''
        key = self._key(out)
        if key in self._registered:
            return True
        backend_error = getattr(dist, "DistBackendError", ())
        payload, why = None, ""
        try:
            payload = self._export_buffer(out)
        except Exception as err:  # noqa: BLE001 — told to everybody below, AFTER the collective every peer is waiting in
            if isinstance(err, backend_error):
                raise
            why = self._describe(err, "exporting its buffer")
        gathered = [None] * self.world
        dist.all_gather_object(gathered, (payload, tuple(out.shape)), group=self.group)
        views, opened = [None] * self.world, []
        if not why:
            for r in range(self.world):
                if r == self.rank:
                    views[r] = out
                    continue
                try:
                    if gathered[r][0] is None:
                        raise RuntimeError(f"rank {r} exported nothing")
                    if tuple(gathered[r][1]) != tuple(out.shape):
                        raise RuntimeError(f"rank {r} registers shape {tuple(gathered[r][1])}, this rank {tuple(out.shape)}")
                    views[r] = self._open_peer(gathered[r][0], out.shape)
                    opened.append(gathered[r][0])
                except Exception as err:  # noqa: BLE001
                    if isinstance(err, backend_error):
                        raise
                    why = self._describe(err, f"mapping rank {r}'s buffer")
                    break
        whys = [None] * self.world
        dist.all_gather_object(whys, why, group=self.group)  # the agreement: everybody sees every failure, in full
        if any(whys):
            del views
            for p in opened:
                try:
                    self._close_peer(p)
                except Exception:  # noqa: BLE001 — already refusing; the first failure is the one reported
                    pass
            dist.barrier(group=self.group)  # every peer has closed before any owner frees its buffer
            self._last_refusal = " | ".join(w for w in whys if w)
            if not _quiet:
                self.fallbacks.append(f"push: output buffer not mappable ({self._last_refusal})")
            return False
        self._serial += 1
        tag = torch.tensor([self._serial, -self._serial], device=self.device, dtype=torch.int32)
        self._registered[key] = {"out": out, "views": views, "opened": opened, "serial": self._serial, "tag": tag}
        return True

    def release_peers(self):
        '''COLLECTIVE whenever the operator runs in a process group of more than one rank (the default group included):
        every rank drops its views of its peers' registered buffers and closes the mappings (mi_ipc_close — each process
        closes its own mapping; there is no shared reference counter to race on), then ONE barrier: every mapping of a
        buffer is closed before its owner may free it.  The registrations (and the operator's own persistent outputs) are
        forgotten.  Call it before an operator that has pushed is dropped; bench.py does.  Without a process group, or in a
        one-rank group, nothing was mapped and nothing is issued.'''
        self._check_tags(final=True)
        regs, self._registered, self._own_out = self._registered, {}, {}
        if self.world <= 1 or self.modelled or not dist.is_initialized():
            return
        errors = []
        for reg in regs.values():
            reg["views"] = None
            for p in reg["opened"]:
                try:
                    self._close_peer(p)
                except Exception as err:  # noqa: BLE001 — close the rest, keep the peers in step, report afterwards
                    errors.append(self._describe(err, "closing a peer mapping"))
        dist.barrier(group=self.group)
        del regs
        if errors:
            raise RuntimeError("; ".join(errors))

    def _check_tags(self, final: bool = False):
        '''The entry fence carries (serial, −serial) of the registration this rank pushes into, reduced with MAX: the two
        differ in magnitude iff two ranks handed in buffers of different registrations (a caller bug that would put rows
        into a buffer its owner is not looking at).  Read one call later from pinned memory — never a synchronisation on
        the product path.'''
        if self._tag_check is None:
            return
        host, ev = self._tag_check
        if ev is not None:
            if not final and not ev.query():
                return
            ev.synchronize()
        self._tag_check = None
        if int(host[0]) != -int(host[1]):
            raise RuntimeError(f"push exchange: the ranks pushed into buffers of different registrations (serials "
                               f"{-int(host[1])} … {int(host[0])}): every rank must pass the buffer of the same alloc_output / "
                               f"register_output call")

    def alloc_output(self, N: int) -> torch.Tensor:
        '''A [padded_rows, N] output buffer.  Under exchange="push" (more than one rank) this is COLLECTIVE: the buffer is
        registered with the peers; should that fail on any rank, every rank moves to the collective forms (settled by the
        same probe as at construction) and the buffer is an ordinary one.'''
        out = self._new_buffer(self.padded_rows, N)
        if self.exchange == "push" and self.world > 1 and not self.modelled and dist.is_initialized():
            if not self.register_output(out):
                self.fallbacks.append("alltoall from now on")
                self.exchange = "alltoall"
                self._probe_exchange()  # (collective, on every rank alike) all_to_all itself may be refused
        return out

    def _global_rule(self, B, out):
        '''Whether the single-GPU product of the WHOLE matrix with this B would sum rows beyond the
        threshold in the split order (host-side plan query; cached per N).'''
        N = B.shape[1]
        if N not in self._rule:
            import custom_mm
            self._rule[N] = bool(custom_mm.spmm_plan(self.nnz, self.M, self.K, B, out[:self.M])[3])
        return self._rule[N]

    def _multiply(self, blk, B, mine, out):
        _, rp, ci, v, nnz, rows, has_long = blk
        if rows == 0:
            return
        if self.mm_op is not None:
            self.mm_op(v, ci, rp, nnz, rows, self.K, B, mine)
            return
        import custom_mm
        # no long row in the block → both rules give the same bits: take the one-launch call
        mode = 1 if (has_long and self._global_rule(B, out)) else 0
        custom_mm.naive_spmm_ex(v, ci, rp, nnz, rows, self.K, B, mine, mode)

    def _alternate(self, B, out):
        '''Blocks of a rank are short launches (≈0.26 ms at 8 GPUs): on one stream every block pays its
        own tail, the last workgroups draining while most CUs idle (measured 8.5 % over single-GPU ÷ 8).
        Alternating the blocks over two streams lets block j+1 fill the CUs block j frees.  Only for
        single-launch plans: a multi-pass plan (the two Infinity-Cache panels of large blocks) wants its
        passes alone on the chip.'''
        N = B.shape[1]
        if N not in self._two_streams:
            ok = self.mm_op is None and self.device.type == 'cuda' and len(self.blocks) > 1
            if ok:
                import custom_mm
                blk = max(self.blocks, key=lambda b: b[4])
                rows = max(blk[5], 1)
                ok = custom_mm.spmm_plan(blk[4], rows, self.K, B, out[:rows])[2] == 1
            self._two_streams[N] = ok
        return self._two_streams[N]

    def _src(self, r):
        return dist.get_global_rank(self.group, r) if self.group is not None else r

    def _on_push_stream(self, wait_for_current: bool = False, wait_for_event=None):
        '''The side stream the pushes and their fences run on (CPU tensors, tests: no streams — program order).'''
        import contextlib
        if self.device.type != "cuda":
            return contextlib.nullcontext()
        if wait_for_current:
            self._push_stream.wait_stream(torch.cuda.current_stream(self.device))
        if wait_for_event is not None:
            self._push_stream.wait_event(wait_for_event)
        return torch.cuda.stream(self._push_stream)

    def forward(self, B: torch.Tensor, out: torch.Tensor = None, gather: bool = True,
                force_collective: bool = False, compute: bool = True) -> torch.Tensor:
        '''Returns C [M, N] (a view of the padded buffer), complete on every rank
        when `gather` is true; with gather=False only this rank's blocks are valid.
        `force_collective` issues the collective even in a one-rank group (tests).
        compute=False issues only the exchanges of a step (what is in `out` travels): the gather-only
        leg bench.py times beside compute-only and end-to-end.'''
        N = B.shape[1]
        collective = gather and (self.world > 1 or force_collective)
        if collective and self.modelled:
            raise RuntimeError("a modelled layout has no process group: call forward(..., gather=False)")
        pushing = collective and self.exchange == "push" and self.world > 1
        if out is None:
            if pushing:
                # the operator's persistent registered buffer for this width (first use: collective, on every rank alike);
                # a failed registration has moved every rank to a collective form and the buffer is an ordinary one
                if N not in self._own_out:
                    self._own_out[N] = self.alloc_output(N)
                out = self._own_out[N]
                pushing = self.exchange == "push"
            else:
                out = self._new_buffer(self.padded_rows, N)
        assert out.shape == (self.padded_rows, N) and out.is_contiguous()
        works, copies = [], []
        reg = peers = None
        on_gpu = self.device.type == "cuda"
        if pushing:
            reg = self._registered.get(self._key(out))
            if reg is None:
                raise ValueError("exchange='push' moves blocks into REGISTERED buffers only: take `out` from alloc_output() or "
                                 "pass it to register_output() first (both collective), or call forward() without `out`")
            peers = reg["views"]
            self._check_tags()  # the previous call's entry fence, if it has landed
            if on_gpu and self._push_stream is None:
                self._push_stream = torch.cuda.Stream(self.device)
            if self._fence is None:
                self._fence = torch.zeros(1, device=self.device, dtype=torch.int32)
                self._fence_in = torch.zeros(2, device=self.device, dtype=torch.int32)
                self._fence_host = torch.zeros(2, dtype=torch.int32).pin_memory() if on_gpu else torch.zeros(2, dtype=torch.int32)
            # ENTRY fence — behind everything this rank has enqueued on `out` so far (its readers of the previous product
            # included), ahead of the first copy: it ends only once EVERY rank's has started, so nobody's rows are
            # overwritten while their owner still reads them.  It delays the copies, not the kernels: the first block is
            # still being computed while it runs.
            with self._on_push_stream(wait_for_current=True):
                self._fence_in.copy_(reg["tag"])
                dist.all_reduce(self._fence_in, op=dist.ReduceOp.MAX, group=self.group)
                self._fence_host.copy_(self._fence_in, non_blocking=True)
                ev = None
                if on_gpu:
                    ev = torch.cuda.Event()
                    ev.record()
                self._tag_check = (self._fence_host, ev)
        streams = None
        if compute and self._alternate(B, out):
            main = torch.cuda.current_stream(self.device)
            if self._side is None:
                self._side = torch.cuda.Stream(self.device)
            self._side.wait_stream(main)  # B (and whatever produced it) is ready
            streams = (main, self._side)

        def step(j, blk):
            r0, r1 = int(self.bounds[blk[0]]), int(self.bounds[blk[0] + 1])
            mine = out[r0:r1]
            if compute:
                self._multiply(blk, B, mine, out)
            if not collective:
                return
            first = j * self.world
            if peers is not None:
                if r1 > r0:
                    done = None
                    if on_gpu:
                        done = torch.cuda.Event()
                        done.record()  # the block is computed (current stream)
                    with self._on_push_stream(wait_for_event=done):
                        for k in range(1, self.world):  # every rank starts with a different peer: one shard per link at a time
                            r = (self.rank + k) % self.world
                            self._push_block(peers[r][r0:r1], mine)
                return
            if self.exchange == "p2p":
                ops = []
                for r in range(self.world):
                    if r == self.rank:
                        continue
                    s0, s1 = int(self.bounds[first + r]), int(self.bounds[first + r + 1])
                    if s1 > s0:
                        ops.append(dist.P2POp(dist.irecv, out[s0:s1], self._src(r), group=self.group))
                    if r1 > r0:
                        ops.append(dist.P2POp(dist.isend, mine, self._src(r), group=self.group))
                if ops:  # one group: the sends and receives of a step run side by side, one peer link each
                    works.extend(dist.batch_isend_irecv(ops))
            elif self.exchange == "alltoall":
                # list form: entry r of the input goes to rank r, entry r of the output comes from rank r; the
                # own entry is empty on both sides (the block is already in place).  Blocks may differ in height.
                empty = mine[:0]
                ins = [mine if (r != self.rank and r1 > r0) else empty for r in range(self.world)]
                outs = [out[int(self.bounds[first + r]):int(self.bounds[first + r + 1])] if r != self.rank else empty
                        for r in range(self.world)]
                works.append(dist.all_to_all(outs, ins, group=self.group, async_op=True))
            elif self.split == "rows":
                span = out[int(self.bounds[first]):int(self.bounds[first + self.world])]
                if self.exchange == "allgather":
                    # in place: `mine` is span[rank*br : (rank+1)*br]; the collective is ordered after
                    # the kernel above (same stream) and runs beside the next step's kernel.  (The aliased form
                    # was probed at construction; an error here is a real failure and propagates.)
                    works.append(dist.all_gather_into_tensor(span, mine, group=self.group, async_op=True))
                else:
                    # one scratch span per step (together: a second C), copied into place after the waits
                    key = (j, tuple(span.shape))
                    if key not in self._scratch:
                        self._scratch[key] = torch.empty_like(span)
                    works.append(dist.all_gather_into_tensor(self._scratch[key], mine.contiguous(), group=self.group,
                                                             async_op=True))
                    copies.append((span, self._scratch[key]))
            else:
                for r in range(self.world):  # blocks of different heights: one in-place broadcast per owner
                    s0, s1 = int(self.bounds[first + r]), int(self.bounds[first + r + 1])
                    if s1 > s0:
                        works.append(dist.broadcast(out[s0:s1], src=self._src(r), group=self.group, async_op=True))

        for j, blk in enumerate(self.blocks):
            if streams is None:
                step(j, blk)
            else:
                with torch.cuda.stream(streams[j % 2]):
                    step(j, blk)
        if streams is not None:
            streams[0].wait_stream(streams[1])
        if peers is not None:
            # EXIT fence: a tiny all-reduce BEHIND this rank's copies on the push stream — it ends only once every
            # rank's has started, i.e. once every rank's pushes are done: all blocks have landed in this rank's C
            with self._on_push_stream():
                dist.all_reduce(self._fence, op=dist.ReduceOp.MAX, group=self.group)  # (zeros stay zeros)
            if on_gpu:
                torch.cuda.current_stream(self.device).wait_stream(self._push_stream)
        for w in works:
            w.wait()
        for span, scratch in copies:
            span.copy_(scratch)
        return out[:self.M]
