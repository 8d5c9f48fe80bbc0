'''
matmuls — autograd wrappers and shape dispatch over the `custom_mm` kernels.

Mirror of the reference's matmuls.py (smoorjani/matrix-multiplication
matmuls.py:1-327): same public names (`custom_matmul`, `sparse_matmul`,
`naive_matmul`, `get_sparse_tensor_properties`, `cublasMM`, `cublasTransaMM`,
`cublasTransbMM`, `cublasTransabMM`, `cusparseMM`, `naiveSpMM`), same
`.apply(m1, m2)` usage, same `mm_op` / `bmm_op` injection points.  Underneath,
`custom_mm` is the MI355X build (hand-written HIP kernels behind
include/mi_spmm.h); there is no CPU path here — CPU tensors reach `custom_mm`
and raise.

Where the reference's behaviour is a defect, this file implements the intended
math = `torch.matmul` and its autograd (SURVEY.md §8a "known defects"):
  * outputs are allocated on the inputs' device, not hard-coded 'cuda'
    (reference matmuls.py:36-37,205-206,274-275);
  * 2-D×3-D, 3-D×2-D and ≥5-D inputs follow torch.matmul broadcasting
    (reference :48-52,62-63,213-220,230-234,282-288);
  * backward uses the correct formulas for the Trans* variants and never hands
    a strided view to a kernel as if it were contiguous (reference :96-102,
    :120-126,:144-150,:168-174,:250-254,:319-325);
  * batched sparse products run as ONE launch over a batched CSR instead of a
    Python recursion with one to_sparse_csr() per slice (reference :289-297).
'''

import os
import weakref

import torch
from torch.autograd.function import InplaceFunction
import custom_mm


# --------------------------------------------------------------------------- #
# helpers
# --------------------------------------------------------------------------- #

def _out_shape(a_shape, b_shape, transa, transb):
    c_rows = a_shape[-2] if not transa else a_shape[-1]
    c_cols = b_shape[-1] if not transb else b_shape[-2]
    batch = torch.broadcast_shapes(tuple(a_shape[:-2]), tuple(b_shape[:-2]))
    return tuple(batch), c_rows, c_cols


def _sum_to_shape(grad, shape):
    '''Reduce a broadcast gradient back to the shape of the input it belongs to.'''
    shape = tuple(shape)
    if tuple(grad.shape) == shape:
        return grad
    lead = grad.dim() - len(shape)
    if lead > 0:
        grad = grad.sum(dim=tuple(range(lead)))
    dims = tuple(i for i, (g, s) in enumerate(zip(grad.shape, shape)) if s == 1 and g != 1)
    if dims:
        grad = grad.sum(dim=dims, keepdim=True)
    return grad


# the HIP-backed extension (tests/fake_custom_mm.py — the oracle-backed stand-in the host-logic tests import instead — is a
# plain Python module: with it every call goes through the dispatch code, which is what those tests are for)
_REAL_EXTENSION = str(getattr(custom_mm, '__file__', '')).endswith('.so')


def _host_operands(a: torch.Tensor, b: torch.Tensor) -> bool:
    '''Two dense float tensors in host memory, with the real extension loaded.'''
    return (_REAL_EXTENSION and not a.is_cuda and not b.is_cuda and a.layout == torch.strided and b.layout == torch.strided
            and a.dim() >= 1 and b.dim() >= 1)


def custom_matmul(a: torch.Tensor,
                  b: torch.Tensor,
                  mm_op=custom_mm.cublas_mmul,
                  bmm_op=custom_mm.cublas_bmm,
                  transa=False,
                  transb=False) -> torch.Tensor:
    '''
    Uses ``mm_op`` or ``bmm_op`` kernel to perform matrix multiplication,
    ``op(a) @ op(b)`` with ``op`` = transpose of the last two dims when the
    flag is set (reference matmuls.py:13-72).

    :param a:
    :param b:
    :param mm_op: kernel to perform basic matrix multiplication,
                  ``mm_op(A, B, C, transa, transb) -> C``
    :param bmm_op: kernel to perform batched matrix multiplication,
                  ``bmm_op(A, B, C, dim, transa, transb) -> C`` with dim 3 or 4
    :param transa: transpose A
    :param transb: transpose B
    :returns: Matrix multiplication output
    '''
    if _host_operands(a, b):
        # BASELINE.json configs[0] as written ("torch.mm dense 8×64 @ 64×8 on CPU via the matmuls.py wrapper"): both operands are
        # dense HOST tensors, nothing asks for the GPU — the reference's own expression for the ranks its kernels do not take,
        # `return a @ b` (reference matmuls.py:39-41).  torch, not a kernel of this package and not the oracle; a device operand
        # (or a mixed pair, or a CSR operand) never comes here: custom_mm raises on host tensors.
        return (a.transpose(-1, -2) if transa and a.dim() > 1 else a) @ (b.transpose(-1, -2) if transb and b.dim() > 1 else b)
    # matrix-vector forms: promote the vector to a matrix, as torch.matmul does.
    if a.dim() == 1 or b.dim() == 1:
        if a.dim() == 0 or b.dim() == 0:
            raise ValueError('custom_matmul: both arguments need to be at least 1-d')
        a2 = a.unsqueeze(0) if a.dim() == 1 else a
        b2 = b.unsqueeze(-1) if b.dim() == 1 else b
        c = custom_matmul(a2, b2, mm_op, bmm_op,
                          transa if a.dim() > 1 else False,
                          transb if b.dim() > 1 else False)
        if b.dim() == 1:
            c = c.squeeze(-1)
        if a.dim() == 1:
            c = c.squeeze(-2 if b.dim() > 1 else -1)
        return c

    batch, c_rows, c_cols = _out_shape(a.shape, b.shape, transa, transb)
    # create tensor C to store results in (beta = 0: no need to pre-zero it)
    c = torch.empty(batch + (c_rows, c_cols), device=a.device, dtype=torch.float32)

    if len(batch) == 0:
        return mm_op(a, b, c, transa, transb)

    if a.dim() >= 3 and b.dim() == 2 and not transa:
        # flatten A into a 2d tensor: one large product instead of a batch
        _a = a.reshape(-1, a.shape[-1])
        mm_op(_a, b, c.view(-1, c_cols), transa, transb)
        return c

    # batched: broadcast both operands to the common batch shape (stride-0 views)
    _a = a.expand(batch + tuple(a.shape[-2:]))
    _b = b.expand(batch + tuple(b.shape[-2:]))
    if len(batch) == 1:
        return bmm_op(_a, _b, c, 3, transa, transb)
    if len(batch) == 2:
        return bmm_op(_a, _b, c, 4, transa, transb)
    # 5-d and larger: fold the batch dims into one (copies only if a view cannot)
    _a = _a.reshape((-1,) + tuple(_a.shape[-2:]))
    _b = _b.reshape((-1,) + tuple(_b.shape[-2:]))
    bmm_op(_a, _b, c.view((-1, c_rows, c_cols)), 3, transa, transb)
    return c


'''
Matrix multiplication classes

Ensure the forward and backward passes are defined for torch.autograd
To add another, just change the mm/bmm operation
'''


_fused_pair = os.environ.get('MI_GEMM_PAIR', '1') != '0'  # 0: the two plain products (developer A/B)


def _dense_backward(ctx, grad_output, transa, transb):
    '''Gradients of C = op(m1)·op(m2) (= torch autograd of torch.matmul).'''
    m1, m2 = ctx.saved_tensors
    grad_m1 = grad_m2 = None
    v1, v2 = m1.dim() == 1, m2.dim() == 1
    a = m1.unsqueeze(0) if v1 else m1
    b = m2.unsqueeze(-1) if v2 else m2
    ta = transa and not v1
    tb = transb and not v2
    g = grad_output
    if v2:
        g = g.unsqueeze(-1)
    if v1:
        g = g.unsqueeze(-2)

    if (_fused_pair and not ta and tb and ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and a.dim() >= 3 and
            tuple(a.shape[:-2]) == tuple(b.shape[:-2]) == tuple(g.shape[:-2]) and hasattr(custom_mm, 'cublas_bmm_pair')):
        # C = A·Bᵀ (the BERT drop-in scores = cublasTransbMM.apply(q, k), README.md:69-77): dA = dC·B and dB = dCᵀ·A both
        # stream dC — one fused launch reads it once (custom_mm.cublas_bmm_pair; same bits as the two plain products,
        # False when the shapes are not its)
        gc, ac, bc = g.contiguous(), a.contiguous(), b.contiguous()
        ga, gb = torch.empty_like(ac), torch.empty_like(bc)
        if custom_mm.cublas_bmm_pair(gc, bc, ac, ga, gb):
            return (ga.squeeze(0) if v1 else ga), (gb.squeeze(-1) if v2 else gb)

    if ctx.needs_input_grad[0]:
        if not ta and not tb:      # dA = dC·Bᵀ
            ga = custom_matmul(g, b, transb=True)
        elif ta and not tb:        # C = Aᵀ·B:  dA = B·dCᵀ
            ga = custom_matmul(b, g, transb=True)
        elif not ta and tb:        # C = A·Bᵀ:  dA = dC·B
            ga = custom_matmul(g, b)
        else:                      # C = Aᵀ·Bᵀ: dA = Bᵀ·dCᵀ
            ga = custom_matmul(b, g, transa=True, transb=True)
        ga = _sum_to_shape(ga, a.shape)
        grad_m1 = ga.squeeze(0) if v1 else ga

    if ctx.needs_input_grad[1]:
        if not ta and not tb:      # dB = Aᵀ·dC
            gb = custom_matmul(a, g, transa=True)
        elif ta and not tb:        # C = Aᵀ·B:  dB = A·dC
            gb = custom_matmul(a, g)
        elif not ta and tb:        # C = A·Bᵀ:  dB = dCᵀ·A
            gb = custom_matmul(g, a, transa=True)
        else:                      # C = Aᵀ·Bᵀ: dB = dCᵀ·Aᵀ
            gb = custom_matmul(g, a, transa=True, transb=True)
        gb = _sum_to_shape(gb, b.shape)
        grad_m2 = gb.squeeze(-1) if v2 else gb

    return grad_m1, grad_m2


class cublasMM(InplaceFunction):
    @staticmethod
    def forward(ctx, m1, m2):
        ctx.save_for_backward(m1, m2)
        return custom_matmul(
            m1, m2)

    @staticmethod
    def backward(ctx, grad_output):
        return _dense_backward(ctx, grad_output, False, False)


class cublasTransaMM(InplaceFunction):
    @staticmethod
    def forward(ctx, m1, m2):
        ctx.save_for_backward(m1, m2)
        return custom_matmul(
            m1, m2, transa=True)

    @staticmethod
    def backward(ctx, grad_output):
        return _dense_backward(ctx, grad_output, True, False)


class cublasTransbMM(InplaceFunction):
    @staticmethod
    def forward(ctx, m1, m2):
        ctx.save_for_backward(m1, m2)
        return custom_matmul(
            m1, m2, transb=True)

    @staticmethod
    def backward(ctx, grad_output):
        return _dense_backward(ctx, grad_output, False, True)


class cublasTransabMM(InplaceFunction):
    @staticmethod
    def forward(ctx, m1, m2):
        ctx.save_for_backward(m1, m2)
        return custom_matmul(
            m1, m2, transa=True, transb=True)

    @staticmethod
    def backward(ctx, grad_output):
        return _dense_backward(ctx, grad_output, True, True)


def get_sparse_tensor_properties(a: torch.Tensor):
    '''
    Retrieve properties of CSR tensor (reference matmuls.py:178-187).
    :param a: CSR Tensor
    :returns: values, col indices (int32), row offsets (int32), number of
              nonzeros, and shape of a — the argument order of
              ``custom_mm.naive_spmm`` / ``custom_mm.cusparse_mmul``; all on
              the device (a CPU CSR tensor is moved, as the reference's
              ``.cuda()`` does); the int64→int32 narrowing happens on the
              device, without the reference's host round trip.
    '''
    assert a.is_sparse_csr
    values = torch.Tensor.values(a)
    nnz = values.numel()
    if nnz >= 2 ** 31:
        raise ValueError('get_sparse_tensor_properties: nnz does not fit int32 indices')
    if not values.is_cuda and torch.cuda.is_available():
        a = a.cuda()  # as the reference's `.cuda()`; without a GPU the kernel call raises
        values = torch.Tensor.values(a)
    return values.contiguous(), torch.Tensor.col_indices(a).to(torch.int32).contiguous(), \
        torch.Tensor.crow_indices(a).to(torch.int32).contiguous(), nnz, \
        a.shape[-2], a.shape[-1]


def _csr_key(a: torch.Tensor):
    vals, crow, ccol = torch.Tensor.values(a), torch.Tensor.crow_indices(a), torch.Tensor.col_indices(a)
    return (vals.data_ptr(), crow.data_ptr(), ccol.data_ptr(), vals._version, crow._version, ccol._version,
            tuple(a.shape), vals.numel())


def _csr_props_cached(a: torch.Tensor):
    '''(values, columns i32, offsets i32, nnz, rows, cols) of a CSR tensor, kept ON the tensor object between
    calls (keyed on the component tensors' storage and version counters; it dies with the tensor): a static sparse
    operand is narrowed to int32 once instead of on every product (two passes over the indices: 0.3 ms at the
    1M × 1M config).  Nothing is read back to the host (round 2 read the longest row back to decide whether the
    long-row helpers were needed: since round 3 the plain entry points cost the main kernel + one empty follow-up
    launch, so the question no longer pays for a synchronisation — and the call stays graph-capturable).'''
    key = _csr_key(a)
    hit = getattr(a, '_mi_csr_props', None)
    if hit is not None and hit[0] == key:
        return hit[1]
    props = get_sparse_tensor_properties(a)
    try:
        a._mi_csr_props = (key, props)
    except (AttributeError, RuntimeError):
        pass  # a tensor type that takes no attributes: just no caching
    return props


def _row_schedule(holder: torch.Tensor, slot: str, key, offsets: torch.Tensor, nnz: int, rows: int, width: int, columns=None, cols: int = 0):
    '''The inspector's row schedule (custom_mm.spmm_schedule: rows handed to waves longest first, like lengths together,
    heavy rows in a launch of their own — the same bits as the plain product) for a CSR pattern that is being used AGAIN:
    kept on the tensor object `holder` under `slot`, per dense width, keyed like the other per-tensor caches (index
    tensors' storage and version counters).  The first product of a pattern runs plain and only leaves a mark — building
    a schedule reads 1 KiB back (it synchronises), which a one-shot operand should not pay for; from the second product
    on the schedule is there.  Never built under stream capture.  Returns None when there is none (yet), or when the
    inspector found no skew worth an indirection (short, alike rows).  Counterpart of the reference's inspect-once /
    multiply-many pair (src/sparse_mm.cu:137-385), without asking the caller to name a layer.'''
    if not hasattr(custom_mm, 'spmm_schedule') or not offsets.is_cuda or rows < 2 or nnz < 4096:
        return None
    book = getattr(holder, slot, None)
    if book is None or book[0] != key:
        book = (key, {})
        try:
            setattr(holder, slot, book)
        except (AttributeError, RuntimeError):
            return None  # a tensor type that takes no attributes: no schedule
    ent = book[1].get(width)
    if ent is None:
        book[1][width] = 'seen'
        return None
    if ent == 'seen':
        if torch.cuda.is_current_stream_capturing():
            return None
        ent = custom_mm.spmm_schedule(offsets, nnz, rows, width, columns, cols)
        book[1][width] = ent
    return ent if ent.info()['active'] else None


def _dense_to_csr(a: torch.Tensor, est_density=None):
    '''(values, columns, offsets, nnz) of a dense tensor's last two dims (batched: the "rowptr of rowptrs" layout).
    With neither capture nor an estimate: the exact arrays (one read-back of the count sizes them; the kernels' plan
    choice sees the true number of non-zeros).  Under capture nothing may be read back, and with `est_density` (a sampled
    share of non-zeros, see sampled_density) nothing needs to be: the arrays get room for EVERY element (capacity =
    a.numel(): the fill cannot overflow) and the product kernels walk the rows through `offsets`.  The count they are told
    is then the capacity (capture) — an upper bound no larger than the arrays, which is what include/mi_spmm.h asks of it
    — or the estimate, which only steers the plan and is allowed where no long-row workspace is sized from it (callers
    pass an estimate only on those routes: rule 0 of naive_spmm_ex, the batched entry); either way the same bits.'''
    if a.is_cuda and torch.cuda.is_current_stream_capturing():
        offsets = custom_mm.dense_row_offsets(a)
        values, columns = custom_mm.dense_to_csr_fill(a, offsets, a.numel())
        return values, columns, offsets, a.numel()
    if est_density is not None and a.is_cuda and 0 < a.numel() <= _NO_READBACK_MAX_ELEMS and hasattr(custom_mm, 'dense_row_offsets'):
        offsets = custom_mm.dense_row_offsets(a)
        values, columns = custom_mm.dense_to_csr_fill(a, offsets, a.numel())
        return values, columns, offsets, min(a.numel(), max(1, int(est_density * a.numel())))
    values, columns, offsets = custom_mm.dense_to_csr(a)
    return values, columns, offsets, values.numel()


# capacity-sized CSR arrays are 8 B per ELEMENT of A, whatever its density (a 1 % dense operand: 100 × what the exact arrays
# need) — transient, but real: the no-read-back route is for operands up to 128 Mi elements (≤ 1 GiB of arrays on a 288 GB
# part; BERT's 384 × 512² probabilities are 100 Mi), larger ones pay the one read-back of the count (round-5 advisor: was 256 Mi)
_NO_READBACK_MAX_ELEMS = 1 << 27


def _csr_of(a: torch.Tensor, est_density=None):
    '''(values, columns, offsets, nnz, rows, cols) of a 2-d dense or CSR tensor.'''
    if a.is_sparse_csr:
        return _csr_props_cached(a)
    values, columns, offsets, nnz = _dense_to_csr(a, est_density)
    return values, columns, offsets.view(-1), nnz, a.shape[-2], a.shape[-1]


def _csr_product(a: torch.Tensor, b: torch.Tensor, c: torch.Tensor, mm_op, default_op, owner=None):
    '''c = a·b for a 2-d CSR (or dense, converted) a through mm_op.  A matrix no wider than
    custom_mm.long_row_threshold() columns cannot hold an over-long row (columns are not repeated inside a row of a
    torch CSR or of a converted dense matrix), so the stock kernel then runs as ONE launch
    (custom_mm.naive_spmm_ex, rule 0: same bits); wider ones take the plain entry, which is the main kernel plus
    one follow-up launch that finds its list of long rows empty.  No host read-back either way for a CSR tensor; a DENSE
    a on the one-launch route is converted without one too (round 5): the count the kernel is told is the sampled
    density's (sampled_density — a count kernel whose result comes back behind an event, only the first call of a shape
    waits), the arrays have room for every element.  The reference converts with `to_sparse_csr()` on every call, which
    synchronises every time (matmuls.py:295-296).'''
    one_launch = mm_op is default_op and hasattr(custom_mm, 'naive_spmm_ex') and a.shape[-1] <= custom_mm.long_row_threshold()
    est = None
    if one_launch and not a.is_sparse_csr and a.is_cuda and a.numel() > 0 and not torch.cuda.is_current_stream_capturing():
        est = sampled_density(a, (tuple(a.shape), tuple(b.shape), a.device.index), a.shape[-1], owner)
    props = _csr_of(a, est)
    # an ESTIMATED count may only meet rule 0 (no long-row workspace is sized from it: include/mi_spmm.h, K1 section)
    assert est is None or one_launch
    if a.is_sparse_csr and mm_op is default_op and b.dim() == 2:
        # a CSR tensor that has been multiplied before (a static sparse operand: weights, an adjacency matrix): its row schedule
        sched = _row_schedule(a, '_mi_csr_sched', _csr_key(a)[1:3] + _csr_key(a)[4:], props[2], props[3], props[4], b.shape[-1], props[1], props[5])
        if sched is not None:
            return custom_mm.naive_spmm_scheduled(sched, *props, b, c, None, 0 if one_launch else -1)
    if one_launch:
        return custom_mm.naive_spmm_ex(*props, b, c, 0)
    return mm_op(*props, b, c)


def fused_skip_pays(items: int, rows: int, cols: int, width: int = 256) -> bool:
    '''Dense A with zeros: the kernel that skips them in place (no CSR, no host read-back, one launch per 256
    output columns) against dense→CSR + the CSR kernels.  Measured on MI355X (tools/bench_skipwide.py,
    benchmarks/random_tensor_benchmark.py): the in-place kernel gives a wave one row to scan, so it wins where
    launches and the read-back dominate or where the batched CSR kernel is the alternative — one small matrix
    (512² at 10 %: 0.013 vs 0.054 ms), batches of small and medium matrices (BERT's 384 × 512²: 0.20 vs 0.32 ms at
    10 % kept, 0.91 vs 1.35 fully dense; 16 × 2048²: 0.19 vs 0.28) — and loses where one large matrix meets the
    single-matrix CSR plans (2048² 0.145 vs 0.067 ms, 16384 × 768 0.39 vs 0.13, 4096² 0.50 vs 0.13), where the rows
    are very long (4 × 4096²: 0.59 vs 0.39), and on a single matrix the denser and wider it gets: what it can win
    there is a launch and a read-back (≈40 µs), what it can lose is unbounded (1024² fully dense × 1024 columns:
    1.2 vs 0.25 ms) — so a single matrix takes it only when tiny.'''
    if items <= 1:
        return rows * cols * -(-width // 256) <= 512 * 512
    return cols <= 3072


def _batched_csr_pattern(a: torch.Tensor, dev, transposed: bool = False):
    '''The index side of a batched CSR tensor ([..., M, K], equal non-zero counts per item) as the batched kernels
    want it, kept ON the tensor object between calls (keyed on the index tensors' storage and version counters):
    `offsets` int32 [batch, M + 1] with every item's base added ("rowptr of rowptrs"), `columns` int32 [nnz]; with
    `transposed`, also what the backward needs — the flat offsets [batch·M + 1] and block-diagonal columns (item i
    shifted by i·K) of the whole batch as ONE matrix, and the pattern of every item's transpose with the permutation
    that carries the values into it (one batched device transpose of 0, 1, 2, …: the kernels move 4-byte values
    untouched).  A training loop with a static pattern pays the narrowing and the transpose once; values are never
    cached (see _csr_cached).'''
    rows, cols = a.shape[-2], a.shape[-1]
    crow = torch.Tensor.crow_indices(a).reshape(-1, rows + 1)
    col = torch.Tensor.col_indices(a)
    nb, per_item = crow.shape[0], col.shape[-1]
    total = nb * per_item
    key = (crow.data_ptr(), col.data_ptr(), crow._version, col._version, tuple(a.shape), per_item, str(dev))
    hit = getattr(a, '_mi_batched_pattern', None)
    if hit is None or hit[0] != key:
        if total >= 2 ** 31 or nb * max(rows, cols) >= 2 ** 31:
            raise ValueError('sparse matmul: the batch holds too many non-zeros / rows for int32 indices')
        if (hasattr(custom_mm, 'batched_csr_narrow') and crow.is_cuda and crow.device == torch.device(dev) and per_item > 0
                and crow.dtype == torch.int64 and col.dtype == torch.int64):
            # ONE launch for both index tensors (round 5: attention probabilities are a new pattern on every step, so
            # this narrowing is per-step work, not a one-off — it used to be five torch kernels)
            off32, col32 = custom_mm.batched_csr_narrow(crow.contiguous(), col.reshape(nb, per_item).contiguous())
            hit = [key, off32, col32, None]
        else:
            base = torch.arange(nb, device=crow.device, dtype=crow.dtype).unsqueeze(1) * per_item
            hit = [key, (crow + base).to(device=dev, dtype=torch.int32).contiguous(),
                   col.reshape(-1).to(device=dev, dtype=torch.int32).contiguous(), None]
        try:
            a._mi_batched_pattern = hit
        except (AttributeError, RuntimeError):
            pass  # a tensor type that takes no attributes: just no caching
    if transposed and hit[3] is None:
        offsets, columns = hit[1], hit[2]
        flat_off = torch.cat([offsets[:, :-1].reshape(-1), offsets[-1:, -1]]).contiguous()
        shift = (torch.arange(nb, device=dev, dtype=torch.int32) * cols).repeat_interleave(per_item)
        iota = torch.arange(total, device=dev, dtype=torch.int32).view(torch.float32)
        if nb <= 65535:
            t_perm, t_col, t_off = custom_mm.csr_transpose_batched(iota, columns, offsets, total, nb, rows, cols)
        else:
            # more items than one launch takes: transpose chunks of ≤ 65535 items on their own slices of the arrays
            # (offsets rebased to the slice, then put back) — the values are 0, 1, 2, … of the WHOLE batch, so the
            # permutation stays global
            parts = []
            for lo in range(0, nb, 65535):
                hi = min(nb, lo + 65535)
                p0, p1 = lo * per_item, hi * per_item
                tp, tc, to = custom_mm.csr_transpose_batched(iota[p0:p1], columns[p0:p1], (offsets[lo:hi] - p0).contiguous(),
                                                             p1 - p0, hi - lo, rows, cols)
                parts.append((tp, tc, to + p0))
            t_perm, t_col, t_off = (torch.cat([x[i] for x in parts]) for i in range(3))
        t_perm = t_perm.view(torch.int32)
        hit[3] = (flat_off, columns + shift, t_perm, t_col, t_off)
    return hit[1], hit[2], hit[3]


def _batched_csr_product(a: torch.Tensor, b: torch.Tensor, mm_op, default_op) -> torch.Tensor:
    '''A batched CSR tensor ([..., M, K], every item with the same number of non-zeros — what torch builds) as the
    sparse operand: the reference recurses over the leading dimension (matmuls.py:289-293); here the whole batch is
    ONE launch of the batched kernel — the items' compressed row offsets get their base offset added, which is the
    "rowptr of rowptrs" layout of custom_mm.naive_spmm_batched.  b: [K, N] (shared) or [..., K, N] (same batch).'''
    a_shape, b_shape = a.shape, b.shape
    rows, cols, n = a_shape[-2], a_shape[-1], b_shape[-1]
    batch = tuple(a_shape[:-2])
    if b.dim() > 2 and tuple(b_shape[:-2]) != batch:
        raise RuntimeError('sparse matmul: a batched CSR tensor needs b of shape [K, N] or with the same batch dimensions')
    crow = torch.Tensor.crow_indices(a).reshape(-1, rows + 1)
    col = torch.Tensor.col_indices(a)
    val = torch.Tensor.values(a)
    nb = crow.shape[0]
    per_item = col.shape[-1]
    total = nb * per_item
    if total >= 2 ** 31:
        raise ValueError('sparse matmul: the batch holds too many non-zeros for int32 indices')
    dev = val.device if val.is_cuda else b.device
    _b = (b.reshape(nb, cols, n) if b.dim() > 2 else b).to(dev)  # a 2-d b is shared by every item
    c = torch.empty((nb, rows, n), device=dev, dtype=torch.float32)
    if mm_op is default_op:
        offsets, columns, _ = _batched_csr_pattern(a, dev)
        values = val.reshape(-1).to(dev).contiguous()
        for lo in range(0, nb, 65535):
            hi = min(nb, lo + 65535)
            custom_mm.naive_spmm_batched(values, columns, offsets[lo:hi].contiguous(), total, hi - lo, rows, cols,
                                         _b[lo:hi].contiguous() if _b.dim() > 2 else _b.contiguous(), c[lo:hi])
    else:
        # a caller-supplied 2-d kernel: slice by slice, as the reference does
        for i in range(nb):
            mm_op(val.reshape(nb, -1)[i].to(dev).contiguous(), col.reshape(nb, -1)[i].to(device=dev, dtype=torch.int32).contiguous(),
                  crow[i].to(device=dev, dtype=torch.int32).contiguous(), per_item, rows, cols,
                  (_b[i] if _b.dim() > 2 else _b).contiguous(), c[i])
    return c.view(batch + (rows, n))


_DENSE_SAMPLE_ROWS = 128
_density_of_shape = {}   # (shape of a, shape of b, device) -> {'est', 'n', 'host', 'event', 'src'}: see _dense_route
_density_of_tensor = {}  # id(a) -> (weakref to a, a._version, density) of an `a` whose own sample has landed

# How a DENSE tensor handed to naiveSpMM / cusparseMM (zeros to be skipped) may be multiplied:
#   'auto'   (default) the exact-fp32 MFMA product where it is faster (dense_route_pays), with the zero-skipping
#            semantics of the reference's `a.to_sparse_csr()` (matmuls.py:295-296) GUARANTEED on the device: a gated
#            launch of the zero-skipping kernel recomputes the product iff `b` holds an inf / nan (the only operands
#            on which the two routes differ: 0·inf = nan) — so the route affects time only, never the result;
#   'never'  always a zero-skipping route (in-kernel skip or dense→CSR + CSR kernels);
#   'always' always the MFMA product, torch.matmul semantics (a zero of `a` facing an inf / nan of `b` gives nan —
#            what the reference's tests compare with, tests/naive_kernel_test.py:30).
# Set with MI_DENSE_ROUTE in the environment, matmuls.set_dense_route(mode), or per call:
# naive_matmul(a, b, dense_route='never').
_DENSE_ROUTE_MODES = ('auto', 'never', 'always')
_dense_route_mode = os.environ.get('MI_DENSE_ROUTE', 'auto').strip().lower() or 'auto'
if _dense_route_mode not in _DENSE_ROUTE_MODES:
    raise ValueError(f"MI_DENSE_ROUTE must be one of {_DENSE_ROUTE_MODES}, not {_dense_route_mode!r}")


def set_dense_route(mode: str) -> str:
    '''Pin how dense-with-zeros inputs of the sparse classes are multiplied ('auto', 'never', 'always' — see
    above); returns the previous mode.'''
    global _dense_route_mode
    if mode not in _DENSE_ROUTE_MODES:
        raise ValueError(f"dense route must be one of {_DENSE_ROUTE_MODES}, not {mode!r}")
    prev, _dense_route_mode = _dense_route_mode, mode
    return prev


def dense_route_pays(density: float, items: int, rows: int, cols: int, width: int) -> bool:
    '''Dense-with-zeros A: the exact-fp32 MFMA product (custom_mm.cublas_mmul / cublas_bmm) against the routes that
    skip the zeros.  Fitted on MI355X (tools/bench_dense_routing.py, profiles/r03_dense_input_routing.log): the MFMA
    product sustains ≈110 TFLOP/s on batches of attention-sized matrices and ≈140 on large ones; the zero-skipping
    routes ≈12–15 TFLOP/s of useful flops plus one pass over A per 256 output columns (≈4 TB/s).  BERT-base probs·V
    (384 × 512² × 64): dense 0.10–0.12 ms whatever the density, skipping 0.96 ms at 100 % kept, 0.19 at 10 %, level
    at ≈2 %; 16384 × 768 × 3072: dense 0.53, CSR route 2.56 / 0.49 / 0.18 ms at 100 / 10 / 2 %.'''
    flops = 2.0 * items * rows * cols * width
    t_dense = flops / (140e12 if rows * cols >= 2048 * 2048 else 110e12) + 4e-6
    t_skip = density * flops / 14e12 + items * rows * cols * 4.0 * -(-width // 256) / 4e12
    return t_dense < t_skip


def _own_density(a: torch.Tensor):
    '''The sampled density of this very tensor OBJECT at its current version, if one has landed (keyed on the Python
    object through a weak reference — not on data_ptr: the caching allocator hands a freed block to the next
    iteration's tensor, which is a different matrix at the same address).'''
    hit = _density_of_tensor.get(id(a))
    if hit is not None and hit[0]() is a and hit[1] == a._version:
        return hit[2]
    return None


def sampled_density(a: torch.Tensor, key, cols: int, owner=None, sample_rows: int = _DENSE_SAMPLE_ROWS):
    '''Density (share of non-zeros) of a dense-with-zeros `a` from an evenly spaced sample of ≤ `sample_rows` rows (one
    small count kernel).  The count comes back WITHOUT stalling the stream: it is copied to pinned memory behind an
    event.  A call gets the count of THIS tensor object (`owner`, at its current version) when that has landed before;
    otherwise the most recent count that has landed under `key` (operands of the same shapes: the previous call's, in a
    loop) — so only the first call of a key waits (the reference converts with `to_sparse_csr()` on every call, which
    synchronises every time, matmuls.py:295-296).  Callers use it to pick a ROUTE, never a result.'''
    if not a.is_cuda:  # (host tensors only meet this in the host-logic tests: no stream to keep running)
        flat = a.reshape(-1, cols)
        sample = flat[::max(1, flat.shape[0] // sample_rows)][:sample_rows]
        return float(torch.count_nonzero(sample)) / max(1, sample.numel())
    owner = a if owner is None else owner  # the caller's tensor object (`a` may be a flattened view of it)
    own = _own_density(owner)
    if own is not None:
        return own
    ent = _density_of_shape.get(key)
    if ent is None:
        if len(_density_of_shape) >= 64:
            _density_of_shape.clear()
        ent = _density_of_shape[key] = {'est': None, 'n': 1, 'event': None, 'src': None,
                                        'host': torch.empty((), dtype=torch.int64, pin_memory=True)}

    def landed():
        ent['est'], ent['event'] = float(ent['host']) / ent['n'], None
        ref, version = ent['src']
        src = ref()
        if src is not None and src._version == version:
            if len(_density_of_tensor) >= 64:
                _density_of_tensor.clear()
            _density_of_tensor[id(src)] = (ref, version, ent['est'])
        ent['src'] = None

    if ent['event'] is not None and ent['event'].query():
        landed()
        own = _own_density(owner)
        if own is not None:  # it was this tensor's own sample
            return own
    if ent['event'] is None:  # no read-back in flight: start one on this call's operand
        flat = a.reshape(-1, cols)
        step = max(1, flat.shape[0] // sample_rows)
        sample = flat[::step][:sample_rows]
        ent['host'].copy_(torch.count_nonzero(sample), non_blocking=True)
        ent['n'] = sample.numel()
        ent['src'] = (weakref.ref(owner), owner._version)
        ent['event'] = torch.cuda.Event()
        ent['event'].record()
    if ent['est'] is None:  # first call of this key: wait for its own count
        ent['event'].synchronize()
        landed()
    return ent['est']


def _dense_route(a: torch.Tensor, b: torch.Tensor, items: int, rows: int, cols: int, width: int, owner=None) -> bool:
    '''Whether a dense-with-zeros `a` is worth the dense MFMA product, decided from its sampled density (see
    sampled_density: nothing stalls the stream after the first call of a shape).  A stale estimate can only cost
    time: in 'auto' mode the result does not depend on the route (see _DENSE_ROUTE_MODES).  Under stream capture
    nothing is read back: the question is not asked.  Products whose dense form takes under ≈20 µs are not worth the
    question either: they take the dense product unless the one-launch in-kernel skip applies.'''
    if not a.is_cuda or torch.cuda.is_current_stream_capturing() or a.numel() == 0 or b.numel() == 0:
        return False
    if 2.0 * items * rows * cols * width / 110e12 < 20e-6:
        # under the gate the dense product is bounded by 20 µs by construction; the dense→CSR route is not (1024² fully
        # dense × 1024: 0.023 against 0.258 ms, profiles/r03_dense_input_routing.log) — so: the one-launch in-kernel skip
        # where a matrix is tiny enough for it (fused_skip_pays), else the matrix cores
        return not fused_skip_pays(items, rows, cols, width)
    est = sampled_density(a, (tuple(a.shape), tuple(b.shape), a.device.index), cols, owner)
    return dense_route_pays(est, items, rows, cols, width)


def _on_matrix_cores(_a: torch.Tensor, _b: torch.Tensor, c: torch.Tensor, mode: str, owner=None) -> bool:
    '''c = _a·_b for a dense-with-zeros _a ([M, K] or [nb, M, K]; _b [K, N] or [nb, K, N]; c preallocated) on the
    MFMA kernel, if `mode` allows it and (auto) the product is worth it.  In 'auto' mode the zero-skipping semantics
    are kept without reading anything back: custom_mm.nonfinite_flag(_b) leaves one int on the device and a gated
    launch of the zero-skipping kernel overwrites c iff that int is set (an inf / nan in _b); with a finite _b both
    products are the same bits (the skipped terms are exact zeros, added in the same ascending-k order).'''
    if mode == 'never':
        return False
    batched = _a.dim() == 3
    nb = _a.shape[0] if batched else 1
    rows, cols, width = _a.shape[-2], _a.shape[-1], c.shape[-1]
    if mode == 'auto':
        if not _dense_route(_a, _b, nb, rows, cols, width, owner):
            return False
        if not custom_mm.naive_spmm_dense_gated(_a, _b, c, c, True):  # dry run: is the gated form available?
            return False
    if batched:
        _bb = _b if _b.dim() == 3 else _b.unsqueeze(0).expand(nb, cols, width)
        custom_mm.cublas_bmm(_a.contiguous(), _bb, c, 3, False, False)
    else:
        custom_mm.cublas_mmul(_a.contiguous(), _b.contiguous(), c, False, False)
    if mode == 'auto':
        custom_mm.naive_spmm_dense_gated(_a, _b, c, custom_mm.nonfinite_flag(_b), False)
    return True


def _spmm_dispatch(a: torch.Tensor, b: torch.Tensor, mm_op, default_op, dense_route=None) -> torch.Tensor:
    '''Shared body of sparse_matmul / naive_matmul: op = ``a @ b`` with ``a``
    taken as sparse (a CSR tensor, or a dense tensor whose exact zeros are
    dropped), semantics of torch.matmul for every rank combination.'''
    if _host_operands(a, b):
        return a @ b  # config C1: dense host operands — the reference's own expression (matmuls.py:279,302); see custom_matmul
    if a.dim() == 1 or b.dim() == 1:
        if a.dim() == 0 or b.dim() == 0:
            raise ValueError('sparse matmul: both arguments need to be at least 1-d')
        a2 = a.unsqueeze(0) if a.dim() == 1 else a
        b2 = b.unsqueeze(-1) if b.dim() == 1 else b
        c = _spmm_dispatch(a2, b2, mm_op, default_op, dense_route)
        if b.dim() == 1:
            c = c.squeeze(-1)
        if a.dim() == 1:
            c = c.squeeze(-2 if b.dim() > 1 else -1)
        return c

    a_shape, b_shape = a.shape, b.shape
    c_rows, c_cols = a_shape[-2], b_shape[-1]
    if a_shape[-1] != b_shape[-2]:
        raise RuntimeError(f'sparse matmul: inner dimensions differ ({a_shape[-1]} vs {b_shape[-2]})')
    dev = b.device if b.is_cuda else a.device

    fused = mm_op is default_op and not a.is_sparse_csr  # dense A + stock kernel: skip zeros in the kernel
    capturing = b.is_cuda and torch.cuda.is_current_stream_capturing()
    # A dense matrix that is not sparse enough belongs on the matrix cores (same result, see _on_matrix_cores):
    # 6× faster on the reference's own naive test shapes (tests/naive_kernel_test.py:48-49 feeds torch.rand).
    if dense_route is not None and dense_route not in _DENSE_ROUTE_MODES:  # (checked whatever the operands: a bad value never passes silently)
        raise ValueError(f"dense_route must be one of {_DENSE_ROUTE_MODES}, not {dense_route!r}")
    mode = (dense_route or _dense_route_mode) if fused and a.is_cuda and b.is_cuda else 'never'

    if a.dim() == 2 and b.dim() == 2:
        c = torch.empty((c_rows, c_cols), device=dev, dtype=torch.float32)
        if _on_matrix_cores(a, b, c, mode, a):
            return c
        # (under capture the conversion's read-back of nnz is not possible: the in-kernel route whenever it applies)
        if fused and (capturing or fused_skip_pays(1, c_rows, a_shape[-1], c_cols)) and custom_mm.naive_spmm_dense(a, b, c):
            return c
        return _csr_product(a, b, c, mm_op, default_op, a)

    if a.dim() == 2:
        # one CSR × a batch of B: C[i] = A·B[i]  ==  A · [K, batch·N]
        batch = tuple(b_shape[:-2])
        _b = b.reshape((-1,) + tuple(b_shape[-2:])).permute(1, 0, 2).reshape(b_shape[-2], -1)
        c = torch.empty((c_rows, _b.shape[1]), device=dev, dtype=torch.float32)
        if not _on_matrix_cores(a, _b, c, mode, a):
            c = _csr_product(a, _b, c, mm_op, default_op, a)
        return c.view(c_rows, -1, c_cols).permute(1, 0, 2).reshape(batch + (c_rows, c_cols))

    if a.is_sparse_csr:
        return _batched_csr_product(a, b, mm_op, default_op)

    if b.dim() == 2:
        # batch of A × one B (the FC-layer call shape): flatten A's rows
        _a = a.reshape(-1, a_shape[-1])
        c = torch.empty((_a.shape[0], c_cols), device=dev, dtype=torch.float32)
        if _on_matrix_cores(_a, b, c, mode, a):
            return c.view(tuple(a_shape[:-1]) + (c_cols,))
        if not (fused and (capturing or fused_skip_pays(1, _a.shape[0], _a.shape[1], c_cols))
                and custom_mm.naive_spmm_dense(_a, b, c)):
            c = _csr_product(_a, b, c, mm_op, default_op, a)
        return c.view(tuple(a_shape[:-1]) + (c_cols,))

    # batch × batch
    batch = torch.broadcast_shapes(tuple(a_shape[:-2]), tuple(b_shape[:-2]))
    _a = a.expand(batch + tuple(a_shape[-2:])).reshape((-1,) + tuple(a_shape[-2:]))
    _b = b.expand(batch + tuple(b_shape[-2:])).reshape((-1,) + tuple(b_shape[-2:]))
    nb = _a.shape[0]
    c = torch.empty((nb, c_rows, c_cols), device=dev, dtype=torch.float32)
    if _on_matrix_cores(_a, _b, c, mode, a):
        pass
    elif fused and (capturing or fused_skip_pays(nb, c_rows, a_shape[-1], c_cols)) and custom_mm.naive_spmm_dense(_a, _b, c):
        pass  # one launch, A read once, no CSR materialised
    elif mm_op is default_op:
        # one dense→CSR conversion and one launch for the whole batch
        est = None
        if _a.is_cuda and not capturing and _a.numel() > 0:  # (the batched entry has no long-row workspace: an estimate is a legal count)
            est = sampled_density(_a, (tuple(_a.shape), tuple(_b.shape), _a.device.index), a_shape[-1], a)
        for lo in range(0, nb, 65535):
            hi = min(nb, lo + 65535)
            values, columns, offsets, nnz = _dense_to_csr(_a[lo:hi], est)
            custom_mm.naive_spmm_batched(values, columns, offsets, nnz, hi - lo,
                                         c_rows, a_shape[-1], _b[lo:hi], c[lo:hi])
    else:
        # a caller-supplied 2-d kernel: apply it slice by slice (reference matmuls.py:289-293)
        for i in range(nb):
            mm_op(*_csr_of(_a[i]), _b[i], c[i])
    return c.view(batch + (c_rows, c_cols))


def sparse_matmul(a: torch.Tensor,
                  b: torch.Tensor,
                  mm_op=custom_mm.cusparse_mmul,
                  dense_route=None) -> torch.Tensor:
    '''
    Uses a sparse kernel to perform matrix multiplication (reference matmuls.py:189-235).

    :param a: This should be a CSR tensor (a dense tensor is converted)
    :param b:
    :param mm_op: kernel to perform basic matrix multiplication,
                  ``mm_op(values, columns, offsets, nnz, rows, cols, B, C) -> C``
    :param dense_route: 'auto' / 'never' / 'always' for a dense `a` (default: matmuls.set_dense_route / MI_DENSE_ROUTE)
    :returns: Matrix multiplication output
    '''
    return _spmm_dispatch(a, b, mm_op, custom_mm.cusparse_mmul, dense_route)


def naive_matmul(a: torch.Tensor,
                 b: torch.Tensor,
                 mm_op=custom_mm.naive_spmm,
                 dense_route=None) -> torch.Tensor:
    '''
    Uses a sparse kernel to perform matrix multiplication (reference matmuls.py:258-303).

    :param a: Torch CSR matrix (a dense tensor is converted)
    :param b:
    :param mm_op: kernel to perform basic matrix multiplication
    :param dense_route: 'auto' / 'never' / 'always' for a dense `a` (default: matmuls.set_dense_route / MI_DENSE_ROUTE)
    :returns: Matrix multiplication output
    '''
    return _spmm_dispatch(a, b, mm_op, custom_mm.naive_spmm, dense_route)


def _csr_cached(m1: torch.Tensor):
    '''(values, columns i32, offsets i32, nnz, rows, cols) and the CSR of m1ᵀ for a CSR tensor.  What is kept ON the
    tensor object between calls is the PATTERN of the transpose — its columns, its offsets and the permutation that
    carries m1's values into it — because in a training loop the pattern is static and the device transpose (1.7 ms
    at the 1M × 1M config) would otherwise be paid on every backward.  The VALUES of m1ᵀ are gathered through the
    permutation on every call (one pass over nnz), so a write to m1's values that no version counter sees
    (`a.values().data.mul_(3)`, a kernel writing through data_ptr) can never leave a stale copy behind.  The entry is
    keyed on the index tensors' storage and version counters; it dies with the tensor.'''
    key = _csr_key(m1)[1:3] + _csr_key(m1)[4:]
    props = _csr_props_cached(m1)
    values, columns, offsets, nnz, rows, cols = props
    hit = getattr(m1, '_mi_csr_cache', None)
    if hit is None or hit[0] != key:
        # the transpose moves 4-byte values untouched: transposing 0, 1, 2, … gives the permutation
        iota = torch.arange(nnz, device=values.device, dtype=torch.int32).view(torch.float32)
        t_perm, t_col, t_off = custom_mm.csr_transpose(iota, columns, offsets, nnz, rows, cols)
        hit = (key, t_perm.view(torch.int32), t_col, t_off)  # int32 permutation: 4 B per non-zero
        try:
            m1._mi_csr_cache = hit
        except (AttributeError, RuntimeError):
            pass  # a tensor type that takes no attributes: just no caching
    t_val = custom_mm.gather_perm(values, hit[1]) if hasattr(custom_mm, 'gather_perm') and values.is_contiguous() \
        else values.index_select(0, hit[1])
    return props, (t_val, hit[2], hit[3])


def _batched_csr_backward(ctx, m1, m2, grad_output):
    '''Both gradients of C[i] = m1[i] @ m2[i] for a batched CSR m1 ([..., M, K], equal non-zero counts per item — what
    torch builds) in a handful of launches for the whole batch (the reference has no backward for this input,
    matmuls.py:250-254):
      grad_m2[i] = m1[i]ᵀ·dC[i] — one batched device transpose (custom_mm.csr_transpose_batched) + one batched
        product, whose B (dC[i], M×N) sits in LDS where it fits; a shared 2-d m2 gets the sum over the items;
      grad_m1 on m1's pattern — ONE SDDMM on the block-diagonal matrix of the batch: rows stacked, item i's columns
        shifted by i·K, against the stacked dC [batch·M, N] and m2 [batch·K, N].'''
    rows, cols = m1.shape[-2], m1.shape[-1]
    vec = m2.dim() == 1  # batched CSR × vector (the forward's unsqueeze): a one-column shared matrix
    if vec:
        m2, grad_output = m2.unsqueeze(-1), grad_output.unsqueeze(-1)
    n = m2.shape[-1]
    val = torch.Tensor.values(m1)
    nb = torch.Tensor.crow_indices(m1).reshape(-1, rows + 1).shape[0]
    total = val.numel()
    per_item = total // max(nb, 1)
    dev = grad_output.device
    # the transposed pattern (a batched device transpose + its permutation + the block-diagonal index arrays) is built only
    # when a step below needs it: for pruned attention (n ≤ 64, an item's m2 in LDS) neither gradient does (round 5)
    offsets, columns, _ = _batched_csr_pattern(m1, dev)
    transposed = lambda: _batched_csr_pattern(m1, dev, transposed=True)[2]  # noqa: E731  (flat_off, diag_columns, t_perm, t_col, t_off)
    g = grad_output.reshape(nb, rows, n).contiguous()
    shared = m2.dim() == 2
    grad_m1 = grad_m2 = None
    if ctx.needs_input_grad[0]:
        # the batched form keeps an item's m2 in LDS where that fits (pruned attention); else (False: nothing ran) ONE
        # SDDMM on the block-diagonal matrix of the batch — the same sums, bit for bit
        gvals = torch.empty(total, device=dev, dtype=torch.float32)
        if not (hasattr(custom_mm, 'sddmm_batched') and
                custom_mm.sddmm_batched(columns, offsets, total, nb, rows, cols, g,
                                        m2.to(dev) if shared else m2.reshape(nb, cols, n).to(dev), gvals)):
            flat_off, diag_columns = transposed()[:2]
            b_stack = (m2.unsqueeze(0).expand(nb, cols, n) if shared else m2.reshape(nb, cols, n)).reshape(nb * cols, n)
            gvals = custom_mm.sddmm(diag_columns, flat_off, total, nb * rows, nb * cols, g.reshape(nb * rows, n),
                                    b_stack.contiguous())
        grad_m1 = torch.sparse_csr_tensor(torch.Tensor.crow_indices(m1), torch.Tensor.col_indices(m1),
                                          gvals.to(val.device).reshape(val.shape), size=m1.shape)
    if ctx.needs_input_grad[1]:
        # m1[i]ᵀ·dC[i] on the cached transposed pattern; the values travel through the cached permutation INSIDE the
        # kernel where its plan allows (the LDS-resident-B kernel: pruned attention), else as one gathered copy
        flat_val = val.reshape(-1).to(dev).contiguous()
        t_val = None
        gb = torch.empty((nb, cols, n), device=dev, dtype=torch.float32)
        # A tensor object seen for the FIRST time (attention probabilities: a new pattern on every step) gets its values
        # transposed directly where the one-workgroup-per-item LDS transpose takes the batch: nothing is kept, no
        # permutation, no block-diagonal index arrays are built (round 5: those cost a fresh pattern 0.39 ms per step,
        # profiles/r05_attention_csr_fresh.log).  From the SECOND backward of the same object on (a static pattern:
        # pruned weights) the transposed pattern and its permutation are kept on the tensor and the values travel
        # through the permutation inside the product kernel, as before.  (The transpose-free product,
        # custom_mm.naive_spmm_batched_at, was built and measured too: 0.54 ms at 10 % kept — its register-indexed FMA
        # per entry and the 16-fold scan of the indices lose to transposing; it stays available as an entry point.)
        launches = range(0, nb, 65535)  # (items per launch, as in the forward)
        hit = getattr(m1, '_mi_batched_pattern', None)
        seen = getattr(m1, '_mi_batched_backwards', 0)
        try:
            m1._mi_batched_backwards = seen + 1
        except (AttributeError, RuntimeError):
            pass
        direct = seen == 0 and (hit is None or hit[3] is None) and hasattr(custom_mm, 'csr_transpose_in_lds') and \
            all(custom_mm.csr_transpose_in_lds((min(nb, lo + 65535) - lo) * per_item, min(nb, lo + 65535) - lo, rows, cols)
                for lo in launches)
        if direct:
            for lo in launches:
                hi = min(nb, lo + 65535)
                p0, p1 = lo * per_item, hi * per_item
                tv, tc, to = custom_mm.csr_transpose_batched(flat_val[p0:p1], columns[p0:p1],
                                                             offsets[lo:hi] if lo == 0 else (offsets[lo:hi] - p0).contiguous(),
                                                             p1 - p0, hi - lo, rows, cols)
                custom_mm.naive_spmm_batched(tv, tc, to, p1 - p0, hi - lo, cols, rows, g[lo:hi], gb[lo:hi])
            launches = range(0)
        else:
            _, _, t_perm, t_col, t_off = transposed()
        for lo in launches:
            hi = min(nb, lo + 65535)
            off_c, g_c = t_off[lo:hi].contiguous(), g[lo:hi]
            if hasattr(custom_mm, 'naive_spmm_batched_perm') and \
                    custom_mm.naive_spmm_batched_perm(flat_val, t_perm, t_col, off_c, total, hi - lo, cols, rows, g_c, gb[lo:hi]):
                continue
            if t_val is None:
                t_val = custom_mm.gather_perm(flat_val, t_perm) if hasattr(custom_mm, 'gather_perm') \
                    else flat_val.index_select(0, t_perm)
            custom_mm.naive_spmm_batched(t_val, t_col, off_c, total, hi - lo, cols, rows, g_c, gb[lo:hi])
        grad_m2 = gb.sum(0) if shared else gb.reshape(m2.shape)
        if vec:
            grad_m2 = grad_m2.squeeze(-1)
    return grad_m1, grad_m2


def _sparse_backward(ctx, grad_output):
    '''Gradients of C = m1 @ m2 with m1 taken as sparse.

    grad_m2 = m1ᵀ·dC.  grad_m1 = dC·m2ᵀ: dense when m1 is a dense tensor (what
    torch autograd of torch.matmul gives), sampled on m1's pattern (SDDMM) when
    m1 is a CSR tensor — a dense M×K gradient cannot exist at the sizes CSR
    inputs are used for.  A CSR m1 may meet a batched m2 ([..., K, N], the forward's
    `A · [K, batch·N]` form, reference call shape matmuls.py:245-256): both gradients are taken
    on the same flattened operands — grad_m2[i] = m1ᵀ·dC[i] is one product with N·batch columns, and
    the SDDMM sums over every item's columns, which is exactly Σ_i dC[i]·m2[i]ᵀ on the pattern.'''
    m1, m2 = ctx.saved_tensors
    grad_m1 = grad_m2 = None

    if m1.is_sparse_csr:
        if m1.dim() > 2:
            return _batched_csr_backward(ctx, m1, m2, grad_output)
        (values, columns, offsets, nnz, rows, cols), (t_val, t_col, t_off) = _csr_cached(m1)
        if m2.dim() == 1:
            g, b = grad_output.reshape(rows, 1), m2.unsqueeze(-1)
        elif m2.dim() == 2:
            g, b = grad_output, m2
        else:
            # [..., K, N] → [K, batch·N] and [..., M, N] → [M, batch·N], item-major columns
            n = m2.shape[-1]
            b = m2.reshape(-1, cols, n).permute(1, 0, 2).reshape(cols, -1)
            g = grad_output.reshape(-1, rows, n).permute(1, 0, 2).reshape(rows, -1)
        if ctx.needs_input_grad[0]:
            gvals = custom_mm.sddmm(columns, offsets, nnz, rows, cols, g, b)
            grad_m1 = torch.sparse_csr_tensor(torch.Tensor.crow_indices(m1), torch.Tensor.col_indices(m1),
                                              gvals.to(m1.device), size=m1.shape)
        if ctx.needs_input_grad[1]:
            gb = torch.empty((cols, g.shape[-1]), device=g.device, dtype=torch.float32)
            # m1ᵀ's pattern is cached on m1: so is its row schedule (the transpose of a skewed matrix is as skewed)
            sched = _row_schedule(m1, '_mi_csr_sched_t', _csr_key(m1)[1:3] + _csr_key(m1)[4:], t_off, nnz, cols, g.shape[-1], t_col, rows) \
                if g.is_contiguous() else None
            if sched is not None:
                gb = custom_mm.naive_spmm_scheduled(sched, t_val, t_col, t_off, nnz, cols, rows, g, gb)
            else:
                gb = custom_mm.naive_spmm(t_val, t_col, t_off, nnz, cols, rows, g, gb)
            if m2.dim() > 2:
                gb = gb.view(cols, -1, m2.shape[-1]).permute(1, 0, 2)
            grad_m2 = gb.reshape(m2.shape)
        return grad_m1, grad_m2

    # dense m1 (sparsified on the fly in forward): both gradients are dense products
    v1, v2 = m1.dim() == 1, m2.dim() == 1
    a = m1.unsqueeze(0) if v1 else m1
    b = m2.unsqueeze(-1) if v2 else m2
    g = grad_output
    if v2:
        g = g.unsqueeze(-1)
    if v1:
        g = g.unsqueeze(-2)
    if ctx.needs_input_grad[0]:
        ga = _sum_to_shape(custom_matmul(g, b, transb=True), a.shape)
        grad_m1 = ga.squeeze(0) if v1 else ga
    if ctx.needs_input_grad[1]:
        gb = _sum_to_shape(custom_matmul(a, g, transa=True), b.shape)
        grad_m2 = gb.squeeze(-1) if v2 else gb
    return grad_m1, grad_m2


class cusparseMM(InplaceFunction):
    @staticmethod
    def forward(ctx, m1, m2):
        ctx.save_for_backward(m1, m2)
        return sparse_matmul(m1, m2)

    @staticmethod
    def backward(ctx, grad_output):
        return _sparse_backward(ctx, grad_output)


class naiveSpMM(InplaceFunction):
    @staticmethod
    def forward(ctx, m1, m2):
        ctx.save_for_backward(m1, m2)
        return naive_matmul(m1, m2)

    @staticmethod
    def backward(ctx, grad_output):
        return _sparse_backward(ctx, grad_output)
