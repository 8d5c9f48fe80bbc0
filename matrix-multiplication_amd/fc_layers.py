'''
fc_layers — the reference's fully-connected call-site modules on the MI355X kernels.

`cublasLinear` and `cusparseLinear` keep the constructor, parameters, initialisation and
`forward(inp)` contract of the reference's modules (benchmarks/cublas_fc_layer.py:12-51,
benchmarks/cusparse_fc_layer.py:12-51): `y = inp @ weight.t() + bias` with `weight`
`[out_features, in_features]`, any number of leading dims on `inp`.

The reference computes `t = cublasMM.apply(inp, self.weight.t()); output = t.clone();
output += self.bias` — a product, a copy and an add, i.e. two extra passes over the output.
Here the bias is added in the kernel epilogue (`custom_mm.cublas_mmul_bias` /
`custom_mm.naive_spmm_bias`) and the backward runs on the same
kernels: grad_inp = dY·W, grad_W = dYᵀ·inp, grad_bias = 1ᵀ·dY.

`cusparseLinear` treats the *activations* as the sparse operand, exactly like the reference
(`cusparseMM.apply(inp, weight.t())`: exact zeros of `inp` are dropped, e.g. after a ReLU).
'''

import math

import torch
import torch.nn as nn
from torch.autograd.function import InplaceFunction

import custom_mm
from matmuls import custom_matmul, sampled_density


def _column_sums(g2d):
    '''Bias gradient: column sums of dY on the device (custom_mm.column_sums).'''
    return custom_mm.column_sums(g2d)


class _LinearBias(InplaceFunction):
    '''y = x·Wᵀ (+ bias) with the dense kernel; x [..., in], W [out, in].'''

    @staticmethod
    def forward(ctx, inp, weight, bias):
        ctx.save_for_backward(inp, weight)
        ctx.has_bias = bias is not None
        x2 = inp.reshape(-1, inp.shape[-1])
        out = torch.empty((x2.shape[0], weight.shape[0]), device=inp.device, dtype=torch.float32)
        if bias is not None:
            custom_mm.cublas_mmul_bias(x2, weight, bias, out, False, True)
        else:
            custom_mm.cublas_mmul(x2, weight, out, False, True)
        return out.view(tuple(inp.shape[:-1]) + (weight.shape[0],))

    @staticmethod
    def backward(ctx, grad_output):
        inp, weight = ctx.saved_tensors
        g2 = grad_output.reshape(-1, grad_output.shape[-1])
        x2 = inp.reshape(-1, inp.shape[-1])
        grad_inp = grad_w = grad_b = None
        if ctx.needs_input_grad[0]:
            grad_inp = custom_matmul(g2, weight).view(inp.shape)       # dY·W
        if ctx.needs_input_grad[1]:
            grad_w = custom_matmul(g2, x2, transa=True)                 # dYᵀ·x
        if ctx.has_bias and ctx.needs_input_grad[2]:
            grad_b = _column_sums(g2)
        return grad_inp, grad_w, grad_b


def sparse_forward_pays(nnz, tokens, in_features, out_features):
    '''Forward cost model of `cusparseLinear` (fitted on MI355X, tools/bench_fc.py,
    profiles/r02_fc_layer_timings.log): the exact-fp32 MFMA product runs at ≈130 TFLOP/s on these
    shapes, the row-split SpMM of ReLU-sparse activations at ≈15 TFLOP/s of useful flops, plus three
    passes over the activations for the exact count and the fill.  Sparse is taken only when the
    model gives it a 10 % lead.'''
    t_dense = 2.0 * tokens * in_features * out_features / 130e12
    t_sparse = 2.0 * nnz * out_features / 15e12 + 3.0 * tokens * in_features * 4 / 4e12
    return t_sparse < 0.9 * t_dense


_SAMPLE_ROWS = 512


def worth_sampling(tokens, in_features, out_features):
    '''A layer whose dense product takes under a quarter of a millisecond cannot win back the sample's
    launches + read-back (≈0.03 ms): measured 0.232 vs 0.200 ms at 16384 × 3072 → 256.'''
    return 2.0 * tokens * in_features * out_features / 130e12 >= 0.25e-3


class _SparseLinearBias(InplaceFunction):
    '''y = sparse(x)·Wᵀ (+ bias): the activations' exact zeros are skipped when that pays.

    The layer first estimates the density from an evenly spaced sample of ≤ 512 token rows (one small
    count kernel).  Since round 4 the count comes back WITHOUT stalling the stream (matmuls.sampled_density:
    pinned memory behind an event; a forward decides from the most recent count that has landed for this
    layer's shapes — the previous step's, in a loop — so only the first forward of a shape waits; the
    reference's cusparseDenseToSparse analysis step synchronises on every call, src/baseline_mm.cu:232-247).
    When the model says the sparse route wins, the conversion needs no read-back either where the layer is no
    wider than the long-row threshold: the CSR arrays get room for every element of x (the fill cannot
    overflow) and the kernels walk the rows through the offsets; the estimate only steers their choice of
    plan.  Otherwise the dense MFMA product runs.  Both routes sum the same terms in the same (column)
    order, the skipped ones being exact zeros, so the result does not depend on the route (finite weights;
    a zero of x facing an inf / nan weight gives nan on the dense route, as in torch.matmul).  Under stream
    capture nothing may be read back: the dense product is used.  (The fused zero-skipping kernel of naiveSpMM is not
    used here: it scans every element of x per 64 output columns and lost to both routes at FC sizes —
    0.46 ms vs 0.20 ms dense at 16384 × 3072 → 256, 99 % zeros.)'''

    @staticmethod
    def forward(ctx, inp, weight, bias):
        ctx.has_bias = bias is not None
        x2 = inp.reshape(-1, inp.shape[-1])
        tokens, fin, fout = x2.shape[0], x2.shape[1], weight.shape[0]
        out = torch.empty((tokens, fout), device=inp.device, dtype=torch.float32)
        csr = None
        capturing = inp.is_cuda and torch.cuda.is_current_stream_capturing()
        ctx.x_density = 1.0
        nnz_arg = 0
        ctx.exact_csr = True
        if not capturing and x2.numel() > 0 and worth_sampling(tokens, fin, fout):
            est = sampled_density(x2, ('fc', tuple(x2.shape), fout, inp.device.index), fin, owner=inp,
                                  sample_rows=_SAMPLE_ROWS)
            ctx.x_density = est
            if sparse_forward_pays(est * x2.numel(), tokens, fin, fout):
                offsets = custom_mm.dense_row_offsets(x2)
                if fin <= custom_mm.long_row_threshold():
                    # no read-back: room for every element.  The count the kernels are told is the estimate — possibly
                    # BELOW the true count (stale or sampled): allowed here because rule 0 runs without the long-row
                    # workspace, the only thing sized from it (include/mi_spmm.h); MI_SPMM_LDS_B takes the clamp of its
                    # 16-byte loads from max(count, offsets' last entry) on the device, every other plan reads through
                    # the offsets alone (tests: test_spmm_count_as_bound_or_estimate_same_bits, test_cusparse_linear_lds_fit_…)
                    values, columns = custom_mm.dense_to_csr_fill(x2, offsets, x2.numel())
                    csr = (values, columns, offsets.view(-1))
                    nnz_arg = min(x2.numel(), max(1, int(est * x2.numel())))
                    ctx.exact_csr = False
                else:
                    nnz = int(offsets.view(-1)[-1])  # wide layers: the long-row rule sizes its lists from the exact count
                    ctx.x_density = nnz / x2.numel()
                    if sparse_forward_pays(nnz, tokens, fin, fout):
                        values, columns = custom_mm.dense_to_csr_fill(x2, offsets, nnz)
                        csr = (values, columns, offsets.view(-1))
                        nnz_arg = nnz
        if csr is not None:
            wt = weight.t().contiguous()                                 # [in, out] row-major B operand
            # rows of x have at most `fin` non-zeros (no duplicate columns): a layer no wider than the long-row
            # threshold runs as ONE launch, without the workspace and the helper kernels of the long-row rule
            rule = 0 if fin <= custom_mm.long_row_threshold() else -1
            if bias is not None:
                custom_mm.naive_spmm_bias_ex(csr[0], csr[1], csr[2], nnz_arg, tokens, fin, wt, bias, out, rule)
            else:
                custom_mm.naive_spmm_ex(csr[0], csr[1], csr[2], nnz_arg, tokens, fin, wt, out, rule)
        elif bias is not None:
            custom_mm.cublas_mmul_bias(x2, weight, bias, out, False, True)
        else:
            custom_mm.cublas_mmul(x2, weight, out, False, True)
        ctx.has_csr = csr is not None
        ctx.save_for_backward(inp, weight, *(csr or ()))
        return out.view(tuple(inp.shape[:-1]) + (fout,))

    @staticmethod
    def backward(ctx, grad_output):
        inp, weight = ctx.saved_tensors[:2]
        g2 = grad_output.reshape(-1, grad_output.shape[-1])
        x2 = inp.reshape(-1, inp.shape[-1])
        grad_inp = grad_w = grad_b = None
        if ctx.needs_input_grad[0]:
            grad_inp = custom_matmul(g2, weight).view(inp.shape)       # dense gradient, as torch autograd gives
        if ctx.needs_input_grad[1] and (ctx.x_density > 0.12 or not ctx.has_csr):
            # x is not sparse enough for the sparse route to pay (in × out is a small output: few
            # workgroups): the dense product sums the same terms in the same token order, the
            # skipped ones being exact zeros.  Measured at 16384 tokens, 3072 → 768, half zeros:
            # forward + backward 9.1 → 3.6 ms (tools/bench_fc.py).
            grad_w = custom_matmul(g2, x2, transa=True)
        elif ctx.needs_input_grad[1]:
            # dYᵀ·x = (xᵀ·dY)ᵀ with x sparse: the CSR kept from forward, transposed, then the row-split kernel
            values, columns, offsets = ctx.saved_tensors[2:]
            # (arrays sized for every element of x: the transpose needs the exact count — the one read-back of this route)
            nnz = values.numel() if ctx.exact_csr else int(offsets[-1])
            t_val, t_col, t_off = custom_mm.csr_transpose(values[:nnz], columns[:nnz], offsets, nnz,
                                                          x2.shape[0], x2.shape[1])
            gwt = torch.empty((x2.shape[1], g2.shape[1]), device=g2.device, dtype=torch.float32)
            custom_mm.naive_spmm(t_val, t_col, t_off, nnz, x2.shape[1], x2.shape[0], g2, gwt)
            grad_w = gwt.t()
        if ctx.has_bias and ctx.needs_input_grad[2]:
            grad_b = _column_sums(g2)
        return grad_inp, grad_w, grad_b


class _LinearBase(nn.Module):
    _fn = None

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        torch.manual_seed(0)  # the reference seeds here (cublas_fc_layer.py:15)

        self.in_features = in_features
        self.out_features = out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_features))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = nn.init._calculate_fan_in_and_fan_out(self.weight)
            bound = 1 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, inp):
        if inp.shape[-1] != self.in_features:
            print('Invalid dimensions')  # reference behaviour (cublas_fc_layer.py:37-40)
            return 0
        return type(self)._fn.apply(inp, self.weight, self.bias)

    def extra_repr(self):
        return 'in_features={}, out_features={}, bias={}'.format(
            self.in_features, self.out_features, self.bias is not None
        )


class cublasLinear(_LinearBase):
    _fn = _LinearBias


class cusparseLinear(_LinearBase):
    _fn = _SparseLinearBias
