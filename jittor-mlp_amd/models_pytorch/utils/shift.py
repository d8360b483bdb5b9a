"""`Shift`, drop-in for the reference's models_pytorch/utils/shift_cuda.py:177-192 -- its only
native op.  Reference: a CUDA C string JIT-compiled through cupy per shape; here: one precompiled
gfx950 kernel behind the C ABI (mlpk_shift_nchw / mlpk_shift_nchw_backward), launched the same way (raw pointers, caller-
allocated output, torch's current stream).  Like the reference's `_shift` it is an autograd.Function with the op's own
backward kernel (shift_cuda.py:75-103,131-162); the models' forward paths around it remain inference-only."""
import torch
from torch import nn

from ... import engine as E


class _shift(torch.autograd.Function):
    """shift_cuda.py:106-162: forward = the gather kernel, backward = its adjoint kernel on grad_output; (grad_input, None, None)."""

    @staticmethod
    def forward(ctx, input, shift, dim):
        assert input.dim() == 4 and input.is_cuda
        x = input.contiguous()
        out = torch.empty_like(x)                                  # caller-allocated, as input.new(...) at shift_cuda.py:112
        with E.on_device(x):
            E.shift_nchw(x, out, shift, dim)
        ctx.shift, ctx.dim = shift, dim
        return out

    @staticmethod
    def backward(ctx, grad_output):
        assert grad_output.is_cuda
        if not ctx.needs_input_grad[0]:
            return None, None, None
        g = grad_output.contiguous()                               # shift_cuda.py:133-134
        grad_input = torch.empty_like(g)
        with E.on_device(g):
            E.shift_nchw_backward(g, grad_input, ctx.shift, ctx.dim)
        return grad_input, None, None


def _shift_gpu(input, shift, dim):
    # same argument checks and error types as _shift_cuda (shift_cuda.py:164-174)
    assert shift >= 3 and shift % 2 == 1
    assert dim == 2 or dim == 3
    if not input.is_cuda:
        raise NotImplementedError
    assert input.dim() == 4
    return _shift.apply(input, shift, dim)


class Shift(nn.Module):
    def __init__(self, kernel_size, dim):
        super(Shift, self).__init__()
        self.kernel_size = kernel_size
        self.dim = dim
        assert dim == 2 or dim == 3
        assert kernel_size % 2 == 1

    def forward(self, x):
        if self.kernel_size == 1:
            return x
        return _shift_gpu(x, self.kernel_size, self.dim)


def _shift_by_slices(x, shift_size, dim):
    """chunk g of ceil(C / shift_size) channels moves by g - shift_size // 2 positions along `dim`, zeros shifted in: the same map as the
    kernel, written with slice assignments (differentiable, any device, any shift size)"""
    C, n = x.shape[1], x.shape[dim]
    per = -(-C // shift_size)
    out = torch.zeros_like(x)
    for g, c0 in enumerate(range(0, C, per)):
        s = g - shift_size // 2
        if abs(s) >= n:
            continue
        src = x[:, c0:c0 + per].narrow(dim, max(0, -s), n - abs(s))
        out[:, c0:c0 + per].narrow(dim, max(0, s), n - abs(s)).copy_(src)
    return out


def torch_shift(x, shift_size, dim):
    """shift_cuda.py:195-205 -- the reference's device-independent restatement of the operation (pad, per-chunk roll, crop) on
    (B, C, H, W).  On a GPU tensor with a shift size the kernel takes (odd, >= 3) it IS the kernel (`Shift`'s autograd.Function);
    CPU tensors and every other shift size run the same index map in torch slice operations, so it stays the cross-check the
    reference uses it as."""
    if shift_size == 1:
        return x
    if x.is_cuda and x.dim() == 4 and shift_size >= 3 and shift_size % 2 == 1:
        return _shift_gpu(x, shift_size, dim)
    return _shift_by_slices(x, shift_size, dim)
