"""`Shift`, drop-in for the reference's models_pytorch/utils/shift_cuda.py:177-192 -- its only
native op.  Reference: a CUDA C string JIT-compiled through cupy per shape; here: one precompiled
gfx950 kernel behind the C ABI (mlpk_shift_nchw), launched the same way (raw pointers, caller-
allocated output, torch's current stream).  Forward only (the north star is the forward path)."""
import torch
from torch import nn

from ... import engine as E


def _shift_gpu(input, shift, dim):
    # same argument checks and error types as _shift_cuda (shift_cuda.py:164-174)
    assert shift >= 3 and shift % 2 == 1
    assert dim == 2 or dim == 3
    if not input.is_cuda:
        raise NotImplementedError
    assert input.dim() == 4
    x = input.contiguous()
    out = torch.empty_like(x)
    with E.on_device(x):
        E.shift_nchw(x, out, shift, dim)
    return out


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


def torch_shift(x, shift_size, dim):
    """shift_cuda.py:195-205 -- the reference's pure-torch restatement of the same operation (pad, per-chunk roll, crop) on
    (B, C, H, W): here the same kernel as `Shift` (chunk g of ceil(C / shift_size) channels moves by g - shift_size // 2 pixels along
    `dim`, zeros shifted in), out of place."""
    if shift_size == 1:
        return x
    return _shift_gpu(x, shift_size, dim)
