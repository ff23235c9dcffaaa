"""Autograd function behind ``F.linear`` on quantized weights (optimum/quanto/tensor/function.py:21-63)."""
import torch

__all__ = ["QuantizedLinearFunction"]


class QuantizedLinearFunction(torch.autograd.Function):
    """``input @ other.t() + bias`` with an explicit backward.

    The backward treats quantization as the identity for the weight (the scale/shift receive no gradient),
    which lets the forward call custom ``quanto::`` ops that have no autograd formula.
    Subclasses override ``forward`` to route to a fused kernel and inherit ``backward``.
    """

    @staticmethod
    def forward(ctx, input, other, bias=None):
        ctx.save_for_backward(input, other)
        out = torch.matmul(input, other.t())  # `other.t()` on a QTensor dequantizes through qfallback
        return out if bias is None else out + bias

    @staticmethod
    def backward(ctx, grad_out):
        input, other = ctx.saved_tensors
        n_out, n_in = other.shape
        grad_in = grad_w = grad_b = None
        if ctx.needs_input_grad[0]:
            grad_in = torch.matmul(grad_out, other)
        if ctx.needs_input_grad[1]:
            grad_w = torch.matmul(grad_out.reshape(-1, n_out).t(), input.reshape(-1, n_in))
        if ctx.needs_input_grad[2]:
            grad_b = grad_out.sum(tuple(range(grad_out.ndim - 1)))
        return grad_in, grad_w, grad_b
