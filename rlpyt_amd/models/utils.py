"""Model helpers (names follow rlpyt/models/utils.py:18-65)."""
import torch


def conv2d_output_shape(h, w, kernel_size=1, stride=1, padding=0, dilation=1):
    """Output (H, W) of a conv / pool layer."""
    pair = lambda x: x if isinstance(x, tuple) else (x, x)  # noqa: E731
    (kh, kw), (sh, sw), (ph, pw) = pair(kernel_size), pair(stride), pair(padding)
    h = (h + 2 * ph - dilation * (kh - 1) - 1) // sh + 1
    w = (w + 2 * pw - dilation * (kw - 1) - 1) // sw + 1
    return h, w


class ScaleGrad(torch.autograd.Function):
    """Identity forward, gradient scaled by ``scale`` backward (dueling heads)."""

    @staticmethod
    def forward(ctx, tensor, scale):
        ctx.scale = scale
        return tensor

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output * ctx.scale, None


scale_grad = ScaleGrad.apply


_DDP_PREFIX = "module."


def strip_ddp_state_dict(state_dict):
    """Keys without DistributedDataParallel's wrapper prefix (same mapping type, same order)."""
    n = len(_DDP_PREFIX)
    return type(state_dict)((k[n:] if k.startswith(_DDP_PREFIX) else k, v)
                            for k, v in state_dict.items())


def update_state_dict(model, state_dict, tau=1, strip_ddp=True):
    """Move ``model`` towards ``state_dict``: ``tau = 1`` copies, ``0 < tau < 1`` blends
    ``tau * new + (1 - tau) * old`` entry by entry (target networks), ``tau <= 0`` does nothing."""
    if tau <= 0:
        return
    new = strip_ddp_state_dict(state_dict) if strip_ddp else state_dict
    if tau != 1:
        old = model.state_dict()
        new = {name: tau * new[name] + (1 - tau) * cur for name, cur in old.items()}
    model.load_state_dict(new)
