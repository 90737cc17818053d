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


def strip_ddp_state_dict(state_dict):
    """Drop DistributedDataParallel's ``module.`` key prefix."""
    out = type(state_dict)()
    for k, v in state_dict.items():
        out[k[7:] if k.startswith("module.") else k] = v
    return out


def update_state_dict(model, state_dict, tau=1, strip_ddp=True):
    """Hard (tau=1) or soft (0<tau<1: tau*new + (1-tau)*old) parameter update."""
    if strip_ddp:
        state_dict = strip_ddp_state_dict(state_dict)
    if tau == 1:
        model.load_state_dict(state_dict)
    elif tau > 0:
        model.load_state_dict({k: tau * state_dict[k] + (1 - tau) * v
                               for k, v in model.state_dict().items()})
