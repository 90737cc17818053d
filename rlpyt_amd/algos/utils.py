"""Return / advantage functions with the reference's names and signatures
(rlpyt/algos/utils.py:8-112), executed by the HIP scan kernels.

Drop-in rule: inputs may be device tensors (zero-copy), CPU tensors or numpy arrays; the
result comes back in the same form (host inputs are staged through HBM -- that is still
the HIP path; there is no CPU implementation here).  ``*_dest`` out-params are honoured.
"""
import numpy as np
import torch

from .. import ops


def _stage(x, device):
    """-> (device tensor, kind) with kind in {"cuda", "cpu", "numpy"}."""
    if isinstance(x, torch.Tensor):
        if x.is_cuda:
            return x, "cuda"
        return x.to(device, non_blocking=True), "cpu"
    return torch.from_numpy(np.ascontiguousarray(x)).to(device, non_blocking=True), "numpy"


def _unstage(y, kind, dest=None):
    if kind == "cuda":
        if dest is not None and dest.data_ptr() != y.data_ptr():
            dest.copy_(y)
            return dest
        return y
    out = y.cpu()
    if kind == "numpy":
        out = out.numpy()
    if dest is not None:
        dest[...] = out
        return dest
    return out


def _device():
    return torch.device("cuda", torch.cuda.current_device())


def discount_return(reward, done, bootstrap_value, discount, return_dest=None):
    """Discounted return-to-go with bootstrapping; resets where done (utils.py:8-21)."""
    dev = reward.device if isinstance(reward, torch.Tensor) and reward.is_cuda else _device()
    r, kind = _stage(reward, dev)
    d, _ = _stage(done, dev)
    bv, _ = _stage(torch.as_tensor(bootstrap_value) if not isinstance(
        bootstrap_value, (torch.Tensor, np.ndarray)) else bootstrap_value, dev)
    dest = return_dest if (kind == "cuda" and return_dest is not None) else None
    ret = ops.discount_return(r, d, bv, discount, return_dest=dest)
    return _unstage(ret, kind, return_dest)


def generalized_advantage_estimation(reward, value, done, bootstrap_value, discount,
                                     gae_lambda, advantage_dest=None, return_dest=None):
    """GAE advantages and returns (utils.py:24-40)."""
    dev = reward.device if isinstance(reward, torch.Tensor) and reward.is_cuda else _device()
    r, kind = _stage(reward, dev)
    v, _ = _stage(value, dev)
    d, _ = _stage(done, dev)
    bv, _ = _stage(torch.as_tensor(bootstrap_value) if not isinstance(
        bootstrap_value, (torch.Tensor, np.ndarray)) else bootstrap_value, dev)
    on_dev = kind == "cuda"
    adv, ret = ops.gae(r, v, d, bv, discount, gae_lambda,
                       advantage_dest=advantage_dest if on_dev else None,
                       return_dest=return_dest if on_dev else None)
    return _unstage(adv, kind, advantage_dest), _unstage(ret, kind, return_dest)


def discount_return_n_step(reward, done, n_step, discount, return_dest=None, done_n_dest=None,
                           do_truncated=False):
    """n-step returns and n-step done flags (utils.py:67-101)."""
    dev = reward.device if isinstance(reward, torch.Tensor) and reward.is_cuda else _device()
    r, kind = _stage(reward, dev)
    d, _ = _stage(done, dev)
    on_dev = kind == "cuda"
    ret, dn = ops.discount_return_n_step(r, d, n_step, discount,
                                         return_dest=return_dest if on_dev else None,
                                         done_n_dest=done_n_dest if on_dev else None,
                                         do_truncated=do_truncated)
    done_dtype = done.dtype if isinstance(done, (torch.Tensor, np.ndarray)) else None
    ret_o, dn_o = _unstage(ret, kind, return_dest), _unstage(dn, kind, None)
    if done_dtype is not None and not on_dev:
        dn_o = dn_o.astype(done_dtype) if kind == "numpy" else dn_o.type(done_dtype)
    if done_n_dest is not None and not on_dev:
        done_n_dest[...] = dn_o
        dn_o = done_n_dest
    return ret_o, dn_o


def valid_from_done(done):
    """Float mask, zero after the first done along time (utils.py:104-112)."""
    dev = done.device if isinstance(done, torch.Tensor) and done.is_cuda else _device()
    d, kind = _stage(done, dev)
    return _unstage(ops.valid_from_done(d), kind)
