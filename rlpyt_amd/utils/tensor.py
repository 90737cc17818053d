"""Small tensor helpers with the reference's names (rlpyt/utils/tensor.py)."""
import torch


def select_at_indexes(indexes, tensor):
    """``tensor[..., indexes]`` along the dim after ``indexes``' dims (tensor.py:5-15)."""
    n = indexes.dim()
    assert tuple(indexes.shape) == tuple(tensor.shape[:n])
    flat = tensor.reshape((-1,) + tuple(tensor.shape[n:]))
    picked = flat[torch.arange(flat.shape[0], device=tensor.device), indexes.reshape(-1)]
    return picked.reshape(tuple(tensor.shape[:n]) + tuple(tensor.shape[n + 1:]))


def to_onehot(indexes, num, dtype=None):
    """One-hot along a new trailing dim (tensor.py:18-27)."""
    dtype = indexes.dtype if dtype is None else dtype
    out = torch.zeros(tuple(indexes.shape) + (num,), dtype=dtype, device=indexes.device)
    out.scatter_(-1, indexes.unsqueeze(-1).long(), 1)
    return out


def from_onehot(onehot, dim=-1, dtype=None):
    idx = torch.argmax(onehot, dim=dim)
    return idx if dtype is None else idx.type(dtype)


def valid_mean(tensor, valid=None, dim=None):
    """Masked mean ``sum(x*v)/sum(v)`` (tensor.py:39-46)."""
    dim = () if dim is None else dim
    if valid is None:
        return tensor.mean(dim=dim)
    valid = valid.type(tensor.dtype)
    return (tensor * valid).sum(dim=dim) / valid.sum(dim=dim)


def infer_leading_dims(tensor, dim):
    """(lead_dim, T, B, shape) for inputs with [], [B] or [T,B] leading dims (tensor.py:49-66)."""
    lead_dim = tensor.dim() - dim
    assert lead_dim in (0, 1, 2)
    if lead_dim == 2:
        T, B = tensor.shape[:2]
    else:
        T, B = 1, (1 if lead_dim == 0 else tensor.shape[0])
    return lead_dim, T, B, tensor.shape[lead_dim:]


def restore_leading_dims(tensors, lead_dim, T=1, B=1):
    """Inverse of ``infer_leading_dims`` on model outputs shaped [T*B, ...] (tensor.py:69-86)."""
    is_seq = isinstance(tensors, (tuple, list))
    ts = tuple(tensors) if is_seq else (tensors,)
    if lead_dim == 2:
        ts = tuple(t.reshape((T, B) + tuple(t.shape[1:])) for t in ts)
    elif lead_dim == 0:
        assert B == 1
        ts = tuple(t.squeeze(0) for t in ts)
    return ts if is_seq else ts[0]
