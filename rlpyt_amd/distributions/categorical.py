"""Categorical action distribution (API of rlpyt/distributions/categorical.py:14-43,
``EPS = 1e-8`` at :9).  The loss-side arithmetic (likelihood ratio, entropy, perplexity,
masked means) is fused inside the HIP loss kernels; the methods here are the thin
tensor-level forms used for sampling and diagnostics (e.g. KL)."""
import torch

from ..utils.collections import namedarraytuple
from ..utils.tensor import from_onehot, select_at_indexes, to_onehot, valid_mean

EPS = 1e-8

DistInfo = namedarraytuple("DistInfo", ["prob"])


class Distribution:
    @property
    def dim(self):
        raise NotImplementedError

    def perplexity(self, dist_info):
        return torch.exp(self.entropy(dist_info))

    def mean_entropy(self, dist_info, valid=None):
        return valid_mean(self.entropy(dist_info), valid)

    def mean_perplexity(self, dist_info, valid=None):
        return valid_mean(self.perplexity(dist_info), valid)


class DiscreteMixin:
    def __init__(self, dim, dtype=torch.long, onehot_dtype=torch.float):
        self._dim = dim
        self.dtype = dtype
        self.onehot_dtype = onehot_dtype

    @property
    def dim(self):
        return self._dim

    def to_onehot(self, indexes, dtype=None):
        return to_onehot(indexes, self._dim, dtype=dtype or self.onehot_dtype)

    def from_onehot(self, onehot, dtype=None):
        return from_onehot(onehot, dtype=dtype or self.dtype)


class Categorical(DiscreteMixin, Distribution):
    def kl(self, old_dist_info, new_dist_info):
        p, q = old_dist_info.prob, new_dist_info.prob
        return torch.sum(p * (torch.log(p + EPS) - torch.log(q + EPS)), dim=-1)

    def mean_kl(self, old_dist_info, new_dist_info, valid=None):
        return valid_mean(self.kl(old_dist_info, new_dist_info), valid)

    def sample(self, dist_info, generator=None):
        """``torch.multinomial`` over the trailing dim, on whatever device prob lives."""
        p = dist_info.prob
        s = torch.multinomial(p.reshape(-1, self.dim), num_samples=1, generator=generator)
        return s.reshape(p.shape[:-1]).type(self.dtype)

    def entropy(self, dist_info):
        p = dist_info.prob
        return -torch.sum(p * torch.log(p + EPS), dim=-1)

    def log_likelihood(self, indexes, dist_info):
        return torch.log(select_at_indexes(indexes, dist_info.prob) + EPS)

    def likelihood_ratio(self, indexes, old_dist_info, new_dist_info):
        num = select_at_indexes(indexes, new_dist_info.prob)
        den = select_at_indexes(indexes, old_dist_info.prob)
        return (num + EPS) / (den + EPS)
