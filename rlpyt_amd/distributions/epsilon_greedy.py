"""Epsilon-greedy action selection over Q-values (rlpyt/distributions/epsilon_greedy.py)."""
import torch

from .categorical import DiscreteMixin, Distribution


class EpsilonGreedy(DiscreteMixin, Distribution):
    """argmax with probability 1-eps, uniform random otherwise; eps scalar or per-env."""

    def __init__(self, epsilon=1, **kwargs):
        super().__init__(**kwargs)
        self._epsilon = epsilon

    def sample(self, q, generator=None):
        arg_select = torch.argmax(q, dim=-1)
        eps = self._epsilon
        if isinstance(eps, torch.Tensor):
            eps = eps.to(q.device)
        mask = torch.rand(arg_select.shape, device=q.device, generator=generator) < eps
        arg_rand = torch.randint(low=0, high=q.shape[-1], size=arg_select.shape,
                                 device=q.device, generator=generator)
        return torch.where(mask, arg_rand, arg_select)

    @property
    def epsilon(self):
        return self._epsilon

    def set_epsilon(self, epsilon):
        self._epsilon = epsilon


class CategoricalEpsilonGreedy(EpsilonGreedy):
    """For distributional Q (p over atoms z): greedy w.r.t. expected value."""

    def __init__(self, z=None, **kwargs):
        super().__init__(**kwargs)
        self.z = z

    def sample(self, p, z=None, generator=None):
        q = torch.tensordot(p, z if z is not None else self.z, dims=1)
        return super().sample(q, generator=generator)

    def set_z(self, z):
        self.z = z
