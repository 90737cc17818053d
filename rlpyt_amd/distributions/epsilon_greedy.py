"""Epsilon-greedy action selection over Q-values, on whatever device the Q-values live
(the reference's rlpyt/distributions/epsilon_greedy.py draws on the host and patches the greedy
choice by boolean indexing; here the draw is branch-free device code -- one ``torch.where`` over
pre-drawn uniform / integer tensors -- so it can sit inside a captured step graph, optionally on
a per-pipeline-group ``torch.Generator``)."""
import torch

from .categorical import DiscreteMixin, Distribution


def _explore(greedy, n_actions, epsilon, generator):
    """Replace each greedy choice by a uniformly random action with probability ``epsilon``
    (scalar, or one value per environment)."""
    dev = greedy.device
    eps = epsilon.to(dev) if isinstance(epsilon, torch.Tensor) else epsilon
    take_random = torch.rand(greedy.shape, device=dev, generator=generator) < eps
    random_action = torch.randint(0, n_actions, greedy.shape, device=dev, generator=generator)
    return torch.where(take_random, random_action, greedy)


class EpsilonGreedy(DiscreteMixin, Distribution):
    """``sample(q)``: argmax with probability 1 - epsilon, uniform otherwise."""

    def __init__(self, epsilon=1, **kwargs):
        super().__init__(**kwargs)
        self._epsilon = epsilon

    epsilon = property(lambda self: self._epsilon)

    def set_epsilon(self, epsilon):
        """Scalar, or a tensor with one epsilon per environment (vector epsilon)."""
        self._epsilon = epsilon

    def sample(self, q, generator=None):
        return _explore(q.argmax(dim=-1), q.shape[-1], self._epsilon, generator)


class CategoricalEpsilonGreedy(EpsilonGreedy):
    """For distributional Q-values ``p[..., A, n_atoms]`` over the atom grid ``z``: greedy with
    respect to the expected value."""

    def __init__(self, z=None, **kwargs):
        super().__init__(**kwargs)
        self.z = z

    def set_z(self, z):
        self.z = z

    def sample(self, p, z=None, generator=None):
        expected = torch.tensordot(p, self.z if z is None else z, dims=1)
        return _explore(expected.argmax(dim=-1), expected.shape[-1], self._epsilon, generator)
