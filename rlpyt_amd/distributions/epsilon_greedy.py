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
    """``sample(q)``: argmax with probability 1 - epsilon, uniform otherwise.

    On the device epsilon lives in PERSISTENT buffers that ``set_epsilon`` overwrites in place
    (``bind_device``): the sampler's captured step graphs compare against device memory, so a new
    epsilon (the schedule of ``sample_mode``, ``eval_mode``) reaches graphs captured earlier -- a
    Python scalar would be frozen into the graph at capture time.  With vector epsilon (one value
    per environment, rlpyt/agents/dqn/epsilon_greedy.py:47-63) a pipeline group sees its own slice
    (``select_envs``)."""

    def __init__(self, epsilon=1, **kwargs):
        super().__init__(**kwargs)
        self._epsilon = epsilon
        self._buf = self._scalar_buf = None
        self._uploaded = None           # what the device buffers hold (float, or a host tensor)
        self._env_slice = None

    epsilon = property(lambda self: self._epsilon)

    def set_epsilon(self, epsilon):
        """Scalar, or a tensor with one epsilon per environment (vector epsilon)."""
        self._epsilon = epsilon
        self._refresh()

    def bind_device(self, device, n_envs=None):
        if torch.device(device).type != "cuda":
            return
        self._scalar_buf = torch.zeros(1, dtype=torch.float32, device=device)
        self._buf = (torch.zeros(int(n_envs), dtype=torch.float32, device=device)
                     if n_envs else None)
        self._uploaded = None
        self._refresh()

    def select_envs(self, lo=None, hi=None):
        self._env_slice = None if lo is None else (int(lo), int(hi))

    def _refresh(self):
        """Bring the device buffers up to date.  A scalar epsilon goes out as ``fill_`` launches: a
        host -> device copy from pageable memory makes the caller wait for everything queued on the
        stream before it (``sample_mode`` runs while the previous iteration's updates are still in
        flight), a fill does not.  Unchanged values are not written again."""
        if self._scalar_buf is None:
            return
        eps = self._epsilon
        if not isinstance(eps, torch.Tensor) or eps.numel() == 1:
            value = float(eps)
            if self._uploaded != value:
                self._scalar_buf.fill_(value)
                if self._buf is not None:
                    self._buf.fill_(value)
                self._uploaded = value
            return
        e = eps.detach().to(dtype=torch.float32, device="cpu").reshape(-1)
        if self._buf is not None and e.numel() != self._buf.numel():
            raise ValueError(f"vector epsilon has {e.numel()} entries for {self._buf.numel()} "
                             "environments")
        if isinstance(self._uploaded, torch.Tensor) and torch.equal(self._uploaded, e):
            return
        self._scalar_buf.fill_(float(e[0]))
        if self._buf is not None:
            self._buf.copy_(e)
        self._uploaded = e.clone()

    def _eps_for(self, n):
        """Epsilon operand for a batch of ``n`` environments."""
        e = self._epsilon
        vector = isinstance(e, torch.Tensor) and e.numel() > 1
        sl = self._env_slice
        if self._scalar_buf is None:                       # host path
            if vector and sl is not None and sl[1] - sl[0] == n:
                return e[sl[0]:sl[1]]
            if vector and e.numel() != n:
                raise ValueError(f"vector epsilon ({e.numel()} envs) cannot serve a batch of {n} "
                                 f"(selected envs: {sl})")
            return e
        if self._buf is not None:
            if sl is not None and sl[1] - sl[0] == n:
                return self._buf[sl[0]:sl[1]]
            if self._buf.numel() == n:
                return self._buf
        if vector:
            # never fall back to the first environment's epsilon silently
            raise ValueError(f"vector epsilon ({e.numel()} envs) cannot serve a batch of {n} "
                             f"(selected envs: {sl}); call select_envs(lo, hi) first")
        return self._scalar_buf

    def _fused(self, q, uniforms):
        """One-launch selection from the sampler's pre-drawn uniforms ``(u_all [T, n], t_dev)``: only
        with epsilon in its device buffers (``bind_device``) and 2-D float32 Q-values on that device."""
        if (uniforms is None or self._scalar_buf is None or q.dim() != 2 or not q.is_cuda
                or q.dtype != torch.float32 or uniforms[0].shape[-1] != q.shape[0]):
            return None
        from .. import ops
        return ops.eps_greedy(q, self._eps_for(q.shape[0]), uniforms[0], uniforms[1])

    def sample(self, q, generator=None, uniforms=None):
        fused = self._fused(q, uniforms)
        if fused is not None:
            return fused
        greedy = q.argmax(dim=-1)
        return _explore(greedy, q.shape[-1], self._eps_for(greedy.shape[0] if greedy.dim() else 1),
                        generator)


class CategoricalEpsilonGreedy(EpsilonGreedy):
    """For distributional Q-values ``p[..., A, n_atoms]`` over the atom grid ``z``: greedy with
    respect to the expected value."""

    def __init__(self, z=None, **kwargs):
        super().__init__(**kwargs)
        self.z = z

    def set_z(self, z):
        self.z = z

    def sample(self, p, z=None, generator=None, uniforms=None):
        expected = torch.tensordot(p, self.z if z is None else z, dims=1)
        fused = self._fused(expected, uniforms)
        if fused is not None:
            return fused
        greedy = expected.argmax(dim=-1)
        return _explore(greedy, expected.shape[-1],
                        self._eps_for(greedy.shape[0] if greedy.dim() else 1), generator)
