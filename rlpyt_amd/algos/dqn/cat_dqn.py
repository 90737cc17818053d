"""Categorical (distributional) DQN on the MI355X (rlpyt/algos/dqn/cat_dqn.py:11-93).

Everything after the three network passes -- the Bellman-shifted atom grid, the [B,P,P']
projection, cross-entropy, IS weighting, KL priorities and dLoss/dp -- is one HIP kernel
(``csrc/catdqn.hip``); the projection tensor is never materialised."""
import torch

from ... import ops
from ...algos.utils import valid_from_done
from .dqn import DQN


class CategoricalDQN(DQN):
    def __init__(self, V_min=-10, V_max=10, **kwargs):
        super().__init__(**kwargs)
        self.V_min = V_min
        self.V_max = V_max
        if "eps" not in self.optim_kwargs:  # assume Adam (cat_dqn.py:22-23)
            self.optim_kwargs["eps"] = 0.01 / self.batch_size

    def initialize(self, *args, **kwargs):
        super().initialize(*args, **kwargs)
        self.agent.give_V_min_max(self.V_min, self.V_max)
        self._z = None

    def loss(self, samples):
        """Returns (loss, KL_div) -- KL_div feeds the replay priorities (cat_dqn.py:34-93)."""
        if getattr(self, "_z", None) is None or self._z.device != self.agent.device:
            self._z = torch.linspace(self.V_min, self.V_max, self.agent.n_atoms,
                                     device=self.agent.device)
        with torch.no_grad():
            target_ps = self.agent.target(*samples.target_inputs)
            next_ps = self.agent(*samples.target_inputs) if self.double_dqn else None
        ps = self.agent(*samples.agent_inputs)
        is_weights = samples.is_weights if self.prioritized_replay else None
        valid = None if self.mid_batch_reset else valid_from_done(samples.done)
        return ops.cat_dqn_loss(ps, target_ps, next_ps, samples.action, samples.return_,
                                samples.done_n, is_weights, valid, self._z, self.V_min,
                                self.V_max, self.discount ** self.n_step_return)
