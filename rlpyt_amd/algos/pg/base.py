"""Policy-gradient base: optimizer set-up and ``process_returns`` on the device
(rlpyt/algos/pg/base.py:10-75)."""
from collections import namedtuple

import torch

from ... import ops
from ..base import RlAlgorithm

OptInfo = namedtuple("OptInfo", ["loss", "gradNorm", "entropy", "perplexity"])
AgentTrain = namedtuple("AgentTrain", ["dist_info", "value"])


class PolicyGradientAlgo(RlAlgorithm):
    bootstrap_value = True
    opt_info_fields = tuple(OptInfo._fields)
    scan_variant = ops.SCAN_EXACT  # bit-identical to the reference's CPU scan

    def initialize(self, agent, n_itr, batch_spec, mid_batch_reset=False, examples=None,
                   world_size=1, rank=0):
        if getattr(agent, "recurrent", False) and not getattr(self, "supports_recurrent", False):
            # refuse here, not in the middle of the first minibatch loop
            raise NotImplementedError(
                f"{type(self).__name__}: recurrent policy-gradient agents "
                "(rlpyt/agents/pg/categorical.py:54-106, rlpyt/algos/pg/ppo.py:84-86) are not "
                "built on this path yet; use a feed-forward agent (AtariFfAgent).")
        self.__dict__.update(agent=agent, n_itr=n_itr, batch_spec=batch_spec,
                             mid_batch_reset=mid_batch_reset, world_size=world_size, rank=rank)
        self.optimizer = self._fresh_optimizer(agent)

    def _fresh_optimizer(self, agent):
        """The constructor's optimizer over the agent's parameters, resumed from
        ``initial_optim_state_dict`` when one was given."""
        opt = self.make_optimizer(agent.parameters(), self.OptimCls, self.learning_rate,
                                  self.optim_kwargs)
        if self.initial_optim_state_dict is not None:
            opt.load_state_dict(self.initial_optim_state_dict)
        return opt

    def process_returns(self, samples):
        """(return_, advantage, valid) as HBM tensors: one fused scan launch computes
        GAE (or the discounted return when ``gae_lambda == 1``) together with the
        ``valid_from_done`` mask; advantage normalisation is a second, in-place pass."""
        mv = self.on_device
        reward, done = mv(samples.env.reward), mv(samples.env.done)
        value, bv = mv(samples.agent.agent_info.value), mv(samples.agent.bootstrap_value)
        want_valid = (not self.mid_batch_reset) or self.agent.recurrent
        if self.gae_lambda == 1:
            out = ops.discount_return(reward, done, bv, self.discount, value=value,
                                      with_valid=want_valid, variant=self.scan_variant)
            return_, advantage = out[0], out[1]
            valid = out[2] if want_valid else None
        else:
            out = ops.gae(reward, value, done, bv, self.discount, self.gae_lambda,
                          with_valid=want_valid, variant=self.scan_variant)
            advantage, return_ = out[0], out[1]
            valid = out[2] if want_valid else None
        if self.normalize_advantage:
            ops.normalize_advantage_(advantage, valid, eps=1e-6)
        return return_, advantage, valid
