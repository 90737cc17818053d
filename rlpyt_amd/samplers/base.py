"""Sampler protocol (rlpyt/samplers/base.py:7-67)."""
from ..utils.quick_args import save__init__args
from .collections import BatchSpec, TrajInfo


class BaseSampler:
    """Holds the configuration; subclasses implement initialize / obtain_samples /
    evaluate_agent / shutdown.  ``batch_spec``, ``batch_size`` and ``mid_batch_reset`` are
    read by the runner and the algorithm."""

    alternating = False

    def __init__(self, EnvCls, env_kwargs, batch_T, batch_B, CollectorCls=None,
                 max_decorrelation_steps=100, TrajInfoCls=TrajInfo, eval_n_envs=0,
                 eval_CollectorCls=None, eval_env_kwargs=None, eval_max_steps=None,
                 eval_max_trajectories=None):
        eval_max_steps = None if eval_max_steps is None else int(eval_max_steps)
        eval_max_trajectories = (None if eval_max_trajectories is None else
                                 int(eval_max_trajectories))
        save__init__args(locals())
        self.batch_spec = BatchSpec(batch_T, batch_B)
        self.mid_batch_reset = getattr(CollectorCls, "mid_batch_reset", True)

    def initialize(self, *args, **kwargs):
        raise NotImplementedError

    def obtain_samples(self, itr):
        raise NotImplementedError

    def evaluate_agent(self, itr):
        raise NotImplementedError

    def shutdown(self):
        pass

    @property
    def batch_size(self):
        return self.batch_spec.size
