"""What a runner needs from a sampler: the constructor arguments of the reference's samplers
(rlpyt/samplers/base.py:7-67 -- kept by name, they arrive from experiment configs), the batch
geometry, and four calls: ``initialize -> examples``, ``obtain_samples(itr) -> (samples,
traj_infos)``, ``evaluate_agent(itr) -> traj_infos``, ``shutdown()``."""
from .collections import BatchSpec, TrajInfo


class BaseSampler:
    alternating = False      # one agent, one model call per step (no alternating halves)

    def __init__(self, EnvCls, env_kwargs, batch_T, batch_B, CollectorCls=None,
                 max_decorrelation_steps=100, TrajInfoCls=TrajInfo, eval_n_envs=0,
                 eval_CollectorCls=None, eval_env_kwargs=None, eval_max_steps=None,
                 eval_max_trajectories=None):
        if int(batch_T) < 1 or int(batch_B) < 1:
            raise ValueError(f"sampler batch must be at least [1, 1], got [{batch_T}, {batch_B}]")
        self.EnvCls, self.env_kwargs = EnvCls, dict(env_kwargs or {})
        self.batch_spec = BatchSpec(int(batch_T), int(batch_B))
        self.batch_T, self.batch_B = self.batch_spec
        self.max_decorrelation_steps = int(max_decorrelation_steps)
        self.TrajInfoCls = TrajInfoCls
        # collector classes only select the reset mode here (the collectors' work is done by the
        # sampler's own env runners): anything with ``mid_batch_reset = False`` means wait-reset
        self.CollectorCls, self.eval_CollectorCls = CollectorCls, eval_CollectorCls
        self.mid_batch_reset = bool(getattr(CollectorCls, "mid_batch_reset", True))
        # offline evaluation
        self.eval_n_envs = int(eval_n_envs or 0)
        self.eval_env_kwargs = eval_env_kwargs
        self.eval_max_steps = None if eval_max_steps is None else int(eval_max_steps)
        self.eval_max_trajectories = (None if eval_max_trajectories is None
                                      else int(eval_max_trajectories))

    @property
    def batch_size(self):
        """Env steps per ``obtain_samples`` call on this rank."""
        return self.batch_spec.size

    # The reference's own samplers disagree on the tail of this signature: SerialSampler has
    # (..., rank=0, world_size=1) (rlpyt/samplers/serial/sampler.py:24-33), ParallelSamplerBase
    # has (..., world_size=1, rank=0, worker_process=None) (rlpyt/samplers/parallel/base.py:28-38)
    # and BaseSampler takes (*args, **kwargs) (rlpyt/samplers/base.py:49).  The base class here
    # takes the same open signature, and subclasses make everything after ``seed`` keyword-only
    # in effect by following THEIR reference class (tests/golden/protocol.json pins both).
    def initialize(self, *args, **kwargs):
        raise NotImplementedError(f"{type(self).__name__}.initialize")

    def obtain_samples(self, itr):
        raise NotImplementedError(f"{type(self).__name__}.obtain_samples")

    def evaluate_agent(self, itr):
        raise NotImplementedError(f"{type(self).__name__}.evaluate_agent")

    def shutdown(self):
        """Stop worker processes and release pinned memory, if any."""
