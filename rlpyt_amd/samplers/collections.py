"""Containers the samplers hand around.

* The ``[T, B]`` batch: ``Samples(agent=AgentSamples[Bsv], env=EnvSamples)`` with the reference's
  field names (rlpyt/samplers/collections.py:7-15) -- algorithms index them by name.  Here every
  leaf except ``env_info`` is an HBM tensor; ``action`` / ``prev_action`` and ``reward`` /
  ``prev_reward`` are the ``[1:]`` / ``[:-1]`` views of one ``[T+1, B]`` array each.
* The per-step hand-off between env workers and the device: ``StepBuffer`` (fork-shared,
  page-locked host arrays, one per pipeline group), ``StepBufferFs`` for frame-stacked envs that
  publish only their newest frame, and the two records an agent receives when it takes over the
  row writes of a step (``StepBinding``) and the frame-stack rebuild (``FramePush``).
* Per-trajectory statistics (``TrajInfo``), logged by the runners.
"""
from collections import namedtuple

from ..utils.collections import AttrDict, namedarraytuple

# ------------------------------------------------------------------------- the [T, B] batch
Samples = namedarraytuple("Samples", ["agent", "env"])
AgentSamples = namedarraytuple("AgentSamples", ["action", "prev_action", "agent_info"])
AgentSamplesBsv = namedarraytuple("AgentSamplesBsv",
                                  AgentSamples._fields + ("bootstrap_value",))
EnvSamples = namedarraytuple("EnvSamples",
                             ["observation", "reward", "prev_reward", "done", "env_info"])
_BatchSpec = namedtuple("BatchSpec", ["T", "B"])


class BatchSpec(_BatchSpec):
    """``T`` time steps of ``B`` environments per sampler batch; ``size`` = env steps in it."""
    __slots__ = ()
    size = property(lambda self: self.T * self.B)


# ------------------------------------------------------------------ per-step hand-off records
StepBuffer = namedarraytuple("StepBuffer", ["observation", "action", "reward", "done"])
# frame-stacked envs additionally publish the newest frame and a "stack was reset" flag
StepBufferFs = namedarraytuple("StepBufferFs", StepBuffer._fields + ("frame", "reset"))
# what an agent's ``step_into`` gets to write the rows of a step itself (BaseAgent.step_into)
StepBinding = namedtuple("StepBinding", ["action_rows", "agent_info_rows", "action_out",
                                         "uniforms", "t_dev", "lo", "push"])
# frame-stack rebuild of row t handed to the agent together with the step: the arguments of
# ``ops.frame_push`` minus the staging copy
FramePush = namedtuple("FramePush", ["obs", "new_frame", "full_rows", "slot", "scalar_rows"])


# ------------------------------------------------------------------ trajectory statistics
class TrajInfo(AttrDict):
    """Running statistics of one trajectory (fields and update rule of
    rlpyt/samplers/collections.py:30-56).  Keys that do not start with ``_`` are what the runners
    log; ``_discount`` is set on the class through the sampler's ``traj_info_kwargs``."""
    _discount = 1

    def __init__(self, **kwargs):
        super().__init__(Length=0, Return=0, NonzeroRewards=0, DiscountedReturn=0, **kwargs)
        self["_cur_discount"] = 1

    def step(self, observation, action, reward, done, agent_info, env_info):
        g = self["_cur_discount"]
        self["Length"] += 1
        self["Return"] += reward
        self["NonzeroRewards"] += reward != 0
        self["DiscountedReturn"] += g * reward
        self["_cur_discount"] = g * self._discount

    def terminate(self, observation):
        """Called with the final observation when the trajectory ends; returns the record."""
        return self


class AtariTrajInfo(TrajInfo):
    """Adds the raw (unclipped) game score (rlpyt/envs/atari/atari_env.py:24-30)."""

    def __init__(self, **kwargs):
        super().__init__(GameScore=0, **kwargs)

    def step(self, observation, action, reward, done, agent_info, env_info):
        TrajInfo.step(self, observation, action, reward, done, agent_info, env_info)
        self["GameScore"] += getattr(env_info, "game_score", 0)
