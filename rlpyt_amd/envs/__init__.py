"""Environments step on host cores, exactly as in the reference; what the samplers rely on is the
reference's environment contract (rlpyt/envs/base.py:5-65) -- ``reset() -> observation``,
``step(action) -> EnvStep(observation, reward, done, env_info)``, ``spaces``, ``horizon`` -- plus
ONE optional declaration that matters on this hardware:

``obs_newest_frame_last = True`` (class attribute) says that the observation is the last ``C``
frames, newest last, shifted by one frame per step (true for rlpyt's AtariEnv,
envs/atari/atari_env.py:115-118).  The HBM-resident sampler then uploads only the newest frame
of every env per step and rebuilds the stack on the device (``rlpyt_frame_push`` /
``rlpyt_atari_sample_convs_f32``); envs without the attribute upload whole observations.
``FrameStack`` below is the host half of that convention."""
from collections import namedtuple

import numpy as np

from ..spaces import EnvSpaces

EnvStep = namedtuple("EnvStep", ["observation", "reward", "done", "env_info"])
EnvInfo = namedtuple("EnvInfo", [])     # envs define their own fields


class Env:
    """Base class of the synthetic environments of this repo; any object with the same methods
    works (the samplers never check the type)."""

    _action_space = _observation_space = None

    def reset(self):
        raise NotImplementedError(f"{type(self).__name__}.reset")

    def step(self, action):
        raise NotImplementedError(f"{type(self).__name__}.step")

    def seed(self, seed):
        """Re-seed the env's private RNG (samplers call it with ``seed + global env index``)."""

    def close(self):
        """Release emulator resources, if any."""

    @property
    def horizon(self):
        raise NotImplementedError(f"{type(self).__name__}.horizon")

    action_space = property(lambda self: self._action_space)
    observation_space = property(lambda self: self._observation_space)
    spaces = property(lambda self: EnvSpaces(observation=self._observation_space,
                                             action=self._action_space))


class FrameStack:
    """``[C, H, W]`` uint8 stack with the newest frame last: ``push`` shifts by one frame and
    returns the slot to draw the new frame into; ``fill`` repeats one frame over the whole stack
    (episode start)."""

    def __init__(self, n_frames, height, width):
        self.frames = np.zeros((n_frames, height, width), dtype=np.uint8)

    @property
    def newest(self):
        return self.frames[-1]

    def push(self):
        f = self.frames
        f[:-1] = f[1:]
        return f[-1]

    def fill(self):
        f = self.frames
        f[:-1] = f[-1]

    def observation(self):
        return self.frames.copy()
