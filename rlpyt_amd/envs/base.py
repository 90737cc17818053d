"""Environment protocol (rlpyt/envs/base.py:5-65): ``step`` -> EnvStep(observation,
reward, done, env_info), ``reset`` -> observation, ``spaces``, ``horizon``."""
from collections import namedtuple

from ..spaces import EnvSpaces

EnvStep = namedtuple("EnvStep", ["observation", "reward", "done", "env_info"])
EnvInfo = namedtuple("EnvInfo", [])


class Env:
    def step(self, action):
        raise NotImplementedError

    def reset(self):
        raise NotImplementedError

    def seed(self, seed):
        pass

    @property
    def action_space(self):
        return self._action_space

    @property
    def observation_space(self):
        return self._observation_space

    @property
    def spaces(self):
        return EnvSpaces(observation=self.observation_space, action=self.action_space)

    @property
    def horizon(self):
        raise NotImplementedError

    def close(self):
        pass
