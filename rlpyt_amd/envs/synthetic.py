"""Synthetic host-side environments for plumbing tests and benches.

No ALE/gym in this image, so "Atari Pong" is ``SyntheticPong``:
an Atari-shaped env -- observation ``uint8[4,104,80]`` frame stack, 6 discrete actions,
reward in {-1,0,+1}, ``EnvInfo(game_score, traj_done)`` exactly like
rlpyt/envs/atari/atari_env.py:14,91-93 -- with real (if tiny) Pong dynamics so that a
policy can learn on it.  ``step_cost_us`` busy-waits to emulate a declared emulator cost.
"""
import time
from collections import namedtuple

import numpy as np

from ..spaces import FloatBox, IntBox
from . import Env, EnvStep, FrameStack

AtariEnvInfo = namedtuple("AtariEnvInfo", ["game_score", "traj_done"])  # typename = module attribute: picklable

H, W = 104, 80


class SyntheticPong(Env):
    """Pong-like: agent paddle on the right column, scripted opponent on the left.

    Actions (ALE Pong's 6): 0,1 = stay; 2,4 = up; 3,5 = down."""

    # the observation is the last ``num_img_obs`` frames, newest last, shifted by one per step
    # (like rlpyt/envs/atari/atari_env.py:115-118): samplers may upload only obs[-1]
    obs_newest_frame_last = True

    def __init__(self, num_img_obs=4, points_to_end=3, max_steps=2000, step_cost_us=0.,
                 opponent_skill=0.6, seed=0, step_cost_ref=None, frozen=False):
        self._n = num_img_obs
        self._observation_space = IntBox(0, 256, shape=(num_img_obs, H, W), dtype="uint8")
        self._action_space = IntBox(0, 6)
        self._stack = FrameStack(num_img_obs, H, W)
        self._points_to_end = points_to_end
        self._max_steps = max_steps
        self._cost = step_cost_us * 1e-6
        # optional fork-shared object with a ``.value`` in microseconds (e.g. mp.RawValue("d")):
        # lets a bench change the declared emulator cost of already forked env workers
        self._cost_ref = step_cost_ref
        self._skill = opponent_skill
        # diagnostics only (bench --frozen-env): step() returns the standing observation with no
        # dynamics and no drawing -- the framework's own per-env-step floor
        self._frozen = bool(frozen)
        self._rng = np.random.RandomState(seed)
        self.reset()

    def seed(self, seed):
        self._rng = np.random.RandomState(seed)

    @property
    def horizon(self):
        return self._max_steps

    def _serve(self):
        self._bx, self._by = W / 2., float(self._rng.randint(10, H - 10))
        self._vx = 2. if self._rng.rand() < 0.5 else -2.
        self._vy = float(self._rng.choice([-1.5, -0.75, 0.75, 1.5]))

    def reset(self):
        self._py = self._oy = H / 2.
        self._points = 0
        self._score = 0.
        self._steps = 0
        self._serve()
        self._draw(self._stack.newest)
        self._stack.fill()
        return self._stack.observation()

    def _draw(self, f):
        f[:] = 0
        p, o = int(self._py), int(self._oy)
        f[max(p - 6, 0):p + 6, W - 4:W - 2] = 200
        f[max(o - 6, 0):o + 6, 2:4] = 120
        y, x = int(self._by), int(self._bx)
        f[max(y - 1, 0):y + 2, max(x - 1, 0):x + 2] = 255

    def step(self, action):
        cost = self._cost if self._cost_ref is None else self._cost_ref.value * 1e-6
        if cost:
            end = time.perf_counter() + cost
            while time.perf_counter() < end:
                pass
        if self._frozen:
            return EnvStep(self._stack.frames, np.float32(0.), False,
                           AtariEnvInfo(game_score=0., traj_done=False))
        a = int(action)
        if a in (2, 4):
            self._py = max(6., self._py - 3.)
        elif a in (3, 5):
            self._py = min(H - 6., self._py + 3.)
        # scripted opponent follows the ball with limited speed and attention
        if self._rng.rand() < self._skill:
            self._oy += min(2., max(-2., self._by - self._oy))
        self._bx += self._vx
        self._by += self._vy
        if self._by < 1. or self._by > H - 2.:
            self._vy = -self._vy
            self._by = min(H - 2., max(1., self._by))
        reward = 0.
        if self._bx >= W - 5:
            if abs(self._by - self._py) <= 7.:
                self._vx = -abs(self._vx)
                self._vy += 0.25 * (self._by - self._py) / 7.
            else:
                reward = -1.
        elif self._bx <= 4:
            if abs(self._by - self._oy) <= 7.:
                self._vx = abs(self._vx)
            else:
                reward = 1.
        if reward != 0.:
            self._points += 1
            self._score += reward
            self._serve()
        self._steps += 1
        self._draw(self._stack.push())
        done = self._points >= self._points_to_end or self._steps >= self._max_steps
        info = AtariEnvInfo(game_score=reward, traj_done=done)
        return EnvStep(self._stack.observation(), np.float32(reward), done, info)


class TinyDiscreteEnv(Env):
    """The "CartPole-like" plumbing env of BASELINE.json config #1: a 1-D chain of length
    ``size``; actions {left, right}; +1 at the right end, episode ends at either end or
    after ``horizon`` steps.  Observation: float32[3] = (position/size, last move, bias)."""

    def __init__(self, size=9, horizon=40, seed=0):
        self._size, self._horizon = size, horizon
        self._observation_space = FloatBox(-1., 1., shape=(3,))
        self._action_space = IntBox(0, 2)
        self._rng = np.random.RandomState(seed)
        self.reset()

    def seed(self, seed):
        self._rng = np.random.RandomState(seed)

    @property
    def horizon(self):
        return self._horizon

    def _ob(self, move):
        return np.array([self._pos / self._size * 2 - 1, move, 1.], dtype=np.float32)

    def reset(self):
        self._pos = self._size // 2 + int(self._rng.randint(-1, 2))
        self._t = 0
        return self._ob(0.)

    def step(self, action):
        move = 1 if int(action) == 1 else -1
        self._pos += move
        self._t += 1
        reward, done = 0., False
        if self._pos >= self._size:
            reward, done = 1., True
        elif self._pos <= 0:
            reward, done = -0.1, True
        elif self._t >= self._horizon:
            done = True
        return EnvStep(self._ob(float(move)), np.float32(reward), done, ())
