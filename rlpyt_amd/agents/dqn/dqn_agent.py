"""DQN agents (rlpyt/agents/dqn/dqn_agent.py:18-81, epsilon_greedy.py:12-131)."""
import torch

from ...distributions.epsilon_greedy import EpsilonGreedy
from ...models.dqn.atari_dqn_model import AtariDqnModel
from ...models.utils import update_state_dict
from ...utils import logger
from ...utils.collections import namedarraytuple
from ..base import AgentStep, BaseAgent

AgentInfo = namedarraytuple("AgentInfo", "q")


class EpsilonSchedule:
    """Exploration rate as a function of the iteration: linear from ``init`` to ``final`` between
    ``itr_min`` and ``itr_max``, held afterwards; ``evaluation`` rate separately (fully random at
    iteration 0, before anything was learned).  ``final`` becomes a per-environment vector when a
    ``final_min`` is given: one log-spaced ladder over the GLOBAL environment index, of which every
    rank takes the rungs of its own envs (values of rlpyt/agents/dqn/epsilon_greedy.py:47-63,
    96-106,120-131; pinned by tests/golden/agents.npz)."""

    def __init__(self, init, final, final_min, itr_min, itr_max, eval_eps):
        self.init, self.final, self.final_min = init, final, final_min
        self.itr_min, self.itr_max, self.eval_eps = itr_min, itr_max, eval_eps
        self._scalar = (init, final)
        self.current = init

    def spread_over_envs(self, global_B, env_ranks):
        init0, final0 = self._scalar
        if self.final_min is not None and self.final_min != final0:
            ladder = torch.logspace(torch.log10(torch.tensor(self.final_min)),
                                    torch.log10(torch.tensor(final0)), global_B)
            self.init, self.final = init0 * torch.ones(len(env_ranks)), ladder[env_ranks]
        self.current = self.init

    def sampling(self, itr):
        if itr <= self.itr_max:
            frac = min(1, max(0, itr - self.itr_min) / (self.itr_max - self.itr_min))
            self.current = frac * self.final + (1 - frac) * self.init
        return self.current

    def evaluation(self, itr):
        return self.eval_eps if itr > 0 else 1.


def _schedule_field(name):
    return property(lambda self: getattr(self.eps, name),
                    lambda self, v: setattr(self.eps, name, v))


class EpsilonGreedyAgentMixin:
    """Agents that act epsilon-greedily while sampling: owns an ``EpsilonSchedule`` (its fields are
    readable / settable on the agent under the reference's attribute names) and applies it whenever
    the agent enters sample or eval mode."""

    def __init__(self, eps_init=1, eps_final=0.01, eps_final_min=None, eps_itr_min=50,
                 eps_itr_max=1000, eps_eval=0.001, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.eps = EpsilonSchedule(eps_init, eps_final, eps_final_min, eps_itr_min, eps_itr_max,
                                   eps_eval)

    eps_init, eps_final = _schedule_field("init"), _schedule_field("final")
    eps_itr_min, eps_itr_max = _schedule_field("itr_min"), _schedule_field("itr_max")
    eps_eval, eps_sample = _schedule_field("eval_eps"), _schedule_field("current")
    eps_final_min = _schedule_field("final_min")     # (the reference keeps it on the agent too)

    def collector_initialize(self, global_B=1, env_ranks=None):
        if env_ranks is not None:
            self.eps.spread_over_envs(global_B, env_ranks)

    def set_epsilon_itr_min_max(self, eps_itr_min, eps_itr_max):
        logger.log(f"Agent: epsilon anneals between iterations {eps_itr_min} and {eps_itr_max}.")
        self.eps.itr_min, self.eps.itr_max = eps_itr_min, eps_itr_max

    def set_sample_epsilon_greedy(self, epsilon):
        self.distribution.set_epsilon(epsilon)

    def select_envs(self, lo=None, hi=None):
        """The sampler is about to step environments [lo, hi) of this rank (one pipeline group):
        vector epsilon is sliced accordingly."""
        self.distribution.select_envs(lo, hi)

    def _enter(self, mode, itr):
        super()._enter(mode, itr)
        if self._mode_repeat:          # the sampler's echo of the runner's call: epsilon is set already
            return
        if mode == "sample":
            self.distribution.set_epsilon(self.eps.sampling(itr))
        elif mode == "eval":
            self.distribution.set_epsilon(self.eps.evaluation(itr))


class DqnAgent(EpsilonGreedyAgentMixin, BaseAgent):
    def _onehot(self, prev_action, model=None):
        """``distribution.to_onehot(prev_action)`` (rlpyt/agents/dqn/dqn_agent.py:28) -- unless the model
        declares that its forward never reads the previous action / reward (``uses_prev_inputs = False``,
        the attribute the HBM sampler reads too: it then passes ``None`` for both)."""
        model = self.model if model is None else model
        if prev_action is None or not getattr(getattr(model, "module", model), "uses_prev_inputs", True):
            return prev_action
        return self.distribution.to_onehot(prev_action)

    def __call__(self, observation, prev_action, prev_reward):
        prev_action = self._onehot(prev_action)
        obs, pa, pr = self._to_model_device(observation, prev_action, prev_reward)
        return self._out(self.model(obs, pa, pr))

    def initialize(self, env_spaces, share_memory=False, global_B=1, env_ranks=None):
        """Online + target network.  An ``initial_model_state_dict`` is a ``{"model": ...}`` record
        and seeds BOTH; without one the target keeps its OWN random initialisation until the first
        target update, as in the reference (dqn_agent.py:39-43) -- pinned by the reference's DQN
        iterations in tests/golden/dqn_iterations.npz."""
        seed_sd, self.initial_model_state_dict = self.initial_model_state_dict, None
        super().initialize(env_spaces, share_memory, global_B=global_B, env_ranks=env_ranks)
        self.initial_model_state_dict = seed_sd
        self.target_model = self._new_model()
        if seed_sd is not None:
            for net in (self.model, self.target_model):
                net.load_state_dict(seed_sd["model"])
        self._n_local_envs = None if env_ranks is None else len(env_ranks)
        self.collector_initialize(global_B, env_ranks)

    def make_distribution(self, n_actions):
        return EpsilonGreedy(dim=n_actions)

    def to_device(self, cuda_idx=None):
        super().to_device(cuda_idx)
        self.target_model.to(self.device)
        # epsilon in persistent device buffers: visible to step graphs captured earlier
        self.distribution.bind_device(self.device, getattr(self, "_n_local_envs", None))
        self.distribution.set_epsilon(self.eps.current)

    @property
    def supports_sample_uniforms(self):
        """The sampler pre-draws one uniform per environment and step; the epsilon-greedy choice is
        then one launch (``rlpyt_eps_greedy_f32``) instead of argmax + two draws + compare + where."""
        return self.device.type == "cuda"

    def state_dict(self):
        return dict(model=self.model.state_dict(), target=self.target_model.state_dict())

    @torch.no_grad()
    def step(self, observation, prev_action, prev_reward):
        prev_action = self._onehot(prev_action)
        obs, pa, pr = self._to_model_device(observation, prev_action, prev_reward)
        q = self.model(obs, pa, pr)
        action = self.distribution.sample(q, generator=self.sample_generator,
                                          uniforms=self.sample_uniforms)
        return self._out(AgentStep(action=action, agent_info=AgentInfo(q=q)))

    def target(self, observation, prev_action, prev_reward):
        prev_action = self._onehot(prev_action, self.target_model)
        obs, pa, pr = self._to_model_device(observation, prev_action, prev_reward)
        return self._out(self.target_model(obs, pa, pr))

    def update_target(self, tau=1):
        update_state_dict(self.target_model, self.model.state_dict(), tau)


class AtariDqnAgent(DqnAgent):
    def __init__(self, ModelCls=AtariDqnModel, **kwargs):
        super().__init__(ModelCls=ModelCls, **kwargs)

    def make_env_to_model_kwargs(self, env_spaces):
        return dict(image_shape=env_spaces.observation.shape,
                    output_size=env_spaces.action.n)
