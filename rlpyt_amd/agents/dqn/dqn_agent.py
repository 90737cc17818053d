"""DQN agents (rlpyt/agents/dqn/dqn_agent.py:18-81, epsilon_greedy.py:12-131)."""
import torch

from ...distributions.epsilon_greedy import EpsilonGreedy
from ...models.dqn.atari_dqn_model import AtariDqnModel
from ...models.utils import update_state_dict
from ...utils import logger
from ...utils.collections import namedarraytuple
from ...utils.quick_args import save__init__args
from ..base import AgentStep, BaseAgent

AgentInfo = namedarraytuple("AgentInfo", "q")


class EpsilonGreedyAgentMixin:
    """Epsilon schedule: linear from eps_init to eps_final between eps_itr_min and
    eps_itr_max; optional log-spaced per-env vector epsilon (``eps_final_min``)."""

    def __init__(self, eps_init=1, eps_final=0.01, eps_final_min=None, eps_itr_min=50,
                 eps_itr_max=1000, eps_eval=0.001, *args, **kwargs):
        super().__init__(*args, **kwargs)
        save__init__args(locals())
        self._eps_final_scalar = eps_final
        self._eps_init_scalar = eps_init

    def collector_initialize(self, global_B=1, env_ranks=None):
        if env_ranks is not None:
            self.make_vec_eps(global_B, env_ranks)

    def make_vec_eps(self, global_B, env_ranks):
        if self.eps_final_min is not None and self.eps_final_min != self._eps_final_scalar:
            self.eps_init = self._eps_init_scalar * torch.ones(len(env_ranks))
            global_eps_final = torch.logspace(torch.log10(torch.tensor(self.eps_final_min)),
                                              torch.log10(torch.tensor(self._eps_final_scalar)),
                                              global_B)
            self.eps_final = global_eps_final[env_ranks]
        self.eps_sample = self.eps_init

    def set_epsilon_itr_min_max(self, eps_itr_min, eps_itr_max):
        logger.log(f"Agent setting min/max epsilon itrs: {eps_itr_min}, {eps_itr_max}")
        self.eps_itr_min, self.eps_itr_max = eps_itr_min, eps_itr_max

    def set_sample_epsilon_greedy(self, epsilon):
        self.distribution.set_epsilon(epsilon)

    def select_envs(self, lo=None, hi=None):
        """The sampler is about to step environments [lo, hi) of this rank (one pipeline group):
        vector epsilon is sliced accordingly."""
        self.distribution.select_envs(lo, hi)

    def sample_mode(self, itr):
        super().sample_mode(itr)
        if itr <= self.eps_itr_max:
            prog = min(1, max(0, itr - self.eps_itr_min) / (self.eps_itr_max - self.eps_itr_min))
            self.eps_sample = prog * self.eps_final + (1 - prog) * self.eps_init
        self.distribution.set_epsilon(self.eps_sample)

    def eval_mode(self, itr):
        super().eval_mode(itr)
        self.distribution.set_epsilon(self.eps_eval if itr > 0 else 1.)


class DqnAgent(EpsilonGreedyAgentMixin, BaseAgent):
    def __call__(self, observation, prev_action, prev_reward):
        prev_action = self.distribution.to_onehot(prev_action)
        obs, pa, pr = self._to_model_device(observation, prev_action, prev_reward)
        return self._out(self.model(obs, pa, pr))

    def initialize(self, env_spaces, share_memory=False, global_B=1, env_ranks=None):
        init_sd = self.initial_model_state_dict
        self.initial_model_state_dict = None
        super().initialize(env_spaces, share_memory, global_B=global_B, env_ranks=env_ranks)
        self.target_model = self.ModelCls(**self.env_model_kwargs, **self.model_kwargs)
        # as the reference (dqn_agent.py:39-43): without an initial state dict the target network
        # keeps its OWN random initialisation until the first target update -- pinned by the
        # reference's DQN iterations in tests/golden/dqn_iterations.npz
        if init_sd is not None:
            self.model.load_state_dict(init_sd["model"])
            self.target_model.load_state_dict(init_sd["model"])
        self.distribution = EpsilonGreedy(dim=env_spaces.action.n)
        self.eps_sample = self.eps_init
        self._n_local_envs = None if env_ranks is None else len(env_ranks)
        if env_ranks is not None:
            self.make_vec_eps(global_B, env_ranks)

    def to_device(self, cuda_idx=None):
        super().to_device(cuda_idx)
        self.target_model.to(self.device)
        # epsilon in persistent device buffers: visible to step graphs captured earlier
        self.distribution.bind_device(self.device, getattr(self, "_n_local_envs", None))
        self.distribution.set_epsilon(self.eps_sample)

    @property
    def supports_sample_uniforms(self):
        """The sampler pre-draws one uniform per environment and step; the epsilon-greedy choice is
        then one launch (``rlpyt_eps_greedy_f32``) instead of argmax + two draws + compare + where."""
        return self.device.type == "cuda"

    def state_dict(self):
        return dict(model=self.model.state_dict(), target=self.target_model.state_dict())

    @torch.no_grad()
    def step(self, observation, prev_action, prev_reward):
        prev_action = self.distribution.to_onehot(prev_action)
        obs, pa, pr = self._to_model_device(observation, prev_action, prev_reward)
        q = self.model(obs, pa, pr)
        action = self.distribution.sample(q, generator=self.sample_generator,
                                          uniforms=self.sample_uniforms)
        return self._out(AgentStep(action=action, agent_info=AgentInfo(q=q)))

    def target(self, observation, prev_action, prev_reward):
        prev_action = self.distribution.to_onehot(prev_action)
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
