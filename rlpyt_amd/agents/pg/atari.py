"""Atari policy-gradient agents (rlpyt/agents/pg/atari.py:21-37)."""
import torch

from ...models.pg.atari_ff_model import AtariFfModel
from ...models.pg.mlp_pg_model import MlpPgModel
from ...models.pg.atari_lstm_model import AtariLstmModel
from .categorical import CategoricalPgAgent, RecurrentCategoricalPgAgent


class AtariMixin:
    def make_env_to_model_kwargs(self, env_spaces):
        return dict(image_shape=env_spaces.observation.shape,
                    output_size=env_spaces.action.n)

    def to_device(self, cuda_idx=None):
        super().to_device(cuda_idx)
        if cuda_idx is not None and not getattr(self.model, "fused_conv", False):
            # MIOpen path: weights in channels-last so its NHWC fp32 kernels run without
            # transposes (the hand-written conv stack takes torch's plain [co,c,ky,kx])
            self.model.to(memory_format=torch.channels_last)
            if hasattr(self, "target_model"):
                self.target_model.to(memory_format=torch.channels_last)

    def gather_observation(self, observation, flat_idx):
        """Minibatch rows of a [T,B,C,H,W] uint8 batch, delivered as the conv stack's input.
        Fused conv stack: just the (batch, indices) pair -- the kernels gather while staging.
        MIOpen path: gather + uint8->float32/255 + NHWC in one kernel."""
        if observation.is_cuda and observation.dtype == torch.uint8 and observation.dim() == 5:
            if getattr(self.sampling_model, "fused_conv", False):
                from ...models.pg.atari_ff_model import ObsGather
                return ObsGather(observation, flat_idx)
            from ... import ops
            return ops.obs_to_nhwc_f32(observation, flat_idx)
        return super().gather_observation(observation, flat_idx)


class AtariFfAgent(AtariMixin, CategoricalPgAgent):
    def __init__(self, ModelCls=AtariFfModel, **kwargs):
        super().__init__(ModelCls=ModelCls, **kwargs)


class AtariLstmAgent(AtariMixin, RecurrentCategoricalPgAgent):
    """rlpyt/agents/pg/atari.py:27-30."""

    def __init__(self, ModelCls=AtariLstmModel, **kwargs):
        super().__init__(ModelCls=ModelCls, **kwargs)


class MlpCategoricalPgAgent(CategoricalPgAgent):
    """Vector-observation categorical agent (BASELINE config #1 plumbing)."""

    def __init__(self, ModelCls=MlpPgModel, **kwargs):
        super().__init__(ModelCls=ModelCls, **kwargs)

    def make_env_to_model_kwargs(self, env_spaces):
        return dict(observation_shape=env_spaces.observation.shape,
                    output_size=env_spaces.action.n)
