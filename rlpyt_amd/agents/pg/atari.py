"""Atari policy-gradient agents (rlpyt/agents/pg/atari.py:21-37)."""
from ...models.pg.atari_ff_model import AtariFfModel
from ...models.pg.mlp_pg_model import MlpPgModel
from .categorical import CategoricalPgAgent


class AtariMixin:
    def make_env_to_model_kwargs(self, env_spaces):
        return dict(image_shape=env_spaces.observation.shape,
                    output_size=env_spaces.action.n)


class AtariFfAgent(AtariMixin, CategoricalPgAgent):
    def __init__(self, ModelCls=AtariFfModel, **kwargs):
        super().__init__(ModelCls=ModelCls, **kwargs)


class MlpCategoricalPgAgent(CategoricalPgAgent):
    """Vector-observation categorical agent (BASELINE config #1 plumbing)."""

    def __init__(self, ModelCls=MlpPgModel, **kwargs):
        super().__init__(ModelCls=ModelCls, **kwargs)

    def make_env_to_model_kwargs(self, env_spaces):
        return dict(observation_shape=env_spaces.observation.shape,
                    output_size=env_spaces.action.n)
