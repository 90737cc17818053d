"""The duck-typed protocol of the hot path as a table shared by
``make_golden.py`` (records the REFERENCE signatures into ``protocol.json``) and
``tests/test_protocol.py`` (holds this repo's classes to them).

reference dotted path -> (this repo's dotted path, members the runner / siblings touch)."""

_SAMPLER = ["__init__", "initialize", "obtain_samples", "evaluate_agent", "shutdown"]
_ALGO = ["__init__", "initialize", "optimize_agent", "optim_state_dict", "bootstrap_value",
         "opt_info_fields"]
_PG_AGENT = ["__init__", "initialize", "to_device", "data_parallel", "sample_mode", "train_mode",
             "eval_mode", "step", "value", "reset", "reset_one", "__call__", "parameters",
             "state_dict", "load_state_dict", "collector_initialize", "sync_shared_memory"]
_DQN_AGENT = [m for m in _PG_AGENT if m != "value"] + ["target", "update_target"]
_REPLAY = ["__init__", "append_samples", "sample_batch"]
_PRI_REPLAY = _REPLAY + ["update_batch_priorities", "set_beta"]

PROTOCOL = {
    # samplers (runners/minibatch_rl.py:74-96,257,329,133)
    "rlpyt.samplers.parallel.gpu.sampler.GpuSampler": ("rlpyt_amd.samplers.gpu.GpuSampler", _SAMPLER),
    "rlpyt.samplers.parallel.gpu.alternating_sampler.AlternatingSampler":
        ("rlpyt_amd.samplers.alternating.AlternatingSampler", _SAMPLER),
    "rlpyt.samplers.parallel.gpu.alternating_sampler.NoOverlapAlternatingSampler":
        ("rlpyt_amd.samplers.alternating.NoOverlapAlternatingSampler", _SAMPLER),
    # algorithms (minibatch_rl.py:88-96,259,144,78,124)
    "rlpyt.algos.pg.ppo.PPO": ("rlpyt_amd.algos.pg.ppo.PPO", _ALGO),
    "rlpyt.algos.pg.a2c.A2C": ("rlpyt_amd.algos.pg.a2c.A2C", _ALGO),
    "rlpyt.algos.dqn.dqn.DQN": ("rlpyt_amd.algos.dqn.dqn.DQN", _ALGO),
    "rlpyt.algos.dqn.cat_dqn.CategoricalDQN": ("rlpyt_amd.algos.dqn.cat_dqn.CategoricalDQN", _ALGO),
    "rlpyt.algos.dqn.r2d1.R2D1": ("rlpyt_amd.algos.dqn.r2d1.R2D1", _ALGO),
    # agents (samplers/parallel/gpu/sampler.py:78-79, action_server.py:53-65, ppo.py:133, ...)
    "rlpyt.agents.pg.atari.AtariFfAgent": ("rlpyt_amd.agents.pg.atari.AtariFfAgent", _PG_AGENT),
    "rlpyt.agents.dqn.atari.atari_dqn_agent.AtariDqnAgent":
        ("rlpyt_amd.agents.dqn.dqn_agent.AtariDqnAgent", _DQN_AGENT),
    "rlpyt.agents.dqn.atari.atari_catdqn_agent.AtariCatDqnAgent":
        ("rlpyt_amd.agents.dqn.catdqn_agent.AtariCatDqnAgent", _DQN_AGENT),
    "rlpyt.agents.dqn.atari.atari_r2d1_agent.AtariR2d1Agent":
        ("rlpyt_amd.agents.dqn.r2d1_agent.AtariR2d1Agent", _DQN_AGENT),
    # replay buffers (algos/dqn/dqn.py:133-156,169-182,279)
    "rlpyt.replays.non_sequence.frame.UniformReplayFrameBuffer":
        ("rlpyt_amd.replays.buffers.UniformReplayFrameBuffer", _REPLAY),
    "rlpyt.replays.non_sequence.frame.PrioritizedReplayFrameBuffer":
        ("rlpyt_amd.replays.buffers.PrioritizedReplayFrameBuffer", _PRI_REPLAY),
    "rlpyt.replays.non_sequence.uniform.UniformReplayBuffer":
        ("rlpyt_amd.replays.buffers.UniformReplayBuffer", _REPLAY),
    "rlpyt.replays.non_sequence.prioritized.PrioritizedReplayBuffer":
        ("rlpyt_amd.replays.buffers.PrioritizedReplayBuffer", _PRI_REPLAY),
    "rlpyt.replays.sequence.frame.UniformSequenceReplayFrameBuffer":
        ("rlpyt_amd.replays.buffers.UniformSequenceReplayFrameBuffer", _REPLAY),
    "rlpyt.replays.sequence.frame.PrioritizedSequenceReplayFrameBuffer":
        ("rlpyt_amd.replays.buffers.PrioritizedSequenceReplayFrameBuffer", _PRI_REPLAY),
    # runners (the classes under which the path drops in)
    "rlpyt.runners.minibatch_rl.MinibatchRl": ("rlpyt_amd.runners.minibatch_rl.MinibatchRl",
                                               ["__init__", "train"]),
    "rlpyt.runners.minibatch_rl.MinibatchRlEval": ("rlpyt_amd.runners.minibatch_rl.MinibatchRlEval",
                                                   ["__init__", "train"]),
    # function seams (algos/utils.py:8-112)
    "rlpyt.algos.utils.discount_return": ("rlpyt_amd.algos.utils.discount_return", []),
    "rlpyt.algos.utils.generalized_advantage_estimation":
        ("rlpyt_amd.algos.utils.generalized_advantage_estimation", []),
    "rlpyt.algos.utils.discount_return_n_step": ("rlpyt_amd.algos.utils.discount_return_n_step", []),
    "rlpyt.algos.utils.valid_from_done": ("rlpyt_amd.algos.utils.valid_from_done", []),
}
