"""rlpyt_amd -- MI355X (gfx950) native hot path for astooke/rlpyt.

Scope: rollout-collection buffers, advantage/return scans, PPO/A2C
minibatch loss + update, prioritized-replay sum tree and frame gathers, DQN loss -- as
hand-written HIP kernels behind the C ABI of ``include/rlpyt_hip.h`` -- plus the host-side
mirror of the reference's Sampler / Algo / Agent / ReplayBuffer protocol that drives them.
"""
__version__ = "0.1.0"
