"""rlpyt_amd -- MI355X (gfx950) native hot path for astooke/rlpyt.

Scope: rollout-collection buffers, advantage/return scans, PPO/A2C
minibatch loss + update, prioritized-replay sum tree and frame gathers, DQN/R2D1 losses -- as
hand-written HIP kernels behind the C ABI of ``include/rlpyt_hip.h`` -- plus the host-side
mirror of the reference's Sampler / Algo / Agent / ReplayBuffer protocol that drives them.
"""
import os

# The conv stack runs in channels-last storage (see models/pg/atari_ff_model.py); this asks
# PyTorch-ROCm to hand NHWC tensors to MIOpen as they are instead of transposing to NCHW.
os.environ.setdefault("PYTORCH_MIOPEN_SUGGEST_NHWC", "1")

__version__ = "0.1.0"
