"""Sampler-parity cases shared by ``make_golden.py`` (which runs the REFERENCE GpuSampler with
its GPU collectors and action server, on CPU) and ``tests/test_sampler_parity.py`` (which runs
this repo's GpuSampler): a deterministic policy that depends on the observation AND on the
previous action / reward it is handed (so the null-after-reset rules are visible in the data),
and the reference's per-worker env seeding."""
import zlib

import numpy as np
import torch

ENV_KWARGS = dict(points_to_end=1, max_steps=25)
B, N_WORKERS, SEED = 4, 2, 7
# (name, collector mode, batch_T, batches): episodes last 18 steps (a point) or 25 (time limit), so
# T=5 puts the dones inside batches and T=6 on the last step of a batch
CASES = [("reset", "reset", 5, 11), ("wait", "wait", 5, 11),
         ("reset_t6", "reset", 6, 7), ("wait_t6", "wait", 6, 7)]


def det_policy(obs, prev_action, prev_reward, n):
    """(action, value) for uint8 image observations with any leading dims."""
    lead = tuple(obs.shape[:-3])
    flat = obs.reshape(*lead, -1).long()
    w = (torch.arange(flat.shape[-1], device=obs.device) % 13) + 1
    s = (flat * w).sum(-1)
    a = (s + 3 * prev_action.long() + (prev_reward * 2).long()) % n
    v = (s % 97).float() / 97 + 0.5 * prev_reward.float() + 0.1 * prev_action.float()
    return a, v


def reference_env_seed(base_seed, i, n_workers=N_WORKERS, batch_B=B):
    """Seed the reference gives env ``i``: worker w = i // per gets seed + w and seeds its k-th
    env with that + k (rlpyt/samplers/parallel/base.py:233, worker.py:51, utils/seed.py:54-61)."""
    per = batch_B // n_workers
    return base_seed + i // per + i % per


# ---- the reference's own AtariFfAgent under the reference GpuSampler (gen_sampler_ff) -----------
# The policy head's bias is zeroed (with the random bias one action wins on every observation) and
# its weight scaled by FF_PI_SCALE after initialisation: every softmax output is then exactly
# one-hot in float32 (top-2 logit gaps of the recorded run: FF_MIN_GAP_REQUIRED at
# least), so torch.multinomial on the reference side and the inverse-CDF draw of the fused head
# kernel pick the SAME action whatever their random streams do -- the recorded batches are a
# function of the observations and the parameters only, and the device chain (frame push + conv1 +
# conv2 -> trunk -> heads + softmax + draw + row writes) can be held to them field by field.
FF_INIT_SEED, FF_PI_SCALE, FF_T, FF_BATCHES, FF_MIN_GAP_REQUIRED = 11, 1.0e7, 5, 11, 300.0
FF_ENV_KWARGS = dict(points_to_end=1, max_steps=25)


def ff_sharpen(model):
    """Scale the policy head in place (both sides apply this to bit-identical parameters)."""
    with torch.no_grad():
        model.pi.weight.mul_(FF_PI_SCALE)
        model.pi.bias.zero_()


def param_checksums(params):
    """CRC32 of every parameter's bytes (exact: a floating-point sum depends on the thread count
    of the host that computes it)."""
    return np.array([zlib.crc32(np.ascontiguousarray(p.detach().cpu().numpy()).tobytes())
                     for p in params], dtype=np.uint32)


EVAL_N_ENVS, EVAL_MAX_STEPS = 4, 4 * 60      # 60 time steps of 4 eval envs
EVAL_ENV_KWARGS = dict(points_to_end=1, max_steps=21)


def obs_crc(observation):
    """uint32 CRC of every [t, b] observation (keeps the fixture small)."""
    o = np.ascontiguousarray(np.asarray(observation))
    out = np.zeros(o.shape[:2], dtype=np.uint32)
    for t in range(o.shape[0]):
        for b in range(o.shape[1]):
            out[t, b] = zlib.crc32(o[t, b].tobytes())
    return out


AgentInfo = None


def bind_agent_info(namedarraytuple):
    """``AgentInfo(value)`` built with the given framework's namedarraytuple and registered as an
    attribute of this module, so it survives the pickling the reference does between processes."""
    global AgentInfo
    AgentInfo = namedarraytuple("AgentInfo", ["value"])
    AgentInfo.__module__ = __name__
    return AgentInfo
