"""CPU oracle for the rlpyt hot path -- TEST INFRASTRUCTURE ONLY.

A plain numpy / torch-CPU restatement of the reference's algorithms for the path named in
BASELINE.json.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this package; nothing under ``rlpyt_amd/``
does.  Each function cites the reference file:line it restates.

Parity pinning: the reference's own tests hold no numeric vectors for this path
(SURVEY.md section 4), so this oracle is pinned against outputs of the reference ITSELF,
imported from /root/reference in the build container by ``tests/golden/make_golden.py``
and committed as ``tests/golden/*.npz`` (checked by tests/test_oracle_golden.py).

Arithmetic notes: all tensor math is float32; Python scalars meeting
an array are rounded to float32 first; ``discount * gae_lambda`` is folded in float64 and
then rounded.  numpy float32 ops are IEEE single ops without FMA contraction, which is
exactly what the torch CPU path of the reference executes.
"""
import numpy as np

EPS_CAT = 1e-8  # rlpyt/distributions/categorical.py:9

f32 = np.float32


# --------------------------------------------------------------------------------------
# rlpyt/algos/utils.py
# --------------------------------------------------------------------------------------
def discount_return(reward, done, bootstrap_value, discount):
    """rlpyt/algos/utils.py:8-21.  reward f32 [T,...], done {0,1} [T,...], bv [...]."""
    reward = np.asarray(reward, dtype=f32)
    nd = (1 - np.asarray(done).astype(f32)).astype(f32)
    g = f32(discount)
    T = reward.shape[0]
    out = np.zeros_like(reward)
    bv = np.broadcast_to(np.asarray(bootstrap_value, dtype=f32).reshape(
        (-1,) + reward.shape[1:])[-1], reward.shape[1:])
    out[T - 1] = reward[T - 1] + (g * bv) * nd[T - 1]
    for t in range(T - 2, -1, -1):
        out[t] = reward[t] + (out[t + 1] * g) * nd[t]
    return out


def generalized_advantage_estimation(reward, value, done, bootstrap_value, discount,
                                     gae_lambda):
    """rlpyt/algos/utils.py:24-40 -> (advantage, return_)."""
    reward = np.asarray(reward, dtype=f32)
    value = np.asarray(value, dtype=f32)
    nd = (1 - np.asarray(done).astype(f32)).astype(f32)
    g = f32(discount)
    gl = f32(float(discount) * float(gae_lambda))  # folded in double first
    T = reward.shape[0]
    adv = np.zeros_like(reward)
    bv = np.broadcast_to(np.asarray(bootstrap_value, dtype=f32).reshape(
        (-1,) + reward.shape[1:])[-1], reward.shape[1:])
    adv[T - 1] = (reward[T - 1] + (g * bv) * nd[T - 1]) - value[T - 1]
    for t in range(T - 2, -1, -1):
        delta = (reward[t] + (g * value[t + 1]) * nd[t]) - value[t]
        adv[t] = delta + (gl * nd[t]) * adv[t + 1]
    ret = adv + value
    return adv, ret


def valid_from_done(done):
    """rlpyt/algos/utils.py:104-112 -> float32 mask, 0 after the first done."""
    d = np.asarray(done).astype(f32)
    valid = np.ones_like(d)
    valid[1:] = 1 - np.minimum(np.cumsum(d[:-1], axis=0), 1)
    return valid.astype(f32)


def discount_return_n_step(reward, done, n_step, discount, do_truncated=False):
    """rlpyt/algos/utils.py:67-101 -> (return_ f32, done_n bool)."""
    reward = np.asarray(reward, dtype=f32)
    done_b = np.asarray(done).astype(bool)
    T_in = reward.shape[0]
    rlen = T_in if do_truncated else T_in - (n_step - 1)
    ret = reward[:rlen].copy()
    dn = done_b[:rlen].copy()
    for n in range(1, n_step):
        c = f32(float(discount) ** n)
        if do_truncated:
            k = T_in - n  # rows that still have an n-th future reward
            if k <= 0:
                break
            ret[:k] = ret[:k] + (c * reward[n:n + k]) * (1 - dn[:k].astype(f32))
            dn[:k] = np.maximum(dn[:k], done_b[n:n + k])
        else:
            ret = ret + (c * reward[n:n + rlen]) * (1 - dn.astype(f32))
            dn = np.maximum(dn, done_b[n:n + rlen])
    return ret.astype(f32), dn


def normalize_advantage(advantage, valid=None, eps=1e-6):
    """rlpyt/algos/pg/base.py:65-73: masked mean, unbiased std, max(std, eps)."""
    a = np.asarray(advantage, dtype=f32)
    sel = a[np.asarray(valid) > 0] if valid is not None else a.reshape(-1)
    mean = f32(sel.astype(np.float64).mean())
    std = f32(sel.astype(np.float64).std(ddof=1))
    den = f32(eps) if f32(eps) > std else std
    return ((a - mean) / den).astype(f32), mean, std


def iterate_mb_idxs(data_length, minibatch_size, rng=None):
    """rlpyt/utils/misc.py:6-17 with shuffle=True (rng: np.random module or RandomState)."""
    rng = rng if rng is not None else np.random
    idxs = np.arange(data_length)
    rng.shuffle(idxs)
    for s in range(0, data_length - minibatch_size + 1, minibatch_size):
        yield idxs[s:s + minibatch_size]


# --------------------------------------------------------------------------------------
# losses (torch CPU autograd restatements)
# --------------------------------------------------------------------------------------
def _valid_mean(x, valid):
    """rlpyt/utils/tensor.py:39-46."""
    if valid is None:
        return x.mean()
    v = valid.type(x.dtype)
    return (x * v).sum() / v.sum()


def ppo_loss_torch(prob_new, value, prob_old, action, advantage, return_, valid, ratio_clip,
                   value_loss_coeff, entropy_loss_coeff):
    """rlpyt/algos/pg/ppo.py:136-153 + distributions/categorical.py:32-43 (torch CPU).

    prob_new/value may require grad.  Returns (loss, pi_loss, value_loss, entropy,
    perplexity) as 0-d tensors."""
    import torch
    ar = torch.arange(action.numel())
    num = prob_new.reshape(-1, prob_new.shape[-1])[ar, action.reshape(-1)]
    den = prob_old.reshape(-1, prob_old.shape[-1])[ar, action.reshape(-1)]
    ratio = ((num + EPS_CAT) / (den + EPS_CAT)).reshape(action.shape)
    surr_1 = ratio * advantage
    surr_2 = torch.clamp(ratio, 1. - ratio_clip, 1. + ratio_clip) * advantage
    pi_loss = -_valid_mean(torch.min(surr_1, surr_2), valid)
    value_loss = value_loss_coeff * _valid_mean(0.5 * (value - return_) ** 2, valid)
    ent = -torch.sum(prob_new * torch.log(prob_new + EPS_CAT), dim=-1)
    entropy = _valid_mean(ent, valid)
    loss = pi_loss + value_loss - entropy_loss_coeff * entropy
    perplexity = _valid_mean(torch.exp(ent), valid)
    return loss, pi_loss, value_loss, entropy, perplexity


def a2c_loss_torch(prob, value, action, advantage, return_, valid, value_loss_coeff,
                   entropy_loss_coeff):
    """rlpyt/algos/pg/a2c.py:85-101 (torch CPU)."""
    import torch
    ar = torch.arange(action.numel())
    sel = prob.reshape(-1, prob.shape[-1])[ar, action.reshape(-1)].reshape(action.shape)
    logli = torch.log(sel + EPS_CAT)
    pi_loss = -_valid_mean(logli * advantage, valid)
    value_loss = value_loss_coeff * _valid_mean(0.5 * (value - return_) ** 2, valid)
    ent = -torch.sum(prob * torch.log(prob + EPS_CAT), dim=-1)
    entropy = _valid_mean(ent, valid)
    loss = pi_loss + value_loss - entropy_loss_coeff * entropy
    perplexity = _valid_mean(torch.exp(ent), valid)
    return loss, pi_loss, value_loss, entropy, perplexity


def dqn_loss_torch(qs, target_qs, next_qs, action, return_, done_n, is_weights, discount,
                   n_step, delta_clip):
    """rlpyt/algos/dqn/dqn.py:231-263 (torch CPU) -> (loss, td_abs_errors)."""
    import torch
    ar = torch.arange(action.numel())
    q = qs[ar, action]
    with torch.no_grad():
        if next_qs is not None:
            target_q = target_qs[ar, torch.argmax(next_qs, dim=-1)]
        else:
            target_q = torch.max(target_qs, dim=-1).values
    disc_target_q = (discount ** n_step) * target_q
    y = return_ + (1 - done_n.float()) * disc_target_q
    delta = y - q
    losses = 0.5 * delta ** 2
    abs_delta = abs(delta)
    if delta_clip is not None:
        b = delta_clip * (abs_delta - delta_clip / 2)
        losses = torch.where(abs_delta <= delta_clip, losses, b)
    if is_weights is not None:
        losses = losses * is_weights
    td = abs_delta.detach()
    if delta_clip is not None:
        td = torch.clamp(td, 0, delta_clip)
    return torch.mean(losses), td


def cat_dqn_loss_torch(ps, target_ps, next_ps, action, return_, done_n, is_weights, done,
                       V_min, V_max, discount, n_step):
    """rlpyt/algos/dqn/cat_dqn.py:42-93 (torch CPU) -> (loss, KL_div).  ``ps`` / ``target_ps``
    / ``next_ps`` [B,A,P] are the network outputs the method obtains from the agent
    (``next_ps`` given = double DQN); ``done`` given = the ``not mid_batch_reset`` branch
    (valid_from_done over the batch axis, exactly as written there)."""
    import torch
    EPS = 1e-6  # cat_dqn.py:8
    n_atoms = ps.shape[-1]
    ar = torch.arange(action.numel())
    delta_z = (V_max - V_min) / (n_atoms - 1)
    z = torch.linspace(V_min, V_max, n_atoms)
    next_z = z * (discount ** n_step)
    next_z = torch.outer(1 - done_n.float(), next_z)
    next_z = torch.clamp(return_.unsqueeze(1) + next_z, V_min, V_max)
    coeffs = torch.clamp(1 - abs(next_z.unsqueeze(1) - z.view(1, -1, 1)) / delta_z, 0, 1)
    with torch.no_grad():
        sel = next_ps if next_ps is not None else target_ps
        next_a = torch.argmax(torch.tensordot(sel, z, dims=1), dim=-1)
        target_p = (target_ps[ar, next_a].unsqueeze(1) * coeffs).sum(-1)
    p = torch.clamp(ps[ar, action], EPS, 1)
    losses = -torch.sum(target_p * torch.log(p), dim=1)
    if is_weights is not None:
        losses = losses * is_weights
    target_p = torch.clamp(target_p, EPS, 1)
    kl = torch.sum(target_p * (torch.log(target_p) - torch.log(p.detach())), dim=1)
    kl = torch.clamp(kl, EPS, 1 / EPS)
    if done is not None:
        valid = torch.from_numpy(valid_from_done(done.numpy().astype(np.float32)))
        loss = _valid_mean(losses, valid)
        kl = kl * valid
    else:
        loss = torch.mean(losses)
    return loss, kl


# --------------------------------------------------------------------------------------
# rlpyt/replays/sum_tree.py
# --------------------------------------------------------------------------------------
class SumTree:
    """rlpyt/replays/sum_tree.py:8-222 restated (float64 array tree, diffs propagated with
    np.add.at in batch order)."""

    def __init__(self, T, B, off_backward, off_forward, default_value=1,
                 enable_input_priorities=False, input_priority_shift=0):
        self.T, self.B, self.size = T, B, T * B
        self.off_backward, self.off_forward = off_backward, off_forward
        self.default_value = default_value
        self.input_priority_shift = input_priority_shift
        self.tree_levels = int(np.ceil(np.log2(self.size + 1)) + 1)
        self.tree = np.zeros(2 ** self.tree_levels - 1)
        self.low_idx = 2 ** (self.tree_levels - 1) - 1
        self.high_idx = self.size + self.low_idx
        self.priorities = self.tree[self.low_idx:self.high_idx].reshape(T, B)
        self.input_priorities = (default_value * np.ones((T, B))
                                 if enable_input_priorities else None)
        self.reset()

    def reset(self):
        self.tree.fill(0)
        self.t = 0
        self._guard = True
        if self.input_priorities is not None:
            self.input_priorities[:] = self.default_value

    def advance(self, T, priorities=None):
        if T == 0:
            return
        t, b, f, TT = self.t, self.off_backward, self.off_forward, self.T
        lo_on, hi_on = (t - b) % TT, ((t + T - b - 1) % TT) + 1
        lo_off, hi_off = (t + T - b) % TT, ((t + T + f - 1) % TT) + 1
        if self._guard:
            lo_on = max(f, t - b)
            hi_on = lo_off = max(lo_on, t + T - b)
            if t + T - b >= f:
                self._guard = False
        if priorities is not None:
            assert self.input_priorities is not None, "Must enable input priorities."
            it = t - self.input_priority_shift
            rows = np.arange(it, it + T) % TT
            self.input_priorities[rows] = priorities
            if self._guard and it < 0:
                self.input_priorities[it:] = self.default_value
        # ranges in the order the reference concatenates them: ON pieces, then OFF pieces
        on, off = [], []
        if hi_on > lo_on:
            on = [(lo_on, hi_on)]
        elif hi_on < lo_on:
            on = [(lo_on, TT), (0, hi_on)]
        if hi_off > lo_off:
            off = [(lo_off, hi_off)]
        else:
            off = [(lo_off, TT), (0, hi_off)]
        idxs, diffs = [], []
        for lo, hi in on:   # all ON diffs are taken before any write in the reference too
            new = (self.default_value if self.input_priorities is None
                   else self.input_priorities[lo:hi])
            diffs.append((new - self.priorities[lo:hi]).reshape(-1))
            idxs.append(np.arange(lo * self.B, hi * self.B) + self.low_idx)
        for lo, hi in on:
            self.priorities[lo:hi] = (self.default_value if self.input_priorities is None
                                      else self.input_priorities[lo:hi])
        for lo, hi in off:
            diffs.append((-self.priorities[lo:hi]).reshape(-1))
            idxs.append(np.arange(lo * self.B, hi * self.B) + self.low_idx)
        for lo, hi in off:
            self.priorities[lo:hi] = 0
        if diffs:
            self._propagate(np.concatenate(idxs), np.concatenate(diffs))
        self.t = (t + T) % TT

    def _propagate(self, tree_idxs, diffs):
        for _ in range(1, self.tree_levels):
            tree_idxs = (tree_idxs - 1) // 2
            np.add.at(self.tree, tree_idxs, diffs)

    def find(self, uniforms):
        v = self.tree[0] * np.asarray(uniforms, dtype=np.float64)
        idx = np.zeros(len(v), dtype=np.int64)
        for _ in range(self.tree_levels - 1):
            idx = 2 * idx + 1
            left = self.tree[idx]
            right = v > left
            idx[right] += 1
            v[right] -= left[right]
        return idx

    def sample_with(self, uniforms):
        """sample() of the reference with the uniforms injected (non-unique mode)."""
        idx = self.find(uniforms)
        self.prev_tree_idxs = idx
        T_idxs, B_idxs = np.divmod(idx - self.low_idx, self.B)
        return (T_idxs, B_idxs), self.tree[idx]

    def update_batch_priorities(self, priorities):
        uniq, first = np.unique(self.prev_tree_idxs, return_index=True)
        self.prev_tree_idxs = uniq
        p = np.asarray(priorities, dtype=np.float64)[first]
        diffs = p - self.tree[uniq]
        self.tree[uniq] = p
        self._propagate(uniq, diffs)


# --------------------------------------------------------------------------------------
# frame / sequence extraction
# --------------------------------------------------------------------------------------
def frames_gather(frames, done, T_idxs, B_idxs, n_frames):
    """rlpyt/replays/non_sequence/frame.py:14-30.  frames [T+C-1,B,...], done [T,B]."""
    C = n_frames
    T = frames.shape[0] - (C - 1)
    out = np.stack([frames[t:t + C, b] for t, b in zip(T_idxs, B_idxs)], axis=0)
    for f in range(1, C):
        blank = np.where(done[(np.asarray(T_idxs) - f) % T, B_idxs])[0]
        out[blank, :C - f] = 0
    return out


def frames_gather_seq(frames, done, T_idxs, B_idxs, n_frames, seq_T):
    """rlpyt/replays/sequence/frame.py:17-50 -> [seq_T, n, C, ...]."""
    C = n_frames
    T = frames.shape[0] - (C - 1)
    n = len(B_idxs)
    out = np.empty((seq_T, n, C) + frames.shape[2:], dtype=frames.dtype)
    for i, (t, b) in enumerate(zip(T_idxs, B_idxs)):
        for s in range(seq_T):
            tt = (t + s) % T
            out[s, i] = frames[tt:tt + C, b]
        drel = np.where(done[np.arange(t - (C - 1), t + seq_T) % T, b])[0] - (C - 1)
        for f in range(1, C):
            tb = drel + f
            tb = tb[(tb >= 0) & (tb < seq_T)]
            out[tb, i, :C - f] = 0
    return out


def extract_sequences(arr, T_idxs, B_idxs, seq_T):
    """rlpyt/utils/misc.py:38-56, including its literal negative-start behaviour."""
    out = np.empty((seq_T, len(B_idxs)) + arr.shape[2:], dtype=arr.dtype)
    L = len(arr)
    for i, (t, b) in enumerate(zip(T_idxs, B_idxs)):
        if t + seq_T > L:
            m = L - t
            out[:m, i] = arr[t:, b]
            out[m:, i] = arr[:seq_T - m, b]
        elif t < 0:
            out[t:, i] = arr[t:, b]
            out[:t, i] = arr[:t + seq_T, b]
        else:
            out[:, i] = arr[t:t + seq_T, b]
    return out
