"""Device-tensor wrappers over the C ABI (``include/rlpyt_hip.h``).

Every function here takes/returns ``torch`` tensors living on the MI355X and launches on
torch's current HIP stream.  Shapes follow the reference's ``[Time, Batch, ...]`` layout.
No function in this module computes anything itself: arithmetic lives in the HIP kernels.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import check, lib, ptr, stream
from .utils import ktimer

SCAN_EXACT = 0
SCAN_SEGMENTED = 1


def _tn(x):
    """[T, ...] -> (T, N) with N = prod(trailing dims)."""
    T = x.shape[0]
    N = 1
    for s in x.shape[1:]:
        N *= s
    return T, N


def _as_done_u8(done):
    """torch.bool / uint8 / float done mask -> contiguous uint8 0/1 view or copy."""
    if done.dtype == torch.bool:
        return done.contiguous().view(torch.uint8)
    if done.dtype == torch.uint8:
        return done.contiguous()
    return (done != 0).contiguous().view(torch.uint8)


def _f32(x):
    return x.contiguous() if x.dtype == torch.float32 else x.float().contiguous()


_unit_seeds = {}


def unit_seed(device):
    """A standing scalar 1 per device for ``loss.backward(unit_seed(dev))``: ``loss.backward()`` fills a
    fresh one per update, and the loss Functions below recognise THIS tensor in their backward and hand
    out their stored gradients as they are instead of multiplying them by it (one fill + one multiply per
    update; any other incoming gradient takes the general path)."""
    device = torch.device(device)
    one = _unit_seeds.get(device)
    if one is None:
        one = _unit_seeds[device] = torch.ones((), dtype=torch.float32, device=device)
    return one


def _is_unit_seed(g):
    one = _unit_seeds.get(g.device)
    return one is not None and g.data_ptr() == one.data_ptr() and g.dim() == 0


# --------------------------------------------------------------------------------------
# scans
# --------------------------------------------------------------------------------------
def gae(reward, value, done, bootstrap_value, discount, gae_lambda, advantage_dest=None,
        return_dest=None, with_valid=False, variant=SCAN_EXACT):
    """rlpyt/algos/utils.py:24-40 on device.  Returns (advantage, return_[, valid])."""
    _lib.require_gpu()
    reward, value = _f32(reward), _f32(value)
    done8 = _as_done_u8(done)
    T, N = _tn(reward)
    bv = _f32(bootstrap_value).reshape(-1)
    if bv.numel() != N:
        bv = _f32(bootstrap_value.expand(reward.shape[1:])).reshape(-1)
    adv = advantage_dest if advantage_dest is not None else torch.empty_like(reward)
    ret = return_dest if return_dest is not None else torch.empty_like(reward)
    valid = torch.empty_like(reward) if with_valid else None
    with ktimer.region("gae", T * N * (21 if with_valid else 17) + 4 * N):
        check(lib.rlpyt_gae_f32(ptr(reward), ptr(value), ptr(done8), ptr(bv), ptr(adv),
                                ptr(ret), ptr(valid), T, N, float(discount),
                                float(gae_lambda), variant, stream()), "rlpyt_gae_f32")
    return (adv, ret, valid) if with_valid else (adv, ret)


def discount_return(reward, done, bootstrap_value, discount, return_dest=None, value=None,
                    with_valid=False, variant=SCAN_EXACT):
    """rlpyt/algos/utils.py:8-21 on device.  With ``value`` also returns advantage=R-V."""
    _lib.require_gpu()
    reward = _f32(reward)
    done8 = _as_done_u8(done)
    T, N = _tn(reward)
    bv = _f32(bootstrap_value).reshape(-1)
    if bv.numel() != N:
        bv = _f32(bootstrap_value.expand(reward.shape[1:])).reshape(-1)
    ret = return_dest if return_dest is not None else torch.empty_like(reward)
    adv = None
    if value is not None:
        value = _f32(value)
        adv = torch.empty_like(reward)
    valid = torch.empty_like(reward) if with_valid else None
    check(lib.rlpyt_discount_return_f32(ptr(reward), ptr(done8), ptr(bv), ptr(ret), ptr(value),
                                        ptr(adv), ptr(valid), T, N, float(discount), variant,
                                        stream()), "rlpyt_discount_return_f32")
    out = (ret,)
    if value is not None:
        out += (adv,)
    if with_valid:
        out += (valid,)
    return out[0] if len(out) == 1 else out


def valid_from_done(done):
    """rlpyt/algos/utils.py:104-112 on device -> float32 mask."""
    _lib.require_gpu()
    done8 = _as_done_u8(done)
    T, N = _tn(done8)
    valid = torch.empty(done8.shape, dtype=torch.float32, device=done8.device)
    check(lib.rlpyt_valid_from_done(ptr(done8), ptr(valid), T, N, stream()),
          "rlpyt_valid_from_done")
    return valid


def discount_return_n_step(reward, done, n_step, discount, return_dest=None, done_n_dest=None,
                           do_truncated=False):
    """rlpyt/algos/utils.py:67-101 on device -> (return_, done_n[bool])."""
    _lib.require_gpu()
    reward = _f32(reward)
    done8 = _as_done_u8(done)
    T_in, N = _tn(reward)
    rlen = T_in if do_truncated else T_in - (n_step - 1)
    shape = (max(rlen, 0),) + tuple(reward.shape[1:])
    ret = return_dest if return_dest is not None else torch.empty(
        shape, dtype=torch.float32, device=reward.device)
    dn = done_n_dest if done_n_dest is not None else torch.empty(
        shape, dtype=torch.bool, device=reward.device)
    assert ret.is_contiguous() and dn.is_contiguous()
    dn8 = dn.view(torch.uint8) if dn.dtype == torch.bool else dn
    check(lib.rlpyt_nstep_return_f32(ptr(reward), ptr(done8), ptr(ret), ptr(dn8), T_in, N,
                                     int(n_step), float(discount), int(bool(do_truncated)),
                                     stream()), "rlpyt_nstep_return_f32")
    return ret, dn


_ws_cache = {}


_ws_retired = []


def _workspace(kind, nbytes, device):
    key = (kind, device, torch.cuda.current_stream().cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        if ws is not None:
            # a captured hipGraph may hold the old buffer's address: outgrown workspaces are kept
            # alive instead of going back to the allocator (they are few and small)
            _ws_retired.append(ws)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


def normalize_advantage_(advantage, valid=None, eps=1e-6, return_stats=False):
    """In-place ``(A - mean) / max(std, eps)`` over valid entries (pg/base.py:65-73)."""
    _lib.require_gpu()
    assert advantage.dtype == torch.float32 and advantage.is_contiguous()
    n = advantage.numel()
    if valid is not None:
        valid = _f32(valid)
        assert valid.numel() == n
    ws = _workspace("norm", lib.rlpyt_adv_normalize_workspace_bytes(n), advantage.device)
    stats = torch.empty(3, dtype=torch.float32, device=advantage.device) if return_stats else None
    check(lib.rlpyt_adv_normalize_f32(ptr(advantage), ptr(valid), n, float(eps), ptr(ws),
                                      ptr(stats), stream()), "rlpyt_adv_normalize_f32")
    return (advantage, stats) if return_stats else advantage


# --------------------------------------------------------------------------------------
# losses (autograd Functions: forward runs the fused fwd+bwd kernel, backward hands the
# saved gradients to autograd scaled by the incoming grad)
# --------------------------------------------------------------------------------------
class _PpoLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prob_new, value, prob_old, action, advantage, return_, valid, ratio_clip,
                value_loss_coeff, entropy_loss_coeff):
        _lib.require_gpu()
        A = prob_new.shape[-1]
        pn = _f32(prob_new).reshape(-1, A)
        M = pn.shape[0]
        v = _f32(value).reshape(-1)
        po = _f32(prob_old).reshape(-1, A)
        act = action.reshape(-1).long().contiguous()
        adv = _f32(advantage).reshape(-1)
        ret = _f32(return_).reshape(-1)
        val = None if valid is None else _f32(valid).reshape(-1)
        out = torch.empty(5, dtype=torch.float32, device=pn.device)
        gp = torch.empty_like(pn)
        gv = torch.empty_like(v)
        ws = _workspace("loss", lib.rlpyt_pg_loss_workspace_bytes(M), pn.device)
        with ktimer.region("ppo_loss", M * (12 * A + 28 + (4 if val is not None else 0))):
            check(lib.rlpyt_ppo_loss_fwd_bwd_f32(
                ptr(pn), ptr(v), ptr(po), ptr(act), ptr(adv), ptr(ret), ptr(val), M, A,
                float(ratio_clip), float(value_loss_coeff), float(entropy_loss_coeff), ptr(out),
                ptr(gp), ptr(gv), ptr(ws), stream()), "rlpyt_ppo_loss_fwd_bwd_f32")
        ctx.save_for_backward(gp, gv)
        ctx.shapes = (prob_new.shape, value.shape)
        ctx.mark_non_differentiable(out)
        return out[0], out

    @staticmethod
    def backward(ctx, g_loss, _g_out):
        gp, gv = ctx.saved_tensors
        ps, vs = ctx.shapes
        return ((gp * g_loss).reshape(ps), (gv * g_loss).reshape(vs), None, None, None, None,
                None, None, None, None)


def ppo_loss(prob_new, value, prob_old, action, advantage, return_, valid, ratio_clip,
             value_loss_coeff, entropy_loss_coeff):
    """PPO.loss (rlpyt/algos/pg/ppo.py:117-154) as one fused kernel.

    Returns ``(loss, scalars)`` where ``loss`` is differentiable w.r.t. ``prob_new`` and
    ``value`` and ``scalars = [loss, pi_loss, value_loss, entropy, perplexity]`` (detached).
    """
    return _PpoLoss.apply(prob_new, value, prob_old, action, advantage, return_, valid,
                          ratio_clip, value_loss_coeff, entropy_loss_coeff)


class _PpoHeadLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, w_pi, b_pi, w_v, b_v, prob_old, action, advantage, return_, valid,
                ratio_clip, value_loss_coeff, entropy_loss_coeff, flat_idx=None, trunk_bias=None,
                unit_grad=False, scalars_out=None):
        _lib.require_gpu()
        ctx.unit_grad = bool(unit_grad)
        rc_dev = None
        if isinstance(ratio_clip, torch.Tensor):     # device scalar (captured update graphs)
            rc_dev, ratio_clip = ratio_clip, 0.
            assert rc_dev.dtype == torch.float32 and rc_dev.is_cuda and rc_dev.numel() == 1
        K = h.shape[-1]
        A = w_pi.shape[0]
        hc = _f32(h).reshape(-1, K)
        M = hc.shape[0]
        wp, bp = _f32(w_pi.detach()), _f32(b_pi.detach())
        wv, bv = _f32(w_v.detach()).reshape(-1), _f32(b_v.detach()).reshape(-1)
        tb = None if trunk_bias is None else _f32(trunk_bias.detach()).reshape(-1)
        T = B = 0
        if flat_idx is not None:     # loss inputs are [T,B,...] batch arrays, indexed in-kernel
            assert valid is None and prob_old.dim() == 3
            T, B = prob_old.shape[:2]
            flat_idx = flat_idx.long().contiguous()
            assert flat_idx.numel() == M
        po = _f32(prob_old).reshape(-1, A)
        act = action.reshape(-1).long().contiguous()
        adv = _f32(advantage).reshape(-1)
        ret = _f32(return_).reshape(-1)
        val = None if valid is None else _f32(valid).reshape(-1)
        if scalars_out is None:
            out = torch.empty(5, dtype=torch.float32, device=hc.device)
        else:      # the caller's diagnostics row (5 consecutive floats)
            out = scalars_out
            assert (out.dtype == torch.float32 and out.numel() == 5 and out.is_contiguous()
                    and out.device == hc.device)
        gh = torch.empty_like(hc)
        gparams = torch.empty(A * K + K + A + 1 + (K if tb is not None else 0),
                              dtype=torch.float32, device=hc.device)
        ws = _workspace("head_loss", lib.rlpyt_ppo_head_loss_workspace_bytes(K, A), hc.device)
        with ktimer.region("ppo_head_loss", M * (8 * K + 8 * A + 28)):
            check(lib.rlpyt_ppo_trunk_head_loss_fwd_bwd_dev_f32(
                ptr(hc), ptr(tb), ptr(wp), ptr(bp), ptr(wv), ptr(bv), ptr(po), ptr(act), ptr(adv),
                ptr(ret), ptr(val), ptr(flat_idx), int(T), int(B), M, K, A, float(ratio_clip),
                ptr(rc_dev), float(value_loss_coeff),
                float(entropy_loss_coeff), ptr(out), ptr(gh), ptr(gparams), ptr(ws), stream()),
                "rlpyt_ppo_trunk_head_loss_fwd_bwd_dev_f32")
        ctx.save_for_backward(gh, gparams)
        ctx.meta = (h.shape, w_pi.shape, b_pi.shape, w_v.shape, b_v.shape, A, K,
                    None if trunk_bias is None else trunk_bias.shape)
        ctx.mark_non_differentiable(out)
        return out[0], out

    @staticmethod
    def backward(ctx, g_loss, _g_out):
        gh, gp = ctx.saved_tensors
        hs, wps, bps, wvs, bvs, A, K, tbs = ctx.meta
        if not ctx.unit_grad:
            # (unit_grad: the caller runs ``loss.backward()`` on the returned loss itself, the
            # incoming gradient is exactly 1 -- two elementwise launches over 16 MB less per update)
            gp, gh = gp * g_loss, gh * g_loss
        o = A * K
        g_tb = None if tbs is None else gp[o + K + A + 1:].reshape(tbs)
        return (gh.reshape(hs), gp[:o].reshape(wps), gp[o + K:o + K + A].reshape(bps),
                gp[o:o + K].reshape(wvs), gp[o + K + A:o + K + A + 1].reshape(bvs)) + \
            (None,) * 9 + (g_tb, None, None)


def ppo_head_loss(h, w_pi, b_pi, w_v, b_v, prob_old, action, advantage, return_, valid,
                  ratio_clip, value_loss_coeff, entropy_loss_coeff, flat_idx=None,
                  trunk_bias=None, unit_grad=False, scalars_out=None):
    """PPO.loss (rlpyt/algos/pg/ppo.py:117-154) with the policy / value heads of
    rlpyt/models/pg/atari_ff_model.py:56-58 fused in: takes the trunk output ``h [M, K]`` and the
    head parameters, returns ``(loss, scalars)`` like ``ppo_loss``; differentiable w.r.t. ``h``
    and the four head parameters (all gradients come out of the same kernel pass).  With
    ``flat_idx`` (int64 ``[M]``) the loss inputs are the whole ``[T,B,...]`` batch arrays and the
    kernel reads sample m at ``(idx % T, idx // T)`` -- no minibatch gather launches.
    With ``trunk_bias [K]``, ``h`` is the trunk's pre-activation WITHOUT its bias (``x W^T``) and
    the kernel applies ``relu(h + trunk_bias)`` itself (the op is then differentiable w.r.t. the
    pre-activation and the bias: the Linear's bias add, the ReLU, its backward and the bias
    gradient reduction never launch).
    ``unit_grad=True``: a promise that backward is seeded with exactly 1 (``loss.backward()`` on the
    returned loss): the gradients saved by the forward kernel are handed on as they are.
    ``scalars_out``: 5 consecutive f32 of the caller (a row of its diagnostics table) that receive the
    scalars instead of a fresh tensor."""
    return _PpoHeadLoss.apply(h, w_pi, b_pi, w_v, b_v, prob_old, action, advantage, return_,
                              valid, ratio_clip, value_loss_coeff, entropy_loss_coeff, flat_idx,
                              trunk_bias, unit_grad, scalars_out)


class _A2cLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prob, value, action, advantage, return_, valid, value_loss_coeff,
                entropy_loss_coeff):
        _lib.require_gpu()
        A = prob.shape[-1]
        pn = _f32(prob).reshape(-1, A)
        M = pn.shape[0]
        v = _f32(value).reshape(-1)
        act = action.reshape(-1).long().contiguous()
        adv = _f32(advantage).reshape(-1)
        ret = _f32(return_).reshape(-1)
        val = None if valid is None else _f32(valid).reshape(-1)
        out = torch.empty(5, dtype=torch.float32, device=pn.device)
        gp = torch.empty_like(pn)
        gv = torch.empty_like(v)
        ws = _workspace("loss", lib.rlpyt_pg_loss_workspace_bytes(M), pn.device)
        check(lib.rlpyt_a2c_loss_fwd_bwd_f32(ptr(pn), ptr(v), ptr(act), ptr(adv), ptr(ret),
                                             ptr(val), M, A, float(value_loss_coeff),
                                             float(entropy_loss_coeff), ptr(out), ptr(gp),
                                             ptr(gv), ptr(ws), stream()),
              "rlpyt_a2c_loss_fwd_bwd_f32")
        ctx.save_for_backward(gp, gv)
        ctx.shapes = (prob.shape, value.shape)
        ctx.mark_non_differentiable(out)
        return out[0], out

    @staticmethod
    def backward(ctx, g_loss, _g_out):
        gp, gv = ctx.saved_tensors
        ps, vs = ctx.shapes
        return ((gp * g_loss).reshape(ps), (gv * g_loss).reshape(vs), None, None, None, None,
                None, None)


def a2c_loss(prob, value, action, advantage, return_, valid, value_loss_coeff,
             entropy_loss_coeff):
    """A2C.loss (rlpyt/algos/pg/a2c.py:63-103) as one fused kernel; see ``ppo_loss``."""
    return _A2cLoss.apply(prob, value, action, advantage, return_, valid, value_loss_coeff,
                          entropy_loss_coeff)


class _DqnLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qs, target_qs, next_qs, action, return_, done_n, is_weights, disc_n,
                delta_clip):
        _lib.require_gpu()
        A = qs.shape[-1]
        q = _f32(qs).reshape(-1, A)
        M = q.shape[0]
        tq = _f32(target_qs).reshape(-1, A)
        nq = None if next_qs is None else _f32(next_qs).reshape(-1, A)
        act = action.reshape(-1).long().contiguous()
        ret = _f32(return_).reshape(-1)
        dn = _as_done_u8(done_n).reshape(-1)
        isw = None if is_weights is None else _f32(is_weights).reshape(-1)
        out = torch.empty(2, dtype=torch.float32, device=q.device)
        td = torch.empty(M, dtype=torch.float32, device=q.device)
        gq = torch.empty_like(q)
        ws = _workspace("loss", lib.rlpyt_pg_loss_workspace_bytes(M), q.device)
        check(lib.rlpyt_dqn_loss_fwd_bwd_f32(ptr(q), ptr(tq), ptr(nq), ptr(act), ptr(ret),
                                             ptr(dn), ptr(isw), M, A, float(disc_n),
                                             float(delta_clip if delta_clip is not None else 0.),
                                             ptr(out), ptr(td), ptr(gq), ptr(ws), stream()),
              "rlpyt_dqn_loss_fwd_bwd_f32")
        ctx.save_for_backward(gq)
        ctx.shape = qs.shape
        ctx.mark_non_differentiable(td)
        return out[0], td

    @staticmethod
    def backward(ctx, g_loss, _g_td):
        (gq,) = ctx.saved_tensors
        if not _is_unit_seed(g_loss):
            gq = gq * g_loss
        return (gq.reshape(ctx.shape), None, None, None, None, None, None, None, None)


def dqn_loss(qs, target_qs, next_qs, action, return_, done_n, is_weights, disc_n, delta_clip):
    """DQN.loss (rlpyt/algos/dqn/dqn.py:231-263): returns (loss, td_abs_errors)."""
    return _DqnLoss.apply(qs, target_qs, next_qs, action, return_, done_n, is_weights, disc_n,
                          delta_clip)


class _R2d1Loss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qs, target_qs, next_qs, action, return_, done_n, valid, is_weights,
                disc_n, delta_clip, value_scale_eps, pri_eta):
        _lib.require_gpu()
        T, B, A = qs.shape
        q = _f32(qs)
        tq = _f32(target_qs)
        nq = None if next_qs is None else _f32(next_qs)
        act = action.long().contiguous()
        ret = _f32(return_)
        dn = _as_done_u8(done_n)
        val = _f32(valid)
        isw = None if is_weights is None else _f32(is_weights)
        out = torch.empty(2, dtype=torch.float32, device=q.device)
        td = torch.empty((T, B), dtype=torch.float32, device=q.device)
        pri = torch.empty(B, dtype=torch.float32, device=q.device)
        gq = torch.empty_like(q)
        ws = _workspace("r2d1", lib.rlpyt_r2d1_loss_workspace_bytes(), q.device)
        check(lib.rlpyt_r2d1_loss_fwd_bwd_f32(
            ptr(q), ptr(tq), ptr(nq), ptr(act), ptr(ret), ptr(dn), ptr(val), ptr(isw), T, B, A,
            float(disc_n), float(delta_clip if delta_clip is not None else 0.),
            float(value_scale_eps), float(pri_eta), ptr(out), ptr(td), ptr(pri), ptr(gq),
            ptr(ws), stream()), "rlpyt_r2d1_loss_fwd_bwd_f32")
        ctx.save_for_backward(gq)
        ctx.mark_non_differentiable(td, pri)
        return out[0], td, pri

    @staticmethod
    def backward(ctx, g_loss, _g_td, _g_pri):
        (gq,) = ctx.saved_tensors
        return (gq if _is_unit_seed(g_loss) else gq * g_loss,) + (None,) * 11


def r2d1_loss(qs, target_qs, next_qs, action, return_, done_n, valid, is_weights, disc_n,
              delta_clip, value_scale_eps, pri_eta):
    """R2D1.loss after the network passes (rlpyt/algos/dqn/r2d1.py:298-345):
    returns (loss, valid_td_abs_errors [T,B], priorities [B])."""
    return _R2d1Loss.apply(qs, target_qs, next_qs, action, return_, done_n, valid, is_weights,
                           disc_n, delta_clip, value_scale_eps, pri_eta)


class _CatDqnLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ps, target_ps, next_ps, action, return_, done_n, is_weights, valid, z,
                v_min, v_max, disc_n):
        _lib.require_gpu()
        A, P = ps.shape[-2], ps.shape[-1]
        p = _f32(ps).reshape(-1, A, P)
        M = p.shape[0]
        tp = _f32(target_ps).reshape(-1, A, P)
        nps = None if next_ps is None else _f32(next_ps).reshape(-1, A, P)
        act = action.reshape(-1).long().contiguous()
        ret = _f32(return_).reshape(-1)
        dn = _as_done_u8(done_n).reshape(-1)
        isw = None if is_weights is None else _f32(is_weights).reshape(-1)
        val = None if valid is None else _f32(valid).reshape(-1)
        zz = _f32(z).to(p.device)
        out = torch.empty(2, dtype=torch.float32, device=p.device)
        kl = torch.empty(M, dtype=torch.float32, device=p.device)
        gp = torch.empty_like(p)
        ws = _workspace("catdqn", lib.rlpyt_cat_dqn_loss_workspace_bytes(), p.device)
        check(lib.rlpyt_cat_dqn_loss_fwd_bwd_f32(
            ptr(p), ptr(tp), ptr(nps), ptr(act), ptr(ret), ptr(dn), ptr(isw), ptr(val), ptr(zz),
            M, A, P, float(v_min), float(v_max), float(disc_n), ptr(out), ptr(kl), ptr(gp),
            ptr(ws), stream()), "rlpyt_cat_dqn_loss_fwd_bwd_f32")
        ctx.save_for_backward(gp)
        ctx.shape = ps.shape
        ctx.mark_non_differentiable(kl)
        return out[0], kl

    @staticmethod
    def backward(ctx, g_loss, _g_kl):
        (gp,) = ctx.saved_tensors
        return ((gp if _is_unit_seed(g_loss) else gp * g_loss).reshape(ctx.shape),) + (None,) * 11


def cat_dqn_loss(ps, target_ps, next_ps, action, return_, done_n, is_weights, valid, z, v_min,
                 v_max, disc_n):
    """CategoricalDQN.loss after the network passes (rlpyt/algos/dqn/cat_dqn.py:42-93):
    returns (loss, KL_div [M])."""
    return _CatDqnLoss.apply(ps, target_ps, next_ps, action, return_, done_n, is_weights, valid,
                             z, v_min, v_max, disc_n)


# --------------------------------------------------------------------------------------
# observation running mean / std (rlpyt/models/running_mean_std.py)
# --------------------------------------------------------------------------------------
def obs_batch_stats(x, n_feature_dims):
    """Per-dimension mean and biased variance over all leading dims of ``x``."""
    _lib.require_gpu()
    x = _f32(x)
    shape = tuple(x.shape[x.dim() - n_feature_dims:]) if n_feature_dims else ()
    D = 1
    for s in shape:
        D *= s
    n = x.numel() // D
    mean = torch.empty(shape, dtype=torch.float32, device=x.device)
    var = torch.empty(shape, dtype=torch.float32, device=x.device)
    ws = _workspace("rms", lib.rlpyt_obs_rms_workspace_bytes(n, D), x.device)
    check(lib.rlpyt_obs_batch_stats_f32(ptr(x), n, D, ptr(mean), ptr(var), ptr(ws), stream()),
          "rlpyt_obs_batch_stats_f32")
    return mean, var, n


def obs_rms_merge_(mean, var, count, batch_mean, batch_var, batch_count):
    """In-place Chan merge into the running (mean, var, count) buffers."""
    _lib.require_gpu()
    check(lib.rlpyt_obs_rms_merge_f32(ptr(mean), ptr(var), ptr(count), ptr(_f32(batch_mean)),
                                      ptr(_f32(batch_var)), float(batch_count), mean.numel(),
                                      stream()), "rlpyt_obs_rms_merge_f32")


def obs_normalize(x, mean, var, var_clip=1e-6, obs_clip=10.):
    """clamp((x - mean) / sqrt(max(var, var_clip)), +-obs_clip) (mujoco_ff_model.py:68-73)."""
    _lib.require_gpu()
    x = _f32(x)
    D = mean.numel()
    out = torch.empty_like(x)
    check(lib.rlpyt_obs_normalize_f32(ptr(x), ptr(mean), ptr(var), ptr(out), x.numel() // D, D,
                                      float(var_clip if var_clip is not None else 0.),
                                      float(obs_clip), stream()), "rlpyt_obs_normalize_f32")
    return out


# --------------------------------------------------------------------------------------
# AtariFfModel convolution stack (fp32 MFMA implicit GEMMs, csrc/conv.hip)
# --------------------------------------------------------------------------------------
ATARI_IMG = (4, 104, 80)
ATARI_P1, ATARI_C1, ATARI_F2 = 25 * 19, 16, 32 * 12 * 9
# algorithmic flops per image (2 * MACs): conv1 475x256x16, conv2 108x256x32
ATARI_MASK2_WORDS = 32 * 4       # sign mask of y2: uint32 [M, 32 co, 4 words of 32 positions]
_FL_C1, _FL_C2 = 2 * 475 * 256 * 16, 2 * 108 * 256 * 32
_FL_C2D = 2 * 475 * 128 * 16      # transposed conv: 2x2 taps x 32 channels per input pixel


def _conv_rows(obs, flat_idx):
    assert obs.dtype == torch.uint8 and obs.is_contiguous()
    assert tuple(obs.shape[-3:]) == ATARI_IMG, "fused conv stack is built for uint8 [4,104,80]"
    if flat_idx is not None:
        assert obs.dim() == 5
        T, B = obs.shape[:2]
        flat_idx = flat_idx.long().contiguous()
        return T, B, flat_idx.numel(), flat_idx
    M = obs.numel() // (4 * 104 * 80)
    return 1, max(M, 1), M, None


class _AtariConvStack(torch.autograd.Function):
    @staticmethod
    def forward(ctx, obs, flat_idx, w1, b1, w2, b2, scale):
        _lib.require_gpu()
        T, B, M, idx = _conv_rows(obs, flat_idx)
        w1c, b1c, w2c, b2c = (x.detach().contiguous() for x in (w1, b1, w2, b2))
        assert w1c.shape == (16, 4, 8, 8) and w2c.shape == (32, 16, 4, 4)
        y1 = torch.empty((M, ATARI_P1, ATARI_C1), dtype=torch.float32, device=obs.device)
        y2 = torch.empty((M, ATARI_F2), dtype=torch.float32, device=obs.device)
        # sign bits of y2, written by the conv2 forward kernel beside it: all that conv2's backward
        # pass needs of y2 (512 B instead of 13.8 KB per image on its load path)
        mask2 = torch.empty((M, ATARI_MASK2_WORDS), dtype=torch.int32, device=obs.device)
        # conv1 -> conv2 in ONE pass over the images at update sizes (y1 goes to conv2 through LDS and to
        # HBM once, for the backward pass); at most one image per CU: the two latency-tuned launches
        with ktimer.region("convs_fwd", M * (33280 + 4 * (7600 + 3456) + 4 * ATARI_MASK2_WORDS),
                           M * (_FL_C1 + _FL_C2)):
            check(lib.rlpyt_atari_convs_fwd_f32(ptr(obs), ptr(idx), T, B, M, ptr(w1c), ptr(b1c), ptr(w2c),
                                                ptr(b2c), float(scale), ptr(y1), ptr(y2), ptr(mask2),
                                                stream()), "rlpyt_atari_convs_fwd_f32")
        if any(ctx.needs_input_grad):
            ctx.save_for_backward(obs, idx, w2c, y1, mask2)
            ctx.dims = (T, B, M, float(scale))
        return y2

    @staticmethod
    def backward(ctx, g2):
        obs, idx, w2c, y1, mask2 = ctx.saved_tensors
        T, B, M, scale = ctx.dims
        g2 = _f32(g2)
        dev = obs.device
        dy1 = torch.empty_like(y1)
        ws = _workspace("conv_wgrad", lib.rlpyt_atari_conv_wgrad_workspace_bytes(), dev)
        dw1 = torch.empty((16, 4, 8, 8), dtype=torch.float32, device=dev)
        db1 = torch.empty(16, dtype=torch.float32, device=dev)
        dw2 = torch.empty((32, 16, 4, 4), dtype=torch.float32, device=dev)
        db2 = torch.empty(32, dtype=torch.float32, device=dev)
        # g2 + mask + y1 in, dy1 out (the kernel: bf16x6, dgrad + both ReLU masks + wgrad fused)
        with ktimer.region("conv2_bwd", M * (4 * (3456 + 2 * 7600) + 4 * ATARI_MASK2_WORDS),
                           M * (_FL_C2D + _FL_C2)):
            check(lib.rlpyt_atari_conv2_bwd_x6_f32(ptr(g2), ptr(mask2), ptr(y1), M, ptr(w2c), ptr(dy1),
                                                   ptr(ws), ptr(dw2), ptr(db2), stream()),
                  "rlpyt_atari_conv2_bwd_x6_f32")
        with ktimer.region("conv1_wgrad", M * (33280 + 4 * 7600), M * _FL_C1):
            check(lib.rlpyt_atari_conv1_wgrad_f32(ptr(obs), ptr(idx), T, B, M, ptr(dy1), scale,
                                                  ptr(ws), ptr(dw1), ptr(db1), stream()),
                  "rlpyt_atari_conv1_wgrad_f32")
        return None, None, dw1, db1, dw2, db2, None


def atari_conv_stack(obs, flat_idx, w1, b1, w2, b2, scale=1. / 255):
    """uint8 observations -> the 3456 conv features of AtariFfModel (differentiable w.r.t.
    the four conv parameters): ``relu(conv2(relu(conv1(obs * scale))))`` flattened in NCHW
    order (rlpyt/models/pg/atari_ff_model.py:50-55, rlpyt/models/conv2d.py).

    ``obs``: uint8 ``[M,4,104,80]``, or ``[T,B,4,104,80]`` together with ``flat_idx`` (int64
    ``[M]``) selecting the minibatch rows ``idx -> (idx % T, idx // T)`` inside the kernel."""
    return _AtariConvStack.apply(obs, flat_idx, w1, b1, w2, b2, scale)


# --------------------------------------------------------------------------------------
# per-time-step sampler kernels (csrc/step.hip)
# --------------------------------------------------------------------------------------
_ROW_COPY_DTYPE = [("dst", "<u8"), ("src", "<u8"), ("row_stride", "<i8"), ("col_off", "<i8"),
                   ("nbytes", "<i8"), ("dt", "<i4"), ("reserved", "<i4"), ("zero_where", "<u8"),
                   ("unit_bytes", "<i8")]


class RowCommit:
    """A fixed list of row writes ``dst[t + dt, lo:hi] = src`` executed by ONE kernel launch
    with ``t`` read from a device counter (``rlpyt_commit_rows``): the sampler's per-step
    writes into the ``[T, B]`` batch (rlpyt/samplers/parallel/gpu/collectors.py:30-47).

    ``entries``: tuples ``(dst, src, lo, dt)`` with ``dst`` a contiguous ``[T', B, ...]``
    device tensor and ``src`` a contiguous ``[hi - lo, ...]`` device tensor, or
    ``(dst, src, None, 0)`` for a plain copy ``dst[...] = src``; a fifth element ``zero_where``
    (bool / uint8 device tensor ``[hi - lo]``) makes the launch write zeros for the rows ``i`` with
    ``zero_where[i]`` (the wait-reset collector's blank rows).  The table lives in device
    memory at a fixed address; ``set_entries`` may be called again (e.g. after hipGraph
    capture, once the sources' addresses are known)."""

    def __init__(self, n_entries, device):
        import numpy as np
        self._np = np
        self.n = int(n_entries)
        self.device = device
        self.table = torch.zeros(self.n * 64, dtype=torch.uint8, device=device)
        self.max_bytes = 0
        self._keep = None

    def set_entries(self, entries):
        np = self._np
        assert len(entries) == self.n
        host = np.zeros(self.n, dtype=_ROW_COPY_DTYPE)
        assert host.itemsize == 64
        keep = []
        for i, entry in enumerate(entries):
            dst, src, lo, dt = entry[:4]
            zw = entry[4] if len(entry) > 4 else None
            assert dst.is_cuda and src.is_cuda and dst.is_contiguous() and src.is_contiguous()
            assert dst.dtype == src.dtype
            nbytes = src.numel() * src.element_size()
            if lo is None:
                assert dst.numel() == src.numel()
                row_stride = col_off = 0
            else:
                assert dst.dim() >= 2 and tuple(dst.shape[2:]) == tuple(src.shape[1:])
                per_col = src.element_size()
                for d in dst.shape[2:]:
                    per_col *= d
                row_stride, col_off = per_col * dst.shape[1], per_col * lo
                assert lo + src.shape[0] <= dst.shape[1]
            zw_ptr = unit = 0
            if zw is not None:
                assert zw.is_cuda and zw.is_contiguous() and zw.element_size() == 1
                assert zw.numel() == src.shape[0] and nbytes % zw.numel() == 0
                zw_ptr, unit = zw.data_ptr(), nbytes // zw.numel()
            host[i] = (dst.data_ptr(), src.data_ptr(), row_stride, col_off, nbytes, dt, 0, zw_ptr, unit)
            keep.append((dst, src, zw))
            self.max_bytes = max(self.max_bytes, nbytes)
        self._keep = keep
        self.table.copy_(torch.from_numpy(host.view(np.uint8).copy()))

    def launch(self, t_dev):
        _lib.require_gpu()
        check(lib.rlpyt_commit_rows(ptr(self.table), self.n, self.max_bytes, ptr(t_dev), stream()),
              "rlpyt_commit_rows")


def fc_small(x, weight, bias=None, relu=True):
    """``act(x @ weight.T + bias)`` for a small batch (M <= 256 rows) on fp32 MFMA with a
    split-K over the whole chip: the no-grad trunk layer of the sampling forward, where a
    library GEMM is latency-bound (rlpyt/models/mlp.py through atari_ff_model.py:52-55)."""
    _lib.require_gpu()
    x = _f32(x)
    M, K = x.shape
    N = weight.shape[0]
    w = _f32(weight.detach())
    b = None if bias is None else _f32(bias.detach())
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    ws = _workspace("fc_small", lib.rlpyt_fc_small_workspace_bytes(M, N), x.device)
    check(lib.rlpyt_fc_small_f32(ptr(x), ptr(w), ptr(b), ptr(y), M, N, K, int(bool(relu)), ptr(ws),
                                 stream()), "rlpyt_fc_small_f32")
    return y


# Which kernel serves which trunk GEMM (M = 8192, gemm_bench on the MI355X, profiles/r3_gemm_*):
#   forward   x W^T : lock-step NT kernel (csrc/gemm.hip) 156-163 us (a producer / consumer-wave
#                     body measured 196-233 us and was removed in round 5);
#   dgrad     g W   : the same kernel on a transposed copy of W, 195-200 (+12 for the copy; a kernel
#                     reading W as stored measured 228-240 and was removed);
#   wgrad     g^T x : gemm_tn (csrc/gemm_tn.hip) 223-240 against 265-274 for the library GEMM.
# (On gfx950 VALU and MFMA instructions of one SIMD do not overlap, whichever wave issues them, so a
# bf16x6 GEMM that splits its operands in-kernel is bounded by matrix-pipe + split-VALU time, ~2200
# cycles per 128 x 128 x 32 step; the lock-step kernel sits within 10 % of that.)
def gemm_nt(a, b, region="gemm_nt"):
    """``a @ b.T`` for f32 ``a [M, K]``, ``b [N, K]`` (K a multiple of 32) on the bf16 matrix pipe
    from exact three-piece bf16 splits of both operands (six products, f32 accumulation, dropped
    terms <= 2^-24 |ab|, 2^-27 rms -- f32-level error, 2.7x less matrix-pipe time than an f32-MFMA
    GEMM): ``rlpyt_gemm_nt_f32`` (lock-step kernel, csrc/gemm.hip)."""
    _lib.require_gpu()
    a, b = _f32(a), _f32(b)
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    with ktimer.region(region, 4 * (M * K + N * K + M * N), 2 * M * N * K):
        check(lib.rlpyt_gemm_nt_f32(ptr(a), ptr(b), ptr(c), M, N, K, stream()), "rlpyt_gemm_nt_f32")
    return c


def gemm_tn(a, b):
    """``a.T @ b`` for f32 ``a [K, M]``, ``b [K, N]`` (K a multiple of 32; M, N of 4), same
    arithmetic as ``gemm_nt``: the weight gradient ``g^T x`` of a Linear -- a contraction over the
    batch axis.  Long contractions are cut into 8 K chunks whose partial tiles are summed in a
    fixed order (``rlpyt_gemm_tn_f32``; run-to-run identical results)."""
    _lib.require_gpu()
    a, b = _f32(a), _f32(b)
    K, M = a.shape
    N = b.shape[1]
    assert b.shape[0] == K
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    nbytes = int(lib.rlpyt_gemm_tn_workspace_bytes(M, N, K))
    ws = _workspace("gemm_tn", nbytes, a.device) if nbytes else None
    with ktimer.region("gemm_tn", 4 * (M * K + N * K + M * N), 2 * M * N * K):
        check(lib.rlpyt_gemm_tn_f32(ptr(a), ptr(b), ptr(c), M, N, K, ptr(ws), stream()),
              "rlpyt_gemm_tn_f32")
    return c


class TransposedMirror:
    """``W^T`` copies of weights whose input-gradient GEMM wants them transposed (``_LinearNoBias``),
    kept current WITHOUT a per-minibatch transposing copy: ``get(W)`` returns the cached ``W^T`` while
    ``W`` is unchanged since it was made (storage address + in-place version counter) and re-makes it
    otherwise; an optimizer that writes the new ``W^T`` beside the new ``W`` in its own launch
    (``ClipAdam``, ``rlpyt_clip_adam_step_mirror_f32``) asks ``buffer_for(W)`` for the destination and
    calls ``written(W)`` once the parameter's version has been bumped.  Any other writer of ``W``
    (``load_state_dict``, another optimizer, DDP's initial broadcast) changes the version: the next
    ``get`` then falls back to the copy."""
    _entries = {}          # id(W) -> [weakref(W), W^T, (data_ptr, version) the copy describes]

    @classmethod
    def _entry(cls, w):
        import weakref
        e = cls._entries.get(id(w))
        if e is not None and e[0]() is w and e[1].device == w.device:
            return e
        if len(cls._entries) > 64:      # dead references of models long gone
            cls._entries = {k: v for k, v in cls._entries.items() if v[0]() is not None}
        e = [weakref.ref(w), torch.empty((w.shape[1], w.shape[0]), dtype=w.dtype, device=w.device), None]
        cls._entries[id(w)] = e
        return e

    @classmethod
    def get(cls, w):
        e = cls._entry(w)
        key = (w.data_ptr(), w._version)
        if e[2] != key:
            e[1].copy_(w.detach().t())
            e[2] = key
        return e[1]

    @classmethod
    def buffer_for(cls, w):
        """The ``[cols, rows]`` buffer an optimizer may fill with the transposed NEW values of ``w``,
        or None when nobody has asked for ``w``'s transpose (nothing to keep current)."""
        e = cls._entries.get(id(w))
        if e is None or e[0]() is not w or e[1].device != w.device:
            return None
        return e[1]

    @classmethod
    def written(cls, w):
        e = cls._entries.get(id(w))
        if e is not None and e[0]() is w:
            e[2] = (w.data_ptr(), w._version)


class _LinearNoBias(torch.autograd.Function):
    """``x @ W.T`` (torch.nn.functional.linear without bias) for the update-size trunk, all three
    GEMMs of forward + backward on the bf16 matrix pipe (rlpyt/models/mlp.py:24-31 under
    autograd): forward ``gemm_nt(x, W)``, input gradient ``gemm_nt(g, W^T copy)``, weight gradient
    ``gemm_tn(g, x)``.  No vendor GEMM."""

    # gemm_nt runs one 128 x 128 tile per CU and has no K split: below ~200 tiles (R2D1's trunk FC at
    # 2 560 - 5 440 rows: 80 - 172 tiles x 216 K steps) the chip is part empty for the whole contraction
    # and the library's smaller tiles are as fast or faster (300 vs 286 - 319 us); both gradient GEMMs
    # have thousands of tiles / eight K chunks at those sizes and stay on the own kernels
    MIN_FORWARD_TILES = 200

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        tiles = -(-x.shape[0] // 128) * -(-weight.shape[0] // 128)
        if tiles < _LinearNoBias.MIN_FORWARD_TILES:
            return torch.nn.functional.linear(x, weight.detach())
        return gemm_nt(x, weight.detach())

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            # g W as g (W^T)^T on the same kernel; W^T from the mirror the optimizer keeps current
            # (a 7 MB transposing copy only when somebody else changed W)
            gx = gemm_nt(g, TransposedMirror.get(weight), region="gemm_nt_dgrad")
        if ctx.needs_input_grad[1]:
            gw = gemm_tn(g, x)
        return gx, gw


def linear_nobias(x, weight):
    """``F.linear(x, weight)`` with ``gemm_nt`` forward / input-gradient (see ``_LinearNoBias``);
    shapes it does not cover (K or N not a multiple of 32, non-f32, CPU) take ``F.linear``."""
    if (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 2
            and x.shape[1] % 32 == 0 and weight.shape[0] % 32 == 0 and x.shape[0] % 32 == 0):
        return _LinearNoBias.apply(x, weight)
    return torch.nn.functional.linear(x, weight)


def fc_small_partials(x, weight):
    """Split-K partial products ``[ksplit, M, N]`` of ``x @ weight.T`` (no bias / activation):
    the first half of ``fc_small``, for consumers that finish the sum themselves."""
    _lib.require_gpu()
    x = _f32(x)
    M, K = x.shape
    N = weight.shape[0]
    w = _f32(weight.detach())
    ws = _workspace("fc_small", lib.rlpyt_fc_small_workspace_bytes(M, N), x.device)
    check(lib.rlpyt_fc_small_f32(ptr(x), ptr(w), None, None, M, N, K, 0, ptr(ws), stream()),
          "rlpyt_fc_small_f32")
    return ws, lib.rlpyt_fc_small_ksplit(K)


def is_weights(priorities, eps, beta):
    """Importance-sampling weights ``float((1 / (p + eps)) ** beta / max)`` of a prioritized batch, float64
    inside, one launch (``rlpyt_is_weights_f64``); ``beta``: a python float or a float64 device scalar
    tensor (captured update graphs)."""
    _lib.require_gpu()
    assert priorities.dtype == torch.float64 and priorities.is_contiguous() and priorities.is_cuda
    n = priorities.numel()
    out = torch.empty(priorities.shape, dtype=torch.float32, device=priorities.device)
    if isinstance(beta, torch.Tensor):
        assert beta.dtype == torch.float64 and beta.device == priorities.device and beta.numel() == 1
        check(lib.rlpyt_is_weights_f64(ptr(priorities), n, float(eps), ptr(beta), 0.0, ptr(out), stream()),
              "rlpyt_is_weights_f64")
    else:
        check(lib.rlpyt_is_weights_f64(ptr(priorities), n, float(eps), None, float(beta), ptr(out), stream()),
              "rlpyt_is_weights_f64")
    return out


def eps_greedy(q, eps, uniforms, t_dev=None):
    """Epsilon-greedy actions for Q-values ``q [n, A]`` from pre-drawn uniforms ``[T, n]`` (row
    ``t_dev[0]``, or row 0): ``u < eps ? floor(u / eps * A) : argmax`` in one launch
    (``rlpyt_eps_greedy_f32``).  ``eps``: f32 device tensor with 1 or n entries."""
    _lib.require_gpu()
    q = _f32(q)
    n, A = q.shape
    eps = _f32(eps).reshape(-1)
    assert eps.numel() in (1, n) and eps.device == q.device
    uniforms = _f32(uniforms)
    assert uniforms.shape[-1] == n and uniforms.device == q.device
    action = torch.empty(n, dtype=torch.int64, device=q.device)
    check(lib.rlpyt_eps_greedy_f32(ptr(q), n, A, ptr(eps), 0 if eps.numel() == 1 else 1,
                                   ptr(uniforms), ptr(t_dev), ptr(action), stream()),
          "rlpyt_eps_greedy_f32")
    return action


class LstmStep:
    """One step of a single-layer ``torch.nn.LSTM`` (no grad) for the per-time-step sampling forward
    of the recurrent agents: ``step(parts, h, c) -> (h', c')`` with ``parts`` the pieces of the input
    row ``x = cat(parts, 1)`` (``[B, *]`` each), ``h, c [B, H]``.

    The gate GEMM ``[x | h] [W_ih | W_hh]^T`` runs as the split-K small-batch kernel
    (``rlpyt_fc_small_f32`` partials), bias + gates + cell as one more launch
    (``rlpyt_lstm_cell_f32``): one concatenation and two launches instead of the library RNN's
    ten-odd kernels per time step.  The concatenated, K-padded weight lives in ONE buffer for the
    life of the object (captured hipGraphs keep its address); ``refresh()`` copies the module's
    current parameters into it when their versions changed -- eager calls do that themselves, a
    captured step graph relies on the agent calling it once per iteration (``BaseAgent.sample_mode``)."""

    def __init__(self, lstm):
        assert lstm.num_layers == 1 and not lstm.bidirectional and lstm.bias
        self.lstm = lstm
        w_ih, w_hh = lstm.weight_ih_l0, lstm.weight_hh_l0
        self.I, self.H = w_ih.shape[1], w_hh.shape[1]
        self.K = self.I + self.H
        self.Kp = (self.K + 15) // 16 * 16
        self.wc = torch.zeros((4 * self.H, self.Kp), dtype=torch.float32, device=w_ih.device)
        self._key = None
        self._pad = {}
        self.refresh(force=True)

    def refresh(self, force=False):
        """Copy the module's current parameters into the concatenated buffer if they changed.  Not
        while a graph is being captured (the copies would become graph nodes that re-read the
        parameters on every replay) -- except for the very first fill, which must never be skipped."""
        lstm = self.lstm
        w_ih, w_hh = lstm.weight_ih_l0, lstm.weight_hh_l0
        key = (w_ih._version, w_hh._version, w_ih.data_ptr(), w_hh.data_ptr())
        if not force and (key == self._key or torch.cuda.is_current_stream_capturing()):
            return
        with torch.no_grad():
            self.wc[:, :self.I].copy_(w_ih)
            self.wc[:, self.I:self.K].copy_(w_hh)
        self._key = key

    def step(self, parts, h, c):
        _lib.require_gpu()
        self.refresh()
        B = h.shape[0]
        parts = [p.reshape(B, -1).float() for p in parts] + [h.reshape(B, self.H)]
        if self.Kp != self.K:
            pad = self._pad.get(B)
            if pad is None:
                pad = torch.zeros((B, self.Kp - self.K), dtype=torch.float32, device=h.device)
                self._pad[B] = pad
            parts.append(pad)
        xh = torch.cat(parts, dim=1)
        assert xh.shape[1] == self.Kp, "input pieces do not add up to the LSTM's input size"
        partial, ksplit = fc_small_partials(xh, self.wc)
        c = _f32(c.reshape(B, self.H))
        h1, c1 = torch.empty_like(c), torch.empty_like(c)
        lstm = self.lstm
        check(lib.rlpyt_lstm_cell_f32(ptr(partial), ksplit, ptr(lstm.bias_ih_l0), ptr(lstm.bias_hh_l0),
                                      ptr(c), ptr(h1), ptr(c1), B, self.H, stream()),
              "rlpyt_lstm_cell_f32")
        return h1, c1

    def step_rows(self, xh, h_state, c_state):
        """The step on rows assembled by ``rnn_step_inputs`` (``xh [B, Kp]``), the new state written
        IN PLACE into ``h_state`` / ``c_state`` ``[B, H]`` (persistent buffers of the sampler's
        pipeline group: no copy-back, a captured graph keeps their addresses)."""
        _lib.require_gpu()
        self.refresh()
        B = xh.shape[0]
        assert xh.shape[1] == self.Kp and h_state.is_contiguous() and c_state.is_contiguous()
        assert h_state.numel() == B * self.H == c_state.numel()
        partial, ksplit = fc_small_partials(xh, self.wc)
        lstm = self.lstm
        check(lib.rlpyt_lstm_cell_f32(ptr(partial), ksplit, ptr(lstm.bias_ih_l0), ptr(lstm.bias_hh_l0),
                                      ptr(c_state), ptr(h_state), ptr(c_state), B, self.H, stream()),
              "rlpyt_lstm_cell_f32")


def rnn_step_inputs(feat, action, reward, done, h, c, n_actions, Kp, relu=False):
    """``[act(feat) | onehot(action) | reward | h | 0-pad]`` rows ``[B, Kp]`` of one recurrent
    sampling step in one launch, with the reset handling folded in (``rlpyt_rnn_step_inputs_f32``):
    rows with ``done`` see the null action 0, zero reward and a zero state; ``c`` is zeroed in place
    there.  Returns ``(xh, prev_h, prev_c)``, the last two being the state the step starts from."""
    _lib.require_gpu()
    B, F = feat.shape
    H = h.shape[-1]
    assert feat.dtype == torch.float32 and feat.is_contiguous()
    assert action.dtype == torch.int64 and action.numel() == B and action.is_contiguous()
    assert reward.dtype == torch.float32 and reward.numel() == B and reward.is_contiguous()
    assert h.dtype == torch.float32 and c.dtype == torch.float32 and h.numel() == B * H == c.numel()
    assert h.is_contiguous() and c.is_contiguous()
    if done is not None:
        assert done.dtype in (torch.bool, torch.uint8) and done.numel() == B and done.is_contiguous()
        done = done.view(torch.uint8)
    xh = torch.empty((B, Kp), dtype=torch.float32, device=feat.device)
    prev_h = torch.empty((B, H), dtype=torch.float32, device=feat.device)
    prev_c = torch.empty((B, H), dtype=torch.float32, device=feat.device)
    check(lib.rlpyt_rnn_step_inputs_f32(ptr(feat), F, int(bool(relu)), ptr(action), int(n_actions),
                                        ptr(reward), ptr(done), ptr(h), ptr(c), H, ptr(xh), int(Kp),
                                        ptr(prev_h), ptr(prev_c), B, stream()),
          "rlpyt_rnn_step_inputs_f32")
    return xh, prev_h, prev_c


LSTM_SEQ_HIDDEN = (256, 512)     # hidden sizes rlpyt_lstm_seq_f32 is instantiated for


def lstm_sequence_ok(lstm, x, h0):
    """Whether ``lstm_sequence`` serves this call: no autograd, single-layer f32 LSTM on the device,
    a hidden size the kernel is built for, ``x [T,B,I]``."""
    return (not torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and x.dim() == 3
            and lstm.num_layers == 1 and not lstm.bidirectional and lstm.bias
            and not lstm.batch_first and getattr(lstm, "proj_size", 0) == 0
            and lstm.hidden_size in LSTM_SEQ_HIDDEN and lstm.weight_hh_l0.dtype == torch.float32
            and (h0 is None or (h0.dim() == 3 and h0.shape[0] == 1 and h0.shape[1] == x.shape[1])))


def lstm_sequence(lstm, x, h0=None, c0=None):
    """No-grad ``torch.nn.LSTM`` forward over a sequence ``x [T,B,I]`` (state ``[1,B,H]`` or None):
    one library GEMM for the input projection of all time steps, then ONE launch per time step
    (``rlpyt_lstm_seq_f32``: recurrent product + gates + cell) instead of the library RNN's five.
    Returns ``(out [T,B,H], (h_T [1,B,H], c_T [1,B,H]))`` like the module."""
    _lib.require_gpu()
    T, B, _ = x.shape
    H = lstm.hidden_size
    xproj = torch.addmm(lstm.bias_ih_l0 + lstm.bias_hh_l0, x.reshape(T * B, -1), lstm.weight_ih_l0.t())
    h0 = (torch.zeros((B, H), dtype=torch.float32, device=x.device) if h0 is None
          else _f32(h0.reshape(B, H)))
    c = (torch.zeros((B, H), dtype=torch.float32, device=x.device) if c0 is None
         else c0.reshape(B, H).to(torch.float32).clone())
    out = torch.empty((T, B, H), dtype=torch.float32, device=x.device)
    w_hh = lstm.weight_hh_l0
    assert w_hh.is_contiguous()
    check(lib.rlpyt_lstm_seq_f32(ptr(xproj), ptr(w_hh), ptr(h0), ptr(c), ptr(out), T, B, H, stream()),
          "rlpyt_lstm_seq_f32")
    hn = out[T - 1:T] if T > 0 else h0.unsqueeze(0)
    return out, (hn, c.unsqueeze(0))


class _LstmSeqTrain(torch.autograd.Function):
    """``torch.nn.LSTM`` (one layer) over ``x [T,B,I]`` UNDER AUTOGRAD on the own step kernels
    (csrc/lstm_seq.hip): forward = input projection of all steps (one GEMM) + one launch per step that
    keeps the activated gates and cell states; backward = one launch per step (recurrent gradient +
    the cell's pointwise backward) + four GEMMs over all steps for dx / dW_ih / dW_hh and the bias sums.
    The online network's training pass of R2D1 (rlpyt/algos/dqn/r2d1.py:286-334 through
    rlpyt/models/dqn/atari_r2d1_model.py:61-63)."""

    @staticmethod
    def forward(ctx, x, h0, c0, w_ih, w_hh, b_ih, b_hh):
        _lib.require_gpu()
        T, B, I = x.shape
        H = w_hh.shape[1]
        x2 = _f32(x).reshape(T * B, I)
        w_ih_d, w_hh_d = w_ih.detach().contiguous(), w_hh.detach().contiguous()
        xproj = torch.addmm(b_ih.detach() + b_hh.detach(), x2, w_ih_d.t())
        h0 = _f32(h0.detach().reshape(B, H))
        c0 = _f32(c0.detach().reshape(B, H))
        c = c0.clone()
        out = torch.empty((T, B, H), dtype=torch.float32, device=x.device)
        gates = torch.empty((T, B, H, 4), dtype=torch.float32, device=x.device)
        c_all = torch.empty((T, B, H), dtype=torch.float32, device=x.device)
        check(lib.rlpyt_lstm_seq_train_f32(ptr(xproj), ptr(w_hh_d), ptr(h0), ptr(c), ptr(out), ptr(gates),
                                           ptr(c_all), T, B, H, stream()), "rlpyt_lstm_seq_train_f32")
        ctx.save_for_backward(x2, h0, c0, w_ih_d, w_hh_d, out, gates, c_all)
        ctx.dims = (T, B, I, H)
        hn = out[T - 1].clone()
        return out, hn, c

    @staticmethod
    def backward(ctx, dout, dhn, dcn):
        x2, h0, c0, w_ih, w_hh, out, gates, c_all = ctx.saved_tensors
        T, B, I, H = ctx.dims
        dout = None if dout is None else _f32(dout)
        dhn = None if dhn is None else _f32(dhn)
        dc = torch.zeros((B, H), dtype=torch.float32, device=x2.device) if dcn is None else _f32(dcn).clone()
        dgates = torch.empty((T, B, 4 * H), dtype=torch.float32, device=x2.device)
        need_state = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dh0 = torch.empty((B, H), dtype=torch.float32, device=x2.device) if need_state else None
        w_hh_t = w_hh.t().contiguous()
        check(lib.rlpyt_lstm_seq_bwd_f32(ptr(dout), ptr(dhn), ptr(gates), ptr(c_all), ptr(c0), ptr(w_hh_t),
                                         ptr(dc), ptr(dgates), ptr(dh0), T, B, H, stream()),
              "rlpyt_lstm_seq_bwd_f32")
        dg2 = dgates.view(T * B, 4 * H)
        dx = torch.mm(dg2, w_ih).view(T, B, I) if ctx.needs_input_grad[0] else None
        dw_ih = torch.mm(dg2.t(), x2)
        h_prev = torch.cat([h0.unsqueeze(0), out[:T - 1]], dim=0).view(T * B, H)
        dw_hh = torch.mm(dg2.t(), h_prev)
        db = dg2.sum(dim=0)
        return dx, dh0, (dc if need_state else None), dw_ih, dw_hh, db, db


def lstm_sequence_train_ok(lstm, x, h0):
    """Whether ``lstm_sequence_train`` serves this call (``lstm_sequence_ok`` under autograd)."""
    return (torch.is_grad_enabled() and LSTM_SEQ_TRAIN and x.is_cuda and x.dtype == torch.float32
            and x.dim() == 3 and x.shape[0] > 0
            and lstm.num_layers == 1 and not lstm.bidirectional and lstm.bias
            and not lstm.batch_first and getattr(lstm, "proj_size", 0) == 0 and lstm.dropout == 0
            and lstm.hidden_size in LSTM_SEQ_HIDDEN and lstm.weight_hh_l0.dtype == torch.float32
            and (h0 is None or (h0.dim() == 3 and h0.shape[0] == 1 and h0.shape[1] == x.shape[1])))


# A/B switch: the sequence under autograd through the library RNN instead
LSTM_SEQ_TRAIN = os.environ.get("RLPYT_LSTM_SEQ_TRAIN", "1") != "0"


def lstm_sequence_train(lstm, x, h0=None, c0=None):
    """``lstm(x, (h0, c0))`` differentiable w.r.t. x, the initial state and the four parameters, on the
    own kernels (``_LstmSeqTrain``).  Returns ``(out [T,B,H], (h_T [1,B,H], c_T [1,B,H]))``."""
    T, B, _ = x.shape
    H = lstm.hidden_size
    if h0 is None:
        h0 = torch.zeros((B, H), dtype=torch.float32, device=x.device)
    if c0 is None:
        c0 = torch.zeros((B, H), dtype=torch.float32, device=x.device)
    out, hn, cn = _LstmSeqTrain.apply(x, h0.reshape(B, H), c0.reshape(B, H), lstm.weight_ih_l0,
                                      lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0)
    return out, (hn.unsqueeze(0), cn.unsqueeze(0))


def update_tick(ctr, table, hyper_cur, idx_all, idx_static, tick_idx):
    """First launch of a captured minibatch update (``rlpyt_update_tick``): row ``*ctr`` of the
    per-update hyper-parameter ``table [n, cols]`` -> ``hyper_cur [cols]``; the update's index chunk
    ``idx_all[cur * M : (cur + 1) * M]`` -> ``idx_static [M]``; ``tick_idx[0] = cur``."""
    _lib.require_gpu()
    n_rows, n_cols = table.shape
    assert table.dtype == torch.float32 and table.is_contiguous() and hyper_cur.numel() >= n_cols
    assert ctr.dtype == torch.int64 and tick_idx.dtype == torch.int64
    M = 0 if idx_static is None else idx_static.numel()
    if idx_all is not None:
        assert idx_all.dtype == torch.int64 and idx_all.is_contiguous() and idx_all.numel() >= n_rows * M
    check(lib.rlpyt_update_tick(ptr(ctr), ptr(table), int(n_rows), int(n_cols), ptr(hyper_cur),
                                ptr(idx_all), ptr(idx_static), int(M), ptr(tick_idx), stream()),
          "rlpyt_update_tick")


def rollout_fc_partials(x, weight):
    """Round-4 trunk of the rollout step: split-K partials ``[ksplit, M, N]`` of ``x @ weight.T``
    in slices of 128 along K, one workgroup per (64 columns, slice, 64 rows)
    (``rlpyt_rollout_fc_f32``); ``rollout_head`` finishes the sum.  Returns (partials, ksplit)."""
    _lib.require_gpu()
    x = _f32(x)
    M, K = x.shape
    N = weight.shape[0]
    w = _f32(weight.detach())
    ws = _workspace("rollout_fc", lib.rlpyt_rollout_fc_workspace_bytes(M, N, K), x.device)
    check(lib.rlpyt_rollout_fc_f32(ptr(x), ptr(w), ptr(ws), M, N, K, stream()),
          "rlpyt_rollout_fc_f32")
    return ws, lib.rlpyt_rollout_fc_ksplit(K)


def rollout_fc_ok(M, N, K):
    return 0 < M <= 1024 and N % 64 == 0 and K % 16 == 0 and 0 < K <= 4096


def rollout_head(partial, ksplit, fc_bias, w_pi, b_pi, w_v, b_v, uniforms, t_dev, n, prob_rows,
                 value_rows, action_rows, lo, action_out, bootstrap_out=None):
    """Trunk finish + heads + softmax + draw + the step's row writes, one workgroup per row
    (``rlpyt_rollout_head_f32``); ``bootstrap_out`` ([n] f32): only the value head, written there
    (every row / uniform argument may then be None)."""
    _lib.require_gpu()
    A, K = w_pi.shape
    if bootstrap_out is None:
        B = prob_rows.shape[1]
        assert prob_rows.is_contiguous() and value_rows.is_contiguous() and action_rows.is_contiguous()
        assert action_rows.dtype == torch.int64 and action_out.dtype == torch.int64
        assert uniforms.shape[-1] == n and uniforms.is_contiguous()
    else:
        B = 0
        assert bootstrap_out.dtype == torch.float32 and bootstrap_out.is_contiguous()
        assert bootstrap_out.numel() == n
    check(lib.rlpyt_rollout_head_f32(
        ptr(partial), int(ksplit), ptr(_f32(fc_bias.detach())), ptr(_f32(w_pi.detach())),
        ptr(_f32(b_pi.detach())), ptr(_f32(w_v.detach()).reshape(-1)),
        ptr(_f32(b_v.detach()).reshape(-1)), ptr(uniforms), ptr(t_dev), int(n), K, A,
        ptr(prob_rows), ptr(value_rows), ptr(action_rows), B, int(lo), ptr(action_out),
        ptr(bootstrap_out), stream()), "rlpyt_rollout_head_f32")


def frame_push(obs, t_dev, lo, new_frame, full_rows, slot, stage=None, scalar_rows=None):
    """Rebuild row ``t`` (device counter) of a frame-stacked uint8 observation batch
    ``[T,B,C,*img]`` for columns ``lo:lo+Bg`` from the newest frames ``[Bg,*img]``:
    shifted previous stack + new frame, or a full row ``full_rows[slot[b]]`` where
    ``slot[b] >= 0`` (reset envs, first step of a batch).  ``scalar_rows`` =
    ``(all_reward [T',B] f32, reward_src [Bg], all_done [T',B] bool, done_src [Bg])`` commits
    the step's reward / done rows in the same launch."""
    _lib.require_gpu()
    assert obs.dtype == torch.uint8 and obs.is_contiguous() and obs.dim() >= 4
    B, C = obs.shape[1], obs.shape[2]
    HW = 1
    for d in obs.shape[3:]:
        HW *= d
    Bg = new_frame.shape[0]
    assert slot.dtype == torch.int32 and slot.numel() == Bg
    rr = rs = dr = ds = None
    if scalar_rows is not None:
        rr, rs, dr, ds = scalar_rows
        assert rr.dtype == torch.float32 and rr.shape[1] == B and dr.shape[1] == B
        dr, ds = dr.view(torch.uint8), ds.view(torch.uint8)
    check(lib.rlpyt_frame_push(ptr(obs), ptr(t_dev), B, int(lo), Bg, C, HW, ptr(new_frame),
                               ptr(full_rows), ptr(slot), ptr(stage), ptr(rr), ptr(rs), ptr(dr),
                               ptr(ds), stream()), "rlpyt_frame_push")


def atari_sample_convs(obs, t_dev, lo, new_frame, full_rows, slot, w1, b1, w2, b2,
                       scalar_rows=None, scale=1. / 255, out=None, dst_stage=None):
    """``frame_push`` + conv1 + conv2 of the AtariFfModel geometry in one launch (sampling
    forward, no grad): rebuilds row ``t`` of ``obs [T,B,4,104,80]`` for columns ``lo:lo+Bg``
    exactly like ``frame_push`` and returns the conv features ``[Bg, 3456]`` of the rebuilt
    stacks (bit-identical to ``atari_conv_stack`` on that row).  ``dst_stage`` (u8
    ``[Bg,4,104,80]``): the rebuilt stacks go there instead of ``obs[t]`` (bootstrap-value pass at
    ``t = T``, which reads ``obs[T-1]`` but must not write a row ``T``)."""
    _lib.require_gpu()
    assert obs.dtype == torch.uint8 and obs.is_contiguous() and tuple(obs.shape[2:]) == (4, 104, 80)
    if dst_stage is not None:
        assert (dst_stage.dtype == torch.uint8 and dst_stage.is_contiguous()
                and tuple(dst_stage.shape) == (new_frame.shape[0], 4, 104, 80))
    B, Bg = obs.shape[1], new_frame.shape[0]
    assert slot.dtype == torch.int32 and slot.numel() == Bg
    rr = rs = dr = ds = None
    if scalar_rows is not None:
        rr, rs, dr, ds = scalar_rows
        assert rr.dtype == torch.float32 and rr.shape[1] == B and dr.shape[1] == B
        dr, ds = dr.view(torch.uint8), ds.view(torch.uint8)
    if out is None:
        out = torch.empty((Bg, 3456), dtype=torch.float32, device=obs.device)
    check(lib.rlpyt_atari_sample_convs_to_f32(
        ptr(obs), ptr(t_dev), B, int(lo), Bg, ptr(new_frame), ptr(full_rows), ptr(slot), ptr(rr),
        ptr(rs), ptr(dr), ptr(ds), ptr(w1.contiguous()), ptr(b1), ptr(w2.contiguous()), ptr(b2),
        float(scale), ptr(out), ptr(dst_stage), stream()), "rlpyt_atari_sample_convs_to_f32")
    return out


def dqn_convs_pack(w1, w2, w3, out=None):
    """The DQN conv stack's weights in the kernels' register order (``rlpyt_dqn_convs_pack_f32``):
    made once per iteration by models that step environments, passed to ``dqn_convs_fwd`` as
    ``packed`` -- valid as long as the three weights do not change."""
    _lib.require_gpu()
    n = int(lib.rlpyt_dqn_convs_packed_floats())
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=w1.device)
    assert out.numel() == n and out.dtype == torch.float32 and out.is_contiguous()
    for x in (w1, w2, w3):
        assert x.dtype == torch.float32 and x.is_contiguous() and x.device == out.device
    check(lib.rlpyt_dqn_convs_pack_f32(ptr(w1), ptr(w2), ptr(w3), ptr(out), stream()),
          "rlpyt_dqn_convs_pack_f32")
    return out


def dqn_convs_fwd(obs, w1, b1, w2, b2, w3, b3, scale=1. / 255, out=None, packed=None):
    """No-grad forward of the DQN-family conv stack (Conv2d(4,32,8,s4) / (32,64,4,s2,p1) /
    (64,64,3,s1,p1), ReLU after each; rlpyt/models/dqn/atari_dqn_model.py:30-37) on uint8 frames
    ``[N,4,104,80]``: returns ``[N, 6912]`` in the order of ``conv(img).view(N, -1)``.  Weights in
    the torch layout; without ``packed`` they are re-packed on the stream in front of the layer
    kernels, so the call (also as a captured graph node) always sees the current parameters; with
    ``packed`` (``dqn_convs_pack`` of the same weights) that launch is skipped."""
    _lib.require_gpu()
    assert obs.dtype == torch.uint8 and obs.is_contiguous() and tuple(obs.shape[1:]) == (4, 104, 80)
    assert tuple(w1.shape) == (32, 4, 8, 8) and tuple(w2.shape) == (64, 32, 4, 4)
    assert tuple(w3.shape) == (64, 64, 3, 3)
    for x in (w1, b1, w2, b2, w3, b3):
        assert x.dtype == torch.float32 and x.is_contiguous() and x.device == obs.device
    N = obs.shape[0]
    ws = torch.empty(int(lib.rlpyt_dqn_convs_workspace_floats(N)), dtype=torch.float32,
                     device=obs.device)
    if out is None:
        out = torch.empty((N, 6912), dtype=torch.float32, device=obs.device)
    assert out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == (N, 6912)
    check(lib.rlpyt_dqn_convs_fwd_f32(ptr(obs), N, ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(w3),
                                      ptr(b3), ptr(packed), float(scale), ptr(ws), ptr(out), stream()),
          "rlpyt_dqn_convs_fwd_f32")
    return out


class _DqnConvStack(torch.autograd.Function):
    """The DQN conv stack UNDER AUTOGRAD at update-batch sizes (the online network's pass of
    rlpyt/algos/dqn/dqn.py:176-180 through rlpyt/models/dqn/atari_dqn_model.py:33-45): the forward is
    the three own layer kernels of ``dqn_convs_fwd`` (uint8 in, bias + ReLU in the epilogues, no
    conversion / bias / clamp / layout launches), whose channels-last activations y1 / y2 stay in the
    call's workspace for the backward pass."""

    @staticmethod
    def forward(ctx, obs, w1, b1, w2, b2, w3, b3, scale):
        _lib.require_gpu()
        N = obs.shape[0]
        params = tuple(x.detach().contiguous() for x in (w1, b1, w2, b2, w3, b3))
        ws = torch.empty(int(lib.rlpyt_dqn_convs_workspace_floats(N)), dtype=torch.float32,
                         device=obs.device)
        out = torch.empty((N, 6912), dtype=torch.float32, device=obs.device)
        check(lib.rlpyt_dqn_convs_fwd_f32(ptr(obs), N, *(ptr(x) for x in params), None, float(scale),
                                          ptr(ws), ptr(out), stream()), "rlpyt_dqn_convs_fwd_f32")
        if any(ctx.needs_input_grad):
            ctx.save_for_backward(obs, params[0], params[2], params[4], ws, out)
            ctx.scale = float(scale)
        return out

    @staticmethod
    def backward(ctx, g):
        obs, w1, w2, w3, ws, out = ctx.saved_tensors
        N = obs.shape[0]
        p0 = int(lib.rlpyt_dqn_convs_packed_floats())
        n1 = N * 475 * 32
        if DQN_CONVS_OWN_BWD:
            # the own backward kernels (csrc/dqn_convs_bwd.hip): masks, bias sums and the three weight
            # gradients from the kept channels-last activations, one call
            g = _f32(g)
            dws = torch.empty(int(lib.rlpyt_dqn_convs_bwd_workspace_floats(N)), dtype=torch.float32,
                              device=obs.device)
            grads = [torch.empty_like(w1), torch.empty(32, dtype=torch.float32, device=obs.device),
                     torch.empty_like(w2), torch.empty(64, dtype=torch.float32, device=obs.device),
                     torch.empty_like(w3), torch.empty(64, dtype=torch.float32, device=obs.device)]
            check(lib.rlpyt_dqn_convs_bwd_f32(
                ptr(obs), N, ptr(w2), ptr(w3), ptr(ws[p0:]), ptr(ws[p0 + n1:]), ptr(out), ptr(g),
                ctx.scale, ptr(dws), *(ptr(x) for x in grads), stream()), "rlpyt_dqn_convs_bwd_f32")
            return (None, *grads, None)
        # logical NCHW views of the kernels' channels-last activations
        y1 = ws[p0:p0 + n1].view(N, 25, 19, 32).permute(0, 3, 1, 2)
        y2 = ws[p0 + n1:p0 + n1 + N * 108 * 64].view(N, 12, 9, 64).permute(0, 3, 1, 2)
        y3 = out.view(N, 64, 12, 9)
        cb = torch.ops.aten.convolution_backward
        dz3 = (_f32(g).view(N, 64, 12, 9) * (y3 > 0)).contiguous(memory_format=torch.channels_last)
        dy2, dw3, db3 = cb(dz3, y2, w3, [64], [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, True, True])
        dz2 = dy2 * (y2 > 0)
        dy1, dw2, db2 = cb(dz2, y1, w2, [64], [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [True, True, True])
        dz1 = dy1 * (y1 > 0)
        x0 = obs_to_nhwc_f32(obs, scale=ctx.scale)
        _, dw1, db1 = cb(dz1, x0, w1, [32], [4, 4], [0, 0], [1, 1], False, [0, 0], 1, [False, True, True])
        return None, dw1, db1, dw2, db2, dw3, db3, None


# A/B switch: the conv stack's backward through the library's convolution_backward instead
DQN_CONVS_OWN_BWD = os.environ.get("RLPYT_DQN_OWN_BWD", "1") != "0"


def dqn_convs(obs, w1, b1, w2, b2, w3, b3, scale=1. / 255):
    """``dqn_convs_fwd`` differentiable w.r.t. the six conv parameters (uint8 ``[N,4,104,80]`` in,
    ``[N, 6912]`` out in the order of ``conv(img).view(N, -1)``)."""
    assert obs.dtype == torch.uint8 and obs.is_contiguous() and tuple(obs.shape[1:]) == (4, 104, 80)
    return _DqnConvStack.apply(obs, w1, b1, w2, b2, w3, b3, scale)


def mlp_q_head_ok(x, lin1, lin2):
    """Whether ``mlp_q_head`` serves ``lin2(relu(lin1(x)))``: no autograd, f32 on the device, at most
    256 rows, hidden width 256 / 512, at most 18 outputs."""
    return (not torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2
            and 0 < x.shape[0] <= 256 and lin1.out_features in (256, 512) and lin1.in_features % 16 == 0
            and 0 < lin2.out_features <= 18 and lin1.bias is not None and lin2.bias is not None
            and lin1.weight.dtype == torch.float32 and lin1.weight.is_cuda)


def mlp_q_head(x, lin1, lin2):
    """``lin2(relu(lin1(x)))`` for the Q-value head of a sampling / target pass: the hidden layer as
    split-K partials (``rlpyt_fc_small_f32``), their sum + bias + ReLU + the output dot products in
    ``rlpyt_q_head_f32`` -- two launches instead of GEMM + clamp + GEMM."""
    _lib.require_gpu()
    partial, ksplit = fc_small_partials(x, lin1.weight)
    n, A, K = x.shape[0], lin2.out_features, lin1.out_features
    q = torch.empty((n, A), dtype=torch.float32, device=x.device)
    check(lib.rlpyt_q_head_f32(ptr(partial), ksplit, ptr(lin1.bias), ptr(_f32(lin2.weight.detach())),
                               ptr(lin2.bias), n, K, A, ptr(q), stream()), "rlpyt_q_head_f32")
    return q


class _MlpQHeadTrain(torch.autograd.Function):
    """``lin2(relu(lin1(x)))`` under autograd on the own kernels: forward = split-K hidden layer +
    ``rlpyt_q_head_train_f32`` (keeps h); backward = ``rlpyt_q_head_bwd_f32`` (output layer's gradients, the
    ReLU mask, the hidden bias gradient: one launch for what autograd issues as eight) + two GEMMs."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        n, K, A = x.shape[0], w1.shape[0], w2.shape[0]
        partial, ksplit = fc_small_partials(x, w1)
        q = torch.empty((n, A), dtype=torch.float32, device=x.device)
        h = torch.empty((n, K), dtype=torch.float32, device=x.device)
        check(lib.rlpyt_q_head_train_f32(ptr(partial), ksplit, ptr(b1), ptr(w2), ptr(b2), n, K, A, ptr(q),
                                         ptr(h), stream()), "rlpyt_q_head_train_f32")
        ctx.save_for_backward(x, w1, w2, h)
        return q

    @staticmethod
    def backward(ctx, dq):
        x, w1, w2, h = ctx.saved_tensors
        n, K, A = x.shape[0], w1.shape[0], w2.shape[0]
        dq = _f32(dq)
        dev = x.device
        dw2 = torch.empty((A, K), dtype=torch.float32, device=dev)
        db2 = torch.empty(A, dtype=torch.float32, device=dev)
        dh = torch.empty((n, K), dtype=torch.float32, device=dev)
        db1 = torch.empty(K, dtype=torch.float32, device=dev)
        check(lib.rlpyt_q_head_bwd_f32(ptr(dq), ptr(h), ptr(w2), n, K, A, ptr(dw2), ptr(db2), ptr(dh),
                                       ptr(db1), stream()), "rlpyt_q_head_bwd_f32")
        dx = dh @ w1 if ctx.needs_input_grad[0] else None
        dw1 = dh.t() @ x      # (gemm_tn here measured neutral: profiles/r6_ab_q_head_wgrad_tn.jsonl)
        return dx, dw1, db1, dw2, db2


def mlp_q_head_train_ok(x, lin1, lin2):
    """Whether ``mlp_q_head_train`` serves ``lin2(relu(lin1(x)))`` under autograd (the online network's head
    in a DQN update): what ``mlp_q_head_ok`` asks for, contiguous parameters, hidden width a multiple of 64."""
    return (torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2
            and 0 < x.shape[0] <= 256 and lin1.out_features in (256, 512) and lin1.in_features % 16 == 0
            and 0 < lin2.out_features <= 18 and lin1.bias is not None and lin2.bias is not None
            and lin1.weight.dtype == torch.float32 and lin1.weight.is_cuda
            and lin1.weight.is_contiguous() and lin2.weight.is_contiguous())


def mlp_q_head_train(x, lin1, lin2):
    _lib.require_gpu()
    return _MlpQHeadTrain.apply(x.contiguous(), lin1.weight, lin1.bias, lin2.weight, lin2.bias)


def categorical_head(h, w_pi, b_pi, w_v=None, b_v=None, uniforms=None, u_row=None):
    """Policy / value heads + softmax (+ inverse-CDF action sampling when ``uniforms`` is
    given) in one kernel -- the no-grad sampling forward of
    rlpyt/models/pg/atari_ff_model.py:56-58 + rlpyt/distributions/categorical.py:28-31.
    ``u_row`` (int64 device scalar tensor): ``uniforms`` is a ``[T', n]`` table and row
    ``*u_row`` is used -- lets a captured hipGraph sample without any RNG state inside.
    Returns ``(prob [n,A], value [n] | None, action int64 [n] | None)``."""
    _lib.require_gpu()
    h = _f32(h)
    n, K = h.shape
    A = w_pi.shape[0]
    w_pi, b_pi = _f32(w_pi.detach()), _f32(b_pi.detach())
    prob = torch.empty((n, A), dtype=torch.float32, device=h.device)
    value = action = wv = bv = None
    if w_v is not None:
        wv, bv = _f32(w_v.detach()).reshape(-1), _f32(b_v.detach()).reshape(-1)
        value = torch.empty(n, dtype=torch.float32, device=h.device)
    if uniforms is not None:
        uniforms = _f32(uniforms)
        assert uniforms.shape[-1] == n and (u_row is not None or uniforms.numel() == n)
        action = torch.empty(n, dtype=torch.int64, device=h.device)
    check(lib.rlpyt_categorical_head_f32(ptr(h), ptr(w_pi), ptr(b_pi), ptr(wv), ptr(bv),
                                         ptr(uniforms), ptr(u_row), n, K, A, ptr(prob),
                                         ptr(value), ptr(action), stream()),
          "rlpyt_categorical_head_f32")
    return prob, value, action


# --------------------------------------------------------------------------------------
# gathers
# --------------------------------------------------------------------------------------
def _row_bytes(x, lead):
    n = x.element_size()
    for s in x.shape[lead:]:
        n *= s
    return n


def gather_tb(src, flat_idx, out=None):
    """``src[idx % T, idx // T]`` for a [T,B,...] tensor (ppo.py:94-100)."""
    _lib.require_gpu()
    assert src.is_contiguous() and src.dim() >= 2
    T, B = src.shape[:2]
    flat_idx = flat_idx.long().contiguous()
    M = flat_idx.numel()
    if out is None:
        out = torch.empty((M,) + tuple(src.shape[2:]), dtype=src.dtype, device=src.device)
    rb = _row_bytes(src, 2)
    with ktimer.region("gather_tb" if rb >= 4096 else "gather_tb_small", 2 * M * rb + 8 * M):
        check(lib.rlpyt_gather_tb(ptr(src), ptr(flat_idx), ptr(out), T, B, rb, M, stream()),
              "rlpyt_gather_tb")
    return out


def obs_to_nhwc_f32(obs, flat_idx=None, scale=1. / 255, out=None):
    """uint8 image batch -> float32 * scale in channels-last storage, optionally gathering
    minibatch rows ``idx -> (idx % T, idx // T)`` of a [T,B,C,H,W] batch in the same pass.

    Returns a logical ``[M, C, H, W]`` tensor whose memory is NHWC (torch.channels_last)."""
    _lib.require_gpu()
    assert obs.dtype == torch.uint8 and obs.is_contiguous() and obs.dim() >= 3
    C, H, W = obs.shape[-3:]
    if flat_idx is not None:
        assert obs.dim() == 5
        T, B = obs.shape[:2]
        flat_idx = flat_idx.long().contiguous()
        M = flat_idx.numel()
    else:
        M = obs.numel() // (C * H * W)
        T, B = 1, M
    if out is None:
        out = torch.empty((M, C, H, W), dtype=torch.float32, device=obs.device,
                          memory_format=torch.channels_last)
    with ktimer.region("obs_to_nhwc", M * C * H * W * 5 + (8 * M if flat_idx is not None else 0)):
        check(lib.rlpyt_obs_to_nhwc_f32(ptr(obs), ptr(flat_idx), ctypes.c_void_p(out.data_ptr()),
                                        T, B, C, H * W, M, float(scale), stream()),
              "rlpyt_obs_to_nhwc_f32")
    return out


def gather_rows(src, t_idx, b_idx, out=None):
    """``src[t_idx, b_idx]`` for a [T,B,...] tensor; negative t wraps once (numpy rule)."""
    _lib.require_gpu()
    assert src.is_contiguous() and src.dim() >= 2
    T, B = src.shape[:2]
    t_idx, b_idx = t_idx.long().contiguous(), b_idx.long().contiguous()
    M = t_idx.numel()
    if out is None:
        out = torch.empty((M,) + tuple(src.shape[2:]), dtype=src.dtype, device=src.device)
    check(lib.rlpyt_gather_rows(ptr(src), ptr(t_idx), ptr(b_idx), ptr(out), T, B,
                                _row_bytes(src, 2), M, stream()), "rlpyt_gather_rows")
    return out


def replay_step_fields(action, reward, done, return_, done_n, t_idx, b_idx, n_step):
    """The small fields of a single-step replay batch (``rlpyt_replay_step_fields``: one launch for
    NStepReturnBuffer.extract_batch minus the observations) -> (prev_action, prev_reward, action,
    return_, done, done_n, target_prev_action, target_prev_reward), each ``[n]``.  Ring arrays
    ``[T, B]``: int64 action, float32 reward / return_, bool done / done_n, all contiguous."""
    _lib.require_gpu()
    T, B = action.shape
    dev = action.device
    assert t_idx.device == dev and b_idx.device == dev, "replay indices must live on the ring's device"
    t_idx, b_idx = t_idx.long().contiguous(), b_idx.long().contiguous()
    n = t_idx.numel()
    assert b_idx.numel() == n
    i64 = lambda: torch.empty(n, dtype=torch.int64, device=dev)        # noqa: E731
    f32 = lambda: torch.empty(n, dtype=torch.float32, device=dev)      # noqa: E731
    bl = lambda: torch.empty(n, dtype=torch.bool, device=dev)          # noqa: E731
    pa, pr, a, r, d, dn, tpa, tpr = i64(), f32(), i64(), f32(), bl(), bl(), i64(), f32()
    check(lib.rlpyt_replay_step_fields(
        ptr(action), ptr(reward), ptr(done), ptr(return_), ptr(done_n), ptr(t_idx),
        ptr(b_idx), n, int(T), int(B), int(n_step), ptr(pa), ptr(pr), ptr(a), ptr(r),
        ptr(d), ptr(dn), ptr(tpa), ptr(tpr), stream()), "rlpyt_replay_step_fields")
    return pa, pr, a, r, d, dn, tpa, tpr


class ReplayAppender:
    """``T`` new time steps into a replay ring in one launch (``rlpyt_replay_append``:
    rlpyt/replays/n_step.py:60-83 and rlpyt/replays/frame.py:39-59).

    ``rings``: the stored fields ``[ring_T, B, ...]`` (contiguous device tensors); ``frames`` (uint8
    ``[ring_T + C - 1, B, *img]``): the unique-frame store, written from a ``[T, B, C, *img]``
    observation with its history / mirror rows.  The field table (ring pointers, row sizes) is
    built once; a call fills in the source pointers."""

    def __init__(self, rings, ring_T, B, frames=None, n_frames=1):
        _lib.require_gpu()
        self.rings, self.ring_T, self.B = list(rings), int(ring_T), int(B)
        self.table = (_lib.AppendField * max(1, len(self.rings)))()
        for k, ring in enumerate(self.rings):
            assert ring.shape[0] == self.ring_T and ring.shape[1] == self.B
            self.table[k].ring = ptr(ring).value
            self.table[k].row_bytes = _row_bytes(ring, 1)
        self._row_elems = [_row_bytes(ring, 1) // ring.element_size() for ring in self.rings]
        self.frames, self.C, self.frame_bytes = frames, 1, 0
        if frames is not None:
            self.C = int(n_frames)
            assert frames.dtype == torch.uint8 and frames.shape[1] == self.B
            assert frames.shape[0] == self.ring_T + self.C - 1
            self.frame_bytes = _row_bytes(frames, 2)
        self._frames_ptr = ptr(frames)
        self._tail = (self.frame_bytes, self.C)

    @staticmethod
    def _ready(new, like):
        """``new`` as a contiguous device tensor of the ring's dtype (what a slice assignment would
        convert it to)."""
        if not isinstance(new, torch.Tensor):
            new = torch.as_tensor(new)
        if new.device != like.device or new.dtype != like.dtype:
            new = new.to(device=like.device, dtype=like.dtype, non_blocking=True)
        return new if new.is_contiguous() else new.contiguous()

    def __call__(self, news, start, observation=None):
        keep = []                         # converted leaves stay alive until the launch is enqueued
        T = None
        table = self.table
        for k, (ring, new) in enumerate(zip(self.rings, news)):
            if not (new.__class__ is torch.Tensor and new.dtype is ring.dtype and new.is_cuda
                    and new.is_contiguous()):
                new = self._ready(new, ring)
                keep.append(new)
            T = new.shape[0] if T is None else T
            assert new.numel() == T * self._row_elems[k], \
                f"replay append: field {k}: {tuple(new.shape)} into ring {tuple(ring.shape)}"
            table[k].src = new.data_ptr()
        obs_ptr = None
        if self.frames is not None:
            if not (observation.__class__ is torch.Tensor and observation.dtype is torch.uint8
                    and observation.is_cuda and observation.is_contiguous()):
                observation = self._ready(observation, self.frames)
            T = observation.shape[0] if T is None else T
            assert tuple(observation.shape[:3]) == (T, self.B, self.C) and \
                observation.shape[3:] == self.frames.shape[2:]
            obs_ptr = ctypes.c_void_p(observation.data_ptr())
        check(lib.rlpyt_replay_append(self.table, len(self.rings), obs_ptr, self._frames_ptr,
                                      self.frame_bytes, self.C, int(T), self.B, int(start),
                                      self.ring_T, stream()), "rlpyt_replay_append")


def frames_gather(frames, done, t_idx, b_idx, n_frames, out=None):
    """NStepFrameBuffer.extract_observation (replays/non_sequence/frame.py:14-30).

    frames: uint8 [T+C-1, B, *img]; done: bool [T, B] -> uint8 [n, C, *img]."""
    _lib.require_gpu()
    assert frames.dtype == torch.uint8 and frames.is_contiguous()
    C = int(n_frames)
    T = frames.shape[0] - (C - 1)
    B = frames.shape[1]
    img = tuple(frames.shape[2:])
    HW = 1
    for s in img:
        HW *= s
    done8 = _as_done_u8(done)
    assert done8.shape[0] == T and done8.shape[1] == B
    t_idx, b_idx = t_idx.long().contiguous(), b_idx.long().contiguous()
    n = t_idx.numel()
    if out is None:
        out = torch.empty((n, C) + img, dtype=torch.uint8, device=frames.device)
    check(lib.rlpyt_frames_gather(ptr(frames), ptr(done8), ptr(t_idx), ptr(b_idx), ptr(out), n,
                                  T, B, C, HW, stream()), "rlpyt_frames_gather")
    return out


def frames_gather_pair(frames, done, t_idx, b_idx, n_frames, n_step, out=None):
    """Agent observation at ``t_idx`` and target observation at ``t_idx + n_step`` of a replay
    batch in one launch (the two ``extract_observation`` calls of
    rlpyt/replays/non_sequence/n_step.py:29-42) -> uint8 [2, n, C, *img]."""
    _lib.require_gpu()
    assert frames.dtype == torch.uint8 and frames.is_contiguous()
    C = int(n_frames)
    T = frames.shape[0] - (C - 1)
    B = frames.shape[1]
    img = tuple(frames.shape[2:])
    HW = 1
    for s in img:
        HW *= s
    done8 = _as_done_u8(done)
    assert done8.shape[0] == T and done8.shape[1] == B
    t_idx, b_idx = t_idx.long().contiguous(), b_idx.long().contiguous()
    n = t_idx.numel()
    if out is None:
        out = torch.empty((2, n, C) + img, dtype=torch.uint8, device=frames.device)
    check(lib.rlpyt_frames_gather_pair(ptr(frames), ptr(done8), ptr(t_idx), ptr(b_idx), ptr(out), n,
                                       int(n_step), T, B, C, HW, stream()),
          "rlpyt_frames_gather_pair")
    return out


def frames_gather_seq(frames, done, t_idx, b_idx, n_frames, seq_T, out=None):
    """SequenceNStepFrameBuffer.extract_observation (replays/sequence/frame.py:17-50)
    -> uint8 [seq_T, n, C, *img]."""
    _lib.require_gpu()
    assert frames.dtype == torch.uint8 and frames.is_contiguous()
    C = int(n_frames)
    T = frames.shape[0] - (C - 1)
    B = frames.shape[1]
    img = tuple(frames.shape[2:])
    HW = 1
    for s in img:
        HW *= s
    done8 = _as_done_u8(done)
    t_idx, b_idx = t_idx.long().contiguous(), b_idx.long().contiguous()
    n = t_idx.numel()
    if out is None:
        out = torch.empty((seq_T, n, C) + img, dtype=torch.uint8, device=frames.device)
    check(lib.rlpyt_frames_gather_seq(ptr(frames), ptr(done8), ptr(t_idx), ptr(b_idx), ptr(out),
                                      n, int(seq_T), T, B, C, HW, stream()),
          "rlpyt_frames_gather_seq")
    return out


def extract_sequences(src, t_idx, b_idx, seq_T, out=None):
    """rlpyt/utils/misc.py:38-56 on device -> [seq_T, n, ...]."""
    _lib.require_gpu()
    assert src.is_contiguous() and src.dim() >= 2
    T, B = src.shape[:2]
    t_idx, b_idx = t_idx.long().contiguous(), b_idx.long().contiguous()
    n = t_idx.numel()
    view = src.view(torch.uint8) if src.dtype == torch.bool else src
    if out is None:
        out = torch.empty((seq_T, n) + tuple(src.shape[2:]), dtype=view.dtype, device=src.device)
    check(lib.rlpyt_gather_sequences(ptr(view), ptr(t_idx), ptr(b_idx), ptr(out), n, int(seq_T),
                                     T, B, _row_bytes(view, 2), stream()),
          "rlpyt_gather_sequences")
    return out.view(torch.bool) if src.dtype == torch.bool else out


# --------------------------------------------------------------------------------------
# sum tree
# --------------------------------------------------------------------------------------
class DeviceSumTree:
    """Opaque handle over ``rlpyt_sumtree`` (f64 tree in HBM)."""

    def __init__(self, T, B, off_backward, off_forward, default_value=1.,
                 enable_input_priorities=False, input_priority_shift=0, device=None):
        _lib.require_gpu()
        self.device = torch.device(device if device is not None else
                                   f"cuda:{torch.cuda.current_device()}")
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            check(lib.rlpyt_sumtree_create(ctypes.byref(h), int(T), int(B), int(off_backward),
                                           int(off_forward), float(default_value),
                                           int(bool(enable_input_priorities)),
                                           int(input_priority_shift)), "rlpyt_sumtree_create")
        self._h = h
        self.T, self.B = int(T), int(B)
        self.tree_levels = lib.rlpyt_sumtree_levels(h)
        self.low_idx = lib.rlpyt_sumtree_low_idx(h)
        self.n_nodes = 2 ** self.tree_levels - 1

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            lib.rlpyt_sumtree_destroy(h)
            self._h = None

    @property
    def t(self):
        return lib.rlpyt_sumtree_cursor(self._h)

    def reset(self):
        check(lib.rlpyt_sumtree_reset(self._h, stream()), "rlpyt_sumtree_reset")

    def advance(self, T, priorities=None):
        if priorities is None:
            check(lib.rlpyt_sumtree_advance(self._h, int(T), None, 0, stream()),
                  "rlpyt_sumtree_advance")
            return
        p = torch.as_tensor(priorities, dtype=torch.float64, device=self.device).contiguous()
        if p.numel() == 1:
            kind = 1
        elif p.numel() == self.B:
            kind = 2
        else:
            assert p.numel() == T * self.B, "priorities must be scalar, [B] or [T,B]"
            kind = 3
        check(lib.rlpyt_sumtree_advance(self._h, int(T), ptr(p), kind, stream()),
              "rlpyt_sumtree_advance")

    def sample(self, uniforms):
        """uniforms: f64 device tensor [n] in [0,1) -> (T_idxs, B_idxs, priorities)."""
        u = uniforms.to(device=self.device, dtype=torch.float64).contiguous()
        n = u.numel()
        T_idxs = torch.empty(n, dtype=torch.int64, device=self.device)
        B_idxs = torch.empty(n, dtype=torch.int64, device=self.device)
        pri = torch.empty(n, dtype=torch.float64, device=self.device)
        check(lib.rlpyt_sumtree_sample(self._h, ptr(u), n, ptr(T_idxs), ptr(B_idxs), ptr(pri),
                                       stream()), "rlpyt_sumtree_sample")
        return T_idxs, B_idxs, pri

    def sample_unique(self, n, max_tries=100):
        """``sample(n, unique=True)`` of rlpyt/replays/sum_tree.py:101-128: ``n`` DISTINCT leaves,
        sorted by tree index.  The descents run on the device; the de-duplication / re-draw loop
        consumes ``np.random.rand`` exactly as the reference does (n values, then 2 x the shortfall
        per retry) and needs the indices on the host -- an off-hot-path option."""
        import numpy as np

        def find(k):
            u = torch.from_numpy(np.random.rand(int(k))).to(self.device)
            T_i, B_i, _ = self.sample(u)
            return (T_i * self.B + B_i).cpu().numpy()

        leaves = find(n)
        for _ in range(max_tries):
            leaves = np.unique(leaves)
            if len(leaves) >= n:
                break
            leaves = np.concatenate([leaves, find(2 * (n - len(leaves)))])
        if len(leaves) < n:
            raise RuntimeError("After 100 tries, unable to get unique indexes.")
        leaves = torch.from_numpy(np.ascontiguousarray(leaves[:n])).to(self.device)
        pri = torch.empty(n, dtype=torch.float64, device=self.device)
        check(lib.rlpyt_sumtree_set_sampled(self._h, ptr(leaves), int(n), ptr(pri), stream()),
              "rlpyt_sumtree_set_sampled")
        return leaves // self.B, leaves % self.B, pri

    def leaf_values(self, leaves):
        """Values the tree holds NOW for ``leaves`` (int64 ``t * B + b``, the last sampled set in
        its drawn order: the call re-declares them as that set, which leaves it unchanged)."""
        leaves = leaves.to(device=self.device, dtype=torch.int64).contiguous()
        pri = torch.empty(leaves.numel(), dtype=torch.float64, device=self.device)
        check(lib.rlpyt_sumtree_set_sampled(self._h, ptr(leaves), leaves.numel(), ptr(pri), stream()),
              "rlpyt_sumtree_set_sampled")
        return pri

    def update_batch_priorities(self, priorities):
        p = priorities.to(device=self.device, dtype=torch.float64).contiguous()
        check(lib.rlpyt_sumtree_update(self._h, ptr(p), p.numel(), stream()),
              "rlpyt_sumtree_update")

    def tree_tensor(self):
        """Copy of the whole f64 tree (for parity tests / inspection)."""
        out = torch.empty(self.n_nodes, dtype=torch.float64, device=self.device)
        check(lib.rlpyt_sumtree_copy_tree(self._h, ptr(out), stream()),
              "rlpyt_sumtree_copy_tree")
        return out
