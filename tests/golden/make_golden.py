"""Generate golden vectors from the REAL reference (astooke/rlpyt at /root/reference).

Run in the build container only (the GPU box has no /root/reference):

    python tests/golden/make_golden.py

It imports the reference's own functions/classes for the hot path,
feeds them seeded synthetic inputs and stores inputs + outputs as small ``.npz`` files
next to this script.  The oracle (``oracle/``) and the HIP path are both tested against
these files, which is what pins parity (the reference's own tests hold no numeric vectors
for this path -- SURVEY.md section 4).
"""
import os
import sys

import numpy as np
import torch

REF = os.environ.get("RLPYT_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))

from rlpyt.algos.utils import (discount_return, discount_return_n_step,  # noqa: E402
                               generalized_advantage_estimation, valid_from_done)
from rlpyt.distributions.categorical import Categorical, DistInfo  # noqa: E402
from rlpyt.replays.sum_tree import SumTree  # noqa: E402
from rlpyt.utils.misc import extract_sequences  # noqa: E402
from rlpyt.utils.tensor import select_at_indexes, valid_mean  # noqa: E402


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def scan_inputs(T, B, p_done, seed):
    g = torch.Generator().manual_seed(seed)
    reward = 0.5 * torch.randn(T, B, generator=g)
    value = torch.randn(T, B, generator=g)
    done = torch.rand(T, B, generator=g) < p_done
    bv = torch.randn(1, B, generator=g)
    return reward, value, done, bv


def gen_scans():
    out = {}
    cases = [("cfg", 128, 256, 0.01, 0.99, 0.98, 0), ("nodone", 16, 8, 0.0, 0.99, 0.95, 1),
             ("dense", 33, 7, 0.2, 0.9, 0.8, 2), ("t1", 1, 5, 0.3, 0.99, 0.97, 3),
             ("t2", 2, 3, 0.5, 0.95, 0.5, 4)]
    for name, T, B, p, gamma, lam, seed in cases:
        reward, value, done, bv = scan_inputs(T, B, p, seed)
        done_f = done.type(reward.dtype)  # as rlpyt/algos/pg/base.py:51
        adv, ret = generalized_advantage_estimation(reward, value, done_f, bv, gamma, lam)
        disc = discount_return(reward, done_f, bv, gamma)
        valid = valid_from_done(done_f)
        out.update({f"{name}_reward": reward.numpy(), f"{name}_value": value.numpy(),
                    f"{name}_done": done.numpy(), f"{name}_bv": bv.numpy(),
                    f"{name}_gamma": np.float64(gamma), f"{name}_lambda": np.float64(lam),
                    f"{name}_adv": adv.numpy(), f"{name}_ret": ret.numpy(),
                    f"{name}_disc": disc.numpy(), f"{name}_valid": valid.numpy()})
    # the survey's hand-checkable known-answer mini
    reward = torch.tensor([[1, 0], [0, 1], [2, -1], [.5, 0]])
    value = torch.tensor([[.5, .1], [.2, -.3], [1, .4], [0, .7]])
    done = torch.tensor([[0, 0], [1, 0], [0, 0], [0, 1]], dtype=torch.float)
    bv = torch.tensor([[2., 3.]])
    adv, ret = generalized_advantage_estimation(reward, value, done, bv, 0.9, 0.8)
    out.update(kat_reward=reward.numpy(), kat_value=value.numpy(), kat_done=done.numpy() > 0,
               kat_bv=bv.numpy(), kat_gamma=np.float64(0.9), kat_lambda=np.float64(0.8),
               kat_adv=adv.numpy(), kat_ret=ret.numpy(),
               kat_disc=discount_return(reward, done, bv, 0.9).numpy(),
               kat_valid=valid_from_done(done).numpy())
    # 1-D [T] input (no batch dim), as the reference allows
    r1, v1, d1, b1 = scan_inputs(9, 1, 0.2, 7)
    a1, rt1 = generalized_advantage_estimation(r1[:, 0], v1[:, 0], d1[:, 0].float(), b1[0, 0],
                                               0.99, 0.9)
    out.update(oned_reward=r1[:, 0].numpy(), oned_value=v1[:, 0].numpy(),
               oned_done=d1[:, 0].numpy(), oned_bv=b1[0, 0].numpy(), oned_adv=a1.numpy(),
               oned_ret=rt1.numpy())
    save("scans", **out)


def gen_nstep():
    out = {}
    for name, T, B, n, gamma, p, seed in [("r2d1", 44, 32, 5, 0.997, 0.05, 10),
                                          ("n3", 12, 4, 3, 0.99, 0.2, 11),
                                          ("n1", 6, 3, 1, 0.99, 0.2, 12),
                                          ("n2", 4, 2, 2, 0.9, 0.3, 13)]:
        reward, _, done, _ = scan_inputs(T, B, p, seed)
        # torch path
        rt, dn = discount_return_n_step(reward, done, n, gamma)
        rt_tr, dn_tr = discount_return_n_step(reward, done, n, gamma, do_truncated=True)
        # numpy path, as rlpyt/replays/n_step.py:81-108 calls it
        rn, dnn = discount_return_n_step(reward.numpy(), done.numpy(), n, gamma)
        assert np.array_equal(rn, rt.numpy()) and np.array_equal(dnn, dn.numpy())
        out.update({f"{name}_reward": reward.numpy(), f"{name}_done": done.numpy(),
                    f"{name}_n": np.int64(n), f"{name}_gamma": np.float64(gamma),
                    f"{name}_ret": rt.numpy(), f"{name}_done_n": dn.numpy(),
                    f"{name}_ret_trunc": rt_tr.numpy(), f"{name}_done_n_trunc": dn_tr.numpy()})
    save("nstep", **out)


def gen_normalize():
    out = {}
    for name, T, B, p, seed in [("cfg", 128, 256, 0.01, 20), ("small", 16, 8, 0.2, 21)]:
        reward, value, done, bv = scan_inputs(T, B, p, seed)
        done_f = done.float()
        adv, _ = generalized_advantage_estimation(reward, value, done_f, bv, 0.99, 0.98)
        valid = valid_from_done(done_f)
        # rlpyt/algos/pg/base.py:65-73, both branches
        a_all = adv.clone()
        a_all[:] = (a_all - a_all.mean()) / max(a_all.std(), 1e-6)
        a_val = adv.clone()
        mask = valid > 0
        a_val[:] = (a_val - a_val[mask].mean()) / max(a_val[mask].std(), 1e-6)
        out.update({f"{name}_adv": adv.numpy(), f"{name}_valid": valid.numpy(),
                    f"{name}_norm_all": a_all.numpy(), f"{name}_norm_valid": a_val.numpy()})
    save("normalize", **out)


def ppo_reference_loss(prob_new, value, prob_old, action, advantage, return_, valid, ratio_clip,
                       c_v, c_e):
    """The arithmetic of rlpyt/algos/pg/ppo.py:136-153 driven through the reference's own
    Categorical / valid_mean objects."""
    dist = Categorical(dim=prob_new.shape[-1])
    new_info, old_info = DistInfo(prob=prob_new), DistInfo(prob=prob_old)
    ratio = dist.likelihood_ratio(action, old_dist_info=old_info, new_dist_info=new_info)
    surr_1 = ratio * advantage
    surr_2 = torch.clamp(ratio, 1. - ratio_clip, 1. + ratio_clip) * advantage
    pi_loss = -valid_mean(torch.min(surr_1, surr_2), valid)
    value_loss = c_v * valid_mean(0.5 * (value - return_) ** 2, valid)
    entropy = dist.mean_entropy(new_info, valid)
    loss = pi_loss + value_loss - c_e * entropy
    perplexity = dist.mean_perplexity(new_info, valid)
    return loss, pi_loss, value_loss, entropy, perplexity


def a2c_reference_loss(prob, value, action, advantage, return_, valid, c_v, c_e):
    """rlpyt/algos/pg/a2c.py:85-101 through the reference's Categorical / valid_mean."""
    dist = Categorical(dim=prob.shape[-1])
    info = DistInfo(prob=prob)
    logli = dist.log_likelihood(action, info)
    pi_loss = -valid_mean(logli * advantage, valid)
    value_loss = c_v * valid_mean(0.5 * (value - return_) ** 2, valid)
    entropy = dist.mean_entropy(info, valid)
    loss = pi_loss + value_loss - c_e * entropy
    perplexity = dist.mean_perplexity(info, valid)
    return loss, pi_loss, value_loss, entropy, perplexity


def loss_inputs(M, A, seed, with_valid):
    g = torch.Generator().manual_seed(seed)
    prob_new = torch.softmax(torch.randn(M, A, generator=g), dim=-1)
    prob_old = torch.softmax(torch.randn(M, A, generator=g) * 0.3 +
                             torch.log(prob_new), dim=-1)
    action = torch.randint(0, A, (M,), generator=g)
    advantage = torch.randn(M, generator=g)
    return_ = torch.randn(M, generator=g)
    value = torch.randn(M, generator=g)
    valid = (torch.rand(M, generator=g) > 0.2).float() if with_valid else None
    return prob_new, value, prob_old, action, advantage, return_, valid


def gen_losses():
    out = {}
    for name, M, A, seed, with_valid, clip in [("ppo_cfg", 2048, 6, 30, False, 0.1),
                                               ("ppo_valid", 777, 6, 31, True, 0.2),
                                               ("ppo_a18", 300, 18, 32, True, 0.1)]:
        pn, v, po, a, adv, ret, valid = loss_inputs(M, A, seed, with_valid)
        pn.requires_grad_(True)
        v.requires_grad_(True)
        res = ppo_reference_loss(pn, v, po, a, adv, ret, valid, clip, 1.0, 0.01)
        res[0].backward()
        out.update({f"{name}_prob_new": pn.detach().numpy(), f"{name}_value": v.detach().numpy(),
                    f"{name}_prob_old": po.numpy(), f"{name}_action": a.numpy(),
                    f"{name}_adv": adv.numpy(), f"{name}_ret": ret.numpy(),
                    f"{name}_clip": np.float64(clip),
                    f"{name}_scalars": np.array([x.item() for x in res], dtype=np.float32),
                    f"{name}_grad_prob": pn.grad.numpy(), f"{name}_grad_value": v.grad.numpy()})
        if valid is not None:
            out[f"{name}_valid"] = valid.numpy()
    for name, M, A, seed, with_valid in [("a2c_cfg", 5 * 32, 6, 40, False),
                                         ("a2c_valid", 640, 4, 41, True)]:
        pn, v, _, a, adv, ret, valid = loss_inputs(M, A, seed, with_valid)
        pn.requires_grad_(True)
        v.requires_grad_(True)
        res = a2c_reference_loss(pn, v, a, adv, ret, valid, 0.5, 0.01)
        res[0].backward()
        out.update({f"{name}_prob": pn.detach().numpy(), f"{name}_value": v.detach().numpy(),
                    f"{name}_action": a.numpy(), f"{name}_adv": adv.numpy(),
                    f"{name}_ret": ret.numpy(),
                    f"{name}_scalars": np.array([x.item() for x in res], dtype=np.float32),
                    f"{name}_grad_prob": pn.grad.numpy(), f"{name}_grad_value": v.grad.numpy()})
        if valid is not None:
            out[f"{name}_valid"] = valid.numpy()
    # DQN.loss arithmetic (rlpyt/algos/dqn/dqn.py:231-263) -- the method needs an agent, so
    # its statements are replayed here on the reference's select_at_indexes.
    for name, M, A, seed, double, clip, pri in [("dqn", 128, 6, 50, False, 1.0, True),
                                                ("ddqn", 32, 18, 51, True, 1.0, True),
                                                ("dqn_mse", 64, 4, 52, False, None, False)]:
        g = torch.Generator().manual_seed(seed)
        qs = (2 * torch.randn(M, A, generator=g)).requires_grad_(True)
        target_qs = 2 * torch.randn(M, A, generator=g)
        next_qs = 2 * torch.randn(M, A, generator=g)
        action = torch.randint(0, A, (M,), generator=g)
        return_ = torch.randn(M, generator=g)
        done_n = torch.rand(M, generator=g) < 0.1
        isw = torch.rand(M, generator=g) if pri else None
        discount, n_step = 0.99, 3
        q = select_at_indexes(action, qs)
        with torch.no_grad():
            if double:
                target_q = select_at_indexes(torch.argmax(next_qs, dim=-1), target_qs)
            else:
                target_q = torch.max(target_qs, dim=-1).values
        disc_target_q = (discount ** n_step) * target_q
        y = return_ + (1 - done_n.float()) * disc_target_q
        delta = y - q
        losses = 0.5 * delta ** 2
        abs_delta = abs(delta)
        if clip is not None:
            b = clip * (abs_delta - clip / 2)
            losses = torch.where(abs_delta <= clip, losses, b)
        if pri:
            losses *= isw
        td = abs_delta.detach()
        if clip is not None:
            td = torch.clamp(td, 0, clip)
        loss = torch.mean(losses)
        loss.backward()
        out.update({f"{name}_qs": qs.detach().numpy(), f"{name}_target_qs": target_qs.numpy(),
                    f"{name}_next_qs": next_qs.numpy(), f"{name}_action": action.numpy(),
                    f"{name}_ret": return_.numpy(), f"{name}_done_n": done_n.numpy(),
                    f"{name}_double": np.bool_(double),
                    f"{name}_clip": np.float64(-1.0 if clip is None else clip),
                    f"{name}_disc_n": np.float64(discount ** n_step),
                    f"{name}_loss": np.float32(loss.item()), f"{name}_td": td.numpy(),
                    f"{name}_grad_qs": qs.grad.numpy()})
        if pri:
            out[f"{name}_isw"] = isw.numpy()
    save("losses", **out)


def sumtree_stream(T, B, ob, of, n_ops, n_sample, adv_T, seed, input_pri=False, shift=0,
                   record_tree=False):
    """Drive the reference SumTree with interleaved advance / sample / update and record the
    uniforms fed, the indices and priorities returned, and root sums (or whole trees)."""
    rng = np.random.RandomState(seed)
    tree = SumTree(T, B, ob, of, default_value=1.0 ** 0.6, enable_input_priorities=input_pri,
                   input_priority_shift=shift)
    rec = dict(uniforms=[], T_idxs=[], B_idxs=[], pri=[], new_pri=[], root=[], adv_pri=[],
               trees=[])
    for op in range(n_ops):
        if input_pri:
            kind = op % 3
            p = (np.abs(rng.randn(adv_T, B)) ** 0.6 if kind == 0 else
                 np.abs(rng.randn(B)) ** 0.6 if kind == 1 else
                 np.abs(rng.randn(1)) ** 0.6)
            rec["adv_pri"].append(np.broadcast_to(p, (adv_T, B)).copy())
            tree.advance(adv_T, priorities=p if kind != 2 else float(p[0]))
        else:
            tree.advance(adv_T)
        if tree.tree[0] <= 0:
            rec["root"].append(tree.tree[0])
            if record_tree:
                rec["trees"].append(tree.tree.copy())
            continue
        u = rng.rand(n_sample)
        # inject the uniforms: np.random.rand is what sample() calls (sum_tree.py:107)
        orig = np.random.rand
        np.random.rand = lambda n, _u=u: _u.copy()
        try:
            (Ti, Bi), pri = tree.sample(n_sample)
        finally:
            np.random.rand = orig
        new_p = np.abs(rng.randn(n_sample)) ** 0.6
        tree.update_batch_priorities(new_p)
        rec["uniforms"].append(u)
        rec["T_idxs"].append(Ti.astype(np.int32))
        rec["B_idxs"].append(Bi.astype(np.int32))
        rec["pri"].append(pri.copy())
        rec["new_pri"].append(new_p)
        rec["root"].append(tree.tree[0])
        if record_tree:
            rec["trees"].append(tree.tree.copy())
    res = {k: np.array(v) for k, v in rec.items() if len(v)}
    res.update(T=np.int64(T), B=np.int64(B), ob=np.int64(ob), of=np.int64(of),
               n_sample=np.int64(n_sample), adv_T=np.int64(adv_T), shift=np.int64(shift),
               input_pri=np.bool_(input_pri), final_tree_root=np.float64(tree.tree[0]),
               final_leaves_head=tree.tree[tree.low_idx:tree.low_idx + min(T * B, 4096)].copy(),
               levels=np.int64(tree.tree_levels), low_idx=np.int64(tree.low_idx))
    return res


def gen_categorical():
    """Every tensor-level method of the reference's Categorical distribution
    (rlpyt/distributions/categorical.py:17-43 + base.py:57-66: kl, mean_kl with and without a valid
    mask, entropy, perplexity, mean_entropy, mean_perplexity, log_likelihood, likelihood_ratio) on
    [T, B, A] probabilities that include exact zeros and ones (where EPS matters)."""
    g = torch.Generator().manual_seed(77)
    T, B, A = 9, 7, 6
    p_old = torch.softmax(2.5 * torch.randn(T, B, A, generator=g), -1)
    p_new = torch.softmax(2.5 * torch.randn(T, B, A, generator=g), -1)
    p_old[0, 0] = torch.tensor([1., 0., 0., 0., 0., 0.])
    p_new[0, 1] = torch.tensor([0., 0., 0.5, 0.5, 0., 0.])
    p_new[1, 0] = p_old[1, 0]
    idx = torch.randint(0, A, (T, B), generator=g)
    valid = (torch.rand(T, B, generator=g) < 0.7).float()
    dist = Categorical(dim=A)
    o, n = DistInfo(prob=p_old), DistInfo(prob=p_new)
    save("categorical", p_old=p_old.numpy(), p_new=p_new.numpy(), idx=idx.numpy(), valid=valid.numpy(),
         kl=dist.kl(o, n).numpy(), mean_kl=np.float32(dist.mean_kl(o, n).item()),
         mean_kl_valid=np.float32(dist.mean_kl(o, n, valid).item()),
         entropy=dist.entropy(n).numpy(), perplexity=dist.perplexity(n).numpy(),
         mean_entropy_valid=np.float32(dist.mean_entropy(n, valid).item()),
         mean_perplexity_valid=np.float32(dist.mean_perplexity(n, valid).item()),
         log_likelihood=dist.log_likelihood(idx, n).numpy(),
         likelihood_ratio=dist.likelihood_ratio(idx, o, n).numpy(),
         onehot=dist.to_onehot(idx).numpy())     # (the reference's DiscreteMixin.from_onehot raises: keyword typo, discrete.py:25)


def gen_sumtree():
    out = {}
    # survey known-answer mini: SumTree(8,2,1,1), advance(4), seed-0 samples (section 8c)
    t = SumTree(8, 2, 1, 1, default_value=1)
    t.advance(4)
    np.random.seed(0)
    (Ti, Bi), p = t.sample(5)
    t.update_batch_priorities(np.array([0.5, 2, 3, 0.1, 4]))
    root_after = t.tree[0]
    (Ti2, Bi2), p2 = t.sample(5)
    np.random.seed(0)
    u_first = np.random.rand(5)
    u_second = np.random.rand(5)
    out.update(kat_T1=Ti, kat_B1=Bi, kat_p1=p, kat_root=np.float64(root_after), kat_T2=Ti2,
               kat_B2=Bi2, kat_p2=p2, kat_u1=u_first, kat_u2=u_second, kat_tree=t.tree.copy())
    for name, kw in [
        ("small", dict(T=16, B=3, ob=2, of=3, n_ops=60, n_sample=9, adv_T=3, seed=1,
                       record_tree=True)),
        ("wrap", dict(T=10, B=2, ob=1, of=1, n_ops=40, n_sample=6, adv_T=4, seed=2,
                      record_tree=True)),
        ("inpri", dict(T=12, B=4, ob=3, of=1, n_ops=50, n_sample=8, adv_T=2, seed=3,
                       input_pri=True, shift=1, record_tree=True)),
        ("dqn1m", dict(T=62500, B=16, ob=1, of=3, n_ops=150, n_sample=128, adv_T=2, seed=4)),
    ]:
        res = sumtree_stream(**kw)
        out.update({f"{name}_{k}": v for k, v in res.items()})
    save("sumtree", **out)


def gen_sumtree_unique():
    """The reference ``SumTree.sample(n, unique=True)`` (rlpyt/replays/sum_tree.py:109-128) on a
    skewed tree: several draws (with their re-draw loops consuming np.random), each followed by
    ``update_batch_priorities``; a near-full draw that needs many retries."""
    out = {}
    T, B = 24, 3
    t = SumTree(T, B, 2, 1, default_value=1)
    rng = np.random.RandomState(7)
    t.advance(10, priorities=rng.rand(10, B) ** 4 + 1e-3) if t.input_priorities is not None \
        else t.advance(10)
    np.random.seed(11)
    for k, n in enumerate([6, 12, 5, 20]):
        (Ti, Bi), p = t.sample(n, unique=True)
        new = rng.rand(n) ** 3 + 1e-3
        t.update_batch_priorities(new)
        out.update({f"T{k}": Ti, f"B{k}": Bi, f"p{k}": p, f"new{k}": new,
                    f"root{k}": np.float64(t.tree[0]), f"tree{k}": t.tree.copy()})
        if k == 1:
            t.advance(5)
    out["after"] = np.random.rand(3)         # where the host RNG stream stands afterwards
    out["geom"] = np.array([T, B, 2, 1, 10, 5])
    save("sumtree_unique", **out)


def gen_frames():
    from rlpyt.replays.non_sequence.frame import NStepFrameBuffer
    from rlpyt.replays.sequence.frame import SequenceNStepFrameBuffer
    out = {}
    rng = np.random.RandomState(5)
    # non-sequence frame gather: call the reference method on a bare object carrying only
    # the attributes it reads (samples_frames, samples.done, n_frames).
    for name, T, B, C, H, W, n, p in [("small", 20, 3, 4, 5, 4, 17, 0.15),
                                      ("c2", 9, 2, 2, 3, 3, 8, 0.3)]:
        frames = rng.randint(0, 256, size=(T + C - 1, B, H, W)).astype(np.uint8)
        frames[:C - 1] = frames[-(C - 1):]  # wrapped state: head rows mirror the tail
        done = rng.rand(T, B) < p
        T_idxs = rng.randint(0, T, size=n)
        T_idxs[:3] = [0, 1, T - 1]  # exercise the negative-index wrap of done[T_idxs - f]
        B_idxs = rng.randint(0, B, size=n)

        class Bare:
            pass
        obj = Bare()
        obj.samples_frames, obj.n_frames = frames, C
        obj.samples = Bare()
        obj.samples.done = done
        obs = NStepFrameBuffer.extract_observation(obj, T_idxs, B_idxs)
        out.update({f"{name}_frames": frames, f"{name}_done": done, f"{name}_T_idxs": T_idxs,
                    f"{name}_B_idxs": B_idxs, f"{name}_C": np.int64(C), f"{name}_obs": obs})
        seq_T = 7 if T > 10 else 4
        obj.T = T
        sT = rng.randint(0, T, size=n)
        sT[:3] = [0, T - 2, T - seq_T]
        seq = SequenceNStepFrameBuffer.extract_observation(obj, sT, B_idxs, seq_T)
        out.update({f"{name}_seq_T_idxs": sT, f"{name}_seq_T": np.int64(seq_T),
                    f"{name}_seq_obs": seq})
    # extract_sequences incl. wrap-at-end and the negative-start branch
    arr = rng.randn(11, 3, 2).astype(np.float32)
    Ti = np.array([0, 5, 9, -1, 10, -2])
    Bi = np.array([0, 1, 2, 0, 1, 2])
    out.update(es_arr=arr, es_T_idxs=Ti, es_B_idxs=Bi, es_seq_T=np.int64(4),
               es_out=extract_sequences(arr, Ti, Bi, 4))
    save("frames", **out)


def gen_replay():
    """End-to-end stream through the reference's PrioritizedReplayFrameBuffer and
    UniformReplayFrameBuffer (append with wraps, n-step returns, sample, priority update)."""
    from rlpyt.replays.non_sequence.frame import (PrioritizedReplayFrameBuffer,
                                                  UniformReplayFrameBuffer)
    from rlpyt.utils.collections import namedarraytuple
    from rlpyt.utils.logging import logger as ref_logger
    ref_logger.log = lambda *a, **k: None
    SamplesToBuffer = namedarraytuple("SamplesToBuffer",
                                      ["observation", "action", "reward", "done"])
    out = {}
    for name, pri, n_step, alpha in [("pri_n3", True, 3, 1.0), ("pri_n1", True, 1, 0.6),
                                     ("uni_n2", False, 2, None)]:
        rng = np.random.RandomState(77)
        B, C, H, W, Tring, Tapp, n_app, nb = 4, 4, 6, 5, 40, 5, 26, 12
        example = SamplesToBuffer(observation=np.zeros((C, H, W), np.uint8),
                                  action=np.int64(0), reward=np.float32(0), done=False)
        kw = dict(example=example, size=Tring * B, B=B, discount=0.99, n_step_return=n_step)
        if pri:
            buf = PrioritizedReplayFrameBuffer(alpha=alpha, beta=0.5, default_priority=1.,
                                               **kw)
        else:
            buf = UniformReplayFrameBuffer(**kw)
        # a consistent frame-stacked observation stream per env
        frames = rng.randint(0, 256, size=(n_app * Tapp + C, B, H, W)).astype(np.uint8)
        rec = dict(obs=[], action=[], reward=[], done=[])
        samp = dict(app_idx=[], agent_obs=[], target_obs=[], prev_action=[], prev_reward=[],
                    action=[], return_=[], done=[], done_n=[], tgt_prev_action=[],
                    is_weights=[], new_pri=[], seeds=[])
        t_abs = 0
        for k in range(n_app):
            obs = np.stack([np.stack([frames[t_abs + i + c] for c in range(C)], axis=1)
                            for i in range(Tapp)])  # [Tapp, B, C, H, W]
            action = rng.randint(0, 6, size=(Tapp, B)).astype(np.int64)
            reward = rng.randn(Tapp, B).astype(np.float32)
            done = rng.rand(Tapp, B) < 0.1
            t_abs += Tapp
            buf.append_samples(SamplesToBuffer(obs, action, reward, done))
            for key, v in zip(("obs", "action", "reward", "done"), (obs, action, reward, done)):
                rec[key].append(v)
            if k >= 2:
                seed = 1000 + k
                np.random.seed(seed)
                batch = buf.sample_batch(nb)
                samp["seeds"].append(seed)
                samp["app_idx"].append(k)
                samp["agent_obs"].append(batch.agent_inputs.observation.numpy().copy())
                samp["target_obs"].append(batch.target_inputs.observation.numpy().copy())
                samp["prev_action"].append(batch.agent_inputs.prev_action.numpy().copy())
                samp["prev_reward"].append(batch.agent_inputs.prev_reward.numpy().copy())
                samp["tgt_prev_action"].append(batch.target_inputs.prev_action.numpy().copy())
                samp["action"].append(batch.action.numpy().copy())
                samp["return_"].append(batch.return_.numpy().copy())
                samp["done"].append(batch.done.numpy().copy())
                samp["done_n"].append(batch.done_n.numpy().copy())
                if pri:
                    samp["is_weights"].append(batch.is_weights.numpy().copy())
                    new_p = torch.from_numpy(np.abs(rng.randn(nb)).astype(np.float32))
                    buf.update_batch_priorities(new_p)
                    samp["new_pri"].append(new_p.numpy())
        out.update({f"{name}_{k}": np.array(v) for k, v in rec.items()})
        out.update({f"{name}_s_{k}": np.array(v) for k, v in samp.items() if len(v)})
        out.update({f"{name}_meta": np.array([B, C, H, W, Tring, Tapp, n_app, nb, n_step]),
                    f"{name}_alpha": np.float64(alpha if alpha else 0.)})
        if pri:
            out[f"{name}_final_root"] = np.float64(buf.priority_tree.tree[0])
    save("replay", **out)


def gen_seq_replay():
    """Stream through the reference's PrioritizedSequenceReplayFrameBuffer (R2D1 geometry in
    miniature: periodic RNN-state storage, n-step returns, input priorities with shift)."""
    from rlpyt.replays.sequence.frame import (PrioritizedSequenceReplayFrameBuffer,
                                              UniformSequenceReplayFrameBuffer)
    from rlpyt.utils.collections import namedarraytuple
    from rlpyt.utils.logging import logger as ref_logger
    ref_logger.log = lambda *a, **k: None
    RnnState = namedarraytuple("RnnState", ["h", "c"])
    S2B = namedarraytuple("SamplesToBufferRnn",
                          ["observation", "action", "reward", "done", "prev_rnn_state"])
    Pri = namedarraytuple("PrioritiesSamplesToBuffer", ["priorities", "samples"])
    out = {}
    for name, pri in [("pseq", True), ("useq", False)]:
        rng = np.random.RandomState(91)
        B, C, H, W, Tring, Tapp, n_app, nb, n_step, rsi, bT = 3, 4, 5, 4, 48, 4, 40, 5, 2, 4, 8
        example = S2B(observation=np.zeros((C, H, W), np.uint8), action=np.int64(0),
                      reward=np.float32(0), done=False,
                      prev_rnn_state=RnnState(np.zeros((1, 3), np.float32),
                                              np.zeros((1, 3), np.float32)))
        kw = dict(example=example, size=Tring * B, B=B, discount=0.99, n_step_return=n_step,
                  rnn_state_interval=rsi, batch_T=bT)
        if pri:
            buf = PrioritizedSequenceReplayFrameBuffer(alpha=1.0, beta=0.5, default_priority=1.,
                                                       input_priorities=True,
                                                       input_priority_shift=1, **kw)
        else:
            buf = UniformSequenceReplayFrameBuffer(**kw)
        frames = rng.randint(0, 256, size=(n_app * Tapp + C, B, H, W)).astype(np.uint8)
        rec = dict(obs=[], action=[], reward=[], done=[], h=[], c=[], in_pri=[])
        samp = dict(seeds=[], all_obs=[], all_action=[], all_reward=[], return_=[], done=[],
                    done_n=[], h=[], c=[], is_weights=[], new_pri=[])
        t_abs = 0
        for k in range(n_app):
            obs = np.stack([np.stack([frames[t_abs + i + c] for c in range(C)], axis=1)
                            for i in range(Tapp)])
            action = rng.randint(0, 6, size=(Tapp, B)).astype(np.int64)
            reward = rng.randn(Tapp, B).astype(np.float32)
            done = rng.rand(Tapp, B) < 0.06
            h = rng.randn(Tapp, B, 1, 3).astype(np.float32)
            c = rng.randn(Tapp, B, 1, 3).astype(np.float32)
            t_abs += Tapp
            smp = S2B(obs, action, reward, done, RnnState(h, c))
            in_pri = np.abs(rng.randn(B)) + 0.1
            buf.append_samples(Pri(priorities=in_pri, samples=smp) if pri else smp)
            for key, v in zip(("obs", "action", "reward", "done", "h", "c", "in_pri"),
                              (obs, action, reward, done, h, c, in_pri)):
                rec[key].append(v)
            if k >= 5:
                seed = 2000 + k
                np.random.seed(seed)
                batch = buf.sample_batch(nb)
                samp["seeds"].append(seed)
                samp["all_obs"].append(batch.all_observation.numpy().copy())
                samp["all_action"].append(batch.all_action.numpy().copy())
                samp["all_reward"].append(batch.all_reward.numpy().copy())
                samp["return_"].append(batch.return_.numpy().copy())
                samp["done"].append(batch.done.numpy().copy())
                samp["done_n"].append(batch.done_n.numpy().copy())
                samp["h"].append(batch.init_rnn_state.h.numpy().copy())
                samp["c"].append(batch.init_rnn_state.c.numpy().copy())
                if pri:
                    samp["is_weights"].append(batch.is_weights.numpy().copy())
                    new_p = torch.from_numpy((np.abs(rng.randn(nb)) + 0.05).astype(np.float32))
                    buf.update_batch_priorities(new_p)
                    samp["new_pri"].append(new_p.numpy())
        out.update({f"{name}_{k}": np.array(v) for k, v in rec.items()})
        out.update({f"{name}_s_{k}": np.array(v) for k, v in samp.items() if len(v)})
        out[f"{name}_meta"] = np.array([B, C, H, W, Tring, Tapp, n_app, nb, n_step, rsi, bT])
        if pri:
            out[f"{name}_final_root"] = np.float64(buf.priority_tree.tree[0])
            out[f"{name}_tree_geom"] = np.array([buf.priority_tree.T,
                                                 buf.priority_tree.off_backward,
                                                 buf.priority_tree.off_forward])
    save("seq_replay", **out)


def gen_r2d1_rms():
    """R2D1 post-network loss arithmetic (rlpyt/algos/dqn/r2d1.py:298-345) driven through
    the reference R2D1 object's own value_scale / inv_value_scale, and the reference
    RunningMeanStdModel over three updates."""
    from rlpyt.algos.dqn.r2d1 import R2D1
    from rlpyt.models.running_mean_std import RunningMeanStdModel
    out = {}
    for name, T, B, A, double, clip, pri in [("r2d1", 80, 64, 6, True, None, True),
                                             ("r2d1_huber", 20, 5, 4, False, 1.0, False)]:
        algo = R2D1(delta_clip=clip, double_dqn=double, prioritized_replay=pri)
        g = torch.Generator().manual_seed(60 + T)
        qs = (3 * torch.randn(T, B, A, generator=g)).requires_grad_(True)
        target_qs = 3 * torch.randn(T, B, A, generator=g)
        next_qs = 3 * torch.randn(T, B, A, generator=g)
        action = torch.randint(0, A, (T, B), generator=g)
        return_ = torch.randn(T, B, generator=g)
        done_n = torch.rand(T, B, generator=g) < 0.05
        done = torch.rand(T, B, generator=g) < 0.02
        isw = torch.rand(B, generator=g) + 0.1
        q = select_at_indexes(action, qs)
        with torch.no_grad():
            if double:
                target_q = select_at_indexes(torch.argmax(next_qs, dim=-1), target_qs)
            else:
                target_q = torch.max(target_qs, dim=-1).values
        disc = algo.discount ** algo.n_step_return
        y = algo.value_scale(return_ + (1 - done_n.float()) * disc *
                             algo.inv_value_scale(target_q))
        delta = y - q
        losses = 0.5 * delta ** 2
        abs_delta = abs(delta)
        if clip is not None:
            b = clip * (abs_delta - clip / 2)
            losses = torch.where(abs_delta <= clip, losses, b)
        if pri:
            losses *= isw.unsqueeze(0)
        valid = valid_from_done(done)
        loss = valid_mean(losses, valid)
        td = abs_delta.detach()
        if clip is not None:
            td = torch.clamp(td, 0, clip)
        vtd = td * valid
        max_d = torch.max(vtd, dim=0).values
        mean_d = valid_mean(td, valid, dim=0)
        priorities = algo.pri_eta * max_d + (1 - algo.pri_eta) * mean_d
        loss.backward()
        out.update({f"{name}_qs": qs.detach().numpy(), f"{name}_target_qs": target_qs.numpy(),
                    f"{name}_next_qs": next_qs.numpy(), f"{name}_action": action.numpy(),
                    f"{name}_ret": return_.numpy(), f"{name}_done_n": done_n.numpy(),
                    f"{name}_valid": valid.numpy(), f"{name}_isw": isw.numpy(),
                    f"{name}_double": np.bool_(double), f"{name}_pri": np.bool_(pri),
                    f"{name}_clip": np.float64(-1. if clip is None else clip),
                    f"{name}_disc_n": np.float64(disc), f"{name}_eps": np.float64(
                        algo.value_scale_eps), f"{name}_eta": np.float64(algo.pri_eta),
                    f"{name}_loss": np.float32(loss.item()), f"{name}_vtd": vtd.numpy(),
                    f"{name}_priorities": priorities.numpy(),
                    f"{name}_grad_qs": qs.grad.numpy()})
    g = torch.Generator().manual_seed(70)
    rms = RunningMeanStdModel((17,))
    xs = [torch.randn(16, 8, 17, generator=g) * (1 + k) + k for k in range(3)]
    for k, x in enumerate(xs):
        rms.update(x)
        out.update({f"rms_x{k}": x.numpy(), f"rms_mean{k}": rms.mean.numpy().copy(),
                    f"rms_var{k}": rms.var.numpy().copy(),
                    f"rms_count{k}": np.float32(rms.count.item())})
    obs_var = torch.clamp(rms.var, min=1e-6)
    out["rms_norm"] = torch.clamp((xs[0] - rms.mean) / obs_var.sqrt(), -10, 10).numpy()
    save("r2d1_rms", **out)


def gen_catdqn():
    """CategoricalDQN.loss (rlpyt/algos/dqn/cat_dqn.py:34-93) -- the reference METHOD itself,
    run on a stub agent that hands back preset network outputs."""
    from collections import namedtuple
    from rlpyt.algos.dqn.cat_dqn import CategoricalDQN
    Samples = namedtuple("Samples", ["agent_inputs", "target_inputs", "action", "return_",
                                     "done", "done_n", "is_weights"])

    class StubAgent:
        def __init__(self, n_atoms, ps, target_ps, next_ps, agent_inputs):
            self.n_atoms, self.ps, self.target_ps, self.next_ps = n_atoms, ps, target_ps, next_ps
            self._agent_inputs = agent_inputs

        def __call__(self, *inputs):
            return self.ps if inputs[0] is self._agent_inputs[0] else self.next_ps

        def target(self, *inputs):
            return self.target_ps

    out = {}
    cases = [  # name, M, A, P, V_min, V_max, double, pri, mid_batch_reset, logit scale, n_step
        ("cat", 128, 6, 51, -10, 10, False, True, True, 1.0, 1),
        ("cat_double", 32, 18, 51, -10, 10, True, False, True, 2.0, 3),
        ("cat_valid", 64, 4, 21, -5, 20, False, True, False, 1.0, 3),
        ("cat_peaky", 48, 3, 64, -1, 1, True, True, True, 12.0, 1),  # p < 1e-6: clamp active
    ]
    for name, M, A, P, v_min, v_max, double, pri, mbr, scale, n_step in cases:
        g = torch.Generator().manual_seed(80 + M)
        ps = torch.softmax(scale * torch.randn(M, A, P, generator=g), dim=-1).requires_grad_(True)
        target_ps = torch.softmax(scale * torch.randn(M, A, P, generator=g), dim=-1)
        next_ps = torch.softmax(scale * torch.randn(M, A, P, generator=g), dim=-1)
        action = torch.randint(0, A, (M,), generator=g)
        return_ = (v_max - v_min) * 0.2 * torch.randn(M, generator=g)
        done_n = torch.rand(M, generator=g) < 0.1
        done = torch.rand(M, generator=g) < 0.03
        isw = torch.rand(M, generator=g) + 0.1
        algo = CategoricalDQN(V_min=v_min, V_max=v_max, discount=0.99, n_step_return=n_step,
                              double_dqn=double, prioritized_replay=pri, batch_size=M)
        a_in, t_in = (torch.zeros(1),), (torch.ones(1),)
        algo.agent = StubAgent(P, ps, target_ps, next_ps if double else None, a_in)
        algo.mid_batch_reset = mbr
        loss, kl = algo.loss(Samples(a_in, t_in, action, return_, done, done_n,
                                     isw.clone() if pri else None))
        loss.backward()
        out.update({f"{name}_ps": ps.detach().numpy(), f"{name}_target_ps": target_ps.numpy(),
                    f"{name}_next_ps": next_ps.numpy(), f"{name}_action": action.numpy(),
                    f"{name}_ret": return_.numpy(), f"{name}_done_n": done_n.numpy(),
                    f"{name}_done": done.numpy(), f"{name}_isw": isw.numpy(),
                    f"{name}_double": np.bool_(double), f"{name}_pri": np.bool_(pri),
                    f"{name}_mbr": np.bool_(mbr), f"{name}_vmin": np.float64(v_min),
                    f"{name}_vmax": np.float64(v_max), f"{name}_n_step": np.int64(n_step),
                    f"{name}_discount": np.float64(0.99),
                    f"{name}_loss": np.float32(loss.item()), f"{name}_kl": kl.detach().numpy(),
                    f"{name}_grad_ps": ps.grad.numpy()})
    save("catdqn", **out)


from model_cases import MODEL_CASES, MODEL_SEED, model_inputs, scalarize  # noqa: E402


def gen_models():
    """Full-size hot-path models built by the REFERENCE classes under a fixed seed: parameter
    names / shapes / per-tensor checksums (the new framework's models must initialise to the
    same values under the same seed), forward outputs and per-parameter gradient norms."""
    import importlib
    out = {}
    for name, path, _mine, kwargs, recurrent in MODEL_CASES:
        mod, cls = path.rsplit(".", 1)
        Model = getattr(importlib.import_module(mod), cls)
        torch.manual_seed(MODEL_SEED)
        model = Model(image_shape=(4, 104, 80), output_size=6, **kwargs)
        sd = model.state_dict()
        out[f"{name}_names"] = np.array(list(sd.keys()))
        out[f"{name}_shapes"] = np.array([str(tuple(v.shape)) for v in sd.values()])
        out[f"{name}_sums"] = np.array([v.double().sum().item() for v in sd.values()])
        out[f"{name}_abs_sums"] = np.array([v.double().abs().sum().item() for v in sd.values()])
        inputs = model_inputs(recurrent)
        res = model(*inputs)
        res = res if isinstance(res, tuple) else (res,)
        scalarize(res).backward()
        k = 0
        for o in res:
            for leaf in ([o] if isinstance(o, torch.Tensor) else list(o)):
                out[f"{name}_out{k}"] = leaf.detach().numpy()
                k += 1
        out[f"{name}_grad_norms"] = np.array([p.grad.double().norm().item()
                                              for p in model.parameters()])
        small = [(n, p) for n, p in model.named_parameters() if p.numel() <= 1024]
        for n, p in small:
            out[f"{name}_grad__{n}"] = p.grad.numpy()
    save("models", **out)


def gen_sampler():
    """The reference GpuSampler itself (GpuResetCollector / GpuWaitResetCollector worker
    processes + ActionServer.serve_actions, rlpyt/samplers/parallel/gpu/*.py) on CPU, stepping
    this repo's synthetic env under a deterministic policy: four consecutive batches, every field
    of the samples buffer, and the completed trajectory infos."""
    import sampler_cases as C
    from rlpyt.agents.base import AgentStep, BaseAgent
    from rlpyt.samplers.parallel.gpu.collectors import GpuResetCollector, GpuWaitResetCollector
    from rlpyt.samplers.parallel.gpu.sampler import GpuSampler as RefGpuSampler
    from rlpyt.utils.collections import namedarraytuple
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from rlpyt_amd.envs.synthetic import SyntheticPong
    AgentInfo = C.bind_agent_info(namedarraytuple)

    class DetAgent(BaseAgent):
        def __init__(self):
            super().__init__(ModelCls=None)

        def initialize(self, env_spaces, share_memory=False, global_B=1, env_ranks=None):
            self.n = env_spaces.action.n
            self.env_spaces, self.share_memory = env_spaces, share_memory

        def to_device(self, cuda_idx=None):
            pass

        def step(self, observation, prev_action, prev_reward):
            a, v = C.det_policy(observation, prev_action, prev_reward, self.n)
            return AgentStep(action=a, agent_info=AgentInfo(value=v))

        def value(self, observation, prev_action, prev_reward):
            return C.det_policy(observation, prev_action, prev_reward, self.n)[1] + 1

        def sample_mode(self, itr):
            pass

        train_mode = eval_mode = sample_mode

        def sync_shared_memory(self):
            pass

    out = {}
    for name, mode, T, n_batches in C.CASES:
        Coll = GpuResetCollector if mode == "reset" else GpuWaitResetCollector
        s = RefGpuSampler(EnvCls=SyntheticPong, env_kwargs=C.ENV_KWARGS, batch_T=T, batch_B=C.B,
                          CollectorCls=Coll, max_decorrelation_steps=0)
        s.initialize(DetAgent(), affinity=dict(workers_cpus=list(range(C.N_WORKERS)), cuda_idx=None,
                                               set_affinity=False),
                     seed=C.SEED, bootstrap_value=True, traj_info_kwargs=dict(discount=0.9))
        for itr in range(n_batches):
            smp, infos = s.obtain_samples(itr)
            k = f"{name}{itr}_"
            out.update({
                k + "obs_crc": C.obs_crc(smp.env.observation.numpy()),
                k + "reward": smp.env.reward.numpy().copy(),
                k + "prev_reward": smp.env.prev_reward.numpy().copy(),
                k + "done": smp.env.done.numpy().copy(),
                k + "game_score": smp.env.env_info.game_score.numpy().copy(),
                k + "traj_done": smp.env.env_info.traj_done.numpy().copy(),
                k + "action": smp.agent.action.numpy().copy(),
                k + "prev_action": smp.agent.prev_action.numpy().copy(),
                k + "value": smp.agent.agent_info.value.numpy().copy(),
                k + "bootstrap_value": smp.agent.bootstrap_value.numpy().copy(),
                k + "traj_len_ret": np.array(sorted((ti["Length"], ti["Return"]) for ti in infos),
                                             dtype=np.float64).reshape(-1, 2),
                # every logged TrajInfo field (samplers/collections.py:30-56), discount set the
                # way the runner sets it through traj_info_kwargs
                k + "traj_fields": np.array(sorted(
                    (ti["Length"], ti["Return"], ti["NonzeroRewards"], ti["DiscountedReturn"])
                    for ti in infos), dtype=np.float64).reshape(-1, 4)})
        s.shutdown()
    # offline evaluation (parallel/base.py:115-145, gpu/action_server.py:76-120,
    # gpu/collectors.py:129-161): every eval env runs eval_max_steps // eval_n_envs steps, all
    # completed trajectories are returned (no trajectory cap: that stop is time-based there)
    s = RefGpuSampler(EnvCls=SyntheticPong, env_kwargs=C.ENV_KWARGS, batch_T=5, batch_B=C.B,
                      max_decorrelation_steps=0, eval_n_envs=C.EVAL_N_ENVS,
                      eval_env_kwargs=C.EVAL_ENV_KWARGS, eval_max_steps=C.EVAL_MAX_STEPS)
    s.initialize(DetAgent(), affinity=dict(workers_cpus=list(range(C.N_WORKERS)), cuda_idx=None,
                                           set_affinity=False),
                 seed=C.SEED, bootstrap_value=True)
    for k in range(2):      # evaluate, train a batch, evaluate again (fresh eval envs each time)
        infos = s.evaluate_agent(k)
        out[f"eval{k}_len_ret"] = np.array(sorted((ti["Length"], ti["Return"]) for ti in infos),
                                           dtype=np.float64).reshape(-1, 2)
        smp, _ = s.obtain_samples(k)
        out[f"eval{k}_next_batch_action"] = smp.agent.action.numpy().copy()
    s.shutdown()
    save("sampler", **out)


def gen_sampler_alt():
    """The reference's ALTERNATING GPU samplers (rlpyt/samplers/parallel/gpu/alternating_sampler.py with
    AlternatingActionServer / NoOverlapAlternatingActionServer, action_server.py:123-363) on CPU over
    this repo's synthetic env under the deterministic policy of ``gen_sampler``: two worker processes
    = one per half, reset and wait-reset collectors; every field of every batch."""
    import sampler_cases as C
    from rlpyt.agents.base import AgentStep, BaseAgent
    from rlpyt.samplers.parallel.gpu.alternating_sampler import (AlternatingSampler,
                                                                 NoOverlapAlternatingSampler)
    from rlpyt.samplers.parallel.gpu.collectors import GpuResetCollector, GpuWaitResetCollector
    from rlpyt.utils.collections import namedarraytuple
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from rlpyt_amd.envs.synthetic import SyntheticPong
    AgentInfo = C.bind_agent_info(namedarraytuple)

    class DetAgent(BaseAgent):
        def __init__(self):
            super().__init__(ModelCls=None)

        def initialize(self, env_spaces, share_memory=False, global_B=1, env_ranks=None):
            self.n = env_spaces.action.n
            self.env_spaces, self.share_memory = env_spaces, share_memory

        def to_device(self, cuda_idx=None):
            pass

        def step(self, observation, prev_action, prev_reward):
            a, v = C.det_policy(observation, prev_action, prev_reward, self.n)
            return AgentStep(action=a, agent_info=AgentInfo(value=v))

        def value(self, observation, prev_action, prev_reward):
            return C.det_policy(observation, prev_action, prev_reward, self.n)[1] + 1

        def sample_mode(self, itr):
            pass

        train_mode = eval_mode = sample_mode

        def sync_shared_memory(self):
            pass

    out = {}
    for tag, Cls in (("alt", AlternatingSampler), ("noalt", NoOverlapAlternatingSampler)):
        for name, mode, T, n_batches in C.CASES:
            Coll = GpuResetCollector if mode == "reset" else GpuWaitResetCollector
            s = Cls(EnvCls=SyntheticPong, env_kwargs=C.ENV_KWARGS, batch_T=T, batch_B=C.B,
                    CollectorCls=Coll, max_decorrelation_steps=0)
            s.initialize(DetAgent(), affinity=dict(workers_cpus=list(range(C.N_WORKERS)), cuda_idx=None,
                                                   set_affinity=False, alternating=True),
                         seed=C.SEED, bootstrap_value=True, traj_info_kwargs=dict(discount=0.9))
            for itr in range(n_batches):
                smp, infos = s.obtain_samples(itr)
                k = f"{tag}_{name}{itr}_"
                out.update({
                    k + "obs_crc": C.obs_crc(smp.env.observation.numpy()),
                    k + "reward": smp.env.reward.numpy().copy(),
                    k + "prev_reward": smp.env.prev_reward.numpy().copy(),
                    k + "done": smp.env.done.numpy().copy(),
                    k + "game_score": smp.env.env_info.game_score.numpy().copy(),
                    k + "traj_done": smp.env.env_info.traj_done.numpy().copy(),
                    k + "action": smp.agent.action.numpy().copy(),
                    k + "prev_action": smp.agent.prev_action.numpy().copy(),
                    k + "value": smp.agent.agent_info.value.numpy().copy(),
                    k + "bootstrap_value": smp.agent.bootstrap_value.numpy().copy(),
                    k + "traj_fields": np.array(sorted(
                        (ti["Length"], ti["Return"], ti["NonzeroRewards"], ti["DiscountedReturn"])
                        for ti in infos), dtype=np.float64).reshape(-1, 4)})
            s.shutdown()
    save("sampler_alt", **out)


def gen_sampler_ff():
    """The reference GpuSampler (GpuResetCollector workers + ActionServer.serve_actions,
    rlpyt/samplers/parallel/gpu/*.py) driving the reference's OWN AtariFfAgent
    (CategoricalPgAgent + AtariFfModel, rlpyt/agents/pg/atari.py, agents/pg/categorical.py:20-51)
    on CPU over this repo's synthetic env, with a policy head sharpened until every sampled action
    is deterministic (sampler_cases.ff_sharpen).  Recorded: every field of 11 consecutive batches,
    incl. dist_info.prob, value and bootstrap_value -- what tests/test_sampler_gpu_parity.py holds
    the fused device chain of the MI355X sampler to."""
    import sampler_cases as C
    from rlpyt.agents.pg.atari import AtariFfAgent
    from rlpyt.models.pg.atari_ff_model import AtariFfModel
    from rlpyt.samplers.parallel.gpu.collectors import GpuResetCollector
    from rlpyt.samplers.parallel.gpu.sampler import GpuSampler as RefGpuSampler
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from rlpyt_amd.envs.synthetic import SyntheticPong
    s = RefGpuSampler(EnvCls=SyntheticPong, env_kwargs=C.FF_ENV_KWARGS, batch_T=C.FF_T, batch_B=C.B,
                      CollectorCls=GpuResetCollector, max_decorrelation_steps=0)
    agent = AtariFfAgent()
    s.initialize(agent, affinity=dict(workers_cpus=list(range(C.N_WORKERS)), cuda_idx=None,
                                      set_affinity=False),
                 seed=C.SEED, bootstrap_value=True, traj_info_kwargs=dict(discount=0.9))
    torch.manual_seed(C.FF_INIT_SEED)
    fresh = AtariFfModel(image_shape=(4, 104, 80), output_size=6)
    agent.model.load_state_dict(fresh.state_dict())
    C.ff_sharpen(agent.model)
    out = {"param_crc": C.param_checksums(list(agent.model.parameters()))}
    try:
        min_gap = _record_ff_batches(s, agent, C, out)
    finally:
        s.shutdown()
    acts = np.concatenate([out[f"ff{i}_action"].reshape(-1) for i in range(C.FF_BATCHES)])
    print("sampler_ff: min top-2 logit gap", min_gap, "action histogram", np.bincount(acts, minlength=6))
    assert min_gap >= C.FF_MIN_GAP_REQUIRED, min_gap
    assert len(np.unique(acts)) >= 3, "degenerate policy: pick another FF_INIT_SEED"
    out["min_logit_gap"] = np.float64(min_gap)
    save("sampler_ff", **out)


def _record_ff_batches(s, agent, C, out):
    min_gap = np.inf
    for itr in range(C.FF_BATCHES):
        agent.sample_mode(itr)
        smp, infos = s.obtain_samples(itr)
        k = f"ff{itr}_"
        prob = smp.agent.agent_info.dist_info.prob.numpy()
        assert np.all((prob == 0) | (prob == 1)), "policy not sharp enough: stochastic draws recorded"
        # top-2 logit gap of every forward of this batch (recomputed: the sampler stores prob only)
        with torch.no_grad():
            obs = smp.env.observation
            m = agent.model
            x = obs.reshape(-1, 4, 104, 80).float().mul(1. / 255)
            h = m.conv(x)
            top = torch.topk(m.pi(h), 2, dim=-1).values
            min_gap = min(min_gap, float((top[:, 0] - top[:, 1]).min()))
        out.update({
            k + "obs_crc": C.obs_crc(smp.env.observation.numpy()),
            k + "reward": smp.env.reward.numpy().copy(),
            k + "prev_reward": smp.env.prev_reward.numpy().copy(),
            k + "done": smp.env.done.numpy().copy(),
            k + "game_score": smp.env.env_info.game_score.numpy().copy(),
            k + "traj_done": smp.env.env_info.traj_done.numpy().copy(),
            k + "action": smp.agent.action.numpy().copy(),
            k + "prev_action": smp.agent.prev_action.numpy().copy(),
            k + "prob": prob.copy(),
            k + "value": smp.agent.agent_info.value.numpy().copy(),
            k + "bootstrap_value": smp.agent.bootstrap_value.numpy().copy(),
            k + "traj_fields": np.array(sorted(
                (ti["Length"], ti["Return"], ti["NonzeroRewards"], ti["DiscountedReturn"])
                for ti in infos), dtype=np.float64).reshape(-1, 4)})
    return min_gap


def gen_algos():
    """Whole update iterations of the REFERENCE algorithms: PPO.optimize_agent / A2C.optimize_agent
    (rlpyt/algos/pg/{ppo,a2c,base}.py) with the reference AtariFfAgent on CPU, two consecutive
    iterations each: the per-update diagnostics and the parameters after every iteration."""
    import algo_cases as C
    from rlpyt.agents.pg.atari import AtariFfAgent
    from rlpyt.agents.pg.base import AgentInfo
    from rlpyt.algos.pg.a2c import A2C
    from rlpyt.algos.pg.ppo import PPO
    from rlpyt.envs.base import EnvSpaces
    from rlpyt.samplers.collections import AgentSamplesBsv, BatchSpec, EnvSamples, Samples
    from rlpyt.spaces.int_box import IntBox
    spaces = EnvSpaces(observation=IntBox(0, 256, shape=(4, 104, 80), dtype="uint8"),
                       action=IntBox(0, C.A))
    inp = C.batch_inputs()
    out = {}
    from rlpyt.agents.pg.atari import AtariLstmAgent
    from rlpyt.agents.pg.base import AgentInfoRnn
    from rlpyt.models.pg.atari_lstm_model import RnnState
    for name, algo_name, kwargs, mbr in C.CASES:
        torch.manual_seed(C.INIT_SEED)
        lstm = name in C.LSTM_CASES
        agent = AtariLstmAgent() if lstm else AtariFfAgent()
        agent.initialize(spaces)
        obs = inp["observation"]
        prev_action, action = inp["all_action"][:-1], inp["all_action"][1:]
        prev_reward, reward = inp["all_reward"][:-1], inp["all_reward"][1:]
        with torch.no_grad():      # the behaviour policy = the initial parameters
            if lstm:
                h0, c0 = C.lstm_init_state()                       # [B, N, H]
                init = RnnState(h=h0.transpose(0, 1).contiguous(), c=c0.transpose(0, 1).contiguous())
                dist_info, value, _ = agent(obs, prev_action, prev_reward, init)
                bv = (value[-1] + 0.25).unsqueeze(0)
                prev_rnn = RnnState(h=torch.zeros((C.T,) + tuple(h0.shape)),
                                    c=torch.zeros((C.T,) + tuple(c0.shape)))
                prev_rnn.h[0], prev_rnn.c[0] = h0, c0              # only row 0 is read by the algos
                agent_info = AgentInfoRnn(dist_info=dist_info, value=value, prev_rnn_state=prev_rnn)
            else:
                dist_info, value = agent(obs, prev_action, prev_reward)
                _, bv = agent(obs[-1], action[-1], reward[-1])
                bv = (bv + 0.25).unsqueeze(0)
                agent_info = AgentInfo(dist_info=dist_info, value=value)
        samples = Samples(
            agent=AgentSamplesBsv(action=action, prev_action=prev_action,
                                  agent_info=agent_info,
                                  bootstrap_value=bv),
            env=EnvSamples(observation=obs, reward=reward, prev_reward=prev_reward,
                           done=inp["done"], env_info=()))
        algo = (PPO if algo_name == "PPO" else A2C)(**kwargs)
        algo.initialize(agent=agent, n_itr=C.N_ITR, batch_spec=BatchSpec(C.T, C.B),
                        mid_batch_reset=mbr, examples=None, world_size=1, rank=0)
        out.update({f"{name}_old_prob": dist_info.prob.numpy().copy(),
                    f"{name}_old_value": value.numpy().copy(),
                    f"{name}_bootstrap_value": bv.numpy().copy()})
        np.random.seed(C.SHUFFLE_SEED)
        for itr in range(C.N_RUN):
            agent.train_mode(itr)
            info = algo.optimize_agent(itr, samples)
            for f in ("loss", "gradNorm", "entropy", "perplexity"):
                out[f"{name}_itr{itr}_{f}"] = np.atleast_1d(np.array(getattr(info, f),
                                                                    dtype=np.float64))
            params = list(agent.parameters())
            sums, abs_sums = C.param_stats(params)
            out[f"{name}_itr{itr}_param_sums"] = sums
            out[f"{name}_itr{itr}_param_abs_sums"] = abs_sums
            for n, p in agent.model.named_parameters():
                if p.numel() <= 4096:
                    out[f"{name}_itr{itr}_param__{n}"] = p.detach().numpy().copy()
    save("algos", **out)


def gen_algos_big():
    """The reference PPO.optimize_agent + AtariFfAgent on CPU at [T=64, B=64] (M = 1024 per
    minibatch: the size from which this repo's update takes its split-GEMM / bf16-split conv
    kernels), SGD, two iterations -- algo_cases.BIG_CASE."""
    import algo_cases as C
    from rlpyt.agents.pg.atari import AtariFfAgent
    from rlpyt.agents.pg.base import AgentInfo
    from rlpyt.algos.pg.ppo import PPO
    from rlpyt.envs.base import EnvSpaces
    from rlpyt.samplers.collections import AgentSamplesBsv, BatchSpec, EnvSamples, Samples
    from rlpyt.spaces.int_box import IntBox
    spaces = EnvSpaces(observation=IntBox(0, 256, shape=(4, 104, 80), dtype="uint8"),
                       action=IntBox(0, C.A))
    name, _algo, kwargs, mbr = C.BIG_CASE
    T, B = C.BIG_T, C.BIG_B
    inp = C.batch_inputs(T, B, seed=78)
    torch.manual_seed(C.INIT_SEED)
    agent = AtariFfAgent()
    agent.initialize(spaces)
    obs = inp["observation"]
    prev_action, action = inp["all_action"][:-1], inp["all_action"][1:]
    prev_reward, reward = inp["all_reward"][:-1], inp["all_reward"][1:]
    with torch.no_grad():
        dist_info, value = agent(obs, prev_action, prev_reward)
        _, bv = agent(obs[-1], action[-1], reward[-1])
        bv = (bv + 0.25).unsqueeze(0)
    samples = Samples(
        agent=AgentSamplesBsv(action=action, prev_action=prev_action,
                              agent_info=AgentInfo(dist_info=dist_info, value=value),
                              bootstrap_value=bv),
        env=EnvSamples(observation=obs, reward=reward, prev_reward=prev_reward,
                       done=inp["done"], env_info=()))
    algo = PPO(**kwargs)
    algo.initialize(agent=agent, n_itr=C.N_ITR, batch_spec=BatchSpec(T, B), mid_batch_reset=mbr,
                    examples=None, world_size=1, rank=0)
    out = {f"{name}_old_prob": dist_info.prob.numpy().copy(),
           f"{name}_old_value": value.numpy().copy(), f"{name}_bootstrap_value": bv.numpy().copy(),
           f"{name}_obs_crc": np.int64(int(obs.to(torch.int64).sum()))}
    np.random.seed(C.SHUFFLE_SEED)
    for itr in range(C.N_RUN):
        agent.train_mode(itr)
        info = algo.optimize_agent(itr, samples)
        for f in ("loss", "gradNorm", "entropy", "perplexity"):
            out[f"{name}_itr{itr}_{f}"] = np.atleast_1d(np.array(getattr(info, f), dtype=np.float64))
        sums, abs_sums = C.param_stats(list(agent.parameters()))
        out[f"{name}_itr{itr}_param_sums"] = sums
        out[f"{name}_itr{itr}_param_abs_sums"] = abs_sums
        for n, p in agent.model.named_parameters():
            if p.numel() <= 8192:
                out[f"{name}_itr{itr}_param__{n}"] = p.detach().numpy().copy()
            else:       # a strided sample of the big tensors (trunk weight: every 433rd element)
                out[f"{name}_itr{itr}_paramsample__{n}"] = p.detach().reshape(-1)[::433].numpy().copy()
    save("algos_big", **out)


def gen_dqn_iterations():
    """The reference DQN.optimize_agent (rlpyt/algos/dqn/dqn.py:158-190) with its AtariDqnAgent
    and frame replay buffers on CPU: append, sample, loss, clip, Adam, priority and target
    updates, over several iterations of fixed sampler batches."""
    import algo_cases as C
    from collections import namedtuple
    from rlpyt.agents.dqn.atari.atari_dqn_agent import AtariDqnAgent
    from rlpyt.algos.dqn.dqn import DQN
    from rlpyt.envs.base import EnvSpaces
    from rlpyt.samplers.collections import BatchSpec
    from rlpyt.spaces.int_box import IntBox
    spaces = EnvSpaces(observation=IntBox(0, 256, shape=(4, 104, 80), dtype="uint8"),
                       action=IntBox(0, C.A))
    Env = namedtuple("Env", ["observation", "reward", "done"])
    Agent = namedtuple("Agent", ["action"])
    Smp = namedtuple("Smp", ["agent", "env"])
    out = {}
    for name, kwargs, n_itr in C.DQN_CASES:
        batches = C.dqn_batches(n_itr)
        torch.manual_seed(C.INIT_SEED)
        if name.startswith("catdqn"):
            from rlpyt.agents.dqn.atari.atari_catdqn_agent import AtariCatDqnAgent
            from rlpyt.algos.dqn.cat_dqn import CategoricalDQN
            agent, algo = AtariCatDqnAgent(n_atoms=51), CategoricalDQN(**kwargs)
        else:
            agent, algo = AtariDqnAgent(), DQN(**kwargs)
        agent.initialize(spaces)
        b0 = batches[0]
        examples = dict(observation=b0["observation"][0, 0], action=b0["action"][0, 0],
                        reward=b0["reward"][0, 0], done=b0["done"][0, 0])
        algo.initialize(agent=agent, n_itr=n_itr, batch_spec=BatchSpec(C.DQN_T, C.DQN_B),
                        mid_batch_reset=True, examples=examples, world_size=1, rank=0)
        np.random.seed(C.SHUFFLE_SEED)
        for itr, b in enumerate(batches):
            agent.train_mode(itr)
            info = algo.optimize_agent(itr, Smp(agent=Agent(action=b["action"]),
                                                env=Env(observation=b["observation"],
                                                        reward=b["reward"], done=b["done"])))
            for f in ("loss", "gradNorm", "tdAbsErr"):
                out[f"{name}_itr{itr}_{f}"] = np.array(getattr(info, f), dtype=np.float64)
            out[f"{name}_itr{itr}_param_abs_sums"] = C.param_stats(list(agent.model.parameters()))[1]
            out[f"{name}_itr{itr}_target_abs_sums"] = C.param_stats(
                list(agent.target_model.parameters()))[1]
            if kwargs["prioritized_replay"]:
                out[f"{name}_itr{itr}_tree_root"] = np.float64(
                    algo.replay_buffer.priority_tree.tree[0])
        out[f"{name}_update_counter"] = np.int64(algo.update_counter)
    save("dqn_iterations", **out)


def gen_r2d1_iterations():
    """The reference R2D1.optimize_agent (rlpyt/algos/dqn/r2d1.py:133-345) with its
    AtariR2d1Agent and prioritized sequence frame replay on CPU: input priorities, append with
    stored LSTM states, sequence sampling, warm-up + training passes, loss, priorities, target
    updates."""
    import algo_cases as C
    from collections import namedtuple
    from rlpyt.agents.dqn.atari.atari_r2d1_agent import AtariR2d1Agent
    from rlpyt.agents.dqn.r2d1_agent import AgentInfo
    from rlpyt.algos.dqn.r2d1 import R2D1
    from rlpyt.envs.base import EnvSpaces
    from rlpyt.models.dqn.atari_r2d1_model import RnnState
    from rlpyt.samplers.collections import BatchSpec
    from rlpyt.spaces.int_box import IntBox
    spaces = EnvSpaces(observation=IntBox(0, 256, shape=(4, 104, 80), dtype="uint8"),
                       action=IntBox(0, C.A))
    Env = namedtuple("Env", ["observation", "reward", "prev_reward", "done"])
    Agent = namedtuple("Agent", ["action", "prev_action", "agent_info"])
    Smp = namedtuple("Smp", ["agent", "env"])
    batches = C.r2d1_batches()
    torch.manual_seed(C.INIT_SEED)
    agent = AtariR2d1Agent(model_kwargs=dict(C.R2D1_MODEL))
    agent.initialize(spaces)
    algo = R2D1(**C.R2D1_KWARGS)
    b0 = batches[0]
    examples = dict(observation=b0["observation"][0, 0], action=b0["all_action"][1, 0],
                    reward=b0["all_reward"][1, 0], done=b0["done"][0, 0],
                    agent_info=AgentInfo(q=b0["q"][0, 0],
                                         prev_rnn_state=RnnState(h=b0["h"][0, 0], c=b0["c"][0, 0])))
    algo.initialize(agent=agent, n_itr=C.R2D1_ITRS, batch_spec=BatchSpec(C.R2D1_T, C.R2D1_B),
                    mid_batch_reset=False, examples=examples, world_size=1, rank=0)
    np.random.seed(C.SHUFFLE_SEED)
    out = {}
    for itr, b in enumerate(batches):
        agent.train_mode(itr)
        smp = Smp(agent=Agent(action=b["all_action"][1:], prev_action=b["all_action"][:-1],
                              agent_info=AgentInfo(q=b["q"],
                                                   prev_rnn_state=RnnState(h=b["h"], c=b["c"]))),
                  env=Env(observation=b["observation"], reward=b["all_reward"][1:],
                          prev_reward=b["all_reward"][:-1], done=b["done"]))
        info = algo.optimize_agent(itr, smp)
        out[f"r2d1_itr{itr}_loss"] = np.array(info.loss, dtype=np.float64)
        out[f"r2d1_itr{itr}_gradNorm"] = np.array(info.gradNorm, dtype=np.float64)
        out[f"r2d1_itr{itr}_priority"] = np.array([float(p) for p in info.priority])
        out[f"r2d1_itr{itr}_param_abs_sums"] = C.param_stats(list(agent.model.parameters()))[1]
        out[f"r2d1_itr{itr}_target_abs_sums"] = C.param_stats(
            list(agent.target_model.parameters()))[1]
        out[f"r2d1_itr{itr}_tree_root"] = np.float64(algo.replay_buffer.priority_tree.tree[0])
    out["r2d1_update_counter"] = np.int64(algo.update_counter)
    save("r2d1_iterations", **out)


def gen_agents():
    """Epsilon schedule of the reference DQN agents (rlpyt/agents/dqn/epsilon_greedy.py): scalar
    and rank-aware vector epsilon (log-spaced over the GLOBAL env index -- the one rank-aware
    piece of the DQN path under SyncRl), through sample_mode / eval_mode."""
    from rlpyt.agents.dqn.atari.atari_dqn_agent import AtariDqnAgent
    from rlpyt.envs.base import EnvSpaces
    from rlpyt.spaces.int_box import IntBox
    spaces = EnvSpaces(observation=IntBox(0, 256, shape=(4, 104, 80), dtype="uint8"),
                       action=IntBox(0, 6))
    out = {}
    for name, kw, global_B, env_ranks in [
            ("scalar", dict(eps_init=1., eps_final=0.05), 4, [0, 1, 2, 3]),
            ("vector_rank1", dict(eps_init=1., eps_final=0.1, eps_final_min=0.001), 8,
             [4, 5, 6, 7])]:
        agent = AtariDqnAgent(**kw)
        agent.initialize(spaces, global_B=global_B, env_ranks=env_ranks)
        agent.set_epsilon_itr_min_max(2, 10)
        eps = []
        for itr in range(13):
            agent.sample_mode(itr)
            eps.append(np.broadcast_to(np.asarray(agent.distribution.epsilon, dtype=np.float64),
                                       (len(env_ranks),)).copy())
        out[f"{name}_sample_eps"] = np.stack(eps)
        ev = []
        for itr in (0, 5):
            agent.eval_mode(itr)
            ev.append(float(agent.distribution.epsilon))
        out[f"{name}_eval_eps"] = np.array(ev)
    save("agents", **out)


def gen_runner_keys():
    """The tabular diagnostics the REFERENCE runners log (MinibatchRl / MinibatchRlEval,
    rlpyt/runners/minibatch_rl.py) -- names and order -- with SerialSampler + PPO on the tiny
    discrete env; ``StepsPerSecond`` among them is BASELINE.json's metric."""
    import json
    import types
    pp = types.ModuleType("pyprind")     # the reference's progress bar dependency is absent here

    class ProgBar:
        def __init__(self, n, **k):
            self.active = True

        def update(self, *a, **k):
            pass

        def stop(self):
            self.active = False
    pp.ProgBar = ProgBar
    sys.modules["pyprind"] = pp
    from rlpyt.agents.pg.categorical import CategoricalPgAgent
    from rlpyt.algos.pg.ppo import PPO
    from rlpyt.runners.minibatch_rl import MinibatchRl, MinibatchRlEval
    from rlpyt.samplers.serial.sampler import SerialSampler
    from rlpyt.utils.logging import logger
    from rlpyt.utils.tensor import infer_leading_dims, restore_leading_dims
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from rlpyt_amd.envs.synthetic import TinyDiscreteEnv

    class TinyModel(torch.nn.Module):
        def __init__(self, n_obs, n_act):
            super().__init__()
            self.body, self.pi = torch.nn.Linear(n_obs, 16), torch.nn.Linear(16, n_act)
            self.v = torch.nn.Linear(16, 1)

        def forward(self, obs, prev_action, prev_reward):
            lead_dim, T, B, _ = infer_leading_dims(obs, 1)
            h = torch.tanh(self.body(obs.view(T * B, -1)))
            return restore_leading_dims((torch.softmax(self.pi(h), -1), self.v(h).squeeze(-1)),
                                        lead_dim, T, B)

    class TinyAgent(CategoricalPgAgent):
        def __init__(self, **kw):
            super().__init__(ModelCls=TinyModel, **kw)

        def make_env_to_model_kwargs(self, env_spaces):
            return dict(n_obs=env_spaces.observation.shape[0], n_act=env_spaces.action.n)

    keys, cur = {}, ["train"]
    orig = logger.dump_tabular

    def capture(*a, **k):
        keys.setdefault(cur[0], []).append([kk for kk, _ in logger._tabular])
        return orig(*a, **k)
    logger.dump_tabular = capture
    for name, Runner, skw in (("train", MinibatchRl, {}),
                              ("eval", MinibatchRlEval, dict(eval_n_envs=2, eval_max_steps=200,
                                                             eval_max_trajectories=10,
                                                             eval_env_kwargs={}))):
        cur[0] = name
        sampler = SerialSampler(EnvCls=TinyDiscreteEnv, env_kwargs={}, batch_T=8, batch_B=4,
                                max_decorrelation_steps=0, **skw)
        Runner(algo=PPO(minibatches=2, epochs=1), agent=TinyAgent(), sampler=sampler,
               n_steps=8 * 4 * 6, log_interval_steps=8 * 4 * 3, affinity=dict(cuda_idx=None),
               seed=0).train()
    logger.dump_tabular = orig
    with open(os.path.join(HERE, "runner_keys.json"), "w") as f:
        json.dump({k: v[-1] for k, v in keys.items()}, f, indent=1)
    print("runner_keys.json:", {k: len(v[-1]) for k, v in keys.items()})


# reference class / function -> the attribute names of the duck-typed protocol the runner and
# the sibling components touch; ``protocol_cases.py`` holds the map to
# this repo's classes.
def gen_protocol():
    """``inspect.signature`` of every protocol method the reference's runners / algos / samplers
    call on each other, recorded from the REFERENCE classes: the drop-in boundary as
    data.  tests/test_protocol.py holds this repo's classes to it (same parameter names, order of
    the positional ones, defaults; extra parameters only with defaults)."""
    import importlib
    import inspect
    import json
    _install_pyprind_shim()
    from protocol_cases import PROTOCOL

    def describe(fn):
        sig = inspect.signature(fn)
        out = []
        for name, prm in sig.parameters.items():
            d = prm.default
            has = d is not inspect.Parameter.empty
            simple = isinstance(d, (int, float, bool, str, type(None)))
            out.append(dict(name=name, kind=prm.kind.name, has_default=has,
                            default=(d if (has and simple) else (repr(type(d).__name__) if has else None))))
        return out

    rec = {}
    for ref_path, (_ours, members) in PROTOCOL.items():
        mod, _, attr = ref_path.rpartition(".")
        obj = getattr(importlib.import_module(mod), attr)
        entry = {}
        if inspect.isclass(obj):
            for m in members:
                if not hasattr(obj, m):
                    raise AttributeError(f"{ref_path} has no {m}")
                member = inspect.getattr_static(obj, m)
                if isinstance(member, property):
                    entry[m] = "property"
                elif callable(getattr(obj, m)):
                    entry[m] = describe(getattr(obj, m))
                else:
                    v = getattr(obj, m)
                    entry[m] = dict(attr=(list(v) if isinstance(v, (tuple, list)) else v))
        else:
            entry["__call__"] = describe(obj)
        rec[ref_path] = entry
    with open(os.path.join(HERE, "protocol.json"), "w") as f:
        json.dump(rec, f, indent=1, sort_keys=True)
    print("protocol.json:", len(rec), "reference classes / functions,",
          sum(len(v) for v in rec.values()), "members")


def _install_pyprind_shim():
    import types
    if "pyprind" in sys.modules:
        return
    pp = types.ModuleType("pyprind")     # the reference's progress bar dependency is absent here

    class ProgBar:
        def __init__(self, n, **k):
            self.active = True

        def update(self, *a, **k):
            pass

        def stop(self):
            self.active = False
    pp.ProgBar = ProgBar
    sys.modules["pyprind"] = pp


if __name__ == "__main__":
    torch.manual_seed(0)
    np.random.seed(0)
    gens = dict(scans=gen_scans, nstep=gen_nstep, normalize=gen_normalize, losses=gen_losses,
                categorical=gen_categorical,
                sumtree=gen_sumtree, sumtree_unique=gen_sumtree_unique, frames=gen_frames, replay=gen_replay,
                seq_replay=gen_seq_replay, r2d1_rms=gen_r2d1_rms, catdqn=gen_catdqn,
                models=gen_models, sampler=gen_sampler, sampler_alt=gen_sampler_alt, sampler_ff=gen_sampler_ff, algos=gen_algos, algos_big=gen_algos_big,
                dqn_iterations=gen_dqn_iterations,
                r2d1_iterations=gen_r2d1_iterations, agents=gen_agents,
                runner_keys=gen_runner_keys, protocol=gen_protocol)
    for name in (sys.argv[1:] or list(gens)):      # python make_golden.py [subset ...]
        gens[name]()
