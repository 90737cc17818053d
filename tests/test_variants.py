"""Kernel-variant coverage: every kernel INSTANTIATION compiled into librlpyt_hip.so is named by
a parity case below, and on the GPU each case (i) checks its result against the oracle / the
torch reference and (ii) asserts -- through the library's own launch counters
(``rlpyt_hip_variant_dump``) -- that this very instantiation produced the result.

Why: several entry points pick a kernel by size or alignment (``rlpyt_frames_gather_seq`` goes
wide at n*seq_T >= 2048, the scans switch at 64 K and 1 M columns, the gathers by pointer / row
alignment, ...).  A parity test that stays below a gate silently validates the wrong kernel.

* CPU (``-m "not gpu"``): the set of instantiations found in the built library (``nm``) must
  equal the keys of ``CASES`` -- adding a kernel or a template instantiation without a parity case
  fails the CPU suite.
* GPU (``-m gpu``): one parametrised test per instantiation.
"""
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from conftest import ROOT
from oracle import np_oracle as O

SO = os.path.join(ROOT, "rlpyt_amd", "csrc", "librlpyt_hip.so")


def _normalise(sym):
    """'void rlpyt::(anonymous namespace)::__device_stub__k<rlpyt::...::B16>(args)' -> 'k<B16>'"""
    s = sym
    for drop in ("rlpyt::", "(anonymous namespace)::", "__device_stub__"):
        s = s.replace(drop, "")
    if s.startswith("void "):
        s = s[5:]
    depth = 0
    for i, ch in enumerate(s):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            s = s[:i]
            break
    return s.strip()


def library_kernels():
    out = subprocess.run(["nm", "-C", SO], check=True, capture_output=True, text=True).stdout
    names = set()
    for line in out.splitlines():
        if "__device_stub__" in line:
            names.add(_normalise(line.split(" ", 2)[2]))
    return names


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def host(x):
    return x.cpu().numpy()


CASES = {}


def case(*names):
    def deco(fn):
        for n in names:
            assert n not in CASES, n
            CASES[n] = fn
        return fn
    return deco


def _ops():
    from rlpyt_amd import ops
    return ops


# ------------------------------------------------------------------------------------ scans
def _scan_case(mode, N, with_valid, T=6, variant=0):
    def run():
        ops = _ops()
        rng = np.random.RandomState(N % 9973 + mode)
        r = (0.5 * rng.randn(T, N)).astype(np.float32)
        v = rng.randn(T, N).astype(np.float32)
        d = rng.rand(T, N) < 0.05
        bv = rng.randn(1, N).astype(np.float32)
        exact = variant == 0
        cmp = (lambda a, b: np.array_equal(a, b)) if exact else \
            (lambda a, b: np.allclose(a, b, rtol=1e-5, atol=1e-5))
        if mode == 0:
            out = ops.gae(dev(r), dev(v), dev(d), dev(bv), 0.99, 0.98, with_valid=with_valid,
                          variant=variant)
            ea, er = O.generalized_advantage_estimation(r, v, d, bv, 0.99, 0.98)
            assert cmp(host(out[0]), ea) and cmp(host(out[1]), er)
            if with_valid:
                assert np.array_equal(host(out[2]), O.valid_from_done(d))
        else:
            out = ops.discount_return(dev(r), dev(d), dev(bv), 0.99, value=dev(v),
                                      with_valid=with_valid, variant=variant)
            er = O.discount_return(r, d, bv, 0.99)
            assert cmp(host(out[0]), er) and cmp(host(out[1]), er - v)
            if with_valid:
                assert np.array_equal(host(out[2]), O.valid_from_done(d))
    return run


for _mode in (0, 1):
    for _valid in (False, True):
        _v = "true" if _valid else "false"
        # (4 columns per lane, chunks of 8): N >= 1 M columns, 16-byte aligned, N % 4 == 0
        CASES[f"scan_exact_kernel<{_mode}, 4, 8, {_v}>"] = _scan_case(_mode, 1 << 20, _valid)
        # (1, 8): 64 K <= N; ragged N keeps it off the 4-wide path at any size
        CASES[f"scan_exact_kernel<{_mode}, 1, 8, {_v}>"] = _scan_case(_mode, 65536 + 3, _valid)
        # (1, 32): few columns, one wave per workgroup (the [128, 256] config shape)
        CASES[f"scan_exact_kernel<{_mode}, 1, 32, {_v}>"] = _scan_case(_mode, 256, _valid, T=128)
        CASES[f"scan_segmented_kernel<{_mode}, {_v}>"] = _scan_case(_mode, 256, _valid, T=128,
                                                                    variant=1)


def _valid_case(N):
    def run():
        d = np.random.RandomState(N % 977).rand(5, N) < 0.1
        assert np.array_equal(host(_ops().valid_from_done(dev(d))), O.valid_from_done(d))
    return run


CASES["valid_kernel<4>"] = _valid_case(4 * 256 * 256)
CASES["valid_kernel<1>"] = _valid_case(1001)


def _nstep_case(N):
    def run():
        rng = np.random.RandomState(N)
        r = rng.randn(9, N).astype(np.float32)
        d = rng.rand(9, N) < 0.1
        ret, dn = _ops().discount_return_n_step(dev(r), dev(d), 3, 0.99)
        er, edn = O.discount_return_n_step(r, d, 3, 0.99)
        assert np.array_equal(host(ret), er) and np.array_equal(host(dn), edn)
    return run


CASES["nstep_kernel<4>"] = _nstep_case(4096)
CASES["nstep_kernel<1>"] = _nstep_case(333)


# ------------------------------------------------------------------------ losses / normalise
@case("norm_pass1_kernel", "norm_pass2_kernel", "norm_pass3_kernel")
def _norm():
    import test_hip_parity as P
    P.test_normalize_golden(_ops(), "cfg")
    P.test_normalize_large(_ops())


@case("pg_loss_kernel<0>", "pg_loss_finalize_kernel", "valid_partial_kernel")
def _ppo_loss():
    import test_hip_parity as P
    P.test_ppo_loss_golden(_ops(), "ppo_cfg")
    P.test_ppo_loss_golden(_ops(), "ppo_valid")


@case("pg_loss_kernel<1>")
def _a2c_loss():
    import test_hip_parity as P
    P.test_a2c_loss_golden(_ops(), "a2c_cfg")
    P.test_a2c_loss_golden(_ops(), "a2c_valid")


@case("dqn_loss_kernel", "dqn_loss_finalize_kernel")
def _dqn_loss():
    import test_hip_parity as P
    for name in ("dqn", "ddqn", "dqn_mse"):
        P.test_dqn_loss_golden(_ops(), name)


@case("r2d1_column_kernel", "r2d1_finalize_kernel", "scale_by_device_scalar_kernel")
def _r2d1_loss():
    import test_hip_parity as P
    P.test_r2d1_loss_golden(_ops(), "r2d1")
    P.test_r2d1_loss_golden(_ops(), "r2d1_huber")


@case("cat_denom_kernel", "cat_dqn_loss_kernel", "cat_dqn_finalize_kernel")
def _cat_loss():
    import test_hip_parity as P
    for name in ("cat", "cat_double", "cat_valid", "cat_peaky"):
        P.test_cat_dqn_loss_golden(_ops(), name)


@case("rms_partial_kernel", "rms_finalize_kernel", "rms_merge_kernel", "rms_count_kernel",
      "rms_normalize_kernel")
def _rms():
    import test_hip_parity as P
    P.test_running_mean_std_golden(_ops())


def _head_loss_case(K, A=6):
    def run():
        """Fused heads + PPO loss against the reference's statement sequence in f64."""
        ops = _ops()
        g = torch.Generator().manual_seed(K + A)
        M = 300
        h = torch.relu(torch.randn(M, K, generator=g))
        wp, bp = torch.randn(A, K, generator=g) * 0.05, torch.randn(A, generator=g) * 0.1
        wv, bv = torch.randn(1, K, generator=g) * 0.05, torch.randn(1, generator=g) * 0.1
        po = torch.softmax(torch.randn(M, A, generator=g), -1)
        act = torch.randint(0, A, (M,), generator=g)
        adv, ret = torch.randn(M, generator=g), torch.randn(M, generator=g)
        valid = (torch.rand(M, generator=g) < 0.8).float()
        ref_in = [t.double().requires_grad_(True) for t in (h, wp, bp, wv, bv)]
        pn = torch.softmax(ref_in[0] @ ref_in[1].t() + ref_in[2], -1)
        v = (ref_in[0] @ ref_in[3].t()).squeeze(-1) + ref_in[4]
        ref = O.ppo_loss_torch(pn, v, po.double(), act, adv.double(), ret.double(), valid.double(),
                               0.1, 1.0, 0.01)
        ref[0].backward()
        d_in = [t.cuda().requires_grad_(True) for t in (h, wp, bp, wv, bv)]
        loss, sc = ops.ppo_head_loss(*d_in, po.cuda(), act.cuda(), adv.cuda(), ret.cuda(),
                                     valid.cuda(), 0.1, 1.0, 0.01)
        loss.backward()
        np.testing.assert_allclose(host(sc), [x.item() for x in ref], rtol=2e-5, atol=1e-6)
        for a, b in zip(d_in, ref_in):
            rg = b.grad.numpy()
            np.testing.assert_allclose(host(a.grad), rg, rtol=1e-4,
                                       atol=2e-6 * float(np.abs(rg).max()) + 1e-9)
    return run


def _trunk_head_loss_case(K, A):
    def run():
        """The same with the trunk's bias add + ReLU fused in (pre-activation in, dL/dz and the
        trunk-bias gradient out) against the same statements in f64."""
        ops = _ops()
        g = torch.Generator().manual_seed(3 * K + A)
        M = 300
        z = torch.randn(M, K, generator=g)
        tb = torch.randn(K, generator=g) * 0.3
        wp, bp = torch.randn(A, K, generator=g) * 0.05, torch.randn(A, generator=g) * 0.1
        wv, bv = torch.randn(1, K, generator=g) * 0.05, torch.randn(1, generator=g) * 0.1
        po = torch.softmax(torch.randn(M, A, generator=g), -1)
        act = torch.randint(0, A, (M,), generator=g)
        adv, ret = torch.randn(M, generator=g), torch.randn(M, generator=g)
        ref_in = [t.double().requires_grad_(True) for t in (z, wp, bp, wv, bv, tb)]
        hh = torch.relu(ref_in[0] + ref_in[5])
        pn = torch.softmax(hh @ ref_in[1].t() + ref_in[2], -1)
        v = (hh @ ref_in[3].t()).squeeze(-1) + ref_in[4]
        ref = O.ppo_loss_torch(pn, v, po.double(), act, adv.double(), ret.double(), None, 0.1, 1.0, 0.01)
        ref[0].backward()
        d_in = [t.cuda().requires_grad_(True) for t in (z, wp, bp, wv, bv, tb)]
        loss, sc = ops.ppo_head_loss(*d_in[:5], po.cuda(), act.cuda(), adv.cuda(), ret.cuda(), None,
                                     0.1, 1.0, 0.01, trunk_bias=d_in[5])
        loss.backward()
        np.testing.assert_allclose(host(sc), [x.item() for x in ref], rtol=2e-5, atol=1e-6)
        for a, b in zip(d_in, ref_in):
            rg = b.grad.numpy()
            np.testing.assert_allclose(host(a.grad), rg, rtol=1e-4,
                                       atol=2e-6 * float(np.abs(rg).max()) + 1e-9)
    return run


for _ki, _K in ((8, 512), (4, 256)):
    for _am, _A in ((4, 3), (6, 6), (8, 8)):      # action slots held in registers: A <= 4, <= 6, <= 8
        CASES[f"ppo_head_loss_kernel<{_ki}, {_am}, false>"] = _head_loss_case(_K, _A)
        CASES[f"ppo_head_loss_kernel<{_ki}, {_am}, true>"] = _trunk_head_loss_case(_K, _A)
CASES["head_reduce_finalize_kernel"] = _head_loss_case(512)


# ---------------------------------------------------------------------------------- gathers
_VEC = {"B16": 16, "unsigned long": 8, "unsigned int": 4, "unsigned char": 1}


def _gather_case(vec, wide, pair):
    def run():
        ops = _ops()
        w = _VEC[vec]
        nvec = 260 if wide else 3
        # row bytes: a multiple of w but of no larger power of two (for w < 16), so the launcher
        # lands exactly on the w-byte vector type
        row = nvec * w
        if w < 16 and (row % (2 * w)) == 0:
            row += w
        rng = np.random.RandomState(row)
        T, B = 11, 7
        src = rng.randint(0, 256, size=(T, B, row)).astype(np.uint8)
        if pair:
            t_idx, b_idx = rng.randint(-1, T, size=40), rng.randint(0, B, size=40)
            out = ops.gather_rows(dev(src), dev(t_idx), dev(b_idx))
            assert np.array_equal(host(out), src[t_idx, b_idx])
        else:
            idx = rng.permutation(T * B)[:50]
            out = ops.gather_tb(dev(src), dev(idx))
            assert np.array_equal(host(out), src[idx % T, idx // T])
    return run


for _vec in _VEC:
    for _pair in (False, True):
        _m = "MapPair" if _pair else "MapTB"
        CASES[f"gather_wide_kernel<{_vec}, {_m}>"] = _gather_case(_vec, True, _pair)
        CASES[f"gather_flat_kernel<{_vec}, {_m}>"] = _gather_case(_vec, False, _pair)


def _frames_ring(rng, T, B, C, H, W, p_done):
    frames = rng.randint(0, 256, size=(T + C - 1, B, H, W)).astype(np.uint8)
    frames[:C - 1] = frames[-(C - 1):]          # rows mirrored after a wrap (replays/frame.py:54-58)
    done = rng.rand(T, B) < p_done
    return frames, done


def _frames_case(H, W):
    def run():
        ops = _ops()
        rng = np.random.RandomState(H * W)
        T, B, C, n = 40, 5, 4, 33
        frames, done = _frames_ring(rng, T, B, C, H, W, 0.1)
        T_idxs, B_idxs = rng.randint(0, T, size=n), rng.randint(0, B, size=n)
        T_idxs[:3] = (0, 1, T - 1)
        obs = ops.frames_gather(dev(frames), dev(done), dev(T_idxs), dev(B_idxs), C)
        assert np.array_equal(host(obs), O.frames_gather(frames, done, T_idxs, B_idxs, C))
        sT = rng.randint(0, T, size=6)
        sT[0] = T - 3
        seq = ops.frames_gather_seq(dev(frames), dev(done), dev(sT), dev(B_idxs[:6]), C, 9)
        assert np.array_equal(host(seq), O.frames_gather_seq(frames, done, sT, B_idxs[:6], C, 9))
    return run


CASES["frames_gather_kernel<B16>"] = _frames_case(104, 80)       # 8320 B frames
CASES["frames_gather_kernel<unsigned int>"] = _frames_case(9, 4)  # 36 B
CASES["frames_gather_kernel<unsigned char>"] = _frames_case(7, 5)  # 35 B


@case("frames_gather_wide_kernel")
def _frames_wide_r2d2_shape():
    """The R2D2 batch shape of BASELINE config #5: out [125, 64, 4, 104, 80] (266 MB), i.e.
    n * seq_T = 8000 >= 2048, the size gate of the wide kernel.  Ring with wrap (mirrored head
    rows), start indices inside the last 3 ring rows, start index 0 and 1 (look-back wraps to
    the ring end), dones inside the C-1 look-back of sequence starts and inside the sequences
    (rlpyt/replays/sequence/frame.py:17-50).  Bit-exact against the oracle."""
    ops = _ops()
    rng = np.random.RandomState(5)
    T, B, C, H, W, n, seq_T = 200, 16, 4, 104, 80, 64, 125
    frames, done = _frames_ring(rng, T, B, C, H, W, 0.02)
    T_idxs, B_idxs = rng.randint(0, T, size=n), rng.randint(0, B, size=n)
    T_idxs[:6] = (T - 1, T - 2, T - 3, 0, 1, T - seq_T)
    for i in range(6, 12):           # done within the C-1 steps before the start of a sequence
        done[(T_idxs[i] - 1 - (i % 3)) % T, B_idxs[i]] = True
    seq = ops.frames_gather_seq(dev(frames), dev(done), dev(T_idxs), dev(B_idxs), C, seq_T)
    assert seq.shape == (seq_T, n, C, H, W)
    exp = O.frames_gather_seq(frames, done, T_idxs, B_idxs, C, seq_T)
    assert np.array_equal(host(seq), exp)
    # non-sequence call above the gate: DQN-style batch of 2048 observations
    n2 = 2048
    T2, B2 = rng.randint(0, T, size=n2), rng.randint(0, B, size=n2)
    obs = ops.frames_gather(dev(frames), dev(done), dev(T2), dev(B2), C)
    assert np.array_equal(host(obs), O.frames_gather(frames, done, T2, B2, C))


def _seq_case(row_bytes):
    def run():
        rng = np.random.RandomState(row_bytes)
        T, B, n, seq_T = 30, 4, 9, 7
        arr = rng.randint(0, 256, size=(T, B, row_bytes)).astype(np.uint8)
        t_idx = rng.randint(0, T - seq_T, size=n)
        t_idx[0], t_idx[1] = -3, T - seq_T - 1      # "wrap beginning" quirk, last legal start
        b_idx = rng.randint(0, B, size=n)
        out = _ops().extract_sequences(dev(arr), dev(t_idx), dev(b_idx), seq_T)
        assert np.array_equal(host(out), O.extract_sequences(arr, t_idx, b_idx, seq_T))
    return run


CASES["gather_seq_kernel<B16>"] = _seq_case(48)
CASES["gather_seq_kernel<unsigned int>"] = _seq_case(12)
CASES["gather_seq_kernel<unsigned char>"] = _seq_case(5)


@case("obs_to_nhwc_f32_kernel<4>", "obs_to_nhwc_f32_generic_kernel")
def _nhwc():
    import test_hip_parity as P
    P.test_obs_to_nhwc_fused_exact(_ops())


# ------------------------------------------------------------------------ sampler step kernels
@case("commit_rows_kernel")
def _commit_rows():
    import test_sampler_gpu as S
    S.test_commit_rows_matches_indexing()


@case("categorical_head_kernel<8>")
def _cat_head8():
    import test_sampler_gpu as S
    S.test_categorical_head_matches_torch(256, 512, 6)


@case("categorical_head_kernel<32>")
def _cat_head32():
    import test_sampler_gpu as S
    S.test_categorical_head_matches_torch(5, 64, 18)


@case("fc_small_kernel<1>")
def _fc1():
    import test_sampler_gpu as S
    S.test_fc_small_matches_torch(1, 64, 16)


@case("fc_small_kernel<2>", "fc_small_finish_kernel")
def _fc2():
    import test_sampler_gpu as S
    S.test_fc_small_matches_torch(128, 3456, 512)


@case("fc_small_kernel<4>")
def _fc4():
    import test_sampler_gpu as S
    S.test_fc_small_matches_torch(256, 3456, 512)


def _rollout_head_case(K):
    def run():
        """Rollout trunk (64 columns x 128-K slices, ksplit up to 27 here) + one-workgroup-per-row
        head vs torch (f64); step mode and value-only (bootstrap) mode; n crosses a 64-row block,
        Kin leaves a short last slice."""
        ops = _ops()
        from rlpyt_amd import _lib
        g = torch.Generator().manual_seed(K + 1)
        A, T, B, lo = 6, 4, 90, 5
        for n, Kin in ((70, 3456), (37, 160), (3, 16)):
            x = torch.randn(n, Kin, generator=g).cuda()
            w = (torch.randn(K, Kin, generator=g) * (2.0 / Kin ** 0.5)).cuda()
            fb = torch.randn(K, generator=g).cuda()
            wp, bp = (torch.randn(A, K, generator=g) * 0.05).cuda(), torch.randn(A, generator=g).cuda()
            wv, bv = (torch.randn(1, K, generator=g) * 0.05).cuda(), torch.randn(1, generator=g).cuda()
            u = torch.rand(T, n, generator=g).cuda()
            t_dev = torch.tensor([2], dtype=torch.int64, device="cuda")
            prob = torch.zeros(T, B, A, device="cuda")
            val = torch.zeros(T, B, device="cuda")
            act = torch.zeros(T + 1, B, dtype=torch.int64, device="cuda")
            out = torch.zeros(n, dtype=torch.int64, device="cuda")
            _lib.variant_reset()
            part, ks = ops.rollout_fc_partials(x, w)
            assert ks == -(-Kin // 128) and part.numel() >= ks * n * K * 4
            ops.rollout_head(part, ks, fb, wp, bp, wv, bv, u, t_dev, n, prob, val, act, lo, out)
            cnt = _lib.variant_counts()
            assert cnt.get("rollout_fc_kernel", 0) == 1
            assert cnt.get(f"rollout_head_kernel<{K // 256}>", 0) == 1
            h = torch.relu(x.double() @ w.double().t() + fb.double())
            rp = torch.softmax(h @ wp.double().t() + bp.double(), -1)
            rv = (h @ wv.double().t()).squeeze(-1) + bv.double()
            np.testing.assert_allclose(host(prob[2, lo:lo + n]), host(rp), rtol=2e-5, atol=1e-7)
            np.testing.assert_allclose(host(val[2, lo:lo + n]), host(rv), rtol=2e-5, atol=2e-5)
            cum = np.cumsum(host(prob[2, lo:lo + n]), axis=1, dtype=np.float32)
            exp = np.minimum((cum <= host(u[2])[:, None]).sum(1), A - 1)
            assert np.array_equal(host(act[3, lo:lo + n]), exp) and np.array_equal(host(out), exp)
            # the partial sums themselves: slice s of x @ w.T
            pv = part[:ks * n * K * 4].view(torch.float32).reshape(ks, n, K)
            ref_s = torch.stack([x[:, 128 * s_:128 * (s_ + 1)].double() @ w[:, 128 * s_:128 * (s_ + 1)].double().t()
                                 for s_ in range(ks)])
            assert (pv.double() - ref_s).abs().max().item() <= 2e-5 * ref_s.abs().max().item() + 1e-6
            # value-only mode writes the same value and nothing else
            keep = [prob.clone(), val.clone(), act.clone(), out.clone()]
            bvout = torch.zeros(n, device="cuda")
            ops.rollout_head(part, ks, fb, wp, bp, wv, bv, None, None, n, None, None, None, 0, None,
                             bootstrap_out=bvout)
            assert torch.equal(bvout, val[2, lo:lo + n])
            for a_, b_ in zip(keep, (prob, val, act, out)):
                assert torch.equal(a_, b_)
            # run-to-run deterministic
            part2, _ = ops.rollout_fc_partials(x, w)
            bv2 = torch.zeros(n, device="cuda")
            ops.rollout_head(part2, ks, fb, wp, bp, wv, bv, None, None, n, None, None, None, 0, None,
                             bootstrap_out=bv2)
            assert torch.equal(bv2, bvout)
            prob[2, lo:lo + n] = 0
            val[2, lo:lo + n] = 0
            act[3, lo:lo + n] = 0
            assert not prob.any() and not val.any() and not act.any()   # nothing else written
    return run


CASES["rollout_head_kernel<2>"] = _rollout_head_case(512)
CASES["rollout_head_kernel<1>"] = _rollout_head_case(256)
CASES["rollout_fc_kernel"] = _rollout_head_case(512)


@case("frame_push_kernel")
def _frame_push():
    """obs[t] = concat(obs[t-1][1:], newest frame), or a full row where slot >= 0; reward / done
    rows in the same launch.  Bit-exact vs torch indexing."""
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    T, B, C, H, W, lo, Bg, t = 5, 9, 4, 13, 16, 2, 6, 3
    obs = torch.randint(0, 256, (T, B, C, H, W), dtype=torch.uint8, generator=g).cuda()
    exp = obs.clone()
    new = torch.randint(0, 256, (Bg, H, W), dtype=torch.uint8, generator=g).cuda()
    full = torch.randint(0, 256, (Bg, C, H, W), dtype=torch.uint8, generator=g).cuda()
    slot = torch.tensor([-1, 0, -1, -1, 1, -1], dtype=torch.int32).cuda()
    rew_rows = torch.zeros(T + 1, B, device="cuda")
    done_rows = torch.zeros(T, B, dtype=torch.bool, device="cuda")
    rs = torch.randn(Bg, generator=g).cuda()
    ds = (torch.rand(Bg, generator=g) < 0.5).cuda()
    t_dev = torch.tensor([t], dtype=torch.int64, device="cuda")
    ops.frame_push(obs, t_dev, lo, new, full, slot, scalar_rows=(rew_rows, rs, done_rows, ds))
    for b in range(Bg):
        if slot[b] >= 0:
            exp[t, lo + b] = full[slot[b]]
        else:
            exp[t, lo + b, :C - 1] = exp[t - 1, lo + b, 1:]
            exp[t, lo + b, C - 1] = new[b]
    assert torch.equal(obs, exp)
    assert torch.equal(rew_rows[t + 1, lo:lo + Bg], rs) or torch.equal(rew_rows[t, lo:lo + Bg], rs)
    assert torch.equal(done_rows[t, lo:lo + Bg], ds)


# ------------------------------------------------------------------------------- conv stack
@case("conv1_fwd_kernel", "conv2_fwd_kernel<2>", "relu_mask_kernel")
def _conv_fwd_small():
    import test_conv_gpu as Cv
    for M in (1, 7, 200):          # conv1 split 4 / 4 / 2; conv2 two workgroups per image
        Cv.test_conv_forward_kernels(_ops(), M)


@case("conv2_fwd_x6_kernel")
def _conv_fwd_large():
    import test_conv_gpu as Cv
    Cv.test_conv_forward_kernels(_ops(), 300)
    Cv.test_conv_identity_weights_asymmetric(_ops())


@case("convs_fwd_fused_kernel")
def _convs_fwd_fused():
    import test_conv_gpu as Cv
    Cv.test_convs_forward_fused_equals_the_two_launches(_ops(), 300, True)
    Cv.test_convs_forward_fused_equals_the_two_launches(_ops(), 1100, False)


@case("conv2_bwd_x6_kernel", "conv1_wgrad_kernel", "reduce_partials_kernel")
def _conv_bwd():
    import test_conv_gpu as Cv
    for M in (1, 5, 700):
        Cv.test_conv_backward_kernels(_ops(), M)


@case("gemm_nt_x6_kernel<128>")
def _gemm_nt():
    import test_conv_gpu as Cv
    Cv.test_gemm_nt_bf16x6_is_f32_accurate(_ops(), 130, 200, 96)


@case("gemm_nt_x6_kernel<256>")
def _gemm_nt_tall():
    import test_conv_gpu as Cv
    Cv.test_gemm_nt_bf16x6_is_f32_accurate(_ops(), 8000, 2100, 64)     # 32 x 17 tiles of 256 x 128


@case("gemm_tn_x6_kernel", "gemm_reduce_slots_kernel")
def _gemm_tn_x6():
    import test_conv_gpu as Cv
    Cv.test_gemm_tn_bf16x6_is_f32_accurate(_ops(), 132, 200, 2080)    # ragged K chunks + partials


@case("dqn_conv1_kernel")
def _dqn_conv1_f32_mfma():
    """The f32-MFMA conv1 (callers that hand over packed weights only): same features as the bf16x3 one."""
    import test_dqn_convs_gpu as D
    from rlpyt_amd._lib import check, lib, ptr, stream
    ops = _ops()
    convs = [c.cuda() for c in D._stack(11)]
    args = [p for c in convs for p in (c.weight.detach(), c.bias.detach())]
    g = torch.Generator().manual_seed(12)
    obs = torch.randint(0, 256, (7, 4, 104, 80), dtype=torch.uint8, generator=g).cuda()
    want = ops.dqn_convs_fwd(obs, *args)
    packed = ops.dqn_convs_pack(args[0], args[2], args[4])
    ws = torch.empty(int(lib.rlpyt_dqn_convs_workspace_floats(7)), dtype=torch.float32, device="cuda")
    out = torch.empty((7, 6912), dtype=torch.float32, device="cuda")
    check(lib.rlpyt_dqn_convs_fwd_f32(ptr(obs), 7, None, ptr(args[1]), None, ptr(args[3]), None, ptr(args[5]),
                                      ptr(packed), 1. / 255, ptr(ws), ptr(out), stream()), "fwd")
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), want.cpu().numpy(), rtol=2e-4, atol=2e-6)


@case("dqn_conv1_x3_kernel", "dqn_t32_pack_kernel",
      "dqn_conv23_t32_kernel<32, 25, 19, 4, 4, 2, 3, false>", "dqn_conv23_t32_kernel<64, 12, 9, 3, 3, 1, 6, true>")
def _dqn_convs():
    import test_dqn_convs_gpu as D
    D.test_dqn_convs_match_torch_conv2d(3)
    D.test_dqn_convs_match_torch_conv2d(300)
    D.test_dqn_convs_identity_like_weights_asymmetric()


@case("dqn_x6_pack_kernel", "dqn_conv23_x6_kernel<32, 25, 19, 4, 4, 2, false>",
      "dqn_conv23_x6_kernel<64, 12, 9, 3, 3, 1, true>")
def _dqn_convs_x6_16x16_tiles():
    """conv2 / conv3 on 16 x 16 tiles (RLPYT_DQN_CONV23_T32=0, the A/B switch): same tests."""
    import test_dqn_convs_gpu as D
    os.environ["RLPYT_DQN_CONV23_T32"] = "0"
    try:
        D.test_dqn_convs_match_torch_conv2d(3)
        D.test_dqn_convs_match_torch_conv2d(300)
        D.test_dqn_convs_identity_like_weights_asymmetric()
    finally:
        del os.environ["RLPYT_DQN_CONV23_T32"]


@case("dqn_pack_weights_kernel", "dqn_conv23_kernel<32, 25, 19, 4, 4, 2, 128, false>",
      "dqn_conv23_kernel<64, 12, 9, 3, 3, 1, 144, true>")
def _dqn_convs_f32_mfma():
    """conv2 / conv3 on the f32 MFMA (RLPYT_DQN_CONV23_X6=0, the A/B switch): same tests."""
    import test_dqn_convs_gpu as D
    os.environ["RLPYT_DQN_CONV23_X6"] = "0"
    try:
        D.test_dqn_convs_match_torch_conv2d(3)
        D.test_dqn_convs_identity_like_weights_asymmetric()
    finally:
        del os.environ["RLPYT_DQN_CONV23_X6"]


@case("dqn_pack_bwd_weights_kernel", "dqn_dgrad3_kernel", "dqn_dgrad2_kernel", "dqn_wgrad1_kernel",
      "dqn_wgrad23_kernel<32, 25, 19, 4, 4, 2, 8, false>",
      "dqn_wgrad23_kernel<64, 12, 9, 3, 3, 1, 12, true>", "dqn_bwd_reduce_kernel")
def _dqn_convs_bwd():
    import test_dqn_convs_gpu as D
    D.test_dqn_convs_under_autograd_gradients_match_the_library_path(32, True)


@case("lstm_seq_step_kernel<8>", "lstm_seq_step_kernel<4>")
def _lstm_seq():
    import test_lstm_seq_gpu as L
    for H, I in ((512, 519), (256, 37)):
        L.test_lstm_sequence_matches_torch_lstm(H, I, 7, 5, True)
        L.test_lstm_sequence_matches_torch_lstm(H, I, 5, 70, False)


@case("lstm_seq_bwd_step_kernel<8>", "lstm_seq_bwd_step_kernel<4>")
def _lstm_seq_bwd():
    import test_lstm_seq_gpu as L
    for H, I in ((512, 519), (256, 37)):
        L.test_lstm_sequence_under_autograd_matches_torch_lstm(H, I, 7, 5, True, True)


@case("rnn_step_inputs_kernel")
def _rnn_step_inputs():
    import test_dqn_gpu as D
    D.test_r2d1_fused_sampling_step_equals_eager_step(dict(fc_size=64, lstm_size=32, head_size=32, dueling=True), True)


@case("q_head_kernel<2>")
def _q_head_512():
    import test_dqn_gpu as D
    D.test_mlp_q_head_matches_torch(8, 6912, 512, 6)


@case("q_head_kernel<1>")
def _q_head_256():
    import test_dqn_gpu as D
    D.test_mlp_q_head_matches_torch(5, 64, 256, 3)


@case("q_head_bwd_kernel")
def _q_head_bwd():
    import test_dqn_gpu as D
    D.test_mlp_q_head_under_autograd_matches_torch(128, 6912, 512, 6)
    D.test_mlp_q_head_under_autograd_matches_torch(7, 48, 256, 1)


@case("replay_append_kernel")
def _replay_append():
    import test_replay_append_gpu as A
    for shape in ((4, 104, 80, 16, False), (4, 3, 2, 2, True), (3, 7, 5, 3, False)):
        A.test_one_launch_append_equals_the_slice_assignments(shape)


@case("replay_step_fields_kernel")
def _replay_step_fields():
    """One-launch field gather of a single-step replay batch vs the row-by-row gathers + selects it
    replaces (and vs the reference's own arithmetic, n_step.py:16-43, in numpy): bit-exact, incl.
    rows 0 / T - 1 (wraps of t - 1 and t + n_step) and done flags on the row before."""
    ops = _ops()
    g = torch.Generator().manual_seed(9)
    T, B, n, n_step = 37, 5, 300, 3
    action = torch.randint(0, 6, (T, B), generator=g).cuda()
    reward = torch.randn(T, B, generator=g).cuda()
    done = (torch.rand(T, B, generator=g) < 0.3).cuda()
    ret = torch.randn(T, B, generator=g).cuda()
    done_n = (torch.rand(T, B, generator=g) < 0.3).cuda()
    t = torch.randint(0, T, (n,), generator=g)
    t[:4] = torch.tensor([0, T - 1, T - n_step, 1])
    b = torch.randint(0, B, (n,), generator=g)
    td, bd = t.cuda(), b.cuda()
    pa, pr, a, r, d, dn, tpa, tpr = ops.replay_step_fields(action, reward, done, ret, done_n, td, bd,
                                                           n_step)
    A, R, D, RT, DN = (host(x) for x in (action, reward, done, ret, done_n))
    tn, bn = t.numpy(), b.numpy()
    was = D[tn - 1, bn]                                   # numpy negative index = the ring wrap
    assert np.array_equal(host(pa), np.where(was, 0, A[tn - 1, bn]))
    assert np.array_equal(host(pr), np.where(was, np.float32(0), R[tn - 1, bn]))
    assert np.array_equal(host(a), A[tn, bn]) and np.array_equal(host(r), RT[tn, bn])
    assert np.array_equal(host(d), D[tn, bn]) and np.array_equal(host(dn), DN[tn, bn])
    nxt = (tn + n_step) % T
    assert np.array_equal(host(tpa), A[nxt - 1, bn]) and np.array_equal(host(tpr), R[nxt - 1, bn])
    # ... and the row-by-row device path gives the same tensors
    row = ops.gather_rows
    assert torch.equal(a, row(action, td, bd)) and torch.equal(tpr, row(reward, (td + n_step) % T - 1, bd))


@case("eps_greedy_kernel")
def _eps_greedy():
    import test_dqn_gpu as D
    D.test_eps_greedy_kernel_semantics()


@case("lstm_cell_kernel")
def _lstm_cell():
    import test_dqn_gpu as D
    D.test_lstm_step_matches_nn_lstm(64, 531, 512)


@case("sample_convs_kernel")
def _sample_convs():
    import test_sampler_gpu as S
    S.test_sample_convs_kernel_matches_separate_launches()


# ------------------------------------------------------------------------------- optimizer
@case("clip_adam_norm_kernel", "clip_adam_apply_kernel")
def _clip_adam():
    """ClipAdam.clip_and_step == torch.nn.utils.clip_grad_norm_ + torch.optim.Adam.step (the
    reference's two statements, rlpyt/algos/pg/ppo.py:100-104) in float64 on the CPU, over several
    steps with clipping active and inactive, odd tensor sizes (scalar tails), and a state_dict
    round trip into a plain torch.optim.Adam.  fp32 tolerance: rtol 2e-6 on parameters."""
    from rlpyt_amd.optim import ClipAdam
    g = torch.Generator().manual_seed(0)
    shapes = [(16, 4, 8, 8), (16,), (3456, 37), (5,), (7, 3), (1,)]
    params64 = [torch.randn(s, generator=g, dtype=torch.float64).requires_grad_(True) for s in shapes]
    params32 = [p.detach().float().cuda().requires_grad_(True) for p in params64]
    ref = torch.optim.Adam(params64, lr=3e-3, betas=(0.9, 0.99), eps=1e-5)
    opt = ClipAdam(params32, lr=3e-3, betas=(0.9, 0.99), eps=1e-5)
    assert opt.supports_fused()
    for it in range(6):
        scale = 10.0 if it % 2 == 0 else 1e-3        # clipped / not clipped
        for k, (p64, p32) in enumerate(zip(params64, params32)):
            gr = torch.randn(p64.shape, generator=g, dtype=torch.float64) * scale
            p64.grad = gr.clone()
            if k == 3:      # a gradient that is a view at an odd offset of a packed buffer
                packed = torch.zeros(p32.numel() + 3, device="cuda")
                packed[3:] = gr.float().cuda().reshape(-1)
                p32.grad = packed[3:].view(p32.shape)
            else:
                p32.grad = gr.float().cuda()
        n_ref = torch.nn.utils.clip_grad_norm_(params64, 1.0)
        ref.step()
        n = opt.clip_and_step(1.0)
        np.testing.assert_allclose(n.item(), n_ref.item(), rtol=1e-6)
        for p64, p32 in zip(params64, params32):
            np.testing.assert_allclose(host(p32.detach()), p64.detach().numpy(), rtol=2e-6, atol=1e-7)
    # same state layout as torch.optim.Adam: the snapshot loads into the plain optimizer
    plain = torch.optim.Adam([torch.zeros_like(p) for p in params32], lr=1.)
    plain.load_state_dict(opt.state_dict())
    st = plain.state[plain.param_groups[0]["params"][2]]
    assert float(st["step"]) == 6 and st["exp_avg"].shape == (3456, 37)
    np.testing.assert_allclose(host(st["exp_avg_sq"]), ref.state[params64[2]]["exp_avg_sq"].numpy(),
                               rtol=1e-5, atol=1e-12)


# --------------------------------------------------------------------------------- sum tree
@case("find_kernel", "advance_apply_kernel", "propagate_kernel", "unique_first_kernel",
      "update_leaves_kernel")
def _sumtree():
    import test_hip_parity as P
    P.test_sumtree_known_answer(_ops())
    P.test_sumtree_vs_oracle_random_stream(_ops())


@case("set_sampled_kernel")
def _sumtree_unique():
    import test_hip_parity as P
    P.test_sumtree_unique_sampling_matches_reference()


@case("update_tick_kernel")
def _update_tick():
    """Row ``*ctr`` of the hyper table + the update's index chunk land at their fixed addresses; the
    counter itself is advanced by the optimizer's apply kernel."""
    ops = _ops()
    n, cols, M = 5, 4, 1000
    g = torch.Generator().manual_seed(3)
    table = torch.randn(n, cols, generator=g).cuda()
    idx_all = torch.randint(0, 10 ** 6, (n * M,), generator=g).cuda()
    hyper = torch.zeros(cols, device="cuda")
    idx = torch.zeros(M, dtype=torch.int64, device="cuda")
    tick = torch.zeros(1, dtype=torch.int64, device="cuda")
    for cur in (0, 3, 4, 9):            # 9: beyond the table -> clamped to its last row
        ctr = torch.tensor([cur], dtype=torch.int64, device="cuda")
        ops.update_tick(ctr, table, hyper, idx_all, idx, tick)
        c = min(cur, n - 1)
        assert torch.equal(hyper, table[c]) and torch.equal(idx, idx_all[c * M:(c + 1) * M])
        assert int(tick.item()) == c and int(ctr.item()) == cur


@case("is_weights_kernel")
def _is_weights():
    import test_hip_parity as P
    P.test_is_weights_match_the_float64_expression(_ops())


@case("fill_f64_kernel", "write_input_pri_kernel")
def _sumtree_input_priorities():
    import test_hip_parity as P
    P.test_sumtree_streams_bit_exact(_ops(), "inpri")


# ===================================================================================== tests
def test_every_kernel_instantiation_has_a_parity_case():
    """CPU: the case table and the library agree (no unnamed kernel, no stale entry)."""
    lib = library_kernels()
    assert lib, "no kernels found in " + SO
    missing = sorted(lib - set(CASES))
    stale = sorted(set(CASES) - lib)
    assert not missing, f"kernel instantiations without a parity case: {missing}"
    assert not stale, f"parity cases naming kernels that are not in the library: {stale}"


def test_every_global_kernel_in_the_sources_is_named():
    """CPU: every ``__global__`` function of csrc/ appears (by base name) in the case table."""
    src_dir = os.path.join(ROOT, "rlpyt_amd", "csrc")
    names = set()
    for fn in os.listdir(src_dir):
        if fn.endswith((".hip", ".cpp")):
            text = open(os.path.join(src_dir, fn)).read()
            names |= set(re.findall(r"__global__[^;{(]*?void\s+(\w+)\s*\(", text))
    assert names
    covered = {k.split("<")[0] for k in CASES}
    assert names <= covered, sorted(names - covered)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_variant_runs_and_matches(name):
    from rlpyt_amd import _lib
    assert torch.cuda.is_available()
    _lib.variant_reset()
    CASES[name]()
    torch.cuda.synchronize()
    counts = _lib.variant_counts()
    assert name in counts, (f"parity case for {name} passed but that kernel never ran; "
                            f"launched instead: {sorted(counts)}")


@pytest.mark.gpu
def test_last_variant_reports_the_gate():
    """``rlpyt_hip_last_variant`` names the kernel behind the last call: the frame gather below
    and above its n*seq_T >= 2048 gate."""
    from rlpyt_amd import _lib, ops
    rng = np.random.RandomState(0)
    frames, done = _frames_ring(rng, 64, 4, 4, 8, 16, 0.05)
    f, d = dev(frames), dev(done)
    for n, expect in ((8, "frames_gather_kernel<B16>"), (2048, "frames_gather_wide_kernel")):
        ti, bi = rng.randint(0, 64, size=n), rng.randint(0, 4, size=n)
        out = ops.frames_gather(f, d, dev(ti), dev(bi), 4)
        assert _lib.last_variant() == expect
        assert np.array_equal(host(out), O.frames_gather(frames, done, ti, bi, 4))
