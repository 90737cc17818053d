"""GPU parity: the HIP path (through the C ABI) against the golden vectors generated from
the real reference, against the CPU oracle on seeded inputs, and -- at BASELINE.json's full
sizes -- through size-independent properties.

Bars: bit-exact for the EXACT scans, n-step returns, valid masks, sum-tree indices /
priorities / whole tree, frame and sequence gathers; fp32 tolerance (written per test) for
the segmented scan, advantage normalisation and the losses (their reductions have no
defined summation order on either side -- SURVEY.md App. B.1)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import np_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from rlpyt_amd import ops as _ops
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    return _ops


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def host(x):
    return x.cpu().numpy()


# ------------------------------------------------------------------------------------ scans
@pytest.mark.parametrize("name", ["cfg", "nodone", "dense", "t1", "t2", "kat"])
def test_scans_golden_bit_exact(ops, name):
    g = load_golden("scans")
    r, v, d, bv = (dev(g[f"{name}_{k}"]) for k in ("reward", "value", "done", "bv"))
    gamma, lam = float(g[f"{name}_gamma"]), float(g[f"{name}_lambda"])
    adv, ret, valid = ops.gae(r, v, d, bv, gamma, lam, with_valid=True)
    assert np.array_equal(host(adv), g[f"{name}_adv"])
    assert np.array_equal(host(ret), g[f"{name}_ret"])
    assert np.array_equal(host(valid), g[f"{name}_valid"])
    adv2, ret2 = ops.gae(r, v, d, bv, gamma, lam)
    assert torch.equal(adv, adv2) and torch.equal(ret, ret2)
    disc, dadv, dvalid = ops.discount_return(r, d, bv, gamma, value=v, with_valid=True)
    assert np.array_equal(host(disc), g[f"{name}_disc"])
    assert np.array_equal(host(dadv), g[f"{name}_disc"] - g[f"{name}_value"])
    assert np.array_equal(host(dvalid), g[f"{name}_valid"])
    assert np.array_equal(host(ops.discount_return(r, d, bv, gamma)), g[f"{name}_disc"])
    assert np.array_equal(host(ops.valid_from_done(d)), g[f"{name}_valid"])


def test_scan_1d_and_trailing_dims(ops):
    g = load_golden("scans")
    adv, ret = ops.gae(dev(g["oned_reward"]), dev(g["oned_value"]), dev(g["oned_done"]),
                       dev(g["oned_bv"]), 0.99, 0.9)
    assert np.array_equal(host(adv), g["oned_adv"]) and np.array_equal(host(ret), g["oned_ret"])
    # [T,B,K] trailing dims: flattened columns
    rng = np.random.RandomState(0)
    r = rng.randn(12, 5, 3).astype(np.float32)
    v = rng.randn(12, 5, 3).astype(np.float32)
    d = rng.rand(12, 5, 3) < 0.2
    bv = rng.randn(1, 5, 3).astype(np.float32)
    adv, ret = ops.gae(dev(r), dev(v), dev(d), dev(bv), 0.99, 0.95)
    ea, er = O.generalized_advantage_estimation(r, v, d, bv, 0.99, 0.95)
    assert np.array_equal(host(adv), ea) and np.array_equal(host(ret), er)


@pytest.mark.parametrize("T,N,p", [(128, 256, 0.01), (128, 4096, 0.01), (7, 65536 + 4, 0.3),
                                   (40, 70001, 0.05), (128, 1 << 20, 0.01), (1, 1, 1.0),
                                   (3, 63, 0.0), (300, 130, 0.02)])
def test_scans_vs_oracle_bit_exact(ops, T, N, p):
    """All launch geometries (1 and 4 columns per lane, ragged N) against the oracle."""
    rng = np.random.RandomState(T * 7 + N % 1000)
    r = (0.5 * rng.randn(T, N)).astype(np.float32)
    v = rng.randn(T, N).astype(np.float32)
    d = rng.rand(T, N) < p
    bv = rng.randn(1, N).astype(np.float32)
    adv, ret, valid = ops.gae(dev(r), dev(v), dev(d), dev(bv), 0.99, 0.98, with_valid=True)
    ea, er = O.generalized_advantage_estimation(r, v, d, bv, 0.99, 0.98)
    assert np.array_equal(host(adv), ea)
    assert np.array_equal(host(ret), er)
    assert np.array_equal(host(valid), O.valid_from_done(d))
    disc = ops.discount_return(dev(r), dev(d), dev(bv), 0.99)
    assert np.array_equal(host(disc), O.discount_return(r, d, bv, 0.99))
    assert np.array_equal(host(ops.valid_from_done(dev(d))), O.valid_from_done(d))


def test_scan_empty(ops):
    z = torch.zeros(0, 4, device="cuda")
    adv, ret = ops.gae(z, z, z.bool(), torch.zeros(1, 4, device="cuda"), 0.99, 0.9)
    assert adv.shape == (0, 4)
    z = torch.zeros(5, 0, device="cuda")
    adv, ret = ops.gae(z, z, z.bool(), torch.zeros(1, 0, device="cuda"), 0.99, 0.9)
    assert ret.shape == (5, 0)


@pytest.mark.parametrize("T,N", [(128, 256), (64, 32), (5, 16), (250, 100), (2, 1)])
def test_segmented_scan_tolerance(ops, T, N):
    """Re-associated latency variant: rtol 1e-5 / atol 1e-5 against the exact oracle (values
    are O(1..5); cancellation near zero makes a pure relative bound meaningless)."""
    rng = np.random.RandomState(T + N)
    r = (0.5 * rng.randn(T, N)).astype(np.float32)
    v = rng.randn(T, N).astype(np.float32)
    d = rng.rand(T, N) < 0.02
    bv = rng.randn(1, N).astype(np.float32)
    adv, ret, valid = ops.gae(dev(r), dev(v), dev(d), dev(bv), 0.99, 0.98, with_valid=True,
                              variant=ops.SCAN_SEGMENTED)
    ea, er = O.generalized_advantage_estimation(r, v, d, bv, 0.99, 0.98)
    np.testing.assert_allclose(host(adv), ea, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(host(ret), er, rtol=1e-5, atol=1e-5)
    assert np.array_equal(host(valid), O.valid_from_done(d))
    disc, dadv = ops.discount_return(dev(r), dev(d), dev(bv), 0.99, value=dev(v),
                                     variant=ops.SCAN_SEGMENTED)
    np.testing.assert_allclose(host(disc), O.discount_return(r, d, bv, 0.99), rtol=1e-5,
                               atol=1e-5)


def test_scan_properties_full_size(ops):
    """Size-independent properties at a scaled shape (T=128, N=2^20):
    lambda=1 GAE return == discounted return (up to fp32 re-association);
    all-done => advantage = reward - value exactly; linearity in reward for done=0."""
    T, N = 128, 1 << 20
    g = torch.Generator(device="cuda").manual_seed(0)
    r = 0.5 * torch.randn(T, N, device="cuda", generator=g)
    v = torch.randn(T, N, device="cuda", generator=g)
    d = torch.rand(T, N, device="cuda", generator=g) < 0.01
    bv = torch.randn(1, N, device="cuda", generator=g)
    adv, ret = ops.gae(r, v, d, bv, 0.99, 1.0)
    disc = ops.discount_return(r, d, bv, 0.99)
    torch.testing.assert_close(ret, disc, rtol=1e-4, atol=1e-4)
    ones = torch.ones_like(d)
    adv, ret = ops.gae(r, v, ones, bv, 0.99, 0.98)
    assert torch.equal(adv, r - v) and torch.equal(ret, (r - v) + v)
    zeros = torch.zeros_like(d)
    a1, _ = ops.gae(r, torch.zeros_like(v), zeros, torch.zeros_like(bv), 0.99, 0.98)
    a2, _ = ops.gae(2 * r, torch.zeros_like(v), zeros, torch.zeros_like(bv), 0.99, 0.98)
    assert torch.equal(a2, 2 * a1)  # scaling by 2 is exact in fp32


# ----------------------------------------------------------------------------------- n-step
@pytest.mark.parametrize("name", ["r2d1", "n3", "n1", "n2"])
def test_nstep_golden_bit_exact(ops, name):
    g = load_golden("nstep")
    r, d = dev(g[f"{name}_reward"]), dev(g[f"{name}_done"])
    n, gamma = int(g[f"{name}_n"]), float(g[f"{name}_gamma"])
    ret, dn = ops.discount_return_n_step(r, d, n, gamma)
    assert np.array_equal(host(ret), g[f"{name}_ret"])
    assert np.array_equal(host(dn), g[f"{name}_done_n"])
    ret, dn = ops.discount_return_n_step(r, d, n, gamma, do_truncated=True)
    assert np.array_equal(host(ret), g[f"{name}_ret_trunc"])
    assert np.array_equal(host(dn), g[f"{name}_done_n_trunc"])


@pytest.mark.parametrize("T,N,n", [(130, 4096, 5), (9, 33, 3), (50, 1 << 16, 2)])
def test_nstep_vs_oracle(ops, T, N, n):
    rng = np.random.RandomState(n)
    r = rng.randn(T, N).astype(np.float32)
    d = rng.rand(T, N) < 0.05
    for trunc in (False, True):
        ret, dn = ops.discount_return_n_step(dev(r), dev(d), n, 0.997, do_truncated=trunc)
        er, edn = O.discount_return_n_step(r, d, n, 0.997, do_truncated=trunc)
        assert np.array_equal(host(ret), er) and np.array_equal(host(dn), edn)


# -------------------------------------------------------------------------------- normalise
@pytest.mark.parametrize("name", ["cfg", "small"])
def test_normalize_golden(ops, name):
    """Reductions re-associate: rtol 2e-5 / atol 2e-6 (statistics accumulate in f64 here)."""
    g = load_golden("normalize")
    a = dev(g[f"{name}_adv"]).clone()
    ops.normalize_advantage_(a)
    np.testing.assert_allclose(host(a), g[f"{name}_norm_all"], rtol=2e-5, atol=2e-6)
    a = dev(g[f"{name}_adv"]).clone()
    _, stats = ops.normalize_advantage_(a, dev(g[f"{name}_valid"]), return_stats=True)
    np.testing.assert_allclose(host(a), g[f"{name}_norm_valid"], rtol=2e-5, atol=2e-6)
    sel = g[f"{name}_adv"][g[f"{name}_valid"] > 0]
    np.testing.assert_allclose(host(stats), [sel.mean(), sel.std(ddof=1), sel.size], rtol=1e-5)


def test_normalize_large(ops):
    x = torch.randn(128 * (1 << 16), device="cuda") * 3 + 1
    y = x.clone()
    ops.normalize_advantage_(y)
    ref = (x - x.mean()) / x.std()
    torch.testing.assert_close(y, ref, rtol=1e-4, atol=1e-5)


# ----------------------------------------------------------------------------------- losses
def _check_pg(ops, g, name, kind):
    t = lambda k: dev(g[f"{name}_{k}"])  # noqa: E731
    valid = t("valid") if f"{name}_valid" in g else None
    if kind == "ppo":
        pn, v = t("prob_new").requires_grad_(True), t("value").requires_grad_(True)
        loss, sc = ops.ppo_loss(pn, v, t("prob_old"), t("action"), t("adv"), t("ret"), valid,
                                float(g[f"{name}_clip"]), 1.0, 0.01)
    else:
        pn, v = t("prob").requires_grad_(True), t("value").requires_grad_(True)
        loss, sc = ops.a2c_loss(pn, v, t("action"), t("adv"), t("ret"), valid, 0.5, 0.01)
    loss.backward()
    # fp32 tolerance: rtol 1e-5 on the scalars (reduction order), 1e-5 / 1e-8 on gradients
    np.testing.assert_allclose(host(sc), g[f"{name}_scalars"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(host(pn.grad), g[f"{name}_grad_prob"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(host(v.grad), g[f"{name}_grad_value"], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("name", ["ppo_cfg", "ppo_valid", "ppo_a18"])
def test_ppo_loss_golden(ops, name):
    _check_pg(ops, load_golden("losses"), name, "ppo")


@pytest.mark.parametrize("name", ["a2c_cfg", "a2c_valid"])
def test_a2c_loss_golden(ops, name):
    _check_pg(ops, load_golden("losses"), name, "a2c")


@pytest.mark.parametrize("M,A,with_valid", [(8192, 6, False), (8192, 6, True), (1, 2, False),
                                            (300000, 4, True), (255, 18, False)])
def test_ppo_loss_vs_oracle(ops, M, A, with_valid):
    g = torch.Generator().manual_seed(M + A)
    pn = torch.softmax(torch.randn(M, A, generator=g), -1)
    po = torch.softmax(torch.randn(M, A, generator=g) * 0.3 + torch.log(pn), -1)
    a = torch.randint(0, A, (M,), generator=g)
    adv, ret, v = (torch.randn(M, generator=g) for _ in range(3))
    valid = (torch.rand(M, generator=g) > 0.3).float() if with_valid else None
    pn_c, v_c = pn.clone().requires_grad_(True), v.clone().requires_grad_(True)
    ref = O.ppo_loss_torch(pn_c, v_c, po, a, adv, ret, valid, 0.1, 1.0, 0.01)
    ref[0].backward()
    pn_d, v_d = pn.cuda().requires_grad_(True), v.cuda().requires_grad_(True)
    loss, sc = ops.ppo_loss(pn_d, v_d, po.cuda(), a.cuda(), adv.cuda(), ret.cuda(),
                            None if valid is None else valid.cuda(), 0.1, 1.0, 0.01)
    (2.0 * loss).backward()  # also checks the incoming-gradient scaling
    np.testing.assert_allclose(host(sc), [x.item() for x in ref], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(host(pn_d.grad), 2 * pn_c.grad.numpy(), rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(host(v_d.grad), 2 * v_c.grad.numpy(), rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("name", ["dqn", "ddqn", "dqn_mse"])
def test_dqn_loss_golden(ops, name):
    g = load_golden("losses")
    t = lambda k: dev(g[f"{name}_{k}"])  # noqa: E731
    qs = t("qs").requires_grad_(True)
    clip = float(g[f"{name}_clip"])
    loss, td = ops.dqn_loss(qs, t("target_qs"), t("next_qs") if bool(g[f"{name}_double"])
                            else None, t("action"), t("ret"), t("done_n"),
                            t("isw") if f"{name}_isw" in g else None,
                            float(g[f"{name}_disc_n"]), None if clip < 0 else clip)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g[f"{name}_loss"], rtol=1e-5)
    np.testing.assert_allclose(host(td), g[f"{name}_td"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(host(qs.grad), g[f"{name}_grad_qs"], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("name", ["cat", "cat_double", "cat_valid", "cat_peaky"])
def test_cat_dqn_loss_golden(ops, name):
    """HIP CategoricalDQN loss vs the outputs of the reference's own method.  fp32 tolerance:
    1e-5 relative on the loss, 2e-5 on KL / gradients (sums of 51-64 products, logf within
    1 ulp); the greedy next action is an argmax over fp32 sums, so the fixtures avoid ties."""
    from oracle import np_oracle as O
    g = load_golden("catdqn")
    t = lambda k: dev(g[f"{name}_{k}"])  # noqa: E731
    ps = t("ps").requires_grad_(True)
    P = ps.shape[-1]
    v_min, v_max = float(g[f"{name}_vmin"]), float(g[f"{name}_vmax"])
    valid = None
    if not bool(g[f"{name}_mbr"]):
        valid = dev(O.valid_from_done(g[f"{name}_done"]))
    disc_n = float(g[f"{name}_discount"]) ** int(g[f"{name}_n_step"])
    loss, kl = ops.cat_dqn_loss(
        ps, t("target_ps"), t("next_ps") if bool(g[f"{name}_double"]) else None, t("action"),
        t("ret"), t("done_n"), t("isw") if bool(g[f"{name}_pri"]) else None, valid,
        torch.linspace(v_min, v_max, P), v_min, v_max, disc_n)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g[f"{name}_loss"], rtol=1e-5)
    np.testing.assert_allclose(host(kl), g[f"{name}_kl"], rtol=2e-5, atol=1e-6)
    ref = g[f"{name}_grad_ps"]
    np.testing.assert_allclose(host(ps.grad), ref, rtol=2e-5, atol=1e-7 * np.abs(ref).max())


def test_cat_dqn_loss_vs_oracle_config_size(ops):
    """BASELINE-style batch (256 samples, 6 actions, 51 atoms): HIP vs the CPU oracle on the
    same seeded inputs, plus two size-independent properties: every projected target
    distribution keeps its mass (sum_i target_p[i] = 1 when next_z stays inside the grid, so
    with p == target the cross-entropy equals the entropy and KL sits at its floor), and
    the gradient is linear in the IS weights."""
    from oracle import np_oracle as O
    g0 = torch.Generator().manual_seed(5)
    M, A, P = 256, 6, 51
    ps = torch.softmax(torch.randn(M, A, P, generator=g0), -1)
    tps = torch.softmax(torch.randn(M, A, P, generator=g0), -1)
    action = torch.randint(0, A, (M,), generator=g0)
    ret = torch.randn(M, generator=g0)
    done_n = torch.rand(M, generator=g0) < 0.1
    isw = torch.rand(M, generator=g0) + 0.1
    z = torch.linspace(-10, 10, P)
    ps_c = ps.clone().requires_grad_(True)
    loss_c, kl_c = O.cat_dqn_loss_torch(ps_c, tps, None, action, ret, done_n, isw, None, -10,
                                        10, 0.99, 1)
    loss_c.backward()
    ps_d = dev(ps.numpy()).requires_grad_(True)
    loss_d, kl_d = ops.cat_dqn_loss(ps_d, dev(tps.numpy()), None, dev(action.numpy()),
                                    dev(ret.numpy()), dev(done_n.numpy()), dev(isw.numpy()),
                                    None, z, -10, 10, 0.99)
    loss_d.backward()
    np.testing.assert_allclose(loss_d.item(), loss_c.item(), rtol=1e-5)
    np.testing.assert_allclose(host(kl_d), kl_c.numpy(), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(host(ps_d.grad), ps_c.grad.numpy(), rtol=2e-5, atol=1e-8)
    # linearity in the IS weights
    ps_e = dev(ps.numpy()).requires_grad_(True)
    loss_e, _ = ops.cat_dqn_loss(ps_e, dev(tps.numpy()), None, dev(action.numpy()),
                                 dev(ret.numpy()), dev(done_n.numpy()), dev(2 * isw.numpy()),
                                 None, z, -10, 10, 0.99)
    loss_e.backward()
    np.testing.assert_allclose(host(ps_e.grad), 2 * host(ps_d.grad), rtol=1e-6, atol=0)
    np.testing.assert_allclose(loss_e.item(), 2 * loss_d.item(), rtol=1e-6)
    # identity Bellman map (reward 0, discount 1, no terminal): projection is the identity,
    # so a network that already equals its target has KL at the clamp floor
    same = dev(tps.numpy())
    act_greedy = torch.argmax(torch.tensordot(tps, z, dims=1), -1)
    _, kl0 = ops.cat_dqn_loss(same.clone().requires_grad_(True), same, None,
                              dev(act_greedy.numpy()), dev(np.zeros(M, np.float32)),
                              dev(np.zeros(M, bool)), None, None, z, -10, 10, 1.0)
    assert float(kl0.max()) <= 2e-6


# ---------------------------------------------------------------------------------- gathers
def test_gather_tb_exact(ops):
    rng = np.random.RandomState(0)
    T, B = 16, 8
    idx = rng.permutation(T * B)[:50]
    for shape, dt in [((4, 13, 10), np.uint8), ((6,), np.float32), ((), np.int64),
                      ((3,), np.uint8), ((4, 104, 80), np.uint8)]:
        src = rng.randint(0, 255, size=(T, B) + shape).astype(dt)
        out = ops.gather_tb(dev(src), dev(idx))
        assert np.array_equal(host(out), src[idx % T, idx // T])
    t_idx = rng.randint(-1, T, size=40)
    b_idx = rng.randint(0, B, size=40)
    src = rng.randn(T, B, 5).astype(np.float32)
    assert np.array_equal(host(ops.gather_rows(dev(src), dev(t_idx), dev(b_idx))),
                          src[t_idx, b_idx])


def test_gather_tb_full_size_permutation_roundtrip(ops):
    """PPO config shape: gathering all 4 minibatches of a permutation moves every row once."""
    T, B = 128, 256
    obs = torch.randint(0, 256, (T, B, 4, 104, 80), dtype=torch.uint8, device="cuda")
    perm = torch.randperm(T * B, device="cuda")
    mb = T * B // 4
    total = torch.zeros((), dtype=torch.int64, device="cuda")
    for k in range(4):
        idx = perm[k * mb:(k + 1) * mb]
        out = ops.gather_tb(obs, idx)
        ref = obs[idx % T, idx // T]
        assert torch.equal(out, ref)
        total += out.sum(dtype=torch.int64)
    assert total.item() == obs.sum(dtype=torch.int64).item()


@pytest.mark.parametrize("name", ["small", "c2"])
def test_frames_golden_bit_exact(ops, name):
    g = load_golden("frames")
    C = int(g[f"{name}_C"])
    obs = ops.frames_gather(dev(g[f"{name}_frames"]), dev(g[f"{name}_done"]),
                            dev(g[f"{name}_T_idxs"]), dev(g[f"{name}_B_idxs"]), C)
    assert np.array_equal(host(obs), g[f"{name}_obs"])
    seq = ops.frames_gather_seq(dev(g[f"{name}_frames"]), dev(g[f"{name}_done"]),
                                dev(g[f"{name}_seq_T_idxs"]), dev(g[f"{name}_B_idxs"]), C,
                                int(g[f"{name}_seq_T"]))
    assert np.array_equal(host(seq), g[f"{name}_seq_obs"])


def test_frames_atari_shape_vs_oracle(ops):
    """Atari frame shape (104x80), ring with wrap duplicates, dones near the samples."""
    rng = np.random.RandomState(3)
    T, B, C, H, W, n = 500, 16, 4, 104, 80, 128
    frames = rng.randint(0, 256, size=(T + C - 1, B, H, W)).astype(np.uint8)
    frames[:C - 1] = frames[-(C - 1):]
    done = rng.rand(T, B) < 0.05
    T_idxs, B_idxs = rng.randint(0, T, size=n), rng.randint(0, B, size=n)
    obs = ops.frames_gather(dev(frames), dev(done), dev(T_idxs), dev(B_idxs), C)
    assert np.array_equal(host(obs), O.frames_gather(frames, done, T_idxs, B_idxs, C))
    sT = rng.randint(0, T, size=8)
    sT[0] = T - 5
    seq = ops.frames_gather_seq(dev(frames), dev(done), dev(sT), dev(B_idxs[:8]), C, 25)
    assert np.array_equal(host(seq), O.frames_gather_seq(frames, done, sT, B_idxs[:8], C, 25))


@pytest.mark.parametrize("n_step", [1, 3])
def test_frames_gather_pair_is_the_two_single_gathers(ops, n_step):
    """Agent + target observation of a replay batch in one launch == the two extract_observation
    calls of rlpyt/replays/non_sequence/n_step.py:29-42 (oracle), incl. target rows past the wrap."""
    rng = np.random.RandomState(5 + n_step)
    T, B, C, H, W, n = 300, 8, 4, 104, 80, 128
    frames = rng.randint(0, 256, size=(T + C - 1, B, H, W)).astype(np.uint8)
    frames[:C - 1] = frames[-(C - 1):]
    done = rng.rand(T, B) < 0.05
    T_idxs, B_idxs = rng.randint(0, T, size=n), rng.randint(0, B, size=n)
    T_idxs[:3] = [T - 1, T - n_step, 0]
    both = ops.frames_gather_pair(dev(frames), dev(done), dev(T_idxs), dev(B_idxs), C, n_step)
    assert np.array_equal(host(both[0]), O.frames_gather(frames, done, T_idxs, B_idxs, C))
    assert np.array_equal(host(both[1]),
                          O.frames_gather(frames, done, (T_idxs + n_step) % T, B_idxs, C))


def test_extract_sequences_golden(ops):
    g = load_golden("frames")
    out = ops.extract_sequences(dev(g["es_arr"]), dev(g["es_T_idxs"]), dev(g["es_B_idxs"]),
                                int(g["es_seq_T"]))
    assert np.array_equal(host(out), g["es_out"])


# --------------------------------------------------------------------------------- sum tree
def _device_tree_api(ops):
    def make(T, B, ob, of, input_pri, shift):
        return ops.DeviceSumTree(T, B, ob, of, default_value=1.0,
                                 enable_input_priorities=input_pri, input_priority_shift=shift)

    def sample(tree, u):
        Ti, Bi, pri = tree.sample(dev(u))
        return host(Ti), host(Bi), host(pri)

    def root(tree):
        return tree.tree_tensor()[0].item()
    return dict(make_tree=make, sample=sample,
                update=lambda tree, p: tree.update_batch_priorities(dev(p)),
                advance=lambda tree, T, p: tree.advance(T, None if p is None else dev(p)),
                root=root, tree_of=lambda tree: host(tree.tree_tensor()))


@pytest.mark.parametrize("name", ["small", "wrap", "inpri", "dqn1m"])
def test_sumtree_streams_bit_exact(ops, name):
    """Indices, priorities, root sums (and whole trees for the small streams) recorded from
    the reference SumTree, replayed on the HBM tree with the same uniforms."""
    from test_oracle_golden import replay_sumtree_stream
    g = load_golden("sumtree")
    tree = replay_sumtree_stream(g, name, **_device_tree_api(ops))
    full = host(tree.tree_tensor())
    assert full[0] == float(g[f"{name}_final_tree_root"])
    n = len(g[f"{name}_final_leaves_head"])
    assert np.array_equal(full[tree.low_idx:tree.low_idx + n], g[f"{name}_final_leaves_head"])
    assert tree.tree_levels == int(g[f"{name}_levels"]) and tree.low_idx == int(g[f"{name}_low_idx"])


def test_sumtree_known_answer(ops):
    g = load_golden("sumtree")
    t = ops.DeviceSumTree(8, 2, 1, 1, default_value=1)
    t.advance(4)
    assert t.tree_tensor()[0].item() == 4.0 and t.tree_levels == 6 and t.low_idx == 31
    Ti, Bi, p = t.sample(dev(g["kat_u1"]))
    assert host(Ti).tolist() == [2, 2, 2, 2, 1] and host(Bi).tolist() == [0, 0, 0, 0, 1]
    t.update_batch_priorities(dev(np.array([0.5, 2, 3, 0.1, 4])))
    assert t.tree_tensor()[0].item() == 6.5
    Ti, Bi, p = t.sample(dev(g["kat_u2"]))
    assert np.array_equal(host(Ti), g["kat_T2"]) and np.array_equal(host(Bi), g["kat_B2"])
    assert np.array_equal(host(p), g["kat_p2"])
    assert np.array_equal(host(t.tree_tensor()), g["kat_tree"])


def test_sumtree_vs_oracle_random_stream(ops):
    """A longer random stream against the oracle incl. whole-tree equality every op."""
    rng = np.random.RandomState(11)
    T, B = 200, 8
    dt = ops.DeviceSumTree(T, B, 3, 3, default_value=1.0)
    ot = O.SumTree(T, B, 3, 3, default_value=1.0)
    for op in range(300):
        k = int(rng.randint(1, 6))
        dt.advance(k)
        ot.advance(k)
        if ot.tree[0] > 0:
            u = rng.rand(64)
            (oT, oB), op_ = ot.sample_with(u)
            Ti, Bi, p = dt.sample(dev(u))
            assert np.array_equal(host(Ti), oT) and np.array_equal(host(Bi), oB)
            assert np.array_equal(host(p), op_)
            newp = np.abs(rng.randn(64)) ** 0.6
            ot.update_batch_priorities(newp)
            dt.update_batch_priorities(dev(newp))
        assert np.array_equal(host(dt.tree_tensor()), ot.tree), op
        assert dt.t == ot.t


def test_sumtree_sampling_distribution(ops):
    """Property at the 1M-leaf size: sampled frequencies follow the priorities."""
    T, B = 62500, 16
    t = ops.DeviceSumTree(T, B, 1, 3, default_value=1.0)
    for _ in range(50):
        t.advance(1000)
    u = torch.rand(200000, dtype=torch.float64, device="cuda")
    Ti, Bi, p = t.sample(u)
    assert (p > 0).all()
    # valid zone only: rows [3, 50000-1)
    assert int(Ti.min()) >= 3 and int(Ti.max()) < 50000 - 1
    frac = (Ti < 25000).double().mean().item()
    assert abs(frac - (25000 - 3) / (49999 - 3)) < 0.01


# ---------------------------------------------------------------------------- replay buffers
@pytest.mark.parametrize("name", ["pri_n3", "pri_n1", "uni_n2"])
def test_replay_buffer_stream_vs_reference(ops, name):
    """Whole-buffer stream recorded from the reference's PrioritizedReplayFrameBuffer /
    UniformReplayFrameBuffer (appends with ring wraps, n-step returns, frame store
    duplication, sampling with the same np.random seed, priority updates): every sampled
    field must be identical; importance weights within fp32 tolerance (device pow)."""
    from rlpyt_amd.replays.buffers import (PrioritizedReplayFrameBuffer,
                                                UniformReplayFrameBuffer)
    from rlpyt_amd.utils.collections import namedarraytuple
    g = load_golden("replay")
    B, C, H, W, Tring, Tapp, n_app, nb, n_step = (int(x) for x in g[f"{name}_meta"])
    S2B = namedarraytuple("SamplesToBuffer", ["observation", "action", "reward", "done"])
    example = S2B(observation=np.zeros((C, H, W), np.uint8), action=np.int64(0),
                  reward=np.float32(0), done=False)
    kw = dict(example=example, size=Tring * B, B=B, discount=0.99, n_step_return=n_step,
              device="cuda")
    pri = name.startswith("pri")
    alpha = float(g[f"{name}_alpha"])
    buf = (PrioritizedReplayFrameBuffer(alpha=alpha, beta=0.5, default_priority=1., **kw)
           if pri else UniformReplayFrameBuffer(**kw))
    exact_pow = alpha == 1.0
    k_s = 0
    for k in range(n_app):
        buf.append_samples(S2B(*(dev(g[f"{name}_{f}"][k]) for f in
                                 ("obs", "action", "reward", "done"))))
        if k < 2:
            continue
        np.random.seed(int(g[f"{name}_s_seeds"][k_s]))
        batch = buf.sample_batch(nb)
        chk = lambda field, val: np.array_equal(host(val), g[f"{name}_s_{field}"][k_s])  # noqa
        same = (chk("agent_obs", batch.agent_inputs.observation) and
                chk("target_obs", batch.target_inputs.observation) and
                chk("prev_action", batch.agent_inputs.prev_action) and
                chk("prev_reward", batch.agent_inputs.prev_reward) and
                chk("tgt_prev_action", batch.target_inputs.prev_action) and
                chk("action", batch.action) and chk("return_", batch.return_) and
                chk("done", batch.done) and chk("done_n", batch.done_n))
        if exact_pow or not pri or k_s == 0:
            assert same, (name, k)
        elif not same:
            # alpha=0.6: device powf may differ from numpy's by an ulp, after which the
            # streams may legitimately part ways; require agreement up to here.
            assert k_s >= 1
            break
        if pri:
            np.testing.assert_allclose(host(batch.is_weights), g[f"{name}_s_is_weights"][k_s],
                                       rtol=2e-6)
            buf.update_batch_priorities(dev(g[f"{name}_s_new_pri"][k_s]))
        k_s += 1
    if pri and exact_pow:
        assert buf.priority_tree.tree_tensor()[0].item() == float(g[f"{name}_final_root"])


@pytest.mark.parametrize("name", ["pseq", "useq"])
def test_sequence_replay_stream_vs_reference(ops, name):
    """Stream recorded from the reference's (Prioritized|Uniform)SequenceReplayFrameBuffer
    (R2D1 geometry in miniature: rnn_state_interval=4, batch_T=8, n_step=2, input
    priorities with shift 1, alpha=1 so the tree stream is exact): every field identical."""
    from rlpyt_amd.replays.buffers import (PrioritizedSequenceReplayFrameBuffer,
                                            UniformSequenceReplayFrameBuffer)
    from rlpyt_amd.utils.collections import namedarraytuple
    g = load_golden("seq_replay")
    B, C, H, W, Tring, Tapp, n_app, nb, n_step, rsi, bT = (int(x) for x in g[f"{name}_meta"])
    RnnState = namedarraytuple("RnnState", ["h", "c"])
    S2B = namedarraytuple("SamplesToBufferRnn",
                          ["observation", "action", "reward", "done", "prev_rnn_state"])
    Pri = namedarraytuple("PrioritiesSamplesToBuffer", ["priorities", "samples"])
    example = S2B(observation=np.zeros((C, H, W), np.uint8), action=np.int64(0),
                  reward=np.float32(0), done=False,
                  prev_rnn_state=RnnState(np.zeros((1, 3), np.float32),
                                          np.zeros((1, 3), np.float32)))
    kw = dict(example=example, size=Tring * B, B=B, discount=0.99, n_step_return=n_step,
              rnn_state_interval=rsi, batch_T=bT, device="cuda")
    pri = name == "pseq"
    if pri:
        buf = PrioritizedSequenceReplayFrameBuffer(alpha=1.0, beta=0.5, default_priority=1.,
                                                   input_priorities=True,
                                                   input_priority_shift=1, **kw)
        geom = g[f"{name}_tree_geom"]
        assert (buf.priority_tree.T, ) == (int(geom[0]),)
    else:
        buf = UniformSequenceReplayFrameBuffer(**kw)
    k_s = 0
    for k in range(n_app):
        smp = S2B(dev(g[f"{name}_obs"][k]), dev(g[f"{name}_action"][k]),
                  dev(g[f"{name}_reward"][k]), dev(g[f"{name}_done"][k]),
                  RnnState(dev(g[f"{name}_h"][k]), dev(g[f"{name}_c"][k])))
        buf.append_samples(Pri(priorities=g[f"{name}_in_pri"][k], samples=smp) if pri else smp)
        if k < 5:
            continue
        np.random.seed(int(g[f"{name}_s_seeds"][k_s]))
        batch = buf.sample_batch(nb)
        for field, val in [("all_obs", batch.all_observation), ("all_action", batch.all_action),
                           ("all_reward", batch.all_reward), ("return_", batch.return_),
                           ("done", batch.done), ("done_n", batch.done_n),
                           ("h", batch.init_rnn_state.h), ("c", batch.init_rnn_state.c)]:
            assert np.array_equal(host(val), g[f"{name}_s_{field}"][k_s]), (name, k, field)
        if pri:
            np.testing.assert_allclose(host(batch.is_weights), g[f"{name}_s_is_weights"][k_s],
                                       rtol=2e-6)
            buf.update_batch_priorities(dev(g[f"{name}_s_new_pri"][k_s]))
        k_s += 1
    if pri:
        assert buf.priority_tree.tree_tensor()[0].item() == float(g[f"{name}_final_root"])


# ------------------------------------------------------------------------- R2D1 loss, obs RMS
@pytest.mark.parametrize("name", ["r2d1", "r2d1_huber"])
def test_r2d1_loss_golden(ops, name):
    """fp32 tolerance.  The inverse value rescaling h^-1(z) = sign(z)(((sqrt(1+4e(|z|+1+e))-1)
    /(2e))^2 - 1) with e=1e-3 is ill-conditioned: the cancellation in sqrt(.)-1 followed by
    the division by 2e = 0.002 amplifies a single fp32 ulp (6e-8) to ~3e-5 absolute before
    squaring, so device and host results may differ by ~1e-4 absolute on targets of O(10).
    Hence rtol 2e-4 / atol 1e-4 on |TD| and priorities, rtol 2e-4 on the loss; the gradient
    is dL/dq = w * clip(td) / (T*B) with 1/(T*B) = 1e-2 here, so the 1e-4 absolute slack on td
    becomes atol 2e-6 on the gradient."""
    g = load_golden("r2d1_rms")
    t = lambda k: dev(g[f"{name}_{k}"])  # noqa: E731
    qs = t("qs").requires_grad_(True)
    clip = float(g[f"{name}_clip"])
    loss, vtd, pri = ops.r2d1_loss(
        qs, t("target_qs"), t("next_qs") if bool(g[f"{name}_double"]) else None, t("action"),
        t("ret"), t("done_n"), t("valid"), t("isw") if bool(g[f"{name}_pri"]) else None,
        float(g[f"{name}_disc_n"]), None if clip < 0 else clip, float(g[f"{name}_eps"]),
        float(g[f"{name}_eta"]))
    loss.backward()
    np.testing.assert_allclose(loss.item(), g[f"{name}_loss"], rtol=2e-4)
    np.testing.assert_allclose(host(vtd), g[f"{name}_vtd"], rtol=2e-4, atol=1e-4)
    np.testing.assert_allclose(host(pri), g[f"{name}_priorities"], rtol=2e-4, atol=1e-4)
    np.testing.assert_allclose(host(qs.grad), g[f"{name}_grad_qs"], rtol=2e-4, atol=2e-6)


def test_running_mean_std_golden(ops):
    """Three successive updates of the reference RunningMeanStdModel; tolerance rtol 1e-5 /
    atol 1e-6 (batch statistics accumulate in f64 here, Welford in f32 there)."""
    from rlpyt_amd.models.running_mean_std import RunningMeanStdModel
    g = load_golden("r2d1_rms")
    rms = RunningMeanStdModel((17,)).cuda()
    for k in range(3):
        rms.update(dev(g[f"rms_x{k}"]))
        np.testing.assert_allclose(host(rms.mean), g[f"rms_mean{k}"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(host(rms.var), g[f"rms_var{k}"], rtol=1e-5, atol=1e-6)
        assert rms.count.item() == float(g[f"rms_count{k}"])
    out = rms.normalize(dev(g["rms_x0"]))
    np.testing.assert_allclose(host(out), g["rms_norm"], rtol=1e-5, atol=1e-5)


def test_obs_to_nhwc_fused_exact(ops):
    """gather + uint8->f32*(1/255) + NHWC in one kernel == torch's ops on the gathered rows
    (bit-exact: one correctly-rounded fp32 multiply per element)."""
    T, B = 6, 5
    obs = torch.randint(0, 256, (T, B, 4, 104, 80), dtype=torch.uint8, device="cuda")
    idx = torch.randperm(T * B, device="cuda")[:17]
    out = ops.obs_to_nhwc_f32(obs, idx)
    ref = obs[idx % T, idx // T].float().mul_(1. / 255)
    assert out.shape == ref.shape and out.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(out, ref)
    out2 = ops.obs_to_nhwc_f32(obs[2])            # identity map, [B,C,H,W]
    assert torch.equal(out2, obs[2].float().mul_(1. / 255))
    odd = torch.randint(0, 256, (3, 2, 3, 5, 7), dtype=torch.uint8, device="cuda")  # generic path
    i2 = torch.tensor([5, 0, 3], device="cuda")
    assert torch.equal(ops.obs_to_nhwc_f32(odd, i2), odd[i2 % 3, i2 // 3].float().mul_(1. / 255))


def test_model_matches_cpu_port_weights(ops):
    """AtariFfModel on the device (fused NHWC input path) vs the same weights on CPU torch:
    fp32 tolerance rtol 1e-4 / atol 1e-5 (different conv algorithms)."""
    from rlpyt_amd.models.pg.atari_ff_model import AtariFfModel
    torch.manual_seed(0)
    cpu = AtariFfModel((4, 104, 80), 6)
    gpu = AtariFfModel((4, 104, 80), 6)
    gpu.load_state_dict(cpu.state_dict())
    gpu = gpu.cuda().to(memory_format=torch.channels_last)
    x = torch.randint(0, 256, (3, 7, 4, 104, 80), dtype=torch.uint8)
    pc, vc = cpu(x, None, None)
    pg, vg = gpu(x.cuda(), None, None)
    np.testing.assert_allclose(host(pg.detach()), pc.detach().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(host(vg.detach()), vc.detach().numpy(), rtol=1e-4, atol=1e-5)


def test_sumtree_unique_sampling_matches_reference():
    """``DeviceSumTree.sample_unique`` == the reference's ``SumTree.sample(n, unique=True)``
    (rlpyt/replays/sum_tree.py:109-128; golden recorded from the reference class,
    make_golden.py gen_sumtree_unique): same distinct, sorted leaves, same priorities, same
    consumption of the host RNG stream by the re-draw loop, same tree after every update."""
    from rlpyt_amd import ops
    g = load_golden("sumtree_unique")
    T, B, ob, of, adv0, adv1 = (int(x) for x in g["geom"])
    tree = ops.DeviceSumTree(T=T, B=B, off_backward=ob, off_forward=of, default_value=1.)
    tree.advance(adv0)
    np.random.seed(11)
    for k, n in enumerate([6, 12, 5, 20]):
        Ti, Bi, p = tree.sample_unique(n)
        assert np.array_equal(Ti.cpu().numpy(), g[f"T{k}"]) and np.array_equal(Bi.cpu().numpy(), g[f"B{k}"])
        assert np.array_equal(p.cpu().numpy(), g[f"p{k}"])
        tree.update_batch_priorities(torch.from_numpy(g[f"new{k}"]).cuda())
        assert np.array_equal(tree.tree_tensor().cpu().numpy(), g[f"tree{k}"])
        if k == 1:
            tree.advance(adv1)
    assert np.array_equal(np.random.rand(3), g["after"])


def test_canary_guard_bands_catch_out_of_bounds_writes():
    """The guard-band debug mode itself (rlpyt_amd/utils/canary.py): an in-bounds kernel leaves the
    guards alone, one byte written past a buffer is reported with the buffer's description, and an
    out-of-bounds float read sees NaN."""
    from rlpyt_amd.utils import canary
    was_on = canary.enabled()
    canary.enable()
    try:
        x = torch.zeros((5, 7), dtype=torch.float32, device="cuda")
        y = torch.empty_like(x)
        y.copy_(x + 1)
        assert canary.check("self-test, clean") >= 2
        assert canary.stats()["live"] >= 2
        before, after = canary.guards_of(x)              # the 0xFF bands around x
        assert before.numel() == canary.GUARD and after.numel() >= canary.GUARD
        assert torch.isnan(after[:4].view(torch.float32)).all()      # an OOB float read: NaN
        after[3] = 7                                      # ... and an OOB write of one byte
        with pytest.raises(AssertionError, match=r"\(5, 7\) torch.float32.*bytes after"):
            canary.check("self-test, dirty")
        assert canary.check("self-test, repaired") >= 2   # (check() restores the pattern)
    finally:
        if not was_on:
            canary.disable()


@pytest.mark.gpu
def test_is_weights_match_the_float64_expression(ops):
    """``rlpyt_is_weights_f64`` (one launch) against the reference's expression in numpy float64
    (rlpyt/replays/non_sequence/prioritized.py:52-56: ``(1 / (p + eps)) ** beta``, divided by its
    maximum, then float32): beta by value and from a device scalar; batch sizes below and above one
    pass of the workgroup; priorities over many orders of magnitude.  float64 ``pow`` may differ from
    libm's in the last bit, i.e. by 1e-16 -- equal after the float32 cast up to one float32 ulp."""
    rng = np.random.RandomState(3)
    for n, eps, beta in ((128, 1e-6, 0.4), (1, 1e-6, 1.0), (3000, 0., 0.6), (7, 1e-6, 0.0)):
        p = np.exp(rng.uniform(-12, 6, n))
        want = (1. / (p + eps)) ** beta
        want = (want / want.max()).astype(np.float32)
        pd = torch.from_numpy(p).cuda()
        got = ops.is_weights(pd, eps, beta).cpu().numpy()
        got_dev = ops.is_weights(pd, eps, torch.tensor([beta], dtype=torch.float64, device="cuda")).cpu().numpy()
        assert got.dtype == np.float32 and got.shape == want.shape
        np.testing.assert_array_equal(got, got_dev)
        np.testing.assert_allclose(got, want, rtol=1.2e-7, atol=0)
        assert got.max() == 1.0
