"""Pins the CPU oracle (oracle/np_oracle.py) against vectors produced by the REAL
reference (tests/golden/make_golden.py, run against /root/reference)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import np_oracle as O

SCAN_CASES = ["cfg", "nodone", "dense", "t1", "t2", "kat"]


@pytest.mark.parametrize("name", SCAN_CASES)
def test_scans_bit_exact(name):
    g = load_golden("scans")
    r, v, d, bv = (g[f"{name}_{k}"] for k in ("reward", "value", "done", "bv"))
    gamma, lam = float(g[f"{name}_gamma"]), float(g[f"{name}_lambda"])
    adv, ret = O.generalized_advantage_estimation(r, v, d, bv, gamma, lam)
    assert np.array_equal(adv, g[f"{name}_adv"])
    assert np.array_equal(ret, g[f"{name}_ret"])
    assert np.array_equal(O.discount_return(r, d, bv, gamma), g[f"{name}_disc"])
    assert np.array_equal(O.valid_from_done(d), g[f"{name}_valid"])


def test_scan_known_answers():
    """The hand-checkable mini of SURVEY.md section 8c (gamma=.9, lambda=.8)."""
    g = load_golden("scans")
    np.testing.assert_allclose(g["kat_adv"], [[0.536, 0.1647585], [-0.2, 0.7427201],
                                              [2.656, -1.274], [2.3, -0.7]], rtol=1e-6)
    np.testing.assert_allclose(g["kat_disc"], [[1.0, 0.09], [0.0, 0.1], [4.07, -1.0],
                                               [2.3, 0.0]], rtol=1e-6, atol=1e-7)
    assert np.array_equal(g["kat_valid"], [[1, 1], [1, 1], [0, 1], [0, 1]])


def test_scan_1d():
    g = load_golden("scans")
    adv, ret = O.generalized_advantage_estimation(g["oned_reward"], g["oned_value"],
                                                  g["oned_done"], g["oned_bv"], 0.99, 0.9)
    assert np.array_equal(adv, g["oned_adv"]) and np.array_equal(ret, g["oned_ret"])


@pytest.mark.parametrize("name", ["r2d1", "n3", "n1", "n2"])
def test_nstep_bit_exact(name):
    g = load_golden("nstep")
    r, d, n, gamma = g[f"{name}_reward"], g[f"{name}_done"], int(g[f"{name}_n"]), float(
        g[f"{name}_gamma"])
    ret, dn = O.discount_return_n_step(r, d, n, gamma)
    assert np.array_equal(ret, g[f"{name}_ret"]) and np.array_equal(dn, g[f"{name}_done_n"])
    ret, dn = O.discount_return_n_step(r, d, n, gamma, do_truncated=True)
    assert np.array_equal(ret, g[f"{name}_ret_trunc"])
    assert np.array_equal(dn, g[f"{name}_done_n_trunc"])


@pytest.mark.parametrize("name", ["cfg", "small"])
def test_normalize(name):
    g = load_golden("normalize")
    out, _, _ = O.normalize_advantage(g[f"{name}_adv"])
    np.testing.assert_allclose(out, g[f"{name}_norm_all"], rtol=2e-5, atol=2e-6)
    out, _, _ = O.normalize_advantage(g[f"{name}_adv"], g[f"{name}_valid"])
    np.testing.assert_allclose(out, g[f"{name}_norm_valid"], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("name", ["ppo_cfg", "ppo_valid", "ppo_a18"])
def test_ppo_loss(name):
    g = load_golden("losses")
    t = lambda k: torch.from_numpy(g[f"{name}_{k}"])  # noqa: E731
    pn, v = t("prob_new").requires_grad_(True), t("value").requires_grad_(True)
    valid = t("valid") if f"{name}_valid" in g else None
    res = O.ppo_loss_torch(pn, v, t("prob_old"), t("action"), t("adv"), t("ret"), valid,
                           float(g[f"{name}_clip"]), 1.0, 0.01)
    res[0].backward()
    np.testing.assert_allclose([x.item() for x in res], g[f"{name}_scalars"], rtol=1e-6)
    np.testing.assert_allclose(pn.grad.numpy(), g[f"{name}_grad_prob"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(v.grad.numpy(), g[f"{name}_grad_value"], rtol=1e-6, atol=1e-9)


def test_categorical_methods_match_the_reference():
    """Every tensor-level method of ``rlpyt_amd.distributions.categorical.Categorical`` against the
    reference class's outputs (rlpyt/distributions/categorical.py:17-43, base.py:57-66; incl. rows
    with exact zeros / ones where EPS decides, and the masked means)."""
    from rlpyt_amd.distributions.categorical import Categorical, DistInfo
    g = load_golden("categorical")
    t = lambda k: torch.from_numpy(g[k])  # noqa: E731
    d = Categorical(dim=g["p_old"].shape[-1])
    o, n, idx, valid = DistInfo(prob=t("p_old")), DistInfo(prob=t("p_new")), t("idx"), t("valid")
    eq = lambda a, k: np.testing.assert_allclose(  # noqa: E731
        a.numpy() if isinstance(a, torch.Tensor) else a, g[k], rtol=1e-6, atol=1e-7)
    eq(d.kl(o, n), "kl")
    eq(d.mean_kl(o, n).item(), "mean_kl")
    eq(d.mean_kl(o, n, valid).item(), "mean_kl_valid")
    eq(d.entropy(n), "entropy")
    eq(d.perplexity(n), "perplexity")
    eq(d.mean_entropy(n, valid).item(), "mean_entropy_valid")
    eq(d.mean_perplexity(n, valid).item(), "mean_perplexity_valid")
    eq(d.log_likelihood(idx, n), "log_likelihood")
    eq(d.likelihood_ratio(idx, o, n), "likelihood_ratio")
    assert np.array_equal(d.to_onehot(idx).numpy(), g["onehot"])
    # (the reference's from_onehot raises -- keyword typo at rlpyt/distributions/discrete.py:25; here it
    #  is the inverse of to_onehot)
    assert torch.equal(d.from_onehot(d.to_onehot(idx)), idx)


@pytest.mark.parametrize("name", ["a2c_cfg", "a2c_valid"])
def test_a2c_loss(name):
    g = load_golden("losses")
    t = lambda k: torch.from_numpy(g[f"{name}_{k}"])  # noqa: E731
    pn, v = t("prob").requires_grad_(True), t("value").requires_grad_(True)
    valid = t("valid") if f"{name}_valid" in g else None
    res = O.a2c_loss_torch(pn, v, t("action"), t("adv"), t("ret"), valid, 0.5, 0.01)
    res[0].backward()
    np.testing.assert_allclose([x.item() for x in res], g[f"{name}_scalars"], rtol=1e-6)
    np.testing.assert_allclose(pn.grad.numpy(), g[f"{name}_grad_prob"], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("name", ["dqn", "ddqn", "dqn_mse"])
def test_dqn_loss(name):
    g = load_golden("losses")
    t = lambda k: torch.from_numpy(g[f"{name}_{k}"])  # noqa: E731
    qs = t("qs").requires_grad_(True)
    clip = float(g[f"{name}_clip"])
    loss, td = O.dqn_loss_torch(qs, t("target_qs"), t("next_qs") if bool(g[f"{name}_double"])
                                else None, t("action"), t("ret"), t("done_n"),
                                t("isw") if f"{name}_isw" in g else None, 0.99, 3,
                                None if clip < 0 else clip)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g[f"{name}_loss"], rtol=1e-6)
    np.testing.assert_allclose(td.numpy(), g[f"{name}_td"], rtol=1e-6)
    np.testing.assert_allclose(qs.grad.numpy(), g[f"{name}_grad_qs"], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("name", ["cat", "cat_double", "cat_valid", "cat_peaky"])
def test_cat_dqn_loss(name):
    """Oracle restatement of CategoricalDQN.loss vs the reference method's own outputs."""
    g = load_golden("catdqn")
    t = lambda k: torch.from_numpy(g[f"{name}_{k}"])  # noqa: E731
    ps = t("ps").requires_grad_(True)
    loss, kl = O.cat_dqn_loss_torch(
        ps, t("target_ps"), t("next_ps") if bool(g[f"{name}_double"]) else None, t("action"),
        t("ret"), t("done_n"), t("isw") if bool(g[f"{name}_pri"]) else None,
        None if bool(g[f"{name}_mbr"]) else t("done"), float(g[f"{name}_vmin"]),
        float(g[f"{name}_vmax"]), float(g[f"{name}_discount"]), int(g[f"{name}_n_step"]))
    loss.backward()
    np.testing.assert_allclose(loss.item(), g[f"{name}_loss"], rtol=1e-6)
    np.testing.assert_allclose(kl.numpy(), g[f"{name}_kl"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(ps.grad.numpy(), g[f"{name}_grad_ps"], rtol=1e-6, atol=1e-9)


def replay_sumtree_stream(g, name, make_tree, sample, update, advance, root, tree_of=None):
    """Shared driver: replays a recorded reference stream against an implementation."""
    T, B = int(g[f"{name}_T"]), int(g[f"{name}_B"])
    adv_T, n_ops = int(g[f"{name}_adv_T"]), len(g[f"{name}_root"])
    input_pri = bool(g[f"{name}_input_pri"])
    tree = make_tree(T, B, int(g[f"{name}_ob"]), int(g[f"{name}_of"]), input_pri,
                     int(g[f"{name}_shift"]))
    k = 0
    n_rec = len(g[f"{name}_uniforms"]) if f"{name}_uniforms" in g else 0
    for op in range(n_ops):
        advance(tree, adv_T, g[f"{name}_adv_pri"][op] if input_pri else None)
        expect_root = g[f"{name}_root"][op]
        if k < n_rec and not (expect_root <= 0 and False):
            # ops with an empty tree were recorded without a sample
            pass
        if f"{name}_trees" in g and tree_of is not None and expect_root <= 0:
            assert np.array_equal(tree_of(tree), g[f"{name}_trees"][op])
        if root(tree) <= 0:
            assert expect_root == root(tree)
            continue
        Ti, Bi, pri = sample(tree, g[f"{name}_uniforms"][k])
        assert np.array_equal(Ti, g[f"{name}_T_idxs"][k]), (name, op)
        assert np.array_equal(Bi, g[f"{name}_B_idxs"][k]), (name, op)
        assert np.array_equal(pri, g[f"{name}_pri"][k]), (name, op)
        update(tree, g[f"{name}_new_pri"][k])
        assert root(tree) == expect_root, (name, op)
        if f"{name}_trees" in g and tree_of is not None:
            assert np.array_equal(tree_of(tree), g[f"{name}_trees"][op]), (name, op)
        k += 1
    assert k == n_rec
    return tree


def _oracle_tree_api():
    def make(T, B, ob, of, input_pri, shift):
        return O.SumTree(T, B, ob, of, default_value=1.0, enable_input_priorities=input_pri,
                         input_priority_shift=shift)

    def sample(tree, u):
        (Ti, Bi), pri = tree.sample_with(u)
        return Ti, Bi, pri
    return dict(make_tree=make, sample=sample,
                update=lambda tree, p: tree.update_batch_priorities(p),
                advance=lambda tree, T, p: tree.advance(T, priorities=p),
                root=lambda tree: tree.tree[0], tree_of=lambda tree: tree.tree)


@pytest.mark.parametrize("name", ["small", "wrap", "inpri", "dqn1m"])
def test_sumtree_streams(name):
    g = load_golden("sumtree")
    tree = replay_sumtree_stream(g, name, **_oracle_tree_api())
    assert tree.tree[0] == float(g[f"{name}_final_tree_root"])
    n = len(g[f"{name}_final_leaves_head"])
    assert np.array_equal(tree.tree[tree.low_idx:tree.low_idx + n], g[f"{name}_final_leaves_head"])
    assert tree.tree_levels == int(g[f"{name}_levels"])


def test_sumtree_known_answer():
    g = load_golden("sumtree")
    t = O.SumTree(8, 2, 1, 1, default_value=1)
    t.advance(4)
    assert t.tree[0] == 4.0 and t.tree_levels == 6 and t.low_idx == 31
    (Ti, Bi), p = t.sample_with(g["kat_u1"])
    assert list(Ti) == [2, 2, 2, 2, 1] and list(Bi) == [0, 0, 0, 0, 1]
    t.update_batch_priorities(np.array([0.5, 2, 3, 0.1, 4]))
    assert t.tree[0] == 6.5 == float(g["kat_root"])
    (Ti, Bi), p = t.sample_with(g["kat_u2"])
    assert np.array_equal(Ti, g["kat_T2"]) and np.array_equal(Bi, g["kat_B2"])
    assert np.array_equal(p, g["kat_p2"]) and np.array_equal(t.tree, g["kat_tree"])


@pytest.mark.parametrize("name", ["small", "c2"])
def test_frames(name):
    g = load_golden("frames")
    C = int(g[f"{name}_C"])
    obs = O.frames_gather(g[f"{name}_frames"], g[f"{name}_done"], g[f"{name}_T_idxs"],
                          g[f"{name}_B_idxs"], C)
    assert np.array_equal(obs, g[f"{name}_obs"])
    seq = O.frames_gather_seq(g[f"{name}_frames"], g[f"{name}_done"], g[f"{name}_seq_T_idxs"],
                              g[f"{name}_B_idxs"], C, int(g[f"{name}_seq_T"]))
    assert np.array_equal(seq, g[f"{name}_seq_obs"])


def test_extract_sequences():
    g = load_golden("frames")
    out = O.extract_sequences(g["es_arr"], g["es_T_idxs"], g["es_B_idxs"], int(g["es_seq_T"]))
    assert np.array_equal(out, g["es_out"])


def test_cpu_port_update_matches_reference_ppo_iteration():
    """The CPU port behind bench.py's ``cpu_baseline`` (oracle/ppo_cpu_port.py) is pinned too: its
    ``optimize`` on the fixed batch of tests/golden/algos.npz reproduces the diagnostics and the
    parameters of the reference ``PPO.optimize_agent`` run (first iteration: lr factor 1)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import algo_cases as C
    from oracle.ppo_cpu_port import AtariFfModelCpu, PpoCpuPort
    g = load_golden("algos")
    name, _algo, kw, _mbr = C.CASES[0]
    assert name == "ppo"
    inp = C.batch_inputs()
    port = PpoCpuPort.__new__(PpoCpuPort)
    port.T, port.B, port.A = C.T, C.B, C.A
    torch.manual_seed(C.INIT_SEED)
    port.model = AtariFfModelCpu((4, 104, 80), C.A)
    port.opt = torch.optim.Adam(port.model.parameters(), lr=kw["learning_rate"])
    port.hp = dict(discount=kw["discount"], c_v=kw["value_loss_coeff"],
                   c_e=kw["entropy_loss_coeff"], clip_grad_norm=kw["clip_grad_norm"],
                   lam=kw["gae_lambda"], minibatches=kw["minibatches"], epochs=kw["epochs"],
                   ratio_clip=kw["ratio_clip"])
    port.buf = dict(observation=inp["observation"].numpy(), action=inp["all_action"][1:].numpy(),
                    reward=inp["all_reward"][1:].numpy(), done=inp["done"].numpy(),
                    prob=g["ppo_old_prob"], value=g["ppo_old_value"])
    # same architecture, same seed => the port's forward is the recorded behaviour policy
    with torch.no_grad():
        pi, v = port.model(inp["observation"])
    np.testing.assert_allclose(pi.numpy(), g["ppo_old_prob"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(v.numpy(), g["ppo_old_value"], rtol=1e-6, atol=1e-7)
    np.random.seed(C.SHUFFLE_SEED)
    infos = np.array(port.optimize(torch.from_numpy(g["ppo_bootstrap_value"])))
    for k, f in enumerate(("loss", "gradNorm", "entropy", "perplexity")):
        np.testing.assert_allclose(infos[:, k], g[f"ppo_itr0_{f}"], rtol=1e-5, atol=1e-6,
                                   err_msg=f)
    abs_sums = [p.detach().double().abs().sum().item() for p in port.model.parameters()]
    np.testing.assert_allclose(abs_sums, g["ppo_itr0_param_abs_sums"], rtol=1e-6)
