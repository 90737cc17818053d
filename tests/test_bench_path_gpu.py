"""The path ``bench.py`` times, pinned AS A WHOLE at its own size: ONE PPO
minibatch of the [T=128, B=256] batch, M = 8192, through the product kernels

    index-mode convs_fwd_fused (conv1 -> conv2 in one pass) -> trunk GEMM x W^T (bf16x6, NT) -> trunk bias/ReLU + heads
    + PPO loss (one kernel) -> trunk input gradient g W (NN) and weight gradient g^T x (TN, split-K
    + fixed-order slot reduction) -> conv2_bwd -> conv1_wgrad -> ClipAdam

against (i) the SAME module through MIOpen convolutions + F.linear + the unfused loss
(``use_fused_conv=False``, ``use_split_gemm=False``, ``RLPYT_TRUNK_FUSION=0``,
``fused_head_loss=False``) and (ii) a float64 torch statement of the model
(rlpyt/models/pg/atari_ff_model.py:40-63) and of ``PPO.loss`` (rlpyt/algos/pg/ppo.py:117-154):
loss scalars and EVERY parameter gradient.  The launch counters prove which kernels produced the
product numbers -- the size-gated ones (split GEMMs, convs_fwd_fused, 32 images per persistent
workgroup) are exactly those that no M = 12 reference-iteration golden selects.

Tolerance (f32 accumulation over M * 475 terms in the conv weight gradients, no defined order on
either side): per parameter, max |g - g64| <= max(2 x the MIOpen path's own error vs float64,
2e-5 * max |g64|); scalars rtol 2e-5."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

T, B, A = 128, 256, 6
M = T * B // 4
CLIP, VC, EC = 0.1, 1.0, 0.01
EPS = 1e-8


def _f64_loss(model64, rows_u8, po, act, adv, ret, chunk=1024):
    """float64 forward + backward in chunks (the loss is a mean over M, so chunk sums add up)."""
    c1, c2 = model64.conv.conv.conv[0], model64.conv.conv.conv[2]
    lin = model64.conv.head.model[0]
    sums = torch.zeros(4, dtype=torch.float64, device="cuda")
    for s in range(0, M, chunk):
        e = min(s + chunk, M)
        x = rows_u8[s:e].double() * (1. / 255)
        h = F.relu(F.conv2d(x, c1.weight, c1.bias, stride=4))
        h = F.relu(F.conv2d(h, c2.weight, c2.bias, stride=2, padding=1)).reshape(e - s, -1)
        h = F.relu(F.linear(h, lin.weight, lin.bias))
        p = F.softmax(model64.pi(h), dim=-1)
        v = model64.value(h).squeeze(-1)
        a = act[s:e, None]
        ratio = (p.gather(1, a).squeeze(1) + EPS) / (po[s:e].gather(1, a).squeeze(1) + EPS)
        surr = torch.min(ratio * adv[s:e], torch.clamp(ratio, 1. - CLIP, 1. + CLIP) * adv[s:e])
        ent = -(p * torch.log(p + EPS)).sum(-1)
        pi_l, v_l, en = -surr.sum() / M, VC * 0.5 * ((v - ret[s:e]) ** 2).sum() / M, ent.sum() / M
        (pi_l + v_l - EC * en).backward()
        sums += torch.stack([pi_l, v_l, en, torch.exp(ent).sum() / M]).detach()
    loss = sums[0] + sums[1] - EC * sums[2]
    return torch.stack([loss, sums[0], sums[1], sums[2], sums[3]]).cpu().numpy()


def test_ppo_minibatch_at_bench_size_product_vs_miopen_vs_f64():
    from rlpyt_amd import _lib
    from rlpyt_amd.agents.base import AgentInputs
    from rlpyt_amd.agents.pg.atari import AtariFfAgent
    from rlpyt_amd.algos.pg.ppo import PPO
    from rlpyt_amd.envs.synthetic import SyntheticPong
    from rlpyt_amd.models.pg.atari_ff_model import AtariFfModel
    from rlpyt_amd.optim import ClipAdam
    torch.manual_seed(7)
    agent = AtariFfAgent()
    agent.initialize(SyntheticPong().spaces)
    agent.to_device(0)
    model = agent.model
    assert model.fused_conv and agent.supports_fused_head_loss
    g = torch.Generator().manual_seed(3)
    # structured images (smooth + noise) so that ReLU patterns and activations are not degenerate
    obs = torch.randint(0, 256, (T, B, 4, 104, 80), dtype=torch.uint8, generator=g).cuda()
    perm = torch.randperm(T * B, generator=g)
    idx = perm[M:2 * M].contiguous().cuda()          # a minibatch of a shuffled epoch
    action = torch.randint(0, A, (T, B), generator=g).cuda()
    adv = torch.randn(T, B, generator=g).cuda()
    ret = torch.randn(T, B, generator=g).cuda()
    # behaviour policy = the current one, perturbed: ratios scatter around 1, on both sides of
    # the clip range
    with torch.no_grad():
        po = torch.empty(T, B, A, device="cuda")
        for t0 in range(0, T, 16):
            pi, _v = model(obs[t0:t0 + 16], None, None)
            po[t0:t0 + 16] = torch.softmax(torch.log(pi) + 0.15 * torch.randn(
                16, B, A, generator=g).cuda(), -1)
    t_i, b_i = idx % T, idx // T
    rows = obs[t_i, b_i]
    mb = dict(po=po[t_i, b_i], act=action[t_i, b_i], adv=adv[t_i, b_i], ret=ret[t_i, b_i])

    # ---- (1) product path, exactly as PPO.optimize_agent issues it --------------------------
    algo = PPO(ratio_clip=CLIP, value_loss_coeff=VC, entropy_loss_coeff=EC)
    algo.agent = agent
    model.zero_grad(set_to_none=True)
    _lib.variant_reset()
    loss, sc = algo.loss(AgentInputs(agent.gather_observation(obs, idx), None, None), action, ret,
                         adv, None, po, flat_idx=idx)
    loss.backward()
    torch.cuda.synchronize()
    ran = {k: v for k, v in _lib.variant_counts().items() if v > 0}
    expected = {"convs_fwd_fused_kernel", "gemm_nt_x6_kernel<128>",
                "ppo_head_loss_kernel<8, 6, true>", "gemm_nt_x6_kernel<256>", "gemm_tn_x6_kernel",
                "gemm_reduce_slots_kernel", "conv2_bwd_x6_kernel", "conv1_wgrad_kernel",
                "head_reduce_finalize_kernel"}
    assert expected <= set(ran), sorted(expected - set(ran))
    # nothing of the alternative paths ran (f32-MFMA conv2 forward, unfused loss, gathers)
    for k in ran:
        assert not k.startswith(("conv1_fwd_kernel", "conv2_fwd", "pg_loss_kernel", "gather_", "obs_to_nhwc")), k
    sc_prod = sc.detach().cpu().numpy()
    g_prod = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    assert all(torch.isfinite(v).all() for v in g_prod.values())

    # ---- (2) the same module through MIOpen + F.linear + unfused loss ------------------------
    os.environ["RLPYT_TRUNK_FUSION"] = "0"
    try:
        model.use_fused_conv = False
        model.use_split_gemm = model.conv.head.use_split_gemm = False
        algo_u = PPO(ratio_clip=CLIP, value_loss_coeff=VC, entropy_loss_coeff=EC,
                     fused_head_loss=False)
        algo_u.agent = agent
        model.zero_grad(set_to_none=True)
        _lib.variant_reset()
        loss_u, sc_u = algo_u.loss(AgentInputs(agent.gather_observation(obs, idx), None, None),
                                   mb["act"], mb["ret"], mb["adv"], None, mb["po"])
        loss_u.backward()
        torch.cuda.synchronize()
        ran_u = {k for k, v in _lib.variant_counts().items() if v > 0}
        assert not (ran_u & expected), ran_u & expected
        assert not any(k.startswith(("conv", "gemm_", "ppo_head_loss")) for k in ran_u), ran_u
        sc_mio = sc_u.detach().cpu().numpy()
        g_mio = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    finally:
        os.environ.pop("RLPYT_TRUNK_FUSION", None)
        model.use_fused_conv = True
        model.use_split_gemm = AtariFfModel.use_split_gemm
        del model.conv.head.use_split_gemm

    # ---- (3) float64 -------------------------------------------------------------------------
    import copy
    m64 = copy.deepcopy(model).double()
    m64.zero_grad(set_to_none=True)
    sc64 = _f64_loss(m64, rows, mb["po"].double(), mb["act"], mb["adv"].double(), mb["ret"].double())
    g64 = {n: p.grad for n, p in m64.named_parameters()}

    np.testing.assert_allclose(sc_prod, sc64, rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(sc_mio, sc64, rtol=2e-5, atol=1e-6)
    report = []
    for n in g64:
        ref = g64[n]
        scale = float(ref.abs().max())
        e_prod = float((g_prod[n].double() - ref).abs().max())
        e_mio = float((g_mio[n].double() - ref).abs().max())
        report.append(f"{n}: product {e_prod / scale:.2e}  miopen {e_mio / scale:.2e} (rel to max |g|)")
        assert e_prod <= max(2. * e_mio, 2e-5 * scale), "\n".join(report)
    print("\n".join(report))

    # ---- (4) ClipAdam on the product gradients == clip_grad_norm_ + torch Adam on the same ----
    ref_params = [p.detach().clone().requires_grad_(True) for p in model.parameters()]
    for rp, p in zip(ref_params, model.parameters()):
        p.grad = g_prod[[n for n, q in model.named_parameters() if q is p][0]].clone()
        rp.grad = p.grad.clone()
    ref_opt = torch.optim.Adam(ref_params, lr=1e-3)
    ref_norm = torch.nn.utils.clip_grad_norm_(ref_params, 1.0)
    ref_opt.step()
    opt = ClipAdam(list(model.parameters()), lr=1e-3)
    _lib.variant_reset()
    norm = opt.clip_and_step(1.0)
    torch.cuda.synchronize()
    ran_o = {k for k, v in _lib.variant_counts().items() if v > 0}
    assert {"clip_adam_norm_kernel", "clip_adam_apply_kernel"} <= ran_o, ran_o
    np.testing.assert_allclose(float(norm), float(ref_norm), rtol=1e-5)
    for p, rp in zip(model.parameters(), ref_params):
        # one Adam step moves a parameter by <= lr; agreement to 1e-3 of that
        assert float((p.detach() - rp.detach()).abs().max()) <= 1e-6
