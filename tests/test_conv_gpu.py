"""GPU parity of the hand-written AtariFfModel conv stack (csrc/conv.hip: fp32 MFMA, and fp32
contractions issued as exact / 2^-24-accurate (2^-27 rms) bf16 splits) against a float64 torch reference of
the same ops (rlpyt/models/pg/atari_ff_model.py:50-55 with rlpyt/models/conv2d.py geometry
4->16 k8 s4 p0, 16->32 k4 s2 p1).

Tolerance (floating point, f32 accumulation chains of length 256 in the forward, up to M*475 in
the weight gradients, different summation order than the reference): max |err| <= 2e-5 *
max |ref| (+1e-6), stated per test; test_bf16_split_kernels_are_f32_accurate pins the bf16-split
kernels to torch's own f32 error level on wide-range operands."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from rlpyt_amd import ops as _ops
    assert torch.cuda.is_available()
    return _ops


def _params(seed=0, dtype=torch.float64):
    g = torch.Generator().manual_seed(seed)
    w1 = (torch.rand(16, 4, 8, 8, generator=g, dtype=dtype) - 0.5) * 0.25   # asymmetric
    b1 = (torch.rand(16, generator=g, dtype=dtype) - 0.5) * 0.2
    w2 = (torch.rand(32, 16, 4, 4, generator=g, dtype=dtype) - 0.5) * 0.25
    b2 = (torch.rand(32, generator=g, dtype=dtype) - 0.5) * 0.2
    return w1, b1, w2, b2


def _ref_stack(obs_u8, params):
    """float64 reference; returns (y1 [M,475,16], y2 [M,3456]) with grad graph on params."""
    w1, b1, w2, b2 = params
    x = obs_u8.double() * (1. / 255)
    a1 = F.relu(F.conv2d(x, w1, b1, stride=4))
    a2 = F.relu(F.conv2d(a1, w2, b2, stride=2, padding=1))
    return a1, a1.permute(0, 2, 3, 1).reshape(x.shape[0], 475, 16), a2.reshape(x.shape[0], -1)


def _close(actual, desired, rel=2e-5, what=""):
    a = actual.detach().double().cpu().numpy()
    d = desired.detach().double().cpu().numpy()
    assert a.shape == d.shape, (what, a.shape, d.shape)
    err = np.abs(a - d).max() if a.size else 0.
    tol = rel * max(np.abs(d).max() if d.size else 0., 1e-30) + 1e-6
    assert err <= tol, f"{what}: max err {err:.3e} > tol {tol:.3e} (max ref {np.abs(d).max():.3e})"


# 1100: five images per persistent workgroup (256 CUs) -- the steady-state iterations of the
# software pipelines (register prefetch two images ahead, hand-counted vmcnt waits, LDS double
# buffers) only exist from the third image of a workgroup on
@pytest.mark.parametrize("M", [1, 7, 200, 300, 1100])
def test_conv_forward_kernels(ops, M):
    from rlpyt_amd._lib import check, lib, ptr, stream
    g = torch.Generator().manual_seed(M)
    obs = torch.randint(0, 256, (M, 4, 104, 80), dtype=torch.uint8, generator=g)
    p64 = _params(1)
    _a1, y1_ref, y2_ref = _ref_stack(obs, p64)
    w1, b1, w2, b2 = (t.float().cuda().contiguous() for t in p64)
    obs_d = obs.cuda()
    y1 = torch.full((M, 475, 16), float("nan"), device="cuda")
    y2 = torch.full((M, 3456), float("nan"), device="cuda")
    check(lib.rlpyt_atari_conv1_fwd_f32(ptr(obs_d), None, 1, M, M, ptr(w1), ptr(b1), 1. / 255,
                                        ptr(y1), stream()), "conv1")
    _close(y1, y1_ref, what="conv1 fwd")
    # conv2 on the float64 reference's y1 (isolates conv2 from conv1's rounding)
    y1_in = y1_ref.float().cuda().contiguous()
    mask = torch.full((M, 128), -1, dtype=torch.int32, device="cuda")
    check(lib.rlpyt_atari_conv2_fwd_f32(ptr(y1_in), M, ptr(w2), ptr(b2), ptr(y2), ptr(mask), stream()),
          "conv2")
    _close(y2, y2_ref, what="conv2 fwd")
    assert not torch.isnan(y1).any() and not torch.isnan(y2).any()   # every element written
    assert torch.equal(mask, _sign_mask(y2))      # the sign bits of the y2 the kernel itself wrote


# 257: one workgroup with two images, 255 with one; 1100: the steady state of the two-role pipeline
# (five images per workgroup); 300 with a minibatch gather out of a [T, B] batch
@pytest.mark.parametrize("M,gather", [(257, False), (300, True), (1100, False), (2100, True)])
def test_convs_forward_fused_equals_the_two_launches(ops, M, gather):
    """``rlpyt_atari_convs_fwd_f32`` (conv1 -> conv2 in one pass, y1 through LDS; rlpyt/models/pg/
    atari_ff_model.py:50-51 at update sizes): y1, y2 and the sign mask are BIT-identical to
    ``rlpyt_atari_conv1_fwd_f32`` + ``rlpyt_atari_conv2_fwd_f32`` (same arithmetic statement for
    statement), every element is written, and y1 / y2 are the float64 convolutions to f32 accuracy."""
    from rlpyt_amd import _lib
    from rlpyt_amd._lib import check, lib, ptr, stream
    g = torch.Generator().manual_seed(M)
    p64 = _params(3)
    w1, b1, w2, b2 = (t.float().cuda().contiguous() for t in p64)
    if gather:
        T, B = 24, 96
        obs = torch.randint(0, 256, (T, B, 4, 104, 80), dtype=torch.uint8, generator=g)
        idx = torch.randperm(T * B, generator=g)[:M]
        rows = obs.view(T * B, 4, 104, 80)[(idx % T) * B + idx // T]
        obs_d, idx_d = obs.cuda(), idx.cuda()
        args = (ptr(obs_d), ptr(idx_d), T, B, M)
    else:
        rows = obs = torch.randint(0, 256, (M, 4, 104, 80), dtype=torch.uint8, generator=g)
        obs_d = obs.cuda()
        args = (ptr(obs_d), None, 1, M, M)
    out = {}
    for tag in ("two", "fused"):
        y1 = torch.full((M, 475, 16), float("nan"), device="cuda")
        y2 = torch.full((M, 3456), float("nan"), device="cuda")
        mask = torch.full((M, 128), -1, dtype=torch.int32, device="cuda")
        if tag == "two":
            check(lib.rlpyt_atari_conv1_fwd_f32(*args, ptr(w1), ptr(b1), 1. / 255, ptr(y1), stream()), "conv1")
            check(lib.rlpyt_atari_conv2_fwd_f32(ptr(y1), M, ptr(w2), ptr(b2), ptr(y2), ptr(mask), stream()),
                  "conv2")
        else:
            check(lib.rlpyt_atari_convs_fwd_f32(*args, ptr(w1), ptr(b1), ptr(w2), ptr(b2), 1. / 255, ptr(y1),
                                                ptr(y2), ptr(mask), stream()), "convs")
            assert _lib.last_variant().startswith("convs_fwd_fused_kernel"), _lib.last_variant()
        torch.cuda.synchronize()
        out[tag] = (y1, y2, mask)
    for a, b, what in zip(out["two"], out["fused"], ("y1", "y2", "mask")):
        assert not torch.isnan(b.float()).any(), what
        assert torch.equal(a, b), (what, int((a != b).sum()))
    _a1, y1_ref, y2_ref = _ref_stack(rows, p64)
    _close(out["fused"][0], y1_ref, what="fused y1")
    _close(out["fused"][1], y2_ref, rel=4e-5, what="fused y2")
    assert torch.equal(out["fused"][2], _sign_mask(out["fused"][1]))


def test_convs_forward_fused_small_batches_take_the_two_launches(ops):
    """At most one image per CU: the entry point forwards to the latency-tuned pair."""
    from rlpyt_amd import _lib
    from rlpyt_amd._lib import check, lib, ptr, stream
    M = 5
    obs = torch.randint(0, 256, (M, 4, 104, 80), dtype=torch.uint8).cuda()
    w1, b1, w2, b2 = (t.float().cuda().contiguous() for t in _params(1))
    y1, y2 = torch.empty((M, 475, 16), device="cuda"), torch.empty((M, 3456), device="cuda")
    mask = torch.empty((M, 128), dtype=torch.int32, device="cuda")
    _lib.variant_reset()
    check(lib.rlpyt_atari_convs_fwd_f32(ptr(obs), None, 1, M, M, ptr(w1), ptr(b1), ptr(w2), ptr(b2), 1. / 255,
                                        ptr(y1), ptr(y2), ptr(mask), stream()), "convs")
    ran = {k for k, v in _lib.variant_counts().items() if v > 0}
    assert {"conv1_fwd_kernel", "conv2_fwd_kernel<2>", "relu_mask_kernel"} <= ran, ran
    assert not any(k.startswith("convs_fwd_fused") for k in ran)
    _a1, y1_ref, y2_ref = _ref_stack(obs.cpu(), _params(1))
    _close(y1, y1_ref, what="y1")
    _close(y2, y2_ref, rel=4e-5, what="y2")


def test_conv_identity_weights_asymmetric(ops):
    """A=I-style check with asymmetric data: w1 picks exactly one input pixel per channel, so
    the output must be that pixel / 255 -- catches transposed row/col or swapped ky/kx maps."""
    M = 3
    g = torch.Generator().manual_seed(5)
    obs = torch.randint(0, 256, (M, 4, 104, 80), dtype=torch.uint8, generator=g)
    w1 = torch.zeros(16, 4, 8, 8)
    for co in range(16):
        w1[co, co % 4, (3 * co) % 8, (5 * co + 1) % 8] = 1.
    from rlpyt_amd._lib import check, lib, ptr, stream
    y1 = torch.empty((M, 475, 16), device="cuda")
    b1 = torch.zeros(16, device="cuda")
    obs_d, w1_d = obs.cuda(), w1.cuda()      # keep the device copies alive across the launch
    check(lib.rlpyt_atari_conv1_fwd_f32(ptr(obs_d), None, 1, M, M, ptr(w1_d), ptr(b1),
                                        1. / 255, ptr(y1), stream()), "conv1")
    y1 = y1.cpu().reshape(M, 25, 19, 16)
    for co in range(16):
        c, ky, kx = co % 4, (3 * co) % 8, (5 * co + 1) % 8
        exp = obs[:, c, ky:ky + 97:4, kx:kx + 73:4].float() / 255
        np.testing.assert_allclose(y1[..., co].numpy(), exp.numpy(), rtol=1e-6, atol=1e-7)


def _sign_mask(y2):
    """uint32 [M, 32, 4] sign mask of y2 [M, 3456] as the conv2 forward kernels write it (bit j of
    word w = y2[m, co, 32 w + j] > 0; positions 108..127 zero), as int32 [M, 128]."""
    M = y2.shape[0]
    pos = (y2.reshape(M, 32, 108) > 0)
    pad = torch.zeros((M, 32, 128), dtype=torch.bool, device=y2.device)
    pad[:, :, :108] = pos
    bits = pad.reshape(M, 32, 4, 32).to(torch.int64)
    w = (bits << torch.arange(32, device=y2.device)).sum(-1)          # [M, 32, 4] in [0, 2^32)
    w = torch.where(w >= 2 ** 31, w - 2 ** 32, w)
    return w.to(torch.int32).reshape(M, 128)


@pytest.mark.parametrize("M", [1, 5, 700, 1100])
def test_conv_backward_kernels(ops, M):
    from rlpyt_amd._lib import check, lib, ptr, stream
    g = torch.Generator().manual_seed(100 + M)
    obs = torch.randint(0, 256, (M, 4, 104, 80), dtype=torch.uint8, generator=g)
    p64 = [t.requires_grad_(True) for t in _params(2)]
    a1, y1_ref, y2_ref = _ref_stack(obs, p64)
    a1.retain_grad()
    g2 = torch.randn(M, 3456, generator=g, dtype=torch.float64)
    (y2_ref * g2).sum().backward()
    # reference dL/d(conv1 pre-activation) = grad wrt a1 masked by relu
    dy1_ref = (a1.grad * (a1 > 0)).permute(0, 2, 3, 1).reshape(M, 475, 16)
    w2 = p64[2].detach().float().cuda().contiguous()
    y1 = y1_ref.detach().float().cuda().contiguous()
    y2 = y2_ref.detach().float().cuda().contiguous()
    g2d = g2.float().cuda().contiguous()
    obs_d = obs.cuda()
    ws = torch.empty(lib.rlpyt_atari_conv_wgrad_workspace_bytes(), dtype=torch.uint8, device="cuda")
    # fused conv2 backward on the bf16 pipe: dgrad + both ReLU masks + weight / bias gradients in one
    # pass; conv2's ReLU mask arrives as the sign bits of y2 (checked against the forward kernels'
    # own mask in test_conv_forward_kernels / test_conv2_sign_mask_*)
    mask = _sign_mask(y2)
    dy1x = torch.full((M, 475, 16), float("nan"), device="cuda")
    dw2x = torch.full((32, 16, 4, 4), float("nan"), device="cuda")
    db2x = torch.full((32,), float("nan"), device="cuda")
    check(lib.rlpyt_atari_conv2_bwd_x6_f32(ptr(g2d), ptr(mask), ptr(y1), M, ptr(w2), ptr(dy1x), ptr(ws),
                                           ptr(dw2x), ptr(db2x), stream()), "conv2 bwd x6")
    _close(dy1x, dy1_ref, what="bf16x6 conv2 dgrad")
    _close(dw2x, p64[2].grad, what="bf16x6 conv2 wgrad")
    _close(db2x, p64[3].grad, what="bf16x6 conv2 bias grad")
    assert not torch.isnan(dy1x).any()
    dw1 = torch.full((16, 4, 8, 8), float("nan"), device="cuda")
    db1 = torch.full((16,), float("nan"), device="cuda")
    dy1_in = dy1_ref.float().cuda().contiguous()
    check(lib.rlpyt_atari_conv1_wgrad_f32(ptr(obs_d), None, 1, M, M, ptr(dy1_in), 1. / 255,
                                          ptr(ws), ptr(dw1), ptr(db1), stream()), "wgrad1")
    _close(dw1, p64[0].grad, what="conv1 wgrad")
    _close(db1, p64[1].grad, what="conv1 bias grad")


def test_conv_stack_autograd_with_gather(ops):
    """ops.atari_conv_stack on a [T,B] batch with the PPO minibatch index map
    idx -> (idx % T, idx // T), forward + backward, against the float64 reference on the
    explicitly gathered rows."""
    T, B, M = 6, 5, 17
    g = torch.Generator().manual_seed(9)
    obs = torch.randint(0, 256, (T, B, 4, 104, 80), dtype=torch.uint8, generator=g)
    idx = torch.randperm(T * B, generator=g)[:M]
    rows = obs[idx % T, idx // T]
    p64 = [t.requires_grad_(True) for t in _params(3)]
    _a1, _y1, y2_ref = _ref_stack(rows, p64)
    g2 = torch.randn(M, 3456, generator=g, dtype=torch.float64)
    (y2_ref * g2).sum().backward()
    p32 = [t.detach().float().cuda().requires_grad_(True) for t in p64]
    y2 = ops.atari_conv_stack(obs.cuda(), idx.cuda(), *p32)
    _close(y2, y2_ref, what="stack fwd")
    (y2 * g2.float().cuda()).sum().backward()
    for name, a, b in zip(["w1", "b1", "w2", "b2"], p32, p64):
        _close(a.grad, b.grad, rel=3e-5, what=f"stack grad {name}")
    # run-to-run deterministic (fixed-order partial reduction, no atomics)
    p32b = [t.detach().clone().requires_grad_(True) for t in p32]
    y2b = ops.atari_conv_stack(obs.cuda(), idx.cuda(), *p32b)
    (y2b * g2.float().cuda()).sum().backward()
    assert torch.equal(y2, y2b)
    for a, b in zip(p32, p32b):
        assert torch.equal(a.grad, b.grad)


def test_model_fused_vs_miopen_path(ops):
    """AtariFfModel: fused MFMA conv stack vs the MIOpen path of the same module (same
    parameters): outputs and all parameter gradients agree to f32 tolerance; leading-dim
    handling [T,B], [B], [] preserved."""
    from rlpyt_amd.models.pg.atari_ff_model import AtariFfModel, ObsGather
    torch.manual_seed(0)
    model = AtariFfModel((4, 104, 80), 6).cuda()
    assert model.fused_conv
    obs = torch.randint(0, 256, (3, 4, 4, 104, 80), dtype=torch.uint8, device="cuda")

    def run(fused, inp):
        model.use_fused_conv = fused
        model.zero_grad(set_to_none=True)
        pi, v = model(inp, None, None)
        ((pi * torch.arange(6, device="cuda")).sum() + (v * v).sum()).backward()
        return pi.detach(), v.detach(), [p.grad.clone() for p in model.parameters()]
    pf, vf, gf = run(True, obs)
    pm, vm, gm = run(False, obs)
    assert pf.shape == (3, 4, 6) and vf.shape == (3, 4)
    np.testing.assert_allclose(pf.cpu().numpy(), pm.cpu().numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(vf.cpu().numpy(), vm.cpu().numpy(), rtol=1e-4, atol=1e-5)
    for a, b in zip(gf, gm):
        _close(a, b, rel=1e-4, what="param grad fused vs MIOpen")
    idx = torch.tensor([11, 0, 7, 5], device="cuda")
    pg, vg, _ = run(True, ObsGather(obs, idx))
    pe, ve, _ = run(False, obs[idx % 3, idx // 3])
    np.testing.assert_allclose(pg.cpu().numpy(), pe.cpu().numpy(), rtol=1e-4, atol=1e-6)
    with torch.no_grad():
        model.use_fused_conv = True
        p1, v1 = model(obs[0, 0], None, None)
        assert p1.shape == (6,) and v1.shape == ()
        p2, _ = model(obs[0], None, None)
        np.testing.assert_allclose(p2[0].cpu().numpy(), p1.cpu().numpy(), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("M,A,with_valid", [(8192, 6, False), (37, 6, True), (5, 3, False)])
def test_ppo_head_loss_fused_vs_reference(ops, M, A, with_valid):
    """ops.ppo_head_loss (heads + softmax + PPO loss, fwd + bwd in one pass) vs the torch
    restatement of the reference statements (oracle.ppo_loss_torch on softmax(h W^T + b)) in
    float64; loss scalars rtol 2e-5, gradients max-err <= 3e-5 * max|ref|."""
    from oracle import np_oracle as O
    K = 512
    g = torch.Generator().manual_seed(M + A)
    h = torch.relu(torch.randn(M, K, generator=g, dtype=torch.float64))
    w_pi = torch.randn(A, K, generator=g, dtype=torch.float64) * 0.05
    b_pi = torch.randn(A, generator=g, dtype=torch.float64) * 0.1
    w_v = torch.randn(1, K, generator=g, dtype=torch.float64) * 0.05
    b_v = torch.randn(1, generator=g, dtype=torch.float64) * 0.1
    po = torch.softmax(torch.randn(M, A, generator=g, dtype=torch.float64), -1)
    act = torch.randint(0, A, (M,), generator=g)
    adv = torch.randn(M, generator=g, dtype=torch.float64)
    ret = torch.randn(M, generator=g, dtype=torch.float64)
    valid = (torch.rand(M, generator=g) < 0.8).double() if with_valid else None
    ref_in = [t.clone().requires_grad_(True) for t in (h, w_pi, b_pi, w_v, b_v)]
    pi = torch.softmax(ref_in[0] @ ref_in[1].t() + ref_in[2], -1)
    v = (ref_in[0] @ ref_in[3].t()).squeeze(-1) + ref_in[4]
    ref = O.ppo_loss_torch(pi, v, po, act, adv, ret, valid, 0.1, 1.0, 0.01)
    ref[0].backward()
    dev_in = [t.float().cuda().requires_grad_(True) for t in (h, w_pi, b_pi, w_v, b_v)]
    f = lambda t: None if t is None else t.float().cuda()  # noqa: E731
    loss, sc = ops.ppo_head_loss(*dev_in, f(po), act.cuda(), f(adv), f(ret), f(valid), 0.1, 1.0,
                                 0.01)
    loss.backward()
    np.testing.assert_allclose(sc.cpu().numpy(), [x.item() for x in ref], rtol=2e-5, atol=1e-6)
    for name, a, b in zip(["h", "w_pi", "b_pi", "w_v", "b_v"], dev_in, ref_in):
        _close(a.grad, b.grad, rel=3e-5, what=f"head-loss grad {name}")


@pytest.mark.parametrize("M,A,with_valid", [(8192, 6, False), (37, 4, True), (5, 8, False)])
def test_ppo_trunk_head_loss_fused_vs_reference(ops, M, A, with_valid):
    """ops.ppo_head_loss(trunk_bias=...): the trunk's bias add + ReLU fused in front of the heads
    (pre-activation z = x W^T in, relu(z + b) applied by the kernel) vs the same statements in
    float64 -- loss scalars, dL/dz (masked by the ReLU), dL/dtrunk_bias and the head gradients."""
    from oracle import np_oracle as O
    K = 512
    g = torch.Generator().manual_seed(7 * M + A)
    z = torch.randn(M, K, generator=g, dtype=torch.float64)
    tb = torch.randn(K, generator=g, dtype=torch.float64) * 0.3
    w_pi = torch.randn(A, K, generator=g, dtype=torch.float64) * 0.05
    b_pi = torch.randn(A, generator=g, dtype=torch.float64) * 0.1
    w_v = torch.randn(1, K, generator=g, dtype=torch.float64) * 0.05
    b_v = torch.randn(1, generator=g, dtype=torch.float64) * 0.1
    po = torch.softmax(torch.randn(M, A, generator=g, dtype=torch.float64), -1)
    act = torch.randint(0, A, (M,), generator=g)
    adv = torch.randn(M, generator=g, dtype=torch.float64)
    ret = torch.randn(M, generator=g, dtype=torch.float64)
    valid = (torch.rand(M, generator=g) < 0.8).double() if with_valid else None
    ref_in = [t.clone().requires_grad_(True) for t in (z, w_pi, b_pi, w_v, b_v, tb)]
    h = torch.relu(ref_in[0] + ref_in[5])
    pi = torch.softmax(h @ ref_in[1].t() + ref_in[2], -1)
    v = (h @ ref_in[3].t()).squeeze(-1) + ref_in[4]
    ref = O.ppo_loss_torch(pi, v, po, act, adv, ret, valid, 0.1, 1.0, 0.01)
    ref[0].backward()
    dev_in = [t.float().cuda().requires_grad_(True) for t in (z, w_pi, b_pi, w_v, b_v, tb)]
    f = lambda t: None if t is None else t.float().cuda()  # noqa: E731
    loss, sc = ops.ppo_head_loss(*dev_in[:5], f(po), act.cuda(), f(adv), f(ret), f(valid), 0.1,
                                 1.0, 0.01, trunk_bias=dev_in[5])
    loss.backward()
    np.testing.assert_allclose(sc.cpu().numpy(), [x.item() for x in ref], rtol=2e-5, atol=1e-6)
    for name, a, b in zip(["z", "w_pi", "b_pi", "w_v", "b_v", "trunk_bias"], dev_in, ref_in):
        _close(a.grad, b.grad, rel=3e-5, what=f"trunk+head-loss grad {name}")
    # the masked entries are exact zeros
    dead = (z.float().cuda() + tb.float().cuda()) <= 0
    assert torch.all(dev_in[0].grad[dead] == 0)


def test_ppo_fused_and_unfused_loss_paths_agree(ops):
    """PPO.loss through the fused head-loss kernel vs through agent() + ops.ppo_loss on the same
    minibatch: same loss, same parameter gradients (f32 tolerance)."""
    from rlpyt_amd.agents.base import AgentInputs
    from rlpyt_amd.agents.pg.atari import AtariFfAgent
    from rlpyt_amd.algos.pg.ppo import PPO
    from rlpyt_amd.envs.synthetic import SyntheticPong
    from rlpyt_amd.models.pg.atari_ff_model import ObsGather
    torch.manual_seed(3)
    agent = AtariFfAgent()
    agent.initialize(SyntheticPong().spaces)
    agent.to_device(0)
    T, B, M = 5, 4, 12
    g = torch.Generator().manual_seed(1)
    obs = torch.randint(0, 256, (T, B, 4, 104, 80), dtype=torch.uint8, generator=g).cuda()
    idx = torch.randperm(T * B, generator=g)[:M].cuda()
    po = torch.softmax(torch.randn(M, 6, generator=g), -1).cuda()
    act = torch.randint(0, 6, (M,), generator=g).cuda()
    adv, ret = torch.randn(M, generator=g).cuda(), torch.randn(M, generator=g).cuda()
    res = []
    for fused in (True, False):
        algo = PPO(fused_head_loss=fused)
        algo.agent = agent
        agent.model.zero_grad(set_to_none=True)
        loss, sc = algo.loss(AgentInputs(ObsGather(obs, idx), None, None), act, ret, adv, None, po)
        loss.backward()
        res.append((sc.detach().cpu().numpy(), [p.grad.clone() for p in agent.parameters()]))
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=2e-5, atol=1e-6)
    for a, b in zip(res[0][1], res[1][1]):
        _close(a, b, rel=1e-4, what="fused vs unfused PPO param grad")


def _wide(shape, g, spread=3.0):
    """Full-mantissa values over ~5 decades with both signs: exercises all three bf16 pieces of a
    split operand (a 2-piece or single bf16 operand fails this by 3..5 orders of magnitude)."""
    return torch.randn(shape, generator=g, dtype=torch.float64) * \
        torch.exp(spread * torch.randn(shape, generator=g, dtype=torch.float64))


def test_bf16_split_kernels_are_f32_accurate(ops):
    """The conv kernels that run on the bf16 matrix pipe (conv1 forward / weight gradient: exact
    bf16x3 split of the f32 operand, uint8 operand exact in bf16; conv2 forward, M > #CUs: bf16x6,
    dropped products <= 2^-24, 2^-27 rms) against float64, beside torch's OWN f32 convolutions on the same
    device and inputs: our error must be f32-accumulation-order noise -- at most 2x torch-f32's
    (+ 2^-22 of the output scale).  A bf16 or 2-piece computation misses this bound by orders
    of magnitude on these wide-range operands."""
    from rlpyt_amd._lib import check, lib, ptr, stream
    M = 320                                            # > #CUs: the bf16x6 conv2 forward
    g = torch.Generator().manual_seed(7)
    obs = torch.randint(0, 256, (M, 4, 104, 80), dtype=torch.uint8, generator=g)
    w1_64 = (_wide((16, 4, 8, 8), g) * 0.02).float().double()       # f32-representable
    b1_64 = (_wide((16,), g, 1.0) * 0.1).float().double()
    w2_64 = (_wide((32, 16, 4, 4), g) * 0.02).float().double()
    b2_64 = (_wide((32,), g, 1.0) * 0.1).float().double()

    def bound(err_torch, scale):
        return 2 * err_torch + scale * 2.0 ** -22

    obs_d = obs.cuda()
    x64 = obs_d.double() / 255
    # ---- conv1 forward
    y1_64 = F.relu(F.conv2d(x64, w1_64.cuda(), b1_64.cuda(), stride=4))
    y1_t = F.relu(F.conv2d(obs_d.float() / 255, w1_64.float().cuda(), b1_64.float().cuda(), stride=4))
    w1, b1 = w1_64.float().cuda().contiguous(), b1_64.float().cuda().contiguous()
    y1 = torch.empty(M, 475, 16, device="cuda")
    check(lib.rlpyt_atari_conv1_fwd_f32(ptr(obs_d), None, 1, M, M, ptr(w1), ptr(b1), 1. / 255,
                                        ptr(y1), stream()), "conv1")
    ours = (y1.reshape(M, 25, 19, 16).permute(0, 3, 1, 2).double() - y1_64).abs().max().item()
    theirs = (y1_t.double() - y1_64).abs().max().item()
    assert ours <= bound(theirs, y1_64.abs().max().item()), ("conv1 fwd", ours, theirs)
    # ---- conv2 forward on wide-range activations
    a1_64 = _wide((M, 16, 25, 19), g, 2.0).abs().float().double().cuda()
    y2_64 = F.relu(F.conv2d(a1_64, w2_64.cuda(), b2_64.cuda(), stride=2, padding=1))
    y2_t = F.relu(F.conv2d(a1_64.float(), w2_64.float().cuda(), b2_64.float().cuda(), stride=2,
                           padding=1))
    y1_in = a1_64.float().permute(0, 2, 3, 1).reshape(M, 475, 16).contiguous()
    w2, b2 = w2_64.float().cuda().contiguous(), b2_64.float().cuda().contiguous()
    y2 = torch.empty(M, 3456, device="cuda")
    mask2 = torch.empty((M, 128), dtype=torch.int32, device="cuda")
    check(lib.rlpyt_atari_conv2_fwd_f32(ptr(y1_in), M, ptr(w2), ptr(b2), ptr(y2), ptr(mask2), stream()),
          "conv2")
    from rlpyt_amd import _lib
    assert _lib.last_variant().startswith("conv2_fwd_x6_kernel"), _lib.last_variant()
    ours = (y2.reshape(M, 32, 12, 9).double() - y2_64).abs().max().item()
    theirs = (y2_t.double() - y2_64).abs().max().item()
    assert ours <= bound(theirs, y2_64.abs().max().item()), ("conv2 fwd", ours, theirs)
    # ---- conv1 weight gradient on wide-range dy1
    dy1_64 = _wide((M, 16, 25, 19), g, 2.0).float().double().cuda()
    xin = x64.clone().requires_grad_(False)
    w_req = w1_64.cuda().clone().requires_grad_(True)
    (F.conv2d(xin, w_req, None, stride=4) * dy1_64).sum().backward()
    dw_t = torch.nn.grad.conv2d_weight(obs_d.float() / 255, (16, 4, 8, 8), dy1_64.float(), stride=4)
    ws = torch.empty(lib.rlpyt_atari_conv_wgrad_workspace_bytes(), dtype=torch.uint8, device="cuda")
    dw1, db1 = torch.empty(16, 4, 8, 8, device="cuda"), torch.empty(16, device="cuda")
    dy1_in = dy1_64.float().permute(0, 2, 3, 1).reshape(M, 475, 16).contiguous()
    check(lib.rlpyt_atari_conv1_wgrad_f32(ptr(obs_d), None, 1, M, M, ptr(dy1_in), 1. / 255,
                                          ptr(ws), ptr(dw1), ptr(db1), stream()), "wgrad1")
    ours = (dw1.double() - w_req.grad).abs().max().item()
    theirs = (dw_t.double() - w_req.grad).abs().max().item()
    assert ours <= bound(theirs, w_req.grad.abs().max().item()), ("conv1 wgrad", ours, theirs)
    db_ref = dy1_64.sum(dim=(0, 2, 3))
    assert (db1.double() - db_ref).abs().max().item() <= 1e-5 * dy1_64.abs().sum(dim=(0, 2, 3)).max().item()


def _gemm_case(ops, layout, M, N, K, **kw):
    """C[M,N] in layout NT (a[M,K] b[N,K]^T) / TN (a[K,M]^T b[K,N]) against float64, beside torch's
    own f32 GEMM of the same operands: (ours, theirs, scale)."""
    g = torch.Generator().manual_seed(M + N + K)
    a64 = _wide((M, K), g, 2.0).float().double().cuda()          # logical A [M, K]
    b64 = _wide((N, K), g, 2.0).float().double().cuda()          # logical B [N, K]
    ref = a64 @ b64.t()
    theirs = ((a64.float() @ b64.float().t()).double() - ref).abs().max().item()
    a, b = a64.float(), b64.float()
    if layout == "NT":
        c = ops.gemm_nt(a, b, **kw)
    else:
        c = ops.gemm_tn(a.t().contiguous(), b.t().contiguous())
    assert c.shape == ref.shape
    return (c.double() - ref).abs().max().item(), theirs, ref.abs().max().item()


# (8192, 3456, 512) and (8000, 2100, 64): >= 512 tiles of 256 x 128 -> the 256-row-tile variant of
# the lock-step kernel; (8192, 512, 3456) / (8192, 3456, 512) / (512, 3456, 8192): the three trunk
# GEMMs of the update at M = 8192 (forward, input gradient, weight gradient)
@pytest.mark.parametrize("M,N,K", [(8192, 512, 3456), (8192, 3456, 512), (8000, 2100, 64),
                                   (1000, 3456, 512), (130, 200, 96), (1, 1, 32)])
def test_gemm_nt_bf16x6_is_f32_accurate(ops, M, N, K):
    """ops.gemm_nt (a b^T from three-piece bf16 splits, six products; lock-step kernel
    rlpyt_gemm_nt_f32, both tile variants) against float64 beside torch's own f32 GEMM on wide-range
    operands: error <= 2x torch-f32's (+ 2^-22 of the output scale), including ragged tile edges."""
    ours, theirs, scale = _gemm_case(ops, "NT", M, N, K)
    assert ours <= 2 * theirs + scale * 2.0 ** -22, (ours, theirs)


@pytest.mark.parametrize("M,N,K", [(8192, 3456, 512), (5001, 3456, 64), (8000, 2100, 64)])
def test_gemm_nt_row_split_is_bit_identical(ops, M, N, K, monkeypatch):
    """Tall shapes run as two launches (256-row tiles for the first row blocks, 128-row tiles behind
    them: gemm_nt_plan); every output element is the same contraction in the same order whichever tile
    holds it -- bit-identical to one launch of 256-row tiles (RLPYT_GEMM_NT_R256=-1), to all 128-row
    tiles (0) and to an odd cut (ragged last block included)."""
    g = torch.Generator().manual_seed(M + K)
    a = torch.randn(M, K, generator=g).cuda()
    b = torch.randn(N, K, generator=g).cuda()
    monkeypatch.delenv("RLPYT_GEMM_NT_R256", raising=False)
    c_plan = ops.gemm_nt(a, b)
    for r in ("-1", "0", "7"):
        monkeypatch.setenv("RLPYT_GEMM_NT_R256", r)
        assert torch.equal(ops.gemm_nt(a, b), c_plan), r


# K >= 2048: the 8-chunk split with partial tiles (K = 8192 with 108 tiles: 96 whole-chunk units +
# 12 tiles in two K parts per XCD; K = 2080: ragged chunks of 8 / 9 steps); below: one unit per tile
@pytest.mark.parametrize("M,N,K", [(512, 3456, 8192), (512, 3456, 1024), (132, 200, 2080),
                                   (4, 4, 32), (260, 136, 4096), (128, 4096, 2048)])
def test_gemm_tn_bf16x6_is_f32_accurate(ops, M, N, K):
    """ops.gemm_tn (a^T b, contraction over the leading axis of both operands: the trunk's weight
    gradient g^T x), same bound; and run-to-run identical (fixed-order partial sums)."""
    ours, theirs, scale = _gemm_case(ops, "TN", M, N, K)
    assert ours <= 2 * theirs + scale * 2.0 ** -22, (ours, theirs)
    g = torch.Generator().manual_seed(1)
    a = torch.randn(K, M, generator=g).cuda()
    b = torch.randn(K, N, generator=g).cuda()
    c0 = ops.gemm_tn(a, b)
    for _ in range(3):
        assert torch.equal(ops.gemm_tn(a, b), c0)


def test_gemm_layouts_agree_on_asymmetric_data(ops):
    """Index-map check on data where a transposed / permuted operand would show: every layout
    reproduces the float64 product of the SAME logical matrices element by element (an error in a
    staging map moves whole rows, i.e. O(1) relative errors, far above this 1e-5 bound)."""
    M, N, K = 200, 264, 96
    g = torch.Generator().manual_seed(4)
    a = (torch.arange(M * K, dtype=torch.float64).reshape(M, K) % 17 - 8) / 8 + \
        torch.rand(M, K, generator=g, dtype=torch.float64)
    b = (torch.arange(N * K, dtype=torch.float64).reshape(N, K) % 13 - 6) / 6 + \
        torch.rand(N, K, generator=g, dtype=torch.float64)
    ref = (a @ b.t()).cuda()
    a32, b32 = a.float().cuda(), b.float().cuda()
    for name, c in (("nt", ops.gemm_nt(a32, b32)),
                    ("tn", ops.gemm_tn(a32.t().contiguous(), b32.t().contiguous()))):
        _close(c, ref, rel=1e-5, what=name)


def test_linear_nobias_autograd(ops):
    """ops.linear_nobias (gemm_nt forward and input gradient, gemm_tn weight gradient) against
    F.linear in float64."""
    g = torch.Generator().manual_seed(3)
    x64 = torch.randn(1024, 3456, generator=g, dtype=torch.float64).cuda().requires_grad_(True)
    w64 = (torch.randn(512, 3456, generator=g, dtype=torch.float64) * 0.02).cuda().requires_grad_(True)
    gy = torch.randn(1024, 512, generator=g, dtype=torch.float64).cuda()
    (F.linear(x64, w64) * gy).sum().backward()
    x = x64.detach().float().requires_grad_(True)
    w = w64.detach().float().requires_grad_(True)
    y = ops.linear_nobias(x, w)
    (y * gy.float()).sum().backward()
    _close(y, F.linear(x64, w64), rel=3e-6, what="linear_nobias fwd")
    _close(x.grad, x64.grad, rel=3e-6, what="linear_nobias dx")
    _close(w.grad, w64.grad, rel=3e-6, what="linear_nobias dw")


@pytest.mark.parametrize("rows", [2560, 1088])
def test_mlp_model_update_size_rows_on_the_split_gemms(ops, rows):
    """MlpModel at update-size row counts (R2D1's trunk FC 6912 -> 512 at 2 560 - 5 440 rows,
    rlpyt/models/mlp.py:24-31 as the head of rlpyt/models/conv2d.py:88-118): Linear layers on
    gemm_nt / gemm_tn under autograd and without, against the same modules in float64; the library
    path (RLPYT_MLP_GEMM=0 / fewer rows) held to the same bound beside it."""
    from rlpyt_amd import _lib
    from rlpyt_amd.models.mlp import MlpModel
    torch.manual_seed(5)
    mlp = MlpModel(6912, [512]).cuda()
    ref = MlpModel(6912, [512]).double().cuda()
    ref.load_state_dict({k: v.double() for k, v in mlp.state_dict().items()})
    g = torch.Generator().manual_seed(8)
    x64 = torch.randn(rows, 6912, generator=g, dtype=torch.float64).cuda().requires_grad_(True)
    gy = torch.randn(rows, 512, generator=g, dtype=torch.float64).cuda()
    y64 = ref.model(x64)
    (y64 * gy).sum().backward()
    x = x64.detach().float().requires_grad_(True)
    _lib.variant_reset()
    y = mlp(x)
    (y * gy.float()).sum().backward()
    counts = _lib.variant_counts()
    assert any(k.startswith("gemm_nt_x6_kernel") for k in counts) and "gemm_tn_x6_kernel" in counts, counts
    # (forward: 80 / 36 tiles of 128 x 128 -> the library's GEMM, ops._LinearNoBias.MIN_FORWARD_TILES)
    lin, lin64 = mlp.model[0], ref.model[0]
    for what, a, d in (("y", y, y64), ("dx", x.grad, x64.grad), ("dW", lin.weight.grad, lin64.weight.grad),
                       ("db", lin.bias.grad, lin64.bias.grad)):
        _close(a, d, rel=5e-6, what=f"MlpModel wide {what}")
    with torch.no_grad():
        lib_y = mlp.model(x.detach())                     # the library's f32 path
    err_own = (y.detach().double() - y64).abs().max().item()
    err_lib = (lib_y.double() - y64).abs().max().item()
    assert err_own <= 2 * err_lib + 1e-6, (err_own, err_lib)


def _check_conv2_bwd_tail(y1, y2, g2, w2, dy1, n):
    """dy1 of the LAST n images against a float64 evaluation of conv2's backward-data pass."""
    M = y1.shape[0]
    a1 = y1[-n:].double().reshape(n, 25, 19, 16).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    z2 = F.conv2d(a1, w2.double(), None, stride=2, padding=1)
    gm2 = g2[-n:].double().reshape(n, 32, 12, 9) * (y2[-n:].reshape(n, 32, 12, 9) > 0)
    (z2 * gm2).sum().backward()
    ref = (a1.grad * (a1 > 0)).permute(0, 2, 3, 1).reshape(n, 475, 16)
    _close(dy1[-n:], ref, what=f"conv2_bwd_x6 dy1, last {n} images of M = {M}")


def test_conv_kernels_run_to_run_identical_at_update_size(ops):
    """Every conv kernel and the trunk GEMM at the update's M = 8192 (32 images per persistent
    workgroup), ten launches on identical inputs: bit-identical outputs.  (A stale-register bug in
    a hand-waited prefetch showed up exactly here -- from the third image of a workgroup on --
    while every small-M parity test passed.)"""
    from rlpyt_amd._lib import check, lib, ptr, stream
    M, T, B = 8192, 64, 160
    g = torch.Generator().manual_seed(0)
    obs = torch.randint(0, 256, (T, B, 4, 104, 80), dtype=torch.uint8, generator=g).cuda()
    idx = torch.randperm(T * B, generator=g)[:M].cuda()
    w1, b1 = (torch.randn(16, 4, 8, 8, generator=g) * 0.05).cuda(), (torch.randn(16, generator=g) * 0.1).cuda()
    w2, b2 = (torch.randn(32, 16, 4, 4, generator=g) * 0.05).cuda(), (torch.randn(32, generator=g) * 0.1).cuda()
    g2 = torch.randn(M, 3456, generator=g).cuda()
    a = torch.randn(M, 3456, generator=g).cuda()
    wt = (torch.randn(512, 3456, generator=g) * 0.02).cuda()
    g512 = torch.randn(M, 512, generator=g).cuda()
    ws = torch.empty(lib.rlpyt_atari_conv_wgrad_workspace_bytes(), dtype=torch.uint8, device="cuda")
    ref = None
    for it in range(10):
        y1 = torch.empty(M, 475, 16, device="cuda")
        y2 = torch.empty(M, 3456, device="cuda")
        dy1 = torch.empty_like(y1)
        dw2, db2, dw1, db1 = (torch.empty_like(t) for t in (w2, b2, w1, b1))
        check(lib.rlpyt_atari_conv1_fwd_f32(ptr(obs), ptr(idx), T, B, M, ptr(w1), ptr(b1), 1. / 255,
                                            ptr(y1), stream()))
        mask = torch.empty((M, 128), dtype=torch.int32, device="cuda")
        check(lib.rlpyt_atari_conv2_fwd_f32(ptr(y1), M, ptr(w2), ptr(b2), ptr(y2), ptr(mask), stream()))
        # the bf16x6 conv2 backward (32 images per persistent workgroup: its double-buffered planes
        # and register pipelines in steady state), fed by the forward kernel's own sign mask
        check(lib.rlpyt_atari_conv2_bwd_x6_f32(ptr(g2), ptr(mask), ptr(y1), M, ptr(w2), ptr(dy1), ptr(ws),
                                               ptr(dw2), ptr(db2), stream()))
        check(lib.rlpyt_atari_conv1_wgrad_f32(ptr(obs), ptr(idx), T, B, M, ptr(dy1), 1. / 255, ptr(ws),
                                              ptr(dw1), ptr(db1), stream()))
        if it == 0:
            assert torch.equal(mask, _sign_mask(y2))      # mask at M = 8192 (bf16x6 epilogue)
            # conv2 backward against float64 on the last 48 images (chunked reference)
            _check_conv2_bwd_tail(y1, y2, g2, w2, dy1, n=48)
        cur = dict(y1=y1, y2=y2, mask=mask, dy1=dy1, dw2=dw2, db2=db2, dw1=dw1, db1=db1,
                   gemm=ops.gemm_nt(a, wt), gemm_tn=ops.gemm_tn(g512, a))
        torch.cuda.synchronize()
        if ref is None:
            ref = {k: v.clone() for k, v in cur.items()}
            # and the last images of the batch are right (float64 reference of 64 of them)
            sel = idx[-64:]
            x = obs.reshape(T * B, 4, 104, 80)[(sel % T) * B + sel // T].double() / 255
            r1 = F.relu(F.conv2d(x, w1.double(), b1.double(), stride=4))
            r2 = F.relu(F.conv2d(r1, w2.double(), b2.double(), stride=2, padding=1))
            _close(y2[-64:], r2.reshape(64, -1), what="conv stack, last 64 images of M = 8192")
        else:
            for k, v in cur.items():
                assert torch.equal(v, ref[k]), (k, it)
