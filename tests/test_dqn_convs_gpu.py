"""No-grad forward of the DQN-family conv stack (``rlpyt_dqn_convs_fwd_f32``, csrc/dqn_convs.hip)
against ``torch.nn.Conv2d`` -- the modules the reference's models are made of
(rlpyt/models/conv2d.py:8-57 under rlpyt/models/dqn/atari_dqn_model.py:30-37,
atari_r2d1_model.py:33-41).  f32 MFMA, f32 accumulate, another summation order: held to three times torch-f32's
own error against float64 (measured: 1.0-2.3x), at every launch geometry (1 image .. beyond one wave of workgroups), and
through the models that dispatch to it."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _stack(seed, scale_w=1.0):
    torch.manual_seed(seed)
    convs = [torch.nn.Conv2d(4, 32, 8, stride=4), torch.nn.Conv2d(32, 64, 4, stride=2, padding=1),
             torch.nn.Conv2d(64, 64, 3, stride=1, padding=1)]
    with torch.no_grad():
        for c in convs:
            c.weight.mul_(scale_w)
            c.bias.uniform_(-0.2, 0.2)       # biases that move ReLU's zero crossings
    return convs


@torch.no_grad()
def _reference(convs, obs, dtype):
    x = obs.to(dtype) * (1. / 255)
    for c in convs:
        x = torch.relu(torch.nn.functional.conv2d(x, c.weight.to(dtype), c.bias.to(dtype),
                                                  stride=c.stride, padding=c.padding))
    return x.reshape(obs.shape[0], -1)


def _check(ops, convs, obs):
    dev = [c.cuda() for c in convs]
    got = ops.dqn_convs_fwd(obs.cuda(), *[p for c in dev for p in (c.weight.detach(), c.bias.detach())])
    torch.cuda.synchronize()
    got = got.cpu().double()
    ref64 = _reference([c.cpu() for c in convs], obs.cpu(), torch.float64)
    ref32 = _reference([c.cpu() for c in convs], obs.cpu(), torch.float32).double()
    scale = ref64.abs().max().item() + 1e-12
    err = (got - ref64).abs().max().item() / scale
    err32 = (ref32 - ref64).abs().max().item() / scale
    assert got.shape == ref64.shape
    assert err <= max(3 * err32, 3e-7), (err, err32)
    return got, ref64


@pytest.mark.parametrize("N", [1, 3, 16, 48, 130, 300])
def test_dqn_convs_match_torch_conv2d(N):
    from rlpyt_amd import ops
    g = torch.Generator().manual_seed(N)
    obs = torch.randint(0, 256, (N, 4, 104, 80), dtype=torch.uint8, generator=g)
    _check(ops, _stack(N), obs)


def test_dqn_convs_edge_images_and_wide_weights():
    """All-black / all-white frames (border handling: padding must contribute exact zeros), single
    hot pixels in every corner, weights 8x the default scale (large activations through three ReLUs)."""
    from rlpyt_amd import ops
    obs = torch.zeros((6, 4, 104, 80), dtype=torch.uint8)
    obs[1] = 255
    obs[2, 0, 0, 0] = 255
    obs[3, 3, 103, 79] = 255
    obs[4, 1, 0, 79] = 200
    obs[5, 2, 103, 0] = 17
    _check(ops, _stack(5), obs)
    g = torch.Generator().manual_seed(1)
    _check(ops, _stack(6, scale_w=8.0), torch.randint(0, 256, (9, 4, 104, 80), dtype=torch.uint8, generator=g))


def test_dqn_convs_identity_like_weights_asymmetric():
    """Weights that pick ONE input tap / channel per output channel: any transposition of (ky, kx),
    channel order or position order inside the packed layout shows up as a wrong pixel, not as a
    small numeric difference."""
    from rlpyt_amd import ops
    convs = _stack(0)
    with torch.no_grad():
        for li, c in enumerate(convs):
            c.weight.zero_()
            c.bias.zero_()
            co, ci, kh, kw = c.weight.shape
            for o in range(co):
                c.weight[o, (3 * o + li) % ci, (o + 2 * li) % kh, (5 * o + 1) % kw] = 1.0 + 0.01 * o
    g = torch.Generator().manual_seed(3)
    obs = torch.randint(0, 256, (5, 4, 104, 80), dtype=torch.uint8, generator=g)
    got, ref = _check(ops, convs, obs)
    assert ref.abs().max() > 0.1
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=3e-6, atol=1e-6)


def test_dqn_convs_see_parameter_updates_without_refresh():
    """The packed weight copy is rebuilt on the stream at every call: an in-place parameter change
    (optimizer step, target update) is visible to the very next call, also from a captured graph."""
    from rlpyt_amd import ops
    convs = [c.cuda() for c in _stack(2)]
    args = [p for c in convs for p in (c.weight.detach(), c.bias.detach())]
    g = torch.Generator().manual_seed(4)
    obs = torch.randint(0, 256, (8, 4, 104, 80), dtype=torch.uint8, generator=g).cuda()
    out = torch.empty((8, 6912), device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ops.dqn_convs_fwd(obs, *args, out=out)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            ops.dqn_convs_fwd(obs, *args, out=out)
    torch.cuda.synchronize()
    first = out.clone()
    with torch.no_grad():
        convs[1].weight.mul_(0.5)
        convs[2].bias.add_(0.05)
    graph.replay()
    torch.cuda.synchronize()
    want = _reference([c.cpu() for c in convs], obs.cpu(), torch.float64)
    assert not torch.allclose(out, first)
    np.testing.assert_allclose(out.cpu().double().numpy(), want.numpy(), rtol=1e-4, atol=2e-6)


@pytest.mark.parametrize("lead", [(), (5,), (2, 3)])
def test_dqn_family_models_dispatch_to_the_fused_convs(lead):
    """AtariDqnModel / AtariCatDqnModel / AtariR2d1Model: the no-grad device forward (fused convs)
    equals the same model's library path (``use_fused_nograd_convs = False``) and its autograd
    forward, and the launch counters show which one ran."""
    from rlpyt_amd import _lib
    from rlpyt_amd.models.dqn.atari_catdqn_model import AtariCatDqnModel
    from rlpyt_amd.models.dqn.atari_dqn_model import AtariDqnModel
    from rlpyt_amd.models.dqn.atari_r2d1_model import AtariR2d1Model
    g = torch.Generator().manual_seed(7)
    obs = torch.randint(0, 256, lead + (4, 104, 80), dtype=torch.uint8, generator=g).cuda()
    pa = torch.zeros(lead + (6,), device="cuda")
    pr = torch.zeros(lead, device="cuda")

    def ran_fused():
        return any("dqn_conv23" in k and v > 0 for k, v in _lib.variant_counts().items())

    for Cls, kw in [(AtariDqnModel, {}), (AtariDqnModel, dict(dueling=True)), (AtariCatDqnModel, {})]:
        torch.manual_seed(1)
        m = Cls(image_shape=(4, 104, 80), output_size=6, **kw).cuda()
        _lib.variant_reset()
        with torch.no_grad():
            fused = m(obs, pa, pr)
        assert ran_fused()
        _lib.variant_reset()
        grad = m(obs, pa, pr)                      # autograd: the same kernels, activations kept
        assert ran_fused() and grad.requires_grad
        m.conv.use_fused_grad_convs = False
        _lib.variant_reset()
        grad = m(obs, pa, pr)                      # autograd through the library convolutions
        assert not ran_fused() and grad.requires_grad
        m.conv.use_fused_nograd_convs = False
        with torch.no_grad():
            lib = m(obs, pa, pr)
        assert not ran_fused()
        np.testing.assert_allclose(fused.cpu().numpy(), lib.cpu().numpy(), rtol=2e-4, atol=2e-6)
        np.testing.assert_allclose(fused.cpu().numpy(), grad.detach().cpu().numpy(), rtol=2e-4, atol=2e-6)
    if len(lead) == 2:
        torch.manual_seed(2)
        m = AtariR2d1Model(image_shape=(4, 104, 80), output_size=6, dueling=True).cuda()
        with torch.no_grad():
            _lib.variant_reset()
            q1, s1 = m(obs, pa, pr, None)
            assert ran_fused()
            m.conv.conv.use_fused_nograd_convs = False
            q2, s2 = m(obs, pa, pr, None)
        np.testing.assert_allclose(q1.cpu().numpy(), q2.cpu().numpy(), rtol=2e-4, atol=2e-6)
        np.testing.assert_allclose(s1.h.cpu().numpy(), s2.h.cpu().numpy(), rtol=2e-4, atol=2e-6)


@pytest.mark.parametrize("own_bwd", [True, False])
@pytest.mark.parametrize("N", [1, 2, 32, 128, 257, 300])
def test_dqn_convs_under_autograd_gradients_match_the_library_path(N, own_bwd):
    """``ops.dqn_convs`` (own forward kernels, channels-last activations kept; backward = the own
    kernels of csrc/dqn_convs_bwd.hip, or -- ``own_bwd=False``, the A/B switch -- the library's
    convolution_backward on the kept activations): features and all six parameter gradients against
    the ``torch.nn.Conv2d`` modules in float64, held to three times the f32 module path's own error.
    N = 1 .. 300: one image per workgroup, several images per weight-gradient workgroup (N > 64 / 128),
    ragged last group (257, 300)."""
    from rlpyt_amd import _lib, ops
    was = ops.DQN_CONVS_OWN_BWD
    ops.DQN_CONVS_OWN_BWD = own_bwd
    try:
        _lib.variant_reset()
        _gradient_case(ops, N)
        ran = any("dqn_wgrad1_kernel" in k for k in _lib.variant_counts())
        assert ran == own_bwd
    finally:
        ops.DQN_CONVS_OWN_BWD = was


def test_dqn_convs_backward_is_deterministic_and_overwrites():
    """The weight-gradient partials are summed in a fixed order: two backward passes over the same
    batch give bit-identical gradients (also into buffers holding stale values)."""
    from rlpyt_amd import ops
    convs = [c.cuda() for c in _stack(77)]
    g = torch.Generator().manual_seed(5)
    obs = torch.randint(0, 256, (130, 4, 104, 80), dtype=torch.uint8, generator=g).cuda()
    cot = torch.randn(130, 6912, generator=g).cuda()
    params = [p for c in convs for p in (c.weight, c.bias)]
    runs = []
    for _ in range(2):
        for p in params:
            p.grad = None
        ops.dqn_convs(obs, *params).backward(cot)
        runs.append([p.grad.clone() for p in params])
    for a, b in zip(*runs):
        assert torch.equal(a, b)


def test_dqn_convs_backward_dead_and_saturated_units():
    """Biases that switch whole channels off (ReLU mask all zero: exact zero gradients through them)
    next to channels that are always on."""
    from rlpyt_amd import ops
    convs = _stack(78)
    with torch.no_grad():
        convs[0].bias[:8] = -50.
        convs[1].bias[5:20] = -50.
        convs[2].bias[60:] = -50.
        convs[2].bias[:4] = 50.
    dev = [torch.nn.Conv2d(c.in_channels, c.out_channels, c.kernel_size, c.stride, c.padding).cuda()
           for c in convs]
    for c, src in zip(dev, convs):
        c.load_state_dict(src.state_dict())
    g = torch.Generator().manual_seed(6)
    obs = torch.randint(0, 256, (9, 4, 104, 80), dtype=torch.uint8, generator=g).cuda()
    cot = torch.randn(9, 6912, generator=g).cuda()
    ops.dqn_convs(obs, *[p for c in dev for p in (c.weight, c.bias)]).backward(cot)
    ref = [torch.nn.Conv2d(c.in_channels, c.out_channels, c.kernel_size, c.stride, c.padding).double()
           for c in convs]
    for c, src in zip(ref, convs):
        c.load_state_dict({k: v.double() for k, v in src.state_dict().items()})
    x = obs.cpu().double() * (1. / 255)
    for c in ref:
        x = torch.relu(c(x))
    x.reshape(9, -1).backward(cot.cpu().double())
    assert torch.equal(dev[0].weight.grad[:8].cpu(), torch.zeros(8, 4, 8, 8))
    assert torch.equal(dev[1].bias.grad[5:20].cpu(), torch.zeros(15))
    assert torch.equal(dev[2].weight.grad[60:].cpu(), torch.zeros(4, 64, 3, 3))
    for c, r in zip(dev, ref):
        for a, b in zip(c.parameters(), r.parameters()):
            scale = b.grad.abs().max().item() + 1e-30
            assert (a.grad.cpu().double() - b.grad).abs().max().item() / scale < 5e-6


def _gradient_case(ops, N):
    convs = _stack(40 + N)
    g = torch.Generator().manual_seed(N)
    obs = torch.randint(0, 256, (N, 4, 104, 80), dtype=torch.uint8, generator=g)
    cot = torch.randn(N, 6912, generator=g)

    def module_path(dtype, device):
        cs = [torch.nn.Conv2d(c.in_channels, c.out_channels, c.kernel_size, c.stride, c.padding)
              .to(device=device, dtype=dtype) for c in convs]
        for c, src in zip(cs, convs):
            c.load_state_dict({k: v.to(dtype) for k, v in src.state_dict().items()})
        x = obs.to(device=device, dtype=dtype) * (1. / 255)
        for c in cs:
            x = torch.relu(c(x))
        y = x.reshape(N, -1)
        y.backward(cot.to(device=device, dtype=dtype))
        return y.detach().cpu().double(), [p.grad.detach().cpu().double() for c in cs for p in c.parameters()]

    y64, g64 = module_path(torch.float64, "cpu")
    y32, g32 = module_path(torch.float32, "cuda")
    dev = [torch.nn.Conv2d(c.in_channels, c.out_channels, c.kernel_size, c.stride, c.padding).cuda()
           for c in convs]
    for c, src in zip(dev, convs):
        c.load_state_dict(src.state_dict())
    y = ops.dqn_convs(obs.cuda(), *[p for c in dev for p in (c.weight, c.bias)])
    y.backward(cot.cuda())
    torch.cuda.synchronize()
    got = [p.grad.detach().cpu().double() for c in dev for p in c.parameters()]

    def rel(a, b):
        return (a - b).abs().max().item() / (b.abs().max().item() + 1e-30)
    assert rel(y.detach().cpu().double(), y64) <= max(3 * rel(y32, y64), 3e-7)
    for k, (a, b32, b64) in enumerate(zip(got, g32, g64)):
        assert a.shape == b64.shape
        assert rel(a, b64) <= max(3 * rel(b32, b64), 2e-6), (k, rel(a, b64), rel(b32, b64))


def test_packed_weights_are_made_once_per_sampling_phase_and_never_stale_in_training():
    """``Conv2dModel.refresh_step_weights`` (the agent calls it on entering sample / eval mode) packs
    the weights once; forwards in eval mode then skip the packing launch; a module in training mode
    (target network, double-DQN pass between optimizer steps) packs per call and sees parameter
    changes at once."""
    from rlpyt_amd import _lib
    from rlpyt_amd.models.dqn.atari_dqn_model import AtariDqnModel
    torch.manual_seed(8)
    m = AtariDqnModel(image_shape=(4, 104, 80), output_size=6).cuda()
    g = torch.Generator().manual_seed(9)
    obs = torch.randint(0, 256, (7, 4, 104, 80), dtype=torch.uint8, generator=g).cuda()
    pa, pr = torch.zeros(7, 6, device="cuda"), torch.zeros(7, device="cuda")

    def packs():
        # (one launch per packing: the bf16 pieces of conv2 / conv3; + the f32 register-order copies when
        #  an f32-MFMA kernel is switched on)
        c = _lib.variant_counts()
        return max(sum(v for k, v in c.items() if "dqn_pack_weights_kernel" in k),
                   sum(v for k, v in c.items() if "dqn_x6_pack_kernel" in k or "dqn_t32_pack_kernel" in k))

    with torch.no_grad():
        m.train()
        _lib.variant_reset()
        q_train = m(obs, pa, pr)
        assert packs() == 1
        m.eval()
        m.refresh_step_weights()
        _lib.variant_reset()
        q_eval = m(obs, pa, pr)
        q_eval2 = m(obs, pa, pr)
        assert packs() == 0
        assert torch.equal(q_train, q_eval) and torch.equal(q_eval, q_eval2)
        # an optimizer step while training: the next no-grad forward is current
        m.train()
        m.conv.conv[2].weight.mul_(0.5)
        q_new = m(obs, pa, pr)
        m.conv.use_fused_nograd_convs = False
        q_lib = m(obs, pa, pr)
    assert not torch.allclose(q_new, q_train)
    np.testing.assert_allclose(q_new.cpu().numpy(), q_lib.cpu().numpy(), rtol=2e-4, atol=2e-6)


def test_packed_weights_are_dropped_when_parameters_change_in_eval_mode():
    """weights can change while the module STAYS in eval mode (``load_state_dict`` after
    ``eval_mode``, an in-place edit): the phase's pack then no longer describes the parameters and the
    forward must pack on the stream again instead of running with stale conv weights."""
    from rlpyt_amd import _lib
    from rlpyt_amd.models.dqn.atari_dqn_model import AtariDqnModel
    torch.manual_seed(18)
    m = AtariDqnModel(image_shape=(4, 104, 80), output_size=6).cuda()
    other = AtariDqnModel(image_shape=(4, 104, 80), output_size=6).cuda()
    g = torch.Generator().manual_seed(19)
    obs = torch.randint(0, 256, (5, 4, 104, 80), dtype=torch.uint8, generator=g).cuda()
    pa, pr = torch.zeros(5, 6, device="cuda"), torch.zeros(5, device="cuda")

    def packs():
        # (one launch per packing: the bf16 pieces of conv2 / conv3; + the f32 register-order copies when
        #  an f32-MFMA kernel is switched on)
        c = _lib.variant_counts()
        return max(sum(v for k, v in c.items() if "dqn_pack_weights_kernel" in k),
                   sum(v for k, v in c.items() if "dqn_x6_pack_kernel" in k or "dqn_t32_pack_kernel" in k))

    with torch.no_grad():
        m.eval()
        m.refresh_step_weights()
        q0 = m(obs, pa, pr)
        m.load_state_dict(other.state_dict())           # still in eval mode, no refresh
        _lib.variant_reset()
        q1 = m(obs, pa, pr)
        assert packs() == 1, "stale pack used after load_state_dict in eval mode"
        other.eval()
        other.conv.use_fused_nograd_convs = False
        q_lib = other(obs, pa, pr)
        m.refresh_step_weights()                        # a refresh makes the pack current again
        _lib.variant_reset()
        q2 = m(obs, pa, pr)
        assert packs() == 0
    assert not torch.allclose(q0, q1)
    assert torch.equal(q1, q2)
    np.testing.assert_allclose(q1.cpu().numpy(), q_lib.cpu().numpy(), rtol=2e-4, atol=2e-6)
