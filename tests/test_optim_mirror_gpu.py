"""``ClipAdam`` keeps the transposed copy of the update trunk's weight current (round 6:
``rlpyt_clip_adam_step_mirror_f32`` + ``ops.TransposedMirror``) -- the new ``W^T`` the input-gradient
GEMM of ``_LinearNoBias`` reads comes out of the optimizer's own launch instead of a 7 MB transposing
copy per minibatch (rlpyt/algos/pg/ppo.py:100-104 + rlpyt/models/mlp.py:24-31 under autograd)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _direct_copies():
    from rlpyt_amd import _lib
    return {k: v for k, v in _lib.variant_counts().items() if v > 0}


@pytest.mark.parametrize("shape", [(512, 3456), (64, 96), (32, 32)])
def test_mirror_equals_transpose_and_step_equals_plain_clip_adam(shape):
    from rlpyt_amd import ops
    from rlpyt_amd.optim import ClipAdam
    g = torch.Generator().manual_seed(3)
    R, C = shape
    w0 = torch.randn(R, C, generator=g) * 0.05
    b0 = torch.randn(R, generator=g) * 0.05
    grads = [(torch.randn(R, C, generator=g), torch.randn(R, generator=g)) for _ in range(3)]

    def run(with_mirror):
        w = torch.nn.Parameter(w0.clone().cuda())
        b = torch.nn.Parameter(b0.clone().cuda())
        opt = ClipAdam([w, b], lr=1e-2)
        norms = []
        for gw, gb in grads:
            if with_mirror:
                wt = ops.TransposedMirror.get(w)              # what _LinearNoBias.backward asks for
                assert torch.equal(wt, w.detach().t())
            w.grad, b.grad = gw.cuda(), gb.cuda()
            norms.append(float(opt.clip_and_step(0.5)))
            if with_mirror:
                buf = ops.TransposedMirror.buffer_for(w)
                assert buf is not None and torch.equal(buf, w.detach().t()), "mirror != new W^T"
                ptr = buf.data_ptr()
                assert ops.TransposedMirror.get(w).data_ptr() == ptr     # a hit: no new copy
        return w.detach().cpu(), b.detach().cpu(), norms

    w_m, b_m, n_m = run(True)
    w_p, b_p, n_p = run(False)
    assert torch.equal(w_m, w_p) and torch.equal(b_m, b_p) and n_m == n_p
    # ... and both equal clip_grad_norm_ + torch.optim.Adam
    w = torch.nn.Parameter(w0.clone().cuda())
    b = torch.nn.Parameter(b0.clone().cuda())
    ref = torch.optim.Adam([w, b], lr=1e-2)
    for gw, gb in grads:
        w.grad, b.grad = gw.cuda(), gb.cuda()
        torch.nn.utils.clip_grad_norm_([w, b], 0.5)
        ref.step()
    np.testing.assert_allclose(w_m.numpy(), w.detach().cpu().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(b_m.numpy(), b.detach().cpu().numpy(), rtol=0, atol=2e-6)


def test_mirror_goes_stale_safely_when_someone_else_writes_the_weight():
    """Any writer other than the mirroring optimizer bumps the version counter: the next ``get`` makes
    a fresh copy (load_state_dict / in-place edits / another optimizer)."""
    from rlpyt_amd import ops
    w = torch.nn.Parameter(torch.randn(64, 96).cuda())
    wt = ops.TransposedMirror.get(w)
    assert torch.equal(wt, w.detach().t())
    with torch.no_grad():
        w.mul_(2.0)
    assert not torch.equal(ops.TransposedMirror.buffer_for(w), w.detach().t())       # stale buffer ...
    assert torch.equal(ops.TransposedMirror.get(w), w.detach().t())                   # ... refreshed
    sgd = torch.optim.SGD([w], lr=0.1)
    w.grad = torch.ones_like(w)
    sgd.step()
    assert torch.equal(ops.TransposedMirror.get(w), w.detach().t())


def test_update_minibatch_issues_no_transposing_copy_after_the_first():
    """Two PPO minibatch updates at M = 1024 through the product path: the second backward finds W^T
    current (no ATen copy kernel between the GEMMs), diagnostics land in the table row."""
    from rlpyt_amd.agents.base import AgentInputs
    from rlpyt_amd.agents.pg.atari import AtariFfAgent
    from rlpyt_amd.algos.pg.ppo import PPO
    from rlpyt_amd.envs.synthetic import SyntheticPong
    from rlpyt_amd import ops
    torch.manual_seed(2)
    agent = AtariFfAgent()
    agent.initialize(SyntheticPong().spaces)
    agent.to_device(0)
    T, B, A, M = 32, 64, 6, 1024
    g = torch.Generator().manual_seed(5)
    obs = torch.randint(0, 256, (T, B, 4, 104, 80), dtype=torch.uint8, generator=g).cuda()
    action = torch.randint(0, A, (T, B), generator=g).cuda()
    adv, ret = torch.randn(T, B, generator=g).cuda(), torch.randn(T, B, generator=g).cuda()
    po = torch.softmax(torch.randn(T, B, A, generator=g), -1).cuda()
    algo = PPO()
    algo.agent = agent
    from rlpyt_amd.optim import ClipAdam
    algo.optimizer = ClipAdam(list(agent.parameters()), lr=1e-3)
    lin = agent.model._single_fc()
    table = torch.zeros(2, 6, device="cuda")
    one = torch.ones((), device="cuda")
    for k in range(2):
        idx = torch.randperm(T * B, generator=g)[:M].cuda()
        algo.optimizer.zero_grad(set_to_none=True)
        v_before = lin.weight._version
        loss, _ = algo.loss(AgentInputs(agent.gather_observation(obs, idx), None, None), action, ret, adv,
                            None, po, flat_idx=idx, unit_grad=True, scalars_out=table[k, :5])
        torch.autograd.backward(loss, grad_tensors=one)
        algo.optimizer.clip_and_step(1.0, norm_out=table[k, 5:])
        assert lin.weight._version > v_before
        buf = ops.TransposedMirror.buffer_for(lin.weight)
        assert buf is not None and torch.equal(buf, lin.weight.detach().t())
    rows = table.cpu().numpy()
    assert np.isfinite(rows).all() and (rows[:, 5] > 0).all() and (rows[:, 3] > 0).all()
    np.testing.assert_allclose(rows[:, 0], rows[:, 1] + rows[:, 2] - 0.01 * rows[:, 3], rtol=1e-4, atol=1e-6)
