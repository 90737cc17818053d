"""``rlpyt_replay_append`` (one launch per ``append_samples``) against the slice-assignment parts of
``replays/store.py`` -- the path the CPU suite pins to the reference's layout contract
(tests/test_host_logic.py::test_replay_store_parts_on_host_tensors; rlpyt/replays/n_step.py:60-83,
rlpyt/replays/frame.py:39-59): every ring array bit-identical after every append, over appends that
start at row 0, wrap inside, close a lap exactly, and cover a whole lap."""
import numpy as np
import pytest
import torch

from rlpyt_amd.utils.collections import namedarraytuple

S2B = namedarraytuple("SamplesToBuffer", ["observation", "action", "reward", "done"])
Nested = namedarraytuple("Nested", ["a", "b"])
S2BX = namedarraytuple("SamplesToBufferX", ["observation", "action", "reward", "done", "extra"])


def _example(C, H, W, extra=False):
    base = dict(observation=np.zeros((C, H, W), np.uint8), action=np.int64(0), reward=np.float32(0),
                done=np.bool_(False))
    if extra:
        return S2BX(**base, extra=Nested(a=np.zeros(3, np.float32), b=np.int16(0)))
    return S2B(**base)


def _record(rng, T, B, C, H, W, extra=False):
    base = dict(observation=torch.from_numpy(rng.randint(0, 256, (T, B, C, H, W)).astype(np.uint8)),
                action=torch.from_numpy(rng.randint(0, 6, (T, B))),
                reward=torch.from_numpy(rng.randn(T, B).astype(np.float32)),
                done=torch.from_numpy(rng.rand(T, B) < 0.2))
    if extra:
        return S2BX(**base, extra=Nested(a=torch.from_numpy(rng.randn(T, B, 3).astype(np.float32)),
                                         b=torch.from_numpy(rng.randint(-9, 9, (T, B)).astype(np.int16))))
    return S2B(**base)


def _leaves(buf):
    from rlpyt_amd.utils.buffer import buffer_leaves
    out = buffer_leaves(buf.samples) + [buf.samples_frames]
    if buf.n_step_return > 1:
        out += [buf.samples_return_, buf.samples_done_n]
    return [x.cpu() for x in out]


def _to(rec, dev):
    from rlpyt_amd.utils.buffer import buffer_to
    return buffer_to(rec, dev)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(4, 104, 80, 16, False), (4, 3, 2, 2, True), (3, 7, 5, 3, False),
                                   (1, 9, 4, 5, False)])
def test_one_launch_append_equals_the_slice_assignments(shape):
    from rlpyt_amd import _lib
    from rlpyt_amd.replays.buffers import UniformReplayFrameBuffer
    C, H, W, B, extra = shape
    ring_T = 10
    kw = dict(example=_example(C, H, W, extra), size=ring_T * B, B=B, discount=0.99, n_step_return=1)
    on_host = UniformReplayFrameBuffer(device="cpu", **kw)
    on_dev = UniformReplayFrameBuffer(device="cuda", **kw)
    rng = np.random.RandomState(3)
    # 10 rows = exactly one lap from row 0; wraps inside appends; 6 + 4: a lap closed exactly on row
    # 0 from a non-zero start; a whole lap from a non-zero start; a whole lap from row 0
    _lib.variant_reset()
    for n, T_new in enumerate((3, 4, 3, 5, 2, 7, 6, 4, 1, 10, 9, 10, 10)):
        rec = _record(rng, T_new, B, C, H, W, extra)
        a = on_host.append_samples(rec)
        b = on_dev.append_samples(_to(rec, "cuda"))
        assert a[0] == b[0] and on_host.t == on_dev.t and on_host._buffer_full == on_dev._buffer_full
        for k, (x, y) in enumerate(zip(_leaves(on_host), _leaves(on_dev))):
            assert x.dtype == y.dtype and torch.equal(x, y), (n, T_new, k)
    assert _lib.variant_counts().get("replay_append_kernel", 0) == 13


@pytest.mark.gpu
def test_one_launch_append_with_n_step_returns_and_host_records():
    """3-step returns: the refresh after the fused write equals the refresh after the slice
    assignments (both on the device); records handed in as numpy / float64 are converted like the
    slice assignment converts them."""
    from rlpyt_amd.replays.buffers import UniformReplayFrameBuffer
    C, H, W, B, ring_T = 4, 104, 80, 16, 12
    kw = dict(example=_example(C, H, W), size=ring_T * B, B=B, discount=0.99, n_step_return=3,
              device="cuda")
    fused, parts = UniformReplayFrameBuffer(**kw), UniformReplayFrameBuffer(**kw)
    rng = np.random.RandomState(5)
    for n, T_new in enumerate((5, 5, 5, 2, 7, 12, 3)):
        rec = _record(rng, T_new, B, C, H, W)
        if n % 2:       # host numpy leaves, reward in float64
            rec = S2B(observation=rec.observation.numpy(), action=rec.action.numpy(),
                      reward=rec.reward.numpy().astype(np.float64), done=rec.done.numpy())
        fused.append_samples(rec)
        claim = parts.cursor.claim(T_new)                 # the parts, as append_samples ran them before
        parts.fields.write(type(parts.fields.data)(*(getattr(rec, f) for f in parts._stored_fields)),
                           claim)
        parts.frame_store.write(rec.observation, claim)
        for k, (x, y) in enumerate(zip(_leaves(fused), _leaves(parts))):
            assert torch.equal(x, y), (n, T_new, k)
