"""Round-6 variants of the bf16x6 GEMM kernels (csrc/gemm.hip, gemm_tn.hip; switch RLPYT_GEMM_EXP, read
per call): PAIR (bit 0: the rows of two K-16 steps requested together) and DOT2 (bit 1: the three-piece
split with v_dot2c_f32_bf16 residuals) change the schedule / the instructions, NOT the arithmetic --
every variant must return the bit pattern of the base kernel, and the split itself must stay exact:
``A @ I^T == A`` bit for bit on operands that stress the residuals (tiny, huge, signed zeros, values on
bf16 rounding ties)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class _exp:
    def __init__(self, v):
        self.v = v

    def __enter__(self):
        self.old = os.environ.get("RLPYT_GEMM_EXP")
        os.environ["RLPYT_GEMM_EXP"] = str(self.v)

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("RLPYT_GEMM_EXP", None)
        else:
            os.environ["RLPYT_GEMM_EXP"] = self.old


def _wide(shape, g):
    """f32 values over ~16 binades with both signs: residuals of every magnitude."""
    return (torch.randn(shape, generator=g) * torch.exp2(torch.randint(-8, 9, shape, generator=g).float())).cuda()


# (M, N, K): forward-like (128-row tiles), input-gradient-like (256-row tiles: ceil(M/256)*ceil(N/128)
# >= 512), ragged edges, the shortest contractions (K = 32, 64, 96: prologue / tail paths of PAIR)
NT_SHAPES = [(1024, 512, 3456), (8192, 3456, 512), (1000, 300, 96), (130, 140, 32), (257, 129, 64),
             (4200, 4000, 160)]


@pytest.mark.parametrize("shape", NT_SHAPES, ids=[str(s) for s in NT_SHAPES])
def test_gemm_nt_variants_are_bit_identical_to_the_base_kernel(shape):
    from rlpyt_amd import _lib, ops
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K)
    a, b = _wide((M, K), g), _wide((N, K), g)
    with _exp(0):
        base = ops.gemm_nt(a, b)
    ref = (a.double() @ b.double().t())
    scale = (a.double().abs() @ b.double().abs().t())
    assert float(((base.double() - ref).abs() / scale).max()) < 2e-5          # f32-level error (K sums)
    for exp in (1, 2, 3):
        _lib.variant_reset()
        with _exp(exp):
            out = ops.gemm_nt(a, b)
        torch.cuda.synchronize()
        ran = [k for k, v in _lib.variant_counts().items() if v > 0 and "gemm_nt_x6_kernel" in k]
        want = f"{'true' if exp & 1 else 'false'}, {'true' if exp & 2 else 'false'}>"
        assert len(ran) == 1 and ran[0].replace(" ", "").endswith(want.replace(" ", "")), (exp, ran)
        assert torch.equal(out, base), (exp, float((out - base).abs().max()))


@pytest.mark.parametrize("shape", [(512, 3456, 8192), (64, 96, 2048), (132, 260, 4096), (40, 36, 64)],
                         ids=str)
def test_gemm_tn_dot2_is_bit_identical_to_the_base_kernel(shape):
    from rlpyt_amd import ops
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K)
    a, b = _wide((K, M), g), _wide((K, N), g)
    with _exp(0):
        base = ops.gemm_tn(a, b)
    with _exp(2):
        out = ops.gemm_tn(a, b)
    assert torch.equal(out, base), float((out - base).abs().max())


@pytest.mark.parametrize("exp", [0, 1, 2, 3])
def test_split_is_exact_identity_contraction(exp):
    """``A @ I^T``: the pieces of I are (1, 0, 0), so every output is lo + mid + hi of one input in the
    kernel's own summation order -- equal to the input bit for bit iff the three pieces are exact."""
    from rlpyt_amd import ops
    K = 256
    g = torch.Generator().manual_seed(9)
    # sign x (1 + u) x 2^e over 190 binades; every piece (down to 2^-24 of the value) stays a NORMAL
    # number -- what the matrix pipe does with subnormal bf16 inputs is not what this test is about
    e = torch.randint(-90, 101, (320, K), generator=g).float()
    sgn = torch.where(torch.rand(320, K, generator=g) < 0.5, -1.0, 1.0)
    rnd = sgn * (1.0 + torch.rand(320, K, generator=g)) * torch.exp2(e)
    # bf16 rounding ties and values just beside them, signed zeros
    base = torch.tensor([1.0, 1.00390625, 1.0078125, 1.01171875, 0.99609375, 257.0, 1.0 + 2 ** -9,
                         1.0 + 2 ** -8 + 2 ** -17, 1.0 - 2 ** -9, 1.0 + 2 ** -16, 1.0 - 2 ** -17, 0.0, -0.0,
                         65535.99609375, 255.998046875, 1.0 + 2 ** -23])
    ties = torch.cat([base, -base]).repeat(K // 32 + 1)[:K].reshape(1, K).repeat(64, 1)
    ties = ties * torch.exp2(torch.arange(64).float() - 40).reshape(64, 1)
    a = torch.cat([rnd, ties]).cuda().contiguous()
    assert bool(torch.isfinite(a).all()) and a.shape[0] == 384
    eye = torch.eye(K, device="cuda")
    with _exp(exp):
        out = ops.gemm_nt(a, eye)
    # (flushed subnormal RESULTS would show here too: the inputs hold values down to 2^-126)
    same = out.view(torch.int32) == a.view(torch.int32)
    zero_sign = (out == 0) & (a == 0)                    # -0 may come back as +0 from the accumulator
    bad = ~(same | zero_sign)
    assert not bool(bad.any()), (int(bad.sum()), a[bad][:8].tolist(), out[bad][:8].tolist())
    b = a[:K].t().contiguous()                           # ... and as the B operand (N = K rows)
    with _exp(exp):
        out_b = ops.gemm_nt(eye, b.t().contiguous())
    a_b = b.t().contiguous().t()
    same = out_b.view(torch.int32) == a_b.contiguous().view(torch.int32)
    assert bool((same | ((out_b == 0) & (a_b == 0))).all())
