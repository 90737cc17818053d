"""CPU check of the arithmetic claim behind the bf16-split kernels (csrc/conv.hip,
csrc/gemm.hip): an f32 value splits EXACTLY into three bf16 pieces, products of pieces are exact in
f32, and the six products of order <= 2 miss the exact product by at most 2^-24 |ab| -- one f32 rounding -- and 2^-27 rms (bf16x6), while
a uint8 operand needs no split at all (bf16x3 is exact).  numpy emulation of round-to-nearest-even
bf16; no GPU, no kernel -- this pins the numerics the kernels rely on, not their implementation."""
import numpy as np


def bf16_rn(x):
    """float32 -> float32 holding the round-to-nearest-even bf16 value."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def bf16_trunc(x):
    u = np.asarray(x, dtype=np.float32).view(np.uint32) & np.uint32(0xFFFF0000)
    return u.view(np.float32)


def split3(x, rnd):
    x = np.asarray(x, dtype=np.float32)
    hi = rnd(x)
    r1 = (x - hi).astype(np.float32)
    mid = rnd(r1)
    r2 = (r1 - mid).astype(np.float32)
    lo = rnd(r2)
    return hi, mid, lo, (r2 - lo).astype(np.float32)


def _wide(rng, n):
    return (rng.standard_normal(n) * np.exp(3 * rng.standard_normal(n))).astype(np.float32)


def test_three_bf16_pieces_are_exact():
    rng = np.random.default_rng(0)
    x = _wide(rng, 200000)
    for rnd in (bf16_rn, bf16_trunc):
        hi, mid, lo, rest = split3(x, rnd)
        assert np.all(rest == 0)                                   # nothing left after three pieces
        s = hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64)
        assert np.all(s == x.astype(np.float64))                   # hi + mid + lo == x exactly
        assert np.all(np.abs(mid) <= np.abs(x) * 2.0 ** -7)
        assert np.all(np.abs(lo) <= np.abs(x) * 2.0 ** -15)
    hi, mid, lo, _ = split3(x, bf16_rn)                            # round to nearest: one bit better
    assert np.all(np.abs(mid) <= np.abs(x) * 2.0 ** -8)
    assert np.all(np.abs(lo) <= np.abs(x) * 2.0 ** -16)


def test_piece_products_are_exact_in_f32_and_bytes_need_no_split():
    rng = np.random.default_rng(1)
    a, b = _wide(rng, 50000), _wide(rng, 50000)
    pa, pb = split3(a, bf16_rn)[:3], split3(b, bf16_rn)[:3]
    for x in pa:
        for y in pb:
            exact = x.astype(np.float64) * y.astype(np.float64)
            f32 = (x * y).astype(np.float32)
            ok = (np.abs(exact) < 1e-37) | (np.abs(exact) > 1e37) | (f32.astype(np.float64) == exact)
            assert np.all(ok)                                      # 8 x 8 significand bits fit in 24
    byte = np.arange(256, dtype=np.float32)
    assert np.all(bf16_rn(byte) == byte) and np.all(bf16_trunc(byte) == byte)
    # bf16x3: byte x (hi + mid + lo) summed exactly == byte x value
    w = _wide(rng, 256)
    hi, mid, lo, _ = split3(w, bf16_trunc)
    s = sum(byte.astype(np.float64) * p.astype(np.float64) for p in (hi, mid, lo))
    assert np.all(s == byte.astype(np.float64) * w.astype(np.float64))


def test_six_products_miss_at_most_2_pow_minus_26():
    rng = np.random.default_rng(2)
    a, b = _wide(rng, 200000), _wide(rng, 200000)
    a0, a1, a2, _ = split3(a, bf16_rn)
    b0, b1, b2, _ = split3(b, bf16_rn)
    d = np.float64
    six = (a0.astype(d) * b0 + a0.astype(d) * b1 + a1.astype(d) * b0 +
           a0.astype(d) * b2 + a2.astype(d) * b0 + a1.astype(d) * b1)
    exact = a.astype(d) * b.astype(d)
    rel = np.abs(six - exact) / np.abs(exact)
    assert rel.max() <= 2.0 ** -24, rel.max()      # worst case of the bound (|x1| <= 2^-8, |x2| <= 2^-16)
    assert np.median(rel) <= 2.0 ** -28            # typical
    # a dot product of 3456 terms through the six products stays below f32-accumulation noise
    K = 3456
    x, y = _wide(rng, K), _wide(rng, K)
    x0, x1, x2, _ = split3(x, bf16_rn)
    y0, y1, y2, _ = split3(y, bf16_rn)
    dot6 = (x0.astype(d) * y0 + x0.astype(d) * y1 + x1.astype(d) * y0 +
            x0.astype(d) * y2 + x2.astype(d) * y0 + x1.astype(d) * y1).sum()
    ref = (x.astype(d) * y.astype(d)).sum()
    scale = (np.abs(x.astype(d) * y.astype(d))).sum()
    assert abs(dot6 - ref) <= scale * 2.0 ** -26
