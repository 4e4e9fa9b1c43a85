"""The arithmetic of the towers' two-term fp16 split (csrc/orl_mlp.h, ORL_TOWER_F16) restated in numpy - what the HIP kernels
rely on, checked without a GPU:

* x = hi + lo with hi = rn16(x), lo = rn16(x - hi): the remainder x - hi is exact in fp32, and |x - hi - lo| <= 2^-22 |x| while
  both terms are normal fp16 numbers;
* three products hi.hi + hi.lo + lo.hi reproduce a.b to ~2^-22 |a||b| (the dropped lo.lo term);
* fp16's range is the catch: unscaled operands of ~1e-6 lose everything, the same operands scaled by a power of two chosen
  from their maximum (the kernels' `scale_exponent`: maximum -> [2^11, 2^12)) do not, and the scaling is exact;
* LayerNorm of a row scaled by 2^k with eps scaled by 4^k returns the same xhat (how fc2's scaled accumulators feed
  LayerNorm 2 without an unscaling pass).
The GPU-side evidence: tools/split_f16_gemm.hip (profiles/r06_split_f16_gemm.txt), tests/test_split_scaling_gpu.py."""
import numpy as np


def split2(x):
    x = np.asarray(x, np.float32)
    hi = x.astype(np.float16)
    rem = x - hi.astype(np.float32)  # exact in fp32
    lo = rem.astype(np.float16)
    return hi, lo, rem


def dot3(a, b):
    """sum_k of the three products, accumulated in float64 (the MFMA accumulates in fp32: its own rounding is not the subject)"""
    ah, al, _ = split2(a)
    bh, bl, _ = split2(b)
    f = lambda v: v.astype(np.float64)
    return f(ah) @ f(bh) + f(ah) @ f(bl) + f(al) @ f(bh)


def scale_exponent(mx):
    """biased exponent of a non-negative maximum, clamped as csrc/orl_mlp.h does"""
    e = int(np.float32(mx).view(np.uint32) >> 23) & 0xFF
    return min(max(e, 20), 254)


def test_remainder_is_exact_and_two_terms_carry_22_bits():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200000) * np.exp(rng.uniform(-2.0, 6.0, 200000))).astype(np.float32)  # |x| in ~[0.01, 4e3]
    hi, lo, rem = split2(x)
    assert np.array_equal(hi.astype(np.float64) + rem.astype(np.float64), x.astype(np.float64))  # x - hi exact
    ok = np.abs(x) >= 0.25  # lo is a normal fp16 number from here on (|lo| <= 2^-11 |x|, fp16's smallest normal is 2^-14)
    err = np.abs(x.astype(np.float64) - hi.astype(np.float64) - lo.astype(np.float64))
    assert (err[ok] <= 2.0 ** -22 * np.abs(x[ok])).all()
    # below that lo is a subnormal: the absolute error is what is bounded (half of fp16's smallest subnormal, 2^-25)
    assert (err[~ok] <= 2.0 ** -25 * (1 + 1e-6)).all()


def test_three_products_reach_fp32_accuracy():
    rng = np.random.default_rng(1)
    A = rng.standard_normal((64, 64)).astype(np.float32)
    B = rng.standard_normal((64, 256)).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64)
    got = dot3(A, B)
    fp32 = (A @ B).astype(np.float64)  # a plain fp32 GEMM, for scale
    rms = np.sqrt((ref ** 2).mean())
    e_split = np.sqrt(((got - ref) ** 2).mean()) / rms
    e_fp32 = np.sqrt(((fp32 - ref) ** 2).mean()) / rms
    assert e_split < 1.5e-7, e_split          # (measured on the MFMA: 8.0e-8)
    assert e_split < 3.0 * e_fp32 + 5e-8, (e_split, e_fp32)


def test_range_unscaled_gradients_are_lost_and_scaled_ones_are_not():
    rng = np.random.default_rng(2)
    W = (0.18 * rng.standard_normal((64, 64))).astype(np.float32)
    G = (1e-6 * rng.standard_normal((64, 16))).astype(np.float32)
    ref = W.astype(np.float64) @ G.astype(np.float64)
    rms = np.sqrt((ref ** 2).mean())
    bad = np.sqrt(((dot3(W, G) - ref) ** 2).mean()) / rms
    assert bad > 1e-4, bad  # fp16 subnormals: nothing left of the second term
    # the kernels' scales: the image's maximum to [2^13, 2^14), the tile's to [2^11, 2^12) - exact powers of two
    eb = int(np.float32(np.abs(W).max()).view(np.uint32) >> 23) & 0xFF
    kw = 13 - (eb - 127)
    ea = scale_exponent(np.abs(G).max())
    Ws, Gs = np.ldexp(W, kw), np.ldexp(G, 138 - ea)
    assert 2.0 ** 13 <= np.abs(Ws).max() < 2.0 ** 14 and 2.0 ** 11 <= np.abs(Gs).max() < 2.0 ** 12
    assert np.array_equal(np.ldexp(Ws, -kw), W) and np.array_equal(np.ldexp(Gs, ea - 138), G)  # exact both ways
    good = np.sqrt(((np.ldexp(dot3(Ws, Gs), -(kw + 138 - ea)) - ref) ** 2).mean()) / rms
    assert good < 1.5e-7, good


def test_a_value_far_below_its_tiles_maximum_degrades_gracefully():
    # one row of the tile 2^-20 below the largest: its own relative accuracy drops, the error relative to the tile's largest
    # output (what the sums over a tile's rows see) stays at the 2^-36 level
    rng = np.random.default_rng(3)
    W = rng.standard_normal((64, 64)).astype(np.float32)
    G = rng.standard_normal((64, 16)).astype(np.float32)
    G[:, 3] *= np.float32(2.0 ** -20)
    ea = scale_exponent(np.abs(G).max())
    Gs = np.ldexp(G, 138 - ea)
    ref = W.astype(np.float64) @ G.astype(np.float64)
    got = np.ldexp(dot3(W, Gs), ea - 138)
    col_max = np.sqrt((ref ** 2).mean(axis=0)).max()
    assert np.abs(got[:, 3] - ref[:, 3]).max() / col_max < 2.0 ** -33


def test_layernorm_of_a_scaled_row_with_scaled_eps():
    rng = np.random.default_rng(4)
    z = (3.0 * rng.standard_normal((16, 64)) + 1.5).astype(np.float32)

    def ln(x, eps):
        x = x.astype(np.float64)
        m = x.mean(-1, keepdims=True)
        v = ((x - m) ** 2).mean(-1, keepdims=True)
        return (x - m) / np.sqrt(v + eps), 1.0 / np.sqrt(v + eps)

    for k in (-12, 5, 14, 30):
        xh, rstd = ln(z, 1e-5)
        xh_s, rstd_s = ln(np.ldexp(z, k), 1e-5 * 4.0 ** k)
        np.testing.assert_allclose(xh_s, xh, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(rstd_s * 2.0 ** k, rstd, rtol=1e-12)  # rstd of the scaled row = the true one x 2^-k
