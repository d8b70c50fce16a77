"""The split3 record of csrc/gemm_s3.h restated in numpy: an f32 value as three bf16 parts (round to nearest even),
x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1).  What the f32x3 kernels rely on, checked on the host:
  * both subtractions are exact in f32 and the three parts carry a normal f32 value EXACTLY ((p0 + p1) + p2 == x bit
    for bit, the order unsplit_s3 adds them in) -- so a tensor of records holds the same information as the f32 tensor
    the learner's backward pass reads;
  * the six products kept by the kernels (i + j <= 2) leave out at most 2^-24 |x w| per product.
(The device round trip over 138 M values runs in tools/ubench/s3_probe.hip: `split_roundtrip_mismatches`: 0.)"""
import numpy as np


def bf16_rne(x):
    """float32 array -> the nearest bf16 (ties to even) as float32"""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32)


def split3(x):
    x = np.ascontiguousarray(x, np.float32)
    p0 = bf16_rne(x)
    r1 = (x - p0).astype(np.float32)
    p1 = bf16_rne(r1)
    r2 = (r1 - p1).astype(np.float32)
    p2 = bf16_rne(r2)
    return p0, p1, p2, r1, r2


def _samples():
    rng = np.random.default_rng(5)
    xs = [rng.normal(0, 1, 200000), rng.normal(0, 1e-3, 50000), rng.uniform(0, 300, 50000),
          np.exp2(rng.integers(-40, 40, 20000)) * (1 + rng.integers(0, 1 << 23, 20000) / float(1 << 23))]
    edge = []
    for e in range(-20, 21):  # just below / above a power of two, all-ones mantissas, ties
        b = np.float32(2.0) ** e
        edge += [b, np.nextafter(b, np.float32(0)), np.nextafter(b, np.float32(np.inf)), b * np.float32(1.5),
                 b * np.float32(1.00390625), b * np.float32(1.99609375), b * np.float32(1 + 2.0 ** -9),
                 b * np.float32(1 + 2.0 ** -9 + 2.0 ** -23), b * np.float32(2 - 2.0 ** -23)]
    xs.append(np.array(edge, np.float64))
    x = np.concatenate(xs).astype(np.float32)
    return np.concatenate([x, -x, np.zeros(3, np.float32)])


def test_three_parts_carry_an_f32_exactly():
    x = _samples()
    p0, p1, p2, r1, r2 = split3(x)
    # the residuals are exact in f32 (checked in f64, where x - p0 is certainly exact)
    assert np.array_equal(r1.astype(np.float64), x.astype(np.float64) - p0.astype(np.float64))
    assert np.array_equal(r2.astype(np.float64), r1.astype(np.float64) - p1.astype(np.float64))
    assert np.array_equal(p2, r2), "the third residual is itself a bf16 number: nothing is left over"
    back = ((p0 + p1).astype(np.float32) + p2).astype(np.float32)  # unsplit_s3's order, each add rounded to f32
    assert np.array_equal(back.view(np.uint32), x.view(np.uint32))


def test_six_products_drop_less_than_an_f32_rounding():
    rng = np.random.default_rng(6)
    x = np.abs(rng.normal(0, 1, 100000)).astype(np.float32)
    w = rng.normal(0, 0.05, 100000).astype(np.float32)
    xp, wp = split3(x)[:3], split3(w)[:3]
    exact = x.astype(np.float64) * w.astype(np.float64)
    six = sum(xp[i].astype(np.float64) * wp[j].astype(np.float64) for i in range(3) for j in range(3) if i + j <= 2)
    err = np.abs(six - exact)
    assert (err <= np.abs(exact) * 2.0 ** -24).all()
    assert err.max() > 0  # (the three dropped terms are really dropped)
