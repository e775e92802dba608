"""CPU: the interpreter's own transcendental forms and sampler (oracle/wgsl_builtins.py - a third, separately written transcription of
the portable forms of DESIGN.md N4) against (a) the two restatements' forms, bit for bit, and (b) binary64: each form must lie within
its stated accuracy of the true function, i.e. be a legal evaluation of the WGSL built-in and not merely what three files agree on."""
import numpy as np
import pytest

from oracle import np_ray as N
from oracle import oracle as O
from oracle import wgsl_builtins as WB

F = np.float32


def bits(x):
    return np.asarray(x, dtype=np.float32).view(np.uint32)


def samples(lo, hi, n, seed):
    rng = np.random.default_rng(seed)
    a = rng.uniform(lo, hi, size=n).astype(np.float32)
    edge = np.array([lo, hi, 0.0, -0.0, 0.5, -0.5, 1.0, -1.0, 0.414213568, 1e-30, -1e-30, 1e-45], dtype=np.float32)
    return np.concatenate([a, edge[(edge >= lo) & (edge <= hi)]])


def test_acos_three_ways_and_within_its_bound():
    x = samples(-1.0, 1.0, 20000, 1)
    mine = np.array([WB.acos(v) for v in x], dtype=np.float32)
    assert np.array_equal(bits(mine), bits(N.bh_acos(x)))
    assert all(bits(O.acos(float(v))) == bits(m) for v, m in zip(x[:2000], mine[:2000]))
    assert all(WB.within_spec("acos", (v,), m) for v, m in zip(x, mine))
    assert np.isnan(WB.acos(F(1.0000001))) and np.isnan(WB.acos(F(np.nan)))


def test_atan2_two_ways_and_within_its_bound():
    y, x = samples(-30.0, 30.0, 20000, 2), samples(-30.0, 30.0, 20000, 3)
    mine = np.array([WB.atan2(a, b) for a, b in zip(y, x)], dtype=np.float32)
    assert np.array_equal(bits(mine), bits(N.bh_atan2(y, x)))
    assert all(WB.within_spec("atan2", (a, b), m) for a, b, m in zip(y, x, mine) if not (a == 0 and b == 0))


def test_sin_cos_tan_two_ways_and_within_their_bound():
    x = samples(-8000.0, 8000.0, 20000, 4)
    s = np.array([WB.sin(v) for v in x], dtype=np.float32)
    c = np.array([WB.cos(v) for v in x], dtype=np.float32)
    assert np.array_equal(bits(s), bits(N.bh_sin(x))) and np.array_equal(bits(c), bits(N.bh_cos(x)))
    assert all(WB.within_spec("sin", (v,), m) for v, m in zip(x, s)) and all(WB.within_spec("cos", (v,), m) for v, m in zip(x, c))
    t = np.array([WB.tan(v) for v in x[:500]], dtype=np.float32)
    with np.errstate(all="ignore"):
        assert np.array_equal(bits(t), bits(N.bh_sin(x[:500]) / N.bh_cos(x[:500])))
    assert np.isnan(WB.sin(F(4e9)))


def test_pow_m001_three_ways_and_within_its_bound():
    rng = np.random.default_rng(5)
    x = np.concatenate([np.exp(rng.uniform(-80, 80, size=20000)).astype(np.float32), np.array([1e-45, 1e-38, 2e-5, 1.0, 3e38], dtype=np.float32)])
    mine = np.array([WB.pow_m001(v) for v in x], dtype=np.float32)
    assert np.array_equal(bits(mine), bits(N.bh_pow_m001(x)))
    assert all(bits(O.pow_m001(float(v))) == bits(m) for v, m in zip(x[:2000], mine[:2000]))
    assert all(WB.within_spec("pow_m001", (v,), m) for v, m in zip(x, mine))
    assert WB.pow_m001(F(np.inf)) == 0.0 and np.isinf(WB.pow_m001(F(0.0))) and np.isnan(WB.pow_m001(F(-1.0)))


def test_sampler_two_ways():
    rng = np.random.default_rng(6)
    tex = rng.integers(0, 256, size=(7, 11, 4), dtype=np.uint8)
    uv = np.concatenate([rng.uniform(-0.3, 1.3, size=(3000, 2)), [[0.0, 0.0], [1.0, 1.0], [0.5 / 11, 0.5 / 7], [np.nan, 0.2]]]).astype(np.float32)
    for u, v in uv:
        a = np.array(WB.sample_bilinear(tex, u, v), dtype=np.float32)
        b = N.sample_bilinear(tex, np.array([u], dtype=np.float32), np.array([v], dtype=np.float32))[0]
        assert np.array_equal(bits(a), bits(b)), (u, v, a, b)
