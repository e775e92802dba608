"""wgsl_builtins.py - the transcendental built-ins the WGSL interpreter (oracle/wgsl_exec.py) calls.  TEST INFRASTRUCTURE ONLY.

WGSL leaves the precision of sin / cos / tan / acos / atan2 / pow to the implementation; the numerics contract fixes them as
PORTABLE FORMS, specified to the bit (argument reductions and polynomial coefficients below).  The product computes them in
bhusie_amd/csrc/bhray_math.h, the C oracle and the NumPy restatement (oracle/np_ray.py, array code) each restate them.  This module is a
THIRD, separately written transcription - scalar code, one numpy.float32 operation per line of the specification - so that the interpreter
which executes the reference's shader text shares no function with the restatements it is used to pin: until round 4 it imported
np_ray's, and a transcription slip there would have gone into the executed-shader frames AND into the restatement compared with them.

Two checks keep it honest (tests/test_wgsl_builtins.py):
  - it agrees bit for bit with np_ray's forms and with the C oracle's exported forms on dense samples of their domains, and
  - every form lies within a stated distance of the binary64 value of the function on the same binary32 arguments (`within_spec`: acos
    1.5 ulp, atan2 3.5 ulp, pow(x, -0.001) 1 ulp, sin / cos 1e-7 absolute for |x| < 8192; measured maxima in SPEC's comment) - i.e. each
    is an accurate, legal evaluation of the WGSL built-in, not merely a function three files agree on.
"""
from __future__ import annotations

import math
import struct

import numpy as np

F = np.float32
_NAN = F(np.nan)


def _bits(x) -> int:
    return struct.unpack("<I", struct.pack("<f", float(x)))[0]


def _from_bits(u: int):
    return F(struct.unpack("<f", struct.pack("<I", u & 0xFFFFFFFF))[0])


def _horner(coeffs, z):
    """((c0 * z + c1) * z + c2) ... : one rounded product and one rounded sum per coefficient"""
    p = F(coeffs[0])
    for c in coeffs[1:]:
        p = F(F(p * z) + F(c))
    return p


# ---- pow(x, -0.001)  (ray.wgsl:459): exp(-0.001 * ln x); ln x = e * ln 2 + 2 atanh((m - 1) / (m + 1)), m in (sqrt(1/2), sqrt(2)]
def pow_m001(x):
    x = F(x)
    if x != x or x < F(0.0):
        return _NAN
    if x == F(0.0):
        return F(np.inf)
    if np.isinf(x):
        return F(0.0)
    u = _bits(x)
    e = (u >> 23) - 127
    if (u >> 23) == 0:                                       # denormal: scaled by 2^23 first
        x = F(x * F(8388608.0))
        u = _bits(x)
        e = (u >> 23) - 127 - 23
    m = _from_bits((u & 0x007FFFFF) | 0x3F800000)
    if m > F(1.41421354):
        m = F(m * F(0.5))
        e += 1
    s = F(F(m - F(1.0)) / F(m + F(1.0)))
    s2 = F(s * s)
    p = _horner((0.111111112, 0.142857149, 0.2, 0.333333343, 1.0), s2)
    lnm = F(F(F(2.0) * s) * p)
    lnx = F(F(F(e) * F(0.693147182)) + lnm)
    t = F(F(-0.001) * lnx)
    return _horner((0.00138888892, 0.00833333377, 0.0416666679, 0.166666672, 0.5, 1.0, 1.0), t)


# ---- acos (ray.wgsl:266): asin on [-1/2, 1/2] by an odd polynomial, the outer thirds by the half-angle identity
def _asin_core(z):
    z2 = F(z * z)
    p = _horner((4.2163199048e-2, 2.4181311049e-2, 4.5470025998e-2, 7.4953002686e-2, 1.6666752422e-1), z2)
    return F(z + F(F(z * z2) * p))


def acos(x):
    x = F(x)
    if x != x or x > F(1.0) or x < F(-1.0):
        return _NAN
    if x > F(0.5):
        return F(F(2.0) * _asin_core(np.sqrt(F(F(F(1.0) - x) * F(0.5)))))
    if x < F(-0.5):
        return F(F(3.14159274) - F(F(2.0) * _asin_core(np.sqrt(F(F(F(1.0) + x) * F(0.5))))))
    return F(F(1.57079637) - _asin_core(x))


# ---- atan2 (ray.wgsl:257-258, 632): atan of min/max in [0, 1], folded at tan(pi/8), then the octant
def atan2(y, x):
    y, x = F(y), F(x)
    ax, ay = F(abs(x)), F(abs(y))
    big, small = (ay, ax) if ax < ay else (ax, ay)
    a = F(0.0) if big == F(0.0) else F(small / big)
    t, base = a, F(0.0)
    if a > F(0.414213568):
        t = F(F(a - F(1.0)) / F(a + F(1.0)))
        base = F(0.785398185)
    z = F(t * t)
    p = F(8.05374449538e-2)
    p = F(F(p * z) - F(1.38776856032e-1))
    p = F(F(p * z) + F(1.99777106478e-1))
    p = F(F(p * z) - F(3.33329491539e-1))
    r = F(base + F(F(F(p * z) * t) + t))
    if ay > ax:
        r = F(F(1.57079637) - r)
    if x < F(0.0):
        r = F(F(3.14159274) - r)
    return F(-r) if (_bits(y) >> 31) else r


# ---- sin / cos (ray.wgsl:634): Cody-Waite reduction by pi/4 in three parts, the octant picks the polynomial and the sign
def _sincos(xin, want_cos: bool):
    xin = F(xin)
    x = F(abs(xin))
    negate = False if want_cos else bool(_bits(xin) >> 31)
    if not (x <= F(3.0e9)):
        return _NAN
    j = int(F(x * F(1.27323954)))                            # truncation toward zero, as the u32 conversion
    j += j & 1
    y = F(j)
    x = F(F(F(x - F(y * F(0.78515625))) - F(y * F(2.4187564849853515625e-4))) - F(y * F(3.77489497744594108e-8)))
    j &= 7
    if j > 3:
        negate = not negate
        j -= 4
    if want_cos and j > 1:
        negate = not negate
    z = F(x * x)
    middle = j in (1, 2)
    if middle != want_cos:                                   # the cosine polynomial: sin in the middle octants, cos in the outer ones
        p = F(2.443315711809948e-5)
        p = F(F(p * z) - F(1.388731625493765e-3))
        p = F(F(p * z) + F(4.166664568298827e-2))
        r = F(F(F(F(p * z) * z) - F(F(0.5) * z)) + F(1.0))
    else:
        p = F(-1.9515295891e-4)
        p = F(F(p * z) + F(8.3321608736e-3))
        p = F(F(p * z) - F(1.6666654611e-1))
        r = F(F(F(p * z) * x) + x)
    return F(-r) if negate else r


def sin(x):
    return _sincos(x, False)


def cos(x):
    return _sincos(x, True)


def tan(x):                                                  # ray.wgsl:279: sin / cos, one rounded quotient
    return F(sin(x) / cos(x))


# ---- textureSampleLevel(t, s, uv, 0) of the reference's sampler (src/renderer/texture.rs:32,61-69): RGBA8 unorm (byte / 255), linear
# min / mag filter, clamp-to-edge, one mip; texel centres at +0.5.  mix(a, b, t) = a * (1 - t) + b * t (N5), x first, then y.
def _axis(t, n: int):
    x = F(F(F(t) * F(n)) - F(0.5))
    if not (x >= F(-1.0)):                                   # also NaN
        x = F(-1.0)
    if x > F(n):
        x = F(n)
    fl = np.floor(x)
    i0 = int(fl)
    frac = F(x - fl)
    return min(max(i0, 0), n - 1), min(max(i0 + 1, 0), n - 1), frac


def sample_bilinear(rgba8, u, v):
    h, w = rgba8.shape[0], rgba8.shape[1]
    x0, x1, fx = _axis(u, w)
    y0, y1, fy = _axis(v, h)
    out = []
    for ch in range(4):
        a, b = F(F(rgba8[y0, x0, ch]) / F(255.0)), F(F(rgba8[y0, x1, ch]) / F(255.0))
        c, d = F(F(rgba8[y1, x0, ch]) / F(255.0)), F(F(rgba8[y1, x1, ch]) / F(255.0))
        top = F(F(a * F(F(1.0) - fx)) + F(b * fx))
        bot = F(F(c * F(F(1.0) - fx)) + F(d * fx))
        out.append(F(F(top * F(F(1.0) - fy)) + F(bot * fy)))
    return tuple(out)


# ---- how far each form is from the correctly rounded binary64 value
def _ulp_distance(got, exact: float) -> float:
    got = float(got)
    if exact == 0.0:
        return abs(got) / float(np.finfo(np.float32).tiny)
    spacing = float(np.spacing(F(abs(exact))))
    return abs(got - exact) / spacing


# asserted bounds; measured maxima over 20 000 seeded samples each: acos 1.24 ulp, atan2 2.90 ulp, pow(x, -0.001) 0.74 ulp, sin / cos 7.2e-8 (|x| < 8000)
SPEC = {"acos": ("ulp", 1.5), "atan2": ("ulp", 3.5), "pow_m001": ("ulp", 1.0), "sin": ("abs", 1.0e-7), "cos": ("abs", 1.0e-7)}


def within_spec(name: str, args, got) -> bool:
    """True when `got` (this module's result for `args`) lies within the accuracy DESIGN.md N4 states for the form, measured against
    the same function evaluated in binary64 on the binary32 arguments."""
    a = [float(F(v)) for v in args]
    exact = {"acos": lambda: math.acos(a[0]), "atan2": lambda: math.atan2(a[0], a[1]), "pow_m001": lambda: a[0] ** -0.001,
             "sin": lambda: math.sin(a[0]), "cos": lambda: math.cos(a[0])}[name]()
    kind, bound = SPEC[name]
    if kind == "abs":
        return abs(float(got) - exact) <= bound
    return _ulp_distance(got, exact) <= bound
