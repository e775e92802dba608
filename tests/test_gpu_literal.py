"""-m gpu: how far is the HIP path from the LITERAL text of ray.wgsl?

The default kernels implement the numerics contract of DESIGN.md §2 (fused multiply-add, reassociation and small-integer powers
in the integrator: N3/N7/N9/N10 — evaluations WGSL permits).  Every other -m gpu test compares them with the oracle under the
SAME contract.  This file pins the distance to the shader text itself, three ways:

 1. BHRAY_F_LITERAL — a trace-kernel variant whose integrator is the shader text operator by operator — must reproduce, bit for
    bit on every direction pixel, (a) the frozen literal fixtures tests/golden/frames_literal.npz (written from the WGSL by the
    NumPy restatement, never regenerated when the contract changes) and (b) the C oracle's literal mode at full frame size.
 2. The DEFAULT kernels against the literal oracle at the bench frame (1920x1080, both integrators): identical pixel classes, a
    bounded median, and >= 99.7 % of the pixels inside the 1e-4 bar of BASELINE.json's north_star.  The remainder are the chaotic
    rays (photon sphere, disk edge) on which any two conforming evaluations disagree; their maximum is recorded, not bounded.
 3. The default kernels against the literal KERNEL (GPU vs GPU): the same statistics without any CPU code in the loop.
"""
import json
import os

import numpy as np
import pytest

import bhusie_amd as B
from oracle import oracle as O
from tests import common as T

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LITERAL_CASES = ["euler_l0", "rk_l0", "rk_ladder", "euler_ladder", "rk_outside", "rk_off_origin"]
REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "literal_distance.jsonl")


def _gpu(cfg, u, tex, **kw):
    rp = B.RayPass(cfg, device=0, **kw)
    rp.set_textures(*tex)
    rp.set_uniforms(*u)
    rp.render()
    return rp


def _same_to_the_bit_where_specified(got, want, what, colour_tol=1e-4):
    assert got.shape == want.shape
    ng, nw = np.isnan(got).any(axis=-1), np.isnan(want).any(axis=-1)
    assert np.array_equal(ng, nw), f"{what}: NaN pixels in different places"
    ok = ~nw
    assert np.array_equal(got[..., 3][ok], want[..., 3][ok]), f"{what}: pixel classes differ"
    d = ok & (want[..., 3] == 0)
    assert np.array_equal(got[d].view(np.uint32), want[d].view(np.uint32)), f"{what}: direction pixels are not bit-identical"
    e = np.abs(got[ok] - want[ok]) / np.maximum(np.abs(want[ok]), T.ABS_FLOOR)
    assert float(e.max(initial=0.0)) <= colour_tol, f"{what}: colour max rel {float(e.max())}"     # pow(., 1.3): device libm vs host libm
    return int(d.sum())


@pytest.mark.parametrize("name", LITERAL_CASES)
def test_literal_kernel_reproduces_the_frozen_literal_fixtures(name):
    g = np.load(os.path.join(GOLD, "frames_literal.npz"))
    tex = (g["t_temp"], g["t_disk"], g["t_sky"])
    u = (g[f"{name}.camera"].tobytes(), g[f"{name}.black_hole"].tobytes(), g[f"{name}.details"].tobytes())
    sizes = [tuple(int(v) for v in s) for s in g[f"{name}.sizes"]]
    cfg = B.ladder_from_base(sizes[0], 3, len(sizes))
    assert cfg.sizes() == sizes
    rp = _gpu(cfg, u, tex, literal=True, counters=True)
    n = 0
    for l in range(len(sizes)):
        n += _same_to_the_bit_where_specified(rp.read_level(l), g[f"{name}.level{l}"], f"literal {name} level {l}")
    assert n > 0
    traced, steps, copied, interp, sky = (int(v) for v in g[f"{name}.stats"])
    c = rp.counters()
    assert (c["traced"], c["steps"], c["copied"], c["interpolated"], c["sky_samples"]) == (traced, steps, copied, interp, sky)
    # and the default kernel is NOT this evaluation (else the variant would pin nothing)
    if name in ("rk_l0", "euler_l0"):
        dflt = _gpu(cfg, u, tex).read_hdr()
        assert not np.array_equal(dflt, rp.read_hdr())


@pytest.mark.parametrize("method", [1, 0])
def test_literal_kernel_equals_literal_oracle_at_1918x1081(method):
    """Reference-native ladder 72x41 -> 1918x1081, every pixel: the literal kernel against the C oracle's literal mode."""
    tex = T.textures(small=False)
    u = T.uniforms(integration_method=method)
    cfg = B.ladder_from_base((72, 41), 3, 4)
    rp = _gpu(cfg, u, tex, literal=True, counters=True)
    cnt = O.Counters()
    O.set_literal(True)
    try:
        want = O.render_ladder(T.oracle_scene(*u, tex), cfg.sizes(), cnt)
    finally:
        O.set_literal(False)
    for l in range(4):
        _same_to_the_bit_where_specified(rp.read_level(l), want[l], f"literal level {l} method {method}")
    assert rp.counters() == cnt.as_dict()


def _distance(got, want):
    same = got[..., 3] == want[..., 3]
    with np.errstate(invalid="ignore"):
        e = (np.abs(got - want) / np.maximum(np.abs(want), T.ABS_FLOOR))[same].max(axis=-1)
    e = e[np.isfinite(e)]
    return {"pixels": int(same.size), "class_differences": int((~same).sum()), "median_rel_err": float(np.median(e)),
            "p99_rel_err": float(np.quantile(e, 0.99)), "fraction_within_1e-4": float((e <= 1e-4).mean()),
            "pixels_beyond_1e-4": int((e > 1e-4).sum()), "max_rel_err": float(e.max())}


def _record(entry):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(json.dumps(entry) + "\n")
    print(json.dumps(entry))


@pytest.mark.parametrize("method", [1, 0])
def test_default_kernels_vs_the_literal_reading_at_the_bench_frame(method):
    """configs[1], 1920x1080 with the adaptive grid: the contract kernels against (a) the literal C oracle, (b) the literal
    kernel.  Bounds: 0 class differences, median < 1e-6, >= 99.7 % of the pixels within 1e-4; the maximum is recorded."""
    tex = T.textures(small=False)
    u = T.uniforms(integration_method=method)
    cfg = B.ladder_for_frame((1920, 1080), 3, 4)
    got = _gpu(cfg, u, tex).read_hdr()
    lit = _gpu(cfg, u, tex, literal=True).read_hdr()
    O.set_literal(True)
    try:
        want = O.render_ladder(T.oracle_scene(*u, tex), cfg.sizes())[-1]
    finally:
        O.set_literal(False)
    cx, cy = int(cfg.crop_x), int(cfg.crop_y)
    want = want[cy:cy + 1080, cx:cx + 1920]
    _same_to_the_bit_where_specified(lit, want, f"literal kernel vs literal oracle, 1920x1080 method {method}")
    for tag, ref in (("literal C oracle", want), ("literal kernel (BHRAY_F_LITERAL)", lit)):
        d = _distance(got, ref)
        d.update(config="1920x1080 " + ("adaptive RK" if method else "Euler"), default_kernels_vs=tag)
        _record(d)
        assert d["class_differences"] == 0, d
        assert d["median_rel_err"] < 1e-6, d
        assert d["fraction_within_1e-4"] >= 0.997, d
