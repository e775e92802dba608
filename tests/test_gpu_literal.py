"""-m gpu: how far is the HIP path from the LITERAL text of ray.wgsl - and is that distance a property of the kernel or of the problem?

The default kernels implement the numerics contract of DESIGN.md §2 (fused multiply-add, reassociation and small-integer powers
in the integrator: N3/N7/N9/N10 — evaluations WGSL permits).  Every other -m gpu test compares them with the oracle under the
SAME contract.  This file pins the distance to the shader text itself, three ways:

 1. BHRAY_F_LITERAL — a trace-kernel variant whose integrator is the shader text operator by operator — must reproduce, bit for
    bit on every direction pixel, (a) tests/golden/wgsl_exec.npz: frames written by EXECUTING the reference's own ray.wgsl
    (oracle/wgsl_exec.py, an interpreter of the shader text), (b) the frozen literal fixtures tests/golden/frames_literal.npz (written
    from the WGSL by the NumPy restatement, never regenerated when the contract changes) and (c) the C oracle's literal mode at full
    frame size.
 2. The DEFAULT kernels against the literal oracle at the bench frame (1920x1080, both integrators): identical pixel classes, a
    bounded median, and >= 99.7 % of the pixels inside the 1e-4 bar of BASELINE.json's north_star.  The remainder are the chaotic
    rays (photon sphere, disk edge) on which any two conforming evaluations disagree; their maximum is recorded, not bounded.
 3. The default kernels against the literal KERNEL (GPU vs GPU): the same statistics without any CPU code in the loop.
 4. A THIRD legal evaluation (BHRAY_F_EVAL_FMA: the text with fused multiply-add contraction only, no reassociation), pinned like
    the literal one (frozen NumPy-written fixtures, the C oracle's eval mode 2), and the literal kernel with ONE INPUT CHANGED BY ONE
    ULP (the camera's z).  The pixels beyond 1e-4 of contract-vs-literal, fma-vs-literal and literal'-vs-literal are the same
    population: same size (within a small factor), mostly the same pixels.  Not even the literal evaluation is within 1e-4 of itself
    when an input moves by an ulp, so no bound tighter than "the same population as a one-ulp perturbation" can be asked of ANY
    implementation that is not bit-identical to the text.  Most of that population is the metric, not the ray: a direction pixel is a
    unit vector carrying ~1e-6 of accumulated rounding (245 steps), which is > 1e-4 RELATIVE in any channel that is near zero;
    measured against the vector's norm the direction-pixel population shrinks 20-fold, the whole population 4-5-fold (recorded, bounded).
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


WGSL_CASES = ["euler_l0", "rk_l0", "rk_ladder", "euler_ladder", "rk_outside", "rk_off_origin", "rk_highlight", "euler_tight", "rk_mesh", "euler_mesh", "rk_mesh_near", "euler_mesh_near"] + [f"fuzz{k:02d}" for k in range(24)]


@pytest.mark.parametrize("name", WGSL_CASES)
def test_literal_kernel_reproduces_frames_made_by_executing_the_shader_text(name, tmp_path):
    """tests/golden/wgsl_exec.npz: frames written by running the reference's own ray.wgsl through oracle/wgsl_exec.py (an interpreter;
    nothing restated).  The literal kernel: same classes, direction pixels bit for bit, colours within the device-pow tolerance."""
    g = np.load(os.path.join(GOLD, "wgsl_exec.npz"))
    tex = (g["t_temp"], g["t_disk"], g["t_sky"])
    u = (g[f"{name}.camera"].tobytes(), g[f"{name}.black_hole"].tobytes(), g[f"{name}.details"].tobytes())
    sizes = [tuple(int(v) for v in s) for s in g[f"{name}.sizes"]]
    cfg = B.ladder_from_base(sizes[0], 3, len(sizes))
    assert cfg.sizes() == sizes
    rp = B.RayPass(cfg, device=0, literal=True)
    rp.set_textures(*tex)
    if int(g[f"{name}.mesh"][0]):
        p = tmp_path / "m.obj"; p.write_bytes(g["mesh.obj"].tobytes())
        rp.upload_model(B.load_model(str(p)))
    rp.set_uniforms(*u)
    rp.render()
    n = 0
    for l in range(len(sizes)):
        n += _same_to_the_bit_where_specified(rp.read_level(l), g[f"{name}.level{l}"], f"executed shader, {name} level {l}")
    assert n > 0 or name.endswith("mesh") or name == "euler_tight" or name.startswith("fuzz")
    if f"{name}.sky" in g.files:         # sky.wgsl executed by the same interpreter over the executed frame: the resolve kernel on the literal frame
        rp.resolve_sky()
        sky, want = rp.read_sky().view(np.uint16), g[f"{name}.sky"]
        direction = g[f"{name}.level{len(sizes) - 1}"][..., 3] == 0
        assert direction.sum() > 100 and np.array_equal(sky[direction], want[direction]), f"{name}: resolved direction pixels"
        assert ((sky == want).all(axis=-1)).mean() > 0.99           # colour pixels pass through: device pow(., 1.3) is 1-2 f32 ulp from glibc's, usually the same binary16


def test_default_kernels_against_the_executed_shader_text(tmp_path):
    """The SHIPPED evaluation (contract: FMA + reassociation in the integrator) against frames made by executing the reference's shader
    text, all 71 frames of tests/golden/wgsl_exec.npz (111 k pixels): every pixel class and every NaN pixel identical; the pixels beyond
    north_star's 1e-4 are the population section 4 of this file's header describes (measured on MI355X: 295 per channel = 0.26 %, 159
    against the pixel's norm = 0.14 %, worst 1.1e-2: profiles/r03_wgsl_default_distance.jsonl) - bounded here at twice that."""
    g = np.load(os.path.join(GOLD, "wgsl_exec.npz"))
    tex = (g["t_temp"], g["t_disk"], g["t_sky"])
    p = tmp_path / "m.obj"; p.write_bytes(g["mesh.obj"].tobytes())
    pixels = beyond_ch = beyond_norm = 0
    worst = 0.0
    for name in WGSL_CASES:
        u = (g[f"{name}.camera"].tobytes(), g[f"{name}.black_hole"].tobytes(), g[f"{name}.details"].tobytes())
        sizes = [tuple(int(v) for v in s) for s in g[f"{name}.sizes"]]
        rp = B.RayPass(B.ladder_from_base(sizes[0], 3, len(sizes)), device=0)
        rp.set_textures(*tex)
        if int(g[f"{name}.mesh"][0]):
            rp.upload_model(B.load_model(str(p)))
        rp.set_uniforms(*u)
        rp.render()
        for l in range(len(sizes)):
            got, want = rp.read_level(l), g[f"{name}.level{l}"]
            nan = np.isnan(want).any(axis=-1)
            assert np.array_equal(np.isnan(got).any(axis=-1), nan), f"{name} level {l}: NaN pixels in different places"
            assert np.array_equal(got[..., 3][~nan], want[..., 3][~nan]), f"{name} level {l}: pixel classes differ from the executed shader"
            a, b = got[~nan][:, :3], want[~nan][:, :3]
            rel = (np.abs(a - b) / np.maximum(np.abs(b), T.ABS_FLOOR)).max(axis=-1)
            nrm = np.linalg.norm(a - b, axis=-1) / np.maximum(np.linalg.norm(b, axis=-1), T.ABS_FLOOR)
            pixels += len(a); beyond_ch += int((rel > 1e-4).sum()); beyond_norm += int((nrm > 1e-4).sum()); worst = max(worst, float(nrm.max(initial=0.0)))
    _record(dict(kind="default kernels vs executed shader text", pixels=pixels, beyond_1e4_per_channel=beyond_ch, beyond_1e4_of_norm=beyond_norm, worst_of_norm=worst))
    assert pixels > 110000
    assert beyond_ch <= 0.0052 * pixels and beyond_norm <= 0.0028 * pixels and worst < 0.05


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


def _per_channel(got, want):
    """max over channels of |got - want| / max(|want|, 1e-3): the metric of BASELINE.json's north_star ("1e-4 relative per channel")"""
    with np.errstate(invalid="ignore"):
        return (np.abs(got - want) / np.maximum(np.abs(want), T.ABS_FLOOR)).max(axis=-1)


def _per_pixel(got, want):
    """error relative to the PIXEL's magnitude: direction pixels ||dv|| / ||v||, colour pixels max|dc| / max(max|c|, 1e-3)"""
    with np.errstate(invalid="ignore", divide="ignore"):
        ang = np.linalg.norm((got - want)[..., :3], axis=-1) / np.linalg.norm(want[..., :3], axis=-1)
        col = np.abs(got - want)[..., :3].max(axis=-1) / np.maximum(np.abs(want[..., :3]).max(axis=-1), T.ABS_FLOOR)
    return np.where(want[..., 3] == 0, ang, col)


def _distance(got, want):
    same = got[..., 3] == want[..., 3]
    e = _per_channel(got, want)
    p = _per_pixel(got, want)
    ok = same & np.isfinite(e)
    ev, pv = e[ok], p[ok & np.isfinite(p)]
    return {"pixels": int(same.size), "class_differences": int((~same).sum()), "median_rel_err": float(np.median(ev)),
            "p99_rel_err": float(np.quantile(ev, 0.99)), "fraction_within_1e-4": float((ev <= 1e-4).mean()),
            "pixels_beyond_1e-4": int((ev > 1e-4).sum()), "max_rel_err": float(ev.max()),
            "per_pixel_norm": {"pixels_beyond_1e-4": int((pv > 1e-4).sum()), "fraction_within_1e-4": float((pv <= 1e-4).mean()),
                               "median": float(np.median(pv)), "max": float(pv.max())}}


def _beyond(got, want):
    """pixel masks: beyond 1e-4 per channel / per pixel norm (class differences count as beyond)"""
    same = got[..., 3] == want[..., 3]
    with np.errstate(invalid="ignore"):
        a = ~same | (_per_channel(got, want) > 1e-4)
        b = ~same | (_per_pixel(got, want) > 1e-4)
    return a, b


def _record(entry):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(json.dumps(entry) + "\n")
    print(json.dumps(entry))


@pytest.mark.parametrize("method", [1, 0])
def test_default_kernels_vs_the_literal_reading_at_the_bench_frame(method):
    """configs[1], 1920x1080 with the adaptive grid: the contract kernels against (a) the literal C oracle, (b) the literal
    kernel.  Bounds: 0 class differences, median < 1e-6; the fraction within 1e-4 is recorded and bounded RELATIVE to what a
    one-ulp change of one input does to the literal evaluation itself (test_the_outliers_are_the_problems_not_the_kernels)."""
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
        assert d["fraction_within_1e-4"] >= 0.997, d                    # 0.9973 (RK) / 0.9978 (Euler) measured; the meaningful bound is the relative one below
        assert d["per_pixel_norm"]["fraction_within_1e-4"] >= 0.999, d  # 0.9994 measured


FMA_CASES = ["euler_l0", "rk_l0", "rk_ladder", "rk_outside"]


@pytest.mark.parametrize("name", FMA_CASES)
def test_fma_kernel_reproduces_the_frozen_fma_fixtures(name):
    g = np.load(os.path.join(GOLD, "frames_fma.npz"))
    tex = (g["t_temp"], g["t_disk"], g["t_sky"])
    u = (g[f"{name}.camera"].tobytes(), g[f"{name}.black_hole"].tobytes(), g[f"{name}.details"].tobytes())
    sizes = [tuple(int(v) for v in s) for s in g[f"{name}.sizes"]]
    cfg = B.ladder_from_base(sizes[0], 3, len(sizes))
    rp = _gpu(cfg, u, tex, eval_fma=True, counters=True)
    n = 0
    for l in range(len(sizes)):
        n += _same_to_the_bit_where_specified(rp.read_level(l), g[f"{name}.level{l}"], f"fma {name} level {l}")
    assert n > 0
    traced, steps, copied, interp, sky = (int(v) for v in g[f"{name}.stats"])
    c = rp.counters()
    assert (c["traced"], c["steps"], c["copied"], c["interpolated"], c["sky_samples"]) == (traced, steps, copied, interp, sky)
    if name in ("rk_l0", "euler_l0"):                       # a third evaluation: neither the contract nor the literal one
        assert not np.array_equal(_gpu(cfg, u, tex).read_hdr(), rp.read_hdr())
        assert not np.array_equal(_gpu(cfg, u, tex, literal=True).read_hdr(), rp.read_hdr())


@pytest.mark.parametrize("method", [1, 0])
def test_fma_kernel_equals_fma_oracle_at_1918x1081(method):
    tex = T.textures(small=False)
    u = T.uniforms(integration_method=method)
    cfg = B.ladder_from_base((72, 41), 3, 4)
    rp = _gpu(cfg, u, tex, eval_fma=True, counters=True)
    cnt = O.Counters()
    O.set_eval(O.EVAL_FMA)
    try:
        want = O.render_ladder(T.oracle_scene(*u, tex), cfg.sizes(), cnt)
    finally:
        O.set_eval(O.EVAL_CONTRACT)
    for l in range(4):
        _same_to_the_bit_where_specified(rp.read_level(l), want[l], f"fma level {l} method {method}")
    assert rp.counters() == cnt.as_dict()


@pytest.mark.parametrize("method", [1, 0])
def test_the_outliers_are_the_problems_not_the_kernels(method):
    """Bench frame, every evaluation on the GPU.  A = pixels where the CONTRACT kernels differ from the literal kernel by more than
    1e-4, B = the same for the FMA-only kernel, P = the same for the LITERAL kernel with the camera's z moved by ONE ULP.
    Claims (each asserted with a factor-of-two margin, all recorded):
      * |A| and |B| are within 3x of |P|: rounding the operations differently costs what moving one input by one ulp costs;
      * A and B are mostly the same pixels (>= 70 % of the smaller set);
      * measured against the pixel's norm instead of per channel, every population shrinks by more than 3x (direction pixels alone: 20x; the per-channel
        metric divides ~1e-6 of absolute error by channels that are near zero) and stays below 0.1 % of the frame."""
    tex = T.textures(small=False)
    cfg = B.ladder_for_frame((1920, 1080), 3, 4)
    u = T.uniforms(integration_method=method)
    z1 = float(np.nextafter(np.float32(-19.0), np.float32(0.0)))
    up = T.uniforms(camera=B.Camera(position=(0.0, 0.0, z1)), integration_method=method)
    lit = _gpu(cfg, u, tex, literal=True).read_hdr()
    con = _gpu(cfg, u, tex).read_hdr()
    fma = _gpu(cfg, u, tex, eval_fma=True).read_hdr()
    per = _gpu(cfg, up, tex, literal=True).read_hdr()
    A, An = _beyond(con, lit)
    Bm, Bn = _beyond(fma, lit)
    P, Pn = _beyond(per, lit)
    C_, Cn = _beyond(fma, con)
    n = A.size
    entry = {"config": "1920x1080 " + ("adaptive RK" if method else "Euler"), "population_test": True, "pixels": int(n),
             "per_channel_beyond_1e-4": {"contract_vs_literal": int(A.sum()), "fma_vs_literal": int(Bm.sum()),
                                         "literal_camera_z_plus_1ulp_vs_literal": int(P.sum()), "fma_vs_contract": int(C_.sum()),
                                         "contract_and_fma": int((A & Bm).sum()), "contract_and_perturbed": int((A & P).sum())},
             "per_pixel_norm_beyond_1e-4": {"contract_vs_literal": int(An.sum()), "fma_vs_literal": int(Bn.sum()),
                                            "literal_camera_z_plus_1ulp_vs_literal": int(Pn.sum()), "fma_vs_contract": int(Cn.sum()),
                                            "contract_and_fma": int((An & Bn).sum()), "contract_and_perturbed": int((An & Pn).sum())},
             "class_differences": {"contract": int((con[..., 3] != lit[..., 3]).sum()), "fma": int((fma[..., 3] != lit[..., 3]).sum()),
                                   "perturbed": int((per[..., 3] != lit[..., 3]).sum())}}
    _record(entry)
    # (a pixel whose CLASS differs - colour against direction, a hit decided in the last bit - counts as beyond; at most a handful)
    # measured (profiles/r03_literal_distance.jsonl): contract 0 / 0, fma 0 (RK) / 1 (Euler), perturbed 0 / 0 - the contract kernel is held to
    # what the other test of this frame demands of it (no class differs); a third evaluation and a perturbed input may flip a pixel or two
    assert entry["class_differences"]["contract"] == 0 and entry["class_differences"]["fma"] <= 2 and entry["class_differences"]["perturbed"] <= 4, entry
    assert P.sum() > 0, "a one-ulp change of the camera position must move some pixel by more than 1e-4 - else the bar would be attainable"
    assert A.sum() <= 3 * P.sum() and Bm.sum() <= 3 * P.sum(), entry
    assert (A & Bm).sum() >= 0.7 * min(A.sum(), Bm.sum()), entry
    assert An.sum() * 3 <= A.sum() and Bn.sum() * 3 <= Bm.sum(), entry
    assert An.sum() <= 1e-3 * n and Bn.sum() <= 1e-3 * n and An.sum() <= 3 * max(1, Pn.sum()) + 0.0002 * n, entry


# ---- the executed-shader pin at the METRIC'S OWN FRAME (tests/golden/wgsl_exec_native_samples.npz: 6 000 seeded pixels of the last level of
# the reference-native 72x41 -> 1918x1081 adaptive-RK ladder, each the result of running the reference's `main` on that pixel)
def _native():
    n = np.load(os.path.join(GOLD, "wgsl_exec_native_samples.npz"))
    u = tuple(n[k].tobytes() for k in ("camera", "black_hole", "details"))
    tex = (n["t_temp"], n["t_disk"], n["t_sky"])
    cfg = B.ladder_from_base((72, 41), 3, 4)
    assert cfg.sizes() == [tuple(int(v) for v in s) for s in n["sizes"]] and (cfg.crop_x, cfg.crop_y) == (0, 0)
    return n, u, tex, cfg


def test_literal_kernel_reproduces_the_executed_shader_at_the_native_frame():
    n, u, tex, cfg = _native()
    frame = _gpu(cfg, u, tex, literal=True).read_hdr()
    px = n["pixels"]
    got, want = frame[px[:, 1], px[:, 0]], n["values"]
    assert np.array_equal(got[:, 3], want[:, 3]), "pixel classes differ from the executed shader"
    d = want[:, 3] == 0
    assert d.sum() >= 2500
    assert np.array_equal(got[d].view(np.uint32), want[d].view(np.uint32)), f"{int((got[d].view(np.uint32) != want[d].view(np.uint32)).any(axis=1).sum())} direction pixels are not bit-identical"
    e = np.abs(got[~d] - want[~d]) / np.maximum(np.abs(want[~d]), T.ABS_FLOOR)
    assert float(e.max()) <= 2e-5, float(e.max())                                   # colours: device powf against glibc's (<= 1.2e-5 measured)
    lvl2 = _gpu(cfg, u, tex, literal=True).read_level(2)
    cp = n["coarse_pixels"]
    gc, wc = lvl2[cp[:, 1], cp[:, 0]], n["coarse_values"]
    dc = wc[:, 3] == 0
    assert np.array_equal(gc[:, 3], wc[:, 3]) and np.array_equal(gc[dc].view(np.uint32), wc[dc].view(np.uint32))


def test_default_kernel_against_the_executed_shader_at_the_native_frame():
    from tests.test_wgsl_pin import native_distance_by_kind
    n, u, tex, cfg = _native()
    frame = _gpu(cfg, u, tex, speculative_levels=2).read_hdr()
    px = n["pixels"]
    got, want = frame[px[:, 1], px[:, 0]], n["values"]
    assert np.array_equal(got[:, 3], want[:, 3]), "pixel classes differ from the executed shader"
    rows = native_distance_by_kind(n, got)
    _record({"config": "reference-native 1918x1081 adaptive RK, 6000 executed-shader sample pixels", "default_kernels_vs": "ray.wgsl as executed (oracle/wgsl_exec.py)",
             "by_kind": {name: {"fraction_within_1e-4_per_channel": round(a, 4), "fraction_within_1e-4_of_the_norm": round(b, 4)} for name, a, b in rows}})
    for name, f_ch, f_norm in rows:                                                  # the bounds of tests/test_wgsl_pin.py (the contract oracle = these kernels, bit for bit)
        assert f_ch >= {"traced_escaped": 0.915, "border": 0.89}.get(name, 0.995) and f_norm >= 0.99, (name, f_ch, f_norm)
