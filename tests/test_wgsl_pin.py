"""CPU: the restatements against frames made by EXECUTING the reference's own shader text.

tests/golden/wgsl_exec.npz was written by tests/golden/make_golden_wgsl.py, which parses /root/reference/src/renderer/shaders/ray.wgsl
and runs it (oracle/wgsl_exec.py: a WGSL-subset interpreter; nothing of the shader is restated in it).  The fixtures are data only -
uniform bytes, small synthetic textures / mesh arrays, frames.  What WGSL leaves to the implementation is fixed as the LITERAL evaluation
fixes it, so:

  * the C oracle in literal mode (oracle_set_eval(1)) must reproduce EVERY WORD of every frame - directions, colours, classes, mesh
    shading, interpolated pixels, ladder levels;
  * the NumPy restatement in literal mode must reproduce classes and direction pixels bit for bit (its float32 pow(., 1.3) is numpy's
    vectorised one, an ulp away from glibc's on ~20 % of arguments, so colours are held to 1e-6);
  * where the reference is present (this container, not the GPU box) a slice of one frame is re-executed from the shader text and must
    equal the committed fixture - the fixture is what the generator says it is.

tests/test_gpu_literal.py holds the BHRAY_F_LITERAL kernel to the same file.  The contract kernels (the shipped default) are tied to the
literal evaluation by the population bounds of tests/test_gpu_literal.py, and the contract oracle to the contract kernels bit for bit.
"""
import os

import numpy as np
import pytest

from oracle import host_oracle as H
from oracle import np_ray as N
from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["euler_l0", "rk_l0", "rk_ladder", "euler_ladder", "rk_outside", "rk_off_origin", "rk_highlight", "euler_tight", "rk_mesh", "euler_mesh", "rk_mesh_near", "euler_mesh_near"]
FUZZ = [f"fuzz{k:02d}" for k in range(24)]            # 24 seeded random points of the UI's uniform space, both integrators, two-level ladders


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLD, "wgsl_exec.npz"))


def scene_of(g, name, structured_nodes=False):
    tex = (g["t_temp"], g["t_disk"], g["t_sky"])
    u = tuple(g[f"{name}.{k}"].tobytes() for k in ("camera", "black_hole", "details"))
    sizes = [tuple(int(v) for v in s) for s in g[f"{name}.sizes"]]
    models = []
    if int(g[f"{name}.mesh"][0]):
        nodes = g["mesh.nodes"].view(H.NODE_DTYPE) if structured_nodes else g["mesh.nodes"]
        models = [dict(position=tuple(float(v) for v in g["mesh.position"]), visible=1, points=g["mesh.points"], normals=g["mesh.normals"],
                       triangles=g["mesh.triangles"], nodes=nodes, bvh_lookup=g["mesh.bvh_lookup"])]
    return u, tex, sizes, models


def test_fixture_covers_every_path_of_the_shader(g):
    """Traced rays of every fate, interpolated and copied pixels, disk and sky colours, mesh shading."""
    seen_mesh = False
    assert sorted(k[:-6] for k in g.files if k.endswith(".sizes")) == sorted(CASES + FUZZ)
    methods = {int(np.frombuffer(g[f"{n}.details"].tobytes(), dtype=np.int32)[3]) for n in FUZZ}
    assert methods == {0, 1}
    for name in CASES:
        sizes = g[f"{name}.sizes"]
        for l in range(len(sizes)):
            im = g[f"{name}.level{l}"]
            assert im.shape == (int(sizes[l][1]), int(sizes[l][0]), 4) and im.dtype == np.float32
            assert set(np.unique(im[..., 3][~np.isnan(im[..., 3])])) <= {0.0, 1.0}
        seen_mesh |= bool(g[f"{name}.mesh"][0])
    top = g["rk_ladder.level2"]
    assert (top[..., 3] == 0).any() and (top[..., 3] == 1).any()
    assert seen_mesh
    # mesh shading is IN the frames: the near-mesh scene without its model is a different picture
    u, tex, sizes, models = scene_of(g, "rk_mesh_near")
    det0 = H.ray_details(integration_method=1, model_count=0)
    O.set_eval(O.EVAL_LITERAL)
    try:
        bare = O.render_ladder(O.OracleScene(u[0], u[1], det0, *tex), sizes)[0]
    finally:
        O.set_eval(O.EVAL_CONTRACT)
    assert int((bare != g["rk_mesh_near.level0"]).any(axis=-1).sum()) >= 40


@pytest.mark.parametrize("name", CASES + FUZZ)
def test_c_oracle_literal_mode_equals_the_executed_shader_word_for_word(g, name):
    u, tex, sizes, models = scene_of(g, name)
    O.set_eval(O.EVAL_LITERAL)
    try:
        imgs = O.render_ladder(O.OracleScene(*u, *tex, models=models), sizes)
    finally:
        O.set_eval(O.EVAL_CONTRACT)
    for l, im in enumerate(imgs):
        want = g[f"{name}.level{l}"]
        bad = im.view(np.uint32) != want.view(np.uint32)
        both_nan = np.isnan(im) & np.isnan(want)                 # NaN payloads are not specified; NaN in the same place is
        assert not (bad & ~both_nan).any(), f"{name} level {l}: {int((bad & ~both_nan).sum())} words differ, first at {np.argwhere(bad & ~both_nan)[:4].tolist()}"


@pytest.mark.parametrize("name", ["euler_l0", "rk_l0", "rk_ladder", "rk_off_origin", "euler_mesh", "rk_mesh_near"])
def test_numpy_literal_mode_equals_the_executed_shader(g, name):
    u, tex, sizes, models = scene_of(g, name, structured_nodes=True)
    N.set_literal(True)
    try:
        imgs = N.render_ladder(N.Scene(*u, *tex, models=models), sizes, {})
    finally:
        N.set_literal(False)
    for l, im in enumerate(imgs):
        want = g[f"{name}.level{l}"]
        ok = ~np.isnan(want).any(axis=-1)
        assert np.array_equal(np.isnan(im).any(axis=-1), ~ok)
        assert np.array_equal(im[..., 3][ok], want[..., 3][ok]), f"{name} level {l}: classes"
        d = ok & (want[..., 3] == 0)
        assert np.array_equal(im[d].view(np.uint32), want[d].view(np.uint32)), f"{name} level {l}: direction pixels"
        e = np.abs(im[ok] - want[ok]) / np.maximum(np.abs(want[ok]), 1e-3)
        assert float(e.max(initial=0.0)) <= 1e-6, f"{name} level {l}: colours {float(e.max())}"


@pytest.mark.parametrize("name", ["rk_ladder", "rk_outside", "euler_mesh_near"])
def test_sky_pass_restatements_equal_the_executed_sky_shader(g, name):
    """sky.wgsl run by the same interpreter over the executed ray frames: rgba16float, every half-word."""
    last = len(g[f"{name}.sizes"]) - 1
    prev, want = g[f"{name}.level{last}"], g[f"{name}.sky"]
    assert want.dtype == np.uint16 and want.shape == prev.shape
    assert np.array_equal(O.sky_resolve(prev, g["t_sky"]).view(np.uint16), want)
    assert np.array_equal(N.sky_resolve(prev, g["t_sky"]).view(np.uint16), want)
    assert (prev[..., 3] == 0).sum() > 100 and np.all(want.view(np.float16)[..., 3] == 1.0)


def test_the_contract_oracle_is_not_this_evaluation_but_close(g):
    """The shipped contract (FMA + reassociation in the integrator) differs from the text in the last bits only."""
    u, tex, sizes, models = scene_of(g, "rk_l0")
    im = O.render_ladder(O.OracleScene(*u, *tex), sizes)[0]
    want = g["rk_l0.level0"]
    assert not np.array_equal(im.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(im[..., 3], want[..., 3])
    d = want[..., 3] == 0
    err = np.linalg.norm(im[d][:, :3] - want[d][:, :3], axis=-1)
    assert float(np.median(err)) < 2e-6


@pytest.mark.skipif(not os.path.exists("/root/reference/src/renderer/shaders/ray.wgsl"), reason="the reference's shader is not on this machine")
def test_fixture_is_what_executing_the_shader_text_gives():
    from oracle import wgsl_exec as W
    g = np.load(os.path.join(GOLD, "wgsl_exec.npz"))
    ns = W.compile_shader()
    for name, level, rows in (("rk_l0", 0, (20, 21)), ("euler_ladder", 1, (17, 19)), ("euler_mesh_near", 1, (9, 10))):
        u, tex, sizes, models = scene_of(g, name)
        W.bind_scene(ns, *u, *tex, models)
        prev = g[f"{name}.level{level - 1}"] if level else None
        got = W.render_level(ns, sizes[level], prev, rows)[rows[0]:rows[1]]
        want = g[f"{name}.level{level}"][rows[0]:rows[1]]
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), name
    sky = W.render_sky(g["euler_mesh_near.level1"], g["t_sky"])
    assert np.array_equal(sky.view(np.uint16), g["euler_mesh_near.sky"])


@pytest.mark.skipif(not os.path.exists("/root/reference/src/renderer/shaders/ray.wgsl"), reason="the reference's shader is not on this machine")
def test_executor_parses_the_whole_shader_and_nothing_is_left_out():
    from oracle import wgsl_exec as W
    src = open(W.SHADER).read()
    decls = W.Parser(W.tokenize(src)).module()
    fns = [d[1] for d in decls if d[0] == "fn"]
    assert src.count("\nfn ") + src.startswith("fn ") == len(fns) and "main" in fns and "trace_ray" in fns and "next_ray_rk" in fns
    ns = W.compile_shader()
    assert all(("fn_" + f) in ns for f in fns)
    # AbstractFloat constants stay binary64 until they meet an f32 (b_1 - b_a_1 is one rounding, not three)
    assert isinstance(ns["C_b_1"], float) and ns["C_b_1"] == 37.0 / 378.0 and isinstance(ns["C_PI"], np.float32)


# ---- the pin at the METRIC'S OWN FRAME: 6 000 seeded pixels of the last level of the reference-native 72x41 -> 1918x1081 adaptive-RK
# ladder, each produced by running the reference's `main` on that pixel (tests/golden/make_golden_wgsl_native.py)
@pytest.fixture(scope="module")
def native():
    return np.load(os.path.join(GOLD, "wgsl_exec_native_samples.npz"))


@pytest.fixture(scope="module")
def native_literal_ladder(native):
    u = tuple(native[k].tobytes() for k in ("camera", "black_hole", "details"))
    tex = (native["t_temp"], native["t_disk"], native["t_sky"])
    sizes = [tuple(int(v) for v in s) for s in native["sizes"]]
    O.set_literal(True)
    try:
        return O.render_ladder(O.OracleScene(*u, *tex), sizes)
    finally:
        O.set_literal(False)


def test_native_samples_cover_every_kind_of_pixel(native):
    kinds = [k.decode() for k in native["kinds"]]
    assert kinds == ["copied", "interpolated", "traced_escaped", "traced_disk", "traced_captured", "border"]
    assert [tuple(s) for s in native["sizes"]] == [(72, 41), (214, 121), (640, 361), (1918, 1081)]       # mod.rs:177-205
    n = np.bincount(native["kind"], minlength=6)
    assert native["pixels"].shape == (6000, 2) and (n == 1000).all()
    v = native["values"]
    assert not np.isnan(v[..., 3]).any() and set(np.unique(v[:, 3])) == {0.0, 1.0}
    k = native["kind"]
    assert (v[k == 2][:, 3] == 0).all() and (v[k == 3][:, 3] == 1).all() and (v[k == 4][:, :3] == 0).all()
    assert int(np.frombuffer(native["details"].tobytes(), dtype=np.int32)[3]) == 1                         # adaptive RK


def test_c_oracle_literal_reproduces_every_word_of_the_native_samples(native, native_literal_ladder):
    px = native["pixels"]
    got = native_literal_ladder[3][px[:, 1], px[:, 0]]
    want = native["values"]
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"{int((got.view(np.uint32) != want.view(np.uint32)).any(axis=1).sum())} of 6000 sample pixels differ"
    cp = native["coarse_pixels"]
    gc, wc = native_literal_ladder[2][cp[:, 1], cp[:, 0]], native["coarse_values"]
    assert np.array_equal(gc.view(np.uint32), wc.view(np.uint32))


def test_contract_oracle_against_the_native_samples(native):
    """The shipped evaluation (the contract: fused multiply-add + reassociation in the integrator) at the metric's frame, against the
    executed shader: no pixel class differs; population bounds as in tests/test_gpu_literal.py (the GPU kernels equal this oracle bit for bit)."""
    u = tuple(native[k].tobytes() for k in ("camera", "black_hole", "details"))
    tex = (native["t_temp"], native["t_disk"], native["t_sky"])
    sizes = [tuple(int(v) for v in s) for s in native["sizes"]]
    img = O.render_ladder(O.OracleScene(*u, *tex), sizes)[3]
    px = native["pixels"]
    got, want = img[px[:, 1], px[:, 0]], native["values"]
    assert np.array_equal(got[:, 3], want[:, 3]), "pixel classes differ from the executed shader"
    for name, f_ch, f_norm in native_distance_by_kind(native, got):
        # measured (CPU contract oracle = the default kernels bit for bit), per channel / against the pixel's norm:
        #   copied 1.000 / 1.000   interpolated 0.999 / 1.000   traced_escaped 0.925 / 0.998   traced_disk 0.997 / 0.998   traced_captured 1 / 1   border 0.901 / 0.994
        # The traced-and-escaped rays of the LAST level are the 0.5 % of a frame's pixels that pass closest to the hole and still escape (everything
        # easier was interpolated): unit vectors carrying ~1e-6 of accumulated rounding, more than 1e-4 RELATIVE in a channel near zero.
        lo_ch = {"traced_escaped": 0.915, "border": 0.89}.get(name, 0.995)
        assert f_ch >= lo_ch and f_norm >= 0.99, (name, f_ch, f_norm)


def native_distance_by_kind(native, got):
    """[(kind, fraction within 1e-4 per channel, fraction within 1e-4 of the pixel's norm)] of `got` against the executed-shader samples"""
    want, k = native["values"], native["kind"]
    err = np.abs(got[:, :3] - want[:, :3]) / np.maximum(np.abs(want[:, :3]), 1e-3)
    per_px = err.max(axis=1)
    nrm = np.linalg.norm(got[:, :3] - want[:, :3], axis=1) / np.maximum(np.linalg.norm(want[:, :3], axis=1), 1e-3)
    return [(name.decode(), float((per_px[k == i] <= 1e-4).mean()), float((nrm[k == i] <= 1e-4).mean())) for i, name in enumerate(native["kinds"])]


@pytest.mark.skipif(not os.path.exists("/root/reference/src/renderer/shaders/ray.wgsl"), reason="the reference is not on this machine")
def test_native_fixture_is_what_executing_the_shader_text_gives(native, native_literal_ladder):
    from oracle import wgsl_exec as W
    ns = W.compile_shader()
    u = tuple(native[k].tobytes() for k in ("camera", "black_hole", "details"))
    W.bind_scene(ns, *u, native["t_temp"], native["t_disk"], native["t_sky"])
    pick = np.concatenate([np.nonzero(native["kind"] == k)[0][:6] for k in range(6)])
    got = W.render_pixels(ns, (1918, 1081), native_literal_ladder[2], [tuple(int(v) for v in native["pixels"][i]) for i in pick])
    assert np.array_equal(got.view(np.uint32), native["values"][pick].view(np.uint32))
