"""CPU: analytic known-answer tests for the oracle: closed-form intersections,
straight-line propagation far from the hole, photon capture vs impact parameter, ladder classes,
portable transcendental accuracy.  These pin the restatement to the maths, not to another copy."""
import numpy as np
import pytest

from bhusie_amd import assets
from oracle import host_oracle as H
from oracle import oracle as O

TEX = (assets.temp_lut(16), assets.disk_texture(32, seed=1), assets.sky_texture(64, 32, seed=2))


def scene(**kw):
    cam = H.camera_uniform(**kw.pop("camera", {}))
    bh = H.black_hole_uniform(**kw.pop("bh", {}))
    return O.OracleScene(cam, bh, H.ray_details(**kw), *TEX)


def test_ray_sphere_closed_form():
    r = O.hit(0, [0, 0, -5, 0, 0, 1], [1.0, 0, 0, 0], 1e-8, 1e5)
    assert r["hit"] and abs(r["t"] - 4.0) < 1e-6 and np.allclose(r["normal"], [0, 0, -1])
    r = O.hit(0, [0, 0, 0, 0, 0, 1], [2.0, 0, 0, 0], 1e-8, 1e5)          # from inside: far root
    assert r["hit"] and abs(r["t"] - 2.0) < 1e-6
    assert not O.hit(0, [0, 2, -5, 0, 0, 1], [1.0, 0, 0, 0], 1e-8, 1e5)["hit"]
    assert not O.hit(0, [0, 0, -5, 0, 0, 1], [1.0, 0, 0, 0], 1e-8, 3.9)["hit"]      # beyond t_max
    r = O.hit(0, [0, 0, -5, 0, 0, 2], [1.0, 0, 0, 0], 1e-8, 1e5)                    # a = d.d != 1
    assert r["hit"] and abs(r["t"] - 2.0) < 1e-6


def test_ray_disk_closed_form():
    n = [0, 1, 0]
    r = O.hit(1, [3, 5, 0, 0, -1, 0], [2.0, 10.0, 0, 0, 0, *n], 1e-8, 1e5)
    assert r["hit"] and abs(r["t"] - 5.0) < 1e-6
    assert not O.hit(1, [1, 5, 0, 0, -1, 0], [2.0, 10.0, 0, 0, 0, *n], 1e-8, 1e5)["hit"]      # inside the inner radius
    assert not O.hit(1, [11, 5, 0, 0, -1, 0], [2.0, 10.0, 0, 0, 0, *n], 1e-8, 1e5)["hit"]     # outside the outer radius
    assert not O.hit(1, [3, 5, 0, 1, 0, 0], [2.0, 10.0, 0, 0, 0, *n], 1e-8, 1e5)["hit"]       # parallel: t = +-inf
    assert not O.hit(1, [3, 5, 0, 0, 1, 0], [2.0, 10.0, 0, 0, 0, *n], 1e-8, 1e5)["hit"]       # behind


def test_ray_aabb_and_triangle_closed_form():
    box = [-1, -1, -1, 1, 1, 1, 0, 0, 0]
    assert abs(O.hit(2, [0, 0, -5, 0, 0, 1], box, 0, 0)["t"] - 4.0) < 1e-6
    assert O.hit(2, [0, 0, 0, 0, 0, 1], box, 0, 0)["t"] == -1.0                   # inside: negative entry (ray.wgsl:721)
    assert O.hit(2, [0, 3, -5, 0, 0, 1], box, 0, 0)["t"] == np.float32(1e8)       # miss
    assert O.hit(2, [0, 0, 5, 0, 0, 1], box, 0, 0)["t"] == np.float32(1e8)        # behind
    assert abs(O.hit(2, [10, 0, -5, 0, 0, 1], [-1, -1, -1, 1, 1, 1, 10, 0, 0], 0, 0)["t"] - 4.0) < 1e-6   # offset
    tri = [0, 0, 0, 1, 0, 0, 0, 1, 0] + [0, 0, 1] * 3
    r = O.hit(3, [0.25, 0.25, -2, 0, 0, 1], tri, 1e-8, 1e5)
    assert r["hit"] and abs(r["t"] - 2.0) < 1e-6
    assert np.allclose(r["normal"], [0, 0, -1])                                    # flipped toward the ray
    assert np.allclose(r["color"], [0.5, 0.5, 0.0])                                # 0.5 - 0.5*n
    assert not O.hit(3, [0.75, 0.75, -2, 0, 0, 1], tri, 1e-8, 1e5)["hit"]          # u+v > 1
    assert not O.hit(3, [0.25, 0.25, -2, 1, 0, 0], tri, 1e-8, 1e5)["hit"]          # parallel


def test_far_rays_go_straight():
    """As h2/dist^5 -> 0 the integrators reduce to straight-line motion."""
    sc = scene()
    ray = np.array([0, 2000.0, -3000.0, 0, 0, 1.0], dtype=np.float32)
    for method in (0, 1):
        tr = O.integrate(sc, ray, 0.5, method, 200)
        assert np.allclose(tr[-1, 3:6], [0, 0, 1], atol=1e-5)      # 1.5*h2/d^4 * path ~ 4e-6: bends toward the hole
        assert tr[-1, 4] <= 0.0
        path = 100.0 if method == 0 else 0.5 * sum(1.0001 ** k for k in range(200))     # RK: h *= 1.0001 per step
        assert abs(tr[-1, 2] - (-3000.0 + path)) < 0.05
        assert np.all(np.abs(np.linalg.norm(tr[:, 3:6], axis=1) - 1) < 1e-6)


def test_rk_step_size_adaptation_rule():
    """h *= 1.0001 while e_max <= 2e-5, h *= 0.9*e_max^-0.001 otherwise (ray.wgsl:458-462)."""
    sc = scene()
    far = O.integrate(sc, np.array([0, 50, -100, 0, 0, 1.0], np.float32), 0.15, 1, 3)
    assert np.float32(far[0, 6]) == np.float32(np.float32(0.15) * np.float32(1.0001))
    near = O.integrate(sc, np.array([0, 1.5, 0, 0, 0, 1.0], np.float32), 0.5, 1, 1)
    e = near[0, 7]
    assert e > 2e-5
    assert np.float32(near[0, 6]) == np.float32(np.float32(0.5) * np.float32(np.float32(0.9) * np.float32(O.pow_m001(float(e)))))


def test_capture_is_monotone_in_impact_parameter():
    """Photon capture threshold near b = 3*sqrt(3)/2 ~ 2.6 r_s (event horizon radius 1 = r_s)."""
    bh = dict(show_disk_texture=0, show_red_shift=0, accretion_disk_inner=1000.0, accretion_disk_outer=1001.0,
              relativity_sphere_radius=200.0)
    sc = O.OracleScene(H.camera_uniform(position=(0, 0, -100)), H.black_hole_uniform(**bh),
                       H.ray_details(integration_method=0, step_size=0.02, max_iterations=40000), *TEX)
    captured = []
    bs = [1.0, 2.0, 2.4, 2.55, 2.7, 3.0, 4.0, 8.0]
    for b in bs:
        out = O.trace_ray(sc, [0, b, -100, 0, 0, 1])
        captured.append(bool(out[3] == 1.0 and np.all(out[:3] == 0)))       # horizon: colour 0, opacity 1
    assert captured[:4] == [True] * 4 and captured[4:] == [False] * 4, list(zip(bs, captured))
    # deflection decreases with b for escaping rays
    defl = [np.degrees(np.arccos(np.clip(O.trace_ray(sc, [0, b, -100, 0, 0, 1])[2], -1, 1))) for b in (3.0, 4.0, 8.0, 16.0)]
    assert defl[0] > defl[1] > defl[2] > defl[3] > 0


def test_ray_missing_the_sphere_is_sky_coloured():
    """Rays that never touch the relativity sphere break at i=0 and are classified colour (i<=5, ray.wgsl:583)."""
    sc = scene(camera=dict(position=(0, 0, -100.0)))
    out = O.trace_ray(sc, [0, 0, -100, 0, 1, 0])
    assert out[3] == 1.0
    d = np.array([0, 1, 0], np.float32)
    theta = np.arctan2(np.hypot(d[0], d[2]), d[1]); phi = np.arctan2(d[2], d[0])
    PI = np.float32(3.1415926)
    u = ((phi + 2.6 * PI) / (2 * PI)) % 1.0; v = ((PI - theta) / PI) % 1.0
    want = O.sample(TEX[2], float(u), float(v))[:3] ** 4
    assert np.allclose(out[:3], want, rtol=1e-4, atol=1e-7)


def test_grid_levels_copy_interpolate_trace():
    sc = scene(integration_method=1)
    sizes = O.ladder_sizes((24, 14), 3, 3)
    assert sizes == [(24, 14), (70, 40), (208, 118)]
    cnt = O.Counters()
    l0, l1, l2 = O.render_ladder(sc, sizes, cnt)
    assert np.array_equal(l1[::3, ::3], l0) and np.array_equal(l2[::3, ::3], l1)     # copies (ray.wgsl:193)
    c = cnt.as_dict()
    assert c["copied"] == 24 * 14 + 70 * 40 and c["pixels"] == 24 * 14 + 70 * 40 + 208 * 118
    assert c["traced"] == c["pixels"] - c["copied"] - c["interpolated"]
    assert c["interpolated"] > 0
    # an interpolated pixel is the bilinear mix of its 4 coarse neighbours (alpha 0, angles < threshold)
    found = 0
    r = np.float32(70) / np.float32(208 + 2)
    for y in range(1, 117):
        for x in range(1, 207):
            if (x % 3 or y % 3) and l2[y, x, 3] == 0:
                nb = l1[y // 3:y // 3 + 2, x // 3:x // 3 + 2]
                if np.all(nb[..., 3] == 0):
                    tx = np.float32(x) * r - np.float32(x // 3); ty = np.float32(y) * np.float32(np.float32(40) / np.float32(120)) - np.float32(y // 3)
                    top = nb[0, 0, :3] * (1 - tx) + nb[0, 1, :3] * tx; bot = nb[1, 0, :3] * (1 - tx) + nb[1, 1, :3] * tx
                    if np.allclose(l2[y, x, :3], top * (1 - ty) + bot * ty, atol=1e-6):
                        found += 1
    assert found >= 0.9 * c["interpolated"]


def test_portable_functions_track_libm():
    rng = np.random.default_rng(5)
    import ctypes as C
    L = O.lib()
    for n in ("bh_atan2", "bh_sin", "bh_cos"):
        getattr(L, n).restype = C.c_float
    L.bh_atan2.argtypes = [C.c_float, C.c_float]; L.bh_sin.argtypes = [C.c_float]; L.bh_cos.argtypes = [C.c_float]

    def ulps(got, want):
        return abs(float(got) - float(want)) / float(np.spacing(np.float32(max(abs(want), 1e-30))))
    for _ in range(4000):
        x = np.float32(rng.uniform(-1, 1))
        assert ulps(O.acos(float(x)), np.arccos(np.float64(x))) <= 2.0
        e = np.float32(10 ** rng.uniform(-20, 5))
        assert ulps(O.pow_m001(float(e)), np.float64(e) ** -0.001) <= 1.0
        y, xx = rng.normal(size=2).astype(np.float32)
        assert ulps(L.bh_atan2(float(y), float(xx)), np.arctan2(np.float64(y), np.float64(xx))) <= 4.0
        a = np.float32(rng.uniform(-200, 200))
        assert abs(L.bh_sin(float(a)) - np.sin(np.float64(a))) <= 1.5e-7
        assert abs(L.bh_cos(float(a)) - np.cos(np.float64(a))) <= 1.5e-7
    assert O.acos(1.0) == 0.0 and np.isnan(O.acos(1.5)) and O.pow_m001(1.0) == 1.0


@pytest.mark.parametrize("method", [0, 1])
def test_contract_vs_literal_evaluation_of_the_integrator(method):
    """The numerics contract evaluates the integrator with the freedoms WGSL grants (N3 integer pow, N7 fused multiply-add,
    N9/N10 reassociation).  oracle_set_literal(1) evaluates the shader text operator by operator instead.  The two must describe
    the same image: identical pixel classes, a median difference at the rounding level, and only the chaotic rays (photon
    sphere, disk edges - where any two conforming implementations disagree) beyond the 1e-4 parity bar.  Measured at
    649x361, every pixel traced: 0 class differences, median 5e-7, 99.88 % of the pixels within 1e-4 (same statistics for
    N3+N7 alone: N9 does not move the contract away from the literal reading)."""
    from tests import common as T
    tex = T.textures()
    u = T.uniforms(integration_method=method)
    sc = T.oracle_scene(*u, tex)
    try:
        O.set_literal(True)
        lit = O.render_level(sc, (217, 121), None)
    finally:
        O.set_literal(False)
    con = O.render_level(sc, (217, 121), None)
    assert not np.array_equal(lit, con), "the switch had no effect"
    cls = lit[..., 3] != con[..., 3]
    assert cls.mean() <= 1e-3
    e = (np.abs(con - lit) / np.maximum(np.abs(lit), 1e-3))[~cls].max(axis=-1)
    print(f"method {method}: class differences {int(cls.sum())}, median {np.median(e):.3g}, p99 {np.quantile(e, 0.99):.3g}, "
          f"beyond 1e-4: {(e > 1e-4).mean():.4%}, max {e.max():.3g}")
    assert np.median(e) < 2e-6
    assert (e > 1e-4).mean() < 0.01


@pytest.mark.parametrize("method", [0, 1])
def test_mirror_symmetry_of_the_restatement(method):
    """With the camera and the disk normal in the plane x = 0, a black sky and the handed shading terms off, the image is
    mirror-symmetric in x bit for bit (sign symmetry of every IEEE operation on the path): escape directions with x negated,
    classes and disk colours unchanged.  The GPU suite checks the same property on the HIP path at 1919x1079."""
    import bhusie_amd as B
    from tests import common as T
    tex = list(T.textures()); tex[2] = np.zeros((1, 1, 4), dtype=np.uint8)
    bh = B.BlackHole(accretion_disk_rotation=(0.3, 0.0, 0.0), show_disk_texture=0, show_red_shift=0)
    u = T.uniforms(black_hole=bh, integration_method=method)
    img = O.render_level(T.oracle_scene(*u, tex), (201, 113), None)
    mir = img[:, ::-1].copy(); d = img[..., 3] == 0
    assert d.any() and (~d).any()
    mir[..., 0] = np.where(d, -mir[..., 0], mir[..., 0])
    assert np.array_equal(img, mir)
