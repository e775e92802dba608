"""-m gpu: the HIP path (through the C ABI) against the CPU oracle on identical bytes.

Bar (BASELINE.json north_star): every channel within 1e-4 relative; pixel classes (alpha) identical.
Direction pixels (alpha = 0) and the integrator are specified to the bit by the numerics contract,
so those are additionally required to be bit-identical.
"""
import numpy as np
import pytest

import bhusie_amd as B
from oracle import oracle as O
from tests import common as T

pytestmark = pytest.mark.gpu


def run_gpu(cfg, cam_b, bh_b, det_b, tex, model=None, **kw):
    rp = B.RayPass(cfg, device=0, **kw)
    rp.set_textures(*tex)
    if model is not None:
        rp.upload_model(model)
    rp.set_uniforms(cam_b, bh_b, det_b)
    rp.render()
    return rp


@pytest.mark.parametrize("method", [0, 1])
def test_level0_parity(method):
    tex = T.textures()
    u = T.uniforms(integration_method=method)
    cfg = B.ladder_from_base((72, 41), 3, 1)
    rp = run_gpu(cfg, *u, tex, counters=True)
    got = rp.read_hdr()
    cnt = O.Counters()
    want = O.render_level(T.oracle_scene(*u, tex), (72, 41), None, cnt)
    mx, exact = T.assert_parity(got, want, f"level0 method {method}")
    d = want[..., 3] == 0
    assert np.array_equal(got[d], want[d]), "direction pixels must be bit-identical"
    assert rp.counters() == cnt.as_dict()
    print(f"method {method}: max rel {mx:.3g}, bit-exact {exact:.4f}")


@pytest.mark.parametrize("method", [0, 1])
def test_ladder_parity_all_levels(method):
    tex = T.textures()
    u = T.uniforms(integration_method=method)
    cfg = B.ladder_from_base((24, 14), 3, 3)           # 24x14 -> 70x40 -> 208x118
    rp = run_gpu(cfg, *u, tex, counters=True)
    cnt = O.Counters()
    want = O.render_ladder(T.oracle_scene(*u, tex), cfg.sizes(), cnt)
    for l in range(3):
        T.assert_parity(rp.read_level(l), want[l], f"level {l}")
    T.assert_parity(rp.read_hdr(), want[-1], "frame")
    assert rp.counters() == cnt.as_dict()
    c = cnt.as_dict()
    assert c["copied"] > 0 and c["interpolated"] > 0 and c["traced"] > 0


def test_camera_outside_sphere():
    tex = T.textures()
    cam = B.Camera(position=(0.0, 3.0, -45.0), forward=tuple(np.array([0.0, -3.0, 45.0]) / np.linalg.norm([0.0, -3.0, 45.0])))
    for method in (0, 1):
        u = T.uniforms(camera=cam, integration_method=method)
        cfg = B.ladder_from_base((40, 24), 3, 2)
        rp = run_gpu(cfg, *u, tex)
        want = O.render_ladder(T.oracle_scene(*u, tex), cfg.sizes())
        T.assert_parity(rp.read_hdr(), want[-1], f"outside camera method {method}")


def test_crop_window_and_row_partition():
    tex = T.textures()
    u = T.uniforms(integration_method=1)
    cfg = B.ladder_for_frame((200, 110), 3, 3)
    sizes = cfg.sizes()
    want = O.render_ladder(T.oracle_scene(*u, tex), sizes)[-1]
    cx, cy = int(cfg.crop_x), int(cfg.crop_y)
    want = want[cy:cy + 110, cx:cx + 200]
    full = run_gpu(cfg, *u, tex).read_hdr()
    T.assert_parity(full, want, "cropped frame")
    canvas = np.full_like(want, np.nan)
    for rank in range(3):
        rp = run_gpu(cfg, *u, tex, row_rank=rank, row_world=3, stripe_rows=9)
        rows = rp.local_rows()
        assert len(rows) > 0 and np.all((rows // 9) % 3 == rank)
        canvas[rows] = rp.read_hdr()
    assert np.array_equal(canvas, full), "row-partitioned render must reassemble bit-identically"
