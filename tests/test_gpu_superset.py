"""-m gpu: bhray_config.superset_levels — the last U ladder levels traced in ONE launch over a conservative superset of the
pixels the shader would trace (tentative classification with PENDING inputs -> one level-tagged queue -> trace -> exact
classification that keeps the traced pixels it wants and overwrites the others).  The frame and every level image must equal the
plain ladder's bit for bit; the copy / interpolate counters are the plain ladder's, the traced count may only be larger."""
import numpy as np
import pytest

import bhusie_amd as B
from oracle import oracle as O
from tests import common as T

pytestmark = pytest.mark.gpu


def _gpu(cfg, u, tex, model=None, **kw):
    rp = B.RayPass(cfg, **kw)
    rp.set_textures(*tex)
    if model is not None:
        rp.upload_model(model)
    rp.set_uniforms(*u)
    rp.render()
    return rp


@pytest.mark.parametrize("method,spec,sup", [(1, 2, 2), (1, 0, 2), (0, 0, 3), (1, 0, 3), (0, 2, 2)])
def test_superset_levels_give_identical_frames(method, spec, sup):
    tex = T.textures()
    u = T.uniforms(integration_method=method)
    cfg = B.ladder_from_base((24, 14), 3, 4)                # 24x14 -> 70x40 -> 208x118 -> 622x352
    normal = _gpu(cfg, u, tex, counters=True)
    fast = _gpu(cfg, u, tex, speculative_levels=spec, superset_levels=sup, counters=True, frames_in_flight=2)
    fast.render(); fast.render()                            # slot reuse: PENDING markers of an earlier frame must not leak
    for l in range(4):
        assert np.array_equal(normal.read_level(l).view(np.uint32), fast.read_level(l).view(np.uint32)), f"level {l}"
    assert np.array_equal(normal.read_hdr().view(np.uint32), fast.read_hdr().view(np.uint32))
    assert not (fast.read_hdr()[..., 3] == 2.0).any()
    cn, cf = normal.counters(), fast.counters()
    for k in ("pixels", "copied", "interpolated"):
        if spec == 0:
            assert cf[k] == cn[k], k                         # (speculative_levels counts its own extra pixels at the coarse end)
    assert cf["traced"] >= cn["traced"]
    print(f"method {method} S={spec} U={sup}: traced {cf['traced']} vs {cn['traced']} for the exact ladder (+{(cf['traced'] - cn['traced']) / cn['traced']:.0%})")
    want = O.render_ladder(T.oracle_scene(*u, tex), cfg.sizes())
    T.assert_parity(fast.read_hdr(), want[-1], "superset frame")
    normal.close(); fast.close()


def test_superset_with_crop_partitions_batches_and_mesh(tmp_path):
    from bhusie_amd import assets
    obj = tmp_path / "m.obj"
    obj.write_text(assets.icosphere_mesh_obj(3, radius=8.0, bump=0.1, seed=4))
    model = B.load_model(str(obj))
    tex = T.textures()
    cam = B.Camera(position=(0.0, 0.0, -40.0), forward=(-0.11914522, 0.0, 0.99287683), fov=1.2)
    u = T.uniforms(camera=cam, integration_method=1, model_count=1)
    cfg = B.ladder_for_frame((200, 110), 3, 3)
    full = _gpu(cfg, u, tex, model=model).read_hdr()
    for rank in range(3):                                   # one ctx = one partition, packed rows
        rp = _gpu(cfg, u, tex, model=model, row_rank=rank, row_world=3, stripe_rows=9, superset_levels=2, frames_in_flight=2, frames_per_batch=2)
        rp.render(); rp.render()
        assert np.array_equal(rp.read_hdr(), full[rp.local_rows()]), f"rank {rank}"
        rp.close()
    tiled = _gpu(cfg, u, tex, model=model, devices=[0, 0, 0, 0], superset_levels=2, frames_in_flight=1)       # in-library gather
    assert np.array_equal(tiled.read_hdr().view(np.uint32), full.view(np.uint32))
    tiled.close()
    for bad in (dict(superset_levels=1), dict(superset_levels=3), dict(superset_levels=2, speculative_levels=2)):
        with pytest.raises(B.BhrayError):
            B.RayPass(cfg, **bad)                            # a 3-level ladder: U = 2 needs the one coarser level to itself


def test_superset_at_the_bench_frame_and_its_latency():
    """1920x1080 RK, one frame at a time (the reference host's mode): S = 2 + U = 2 is two dependent trace launches instead of three."""
    import time
    tex = T.textures(small=False)
    u = T.uniforms(integration_method=1)
    cfg = B.ladder_for_frame((1920, 1080), 3, 4)
    want = _gpu(cfg, u, tex, frames_in_flight=1, speculative_levels=2)
    lat = {}
    for name, kw in (("S2", dict(speculative_levels=2)), ("S2+U2", dict(speculative_levels=2, superset_levels=2)), ("U3", dict(superset_levels=3)), ("S3", dict(speculative_levels=3))):
        rp = _gpu(cfg, u, tex, frames_in_flight=1, counters=(name == "count"), **kw)
        rp.sync()
        ts = []
        for _ in range(12):
            t0 = time.perf_counter(); rp.render(); rp.sync(); ts.append(time.perf_counter() - t0)
        lat[name] = sorted(ts)[len(ts) // 2] * 1e3
        assert np.array_equal(rp.read_hdr().view(np.uint32), want.read_hdr().view(np.uint32)), name
        rp.close()
    print("latency ms per 1920x1080 frame, one frame in flight:", {k: round(v, 3) for k, v in lat.items()})
    for name, kw in (("S2", dict(speculative_levels=2)), ("S2+U2", dict(speculative_levels=2, superset_levels=2)), ("U3", dict(superset_levels=3))):
        rp = _gpu(cfg, u, tex, frames_in_flight=1, counters=True, **kw)
        print(name, "traced rays per frame:", rp.counters()["traced"], rp.scheduling_counters())
        rp.close()
    want.close()
