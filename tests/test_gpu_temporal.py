"""-m gpu: BHRAY_F_TEMPORAL — one launch first traces, at every ladder level, the pixels the previous frame had to trace; the
ladder then traces only what that prediction missed.  Every frame must equal the frame a prediction-less ctx renders from the
same uniforms, bit for bit, whatever the prediction was worth: perfect (static camera), partly right (moving camera), useless
(scene cut, other integrator), absent (first frame)."""
import time

import numpy as np
import pytest

import bhusie_amd as B
from tests import common as T

pytestmark = pytest.mark.gpu


def _plain(cfg, u, tex, **kw):
    rp = B.RayPass(cfg, frames_in_flight=1, **kw)
    rp.set_textures(*tex)
    rp.set_uniforms(*u)
    rp.render()
    out = rp.read_hdr()
    rp.close()
    return out


def _camera_path(n):
    out = []
    for i in range(n):
        a = 0.02 * i
        pos = (19.0 * np.sin(a), 0.3 * i, -19.0 * np.cos(a))
        fwd = tuple(-np.array(pos) / np.linalg.norm(pos))
        out.append(B.Camera(position=tuple(float(v) for v in pos), forward=tuple(float(v) for v in fwd), fov=1.0 + 0.01 * i))
    return out


@pytest.mark.parametrize("fif,fpb", [(1, 1), (3, 1), (2, 2)])
def test_temporal_speculation_never_changes_a_frame(fif, fpb):
    tex = T.textures()
    cfg = B.ladder_from_base((24, 14), 3, 4)
    cams = _camera_path(4)
    seq = ([T.uniforms(integration_method=1)] * 3                                            # static: perfect prediction from the 2nd frame on
           + [T.uniforms(integration_method=1, camera=c, time=0.1 * i) for i, c in enumerate(cams)]      # moving camera, rotating disk
           + [T.uniforms(integration_method=0, camera=cams[-1])] * 2                          # other integrator (kernel variant switch)
           + [T.uniforms(integration_method=1, black_hole=B.BlackHole(relativity_sphere_radius=12.0), camera=B.Camera(position=(0.0, 3.0, -45.0), forward=(0.0, -0.06651901, 0.99778515)))]  # cut
           + [T.uniforms(integration_method=1)])
    want = {}
    rp = B.RayPass(cfg, temporal=True, frames_in_flight=fif, frames_per_batch=fpb, counters=True)
    rp.set_textures(*tex)
    for i, u in enumerate(seq):
        if u not in want:
            want[u] = _plain(cfg, u, tex)
        rp.set_uniforms(*u)
        rp.render()
        got = rp.read_hdr()
        assert np.array_equal(got.view(np.uint32), want[u].view(np.uint32)), f"frame {i}"
    rp.close()


def test_temporal_counters_and_perfect_prediction(monkeypatch):
    """Static camera: from the second frame on the predicted launch delivers every traced pixel — the frame's own trace launches find
    empty queues — and copy / interpolate counts never change.  With the prediction reduced to last frame's traced set
    (BHRAY_TEMPORAL_MARGIN=1, BHRAY_TEMPORAL_RADIUS=0) the only extra rays are the clamped border pixels, which are always predicted
    (their classification hangs on the last bit of a quotient that should be 1: predict_kernel)."""
    tex = T.textures()
    u = T.uniforms(integration_method=1)
    cfg = B.ladder_from_base((24, 14), 3, 4)
    ref = B.RayPass(cfg, counters=True, frames_in_flight=1)
    ref.set_textures(*tex); ref.set_uniforms(*u); ref.render()
    cr = ref.counters()
    border = sum(w + h for (w, h) in cfg.sizes()[1:])
    for env, bound in (({"BHRAY_TEMPORAL_MARGIN": "1", "BHRAY_TEMPORAL_RADIUS": "0"}, border), ({}, None)):      # exact marks only; the defaults
        for k in ("BHRAY_TEMPORAL_MARGIN", "BHRAY_TEMPORAL_RADIUS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        rp = B.RayPass(cfg, temporal=True, counters=True, frames_in_flight=1)
        rp.set_textures(*tex); rp.set_uniforms(*u)
        for i in range(3):
            rp.render()
            c = rp.counters()
            assert (c["pixels"], c["copied"], c["interpolated"]) == (cr["pixels"], cr["copied"], cr["interpolated"])
            assert c["traced"] >= cr["traced"], (i, c, cr)                                    # nothing missed ...
            if i == 0 or bound is not None:                                                   # (frame 0: no marks yet - the plain ladder + the border)
                assert c["traced"] <= cr["traced"] + border, (i, c, cr)                       # ... and nothing traced twice
            if i >= 1:
                assert all(rp.level_counters(l)["traced"] == 0 for l in range(1, 4))          # every fix-up queue was empty
        rp.close()
    with pytest.raises(B.BhrayError):
        B.RayPass(cfg, temporal=True, speculative_levels=2)
    with pytest.raises(B.BhrayError):
        B.RayPass(B.ladder_from_base((24, 14), 3, 5), temporal=True)                         # level tags are 2 bits
    ref.close()


def test_temporal_with_partitions_and_in_library_gather():
    tex = T.textures()
    cfg = B.ladder_for_frame((200, 110), 3, 3)
    cams = _camera_path(3)
    rp = B.RayPass(cfg, devices=[0, 0, 0], temporal=True, frames_in_flight=2, stripe_rows=9)
    rp.set_textures(*tex)
    for i, cam in enumerate(cams + cams[::-1]):
        u = T.uniforms(integration_method=1, camera=cam)
        rp.set_uniforms(*u)
        rp.render()
        assert np.array_equal(rp.read_hdr().view(np.uint32), _plain(cfg, u, tex).view(np.uint32)), f"frame {i}"
    rp.close()


def test_temporal_latency_at_the_bench_frame():
    """1920x1080 RK, one frame at a time: static camera (perfect prediction) and a moving camera (the prediction is one frame old)."""
    tex = T.textures(small=False)
    cfg = B.ladder_for_frame((1920, 1080), 3, 4)
    u0 = T.uniforms(integration_method=1)
    want = _plain(cfg, u0, tex)

    def median_latency(rp, uniforms):
        ts = []
        for u in uniforms:
            rp.set_uniforms(*u)
            t0 = time.perf_counter(); rp.render(); rp.sync(); ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2] * 1e3

    cams = _camera_path(12)
    moving = [T.uniforms(integration_method=1, camera=c) for c in cams]
    res = {}
    for name, kw in (("ladder", dict()), ("S2", dict(speculative_levels=2)), ("temporal", dict(temporal=True))):
        rp = B.RayPass(cfg, frames_in_flight=1, **kw)
        rp.set_textures(*tex); rp.set_uniforms(*u0); rp.render(); rp.render(); rp.sync()
        res[name + " static"] = round(median_latency(rp, [u0] * 12), 3)
        assert np.array_equal(rp.read_hdr().view(np.uint32), want.view(np.uint32)), name
        res[name + " moving"] = round(median_latency(rp, moving), 3)
        rp.close()
    print("latency ms per 1920x1080 frame, one frame in flight:", res)
    assert res["temporal static"] < res["S2 static"]


def test_moving_camera_leaves_one_fixup_launch_at_the_bench_frame():
    """The prediction's superset (threshold margin, radius, clamped border pixels: predict_kernel) is sized so that a camera in motion
    misses nothing below the last level — the frame then costs ONE fix-up launch, not one per level.  Orbit of 0.002 rad + 0.03 up per
    frame (half a pixel of the last level) and bench.py's 0.02 rad + 0.3: levels 1 and 2 trace nothing in their fix-up launches, the
    last level a few hundred aliased pixels, and every frame equals the plain ladder's."""
    tex = T.textures(small=False)
    cfg = B.ladder_for_frame((1920, 1080), 3, 4)
    for da, dy in ((0.002, 0.03), (0.02, 0.3)):
        rp = B.RayPass(cfg, temporal=True, frames_in_flight=1, counters=True)
        rp.set_textures(*tex)
        for i in range(6):
            a = da * i
            pos = (19.0 * np.sin(a), dy * i, -19.0 * np.cos(a))
            fwd = tuple(float(v) for v in -np.array(pos) / np.linalg.norm(pos))
            u = T.uniforms(integration_method=1, camera=B.Camera(position=tuple(float(v) for v in pos), forward=fwd))
            rp.set_uniforms(*u)
            rp.render()
            if i >= 2:
                fix = [rp.level_counters(l)["traced"] for l in (1, 2, 3)]
                assert fix[0] == 0 and fix[1] == 0, (da, i, fix)
                assert 0 < fix[2] < 4000, (da, i, fix)
            if i == 5:
                assert np.array_equal(rp.read_hdr().view(np.uint32), _plain(cfg, u, tex).view(np.uint32))
        rp.close()


def test_more_than_22_frame_slots_are_served_with_22():
    """bhray_create clamps the slots (24+ put the HIP runtime into an intermittent stall): a ctx asked for 28 behaves,
    frames are the single-slot frames."""
    tex = T.textures()
    cfg = B.ladder_from_base((24, 14), 3, 3)
    u = T.uniforms(integration_method=1)
    want = _plain(cfg, u, tex)
    rp = B.RayPass(cfg, frames_in_flight=28)
    rp.set_textures(*tex); rp.set_uniforms(*u)
    for _ in range(60):
        rp.render()
    assert np.array_equal(rp.read_hdr().view(np.uint32), want.view(np.uint32))
    rp.close()
    with pytest.raises(B.BhrayError):
        B.RayPass(cfg, frames_in_flight=33)

