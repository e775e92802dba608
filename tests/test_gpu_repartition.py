"""-m gpu: the partition of a ctx changed at run time (bhray_set_partition) and balanced by the ctx's own measurements
(bhray_rebalance) - the reference's camera moves every frame (src/app.rs:98-102), so the rows that hold the work move too.

Every frame of a partitioned ctx, before and after any change of its bounds, is compared byte for byte with the frame ONE
partition-less ctx renders from the same uniforms."""
import math

import numpy as np
import pytest

import bhusie_amd as B
from tests import common as T

pytestmark = pytest.mark.gpu


def _pitching_camera(i, n, amplitude=0.45):
    """a camera that looks further up frame by frame: the hole's projection travels down the frame"""
    a = -amplitude + 2.0 * amplitude * i / max(1, n - 1)
    return B.Camera(position=(0.0, 0.0, -19.0), forward=(0.0, math.sin(a), math.cos(a)))


def _frames_of_one_ctx(cfg, tex, uniforms, **kw):
    one = B.RayPass(cfg, device=0, **kw)
    one.set_textures(*tex)
    out = []
    for u in uniforms:
        one.set_uniforms(*u); one.render(); out.append(one.read_hdr())
    one.close()
    return out


@pytest.mark.parametrize("kw", [dict(), dict(frames_per_batch=3, frames_in_flight=2, speculative_levels=2), dict(temporal=True, frames_in_flight=2),
                                dict(gather_root=2, frames_in_flight=3)])
def test_set_partition_moves_the_bounds_of_a_multi_device_ctx_and_the_frames_stay_the_same(kw):
    tex = T.textures()
    cfg = B.ladder_for_frame((200, 110), 3, 3)
    frames = [T.uniforms(integration_method=1, time=0.1 * i, camera=_pitching_camera(i, 9)) for i in range(9)]
    want = _frames_of_one_ctx(cfg, tex, frames)
    rp = B.RayPass(cfg, devices=[0] * 4, slab_row0=[0, 30, 55, 80, 110], **kw)
    rp.set_textures(*tex)
    plans = [None, None, [0, 10, 60, 61, 110], None, [0, 0, 5, 110, 110], [0, 27, 54, 81, 110], None, [0, 100, 104, 108, 110], None]
    for i, (u, plan) in enumerate(zip(frames, plans)):
        if plan is not None:
            rp.set_partition(plan)
            assert rp.get_partition() == plan
        rp.set_uniforms(*u); rp.render()
        if i % 2 == 0 or plan is not None or i == len(frames) - 1:
            assert np.array_equal(rp.read_hdr().view(np.uint32), want[i].view(np.uint32)), (kw, i, plan)
    assert rp.gather_info()["bytes_received_per_frame"] == (110 - (108 - 104 if kw.get("gather_root") == 2 else 100)) * (3 * 200 + 7) * 4
    with pytest.raises(B.BhrayError):
        rp.set_partition([0, 50, 40, 80, 110])
    with pytest.raises(B.BhrayError):
        rp.set_partition([0, 30, 55, 80, 109])
    rp.set_uniforms(*frames[0]); rp.render()                     # a refused partition changes nothing
    assert np.array_equal(rp.read_hdr().view(np.uint32), want[0].view(np.uint32))
    rp.close()


def test_set_partition_on_a_ctx_created_with_stripes_and_on_one_rank_of_a_partition():
    tex = T.textures()
    cfg = B.ladder_for_frame((200, 110), 3, 3)
    u = T.uniforms(integration_method=0)
    want = _frames_of_one_ctx(cfg, tex, [u])[0]
    rp = B.RayPass(cfg, devices=[0] * 3, stripe_rows=7, frames_in_flight=2)
    rp.set_textures(*tex); rp.set_uniforms(*u); rp.render()
    assert np.array_equal(rp.read_hdr().view(np.uint32), want.view(np.uint32))
    with pytest.raises(B.BhrayError):
        rp.get_partition()                                       # interleaved stripes have no bounds
    rp.set_partition([0, 40, 70, 110])
    with pytest.raises(ValueError):                        # a short list never reaches the library
        rp.set_partition([0, 40, 110])
    rp.render()
    assert np.array_equal(rp.read_hdr().view(np.uint32), want.view(np.uint32))
    rp.close()
    # one process per GPU without the gather: a rank's packed rows follow its bounds (an empty slab, then a larger one than it was created with)
    rk = B.RayPass(cfg, device=0, row_rank=1, row_world=3, slab_row0=[0, 50, 60, 110])
    rk.set_textures(*tex); rk.set_uniforms(*u)
    for bounds in ([0, 50, 60, 110], [0, 20, 20, 110], [0, 5, 100, 110], [0, 50, 60, 110]):
        rk.set_partition(bounds)
        rk.render()
        rows = rk.local_rows()
        assert rows.tolist() == list(range(bounds[1], bounds[2]))
        got = rk.read_hdr()
        assert got.shape[0] == len(rows)
        if len(rows):
            assert np.array_equal(got.view(np.uint32), want[rows].view(np.uint32)), bounds
    rk.close()


def test_gathered_sky_image_survives_a_change_of_partition():
    tex = T.textures()
    cfg = B.ladder_for_frame((200, 110), 3, 3)
    u = T.uniforms(integration_method=1)
    one = B.RayPass(cfg, device=0); one.set_textures(*tex); one.set_uniforms(*u); one.render(); one.resolve_sky(); want = one.read_sky(); one.close()
    rp = B.RayPass(cfg, devices=[0] * 3, slab_row0=[0, 40, 70, 110], gather_sky=True, frames_per_batch=2, frames_in_flight=2)
    rp.set_textures(*tex); rp.set_uniforms(*u)
    for bounds in (None, [0, 10, 100, 110], [0, 60, 60, 110]):
        if bounds:
            rp.set_partition(bounds)
        for _ in range(3):
            rp.render()
        assert np.array_equal(rp.read_sky().view(np.uint16), want.view(np.uint16)), bounds
    rp.close()


def test_rebalance_follows_a_pitching_camera_and_every_frame_is_the_undivided_frame():
    """200 frames of a camera that pitches so that the hole's projection crosses most of the frame; the ctx re-balances itself every
    20 frames from the work its kernels count (steps issued by the trace waves + classified pixels: no counting build, no calibration
    frame, no timing flags).  Every 10th frame and every frame around a change of bounds is compared with the undivided frame; the
    bounds follow the hole."""
    tex = T.textures()
    cfg = B.ladder_for_frame((320, 180), 3, 3)
    n = 200
    cams = [_pitching_camera(i, n, amplitude=0.40) for i in range(n)]
    frames = [T.uniforms(integration_method=1, time=i / 60.0, camera=cams[i]) for i in range(n)]
    check = sorted(set(list(range(0, n, 10)) + [i + d for i in range(20, n, 20) for d in (-1, 0, 1)] + [n - 1]))
    one = B.RayPass(cfg, device=0); one.set_textures(*tex)
    want = {}
    for i in check:
        one.set_uniforms(*frames[i]); one.render(); want[i] = one.read_hdr()
    one.close()
    rp = B.RayPass(cfg, devices=[0] * 4, slab_row0=[0, 45, 90, 135, 180], frames_in_flight=3, speculative_levels=2)
    rp.set_textures(*tex)
    history, applied = [], 0
    for i in range(n):
        if i and i % 20 == 0:
            cost, extra = rp.partition_costs()                   # the numbers a launcher with a channel of its own would sum over its ranks
            info = rp.rebalance()
            assert np.allclose(cost, info["part_cost"], rtol=1e-5) and np.allclose(extra, info["extra_cost"], rtol=1e-5), (cost, info)
            assert info["partitions"] == 4 and info["frames"] > 0 and min(info["part_cost"]) > 0.0 and info["extra_cost"][0] > 0.0
            assert info["slab_row0"][0] == 0 and info["slab_row0"][-1] == 180 and info["slab_row0"] == rp.get_partition()
            applied += info["applied"]
            history.append(info["slab_row0"])
        rp.set_uniforms(*frames[i]); rp.render()
        if i in want:
            assert np.array_equal(rp.read_hdr().view(np.uint32), want[i].view(np.uint32)), (i, history[-1:] )
    rp.close()
    assert applied >= 3, history
    # the thin slabs sit where the hole is: the middle bounds travel with the hole's projection
    mids = [h[2] for h in history]
    assert max(mids) - min(mids) >= 20, history


def test_work_counters_count_the_steps_the_waves_issue():
    """bhray_get_work: the integrator steps the trace waves issued per frame (in whole batches of 16) - between the frame's ray
    iterations / 64 (perfectly packed waves) and its ray iterations (one ray per wave; + the rounding to batches); about the same
    whether two or three frames are in flight (the same kernel build packs the same rays the same way); and a slab of the sky costs
    less than a slab of the hole."""
    tex = T.textures()
    cfg = B.ladder_for_frame((320, 180), 3, 3)
    u = T.uniforms(integration_method=1)
    res = []
    for fif in (2, 3):
        rp = B.RayPass(cfg, device=0, frames_in_flight=fif)
        rp.set_textures(*tex); rp.set_uniforms(*u)
        for _ in range(8):
            rp.render()
        ws, px, n = rp.work()
        rp.close()
        assert n == fif and ws > 0 and px >= 320 * 180, (fif, ws, px, n)
        res.append(ws)
    assert abs(res[0] - res[1]) <= 0.25 * res[0], res
    rc = B.RayPass(cfg, device=0, frames_in_flight=1, counters=True)
    rc.set_textures(*tex); rc.set_uniforms(*u); rc.render()
    c = rc.counters(); wsc, _, _ = rc.work()
    rc.close()
    assert c["steps"] / 64.0 <= wsc <= 1.2 * c["steps"] + 16 * c["traced"], (wsc, c["steps"], c["traced"])
    parts = []
    for rank in range(3):
        rk = B.RayPass(cfg, device=0, row_rank=rank, row_world=3, slab_row0=[0, 40, 140, 180])
        rk.set_textures(*tex); rk.set_uniforms(*u); rk.render()
        parts.append(rk.work()[0]); rk.close()
    assert parts[1] > 1.15 * parts[0] and parts[1] > 1.15 * parts[2], parts           # (each rank also traces the coarse rows under its slab: small frames share much)
