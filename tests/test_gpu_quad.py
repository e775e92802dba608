"""-m gpu: the quad march (bhray_quad.inc: x, y, z of one ray on three lanes of a quad) and wave priority (s_setprio by predicted ray length) deliver
the frames of the scalar kernels, byte for byte, and those frames are the oracle's.

Both are scheduling / layout forms of the same operations, enabled by the host for a ctx that renders one frame at a time (one frame per launch, at
most two frame slots / one).  BHRAY_QUAD (waves per SIMD the quad march may use, 0 = off) and BHRAY_PRIO (0 / 1) are read at bhray_create, so one
process renders every setting.  Covered: both integrators, every ladder mode, frames small enough that EVERY launch is thin (all rays take the quad
path) and large enough that the quad, the scalar thin and the queue paths all run in one frame, a camera outside the sphere (rays start in flat space),
an off-origin tilted hole (NaN pixels of the shader's own arithmetic), the iteration limit, a rank of a row partition, and a queue of exactly 16 / 17
rays per wave (the boundary between one and two rounds)."""
import os

import numpy as np
import pytest

import bhusie_amd as B
from oracle import oracle as O
from tests import common as T

pytestmark = pytest.mark.gpu


def _frame(cfg, u, tex, quad, prio, read="hdr", **kw):
    old = {k: os.environ.get(k) for k in ("BHRAY_QUAD", "BHRAY_PRIO")}
    os.environ["BHRAY_QUAD"], os.environ["BHRAY_PRIO"] = str(quad), str(prio)
    try:
        rp = B.RayPass(cfg, device=0, **kw)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    rp.set_textures(*tex)
    rp.set_uniforms(*u)
    rp.render(); rp.render()                                 # (the second frame of a BHRAY_F_TEMPORAL ctx runs the predicted launch and the thin fix-up launches)
    out = rp.read_hdr() if read == "hdr" else [rp.read_level(l) for l in range(cfg.levels - 1)] + [rp.read_hdr()]
    rp.close()
    return out


def _same(a, b, what):
    if isinstance(a, list):
        for l, (x, y) in enumerate(zip(a, b)):
            _same(x, y, f"{what}, level {l}")
        return
    assert a.shape == b.shape
    diff = a.view(np.uint32) != b.view(np.uint32)
    # (a NaN's sign is not specified: DESIGN.md section 2)
    diff &= ~(np.isnan(a) & np.isnan(b))
    assert not diff.any(), f"{what}: {int(diff.any(axis=-1).sum())} pixels differ"


SCENES = {
    "default": dict(),
    "camera_outside_the_sphere": dict(camera=lambda: B.Camera(position=(0.0, 3.0, -45.0))),
    "off_origin_tilted_hole": dict(black_hole=lambda: B.BlackHole(position=(1.5, -0.75, 2.0), accretion_disk_rotation=(0.6, 0.2, -0.4))),
    "iteration_limit": dict(details=dict(max_iterations=150)),
    "looking_past_the_sphere": dict(camera=lambda: B.Camera(position=(30.0, 2.0, -30.0), forward=(0.0, 0.0, 1.0))),
}


def _uniforms(scene, method):
    s = SCENES[scene]
    cam = s["camera"]() if "camera" in s else None
    bh = s["black_hole"]() if "black_hole" in s else None
    return T.uniforms(camera=cam, black_hole=bh, integration_method=method, **s.get("details", {}))


@pytest.mark.parametrize("method", [1, 0])
@pytest.mark.parametrize("scene", list(SCENES))
def test_quad_march_and_wave_priority_deliver_the_scalar_kernels_frame(scene, method):
    tex = T.textures()
    u = _uniforms(scene, method)
    for size, levels, modes in (((96, 54), 3, [dict(), dict(speculative_levels=2)]),                      # every launch is thin: every ray takes the quad path
                                ((640, 360), 4, [dict(speculative_levels=2), dict(speculative_levels=3), dict(temporal=True), dict(superset_levels=2, speculative_levels=2)])):
        cfg = B.ladder_for_frame(size, 3, levels)
        for kw in modes:
            ref = _frame(cfg, u, tex, 0, 0, read="levels", frames_in_flight=1, **kw)
            for quad, prio in ((1, 0), (2, 1), (4, 1), (0, 1)):
                got = _frame(cfg, u, tex, quad, prio, read="levels", frames_in_flight=1, **kw)
                _same(got, ref, f"{scene} method {method} {size} {kw} BHRAY_QUAD={quad} BHRAY_PRIO={prio}")


@pytest.mark.parametrize("method", [1, 0])
def test_a_frame_whose_every_ray_takes_the_quad_path_is_the_oracles(method):
    tex = T.textures()
    u = T.uniforms(integration_method=method)
    cfg = B.ladder_from_base((24, 14), 3, 3)                                 # 24x14 -> 70x40 -> 208x118: at most 24.5 k rays per launch
    got = _frame(cfg, u, tex, 4, 1, frames_in_flight=1)
    want = O.render_ladder(T.oracle_scene(*u, tex), cfg.sizes())[-1]
    T.assert_parity(got, want, f"quad march, method {method}")
    d = want[..., 3] == 0
    assert np.array_equal(got[d], want[d]), "escape directions are specified to the bit"


def test_the_default_policy_is_one_frame_at_a_time_only_and_the_frames_agree():
    """what the host enables by itself: a ctx with one frame slot (quad + priority), two (quad), four (neither) - the same frame from all three"""
    tex = T.textures()
    u = T.uniforms(integration_method=1)
    cfg = B.ladder_for_frame((320, 180), 3, 3)
    frames = []
    for fif in (1, 2, 4):
        rp = B.RayPass(cfg, device=0, frames_in_flight=fif, speculative_levels=2)
        rp.set_textures(*tex); rp.set_uniforms(*u)
        for _ in range(3):
            rp.render()
        frames.append(rp.read_hdr()); rp.close()
    _same(frames[1], frames[0], "two slots against one")
    _same(frames[2], frames[0], "four slots against one")


def test_queue_lengths_around_the_round_boundaries_and_a_rank_of_a_partition():
    """level 0 of a w x h ladder holds w * h rays: 16 * 1024 = 16 384 is one full round of one wave per SIMD, one more ray needs a second round; and a rank
    of a 3-way partition (its own rows only) with the quad march forced on for its batches"""
    tex = T.textures()
    u = T.uniforms(integration_method=1)
    for base in ((128, 128), (129, 127), (127, 129), (182, 181)):            # 16 384 / 16 383 / 16 383 / 32 942 rays
        cfg = B.ladder_from_base(base, 3, 1)
        _same(_frame(cfg, u, tex, 4, 1, frames_in_flight=1), _frame(cfg, u, tex, 0, 0, frames_in_flight=1), f"level of {base}")
    cfg = B.ladder_for_frame((200, 110), 3, 3)
    for rank in range(3):
        kw = dict(row_rank=rank, row_world=3, slab_row0=[0, 30, 75, 110], frames_in_flight=2, frames_per_batch=3)
        a, b = _frame(cfg, u, tex, 2, 1, **kw), _frame(cfg, u, tex, 0, 0, **kw)
        _same(a, b, f"rank {rank} of 3, batches of 3 frames")
