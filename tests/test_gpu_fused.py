"""-m gpu: the fused ladder (BHRAY_F_FUSED) - ONE persistent launch per batch runs every ladder level; the grid classification
(ray.wgsl:167-243) is folded into the trace kernel as tile work items with explicit dependencies.  Scheduling only: every frame must
equal, byte for byte, the frame the launch-per-level path delivers, in every combination the path supports (both integrators,
speculative levels, frame batches with different uniforms per frame, slot reuse, row partitions + in-library gather, the mesh
variant, a camera outside the relativity sphere, the literal evaluation), and the classification counters must agree."""
import numpy as np
import pytest

import bhusie_amd as B
from tests import common as T

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


@pytest.fixture(scope="module", autouse=True)
def fused_library():
    """The fused ladder is a build option (`make -C bhusie_amd/csrc fused`, also run by __graft_entry__.build()): this module runs
    against bhusie_amd/libbhray_fused.so - the same sources with -DBHRAY_WITH_FUSED=1 - for the fused AND the launch-per-level frames
    it compares; the default library refuses BHRAY_F_FUSED (test_the_default_library_refuses_the_flag)."""
    import ctypes as C
    import os
    from bhusie_amd import _lib, layouts
    path = T.variant_library("fused")
    saved = _lib.lib()
    L = C.CDLL(path)
    layouts.declare(L)
    _lib._lib = L
    yield saved
    _lib._lib = saved


def test_the_default_library_refuses_the_flag(fused_library):
    import ctypes as C
    from bhusie_amd import layouts
    cfg = B.ladder_from_base((24, 14), 3, 3)
    cfg.struct_size = C.sizeof(layouts.BhrayConfig)
    cfg.flags = layouts.F_FUSED
    h = C.c_void_p()
    rc = fused_library.bhray_create(C.byref(cfg), C.byref(h))
    assert rc == -1 and h.value is None                       # BHRAY_E_INVALID, with a message that names the build option
    assert b"make" in fused_library.bhray_last_error(None)


def _render(cfg, u, tex, model=None, **kw):
    rp = B.RayPass(cfg, device=0, **kw)
    rp.set_textures(*tex)
    if model is not None:
        rp.upload_model(model)
    rp.set_uniforms(*u)
    rp.render()
    return rp


def _same(a, b, what):
    assert a.shape == b.shape, what
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"{what}: {int((a.view(np.uint32) != b.view(np.uint32)).any(axis=-1).sum())} pixels differ"


@pytest.mark.parametrize("method", [1, 0])
@pytest.mark.parametrize("spec", [0, 2])
def test_fused_ladder_equals_the_launch_per_level_ladder(method, spec):
    tex = T.textures()
    u = T.uniforms(integration_method=method)
    cfg = B.ladder_from_base((24, 14), 3, 3)
    one = _render(cfg, u, tex, counters=True, speculative_levels=spec)
    fz = _render(cfg, u, tex, counters=True, speculative_levels=spec, fused=True)
    for l in range(3):
        _same(fz.read_level(l), one.read_level(l), f"level {l} method {method} spec {spec}")
    _same(fz.read_hdr(), one.read_hdr(), "frame")
    a, b = one.counters(), fz.counters()
    for k in ("pixels", "copied", "interpolated", "traced", "steps", "flat_iters", "disk_hits", "sky_samples"):
        assert a[k] == b[k], (k, a[k], b[k])
    one.close(); fz.close()


@pytest.mark.parametrize("fif", [1, 3])
def test_fused_slot_reuse_and_changing_uniforms(fif):
    tex = T.textures()
    cam2 = B.Camera(position=(1.0, 2.0, -17.0), forward=(0.0, -0.1, 1.0), fov=1.1)
    frames = [T.uniforms(integration_method=1), T.uniforms(integration_method=1, camera=cam2, time=0.7), T.uniforms(integration_method=0, step_size=0.2)]
    cfg = B.ladder_for_frame((200, 110), 3, 3)
    want = []
    for u in frames:
        r = _render(cfg, u, tex); want.append(r.read_hdr()); r.close()
    rp = B.RayPass(cfg, device=0, frames_in_flight=fif, fused=True, speculative_levels=2)
    rp.set_textures(*tex)
    for i in range(9):
        rp.set_uniforms(*frames[i % 3])
        rp.render()
        if i % 2 == 0 or i == 8:
            _same(rp.read_hdr(), want[i % 3], f"frame {i}")
    rp.close()


def test_fused_frame_batches():
    tex = T.textures()
    cam2 = B.Camera(position=(1.0, 2.0, -17.0), forward=(0.0, -0.1, 1.0), fov=1.1)
    frames = [T.uniforms(integration_method=1), T.uniforms(integration_method=1, camera=cam2, time=0.7)]
    cfg = B.ladder_from_base((24, 14), 3, 3)
    want = []
    for u in frames:
        r = _render(cfg, u, tex); want.append(r.read_hdr()); r.close()
    h, w = want[0].shape[:2]
    fb = h * w * 16
    rp = B.RayPass(cfg, device=0, frames_in_flight=2, frames_per_batch=4, fused=True, speculative_levels=2)
    rp.set_textures(*tex)
    order = [0, 1, 1, 0, 0, 1, 0, 1, 1, 0]
    out = T.DeviceBuffer(len(order) * fb)
    for i, f in enumerate(order):
        rp.set_uniforms(*frames[f])
        rp.bind_output(out.ptr.value + i * fb, fb)
        rp.render()
    rp.sync()
    got = out.read().reshape(len(order), h, w, 4)
    for i, f in enumerate(order):
        _same(got[i], want[f], f"frame {i} of the batches")
    rp.close(); out.free()


def test_fused_bench_frame_1920x1080():
    tex = T.textures(small=False)
    u = T.uniforms(integration_method=1)
    cfg = B.ladder_for_frame((1920, 1080), 3, 4)
    one = _render(cfg, u, tex, frames_in_flight=1)
    want = one.read_hdr(); one.close()
    for spec in (2, 0):
        rp = B.RayPass(cfg, device=0, frames_in_flight=1, fused=True, speculative_levels=spec)
        rp.set_textures(*tex); rp.set_uniforms(*u)
        for _ in range(3):
            rp.render()
        _same(rp.read_hdr(), want, f"1920x1080 fused, speculative levels {spec}")
        rp.close()


def test_fused_row_partitions_with_the_in_library_gather():
    tex = T.textures()
    u = T.uniforms(integration_method=1)
    cfg = B.ladder_for_frame((200, 110), 3, 3)
    one = _render(cfg, u, tex)
    want = one.read_hdr(); one.close()
    rp = B.RayPass(cfg, devices=[0, 0, 0], stripe_rows=9, frames_in_flight=2, frames_per_batch=2, speculative_levels=2, fused=True)
    rp.set_textures(*tex); rp.set_uniforms(*u)
    for _ in range(5):
        rp.render()
    _same(rp.read_hdr(), want, "3 partitions, fused")
    rp.close()


def test_fused_camera_outside_the_sphere_and_literal_evaluation():
    tex = T.textures()
    cam = B.Camera(position=(0.0, 3.0, -45.0), forward=(0.0, -0.06651901, 0.99778515))
    cfg = B.ladder_from_base((40, 24), 3, 3)
    for kw in (dict(), dict(literal=True)):
        u = T.uniforms(camera=cam, integration_method=1)
        one = _render(cfg, u, tex, **kw)
        fz = _render(cfg, u, tex, fused=True, speculative_levels=2, **kw)
        _same(fz.read_hdr(), one.read_hdr(), f"outside the sphere {kw}")
        one.close(); fz.close()


def test_fused_mesh_variant(tmp_path):
    from bhusie_amd import assets
    obj = tmp_path / "m.obj"
    obj.write_text(assets.icosphere_mesh_obj(3, radius=8.0, bump=0.1, seed=4))
    model = B.load_model(str(obj))
    tex = T.textures()
    cam = B.Camera(position=(0.0, 0.0, -40.0), forward=(-0.11914522, 0.0, 0.99287683), fov=1.2)
    u = T.uniforms(camera=cam, integration_method=1, model_count=1)
    cfg = B.ladder_for_frame((200, 110), 3, 3)
    one = _render(cfg, u, tex, model=model, counters=True)
    fz = _render(cfg, u, tex, model=model, counters=True, fused=True)
    _same(fz.read_hdr(), one.read_hdr(), "mesh")
    assert one.counters()["triangles"] == fz.counters()["triangles"] > 0
    one.close(); fz.close()


def test_fused_refuses_what_it_does_not_support():
    cfg = B.ladder_from_base((24, 14), 3, 3)
    with pytest.raises(B.BhrayError):
        B.RayPass(cfg, device=0, fused=True, temporal=True)
    with pytest.raises(B.BhrayError):
        B.RayPass(cfg, device=0, fused=True, superset_levels=2)
    with pytest.raises(B.BhrayError):
        B.RayPass(B.ladder_for_frame((7680, 4320), 3, 5), device=0, fused=True)
