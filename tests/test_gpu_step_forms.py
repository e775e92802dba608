"""-m gpu: the UNIFIED march of the shipped no-mesh contract kernels (bhray_step_u.inc: pairs of steps over two position register sets, the state the
other phases read written when a lane leaves the march) against the GENERAL step (bhray_step.inc) - the same sources built with -DBHRAY_UNIFIED=0
(`make -C bhusie_amd/csrc general` -> libbhray_general.so, built by __graft_entry__.build(); test infrastructure).  Same operations on the same values in
the same order per ray, so every frame must be the same BYTES: the scenes below are the ones in which the two forms take different code paths - disk hits
(a lane leaves and re-enters the march), a camera outside the sphere (the Cash-Karp ray and the hit-test ray differ at the entry, ray.wgsl keeps two),
iteration limits of both parities and 0, feather 0 (NaN directions), a seeded sweep of the UI's uniform space - rendered by the latency builds (one frame)
and by the dense builds (a full set of frame slots).  The oracle parity of the shipped kernels is every other -m gpu test's business."""
import ctypes as C

import numpy as np
import pytest

import bhusie_amd as B
from tests import common as T

pytestmark = pytest.mark.gpu


def frames_of(scenes, tex, **kw):
    out = []
    for cfg, u, n in scenes:
        rp = B.RayPass(cfg, device=0, **kw)
        rp.set_textures(*tex)
        rp.set_uniforms(*u)
        for _ in range(n):
            rp.render()
        rp.sync()
        out.append(rp.read_hdr().copy())
        rp.close()
    return out


@pytest.fixture(params=["general", "redo"])
def general_library(request):
    """`general`: the general step everywhere (-DBHRAY_UNIFIED=0).  `redo`: the shipped sources with the one-test step's exact second pass forced on an eighth of the
    steps (-DBHRAY_ONE_TEST=5; the latency builds): the frames must still be the product's, byte for byte - the path real scenes almost never take."""
    from bhusie_amd import _lib, layouts
    path = T.variant_library(request.param)
    saved = _lib.lib()
    L = C.CDLL(path)
    layouts.declare(L)

    def use(general: bool):
        _lib._lib = L if general else saved
    yield use
    _lib._lib = saved


def same_bytes(a, b, what):
    assert len(a) == len(b)
    for i, (x, y) in enumerate(zip(a, b)):
        assert x.shape == y.shape
        d = x.view(np.uint32) != y.view(np.uint32)
        assert not d.any(), f"{what}, scene {i}: {int(d.any(axis=-1).sum())} pixels differ between the unified march and the general step, first at {np.argwhere(d.any(axis=-1))[:3].tolist()}"


def edge_scenes(levels=2):
    cfg = B.ladder_from_base((20, 12) if levels == 2 else (10, 6), 3, levels)
    sc = []
    for method in (0, 1):
        for mi in (0, 1, 2, 5, 6, 7, 40, 41, 291, 496):                                    # the iteration limit on either step of a pair, and before the first
            sc.append((cfg, T.uniforms(integration_method=method, max_iterations=mi), 1))
            sc.append((cfg, T.uniforms(integration_method=method, max_iterations=mi, step_size=0.05), 1))
        outside = B.Camera(position=(0.0, 3.0, -45.0), forward=(0.0, -3.0 / 45.1, 45.0 / 45.1), fov=1.0)
        sc.append((cfg, T.uniforms(camera=outside, integration_method=method), 1))          # flat -> relativity -> flat: the sphere entry moves cpos, not the Cash-Karp ray
        sc.append((cfg, T.uniforms(camera=outside, integration_method=method, black_hole=B.BlackHole(relativity_sphere_radius=9.0, accretion_disk_outer=14.0)), 1))   # disk outside the sphere
        sc.append((cfg, T.uniforms(integration_method=method, black_hole=B.BlackHole(feather_amount=0.0)), 1))      # 0/0 at the exit: NaN directions (ray.wgsl:548)
        sc.append((cfg, T.uniforms(integration_method=method, black_hole=B.BlackHole(position=(3.0, -2.0, 5.0))), 1))     # hole off the origin
        sc.append((cfg, T.uniforms(integration_method=method, step_size=2.5), 1))           # steps that jump over the hole
    return sc


def fuzz_scenes(n=24, levels=2):
    rng = np.random.default_rng(20260929)
    cfg = B.ladder_from_base((20, 12) if levels == 2 else (10, 6), 3, levels)
    sc = []
    for _ in range(n):
        pos = rng.normal(size=3) * np.array([6.0, 4.0, 6.0]) + np.array([0.0, 0.0, -16.0])
        fwd = -pos + rng.normal(size=3) * 4.0
        fwd = fwd / np.linalg.norm(fwd)
        inner = float(rng.uniform(1.2, 4.0))
        bh = B.BlackHole(accretion_disk_rotation=tuple(rng.uniform(-1.5, 1.5, size=3)), accretion_disk_inner=inner, accretion_disk_outer=inner + float(rng.uniform(1.0, 12.0)),
                         rotation_speed=float(rng.uniform(0, 10)), relativity_sphere_radius=float(rng.uniform(8.0, 40.0)), feather_amount=float(rng.uniform(0.05, 1.0)))
        sc.append((cfg, T.uniforms(camera=B.Camera(position=tuple(pos), forward=tuple(fwd), fov=float(rng.uniform(0.3, 2.2))), black_hole=bh,
                                   integration_method=int(rng.integers(0, 2)), step_size=float(rng.uniform(0.05, 0.6)), max_iterations=int(rng.integers(50, 900)),
                                   angle_division_threshold=float(rng.uniform(0.0, 0.2)), time=float(rng.uniform(0, 100))), 1))
    return sc


def test_unified_march_equals_the_general_step_latency_builds(general_library):
    tex = T.textures()
    for levels, kw in ((2, dict(speculative_levels=0)), (3, dict(speculative_levels=2, frames_in_flight=1))):
        scenes = edge_scenes(levels) + fuzz_scenes(24, levels)
        general_library(False); a = frames_of(scenes, tex, **kw)
        general_library(True); b = frames_of(scenes, tex, **kw)
        general_library(False)
        same_bytes(a, b, f"latency builds {kw}")


@pytest.mark.parametrize("method", [0, 1])
def test_unified_march_equals_the_general_step_dense_builds(general_library, method):
    """A ctx with a full set of frame slots at 1920x1080 marches with the dense builds (bhray_api: which trace build a batch gets); the last of 24 frames."""
    tex = T.textures()
    cfg = B.ladder_for_frame((1920, 1080), 3, 4)
    scenes = [(cfg, T.uniforms(integration_method=method), 24),
              (cfg, T.uniforms(integration_method=method, max_iterations=301, camera=B.Camera(position=(2.0, 1.0, -25.0), forward=(-0.08, -0.04, 1.0), fov=1.3)), 24),
              (cfg, T.uniforms(integration_method=method, black_hole=B.BlackHole(position=(1.5, -0.75, 2.0))), 24),      # hole off the origin: the pairs that form position - bpos
              (cfg, T.uniforms(integration_method=method, black_hole=B.BlackHole(position=(-0.0, 0.0, 0.0))), 24)]       # a NEGATIVE zero is not the origin path's case (x - (-0) is not x for x = -0)
    general_library(False); a = frames_of(scenes, tex, frames_in_flight=22, speculative_levels=2)
    general_library(True); b = frames_of(scenes, tex, frames_in_flight=22, speculative_levels=2)
    general_library(False)
    same_bytes(a, b, f"dense builds, method {method}")


@pytest.mark.parametrize("method", [0, 1])
def test_unified_march_equals_the_general_step_mesh_variant(general_library, tmp_path, method):
    """The mesh variant's kernels march unified too (the flat phase's BVH traversal between batches finds the general step's state): latency build (one frame)
    and the dense build with its parked traversal (a full set of slots), a mesh in front of and beside the hole."""
    from bhusie_amd import assets
    tex = T.textures()
    p = tmp_path / "mesh.obj"
    p.write_text(assets.icosphere_mesh_obj(3, radius=6.0, bump=0.2, seed=11))
    model = B.load_model(str(p))
    model.set_transform((-7.0, 1.0, 24.0), 1)

    def frames(cfg, u, n, **kw):
        rp = B.RayPass(cfg, device=0, **kw)
        rp.set_textures(*tex); rp.upload_model(model); rp.set_uniforms(*u)
        for _ in range(n):
            rp.render()
        rp.sync()
        f = rp.read_hdr().copy()
        rp.close()
        return f
    small = B.ladder_from_base((24, 14), 3, 2)
    big = B.ladder_for_frame((1920, 1080), 3, 4)
    cases = [(small, T.uniforms(integration_method=method, model_count=1), 1, dict(speculative_levels=0)),
             (small, T.uniforms(integration_method=method, model_count=1, camera=B.Camera(position=(0.0, 3.0, -45.0), forward=(0.0, -3.0 / 45.1, 45.0 / 45.1), fov=1.0)), 1, dict(speculative_levels=0)),
             (big, T.uniforms(integration_method=method, model_count=1), 24, dict(frames_in_flight=22, speculative_levels=2))]
    general_library(False); a = [frames(c, u, n, **kw) for c, u, n, kw in cases]
    general_library(True); b = [frames(c, u, n, **kw) for c, u, n, kw in cases]
    general_library(False)
    same_bytes(a, b, f"mesh variant, method {method}")
