"""-m gpu: the pair march (bhusie_amd/csrc/bhray_pair.inc, `make pair` -> libbhray_pair.so): two rays per lane, the Cash-Karp step on
packed FP32, in the dense RK kernel without meshes.  It is the same operations per ray in the same order, so every frame, every ladder
level and every frame counter must equal, byte for byte, what the default library renders - in every way that kernel is launched (one
level per launch, speculative multi-level launches, temporal speculation, frame batches with different uniforms per frame, row
partitions) and on the scenes that reach its rare paths (disk hits that resume the march, rays that leave and re-enter the sphere, a
camera outside the sphere, rays that use up their iterations at every point of their life)."""
import contextlib
import ctypes as C
import os

import numpy as np
import pytest

import bhusie_amd as B
from bhusie_amd import _lib, layouts
from tests import common as T

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]
PAIR = os.path.join(os.path.dirname(os.path.abspath(_lib.__file__)), "libbhray_pair.so")
_loaded = {}


@contextlib.contextmanager
def library(path):
    """Run the body against another in-tree build of the same sources; BHRAY_TRACE_DENSE=1 (read by bhray_create) makes every trace
    launch of the ctx the dense build - the kernel the pair march replaces - whatever the size of the frame."""
    if path not in _loaded:
        if path == PAIR:
            T.variant_library("pair")                              # built here if a fresh checkout lacks it
        L = C.CDLL(path)
        layouts.declare(L)
        _loaded[path] = L
    saved, old = _lib.lib(), os.environ.get("BHRAY_TRACE_DENSE")
    _lib._lib = _loaded[path]
    os.environ["BHRAY_TRACE_DENSE"] = "1"
    try:
        yield
    finally:
        _lib._lib = saved
        if old is None:
            del os.environ["BHRAY_TRACE_DENSE"]
        else:
            os.environ["BHRAY_TRACE_DENSE"] = old


def _render_all(cfg, frames, tex, **kw):
    """every level of every frame + the frame counters"""
    out = []
    rp = B.RayPass(cfg, device=0, counters=True, **kw)
    rp.set_textures(*tex)
    for u in frames:
        rp.set_uniforms(*u)
        rp.render()
        lv = [rp.read_level(l) for l in range(cfg.levels)]
        sched = rp.scheduling_counters()
        out.append((lv, rp.read_hdr(), rp.counters(), sched["max_ray_iterations"], sched["wave_steps"]))
    rp.close()
    return out


def _compare(cfg, frames, tex, what, **kw):
    with library(_lib.LIB_PATH):
        want = _render_all(cfg, frames, tex, **kw)
    with library(PAIR):
        got = _render_all(cfg, frames, tex, **kw)
    for i, (g, w) in enumerate(zip(got, want)):
        for l, (a, b) in enumerate(zip(g[0], w[0])):
            bad = (a.view(np.uint32) != b.view(np.uint32)).any(axis=-1)
            assert not bad.any(), f"{what}: frame {i} level {l}: {int(bad.sum())} of {bad.size} pixels differ, first at {np.argwhere(bad)[0]}"
        assert np.array_equal(g[1].view(np.uint32), w[1].view(np.uint32)), f"{what}: frame {i}"
        assert g[2] == w[2], f"{what}: frame {i} counters {g[2]} != {w[2]}"
        assert g[3] == w[3], f"{what}: frame {i} longest ray"
    return want


def _scenes():
    far = B.Camera(position=(0.0, 3.0, -45.0), forward=(0.0, -0.0665190, 0.9977851))        # outside the relativity sphere (radius 20)
    side = B.Camera(position=(2.0, 1.5, -24.0), forward=(-0.0830455, -0.0622841, 0.9945987))
    near = B.Camera(position=(1.0, 0.6, -7.0), forward=(-0.1, -0.05, 1.0), fov=1.3)         # inside the disk's outer radius: many disk hits per ray
    small = B.BlackHole(relativity_sphere_radius=12.0, feather_amount=0.5)                   # default camera outside; exits and re-entries near the rim
    return {
        "default": T.uniforms(integration_method=1),
        "moving": T.uniforms(integration_method=1, camera=side, time=1.7),
        "outside": T.uniforms(integration_method=1, camera=far, time=3.1),
        "near": T.uniforms(integration_method=1, camera=near, time=0.4),
        "small sphere": T.uniforms(integration_method=1, black_hole=small),
        "no textures": T.uniforms(integration_method=1, black_hole=B.BlackHole(show_disk_texture=0, show_red_shift=0)),
        "coarse steps": T.uniforms(integration_method=1, step_size=0.6),
    }


@pytest.mark.parametrize("spec", [0, 2])
def test_pair_march_renders_the_default_kernels_frames(spec):
    tex = T.textures()
    cfg = B.ladder_for_frame((200, 110), 3, 3)
    sc = _scenes()
    want = _compare(cfg, list(sc.values()), tex, f"spec {spec}", speculative_levels=spec)
    steps = [w[2]["steps"] for w in want]
    with library(PAIR):                                                      # ... and it IS the other kernel: a wave-step marches 128 slots
        pair = _render_all(cfg, [sc["default"]], tex, speculative_levels=spec)
    assert pair[0][4] < 0.75 * want[0][4], (pair[0][4], want[0][4])
    assert min(steps) > 0 and want[list(sc).index("outside")][2]["flat_iters"] > 0
    assert want[list(sc).index("near")][2]["disk_hits"] > 1000               # the scenes do reach the rare paths


@pytest.mark.parametrize("max_iterations", [0, 1, 2, 5, 6, 7, 23, 60, 131])
def test_pair_march_rays_that_use_up_their_iterations(max_iterations):
    """The iteration limit ends a ray at the top of a step, after a disk hit, on the way out of the sphere or back in - and `it <= 5`
    decides what the epilogue writes: every small limit, and limits in the middle of the rays' lives."""
    tex = T.textures()
    cfg = B.ladder_from_base((24, 14), 3, 3)
    near = B.Camera(position=(1.0, 0.6, -7.0), forward=(-0.1, -0.05, 1.0), fov=1.3)
    far = B.Camera(position=(0.0, 3.0, -45.0), forward=(0.0, -0.0665190, 0.9977851))
    frames = [T.uniforms(integration_method=1, max_iterations=max_iterations),
              T.uniforms(integration_method=1, max_iterations=max_iterations, camera=near, step_size=0.4),
              T.uniforms(integration_method=1, max_iterations=max_iterations, camera=far, step_size=0.8),
              T.uniforms(integration_method=1, max_iterations=max_iterations, step_size=1.5,
                         black_hole=B.BlackHole(relativity_sphere_radius=12.0))]
    _compare(cfg, frames, tex, f"max_iterations {max_iterations}", speculative_levels=2)


def test_pair_march_in_batches_in_flight_temporal_and_row_partitions():
    tex = T.textures()
    cfg = B.ladder_for_frame((320, 180), 3, 4)
    sc = list(_scenes().values())
    frames = [sc[i % len(sc)] for i in range(9)]
    _compare(cfg, frames, tex, "batches of 3, 2 slots", frames_per_batch=3, frames_in_flight=2, speculative_levels=2)
    _compare(cfg, frames[:5], tex, "temporal", temporal=True, frames_in_flight=2)
    with library(_lib.LIB_PATH):
        want = _render_all(cfg, frames[:3], tex)
    with library(PAIR):
        rp = B.RayPass(cfg, devices=[0, 0, 0], frames_in_flight=1)
        rp.set_textures(*tex)
        for i, u in enumerate(frames[:3]):
            rp.set_uniforms(*u); rp.render()
            assert np.array_equal(rp.read_hdr().view(np.uint32), want[i][1].view(np.uint32)), f"3 partitions: frame {i}"
        rp.close()


def test_pair_march_at_the_metrics_frame():
    """1920x1080 from the 72x41 ladder (BASELINE.json configs[1]), the default scene and the moving one: 2 x 660 k rays through the kernel."""
    tex = T.textures(small=False)
    cfg = B.ladder_for_frame((1920, 1080), 3, 4)
    sc = _scenes()
    _compare(cfg, [sc["default"], sc["moving"], sc["near"]], tex, "1080p", speculative_levels=2, frames_in_flight=2)


def test_the_pair_library_only_replaces_that_kernel():
    """Euler, the mesh variant and the literal evaluation run the default kernels in the pair library too: same frames."""
    tex = T.textures()
    cfg = B.ladder_from_base((24, 14), 3, 3)
    _compare(cfg, [T.uniforms(integration_method=0, step_size=0.2)], tex, "euler")
    _compare(cfg, [T.uniforms(integration_method=1)], tex, "literal", literal=True)
