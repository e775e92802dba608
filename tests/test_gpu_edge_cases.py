"""-m gpu: edge cases and a seeded fuzz sweep of the uniform space the reference's UI exposes
(src/ui/render_settings.rs, black_hole_settings.rs, camera_settings.rs — SURVEY.md §5), HIP vs oracle.

NaN handling is part of the contract (compare-select clamp/min/max): where the shader's own arithmetic
produces NaN (e.g. pow of a negative disk density when the hole is off-origin, ray.wgsl:619-623; feather 0 -> 0/0,
ray.wgsl:548) both sides must produce NaN in the same pixels."""
import numpy as np
import pytest

import bhusie_amd as B
from oracle import oracle as O
from tests import common as T

pytestmark = pytest.mark.gpu


def gpu_frame(cfg, u, tex, **kw):
    rp = B.RayPass(cfg, device=0, **kw)
    rp.set_textures(*tex)
    rp.set_uniforms(*u)
    rp.render()
    return rp


def check(got, want, what):
    assert got.shape == want.shape
    gn, wn = np.isnan(got), np.isnan(want)
    assert np.array_equal(gn, wn), f"{what}: NaN pixels differ ({int(gn.sum())} vs {int(wn.sum())})"
    ok = ~wn.any(axis=-1)
    gi, wi = np.isinf(got[ok]), np.isinf(want[ok])
    assert np.array_equal(gi, wi) and np.array_equal(got[ok][gi], want[ok][wi]), f"{what}: infinities differ"
    fin = ok & np.isfinite(want).all(axis=-1)
    assert np.array_equal(got[..., 3][fin], want[..., 3][fin]), f"{what}: classes differ"
    e = T.rel_err(got[fin], want[fin])
    assert float(e.max(initial=0.0)) <= T.REL_TOL, f"{what}: max rel err {float(e.max()):.3g}"
    d = fin & (want[..., 3] == 0)
    assert np.array_equal(got[d], want[d]), f"{what}: direction pixels not bit-identical"


def test_tiny_and_ragged_frames():
    tex = T.textures()
    for base, levels, method in (((2, 2), 1, 0), ((2, 3), 2, 1), ((5, 2), 3, 1), ((9, 7), 2, 0), ((3, 17), 2, 1)):
        u = T.uniforms(integration_method=method)
        cfg = B.ladder_from_base(base, 3, levels)
        rp = gpu_frame(cfg, u, tex)
        want = O.render_ladder(T.oracle_scene(*u, tex), cfg.sizes())
        check(rp.read_hdr(), want[-1], f"base {base} x{levels}")


def test_single_pixel_window_and_empty_partitions():
    tex = T.textures()
    u = T.uniforms(integration_method=1)
    cfg = B.ladder_from_base((8, 6), 3, 3)                       # 8x6 -> 22x16 -> 64x46
    want = O.render_ladder(T.oracle_scene(*u, tex), cfg.sizes())[-1]
    for (cx, cy, fw, fh) in ((0, 0, 1, 1), (63, 45, 1, 1), (10, 7, 5, 1), (31, 0, 1, 46)):
        c = B.BhrayConfig.from_buffer_copy(bytes(cfg))
        c.crop_x, c.crop_y, c.frame_w, c.frame_h = cx, cy, fw, fh
        got = gpu_frame(c, u, tex).read_hdr()
        check(got, want[cy:cy + fh, cx:cx + fw], f"window {cx},{cy} {fw}x{fh}")
    # more ranks than stripes: some ranks own no rows and must render nothing without error
    rows_seen = []
    for rank in range(4):
        rp = gpu_frame(cfg, u, tex, row_rank=rank, row_world=4, stripe_rows=27)
        rows = rp.local_rows()
        rows_seen += rows.tolist()
        out = rp.read_hdr()
        assert out.shape == (len(rows), 64, 4)
        if len(rows):
            check(out, want[rows], f"rank {rank}")
    assert sorted(rows_seen) == list(range(46))


def test_iteration_limit_classes():
    """`hit || i <= 5` (ray.wgsl:583): tiny iteration limits turn escaping rays into colour pixels."""
    tex = T.textures()
    cfg = B.ladder_from_base((16, 9), 3, 2)
    for mi in (0, 1, 5, 6, 7, 40):
        for method in (0, 1):
            u = T.uniforms(integration_method=method, max_iterations=mi)
            want = O.render_ladder(T.oracle_scene(*u, tex), cfg.sizes())
            check(gpu_frame(cfg, u, tex).read_hdr(), want[-1], f"max_iterations {mi} method {method}")
            if mi <= 5:
                assert np.all(want[-1][..., 3] == 1.0)


def test_degenerate_uniforms_propagate_nan_identically():
    tex = T.textures()
    cfg = B.ladder_from_base((24, 14), 3, 2)
    cases = {
        "hole off origin (negative density -> pow NaN, ray.wgsl:619-623)": dict(bh=B.BlackHole(position=(4.0, -2.0, 1.0))),
        "feather 0 (0/0 in the exit blend, ray.wgsl:546-548)": dict(bh=B.BlackHole(feather_amount=0.0)),
        "inner > outer disk": dict(bh=B.BlackHole(accretion_disk_inner=8.0, accretion_disk_outer=4.0)),
        "sphere smaller than the disk": dict(bh=B.BlackHole(relativity_sphere_radius=6.0), cam=B.Camera(position=(0.0, 1.0, -5.0))),
        "camera on the horizon side": dict(cam=B.Camera(position=(0.0, 0.0, -1.5))),
        "zero forward vector": dict(cam=B.Camera(forward=(0.0, 0.0, 0.0))),
        "forward parallel to plane_up": dict(cam=B.Camera(forward=(0.0, -1.0, 0.0))),
        "large time (range reduction of sin/cos)": dict(time=7321.25),
        "step size limits of the UI": dict(step_size=1.0),
        "fine steps": dict(step_size=0.005, max_iterations=300),
    }
    for name, kw in cases.items():
        for method in (0, 1):
            det = {k: v for k, v in kw.items() if k not in ("bh", "cam")}
            u = T.uniforms(camera=kw.get("cam"), black_hole=kw.get("bh"), integration_method=method, **det)
            want = O.render_ladder(T.oracle_scene(*u, tex), cfg.sizes())
            check(gpu_frame(cfg, u, tex).read_hdr(), want[-1], f"{name} / method {method}")


@pytest.mark.parametrize("offset", [(1000.0, -700.0, 2500.0), (-9000.0, 4000.0, 7000.0)])
def test_hole_and_camera_far_from_the_origin(offset):
    """Hole (and camera) 1e3..1e4 units from the origin: the conservative culls of the horizon / disk tests must stay
    conservative when |bh.position| is large (their quantities come from hole-relative vectors), so classes, NaN positions
    (negative density -> pow NaN, ray.wgsl:619-623) and direction bits still equal the oracle's, which evaluates every test."""
    tex = T.textures()
    cfg = B.ladder_from_base((40, 24), 3, 2)
    off = np.array(offset, dtype=np.float32)
    for method in (0, 1):
        for rel in ((0.0, 0.0, -19.0), (0.0, 3.0, -45.0)):
            cam = B.Camera(position=tuple(float(v) for v in (off + np.array(rel, dtype=np.float32))),
                           forward=tuple(float(v) for v in (-np.array(rel) / np.linalg.norm(rel))))
            u = T.uniforms(camera=cam, black_hole=B.BlackHole(position=tuple(float(v) for v in off)), integration_method=method)
            want = O.render_ladder(T.oracle_scene(*u, tex), cfg.sizes())
            check(gpu_frame(cfg, u, tex, counters=True).read_hdr(), want[-1], f"offset {offset} camera {rel} method {method}")


def test_one_texel_textures_and_empty_model():
    tex = tuple(np.full((1, 1, 4), v, dtype=np.uint8) for v in (200, 128, 64))
    u = T.uniforms(integration_method=1, model_count=1)
    cfg = B.ladder_from_base((16, 9), 3, 2)
    rp = B.RayPass(cfg, device=0)
    rp.set_textures(*tex)
    m = B.Model(); m.build_bvh()                                  # no triangles
    rp.upload_model(m)
    rp.set_uniforms(*u)
    rp.render()
    want = O.render_ladder(T.oracle_scene(*T.uniforms(integration_method=1, model_count=0), tex), cfg.sizes())
    check(rp.read_hdr(), want[-1], "1x1 textures, empty model")


def test_fuzz_uniform_space():
    """40 seeded random points of the UI's parameter space (ranges from src/ui/*_settings.rs)."""
    tex = T.textures()
    rng = np.random.default_rng(20260928)
    cfg = B.ladder_from_base((20, 12), 3, 2)
    for k in range(40):
        pos = rng.normal(size=3) * np.array([6.0, 4.0, 6.0]) + np.array([0.0, 0.0, -16.0])
        fwd = -pos + rng.normal(size=3) * 4.0
        fwd = fwd / np.linalg.norm(fwd)
        cam = B.Camera(position=tuple(pos), forward=tuple(fwd), fov=float(rng.uniform(0.3, 2.2)))
        inner = float(rng.uniform(1.2, 4.0))
        bh = B.BlackHole(accretion_disk_rotation=tuple(rng.uniform(-1.5, 1.5, size=3)), accretion_disk_inner=inner,
                         accretion_disk_outer=inner + float(rng.uniform(1.0, 12.0)), rotation_speed=float(rng.uniform(0, 10)),
                         relativity_sphere_radius=float(rng.uniform(8.0, 40.0)), show_disk_texture=int(rng.integers(0, 2)),
                         show_red_shift=int(rng.integers(0, 2)), feather_amount=float(rng.uniform(0.05, 1.0)))
        method = int(rng.integers(0, 2))
        u = T.uniforms(camera=cam, black_hole=bh, integration_method=method, step_size=float(rng.uniform(0.05, 0.6)),
                       max_iterations=int(rng.integers(50, 900)), angle_division_threshold=float(rng.uniform(0.0, 0.2)),
                       time=float(rng.uniform(0, 100)))
        want = O.render_ladder(T.oracle_scene(*u, tex), cfg.sizes())
        rp = gpu_frame(cfg, u, tex, speculative_levels=0)
        check(rp.read_hdr(), want[-1], f"fuzz case {k} (method {method})")


def test_fuzz_mesh_scenes(tmp_path):
    """12 seeded scenes with a mesh at random places - beside the hole, inside the relativity sphere (where the reference never
    tests it), across the disk, behind the camera, invisible - both integrators, both BVH builders; counters
    (node visits, triangle tests) must equal the oracle's too: the traversal order is part of the path."""
    from bhusie_amd import assets
    tex = T.textures()
    rng = np.random.default_rng(7)
    p = tmp_path / "m.obj"; p.write_text(assets.sphere_mesh_obj(10, 14, radius=6.0, bump=0.3, seed=11))
    cfg = B.ladder_from_base((24, 14), 3, 2)
    tested = visited = 0
    for k in range(12):
        model = B.load_model(str(p))
        if k % 4 == 3:
            model.build_bvh_sah()
        mpos = tuple(rng.normal(size=3) * np.array([14.0, 6.0, 14.0]))
        visible = 0 if k == 5 else 1
        pos = rng.normal(size=3) * np.array([8.0, 4.0, 8.0]) + np.array([0.0, 0.0, -24.0])
        fwd = (np.array(mpos) * rng.uniform(0.0, 1.0) - pos); fwd = fwd / np.linalg.norm(fwd)
        cam = B.Camera(position=tuple(pos), forward=tuple(fwd), fov=float(rng.uniform(0.6, 1.8)))
        bh = B.BlackHole(relativity_sphere_radius=float(rng.uniform(10.0, 30.0)), accretion_disk_rotation=tuple(rng.uniform(-1, 1, size=3)))
        method = k % 2
        u = T.uniforms(camera=cam, black_hole=bh, integration_method=method, model_count=1, step_size=float(rng.uniform(0.1, 0.4)),
                       max_iterations=int(rng.integers(200, 900)))
        model.set_transform(mpos, visible)
        arrays = model.arrays()
        cnt = O.Counters()
        want = O.render_ladder(T.oracle_scene(*u, tex, [arrays]), cfg.sizes(), cnt)
        rp = B.RayPass(cfg, device=0, counters=True, speculative_levels=0)
        rp.set_textures(*tex); rp.upload_model(model); rp.set_uniforms(*u); rp.render()
        check(rp.read_hdr(), want[-1], f"mesh fuzz case {k} (method {method}, mesh at {np.round(mpos, 1)})")
        assert rp.counters() == cnt.as_dict(), f"mesh fuzz case {k}: counters"
        rp.close()
        tested += cnt.triangles; visited += cnt.node_pairs
    assert tested > 1000 and visited > 10000            # the sweep does exercise the traversal


@pytest.mark.parametrize("thr", [0.0, -1.0, 1e-9, 0.02, 0.5235988, 1.5707964, 3.1415925, 3.1415927, 4.0, float("nan"), float("inf")])
def test_angle_threshold_values(thr):
    """The grid classification compares c = cos(angle) with a host-computed c* instead of acos(c) with the threshold
    (bh_acos is monotone: bhray_selftest).  Every threshold - none / all pixels interpolated, NaN, the branch points of the
    portable acos - must classify exactly as the oracle's literal acos(c) < threshold."""
    tex = T.textures()
    u = T.uniforms(integration_method=1, angle_division_threshold=thr)
    cfg = B.ladder_from_base((24, 14), 3, 3)
    cnt = O.Counters()
    want = O.render_ladder(T.oracle_scene(*u, tex), cfg.sizes(), cnt)
    rp = B.RayPass(cfg, device=0, counters=True); rp.set_textures(*tex); rp.set_uniforms(*u); rp.render()
    check(rp.read_hdr(), want[-1], f"threshold {thr}")
    assert rp.counters() == cnt.as_dict()


def test_fuzz_every_build_and_mode_delivers_the_same_frame(monkeypatch):
    """Seeded random scenes (camera, hole, disk, step size, iteration limit, threshold, time, frame size, 1-4 levels, both integrators):
    the latency build (lean step form, short queues dealt out), the dense build, the speculative ladder on four slots and the temporal
    mode (third frame of a static sequence) deliver the same bits — except the SIGN of a NaN, which neither WGSL nor the contract
    specifies (the builds select different instructions for the same operation: 0x7fc00000 / 0xffc00000 in the same pixels)."""
    rng = np.random.default_rng(1234)
    tex = T.textures()
    for trial in range(14):
        method = int(rng.integers(0, 2))
        pos = rng.normal(size=3) * np.array([12.0, 4.0, 12.0]); pos[2] -= 18.0
        fwd = -pos / np.linalg.norm(pos) + rng.normal(size=3) * 0.15; fwd /= np.linalg.norm(fwd)
        cam = B.Camera(position=tuple(float(v) for v in pos), forward=tuple(float(v) for v in fwd), fov=float(rng.uniform(0.6, 1.6)))
        bh = B.BlackHole(position=tuple(float(v) for v in rng.normal(size=3) * 2.0), relativity_sphere_radius=float(rng.uniform(8.0, 30.0)),
                         accretion_disk_inner=float(rng.uniform(1.5, 4.0)), accretion_disk_outer=float(rng.uniform(5.0, 12.0)))
        u = T.uniforms(integration_method=method, camera=cam, black_hole=bh, step_size=float(rng.uniform(0.05, 0.3)), max_iterations=int(rng.integers(50, 2500)),
                       angle_division_threshold=float(rng.uniform(0.005, 0.08)), time=float(rng.uniform(0, 5)))
        cfg = B.ladder_for_frame((int(rng.integers(60, 420)), int(rng.integers(40, 260))), 3, int(rng.integers(1, 5)))
        nl = len(cfg.sizes())
        frames = {}
        for name, env, kw, renders in (("latency", "0", dict(frames_in_flight=1), 1), ("dense", "1", dict(frames_in_flight=1), 1),
                                       ("speculative, 4 slots", None, dict(frames_in_flight=4, speculative_levels=2 if nl >= 3 else 0), 1),
                                       ("temporal", None, dict(frames_in_flight=1, temporal=True), 3)):
            monkeypatch.delenv("BHRAY_TRACE_DENSE", raising=False)
            if env is not None:
                monkeypatch.setenv("BHRAY_TRACE_DENSE", env)
            rp = B.RayPass(cfg, **kw)
            rp.set_textures(*tex); rp.set_uniforms(*u)
            for _ in range(renders):
                rp.render()
            frames[name] = rp.read_hdr()
            rp.close()
        ref = frames["latency"]
        for name, f in frames.items():
            both_nan = np.isnan(f) & np.isnan(ref)
            a, b = np.where(both_nan, 0.0, f).astype(np.float32), np.where(both_nan, 0.0, ref).astype(np.float32)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"trial {trial}: {name} differs from the latency build ({cfg.sizes()}, method {method})"


@pytest.mark.parametrize("grid", [1, 3, 8, 40])
def test_a_batch_on_a_small_persistent_grid_traces_every_frame(monkeypatch, grid):
    """The short queues of a batch's coarse launches are dealt out over the blocks whose own frame it is - which needs blocks for every
    frame of the batch (bhray_kernels.hip: gridDim.x >= 4 * nb).  A grid smaller than that (BHRAY_TRACE_GRID: tuning runs, tiny devices)
    must fall back to pulling from the queue head: every frame of the batch complete, byte for byte the frame of a ctx without batches."""
    tex = T.textures()
    cfg = B.ladder_for_frame((160, 90), 3, 3)
    frames = [T.uniforms(integration_method=k & 1, time=0.3 * k) for k in range(6)]
    want = []
    monkeypatch.delenv("BHRAY_TRACE_GRID", raising=False)
    monkeypatch.setenv("BHRAY_TRACE_DENSE", "0")                 # the latency build is the one that deals short queues out
    for u in frames:
        rp = gpu_frame(cfg, u, tex, frames_in_flight=1)
        want.append(rp.read_hdr()); rp.close()
    monkeypatch.setenv("BHRAY_TRACE_GRID", str(grid))            # persistent trace blocks per launch: fewer than 4 x frames_per_batch for 1, 3, 8
    for spec in (0, 2):
        rp = B.RayPass(cfg, device=0, frames_per_batch=3, frames_in_flight=2, speculative_levels=spec)
        rp.set_textures(*tex)
        for k, u in enumerate(frames):                          # batches of 3: two integrators alternate, so batches are also cut short by the variant change
            rp.set_uniforms(*u); rp.render()
            if k in (1, 2, 5):
                assert np.array_equal(rp.read_hdr().view(np.uint32), want[k].view(np.uint32)), (grid, spec, k)
        rp.close()
    same = [T.uniforms(integration_method=1, time=0.1 * k) for k in range(5)]
    ref = []
    monkeypatch.delenv("BHRAY_TRACE_GRID", raising=False)
    for u in same:
        rp = gpu_frame(cfg, u, tex, frames_in_flight=1); ref.append(rp.read_hdr()); rp.close()
    monkeypatch.setenv("BHRAY_TRACE_GRID", str(grid))
    rp = B.RayPass(cfg, device=0, frames_per_batch=5, frames_in_flight=1)       # ONE batch of five frames of one variant: nb = 5
    rp.set_textures(*tex)
    bufs = [T.DeviceBuffer(160 * 90 * 16) for _ in same]
    for u, b in zip(same, bufs):
        rp.set_uniforms(*u); rp.bind_output(b.ptr.value, b.nbytes); rp.render()
    rp.sync()
    for k, b in enumerate(bufs):
        assert np.array_equal(b.read(np.uint32), ref[k].view(np.uint32).ravel()), (grid, k)
        b.free()
    rp.close()
