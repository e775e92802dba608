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


def _mesh_model(tmp_path, n_lat=20, n_lon=28, with_normals=True):
    from bhusie_amd import assets
    p = tmp_path / "mesh.obj"
    p.write_text(assets.sphere_mesh_obj(n_lat, n_lon, radius=8.0, bump=0.2, seed=5, with_normals=with_normals))
    return B.load_model(str(p))


@pytest.mark.parametrize("with_normals", [True, False])
def test_mesh_parity(tmp_path, with_normals):
    """configs[2]: BVH mesh in the flat-space phases (ray.wgsl:556), default model position (-10,0,30)."""
    tex = T.textures()
    model = _mesh_model(tmp_path, with_normals=with_normals)
    # camera outside the sphere looking at both the hole and the mesh
    d = np.array([-0.12, 0.0, 1.0]); d /= np.linalg.norm(d)
    cam = B.Camera(position=(0.0, 0.0, -40.0), forward=tuple(d), fov=1.2)
    for method in (0, 1):
        u = T.uniforms(camera=cam, integration_method=method, model_count=1)
        cfg = B.ladder_from_base((40, 24), 3, 2)
        rp = run_gpu(cfg, *u, tex, model=model, counters=True)
        cnt = O.Counters()
        want = O.render_ladder(T.oracle_scene(*u, tex, [model.arrays()]), cfg.sizes(), cnt)
        T.assert_parity(rp.read_hdr(), want[-1], f"mesh method {method}")
        c = rp.counters()
        assert c == cnt.as_dict()
        assert c["triangles"] > 0 and c["node_pairs"] > 0


def test_mesh_behind_hole_is_lensed(tmp_path):
    """Camera inside the sphere, mesh beyond it: rays leave the sphere bent and then hit the mesh."""
    tex = T.textures()
    model = _mesh_model(tmp_path)
    model.set_transform((-6.0, 0.0, 32.0), 1)
    u = T.uniforms(integration_method=1, model_count=1)
    cfg = B.ladder_from_base((48, 27), 3, 2)
    rp = run_gpu(cfg, *u, tex, model=model, counters=True)
    want = O.render_ladder(T.oracle_scene(*u, tex, [model.arrays()]), cfg.sizes())
    T.assert_parity(rp.read_hdr(), want[-1], "mesh behind hole")
    assert rp.counters()["triangles"] > 0


def test_model_uniform_blob_equals_compact_upload(tmp_path):
    """The exact 48 234 572-byte ModelUniform (triangle.rs:268-325) and the compact upload give the same frame;
    set_model_transform replaces the per-frame 48 MB re-upload (mod.rs:391)."""
    tex = T.textures()
    model = _mesh_model(tmp_path, 10, 14)
    cam = B.Camera(position=(0.0, 0.0, -40.0), forward=(-0.12, 0.0, 0.99), fov=1.2)
    u = T.uniforms(camera=cam, integration_method=0, model_count=1)
    cfg = B.ladder_from_base((40, 24), 3, 1)
    a = run_gpu(cfg, *u, tex, model=model).read_hdr()
    rp = B.RayPass(cfg, device=0)
    rp.set_textures(*tex)
    rp.upload_model_uniform(model.pack_uniform())
    rp.set_uniforms(*u)
    rp.render()
    b = rp.read_hdr()
    assert np.array_equal(a, b)
    rp.set_model_transform((0.0, 0.0, 60.0), 0)       # invisible: no triangle may be hit
    rp.render()
    c = rp.read_hdr()
    want = O.render_level(T.oracle_scene(*T.uniforms(camera=cam, integration_method=0, model_count=0), tex), (40, 24))
    T.assert_parity(c, want, "invisible model")


def test_reference_native_ladder_1918x1081():
    """The reference's shipped configuration (mod.rs:177-179): 72x41 x3 x4 -> 1918x1081, adaptive RK."""
    tex = T.textures()
    u = T.uniforms(integration_method=1)
    cfg = B.ladder_from_base((72, 41), 3, 4)
    assert cfg.sizes()[-1] == (1918, 1081)
    rp = run_gpu(cfg, *u, tex, counters=True)
    cnt = O.Counters()
    want = O.render_ladder(T.oracle_scene(*u, tex), cfg.sizes(), cnt)
    mx, exact = T.assert_parity(rp.read_hdr(), want[-1], "1918x1081")
    assert rp.counters() == cnt.as_dict()
    print(f"1918x1081: max rel {mx:.3g}, bit-exact {exact:.5f}, counters {cnt.as_dict()}")


@pytest.mark.parametrize("name", ["euler_l0", "rk_l0", "rk_ladder", "euler_ladder", "rk_outside", "euler_flags_off"])
def test_gpu_against_committed_golden_frames(name):
    """HIP path vs tests/golden/frames.npz (written by the independent NumPy restatement)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frames.npz"))
    sizes = [tuple(int(v) for v in s) for s in g[f"{name}.sizes"]]
    cfg = B.ladder_from_base(sizes[0], 3, len(sizes))
    assert cfg.sizes() == sizes
    rp = run_gpu(cfg, g[f"{name}.camera"].tobytes(), g[f"{name}.black_hole"].tobytes(), g[f"{name}.details"].tobytes(),
                 (g["t_temp"], g["t_disk"], g["t_sky"]), counters=True)
    for l in range(len(sizes)):
        want = g[f"{name}.level{l}"]
        got = rp.read_level(l)
        T.assert_parity(got, want, f"{name} level {l}")
        d = want[..., 3] == 0
        assert np.array_equal(got[d], want[d]), "direction pixels must be bit-identical to the golden frame"
    traced, steps, copied, interp, sky = (int(v) for v in g[f"{name}.stats"])
    c = rp.counters()
    assert (c["traced"], c["steps"], c["copied"], c["interpolated"], c["sky_samples"]) == (traced, steps, copied, interp, sky)


def test_gpu_against_committed_golden_mesh(tmp_path):
    import os
    gd = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g, f = np.load(os.path.join(gd, "mesh.npz")), np.load(os.path.join(gd, "frames.npz"))
    p = tmp_path / "m.obj"; p.write_bytes(g["obj"].tobytes())
    model = B.load_model(str(p))
    for method in (0, 1):
        rp = run_gpu(B.ladder_from_base((40, 24), 3, 1), g["camera"].tobytes(), g["black_hole"].tobytes(),
                     g[f"details{method}"].tobytes(), (f["t_temp"], f["t_disk"], f["t_sky"]), model=model)
        T.assert_parity(rp.read_hdr(), g[f"frame{method}"], f"golden mesh frame {method}")


def test_1920x1080_bench_config_matches_oracle():
    """configs[1] at full size: the 1920x1080 window of the 73x41 x3 x4 ladder; the oracle renders the whole
    ladder once (a few seconds on 8 cores) and every pixel is compared."""
    tex = T.textures()
    u = T.uniforms(integration_method=1)
    cfg = B.ladder_for_frame((1920, 1080), 3, 4)
    rp = run_gpu(cfg, *u, tex)
    got = rp.read_hdr()
    assert got.shape == (1080, 1920, 4)
    cx, cy = int(cfg.crop_x), int(cfg.crop_y)
    want = O.render_ladder(T.oracle_scene(*u, tex), cfg.sizes())[-1][cy:cy + 1080, cx:cx + 1920]
    T.assert_parity(got, want, "1920x1080 window")
    # idempotence: rendering again gives the same bytes
    rp.render()
    assert np.array_equal(rp.read_hdr(), got)


def test_sky_resolve_pass_bit_exact():
    """sky.wgsl on the GPU (bhray_resolve_sky) vs the oracle: rgba16float, bit for bit; also against the golden image."""
    import os
    tex = T.textures()
    u = T.uniforms(integration_method=1)
    cfg = B.ladder_from_base((24, 14), 3, 3)
    rp = run_gpu(cfg, *u, tex)
    rp.resolve_sky()
    got = rp.read_sky()
    hdr = rp.read_hdr()
    want = O.sky_resolve(hdr, tex[2])
    assert got.dtype == np.float16 and got.shape == want.shape
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))
    assert np.all(got[..., 3] == 1.0)
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frames.npz"))
    rp = run_gpu(B.ladder_from_base((24, 14), 3, 3), g["rk_ladder.camera"].tobytes(), g["rk_ladder.black_hole"].tobytes(),
                 g["rk_ladder.details"].tobytes(), (g["t_temp"], g["t_disk"], g["t_sky"]))
    rp.resolve_sky()
    sky = rp.read_sky().view(np.uint16)
    ref = g["rk_ladder.sky"]
    # colour pixels differ from the NumPy golden only through pow(.,1.3) (1-2 ulp in f32, usually the same binary16)
    same = (sky == ref).all(axis=-1)
    assert same.mean() > 0.995
    direction = g["rk_ladder.level2"][..., 3] == 0
    assert np.array_equal(sky[direction], ref[direction])
    # row partition: the resolve pass works on the packed rows of a rank
    rp2 = run_gpu(cfg, *u, tex, row_rank=1, row_world=2, stripe_rows=9)
    rp2.resolve_sky()
    assert np.array_equal(rp2.read_sky().view(np.uint16), want.view(np.uint16)[rp2.local_rows()])


@pytest.mark.parametrize("method,spec", [(1, 2), (1, 3), (0, 3)])
def test_speculative_levels_give_identical_frames(method, spec):
    """bhray_config.speculative_levels: levels 0..S-1 traced in one launch, then classified — every level image and the
    frame must equal the normal ladder's bit for bit (and hence the oracle's within the usual bar)."""
    tex = T.textures()
    u = T.uniforms(integration_method=method)
    cfg = B.ladder_from_base((24, 14), 3, 4)                # 24x14 -> 70x40 -> 208x118 -> 622x352
    normal = run_gpu(cfg, *u, tex)
    fast = run_gpu(cfg, *u, tex, speculative_levels=spec, counters=True)
    for l in range(4):
        assert np.array_equal(normal.read_level(l), fast.read_level(l), equal_nan=True), f"level {l}"
    assert np.array_equal(normal.read_hdr(), fast.read_hdr())
    want = O.render_ladder(T.oracle_scene(*u, tex), cfg.sizes())
    T.assert_parity(fast.read_hdr(), want[-1], "speculative frame")
    c = fast.counters()
    assert c["traced"] >= sum(w * h for w, h in cfg.sizes()[:spec])          # all pixels of the speculated levels were traced


def test_speculative_levels_with_crop_rows_and_mesh(tmp_path):
    tex = T.textures()
    model = _mesh_model(tmp_path, 12, 16)
    cam = B.Camera(position=(0.0, 0.0, -40.0), forward=(-0.11914522, 0.0, 0.99287683), fov=1.2)
    u = T.uniforms(camera=cam, integration_method=1, model_count=1)
    cfg = B.ladder_for_frame((200, 110), 3, 3)
    full = run_gpu(cfg, *u, tex, model=model).read_hdr()
    for rank in range(2):
        rp = run_gpu(cfg, *u, tex, model=model, row_rank=rank, row_world=2, stripe_rows=9, speculative_levels=2, frames_in_flight=2)
        rp.render(); rp.render()                              # slots are reused: still the same bytes
        assert np.array_equal(rp.read_hdr(), full[rp.local_rows()])
    with pytest.raises(B.BhrayError):
        B.RayPass(cfg, device=0, speculative_levels=3)         # must leave at least the last level to the normal path


def test_cpp_host_program_matches_python_host(tmp_path):
    """bhusie_amd/bhray_render (C++ over host/renderer.hpp: Renderer / RayPipeline / Model mirrors) renders the same bytes as
    the Python host, and as the oracle within the bar."""
    import os
    import subprocess
    from bhusie_amd import assets
    exe = os.path.join(os.path.dirname(B.LIB_PATH), "bhray_render")
    obj = tmp_path / "m.obj"
    obj.write_text(assets.icosphere_mesh_obj(3, radius=8.0, bump=0.1, seed=4))
    out = tmp_path / "o.f32"
    r = subprocess.run([exe, str(out), "--rk", "--base", "24", "14", "--levels", "2", "--disk-size", "64", "--obj", str(obj)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip() == "70x40"
    got = np.fromfile(out, dtype=np.float32).reshape(40, 70, 4)
    grey = np.full((1, 1, 4), 160, dtype=np.uint8); grey[..., 3] = 255
    tex = (grey, assets.reference_disk_texture(64), grey)
    model = B.load_model(str(obj))
    u = T.uniforms(integration_method=1, model_count=1)
    rp = run_gpu(B.ladder_from_base((24, 14), 3, 2), *u, tex, model=model)
    assert np.array_equal(got, rp.read_hdr())
    want = O.render_ladder(T.oracle_scene(*u, tex, [model.arrays()]), [(24, 14), (70, 40)])
    T.assert_parity(got, want[-1], "C++ host program")


def test_config1_256x256_euler_single_level():
    """BASELINE.json configs[0]: 256x256 single frame, Euler integrator, accretion disk only, one level (every pixel traced).
    The reference's "CPU-runnable plumbing case": here the CPU side is the oracle, the HIP path must match it."""
    tex = T.textures()
    u = T.uniforms(integration_method=0, model_count=0)
    cfg = B.ladder_from_base((256, 256), 3, 1)
    rp = run_gpu(cfg, *u, tex, counters=True)
    cnt = O.Counters()
    want = O.render_level(T.oracle_scene(*u, tex), (256, 256), None, cnt)
    mx, exact = T.assert_parity(rp.read_hdr(), want, "config 1")
    d = want[..., 3] == 0
    assert np.array_equal(rp.read_hdr()[d], want[d])
    assert rp.counters() == cnt.as_dict() and cnt.traced == 256 * 256 and cnt.copied == 0 and cnt.interpolated == 0


def test_sah_tree_gives_the_same_closest_hits(tmp_path):
    """The flagged SAH builder changes the tree, not the geometry: the HIP path with the SAH tree matches the oracle run on the
    SAME tree (traversal semantics), and matches the reference-tree frame except where two triangles are hit at equal t."""
    tex = T.textures()
    model = _mesh_model(tmp_path, 24, 32)
    d = np.array([-0.12, 0.0, 1.0]); d /= np.linalg.norm(d)
    cam = B.Camera(position=(0.0, 0.0, -40.0), forward=tuple(d), fov=1.2)
    u = T.uniforms(camera=cam, integration_method=1, model_count=1)
    cfg = B.ladder_from_base((40, 24), 3, 2)
    ref_frame = run_gpu(cfg, *u, tex, model=model).read_hdr()
    model.build_bvh_sah()
    rp = run_gpu(cfg, *u, tex, model=model, counters=True)
    got = rp.read_hdr()
    cnt = O.Counters()
    want = O.render_ladder(T.oracle_scene(*u, tex, [model.arrays()]), cfg.sizes(), cnt)
    T.assert_parity(got, want[-1], "SAH tree vs oracle on the same tree")
    assert rp.counters() == cnt.as_dict()
    same = (got == ref_frame).all(axis=-1)
    assert same.mean() > 0.995, same.mean()


def _frames_for_batch():
    """Three frames with different uniforms (camera, disk rotation time, flags)."""
    cams = [B.Camera(), B.Camera(position=(2.0, 1.5, -24.0), forward=(-0.0830455, -0.0622841, 0.9945987)),
            B.Camera(position=(0.0, 3.0, -45.0), forward=(0.0, -0.0665190, 0.9977851))]
    return [T.uniforms(camera=cams[0], integration_method=1, time=0.0),
            T.uniforms(camera=cams[1], integration_method=1, time=1.7),
            T.uniforms(camera=cams[2], integration_method=1, time=3.1)]


@pytest.mark.parametrize("spec", [0, 2])
def test_frame_batches_give_identical_frames(spec):
    """bhray_config.frames_per_batch: bhray_render stages, every launch covers the staged frames (different uniforms per
    frame).  Each frame must equal the frame a batch-less ctx renders from the same uniforms, bit for bit."""
    tex = T.textures()
    frames = _frames_for_batch()
    cfg = B.ladder_from_base((24, 14), 3, 3)
    want = [run_gpu(cfg, *u, tex, speculative_levels=spec).read_hdr() for u in frames]
    assert not np.array_equal(want[0], want[1]) and not np.array_equal(want[1], want[2])
    rp = B.RayPass(cfg, device=0, frames_per_batch=3, frames_in_flight=2, speculative_levels=spec, counters=True)
    rp.set_textures(*tex)
    h, w = want[0].shape[:2]
    fb = h * w * 16
    out = T.DeviceBuffer(7 * fb)                               # NaN-filled
    order = [0, 1, 2, 2, 0, 1, 1]                              # 2 full batches + a partial one; slots are reused
    for i, f in enumerate(order):
        rp.set_uniforms(*frames[f])
        rp.bind_output(out.ptr.value + i * fb, fb)
        rp.render()
    rp.sync()                                                  # flushes the partial batch
    got = out.read().reshape(7, h, w, 4)
    for i, f in enumerate(order):
        assert np.array_equal(got[i], want[f]), f"frame {i} (uniform set {f})"
    assert np.array_equal(rp.read_hdr(), want[order[-1]])      # the most recent frame, through the bound buffer
    single = run_gpu(cfg, *frames[order[-1]], tex, speculative_levels=spec, counters=True)
    assert rp.counters() == single.counters()
    rp.close(); out.free()


def test_frame_batch_switches_kernel_variant_and_partition(tmp_path):
    """A batch is homogeneous in its kernel variant: a frame that needs another integrator (or the mesh variant) first
    launches what is staged.  Also: batches on a row partition with a mesh."""
    tex = T.textures()
    model = _mesh_model(tmp_path, 12, 16)
    cam = B.Camera(position=(0.0, 0.0, -40.0), forward=(-0.11914522, 0.0, 0.99287683), fov=1.2)
    cfg = B.ladder_for_frame((200, 110), 3, 3)
    us = [T.uniforms(camera=cam, integration_method=1, model_count=1), T.uniforms(camera=cam, integration_method=0, model_count=1),
          T.uniforms(camera=cam, integration_method=0, model_count=0), T.uniforms(camera=cam, integration_method=1, model_count=1)]
    want = [run_gpu(cfg, *u, tex, model=model).read_hdr() for u in us]
    for rank in range(2):
        rp = B.RayPass(cfg, device=0, frames_per_batch=4, frames_in_flight=2, row_rank=rank, row_world=2, stripe_rows=9, speculative_levels=2)
        rp.set_textures(*tex); rp.upload_model(model)
        for i, u in enumerate(us):
            rp.set_uniforms(*u); rp.render()
            assert np.array_equal(rp.read_hdr(), want[i][rp.local_rows()]), f"rank {rank} frame {i}"     # read flushes
        for u in us:                                           # the same four frames staged back to back: three variant switches
            rp.set_uniforms(*u); rp.render()
        assert np.array_equal(rp.read_hdr(), want[3][rp.local_rows()])
    with pytest.raises(B.BhrayError):
        B.RayPass(cfg, device=0, frames_per_batch=33)                # BHRAY_MAX_FRAMES_PER_BATCH = 32


def test_exact_math_selftest():
    """N8: the integrator's short 1/x and sqrt sequences equal the IEEE results on every binary32 input of this device."""
    rp = B.RayPass(B.ladder_from_base((8, 8), 3, 1), device=0)
    assert rp.selftest() == (0, 0, 0)


def test_batch_timing_and_flush_accounting():
    """bhray_timing counts frames and batches separately; bhray_flush launches a partial batch and is a no-op otherwise."""
    tex = T.textures()
    u = T.uniforms(integration_method=1)
    cfg = B.ladder_from_base((24, 14), 3, 2)
    rp = B.RayPass(cfg, device=0, timing=True, frames_per_batch=3, frames_in_flight=2)
    rp.set_textures(*tex); rp.set_uniforms(*u)
    for _ in range(7):
        rp.render()
    rp.flush(); rp.flush()
    t = rp.timing()
    assert (t.frames, t.batches) == (7, 3)
    assert t.trace_launches == 3 * 2 and t.classify_launches == 3 * 2
    want = O.render_ladder(T.oracle_scene(*u, tex), cfg.sizes())[-1]
    T.assert_parity(rp.read_hdr(), want, "last frame of a partial batch")
    t = rp.timing()
    assert (t.frames, t.batches) == (0, 0)


@pytest.mark.parametrize("method", [0, 1])
def test_mirror_symmetry_at_full_size(method):
    """A size-independent property at the bench size, without the oracle: with the camera and the disk normal in the plane
    x = 0, a black sky and the (handed) disk texture / Doppler shading off, the scene is mirror-symmetric in x, and so is
    every IEEE operation of the path (sign symmetry of +, x, fma; squares in the norms).  Every one of the 2 M pixels of a
    fully traced 1919x1079 level must therefore equal its mirror image bit for bit: escape directions with x negated, the
    pixel classes and the disk colours unchanged."""
    tex = list(T.textures())
    tex[2] = np.zeros((1, 1, 4), dtype=np.uint8)                               # black sky: even a uniform one is not enough, a(1-t) + a t != a
    bh = B.BlackHole(accretion_disk_rotation=(0.3, 0.0, 0.0), show_disk_texture=0, show_red_shift=0)
    u = T.uniforms(black_hole=bh, integration_method=method)
    cfg = B.ladder_from_base((1919, 1079), 3, 1)
    img = run_gpu(cfg, *u, tex).read_hdr()
    assert img.shape == (1079, 1919, 4) and np.isfinite(img).all()
    mir = img[:, ::-1].copy()
    direction = img[..., 3] == 0
    assert 0.2 < direction.mean() < 0.98 and (img[..., 3] == 1).any()           # both classes present
    assert np.array_equal(img[..., 3], mir[..., 3])
    mir[..., 0] = np.where(direction, -mir[..., 0], mir[..., 0])
    assert np.array_equal(img.view(np.uint32) & 0x7FFFFFFF, mir.view(np.uint32) & 0x7FFFFFFF)      # up to the sign of zeros
    assert np.array_equal(img, mir)


def test_production_configuration_is_invariant_at_full_size():
    """1920x1080 adaptive RK through the configuration bench.py runs on 8 GPUs (row partition 8 x 27-row stripes,
    speculative levels 2, frame batches, several slots): the eight partitions together must be the whole frame rendered by
    a plain ctx (one slot, no speculation, no batches), byte for byte."""
    tex = T.textures()
    u = T.uniforms(integration_method=1)
    cfg = B.ladder_for_frame((1920, 1080), 3, 4)
    whole = run_gpu(cfg, *u, tex, frames_in_flight=1).read_hdr()
    assert whole.shape == (1080, 1920, 4) and not np.isnan(whole).any()
    seen = np.zeros(1080, dtype=bool)
    for rank in range(8):
        rp = B.RayPass(cfg, device=0, row_rank=rank, row_world=8, stripe_rows=27, speculative_levels=2, frames_per_batch=2, frames_in_flight=3)
        rp.set_textures(*tex); rp.set_uniforms(*u)
        for _ in range(5):
            rp.render()
        rows = rp.local_rows()
        assert np.array_equal(rp.read_hdr(), whole[rows]), f"rank {rank}"
        seen[rows] = True
        rp.close()
    assert seen.all()


@pytest.mark.parametrize("method", [1, 0])
def test_the_mesh_variant_s_two_builds_deliver_the_same_frame(tmp_path, method, monkeypatch):
    """The mesh variant has a build for lone launches (5 waves per SIMD, the traversal inline in the flat phase) and one for a saturated
    device (6 waves: the traversal in a region of its own with the marching state stored around it, while-while form - trace_kernel's
    MESH_DENSE).  Same pixels at every level and the same counters, with the camera outside the sphere (a traversal before the march and one
    after it) and inside it; the build a ctx picks by itself is one of the two."""
    tex = T.textures()
    model = _mesh_model(tmp_path)
    d = np.array([-0.12, 0.0, 1.0]); d /= np.linalg.norm(d)
    for cam, pos in ((B.Camera(position=(0.0, 0.0, -40.0), forward=tuple(d), fov=1.2), None), (B.Camera(), (-6.0, 0.0, 32.0))):
        if pos is not None:
            model.set_transform(pos, 1)
        u = T.uniforms(camera=cam, integration_method=method, model_count=1)
        cfg = B.ladder_for_frame((240, 135), 3, 3)
        got = {}
        for name, env, fif in (("latency", "0", 1), ("dense", "1", 1), ("auto 1 slot", None, 1), ("auto 6 slots", None, 6)):
            monkeypatch.delenv("BHRAY_TRACE_DENSE", raising=False)
            if env is not None:
                monkeypatch.setenv("BHRAY_TRACE_DENSE", env)
            rp = run_gpu(cfg, *u, tex, model=model, frames_in_flight=fif, counters=True)
            got[name] = ([rp.read_level(l) for l in range(3)], rp.counters())
            rp.close()
        ref_levels, ref_counters = got["latency"]
        assert ref_counters["triangles"] > 0 and ref_counters["node_pairs"] > ref_counters["flat_iters"] // 2
        for name, (levels, counters) in got.items():
            for a, b in zip(levels, ref_levels):
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (name, pos)
            assert counters == ref_counters, (name, pos)


@pytest.mark.parametrize("method", [1, 0])
def test_latency_build_and_dense_build_deliver_the_same_frame(method, monkeypatch):
    """One frame slot selects the latency build (lean step form, short queues dealt out one wave per SIMD: bhray_step.inc, trace_kernel),
    four slots the dense build; BHRAY_TRACE_DENSE forces either.  Same pixels at every level, same counters — scheduling and control
    flow differ, the operations of a ray do not."""
    tex = T.textures()
    u = T.uniforms(integration_method=method)
    cfg = B.ladder_for_frame((400, 225), 3, 3)
    got = {}
    for name, env, fif in (("latency", "0", 1), ("dense", "1", 1), ("auto 1 slot", None, 1), ("auto 4 slots", None, 4)):
        monkeypatch.delenv("BHRAY_TRACE_DENSE", raising=False)
        if env is not None:
            monkeypatch.setenv("BHRAY_TRACE_DENSE", env)
        rp = run_gpu(cfg, *u, tex, frames_in_flight=fif, counters=True)
        got[name] = ([rp.read_level(l) for l in range(3)], rp.counters())
        rp.close()
    ref_levels, ref_counters = got["latency"]
    for name, (levels, counters) in got.items():
        for a, b in zip(levels, ref_levels):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), name
        assert counters == ref_counters, name
