"""-m gpu: the multi-GPU row tiling INSIDE libbhray (bhray_config.device_count >= 2): ONE ctx, N row partitions, the RCCL gather
and the HIP de-interleave enqueued by bhray_render — everything through the C ABI, no PyTorch anywhere.

On a one-GPU box the device list repeats device 0: the partitions then share one RCCL rank and their tiles travel as grouped
ncclSend/ncclRecv to self (RCCL refuses a communicator with a duplicated device).  That exercises everything except the xGMI
hop itself: per-partition engines in lock step, output bindings into send buffers / the assembled frame, the group call, the
stream ordering between render, gather and the next batch, staging layout, the de-interleave kernel, frame batches, bound
outputs, the sky pass over the assembled frame.  Every comparison is byte for byte against the frame ONE partition-less ctx
renders from the same uniforms."""
import os
import subprocess

import numpy as np
import pytest

import bhusie_amd as B
from tests import common as T

pytestmark = pytest.mark.gpu


def _single(cfg, u, tex, model=None, **kw):
    rp = B.RayPass(cfg, device=0, **kw)
    rp.set_textures(*tex)
    if model is not None:
        rp.upload_model(model)
    rp.set_uniforms(*u)
    rp.render()
    return rp


def _frames():
    cam2 = B.Camera(position=(1.0, 2.0, -17.0), forward=(0.0, -0.1, 1.0), fov=1.1)
    return [T.uniforms(integration_method=1), T.uniforms(integration_method=1, camera=cam2, time=0.7),
            T.uniforms(integration_method=0, step_size=0.2)]


@pytest.mark.parametrize("nparts,root,stripe", [(2, 0, 27), (3, 1, 9), (8, 0, 5), (5, 4, 27)])
def test_one_ctx_many_partitions_assembles_the_frame(nparts, root, stripe):
    """Partitions with unequal row counts, partitions without rows (5 x 27-row stripes on a 110-row frame), a root that is
    not partition 0: the assembled frame, every ladder level and the sky pass equal the single ctx's."""
    tex = T.textures()
    u = T.uniforms(integration_method=1)
    cfg = B.ladder_for_frame((200, 110), 3, 3)
    one = _single(cfg, u, tex)
    want = one.read_hdr()
    rp = B.RayPass(cfg, devices=[0] * nparts, gather_root=root, stripe_rows=stripe, frames_in_flight=2)
    info = rp.gather_info()
    assert info["partitions"] == nparts and info["local_partitions"] == nparts and info["root"] == root and info["root_is_local"] == 1
    assert info["comm_ranks"] == 1 and info["rccl_version"] >= 20000         # duplicated device: one shared RCCL rank
    rows = B.partition_rows(110, nparts, stripe)
    assert info["bytes_received_per_frame"] == sum(len(r) for k, r in enumerate(rows) if k != root) * (3 * 200 + 7) * 4      # a row travels packed: 3 floats per pixel + ceil(200 / 32) words of alpha bits
    rp.set_textures(*tex)
    rp.set_uniforms(*u)
    for _ in range(3):                                       # slots are reused: the third frame overwrites the first one's buffers
        rp.render()
    got = rp.read_hdr()
    assert got.shape == want.shape and rp.local_rows().tolist() == list(range(110))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    for l in range(2):                                       # coarse levels: every partition computed the rows it needs; the overlay is the level
        a, b = rp.read_level(l).view(np.uint32), one.read_level(l).view(np.uint32)
        have = ~(a == 0xFFFFFFFF).all(axis=-1)
        assert have.any() and np.array_equal(a[have], b[have])
    rp.resolve_sky(); one.resolve_sky()
    assert np.array_equal(rp.read_sky().view(np.uint16), one.read_sky().view(np.uint16))
    rp.close(); one.close()


@pytest.mark.parametrize("spec", [0, 2])
def test_gather_with_frame_batches_bound_outputs_and_variant_switches(spec):
    """frames_per_batch = 3: one gather per batch (partial batches included), different uniforms and integrators per frame
    (a variant switch launches — and gathers — what is staged), caller-bound frame destinations, slot reuse."""
    tex = T.textures()
    frames = _frames()
    cfg = B.ladder_from_base((24, 14), 3, 3)
    want = [_single(cfg, u, tex, speculative_levels=spec).read_hdr() for u in frames]
    h, w = want[0].shape[:2]
    fb = h * w * 16
    rp = B.RayPass(cfg, devices=[0, 0, 0], frames_per_batch=3, frames_in_flight=2, speculative_levels=spec, stripe_rows=7, timing=True)
    rp.set_textures(*tex)
    order = [0, 1, 1, 0, 2, 2, 1, 0, 0, 1]                   # integrator switches at 3->4 and 5->6; partial batches
    out = T.DeviceBuffer(len(order) * fb)                    # NaN-filled
    for i, f in enumerate(order):
        rp.set_uniforms(*frames[f])
        if i % 2 == 0:
            rp.bind_output(out.ptr.value + i * fb, fb)       # one-shot: the odd frames go to the ctx's own buffers
        rp.render()
        if i % 2 == 1:
            assert np.array_equal(rp.read_hdr(), want[f]), f"frame {i} through the ctx's own buffer"
    rp.sync()
    got = out.read().reshape(len(order), h, w, 4)
    for i, f in enumerate(order):
        if i % 2 == 0:
            assert np.array_equal(got[i], want[f]), f"frame {i} (uniform set {f}) through the bound buffer"
        else:
            assert np.isnan(got[i]).all()
    tm = rp.timing()
    assert tm.gathers >= 4 and tm.gather_ms > 0 and tm.deinterleave_ms > 0
    rp.close(); out.free()


def test_gather_with_mesh_and_counters(tmp_path):
    """Mesh variant on four partitions; the counters of the ctx are the sum over its partitions (coarse rows recomputed by
    several partitions are counted by each)."""
    from bhusie_amd import assets
    obj = tmp_path / "m.obj"
    obj.write_text(assets.icosphere_mesh_obj(3, radius=8.0, bump=0.1, seed=4))
    model = B.load_model(str(obj))
    tex = T.textures()
    cam = B.Camera(position=(0.0, 0.0, -40.0), forward=(-0.11914522, 0.0, 0.99287683), fov=1.2)
    u = T.uniforms(camera=cam, integration_method=1, model_count=1)
    cfg = B.ladder_for_frame((200, 110), 3, 3)
    one = _single(cfg, u, tex, model=model, counters=True)
    rp = B.RayPass(cfg, devices=[0] * 4, stripe_rows=9, counters=True, frames_in_flight=1)
    rp.set_textures(*tex); rp.upload_model(model); rp.set_materials()
    rp.set_uniforms(*u)
    rp.render()
    assert np.array_equal(rp.read_hdr().view(np.uint32), one.read_hdr().view(np.uint32))
    c1, cn = one.counters(), rp.counters()
    assert cn["traced"] >= c1["traced"] and cn["triangles"] >= c1["triangles"] > 0
    last = len(cfg.sizes()) - 1
    assert rp.level_counters(last)["pixels"] == one.level_counters(last)["pixels"]     # the last level is partitioned exactly
    assert rp.selftest() == (0, 0, 0)
    rp.close(); one.close()


def test_per_frame_scene_state_travels_with_the_frame_through_the_issue_threads(tmp_path):
    """A multi-device ctx hands every frame to its issue threads: uniforms AND the model's per-frame transform (mod.rs:391 re-uploads the model
    every frame; bhray_set_model_transform replaces that) must be the ones in force when bhray_render was called, although the engines pick
    the frame up later.  The mesh comes and goes between frames (a change of kernel variant cuts the staged batch short), frames in flight
    and batches on; every frame against the single ctx."""
    from bhusie_amd import assets
    obj = tmp_path / "m.obj"
    obj.write_text(assets.icosphere_mesh_obj(3, radius=8.0, bump=0.1, seed=4))
    model = B.load_model(str(obj))
    tex = T.textures()
    cam = B.Camera(position=(0.0, 0.0, -40.0), forward=(-0.11914522, 0.0, 0.99287683), fov=1.2)
    cfg = B.ladder_for_frame((200, 110), 3, 3)
    steps = [((-10.0, 0.0, 30.0), 1, 1), ((-10.0, 0.0, 30.0), 1, 0), ((-6.0, 2.0, 33.0), 1, 1), ((-6.0, 2.0, 33.0), 0, 1), ((-12.0, -1.0, 28.0), 1, 1),
             ((-12.0, -1.0, 28.0), 1, 1), ((0.0, 0.0, 60.0), 1, 0)]                          # (model position, visible, integrator)
    one = B.RayPass(cfg, device=0); one.set_textures(*tex); one.upload_model(model)
    want = []
    for k, (pos, vis, method) in enumerate(steps):
        one.set_model_transform(pos, vis); one.set_uniforms(*T.uniforms(camera=cam, integration_method=method, model_count=1, time=0.2 * k)); one.render()
        want.append(one.read_hdr())
    one.close()
    for kw in (dict(devices=[0] * 3, stripe_rows=9, frames_in_flight=2, frames_per_batch=2), dict(devices=[0, 0], slab_row0=[0, 70, 110], frames_in_flight=3)):
        rp = B.RayPass(cfg, **kw)
        rp.set_textures(*tex); rp.upload_model(model)
        bufs = [T.DeviceBuffer(200 * 110 * 16) for _ in steps]
        for k, (pos, vis, method) in enumerate(steps):                                       # all frames enqueued back to back, nothing read in between
            rp.set_model_transform(pos, vis); rp.set_uniforms(*T.uniforms(camera=cam, integration_method=method, model_count=1, time=0.2 * k))
            rp.bind_output(bufs[k].ptr.value, bufs[k].nbytes); rp.render()
        rp.sync()
        for k, b in enumerate(bufs):
            assert np.array_equal(b.read(np.uint32), want[k].view(np.uint32).ravel()), (kw, k)
            b.free()
        rp.close()
    assert not np.array_equal(want[0], want[2]) and not np.array_equal(want[2], want[3])      # the mesh did move and did disappear


def test_bench_frame_on_eight_partitions_is_the_single_gpu_frame():
    """configs[1]/[3] shape at full 1920x1080: 8 partitions x 27-row stripes, 8 frames per batch, speculative levels — the
    production multi-GPU configuration of bench.py — equals the single-ctx frame byte for byte."""
    tex = T.textures(small=False)
    u = T.uniforms(integration_method=1)
    cfg = B.ladder_for_frame((1920, 1080), 3, 4)
    one = _single(cfg, u, tex, frames_in_flight=1)
    want = one.read_hdr()
    one.close()
    rp = B.RayPass(cfg, devices=[0] * 8, frames_per_batch=8, frames_in_flight=2, speculative_levels=2)
    rp.set_textures(*tex)
    rp.set_uniforms(*u)
    for _ in range(19):                                      # two full batches + a partial one, slots reused
        rp.render()
    got = rp.read_hdr()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    rp.close()


def test_cpp_host_program_row_tiled(tmp_path):
    """The C++ host mirror with a device list: Renderer({..}, devices) -> the same bytes as the single-device program."""
    exe = os.path.join(os.path.dirname(B.LIB_PATH), "bhray_render")
    a, b = tmp_path / "a.f32", tmp_path / "b.f32"
    base = ["--rk", "--base", "24", "14", "--levels", "3", "--disk-size", "64"]
    env = dict(os.environ, GPU_MAX_HW_QUEUES="16")
    r1 = subprocess.run([exe, str(a)] + base, capture_output=True, text=True, timeout=120, env=env)
    r2 = subprocess.run([exe, str(b)] + base + ["--devices", "0,0,0"], capture_output=True, text=True, timeout=120, env=env)
    assert r1.returncode == 0 and r2.returncode == 0, r1.stderr + r2.stderr
    assert r1.stdout.strip().splitlines()[-1] == r2.stdout.strip().splitlines()[-1] == "208x118"      # RCCL prints a version banner on stdout
    assert np.array_equal(np.fromfile(a, dtype=np.uint32), np.fromfile(b, dtype=np.uint32))


def test_bad_multi_device_configs_are_errors():
    cfg = B.ladder_from_base((24, 14), 3, 2)
    with pytest.raises(B.BhrayError):
        B.RayPass(cfg, devices=[0, 99])                      # no such device
    with pytest.raises(B.BhrayError):
        B.RayPass(cfg, devices=[0, 0], gather_root=2)        # root is not a partition
    with pytest.raises(B.BhrayError):
        B.RayPass(cfg, devices=[0] * 17)                     # more than BHRAY_MAX_DEVICES
    rp = B.RayPass(cfg, devices=[0, 0])
    with pytest.raises(B.BhrayError):
        rp.set_materials(bytes(64))                          # MaterialUniform x 8 = 128 bytes
    rp.close()


@pytest.mark.parametrize("kw", [dict(devices=[0, 0, 0], stripe_rows=9), dict(devices=[0, 0, 0, 0], gather_root=2, slab_row0=[0, 20, 55, 55, 110]),
                                dict(devices=[0, 0], stripe_rows=27, frames_per_batch=3, frames_in_flight=2)])
def test_gather_of_the_sky_image_is_the_sky_pass_over_the_whole_frame(kw):
    """BHRAY_F_GATHER_SKY: every partition runs the sky pass over its own rows and 8-byte pixels travel; the image the root assembles is
    byte for byte the sky pass over the frame rendered whole - stripes, slabs (one empty), a root in the middle, batches, slot reuse,
    the asynchronous read - and the RGBA32F outputs are refused."""
    tex = T.textures()
    frames = [T.uniforms(integration_method=1), T.uniforms(integration_method=1, time=0.7, camera=B.Camera(position=(1.0, 2.0, -17.0), forward=(0.0, -0.1, 1.0), fov=1.1)),
              T.uniforms(integration_method=0, step_size=0.2)]
    cfg = B.ladder_for_frame((200, 110), 3, 3)
    want = []
    for u in frames:
        one = B.RayPass(cfg, device=0); one.set_textures(*tex); one.set_uniforms(*u); one.render(); one.resolve_sky(); want.append(one.read_sky()); one.close()
    rp = B.RayPass(cfg, gather_sky=True, **kw)
    rp.set_textures(*tex)
    order = [0, 1, 2, 2, 1, 0, 1]
    bufs = [B.PinnedFrame(110, 200, channels16=True) for _ in order]
    tickets = []
    for i, f in enumerate(order):
        rp.set_uniforms(*frames[f]); rp.render()
        if i % 2:
            rp.resolve_sky()                                     # allowed, does nothing
        tickets.append(rp.read_sky_async(bufs[i]))
    for t in reversed(tickets):
        rp.wait_read(t)
    for i, f in enumerate(order):
        assert np.array_equal(bufs[i].array.view(np.uint16), want[f].view(np.uint16)), f"{kw}: frame {i}"
    assert np.array_equal(rp.read_sky().view(np.uint16), want[order[-1]].view(np.uint16))
    for call in (rp.read_hdr, lambda: rp.read_hdr_async(B.PinnedFrame(110, 200))):
        with pytest.raises(B.BhrayError) as e:
            call()
        assert e.value.code == -5
    info = rp.gather_info()
    rows_elsewhere = 110 - len(B.config_partition_rows(rp.cfg)[kw.get("gather_root", 0)])
    assert info["bytes_received_per_frame"] == rows_elsewhere * 200 * 8, info          # 8 bytes per pixel travel, not 16
    for b in bufs:
        b.free()
    rp.close()


def test_gather_sky_needs_a_multi_gpu_ctx():
    cfg = B.ladder_from_base((24, 14), 3, 2)
    with pytest.raises(B.BhrayError) as e:
        B.RayPass(cfg, device=0, gather_sky=True)
    assert e.value.code == -1


def test_gather_of_the_sky_image_with_an_odd_row_length():
    """rows of an odd number of 8-byte pixels do not start on 16 bytes: the de-interleave's one-pixel path"""
    tex = T.textures()
    u = T.uniforms(integration_method=1)
    cfg = B.ladder_for_frame((199, 111), 3, 3)
    one = B.RayPass(cfg, device=0); one.set_textures(*tex); one.set_uniforms(*u); one.render(); one.resolve_sky(); want = one.read_sky(); one.close()
    rp = B.RayPass(cfg, devices=[0, 0, 0], stripe_rows=5, gather_sky=True)
    rp.set_textures(*tex); rp.set_uniforms(*u); rp.render()
    assert np.array_equal(rp.read_sky().view(np.uint16), want.view(np.uint16))
    rp.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7, 8])
def test_random_partitions_batches_and_roots_assemble_the_single_ctx_frame(seed):
    """Seeded random combinations of frame size, ladder depth, number of partitions, stripes or uneven slabs (empty ones included), root,
    frames per batch, frame slots, integrator and gathered image (RGBA32F frame / RGBA16F sky image): frames in random order through the
    multi-partition ctx equal the frames of a partition-less ctx byte for byte."""
    rng = np.random.default_rng(1000 + seed)
    w, h = int(rng.integers(40, 260)), int(rng.integers(30, 160))
    levels = int(rng.integers(2, 5))
    cfg = B.ladder_for_frame((w, h), 3, levels)
    n = int(rng.integers(2, 6))
    kw = dict(devices=[0] * n, gather_root=int(rng.integers(0, n)), frames_per_batch=int(rng.integers(1, 4)), frames_in_flight=int(rng.integers(1, 4)),
              speculative_levels=int(rng.choice([0, 2])) if levels >= 3 else 0)
    if rng.random() < 0.5:
        kw["stripe_rows"] = int(rng.integers(1, 30))
    else:
        cuts = sorted(int(v) for v in rng.integers(0, h + 1, n - 1))
        kw["slab_row0"] = [0] + cuts + [h]
    sky = bool(rng.random() < 0.5)
    method = int(rng.integers(0, 2))
    tex = T.textures()
    frames = [T.uniforms(integration_method=method, step_size=0.15 if method else 0.2, time=float(t)) for t in (0.0, 0.9, 2.3)]
    want = []
    for u in frames:
        one = B.RayPass(cfg, device=0, speculative_levels=kw["speculative_levels"]); one.set_textures(*tex); one.set_uniforms(*u); one.render()
        if sky:
            one.resolve_sky(); want.append(one.read_sky())
        else:
            want.append(one.read_hdr())
        one.close()
    rp = B.RayPass(cfg, gather_sky=sky, **kw)
    rp.set_textures(*tex)
    order = [int(v) for v in rng.integers(0, 3, 7)]
    word = np.uint16 if sky else np.uint32
    for i, f in enumerate(order):
        rp.set_uniforms(*frames[f]); rp.render()
        if i % 3 == 2 or i == len(order) - 1:                   # read some frames (a read flushes a partial batch), let others pass unread
            got = rp.read_sky() if sky else rp.read_hdr()
            assert np.array_equal(got.view(word), want[f].view(word)), (seed, kw, sky, method, i)
    rp.close()
