"""-m gpu: the hand-off of the ray pass's output to a consumer that is not a HIP client of the same GPU.

In the reference the output is a wgpu texture that SkyPipeline samples (ray_pipeline.rs:297-299, mod.rs:215, sky.wgsl:4,17).
Behind the C ABI that is (1) bhray_read_hdr_async into pinned host memory on the library's copy streams - frame k's copy
overlaps frame k+1's render -, or (2) memory the consumer exports as a file descriptor, imported with bhray_import_external_fd
and bound as the output (zero copy).  Everything here compares bytes with the synchronous bhray_read_hdr."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import bhusie_amd as B
from tests import common as T

pytestmark = pytest.mark.gpu
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def _frames():
    cam2 = B.Camera(position=(1.0, 2.0, -17.0), forward=(0.0, -0.1, 1.0), fov=1.1)
    return [T.uniforms(integration_method=1), T.uniforms(integration_method=1, camera=cam2, time=0.7),
            T.uniforms(integration_method=0, step_size=0.2)]


def _want(cfg, tex, frames):
    out = []
    for u in frames:
        rp = B.RayPass(cfg, device=0)
        rp.set_textures(*tex); rp.set_uniforms(*u); rp.render()
        out.append(rp.read_hdr()); rp.close()
    return out


@pytest.mark.parametrize("fif", [1, 3])
def test_async_read_delivers_every_frame_and_protects_the_slot(fif):
    """Render A, enqueue its copy, render B, C, ... WITHOUT waiting: with one frame slot the next render reuses the image the copy is
    still reading - the library must order it behind the copy.  Every pinned buffer must hold its own frame."""
    tex = T.textures()
    frames = _frames()
    cfg = B.ladder_for_frame((200, 110), 3, 3)
    want = _want(cfg, tex, frames)
    rp = B.RayPass(cfg, device=0, frames_in_flight=fif)
    rp.set_textures(*tex)
    order = [0, 1, 2, 1, 0, 2, 2, 1]
    bufs = [B.PinnedFrame(110, 200) for _ in order]
    for b in bufs:
        b.array[...] = -1.0
    tickets = []
    for i, f in enumerate(order):
        rp.set_uniforms(*frames[f])
        rp.render()
        tickets.append(rp.read_hdr_async(bufs[i]))
    for i in reversed(range(len(order))):                     # any order
        rp.wait_read(tickets[i])
    for i, f in enumerate(order):
        assert np.array_equal(bufs[i].array.view(np.uint32), want[f].view(np.uint32)), f"frame {i}"
    assert np.array_equal(rp.read_hdr().view(np.uint32), want[order[-1]].view(np.uint32))      # the synchronous read still works
    with pytest.raises(B.BhrayError):
        rp.wait_read(10 ** 6)
    for b in bufs:
        b.free()
    rp.close()


def test_async_read_more_tickets_than_the_ring_and_a_padded_pitch():
    """> 64 outstanding tickets (the 65th call waits for the oldest) and a destination pitch wider than a row."""
    tex = T.textures()
    u = T.uniforms(integration_method=1)
    cfg = B.ladder_from_base((24, 14), 3, 2)
    want = _want(cfg, tex, [u])[0]
    h, w = want.shape[:2]
    rp = B.RayPass(cfg, device=0, frames_in_flight=2)
    rp.set_textures(*tex); rp.set_uniforms(*u)
    buf = B.PinnedFrame(h, w + 4)                             # pitch = (w + 4) * 16
    buf.array[...] = -2.0
    L = B.lib()
    t = C.c_uint64()
    for _ in range(70):
        rp.render()
        B.check(L.bhray_read_hdr_async(rp._h, C.c_void_p(buf.ptr), (w + 4) * 16, C.byref(t)), rp._h)
    rp.wait_read(int(t.value))
    assert t.value == 69
    rp.wait_read(0)                                           # long recycled: returns at once
    assert np.array_equal(buf.array[:, :w].view(np.uint32), want.view(np.uint32))
    assert (buf.array[:, w:] == -2.0).all()
    buf.free(); rp.close()


def test_async_read_of_the_sky_pass_image():
    """RGBA16F hand-off (half the bytes): render, resolve the sky, enqueue the copy, go on rendering; single ctx and gathered frame."""
    tex = T.textures()
    frames = _frames()
    cfg = B.ladder_for_frame((200, 110), 3, 3)
    want = []
    for u in frames:
        rp = B.RayPass(cfg, device=0); rp.set_textures(*tex); rp.set_uniforms(*u); rp.render(); rp.resolve_sky(); want.append(rp.read_sky()); rp.close()
    for kw in (dict(device=0, frames_in_flight=2), dict(devices=[0, 0], stripe_rows=9, frames_in_flight=1)):
        rp = B.RayPass(cfg, **kw)
        rp.set_textures(*tex)
        bufs = [B.PinnedFrame(110, 200, channels16=True) for _ in range(5)]
        tk = []
        for i in range(5):
            rp.set_uniforms(*frames[i % 3]); rp.render(); rp.resolve_sky()
            tk.append(rp.read_sky_async(bufs[i]))
        for t in tk:
            rp.wait_read(t)
        for i in range(5):
            assert np.array_equal(bufs[i].array.view(np.uint16), want[i % 3].view(np.uint16)), f"{kw}: frame {i}"
        for b in bufs:
            b.free()
        rp.close()
    rp = B.RayPass(cfg, device=0); rp.set_textures(*tex); rp.set_uniforms(*frames[0]); rp.render()
    with pytest.raises(B.BhrayError):
        rp.read_sky_async(B.PinnedFrame(110, 200, channels16=True))       # the sky pass has not run for this frame
    rp.close()


def test_async_read_of_a_gathered_frame():
    """Multi-partition ctx: the copy reads the ASSEMBLED frame on the root, behind the gather and the de-interleave; the root's next
    render into the same slot (its own rows go straight into that frame) waits for it."""
    tex = T.textures()
    frames = _frames()
    cfg = B.ladder_for_frame((200, 110), 3, 3)
    want = _want(cfg, tex, frames)
    rp = B.RayPass(cfg, devices=[0, 0, 0], stripe_rows=9, frames_in_flight=1)
    rp.set_textures(*tex)
    bufs = [B.PinnedFrame(110, 200) for _ in range(6)]
    tickets = []
    for i in range(6):
        rp.set_uniforms(*frames[i % 3])
        rp.render()
        tickets.append(rp.read_hdr_async(bufs[i]))
    for t in tickets:
        rp.wait_read(t)
    for i in range(6):
        assert np.array_equal(bufs[i].array.view(np.uint32), want[i % 3].view(np.uint32)), f"frame {i}"
    for b in bufs:
        b.free()
    rp.close()


def test_sky_pass_of_a_gathered_frame_is_not_torn_by_the_next_render():
    """in gather mode the sky pass runs on the root's communication stream; the root's next render into the same slot
    writes its own rows straight into the frame the sky pass is reading.  Render A, resolve its sky, render B (different camera)
    at once: A's sky image, whose copy to the host was enqueued before B, must be sky(A), not a mixture."""
    tex = T.textures(small=False)
    frames = _frames()
    cfg = B.ladder_for_frame((1920, 1080), 3, 4)
    one = B.RayPass(cfg, device=0)
    one.set_textures(*tex); one.set_uniforms(*frames[0]); one.render(); one.resolve_sky()
    want = one.read_sky()
    one.close()
    rp = B.RayPass(cfg, devices=[0, 0], frames_in_flight=1)
    rp.set_textures(*tex)
    buf = B.PinnedFrame(1080, 1920, channels16=True)
    for _ in range(3):
        buf.array[...] = 0
        rp.set_uniforms(*frames[0]); rp.render(); rp.resolve_sky()
        t = rp.read_sky_async(buf)                            # A's sky image, copy enqueued - nothing waited for
        rp.set_uniforms(*frames[1]); rp.render()              # same slot, different frame
        with pytest.raises(B.BhrayError):
            rp.read_sky()                                     # the image on hand is A's, the current frame is B: refused (as on one GPU)
        rp.wait_read(t)
        assert np.array_equal(buf.array.view(np.uint16), want.view(np.uint16))
    buf.free()
    rp.close()


def test_two_wait_streams_before_one_render_are_two_dependencies():
    """bhray_wait_stream kept ONE event; a second call before the render re-recorded it and the first stream's
    dependency was lost.  Two caller streams each hold back the render until their own (slow) memset has finished."""
    hip = C.CDLL("libamdhip64.so")
    tex = T.textures()
    u = T.uniforms(integration_method=1)
    cfg = B.ladder_from_base((24, 14), 3, 2)
    want = _want(cfg, tex, [u])[0]
    h, w = want.shape[:2]
    n = h * w * 16
    rp = B.RayPass(cfg, device=0, frames_in_flight=1)
    rp.set_textures(*tex); rp.set_uniforms(*u)
    out = T.DeviceBuffer(n)
    s = [C.c_void_p(), C.c_void_p()]
    big = T.DeviceBuffer(1 << 30, fill=0)
    for k in range(2):
        assert hip.hipStreamCreateWithFlags(C.byref(s[k]), 1) == 0
    # each caller stream: a long memset, then a memset of the output to a pattern; the render must come after BOTH
    for k in range(2):
        for _ in range(4):
            assert hip.hipMemsetAsync(C.c_void_p(big.ptr.value + k * (1 << 29)), k + 1, C.c_size_t(1 << 29), s[k]) == 0
        assert hip.hipMemsetAsync(out.ptr, 0x40 + k, C.c_size_t(n), s[k]) == 0
        rp.wait_stream(s[k].value)
    rp.bind_output(out.ptr.value, n)
    rp.render()
    rp.sync()
    got = out.read().reshape(h, w, 4)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "a caller stream's memset landed after the render"
    for k in range(2):
        hip.hipStreamSynchronize(s[k]); hip.hipStreamDestroy(s[k])
    out.free(); big.free(); rp.close()


def test_external_memory_import_is_a_zero_copy_output():
    """What a Vulkan / wgpu host does with VK_KHR_external_memory_fd, replayed with HIP on both sides: an allocation is exported as a
    dma-buf file descriptor (hipMemGetHandleForAddressRange), imported through the ABI (bhray_import_external_fd ->
    hipImportExternalMemory), bound as the output; the frame must appear in the ORIGINAL allocation.  Records what the runtime
    supports in gpurun_out/external_import.json; a runtime that cannot export a dma-buf skips (nothing to import)."""
    hip = C.CDLL("libamdhip64.so")
    tex = T.textures()
    u = T.uniforms(integration_method=1)
    cfg = B.ladder_from_base((24, 14), 3, 2)
    want = _want(cfg, tex, [u])[0]
    h, w = want.shape[:2]
    n = h * w * 16
    size = (n + (1 << 21) - 1) & ~((1 << 21) - 1)             # whole 2 MiB pages
    os.makedirs(OUT, exist_ok=True)
    rec = {"bytes": size}
    buf = T.DeviceBuffer(size)
    fd = C.c_int(-1)
    if not hasattr(hip, "hipMemGetHandleForAddressRange"):
        rec["export"] = "hipMemGetHandleForAddressRange not in this runtime"
        json.dump(rec, open(os.path.join(OUT, "external_import.json"), "w"))
        pytest.skip(rec["export"])
    rc = hip.hipMemGetHandleForAddressRange(C.byref(fd), buf.ptr, C.c_size_t(size), 1, C.c_ulonglong(0))     # hipMemRangeHandleTypeDmaBufFd = 1
    rec["export_rc"] = int(rc); rec["fd"] = int(fd.value)
    if rc != 0 or fd.value < 0:
        json.dump(rec, open(os.path.join(OUT, "external_import.json"), "w"))
        pytest.skip("the runtime does not export a dma-buf fd for a hipMalloc allocation (rc %d)" % rc)
    rp = B.RayPass(cfg, device=0, frames_in_flight=1)
    rp.set_textures(*tex); rp.set_uniforms(*u)
    try:
        p = rp.import_external_fd(fd.value, size)
    except B.BhrayError as e:
        rec["import"] = str(e)
        json.dump(rec, open(os.path.join(OUT, "external_import.json"), "w"))
        rp.close(); buf.free()
        pytest.skip("hipImportExternalMemory refused the dma-buf fd: %s" % e)
    rec["import"] = "ok"; rec["same_address"] = bool(p == buf.ptr.value)
    rp.bind_output(p, n)
    rp.render(); rp.sync()
    got = buf.read()[: h * w * 4].reshape(h, w, 4)
    rec["frame_visible_through_the_exporters_pointer"] = bool(np.array_equal(got.view(np.uint32), want.view(np.uint32)))
    json.dump(rec, open(os.path.join(OUT, "external_import.json"), "w"))
    assert rec["frame_visible_through_the_exporters_pointer"]
    rp.release_external(p)
    with pytest.raises(B.BhrayError):
        rp.release_external(p)
    os.close(fd.value)
    rp.close(); buf.free()


def test_a_stale_sky_image_is_refused_until_the_sky_pass_runs_again():
    """the sky image belongs to the frame it was resolved from.  After the NEXT render the image is stale: reading it
    (either read) is a state error, not the previous frame's pixels; resolve_sky() makes it current again."""
    tex = T.textures()
    frames = _frames()
    cfg = B.ladder_for_frame((200, 110), 3, 3)
    want = []
    for u in frames[:2]:
        rp = B.RayPass(cfg, device=0); rp.set_textures(*tex); rp.set_uniforms(*u); rp.render(); rp.resolve_sky(); want.append(rp.read_sky()); rp.close()
    for kw in (dict(device=0), dict(device=0, frames_in_flight=3, frames_per_batch=2)):
        rp = B.RayPass(cfg, **kw)
        rp.set_textures(*tex)
        buf = B.PinnedFrame(110, 200, channels16=True)
        rp.set_uniforms(*frames[0]); rp.render(); rp.resolve_sky()
        assert np.array_equal(rp.read_sky().view(np.uint16), want[0].view(np.uint16))
        rp.set_uniforms(*frames[1]); rp.render()                  # a new frame: the sky image on hand is frame 0's
        with pytest.raises(B.BhrayError) as e:
            rp.read_sky_async(buf)
        assert e.value.code == -5                                 # BHRAY_E_STATE
        with pytest.raises(B.BhrayError):
            rp.read_sky()
        rp.resolve_sky()
        rp.wait_read(rp.read_sky_async(buf))
        assert np.array_equal(buf.array.view(np.uint16), want[1].view(np.uint16)), kw
        buf.free(); rp.close()


def test_a_refused_async_read_of_a_gathered_frame_consumes_no_ticket():
    """multi-partition ctx, bhray_read_hdr_async with a pitch narrower than a row is refused BEFORE a ticket is taken: the
    next valid call gets the ticket the refused one would have had, and waiting on it delivers the frame."""
    tex = T.textures()
    u = _frames()[0]
    cfg = B.ladder_for_frame((200, 110), 3, 3)
    want = _want(cfg, tex, [u])[0]
    L = B.lib()
    for kw in (dict(device=0), dict(devices=[0, 0], stripe_rows=9)):
        rp = B.RayPass(cfg, **kw)
        rp.set_textures(*tex); rp.set_uniforms(*u); rp.render()
        buf = B.PinnedFrame(110, 200)
        buf.array[...] = -3.0
        first = rp.read_hdr_async(buf)
        rp.wait_read(first)
        t = C.c_uint64(12345)
        rc = L.bhray_read_hdr_async(rp._h, buf.array.ctypes.data_as(C.POINTER(C.c_float)), 200 * 16 - 16, C.byref(t))
        assert rc != 0, kw
        rc = L.bhray_read_hdr_async(rp._h, None, 200 * 16, C.byref(t))
        assert rc != 0, kw
        buf.array[...] = -3.0
        second = rp.read_hdr_async(buf)
        assert second == first + 1, (kw, first, second)           # the two refused calls took none
        rp.wait_read(second)
        assert np.array_equal(buf.array.view(np.uint32), want.view(np.uint32)), kw
        buf.free(); rp.close()


def _handoff_frames(tmp_path, form, frames, pixel_bytes):
    import subprocess
    exe = os.path.join(os.path.dirname(B.LIB_PATH), "bhray_render")
    out = tmp_path / f"{form}.bin"
    r = subprocess.run([exe, "--handoff", form, str(frames), str(out), "--rk", "--base", "24", "14", "--levels", "3"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    size, delivered = r.stdout.split()
    w, h = (int(v) for v in size.split("x"))
    assert int(delivered) == frames, (form, r.stdout)             # every frame reaches the host, the last ones after the drain
    data = np.fromfile(out, dtype=np.uint8)
    assert data.size == frames * w * h * pixel_bytes
    return data.reshape(frames, h, w, pixel_bytes)


def test_the_shim_hands_the_host_the_same_frames_in_every_form(tmp_path):
    """INTEGRATION.md §3's shim as the C++ program runs it (host/renderer.hpp, Renderer::render_handoff; time advances 1/60 per frame):
    the asynchronous forms - two frames in flight, the host shown frame k-1 while k renders - deliver, in order, exactly the frames the
    synchronous form delivers; the temporal flag changes when pixels are computed, not their values."""
    n = 7
    sync = _handoff_frames(tmp_path, "sync", n, 16)
    hdr = _handoff_frames(tmp_path, "hdr", n, 16)
    assert np.array_equal(sync, hdr)
    assert not np.array_equal(sync[0], sync[1])                   # the scene does move (disk rotation), so frame order is tested
    sky1 = _handoff_frames(tmp_path, "sky1", n, 8)
    sky = _handoff_frames(tmp_path, "sky", n, 8)
    skyt = _handoff_frames(tmp_path, "sky-temporal", n, 8)
    assert np.array_equal(sky1, sky)
    assert np.array_equal(sky1, skyt)
    assert not np.array_equal(sky1[0], sky1[1])
    # the sky image is the sky pass over the same HDR frame: where the HDR pixel already carries its colour (alpha 1: disk, horizon) the
    # two agree to half precision
    hdr_f = sync.view(np.float32).reshape(n, sync.shape[1], sync.shape[2], 4)
    sky_f = sky1.view(np.float16).reshape(n, sky1.shape[1], sky1.shape[2], 4).astype(np.float32)
    direct = hdr_f[..., 3] == 1.0
    assert direct.any()
    assert np.allclose(sky_f[direct][:, :3], hdr_f[direct][:, :3], rtol=2e-3, atol=1e-4)


def test_the_dropin_measurement_mode_reports_every_leg(tmp_path):
    """bench.py's `dropin` object comes from `bhray_render --dropin`: a tiny run must finish and name the six hand-off legs."""
    import subprocess
    exe = os.path.join(os.path.dirname(B.LIB_PATH), "bhray_render")
    r = subprocess.run([exe, "--dropin", "4", "--rk", "--base", "24", "14", "--levels", "3"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    legs = rep["legs"] if "legs" in rep else rep
    names = [l["name"] for l in legs] if isinstance(legs, list) else list(legs)
    for want in ("sync_read_hdr", "async_sky_rgba16f"):
        assert any(want in x for x in names), names
    assert len(names) >= 6
