"""RayPass / Renderer — the Python mirror of the reference's ray-pass surface.

RayPass   = the chain of RayPipelines (src/renderer/pipelines/ray_pipeline.rs:28-310) as one
            object over a bhray_ctx: textures, model, per-frame uniforms, dispatch, output.
Renderer  = the part of Renderer::{new,render} on the path (src/renderer/mod.rs:113-207,
            378-420): default RayDetails, the 72x41 x3 x4 ladder, per-frame call order.
All pixels come from libbhray (gfx950); nothing is computed here.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import lib
from .layouts import (PARTITION_SLABS, F_COUNTERS, F_EVAL_FMA, F_GATHER_SKY, F_LITERAL, F_TEMPORAL, F_TIMING, F_TIMING_SPARSE, GATHER_RCCL, TEX_DISK, TEX_SKY, TEX_TEMP_LUT, BhrayConfig, BhrayCounters,
                      BhrayGatherInfo, BhrayRebalanceInfo, BhrayTiming, check)
from .model import Model
from .scene import BlackHole, Camera, RayDetails


def ladder_from_base(base=(72, 41), multiplier=3, levels=4) -> BhrayConfig:
    cfg = BhrayConfig()
    check(lib().bhray_ladder_from_base(base[0], base[1], multiplier, levels, C.byref(cfg)))
    return cfg


def ladder_for_frame(frame=(1920, 1080), multiplier=3, levels=4) -> BhrayConfig:
    cfg = BhrayConfig()
    check(lib().bhray_ladder_for_frame(frame[0], frame[1], multiplier, levels, C.byref(cfg)))
    return cfg


def comm_unique_id() -> bytes:
    """A fresh RCCL communicator id (one process per GPU: call on ONE rank, hand the bytes to every rank's RayPass)."""
    buf = (C.c_uint8 * 128)()
    check(lib().bhray_comm_unique_id(buf))
    return bytes(buf)


def partition_rows(frame_h: int, world: int, stripe_rows: int = 27):
    """rows[part] = frame rows of every partition, from the library's own partition arithmetic (host only)."""
    L = lib()
    out, r = [], C.c_uint32()
    for part in range(world):
        n = int(L.bhray_partition_rows(frame_h, world, stripe_rows, part))
        rows = np.zeros(n, dtype=np.int64)
        for i in range(n):
            check(L.bhray_partition_row_index(frame_h, world, stripe_rows, part, i, C.byref(r)))
            rows[i] = r.value
        out.append(rows)
    return out


def config_partition_rows(cfg: BhrayConfig):
    """rows[part] for the partition a config describes (stripes or slabs)."""
    L = lib()
    world = cfg.device_count if cfg.device_count >= 2 else max(1, cfg.row_world)
    out, r = [], C.c_uint32()
    for part in range(world):
        n = int(L.bhray_config_partition_rows(C.byref(cfg), part))
        rows = np.zeros(n, dtype=np.int64)
        for i in range(n):
            check(L.bhray_config_partition_row_index(C.byref(cfg), part, i, C.byref(r)))
            rows[i] = r.value
        out.append(rows)
    return out


def balance_slabs(cfg: BhrayConfig, row_work, world: int):
    """slab_row0[0..world] that minimise the largest partition's work; row_work[l] = per-row work of ladder level l of a calibration
    frame rendered whole (RayPass.row_work()) - bhray_balance_slabs, pure host arithmetic."""
    assert len(row_work) == cfg.levels
    arrs = [np.ascontiguousarray(a, dtype=np.uint64) for a in row_work]
    for l, a in enumerate(arrs):
        assert a.shape == (cfg.level_h[l],), (l, a.shape)
    ptrs = (C.POINTER(C.c_uint64) * cfg.levels)(*[a.ctypes.data_as(C.POINTER(C.c_uint64)) for a in arrs])
    out = (C.c_uint32 * (world + 1))()
    check(lib().bhray_balance_slabs(C.byref(cfg), ptrs, world, out))
    return [int(v) for v in out]


def rebalance_slabs(frame_h: int, slab_row0, part_ms, row_weight: np.ndarray, extra_ms=None, shift_rows: float = 0.0):
    """bhray_rebalance_slabs: (new bounds, predicted slowest partition); row_weight (float64[frame_h], zeros before the first call) is updated in place."""
    world = len(part_ms)
    if len(slab_row0) != world + 1:
        raise ValueError(f"rebalance_slabs: {world} partitions need {world + 1} bounds, got {len(slab_row0)}")
    if extra_ms is not None and len(extra_ms) != world:
        raise ValueError(f"rebalance_slabs: extra_ms has {len(extra_ms)} entries for {world} partitions")
    assert row_weight.dtype == np.float64 and row_weight.shape == (frame_h,) and row_weight.flags.c_contiguous
    b_in = (C.c_uint32 * (world + 1))(*[int(v) for v in slab_row0])
    ms = (C.c_double * world)(*[float(v) for v in part_ms])
    ex = (C.c_double * world)(*[float(v) for v in extra_ms]) if extra_ms is not None else None
    out = (C.c_uint32 * (world + 1))()
    pred = C.c_double()
    check(lib().bhray_rebalance_slabs(frame_h, world, b_in, ms, ex, float(shift_rows), row_weight.ctypes.data_as(C.POINTER(C.c_double)), out, C.byref(pred)))
    return [int(v) for v in out], float(pred.value)


class RayPass:
    """devices=[d0, d1, ...]: ONE ctx drives several GPUs (partition i on devices[i]); bhray_render then also gathers the row
    tiles to `gather_root`'s GPU over RCCL and de-interleaves them, and every output call refers to the whole frame.
    comm_id + row_rank/row_world: one process per GPU, the same gather enqueued by every rank's library."""

    def __init__(self, cfg: BhrayConfig, device=0, counters=False, timing=False, row_rank=0, row_world=1, stripe_rows=27,
                 frames_in_flight=0, speculative_levels=0, frames_per_batch=0, devices=None, gather_root=0, comm_id=None,
                 literal=False, superset_levels=0, temporal=False, eval_fma=False, slab_row0=None, gather_sky=False):
        cfg = BhrayConfig.from_buffer_copy(bytes(cfg))
        cfg.struct_size = C.sizeof(BhrayConfig)
        cfg.device = device
        if devices is not None:
            cfg.device_count = len(devices)                      # more than BHRAY_MAX_DEVICES: refused by bhray_create
            for i, d in enumerate(list(devices)[:len(cfg.devices)]):
                cfg.devices[i] = int(d)
        cfg.gather_root = gather_root
        if comm_id is not None:
            assert len(comm_id) == 128
            cfg.gather = GATHER_RCCL
            C.memmove(cfg.comm_id, bytes(comm_id), 128)
        cfg.flags = (F_COUNTERS if counters else 0) | (F_TIMING_SPARSE if timing == "sparse" else (F_TIMING if timing else 0)) | (F_LITERAL if literal else 0) | (F_TEMPORAL if temporal else 0) | (F_EVAL_FMA if eval_fma else 0) | (F_GATHER_SKY if gather_sky else 0)
        cfg.row_rank, cfg.row_world, cfg.stripe_rows = row_rank, row_world, stripe_rows
        if slab_row0 is not None:                                # contiguous slabs instead of interleaved stripes (bhray_config.partition)
            cfg.partition = PARTITION_SLABS
            for i, v in enumerate(list(slab_row0)[:len(cfg.slab_row0)]):
                cfg.slab_row0[i] = int(v)
        cfg.frames_in_flight = frames_in_flight
        cfg.speculative_levels = speculative_levels
        cfg.frames_per_batch = frames_per_batch
        cfg.superset_levels = superset_levels
        self.cfg = cfg
        h = C.c_void_p()
        self._L = lib()                                          # the library this ctx belongs to: EVERY call on self._h goes through it (a handle of one .so must never be driven by another)
        check(self._L.bhray_create(C.byref(cfg), C.byref(h)))
        self._h = h

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.bhray_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- static inputs
    def set_texture(self, slot: int, rgba: np.ndarray):
        a = np.ascontiguousarray(rgba, dtype=np.uint8)
        assert a.ndim == 3 and a.shape[2] == 4
        check(self._L.bhray_set_texture(self._h, slot, a.ctypes.data, a.shape[1], a.shape[0]), self._h, self._L)

    def set_textures(self, temp_lut, disk, sky):
        self.set_texture(TEX_TEMP_LUT, temp_lut); self.set_texture(TEX_DISK, disk); self.set_texture(TEX_SKY, sky)

    def upload_model(self, model: Model, index=0):
        d = model.desc()
        check(self._L.bhray_upload_model(self._h, index, C.byref(d)), self._h, self._L)

    def upload_model_uniform(self, blob: bytes, index=0):
        check(self._L.bhray_upload_model_uniform(self._h, index, blob, len(blob)), self._h, self._L)

    def set_materials(self, blob: bytes = bytes(128)):
        """mod.rs:389 — accepted and ignored (the shader never reads the materials)."""
        check(self._L.bhray_set_materials(self._h, blob, len(blob)), self._h, self._L)

    def row_work(self):
        """[per-row iterations of the last render, one uint64 array per ladder level] (needs counters=True) - the input of balance_slabs."""
        out = []
        for l in range(self.cfg.levels):
            a = np.zeros(self.cfg.level_h[l], dtype=np.uint64)
            check(self._L.bhray_get_row_work(self._h, l, a.ctypes.data_as(C.POINTER(C.c_uint64)), a.size), self._h, self._L)
            out.append(a)
        return out

    def gather_info(self) -> dict:
        g = BhrayGatherInfo()
        check(self._L.bhray_get_gather_info(self._h, C.byref(g)), self._h, self._L)
        return g.as_dict()

    # -- run-time partition (multi-GPU balance that follows the scene)
    def set_partition(self, slab_row0):
        """From the next render on partition p owns frame rows [slab_row0[p], slab_row0[p + 1]) - no re-create (bhray_set_partition)."""
        parts = int(self.cfg.device_count) if self.cfg.device_count >= 2 else max(1, int(self.cfg.row_world))
        if len(slab_row0) != parts + 1:                     # the library reads partitions + 1 entries whatever it is handed
            raise ValueError(f"set_partition: {parts} partitions need {parts + 1} bounds, got {len(slab_row0)}")
        a = (C.c_uint32 * (parts + 1))(*[int(v) for v in slab_row0])
        check(self._L.bhray_set_partition(self._h, a), self._h, self._L)

    def get_partition(self):
        a, n = (C.c_uint32 * 17)(), C.c_uint32()
        check(self._L.bhray_get_partition(self._h, a, C.byref(n)), self._h, self._L)
        return [int(v) for v in a[:n.value + 1]]

    def work(self):
        """(integrator steps the trace waves issued, pixels the classify launches visited) per frame, over the frames the slots still hold (bhray_get_work)"""
        ws, px, n = C.c_double(), C.c_double(), C.c_uint32()
        check(self._L.bhray_get_work(self._h, C.byref(ws), C.byref(px), C.byref(n)), self._h, self._L)
        return float(ws.value), float(px.value), int(n.value)

    def partition_costs(self):
        """(cost[partitions], extra[partitions]) of the partitions this ctx renders - bhray_rebalance's own numbers (bhray_get_partition_costs)"""
        n = int(self.cfg.device_count) if self.cfg.device_count >= 2 else max(1, int(self.cfg.row_world))
        cost, extra = (C.c_double * n)(), (C.c_double * n)()
        check(self._L.bhray_get_partition_costs(self._h, cost, extra, None), self._h, self._L)
        return [float(v) for v in cost], [float(v) for v in extra]

    def rebalance(self) -> dict:
        """New slab bounds from the work the ctx's kernels counted for the frames in its slots; applied when they promise >= 2 % (bhray_rebalance)."""
        info = BhrayRebalanceInfo()
        check(self._L.bhray_rebalance(self._h, C.byref(info)), self._h, self._L)
        return info.as_dict()

    def set_model_transform(self, position, visible=1, index=0):
        check(self._L.bhray_set_model_transform(self._h, index, (C.c_float * 3)(*[float(x) for x in position]), int(visible)), self._h, self._L)

    # -- per frame
    def set_uniforms(self, camera: bytes, black_hole: bytes, details: bytes):
        assert len(camera) == 32 and len(black_hole) == 132 and len(details) == 32
        check(self._L.bhray_set_uniforms(self._h, camera, black_hole, details), self._h, self._L)

    def render(self):
        check(self._L.bhray_render(self._h), self._h, self._L)

    def flush(self):
        """frames_per_batch > 1: enqueue the launches of the frames staged so far."""
        check(self._L.bhray_flush(self._h), self._h, self._L)

    def sync(self):
        check(self._L.bhray_sync(self._h), self._h, self._L)

    # -- output
    @property
    def frame_size(self):
        return int(self.cfg.frame_w), int(self.cfg.frame_h)

    def local_rows(self) -> np.ndarray:
        n = int(self._L.bhray_local_rows(self._h))
        out = np.zeros(n, dtype=np.uint32)
        r = C.c_uint32()
        for i in range(n):
            check(self._L.bhray_local_row_index(self._h, i, C.byref(r)), self._h, self._L)
            out[i] = r.value
        return out

    def read_hdr(self) -> np.ndarray:
        n = int(self._L.bhray_local_rows(self._h))
        out = np.empty((n, int(self.cfg.frame_w), 4), dtype=np.float32)
        check(self._L.bhray_read_hdr(self._h, out.ctypes.data, int(self.cfg.frame_w) * 16), self._h, self._L)
        return out

    def read_hdr_async(self, dst: "PinnedFrame") -> int:
        """Enqueue the device->host copy of the most recently enqueued frame into pinned memory, behind its kernels; returns a ticket."""
        t = C.c_uint64()
        check(self._L.bhray_read_hdr_async(self._h, C.c_void_p(dst.ptr), int(self.cfg.frame_w) * 16, C.byref(t)), self._h, self._L)
        return int(t.value)

    def read_sky_async(self, dst: "PinnedFrame") -> int:
        """The RGBA16F image of the sky pass (resolve_sky first) into pinned memory (a PinnedFrame(rows, width, channels16=True))."""
        t = C.c_uint64()
        check(self._L.bhray_read_sky_async(self._h, C.c_void_p(dst.ptr), int(self.cfg.frame_w) * 8, C.byref(t)), self._h, self._L)
        return int(t.value)

    def wait_read(self, ticket: int):
        check(self._L.bhray_wait_read(self._h, C.c_uint64(ticket)), self._h, self._L)

    def import_external_fd(self, fd: int, nbytes: int) -> int:
        """Map memory exported by another API (Vulkan OPAQUE_FD / dma-buf) on the GPU that delivers the frame; returns a device pointer for bind_output."""
        p = C.c_void_p()
        check(self._L.bhray_import_external_fd(self._h, int(fd), nbytes, C.byref(p)), self._h, self._L)
        return p.value

    def release_external(self, ptr: int):
        check(self._L.bhray_release_external(self._h, C.c_void_p(ptr)), self._h, self._L)

    def read_level(self, level: int) -> np.ndarray:
        w, h = int(self.cfg.level_w[level]), int(self.cfg.level_h[level])
        out = np.empty((h, w, 4), dtype=np.float32)
        check(self._L.bhray_read_level(self._h, level, out.ctypes.data, out.strides[0]), self._h, self._L)
        return out

    def resolve_sky(self):
        """sky.wgsl behind the last frame (mod.rs:419): direction pixels -> sky^4; RGBA16F."""
        check(self._L.bhray_resolve_sky(self._h), self._h, self._L)

    def read_sky(self) -> np.ndarray:
        n = int(self._L.bhray_local_rows(self._h))
        out = np.empty((n, int(self.cfg.frame_w), 4), dtype=np.float16)
        check(self._L.bhray_read_sky(self._h, out.ctypes.data, int(self.cfg.frame_w) * 8), self._h, self._L)
        return out

    def device_ptr(self):
        p, n = C.c_void_p(), C.c_size_t()
        check(self._L.bhray_hdr_device_ptr(self._h, C.byref(p), C.byref(n)), self._h, self._L)
        return p.value, n.value

    def bind_output(self, ptr, nbytes):
        check(self._L.bhray_bind_output(self._h, C.c_void_p(ptr), nbytes), self._h, self._L)

    def wait_stream(self, s):
        """The next render starts after everything enqueued so far on hipStream_t `s`."""
        check(self._L.bhray_wait_stream(self._h, C.c_void_p(s)), self._h, self._L)

    def next_stream(self) -> int:
        """hipStream_t (as int) of the slot the next render will use."""
        s = C.c_void_p()
        check(self._L.bhray_next_stream(self._h, C.byref(s)), self._h, self._L)
        return s.value

    def signal_stream(self, s):
        """Work enqueued on hipStream_t `s` from now on starts after the last render."""
        check(self._L.bhray_signal_stream(self._h, C.c_void_p(s)), self._h, self._L)

    def counters(self) -> dict:
        c = BhrayCounters()
        check(self._L.bhray_get_counters(self._h, C.byref(c)), self._h, self._L)
        return c.as_dict()

    def scheduling_counters(self) -> dict:
        """wave steps, rays adopted through the drain-merging mailbox, lane occupancy of the step loop (needs counters=True)"""
        c = BhrayCounters()
        check(self._L.bhray_get_counters(self._h, C.byref(c)), self._h, self._L)
        return c.scheduling()

    def level_counters(self, level: int) -> dict:
        c = BhrayCounters()
        check(self._L.bhray_get_level_counters(self._h, level, C.byref(c)), self._h, self._L)
        return c.as_dict()

    def selftest(self):
        """(1/x mismatches, sqrt mismatches, places where the portable acos increases): exhaustive device checks of the
        properties the exact shortcuts rest on; all must be 0."""
        m = (C.c_uint64 * 3)()
        check(self._L.bhray_selftest(self._h, m), self._h, self._L)
        return int(m[0]), int(m[1]), int(m[2])

    def timing(self) -> BhrayTiming:
        t = BhrayTiming()
        check(self._L.bhray_get_timing(self._h, C.byref(t)), self._h, self._L)
        return t


class PinnedFrame:
    """Pinned host memory for one RGBA32F frame (bhray_host_alloc), viewed as a (rows, width, 4) float32 array."""

    def __init__(self, rows: int, width: int, channels16: bool = False):
        p = C.c_void_p()
        self.nbytes = rows * width * (8 if channels16 else 16)
        check(lib().bhray_host_alloc(self.nbytes, C.byref(p)))
        self.ptr = p.value
        if channels16:          # RGBA16F (the sky pass's image)
            self.array = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint16)), shape=(rows, width, 4)).view(np.float16)
        else:
            self.array = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(rows, width, 4))

    def free(self):
        p, self.ptr = self.ptr, None
        if p:
            self.array = None
            check(lib().bhray_host_free(C.c_void_p(p)))

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Renderer:
    """Renderer::{new,render} restricted to the ray pass (mod.rs:113-207, 378-420)."""

    def __init__(self, cfg: BhrayConfig | None = None, device=0, **kw):
        self.camera = Camera()
        self.black_hole = BlackHole()
        self.ray_details = RayDetails()                      # mod.rs:116-121
        self.ray_pass = RayPass(cfg if cfg is not None else ladder_from_base((72, 41), 3, 4), device=device, **kw)
        self.model: Model | None = None

    def set_model(self, model: Model):
        self.model = model
        self.ray_pass.upload_model(model, 0)
        self.ray_details.model_count = 1                     # mod.rs:384 (scene.models.size())

    def render(self, dt: float = 0.0):
        self.ray_details.time += dt                          # mod.rs:382
        self.ray_pass.set_uniforms(self.camera.uniform(), self.black_hole.uniform(), self.ray_details.uniform())
        self.ray_pass.render()

    def read_hdr(self):
        return self.ray_pass.read_hdr()
