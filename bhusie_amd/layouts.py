"""ctypes mirrors of include/bhray.h (which mirrors the reference's #[repr(C)] structs)."""
from __future__ import annotations

import ctypes as C

MAX_LEVELS = 8
MAX_DEVICES = 16
COMM_ID_BYTES = 128
GATHER_NONE, GATHER_RCCL = 0, 1
PARTITION_STRIPES, PARTITION_SLABS = 0, 1
MAX_MODEL_VERTICES = 524288
MODEL_UNIFORM_BYTES = 48234572
F_COUNTERS = 1
F_TIMING = 2
F_LITERAL = 4
F_TEMPORAL = 8
F_TIMING_SPARSE = 16
F_EVAL_FMA = 32
F_GATHER_SKY = 128
TEX_TEMP_LUT, TEX_DISK, TEX_SKY = 0, 1, 2


class BhrayError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"bhray error {code}: {msg}")
        self.code = code


class BhrayDetails(C.Structure):            # ray_pipeline.rs:3-14
    _fields_ = [("material_count", C.c_int32), ("model_count", C.c_int32), ("time", C.c_float),
                ("integration_method", C.c_int32), ("step_size", C.c_float), ("max_iterations", C.c_int32),
                ("angle_division_threshold", C.c_float), ("highlight_interpolation", C.c_int32)]


class BhrayCameraUniform(C.Structure):      # camera.rs:66-73
    _fields_ = [("position", C.c_float * 3), ("_padding", C.c_uint32), ("forward", C.c_float * 3), ("fov", C.c_float)]


class BhrayBlackHoleUniform(C.Structure):   # blackhole.rs:37-51
    _fields_ = [("accretion_disk_inner", C.c_float), ("accretion_disk_outer", C.c_float),
                ("rotation_speed", C.c_float), ("relativity_sphere_radius", C.c_float),
                ("position", C.c_float * 3), ("show_disk_texture", C.c_int32),
                ("normal", C.c_float * 3), ("show_red_shift", C.c_int32),
                ("rotation_matrix", C.c_float * 12), ("feather_amount", C.c_float), ("pad", C.c_int32 * 8)]


class BhrayBlackHole(C.Structure):          # blackhole.rs:3-13
    _fields_ = [("position", C.c_float * 3), ("accretion_disk_rotation", C.c_float * 3),
                ("accretion_disk_inner", C.c_float), ("accretion_disk_outer", C.c_float),
                ("rotation_speed", C.c_float), ("relativity_sphere_radius", C.c_float),
                ("show_disk_texture", C.c_int32), ("show_red_shift", C.c_int32), ("feather_amount", C.c_float)]


class BhrayNode(C.Structure):               # triangle.rs:45-52
    _fields_ = [("min_corner", C.c_float * 3), ("left_child", C.c_int32), ("max_corner", C.c_float * 3), ("obj_count", C.c_int32)]


class BhrayTriangle(C.Structure):           # triangle.rs:54-63
    _fields_ = [(n, C.c_int32) for n in ("p1", "p2", "p3", "n1", "n2", "n3")]


class BhrayModelDesc(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("visible", C.c_int32),
                ("points", C.c_void_p), ("normals", C.c_void_p), ("triangles", C.c_void_p),
                ("nodes", C.c_void_p), ("bvh_lookup", C.c_void_p),
                ("point_count", C.c_int32), ("normal_count", C.c_int32), ("triangle_count", C.c_int32), ("node_count", C.c_int32)]


class BhrayConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("levels", C.c_uint32),
                ("level_w", C.c_uint32 * MAX_LEVELS), ("level_h", C.c_uint32 * MAX_LEVELS),
                ("crop_x", C.c_uint32), ("crop_y", C.c_uint32), ("frame_w", C.c_uint32), ("frame_h", C.c_uint32),
                ("row_rank", C.c_uint32), ("row_world", C.c_uint32), ("stripe_rows", C.c_uint32), ("flags", C.c_uint32),
                ("frames_in_flight", C.c_uint32), ("speculative_levels", C.c_uint32), ("frames_per_batch", C.c_uint32),
                ("superset_levels", C.c_uint32), ("device_count", C.c_uint32), ("devices", C.c_int32 * MAX_DEVICES), ("gather", C.c_uint32),
                ("gather_root", C.c_uint32), ("comm_id", C.c_uint8 * COMM_ID_BYTES),
                ("partition", C.c_uint32), ("slab_row0", C.c_uint32 * (MAX_DEVICES + 1))]

    def sizes(self):
        return [(int(self.level_w[i]), int(self.level_h[i])) for i in range(self.levels)]


class BhrayCounters(C.Structure):
    FRAME = ("pixels", "copied", "interpolated", "traced", "steps", "flat_iters", "node_pairs", "triangles", "disk_hits", "sky_samples")
    SCHED = ("wave_steps", "rays_adopted", "max_ray_iterations")
    _fields_ = [(n, C.c_uint64) for n in FRAME + SCHED]

    def as_dict(self):
        """the counters that are a property of the frame (equal to the oracle's)"""
        return {n: int(getattr(self, n)) for n in self.FRAME}

    def scheduling(self):
        """how the trace kernel scheduled that work: depends on the build, frames in flight, batches"""
        d = {n: int(getattr(self, n)) for n in self.SCHED}
        d["step_lane_occupancy"] = (self.steps / (64.0 * self.wave_steps)) if self.wave_steps else None
        return d


class BhrayTiming(C.Structure):
    _fields_ = [("frames", C.c_uint32), ("batches", C.c_uint32), ("total_ms", C.c_float), ("trace_ms", C.c_float), ("classify_ms", C.c_float),
                ("trace_launches", C.c_uint32), ("classify_launches", C.c_uint32),
                ("level_trace_ms", C.c_float * MAX_LEVELS), ("level_classify_ms", C.c_float * MAX_LEVELS),
                ("sky_ms", C.c_float), ("sky_launches", C.c_uint32),
                ("gather_ms", C.c_float), ("deinterleave_ms", C.c_float), ("gathers", C.c_uint32),
                ("predicted_trace_ms", C.c_float), ("predicted_launches", C.c_uint32),
                ("trace_exec_ms", C.c_float), ("trace_exec_launches", C.c_uint32)]


class BhrayGatherInfo(C.Structure):
    _fields_ = [("partitions", C.c_uint32), ("local_partitions", C.c_uint32), ("root", C.c_uint32), ("root_is_local", C.c_uint32),
                ("comm_ranks", C.c_uint32), ("rccl_version", C.c_uint32),
                ("bytes_sent_per_frame", C.c_uint64), ("bytes_received_per_frame", C.c_uint64)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


assert C.sizeof(BhrayDetails) == 32 and C.sizeof(BhrayCameraUniform) == 32 and C.sizeof(BhrayBlackHoleUniform) == 132
assert C.sizeof(BhrayNode) == 32 and C.sizeof(BhrayTriangle) == 24

# every symbol include/bhray.h declares: name -> (restype, argtypes)
class BhrayRebalanceInfo(C.Structure):
    """bhray_rebalance_info (include/bhray.h)"""
    _fields_ = [("partitions", C.c_uint32), ("applied", C.c_uint32), ("slab_row0", C.c_uint32 * 17), ("part_cost", C.c_float * 16), ("extra_cost", C.c_float * 16),
                ("slowest_before", C.c_float), ("slowest_predicted", C.c_float), ("frames", C.c_uint32)]

    def as_dict(self):
        n = int(self.partitions)
        return {"partitions": n, "applied": bool(self.applied), "slab_row0": [int(v) for v in self.slab_row0[:n + 1]], "part_cost": [float(v) for v in self.part_cost[:n]],
                "extra_cost": [float(v) for v in self.extra_cost[:n]], "slowest_before": float(self.slowest_before), "slowest_predicted": float(self.slowest_predicted),
                "frames": int(self.frames)}


P = C.POINTER
vp, u32, i32, sz = C.c_void_p, C.c_uint32, C.c_int32, C.c_size_t
SYMBOLS = {
    "bhray_ladder_from_base": (C.c_int, [u32, u32, u32, u32, P(BhrayConfig)]),
    "bhray_ladder_for_frame": (C.c_int, [u32, u32, u32, u32, P(BhrayConfig)]),
    "bhray_create": (C.c_int, [P(BhrayConfig), P(vp)]),
    "bhray_destroy": (None, [vp]),
    "bhray_last_error": (C.c_char_p, [vp]),
    "bhray_strerror": (C.c_char_p, [C.c_int]),
    "bhray_version": (u32, []),
    "bhray_device_count": (C.c_int, []),
    "bhray_partition_rows": (u32, [u32, u32, u32, u32]),
    "bhray_partition_row_index": (C.c_int, [u32, u32, u32, u32, u32, P(u32)]),
    "bhray_config_partition_rows": (u32, [P(BhrayConfig), u32]),
    "bhray_config_partition_row_index": (C.c_int, [P(BhrayConfig), u32, u32, P(u32)]),
    "bhray_balance_slabs": (C.c_int, [P(BhrayConfig), P(P(C.c_uint64)), u32, P(u32)]),
    "bhray_comm_unique_id": (C.c_int, [vp]),
    "bhray_get_gather_info": (C.c_int, [vp, P(BhrayGatherInfo)]),
    "bhray_set_partition": (C.c_int, [vp, P(u32)]),
    "bhray_get_partition": (C.c_int, [vp, P(u32), P(u32)]),
    "bhray_rebalance_slabs": (C.c_int, [u32, u32, P(u32), P(C.c_double), P(C.c_double), C.c_double, P(C.c_double), P(u32), P(C.c_double)]),
    "bhray_rebalance": (C.c_int, [vp, P(BhrayRebalanceInfo)]),
    "bhray_get_work": (C.c_int, [vp, P(C.c_double), P(C.c_double), P(u32)]),
    "bhray_get_partition_costs": (C.c_int, [vp, P(C.c_double), P(C.c_double), P(u32)]),
    "bhray_set_materials": (C.c_int, [vp, vp, sz]),
    "bhray_set_texture": (C.c_int, [vp, C.c_int, vp, u32, u32]),
    "bhray_upload_model_uniform": (C.c_int, [vp, u32, vp, sz]),
    "bhray_upload_model": (C.c_int, [vp, u32, P(BhrayModelDesc)]),
    "bhray_set_model_transform": (C.c_int, [vp, u32, P(C.c_float), i32]),
    "bhray_set_uniforms": (C.c_int, [vp, vp, vp, vp]),
    "bhray_render": (C.c_int, [vp]),
    "bhray_flush": (C.c_int, [vp]),
    "bhray_sync": (C.c_int, [vp]),
    "bhray_read_hdr": (C.c_int, [vp, vp, sz]),
    "bhray_read_level": (C.c_int, [vp, u32, vp, sz]),
    "bhray_local_rows": (u32, [vp]),
    "bhray_local_row_index": (C.c_int, [vp, u32, P(u32)]),
    "bhray_hdr_device_ptr": (C.c_int, [vp, P(vp), P(sz)]),
    "bhray_bind_output": (C.c_int, [vp, vp, sz]),
    "bhray_read_hdr_async": (C.c_int, [vp, vp, sz, P(C.c_uint64)]),
    "bhray_read_sky_async": (C.c_int, [vp, vp, sz, P(C.c_uint64)]),
    "bhray_wait_read": (C.c_int, [vp, C.c_uint64]),
    "bhray_host_alloc": (C.c_int, [sz, P(vp)]),
    "bhray_host_free": (C.c_int, [vp]),
    "bhray_import_external_fd": (C.c_int, [vp, C.c_int, sz, P(vp)]),
    "bhray_release_external": (C.c_int, [vp, vp]),
    "bhray_resolve_sky": (C.c_int, [vp]),
    "bhray_read_sky": (C.c_int, [vp, vp, sz]),
    "bhray_sky_device_ptr": (C.c_int, [vp, P(vp), P(sz)]),
    "bhray_wait_stream": (C.c_int, [vp, vp]),
    "bhray_signal_stream": (C.c_int, [vp, vp]),
    "bhray_next_stream": (C.c_int, [vp, P(vp)]),
    "bhray_get_counters": (C.c_int, [vp, P(BhrayCounters)]),
    "bhray_get_level_counters": (C.c_int, [vp, u32, P(BhrayCounters)]),
    "bhray_get_row_work": (C.c_int, [vp, u32, P(C.c_uint64), u32]),
    "bhray_get_timing": (C.c_int, [vp, P(BhrayTiming)]),
    "bhray_selftest": (C.c_int, [vp, P(C.c_uint64)]),
    "bhray_camera_uniform_update": (None, [P(BhrayCameraUniform), P(C.c_float), P(C.c_float), C.c_float]),
    "bhray_black_hole_default": (None, [P(BhrayBlackHole)]),
    "bhray_black_hole_uniform_update": (None, [P(BhrayBlackHoleUniform), P(BhrayBlackHole)]),
    "bhray_details_default": (None, [P(BhrayDetails)]),
    "bhray_model_new": (C.c_int, [P(vp)]),
    "bhray_model_free": (None, [vp]),
    "bhray_model_add_vertex": (C.c_int, [vp, P(C.c_float)]),
    "bhray_model_add_normal": (C.c_int, [vp, P(C.c_float)]),
    "bhray_model_add_triangle": (C.c_int, [vp, P(BhrayTriangle)]),
    "bhray_model_build_bvh": (C.c_int, [vp]),
    "bhray_model_max_depth": (C.c_int, [vp]),
    "bhray_model_build_bvh_sah": (C.c_int, [vp]),
    "bhray_model_desc_get": (C.c_int, [vp, P(BhrayModelDesc)]),
    "bhray_model_set_transform": (C.c_int, [vp, P(C.c_float), i32]),
    "bhray_model_pack_uniform": (C.c_int, [vp, vp, sz]),
    "bhray_load_model": (C.c_int, [C.c_char_p, P(vp)]),
    "bhray_generate_disk_texture": (C.c_int, [u32, vp]),
}


def declare(L):
    import os
    old_build = os.environ.get("BHRAY_AB_OLD_BUILD") == "1"      # kernel A/B against a library built from an earlier tree (profiles/jobs/ab.sh): symbols it lacks are not bound
    for name, (res, args) in SYMBOLS.items():
        if old_build and not hasattr(L, name):
            continue
        f = getattr(L, name)          # AttributeError if the library does not export it
        f.restype = res
        f.argtypes = args


def check(rc, ctx=None, L=None):
    """Raise BhrayError for a non-zero return code; the message comes from the library the ctx belongs to (L)."""
    if rc != 0:
        if L is None:
            from ._lib import lib
            L = lib()
        msg = L.bhray_last_error(ctx) or b""
        if not msg:
            msg = L.bhray_strerror(rc)
        raise BhrayError(rc, msg.decode("utf-8", "replace"))
