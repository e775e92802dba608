"""ctypes front-end of the CPU oracle (oracle/ray_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (bhusie_amd) never does.  Pinned to the reference's shader text as executed by oracle/wgsl_exec.py
(tests/test_wgsl_pin.py), not to a driver run (the reference has no tests and cannot be built here) — see the header of ray_oracle.c.

The ladder driver below restates src/renderer/mod.rs:170-207 (level sizes r <- 3r-2, each level
reads the previous one, level 0 reads a 1x1 base texture) on top of the per-level entry point.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "ray_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class _Tex(C.Structure):
    _fields_ = [("rgba", C.c_void_p), ("w", C.c_int32), ("h", C.c_int32)]


class _Model(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("visible", C.c_int32),
                ("points", C.c_void_p), ("normals", C.c_void_p), ("triangles", C.c_void_p),
                ("nodes", C.c_void_p), ("bvh_lookup", C.c_void_p),
                ("point_count", C.c_int32), ("normal_count", C.c_int32),
                ("triangle_count", C.c_int32), ("node_count", C.c_int32)]


class _Scene(C.Structure):
    _fields_ = [("camera", C.c_void_p), ("details", C.c_void_p), ("bh", C.c_void_p),
                ("models", C.c_void_p), ("t_temp", _Tex), ("t_disk", _Tex), ("t_sky", _Tex)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("pixels", "copied", "interpolated", "traced", "steps",
                                           "flat_iters", "node_pairs", "triangles", "disk_hits",
                                           "sky_samples")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.oracle_render_level.restype = C.c_int
        L.oracle_render_level.argtypes = [C.POINTER(_Scene), C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                          C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                          C.POINTER(Counters)]
        L.oracle_create_ray.argtypes = [C.POINTER(_Scene), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.oracle_trace_ray.argtypes = [C.POINTER(_Scene), C.c_void_p, C.c_void_p, C.POINTER(Counters)]
        L.oracle_integrate.argtypes = [C.POINTER(_Scene), C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_void_p]
        L.oracle_hit.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        L.oracle_sample.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]
        L.oracle_angle_between.restype = C.c_float
        L.oracle_angle_between.argtypes = [C.c_void_p, C.c_void_p]
        L.bh_acos.restype = C.c_float
        L.bh_acos.argtypes = [C.c_float]
        L.bh_pow_m001.restype = C.c_float
        L.bh_pow_m001.argtypes = [C.c_float]
        L.oracle_sky_resolve.restype = C.c_int
        L.oracle_sky_resolve.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.oracle_f32_to_f16.restype = C.c_uint16
        L.oracle_f32_to_f16.argtypes = [C.c_float]
        L.oracle_set_literal.argtypes = [C.c_int]
        L.oracle_get_literal.restype = C.c_int
        L.oracle_set_eval.argtypes = [C.c_int]
        L.oracle_get_eval.restype = C.c_int
        L.oracle_render_aux.restype = C.c_int
        L.oracle_render_aux.argtypes = [C.POINTER(_Scene), C.c_int, C.c_int, C.c_void_p]
        L.oracle_classify_level.restype = C.c_int
        L.oracle_classify_level.argtypes = [C.POINTER(_Scene), C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.oracle_num_threads.restype = C.c_int
        L.oracle_set_threads.argtypes = [C.c_int]
        _lib = L
    return _lib


def _tex(a: np.ndarray) -> _Tex:
    assert a.dtype == np.uint8 and a.ndim == 3 and a.shape[2] == 4 and a.flags.c_contiguous
    return _Tex(a.ctypes.data, a.shape[1], a.shape[0])


@dataclass
class OracleScene:
    """Everything the shader binds (ray.wgsl:5-19) as raw bytes / arrays.

    camera: 32 B CameraUniform, black_hole: 132 B BlackHoleUniform, details: 32 B RayDetails
    (bytes objects or uint8 arrays); textures: (H,W,4) uint8; models: list of dicts with
    position(3 f32), visible, points(N,4) f32, normals(M,4) f32, triangles(T,6) i32,
    nodes (structured, 32 B each; any array of K*32 bytes), bvh_lookup(T) i32.
    """
    camera: bytes
    black_hole: bytes
    details: bytes
    t_temp: np.ndarray
    t_disk: np.ndarray
    t_sky: np.ndarray
    models: list = field(default_factory=list)

    def _pack(self):
        keep = []
        cam = np.frombuffer(bytes(self.camera), dtype=np.uint8).copy(); assert cam.size == 32
        bh = np.frombuffer(bytes(self.black_hole), dtype=np.uint8).copy(); assert bh.size == 132
        det = np.frombuffer(bytes(self.details), dtype=np.uint8).copy(); assert det.size == 32
        keep += [cam, bh, det]
        ms = (_Model * max(1, len(self.models)))()
        for i, m in enumerate(self.models):
            pts = np.ascontiguousarray(m["points"], dtype=np.float32).reshape(-1, 4)
            nrm = np.ascontiguousarray(m["normals"], dtype=np.float32).reshape(-1, 4)
            tri = np.ascontiguousarray(m["triangles"], dtype=np.int32).reshape(-1, 6)
            nodes = np.ascontiguousarray(m["nodes"]).view(np.uint8).reshape(-1, 32)
            lut = np.ascontiguousarray(m["bvh_lookup"], dtype=np.int32)
            keep += [pts, nrm, tri, nodes, lut]
            ms[i].position[:] = [float(x) for x in m["position"]]
            ms[i].visible = int(m.get("visible", 1))
            ms[i].points = pts.ctypes.data; ms[i].normals = nrm.ctypes.data
            ms[i].triangles = tri.ctypes.data; ms[i].nodes = nodes.ctypes.data
            ms[i].bvh_lookup = lut.ctypes.data
            ms[i].point_count = pts.shape[0]; ms[i].normal_count = nrm.shape[0]
            ms[i].triangle_count = tri.shape[0]; ms[i].node_count = nodes.shape[0]
        keep.append(ms)
        s = _Scene(cam.ctypes.data, det.ctypes.data, bh.ctypes.data, C.addressof(ms),
                   _tex(self.t_temp), _tex(self.t_disk), _tex(self.t_sky))
        return s, keep


def render_level(scene: OracleScene, size, prev: np.ndarray | None = None, counters: Counters | None = None,
                 rows=None) -> np.ndarray:
    """One ladder level (ray.wgsl main over every pixel).  size=(W,H); prev = previous level
    image (H',W',4) f32 or None for the base case (1x1 t_prev).  rows=(y0,y1) optionally restricts."""
    w, h = size
    s, keep = scene._pack()
    out = np.full((h, w, 4), np.nan, dtype=np.float32)
    if prev is None:
        pbuf, pw, ph = None, 1, 1
    else:
        prev = np.ascontiguousarray(prev, dtype=np.float32)
        pbuf, pw, ph = prev.ctypes.data, prev.shape[1], prev.shape[0]
    y0, y1 = rows if rows else (0, h)
    rc = lib().oracle_render_level(C.byref(s), w, h, pbuf, pw, ph, out.ctypes.data, 0, y0, w, y1, None,
                                   C.byref(counters) if counters is not None else None)
    assert rc == 0
    del keep
    return out


def ladder_sizes(base, multiplier: int = 3, levels: int = 4):
    """mod.rs:177-205: current_res <- current_res*m - (m-1)."""
    w, h = base
    out = []
    for i in range(levels):
        out.append((int(w), int(h)))
        w, h = w * multiplier - (multiplier - 1), h * multiplier - (multiplier - 1)
    return out


def render_ladder(scene: OracleScene, sizes, counters: Counters | None = None):
    """All levels in order; returns the list of level images."""
    imgs, prev = [], None
    for sz in sizes:
        img = render_level(scene, sz, prev, counters)
        imgs.append(img)
        prev = img
    return imgs


def trace_ray(scene: OracleScene, ray6, counters: Counters | None = None) -> np.ndarray:
    s, keep = scene._pack()
    r = np.ascontiguousarray(ray6, dtype=np.float32)
    out = np.zeros(4, dtype=np.float32)
    lib().oracle_trace_ray(C.byref(s), r.ctypes.data, out.ctypes.data,
                           C.byref(counters) if counters is not None else None)
    return out


def create_ray(scene: OracleScene, px, py, sw, sh) -> np.ndarray:
    s, keep = scene._pack()
    out = np.zeros(6, dtype=np.float32)
    lib().oracle_create_ray(C.byref(s), px, py, sw, sh, out.ctypes.data)
    return out


def integrate(scene: OracleScene, ray6, h: float, method: int, n: int) -> np.ndarray:
    s, keep = scene._pack()
    r = np.ascontiguousarray(ray6, dtype=np.float32)
    out = np.zeros((n, 8), dtype=np.float32)
    lib().oracle_integrate(C.byref(s), r.ctypes.data, h, method, n, out.ctypes.data)
    return out


def hit(kind: int, ray6, params, t_min: float, t_max: float) -> dict:
    r = np.ascontiguousarray(ray6, dtype=np.float32)
    p = np.ascontiguousarray(params, dtype=np.float32)
    out = np.zeros(9, dtype=np.float32)
    lib().oracle_hit(kind, r.ctypes.data, p.ctypes.data, t_min, t_max, out.ctypes.data)
    return {"color": out[0:3].copy(), "opacity": float(out[3]), "t": float(out[4]),
            "normal": out[5:8].copy(), "hit": bool(out[8])}


def sample(tex: np.ndarray, u: float, v: float) -> np.ndarray:
    out = np.zeros(4, dtype=np.float32)
    lib().oracle_sample(tex.ctypes.data, tex.shape[1], tex.shape[0], u, v, out.ctypes.data)
    return out


def acos(x: float) -> float:
    return float(lib().bh_acos(x))


def pow_m001(x: float) -> float:
    return float(lib().bh_pow_m001(x))


def num_threads() -> int:
    return int(lib().oracle_num_threads())


def set_threads(n: int) -> None:
    lib().oracle_set_threads(n)


def sky_resolve(prev: np.ndarray, t_sky: np.ndarray) -> np.ndarray:
    """sky.wgsl over a whole RGBA32F image -> (H, W, 4) float16 (the reference target is rgba16float)."""
    prev = np.ascontiguousarray(prev, dtype=np.float32)
    h, w = prev.shape[:2]
    out = np.zeros((h, w, 4), dtype=np.uint16)
    rc = lib().oracle_sky_resolve(prev.ctypes.data, w, h, t_sky.ctypes.data, t_sky.shape[1], t_sky.shape[0], out.ctypes.data)
    assert rc == 0
    return out.view(np.float16)


def set_literal(on: bool) -> None:
    """Integrator evaluation: False = the numerics contract (N3, N7, N9; what the HIP kernel computes), True = the shader text
    operator by operator (N0-N2).  Process-wide switch of the C oracle; tests restore it."""
    lib().oracle_set_literal(1 if on else 0)


EVAL_CONTRACT, EVAL_LITERAL, EVAL_FMA = 0, 1, 2


def set_eval(mode: int) -> None:
    """0 = the numerics contract, 1 = the literal text, 2 = the literal text with fused multiply-add contraction only (no
    reassociation): the THIRD legal evaluation (kernel flag BHRAY_F_EVAL_FMA)."""
    lib().oracle_set_eval(int(mode))


def render_aux(scene: OracleScene, size) -> np.ndarray:
    """Every pixel of a (W, H) frame traced from the camera: (H, W, 4) = closest approach to the hole, iterations, smallest
    distance of a disk-plane crossing from the disk's rims (1e30 = never crossed), disk hits.  Diagnostics for the parity tests."""
    w, h = size
    s, keep = scene._pack()
    out = np.zeros((h, w, 4), dtype=np.float32)
    assert lib().oracle_render_aux(C.byref(s), w, h, out.ctypes.data) == 0
    del keep
    return out


def classify_level(scene: OracleScene, size, prev: np.ndarray) -> np.ndarray:
    """The grid decision alone for every pixel of a (W, H) level: 0 copy, 1 interpolate, 2 trace (diagnostics)."""
    w, h = size
    s, keep = scene._pack()
    prev = np.ascontiguousarray(prev, dtype=np.float32)
    out = np.zeros((h, w), dtype=np.uint8)
    assert lib().oracle_classify_level(C.byref(s), w, h, prev.ctypes.data, prev.shape[1], prev.shape[0], out.ctypes.data) == 0
    del keep
    return out
