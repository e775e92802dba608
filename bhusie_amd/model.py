"""Model / load_model mirrors (reference: src/renderer/triangle.rs:65-259, src/renderer/model.rs:7-87).
The builder and the OBJ reader are C++ behind the C ABI; this wraps the handle."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import lib
from .layouts import MODEL_UNIFORM_BYTES, BhrayModelDesc, BhrayTriangle, check

NODE_DTYPE = np.dtype([("min_corner", "<f4", 3), ("left_child", "<i4"), ("max_corner", "<f4", 3), ("obj_count", "<i4")])


class Model:
    def __init__(self, handle=None):
        if handle is None:
            h = C.c_void_p()
            check(lib().bhray_model_new(C.byref(h)))
            handle = h
        self._h = handle

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                lib().bhray_model_free(h)
            except Exception:
                pass

    def add_vertex(self, p):
        check(lib().bhray_model_add_vertex(self._h, (C.c_float * 4)(*[float(x) for x in (list(p) + [0.0])[:4]])))

    def add_normal(self, n):
        check(lib().bhray_model_add_normal(self._h, (C.c_float * 4)(*[float(x) for x in (list(n) + [0.0])[:4]])))

    def add_triangle(self, t):
        check(lib().bhray_model_add_triangle(self._h, C.byref(BhrayTriangle(*[int(x) for x in t]))))

    def build_bvh(self):
        check(lib().bhray_model_build_bvh(self._h))

    def build_bvh_sah(self):
        """Binned-SAH builder behind a flag (not the reference's tree; see include/bhray.h)."""
        check(lib().bhray_model_build_bvh_sah(self._h))

    def max_depth(self) -> int:
        return int(lib().bhray_model_max_depth(self._h))

    def set_transform(self, position, visible=1):
        check(lib().bhray_model_set_transform(self._h, (C.c_float * 3)(*[float(x) for x in position]), int(visible)))

    def desc(self) -> BhrayModelDesc:
        d = BhrayModelDesc()
        check(lib().bhray_model_desc_get(self._h, C.byref(d)))
        return d

    def arrays(self) -> dict:
        """Copies of the arrays (for comparisons / feeding the oracle in tests)."""
        d = self.desc()

        def arr(ptr, n, dt, shape):
            if n == 0:
                return np.zeros((0,) + shape[1:], dtype=dt)
            buf = (C.c_uint8 * (n * np.dtype(dt).itemsize * int(np.prod(shape[1:]) or 1))).from_address(ptr)
            return np.frombuffer(buf, dtype=dt).reshape((n,) + shape[1:]).copy()

        return dict(position=np.array(d.position[:], dtype=np.float32), visible=int(d.visible),
                    points=arr(d.points, d.point_count, np.float32, (0, 4)),
                    normals=arr(d.normals, d.normal_count, np.float32, (0, 4)),
                    triangles=arr(d.triangles, d.triangle_count, np.int32, (0, 6)),
                    nodes=arr(d.nodes, d.node_count, NODE_DTYPE, (0,)),
                    bvh_lookup=arr(d.bvh_lookup, d.triangle_count, np.int32, (0,)))

    def pack_uniform(self) -> bytes:
        buf = C.create_string_buffer(MODEL_UNIFORM_BYTES)
        check(lib().bhray_model_pack_uniform(self._h, buf, MODEL_UNIFORM_BYTES))
        return buf.raw


def load_model(path: str) -> Model:
    h = C.c_void_p()
    check(lib().bhray_load_model(path.encode(), C.byref(h)))
    return Model(h)
