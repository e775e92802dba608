"""Python restatement of the reference HOST code on the ray-pass path.  TEST INFRASTRUCTURE ONLY.

What is restated (binary32 arithmetic via numpy.float32 scalars, operation order as written):
  * CameraUniform::update            /root/reference/src/scene/camera.rs:66-90
  * BlackHole::new + BlackHoleUniform::update   src/scene/blackhole.rs:16-28, 68-98
      third-party math (cgmath 0.18.0, Cargo.lock:632 — not vendored under /root/reference):
      Quaternion::from(Euler) (XYZ order, NASA 19770024290 p. A-2), Quaternion * Vector3
      (v + 2 q.v x (q.v x v + s v)), InnerSpace::normalize (v * (1/|v|)).  PARITY UNPINNED:
      restated from the published algorithm; no reference test pins these bytes.
  * RayDetails defaults              src/renderer/mod.rs:116-121
  * Model::build_bvh / update_bounds / subdivide   src/renderer/triangle.rs:143-259
  * load_model                       src/renderer/model.rs:7-87  (tobj 4.0.2 default options,
      Cargo.lock:2951: no triangulation, separate position / normal index streams)
  * ModelUniform::update byte image  src/renderer/triangle.rs:268-325
  * the disk-texture generator       perlin/src/main.rs:1-148 — the one place where the reference
      itself supplies a checkable artefact: its output src/renderer/textures/disk.png

Only tests/ (and fixture generators) import this; the product never does.
"""
from __future__ import annotations

import math
import struct
import sys

import numpy as np

f32 = np.float32
MAX_MODEL_VERTICES = 524288
MODEL_UNIFORM_BYTES = 48 + 92 * MAX_MODEL_VERTICES + 28


def _v(x, y, z):
    return np.array([x, y, z], dtype=np.float32)


def _cross(a, b):
    return _v(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def _dot(a, b):  # cgmath Vector3::dot = x*x + y*y + z*z (left to right)
    return f32(f32(a[0] * b[0] + a[1] * b[1]) + a[2] * b[2])


def _normalize(a):
    mag = f32(np.sqrt(_dot(a, a)))
    return a * f32(f32(1.0) / mag)


# ---------------------------------------------------------------- camera.rs:66-90
def camera_uniform(position=(0.0, 0.0, -19.0), forward=(0.0, 0.0, 1.0), fov=1.0) -> bytes:
    """Defaults are Camera::new (camera.rs:10-16)."""
    return struct.pack("<3fI3ff", *[float(f32(x)) for x in position], 0,
                       *[float(f32(x)) for x in forward], float(f32(fov)))


# ---------------------------------------------------------------- mod.rs:116-121 + ray_pipeline.rs:3-14
def ray_details(material_count=0, model_count=0, time=0.0, integration_method=0, step_size=0.15,
                max_iterations=2000, angle_division_threshold=0.02, highlight_interpolation=0) -> bytes:
    return struct.pack("<iifififi", material_count, model_count, float(f32(time)), integration_method,
                       float(f32(step_size)), max_iterations, float(f32(angle_division_threshold)),
                       highlight_interpolation)


# ---------------------------------------------------------------- blackhole.rs
BLACK_HOLE_DEFAULT = dict(position=(0.0, 0.0, 0.0), accretion_disk_rotation=(0.15, 0.0, 0.25),
                          accretion_disk_inner=2.0, accretion_disk_outer=10.0, rotation_speed=1.0,
                          relativity_sphere_radius=20.0, show_disk_texture=1, show_red_shift=1,
                          feather_amount=0.3)


def _quat_from_euler(x, y, z):
    half = f32(0.5)
    sx, cx = f32(np.sin(f32(x) * half)), f32(np.cos(f32(x) * half))
    sy, cy = f32(np.sin(f32(y) * half)), f32(np.cos(f32(y) * half))
    sz, cz = f32(np.sin(f32(z) * half)), f32(np.cos(f32(z) * half))
    s = f32(f32(f32(-sx) * sy) * sz) + f32(f32(cx * cy) * cz)
    vx = f32(f32(sx * cy) * cz) + f32(f32(sy * sz) * cx)
    vy = f32(f32(f32(-sx) * sz) * cy) + f32(f32(sy * cx) * cz)
    vz = f32(f32(sx * sy) * cz) + f32(f32(sz * cx) * cy)
    return f32(s), _v(vx, vy, vz)


def _quat_rotate(s, v, vec):
    tmp = _cross(v, vec) + vec * s
    return _cross(v, tmp) * f32(2.0) + vec


def black_hole_uniform(**kw) -> bytes:
    bh = dict(BLACK_HOLE_DEFAULT); bh.update(kw)
    rx, ry, rz = bh["accretion_disk_rotation"]
    s, v = _quat_from_euler(rx, ry, rz)
    up = _normalize(_quat_rotate(s, v, _v(0.0, -1.0, 0.0)))
    right = _cross(_v(0.0, 0.0, 1.0), up)
    fwd = _cross(right, up)
    mat = [right[0], right[1], right[2], 0.0, up[0], up[1], up[2], 0.0, fwd[0], fwd[1], fwd[2], 0.0]
    return struct.pack("<4f3fi3fi12ff8i",
                       float(f32(bh["accretion_disk_inner"])), float(f32(bh["accretion_disk_outer"])),
                       float(f32(bh["rotation_speed"])), float(f32(bh["relativity_sphere_radius"])),
                       *[float(f32(x)) for x in bh["position"]], int(bh["show_disk_texture"]),
                       *[float(x) for x in up], int(bh["show_red_shift"]),
                       *[float(x) for x in mat], float(f32(bh["feather_amount"])), *([0] * 8))


# ---------------------------------------------------------------- triangle.rs:143-259
NODE_DTYPE = np.dtype([("min_corner", "<f4", 3), ("left_child", "<i4"), ("max_corner", "<f4", 3),
                       ("obj_count", "<i4")])
assert NODE_DTYPE.itemsize == 32


class Model:
    """triangle.rs:65-141 with growable storage (capacity limits checked at pack time)."""

    def __init__(self):
        self.position = _v(-10.0, 0.0, 30.0)          # triangle.rs:100
        self.rotation = _v(0.0, 0.0, 0.0)
        self.visible = 1
        self.points: list = []
        self.normals: list = []
        self.triangles: list = []
        self.nodes = None
        self.bvh_lookup = None
        self.nodes_used = 0

    def add_vertex(self, p):
        self.points.append([f32(p[0]), f32(p[1]), f32(p[2]), f32(p[3] if len(p) > 3 else 0.0)])

    def add_normal(self, n):
        self.normals.append([f32(n[0]), f32(n[1]), f32(n[2]), f32(n[3] if len(n) > 3 else 0.0)])

    def add_triangle(self, t):
        self.triangles.append([int(x) for x in t])

    # -- build_bvh (triangle.rs:143-157)
    def build_bvh(self):
        T = len(self.triangles)
        P = np.array(self.points, dtype=np.float32).reshape(-1, 4)
        tri = np.array(self.triangles, dtype=np.int32).reshape(-1, 6)
        self._P, self._tri = P, tri
        self.bvh_lookup = np.arange(T, dtype=np.int32)
        self.nodes = np.zeros(max(1, 2 * T + 1), dtype=NODE_DTYPE)
        self.nodes[0]["left_child"] = 0
        self.nodes[0]["obj_count"] = T
        self.nodes_used = 1
        # centroids exactly as subdivide computes them: ((a + b) + c) / 3.0 in f32
        a, b, c = P[tri[:, 0], :3], P[tri[:, 1], :3], P[tri[:, 2], :3]
        self._cent = ((a + b) + c) / f32(3.0) if T else np.zeros((0, 3), np.float32)
        self._update_bounds(0)
        old = sys.getrecursionlimit()
        sys.setrecursionlimit(max(old, 10000))       # the reference grows its stack to 1 GiB (main.rs:1-5)
        try:
            self._subdivide(0)
        finally:
            sys.setrecursionlimit(old)
        self.nodes = self.nodes[: self.nodes_used].copy()

    def _update_bounds(self, ni):                    # triangle.rs:159-194
        n = self.nodes[ni]
        lo = np.full(3, np.finfo(np.float32).max, dtype=np.float32)
        hi = np.full(3, np.finfo(np.float32).min, dtype=np.float32)   # f32::MIN = -MAX
        first, cnt = int(n["left_child"]), int(n["obj_count"])
        if cnt:
            idx = self.bvh_lookup[first:first + cnt]
            pts = self._P[self._tri[idx, :3].reshape(-1), :3]
            lo = np.minimum(lo, pts.min(axis=0)); hi = np.maximum(hi, pts.max(axis=0))
        # zero bounds canonicalised to +0 (f32::min/max leave the sign of a zero result unspecified)
        self.nodes[ni]["min_corner"] = lo + f32(0.0)
        self.nodes[ni]["max_corner"] = hi + f32(0.0)

    def _subdivide(self, ni):                        # triangle.rs:196-259
        n = self.nodes[ni]
        cnt = int(n["obj_count"])
        if cnt <= 2:
            return
        lo = n["min_corner"].astype(np.float32); hi = n["max_corner"].astype(np.float32)
        extent = hi - lo
        axis = 0
        if extent[1] > extent[axis]:
            axis = 1
        if extent[2] > extent[axis]:
            axis = 2
        split = f32(lo[axis] + extent[axis] / f32(2.0))
        first = int(n["left_child"])
        i, j = first, first + cnt - 1
        L = self.bvh_lookup
        while i <= j:
            if self._cent[L[i], axis] < split:
                i += 1
            else:
                L[i], L[j] = L[j], L[i]
                j -= 1
        left_count = i - first
        if left_count == 0 or left_count == cnt:
            return
        li = self.nodes_used; ri = li + 1
        self.nodes_used += 2
        self.nodes[li]["left_child"] = first; self.nodes[li]["obj_count"] = left_count
        self.nodes[ri]["left_child"] = i; self.nodes[ri]["obj_count"] = cnt - left_count
        self.nodes[ni]["left_child"] = li; self.nodes[ni]["obj_count"] = 0
        self._update_bounds(li); self._update_bounds(ri)
        self._subdivide(li); self._subdivide(ri)

    # -- views
    def as_oracle_dict(self):
        return dict(position=self.position, visible=self.visible,
                    points=np.array(self.points, dtype=np.float32).reshape(-1, 4),
                    normals=np.array(self.normals, dtype=np.float32).reshape(-1, 4),
                    triangles=np.array(self.triangles, dtype=np.int32).reshape(-1, 6),
                    nodes=self.nodes, bvh_lookup=self.bvh_lookup)

    def max_depth(self):
        depth, stack = 0, [(0, 1)]
        while stack:
            ni, d = stack.pop()
            depth = max(depth, d)
            if int(self.nodes[ni]["obj_count"]) == 0 and self.nodes_used > 1:
                lc = int(self.nodes[ni]["left_child"])
                stack += [(lc, d + 1), (lc + 1, d + 1)]
        return depth

    def pack_uniform(self) -> bytes:                 # triangle.rs:268-325
        N = MAX_MODEL_VERTICES
        buf = bytearray(MODEL_UNIFORM_BYTES)
        # normal_count is never copied by ModelUniform::update (triangle.rs:308-325) -> stays 0
        struct.pack_into("<3fi3fIiiiI", buf, 0, *[float(x) for x in self.position], int(self.visible),
                         *[float(x) for x in self.rotation], 0, len(self.points), 0, len(self.triangles), 0)
        d = self.as_oracle_dict()
        for arr, off in ((d["points"], 48), (d["normals"], 48 + 16 * N), (d["triangles"], 48 + 32 * N),
                         (d["nodes"], 48 + 56 * N), (d["bvh_lookup"], 48 + 88 * N)):
            raw = np.ascontiguousarray(arr).tobytes()
            buf[off:off + len(raw)] = raw
        return bytes(buf)


# ---------------------------------------------------------------- model.rs:7-87
def parse_obj(text: str):
    """Minimal tobj-default-options equivalent: one mesh, `v`, `vn`, `f` with 3 vertices
    (a, a/t, a//n, a/t/n; negative indices relative).  Returns (positions, normals, idx, nidx)."""
    pos, nrm, idx, nidx = [], [], [], []
    for line in text.splitlines():
        t = line.split("#", 1)[0].split()
        if not t:
            continue
        if t[0] == "v":
            pos.append([f32(float(t[1])), f32(float(t[2])), f32(float(t[3]))])
        elif t[0] == "vn":
            nrm.append([f32(float(t[1])), f32(float(t[2])), f32(float(t[3]))])
        elif t[0] == "f":
            if len(t) != 4:
                raise ValueError("only triangles (tobj default options do not triangulate)")
            for w in t[1:]:
                parts = w.split("/")
                a = int(parts[0]); a = a - 1 if a > 0 else len(pos) + a
                idx.append(a)
                if len(parts) >= 3 and parts[2] != "":
                    b = int(parts[2]); b = b - 1 if b > 0 else len(nrm) + b
                    nidx.append(b)
    return pos, nrm, idx, nidx


def load_model(text: str) -> Model:
    pos, nrm, idx, nidx = parse_obj(text)
    m = Model()
    mesh_offset, normal_offset = 0, 0                # single object: offsets are 0 (model.rs:22-23)
    for n in nrm:
        m.add_normal([n[0], n[1], n[2], 0.0])
    for p in pos:
        m.add_vertex([f32(p[0] * f32(0.5)), f32(p[1] * f32(-0.5)), f32(p[2] * f32(0.5)), 0.0])
    for i in range(len(idx) // 3):
        p1, p2, p3 = idx[3 * i:3 * i + 3]
        if nidx:
            n1, n2, n3 = nidx[3 * i:3 * i + 3]
        else:
            a = np.array(m.points[p1][:3], dtype=np.float32); b = np.array(m.points[p2][:3], dtype=np.float32)
            c = np.array(m.points[p3][:3], dtype=np.float32)
            d = _normalize(_cross(b - a, c - a))
            n1 = n2 = n3 = len(m.normals)
            m.add_normal([d[0], d[1], d[2], 0.0])
        m.add_triangle([p1 + mesh_offset, p2 + mesh_offset, p3 + mesh_offset,
                        n1 + normal_offset, n2 + normal_offset, n3 + normal_offset])
    m.build_bvh()
    return m


# ---------------------------------------------------------------- perlin/src/main.rs (offline asset tool)
def _random_gradient(ix, iy):
    """perlin/src/main.rs:6-24: integer hash (wrapping_mul, rotate_left by 16) -> angle in [0, pi] -> (cos, sin)."""
    a = ix.astype(np.uint32); b = iy.astype(np.uint32)
    a = a * np.uint32(3284157443)
    b = b ^ ((a << np.uint32(16)) | (a >> np.uint32(16)))
    b = b * np.uint32(1911520717)
    a = a ^ ((b << np.uint32(16)) | (b >> np.uint32(16)))
    a = a * np.uint32(2048419325)
    k = f32(f32(np.pi) / f32(4294967295))            # PI / (!(0u32 >> 1)) as f32   (u32::MAX as f32 = 2^32)
    rnd = a.astype(np.float32) * k
    return np.cos(rnd).astype(np.float32), np.sin(rnd).astype(np.float32)


def _dot_grid_gradient(ix, iy, x, y):                # main.rs:26-33
    gx, gy = _random_gradient(ix, iy)
    dx = x - ix.astype(np.float32); dy = y - iy.astype(np.float32)
    return dx * gx + dy * gy


def _interpolate(a0, a1, w):                         # main.rs:35-38
    return (a1 - a0) * ((w * (w * f32(6.0) - f32(15.0)) + f32(10.0)) * w * w * w) + a0


def _perlin(x, y):                                   # main.rs:40-58
    x0 = np.floor(x).astype(np.uint32); y0 = np.floor(y).astype(np.uint32)
    x1 = x0 + np.uint32(1); y1 = y0 + np.uint32(1)
    sx = x - x0.astype(np.float32); sy = y - y0.astype(np.float32)
    ix0 = _interpolate(_dot_grid_gradient(x0, y0, x, y), _dot_grid_gradient(x1, y0, x, y), sx)
    ix1 = _interpolate(_dot_grid_gradient(x0, y1, x, y), _dot_grid_gradient(x1, y1, x, y), sx)
    return _interpolate(ix0, ix1, sy) * f32(0.5) + f32(0.5)


def _as_u8(v):                                       # Rust `as u8` from f32: truncation, saturating, NaN -> 0
    v = np.nan_to_num(v, nan=0.0)
    return np.clip(np.trunc(v), 0, 255).astype(np.uint8)


def _generate(w, h, density):                        # main.rs:61-77 (value[x, y]; image row = y)
    d = f32(f32(density) / f32(w))
    xs, ys = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32), indexing="ij")
    return _as_u8(_perlin(xs * d, ys * d) * f32(256.0))            # indexed [x, y]


def _spiral(buf, amount, power):                     # main.rs:79-110
    w, h = buf.shape
    xs, ys = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32), indexing="ij")
    rx = (xs / f32(w)) * f32(2.0) - f32(1.0); ry = (ys / f32(h)) * f32(2.0) - f32(1.0)
    r = np.sqrt(rx * rx + ry * ry).astype(np.float32)
    theta = np.arctan2(ry, rx).astype(np.float32)
    PI = f32(np.pi)
    t = theta + PI + np.power(r, f32(power)).astype(np.float32) * PI * f32(amount)
    theta = np.fmod(t, f32(2.0) * PI).astype(np.float32) - PI        # Rust % on f32 = fmod
    rx = r * np.cos(theta).astype(np.float32); ry = r * np.sin(theta).astype(np.float32)
    nx = _as_u32((rx * f32(0.5) + f32(0.5)) * f32(w)) % np.uint32(w)
    ny = _as_u32((ry * f32(0.5) + f32(0.5)) * f32(h)) % np.uint32(h)
    return buf[nx, ny]


def _as_u32(v):
    v = np.nan_to_num(v, nan=0.0)
    return np.clip(np.trunc(v), 0, 4294967295).astype(np.uint32)


def _merge(b1, b2, amount):                          # main.rs:113-131
    a = f32(amount)
    return _as_u8(b1.astype(np.float32) * a + b2.astype(np.float32) * (f32(1.0) - a))


def perlin_disk(size: int = 1000) -> np.ndarray:
    """main(): four noise octaves (density 4, 20, 50, 100), each spiral-warped (amount 2, power 0.5), merged pairwise at 0.5
    (main.rs:133-147).  Returns (size, size, 4) uint8 with the value replicated into R, G, B and A, image row = y."""
    sp = [_spiral(_generate(size, size, d), 2.0, 0.5) for d in (4, 20, 50, 100)]
    m = _merge(_merge(_merge(sp[3], sp[2], 0.5), sp[1], 0.5), sp[0], 0.5)
    img = m.T                                        # buffer.put_pixel(x, y) -> row y, column x
    return np.ascontiguousarray(np.repeat(img[:, :, None], 4, axis=2))
