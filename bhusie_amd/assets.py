"""Seeded synthetic inputs for the ray pass.

The reference's textures are baked in with include_bytes!
(/root/reference/src/renderer/pipelines/ray_pipeline.rs:63-70); sky.png and lucy.obj are missing
from the reference checkout (.MISSING_LARGE_BLOBS), and reference assets are not copied here.
These generators produce inputs of the same kind and shape: an RGBA8 temperature
LUT (256x256, like color.png), a disk texture whose four channels carry the same noise value (what
perlin/src/main.rs writes), an equirectangular star field, and a triangulated OBJ mesh.

Everything is integer-hash based (no RNG state), so the bytes are identical on every machine.
"""
from __future__ import annotations

import numpy as np


def _hash_u32(x: np.ndarray) -> np.ndarray:
    """32-bit avalanche hash (lowbias32) on uint32 arrays."""
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


def _lattice(ix, iy, seed):
    h = _hash_u32(ix.astype(np.uint32) * np.uint32(0x9E3779B1) ^ _hash_u32(iy.astype(np.uint32) + np.uint32(seed * 7919 + 13)))
    return (h >> np.uint32(8)).astype(np.float64) / float(1 << 24)


def value_noise(w: int, h: int, cells: int, seed: int) -> np.ndarray:
    """Smooth value noise in [0,1], `cells` lattice cells across the image."""
    ys, xs = np.mgrid[0:h, 0:w]
    fx = xs * (cells / w); fy = ys * (cells / h)
    x0 = np.floor(fx).astype(np.int64); y0 = np.floor(fy).astype(np.int64)
    tx = fx - x0; ty = fy - y0
    tx = tx * tx * (3 - 2 * tx); ty = ty * ty * (3 - 2 * ty)
    v00 = _lattice(x0, y0, seed); v10 = _lattice(x0 + 1, y0, seed)
    v01 = _lattice(x0, y0 + 1, seed); v11 = _lattice(x0 + 1, y0 + 1, seed)
    return (v00 * (1 - tx) + v10 * tx) * (1 - ty) + (v01 * (1 - tx) + v11 * tx) * ty


def disk_texture(size: int = 1000, seed: int = 1) -> np.ndarray:
    """(size,size,4) uint8; all four channels equal (perlin/src/main.rs writes the value into RGBA)."""
    v = 0.5 * value_noise(size, size, 8, seed) + 0.3 * value_noise(size, size, 32, seed + 1) \
        + 0.2 * value_noise(size, size, 128, seed + 2)
    v = np.clip((v - 0.15) / 0.7, 0.0, 1.0)
    b = np.round(v * 255.0).astype(np.uint8)
    return np.ascontiguousarray(np.repeat(b[:, :, None], 4, axis=2))


def temp_lut(size: int = 256) -> np.ndarray:
    """(size,size,4) uint8 ramp along u: deep red -> orange -> white -> pale blue (a colour-
    temperature palette like color.png); constant along v; alpha 255."""
    u = (np.arange(size) + 0.5) / size
    r = np.clip(0.35 + 1.6 * u, 0, 1) * np.clip(1.9 - 1.0 * u, 0, 1)
    g = np.clip(-0.05 + 1.5 * u, 0, 1) * np.clip(1.8 - 0.9 * u, 0, 1)
    b = np.clip(-0.5 + 2.0 * u, 0, 1)
    row = np.stack([r, g, b, np.ones_like(u)], axis=1)
    img = np.broadcast_to(row[None, :, :], (size, size, 4))
    return np.ascontiguousarray(np.round(img * 255.0).astype(np.uint8))


def sky_texture(w: int = 4096, h: int = 2048, seed: int = 2) -> np.ndarray:
    """(h,w,4) uint8 equirect star field: ~0.1 % bright texels over a faint noisy band."""
    ys, xs = np.mgrid[0:h, 0:w]
    hsh = _hash_u32(xs.astype(np.uint32) * np.uint32(73856093) ^ ys.astype(np.uint32) * np.uint32(19349663)
                    ^ np.uint32(seed * 83492791))
    star = (hsh % np.uint32(1000)) == 0
    mag = ((hsh >> np.uint32(10)) & np.uint32(0xFF)).astype(np.float64) / 255.0
    tint = ((hsh >> np.uint32(18)) & np.uint32(0x3F)).astype(np.float64) / 63.0
    band = np.exp(-((ys / h - 0.5) / 0.12) ** 2) * value_noise(w, h, 16, seed + 5)
    base = 0.25 + 0.35 * band
    img = np.zeros((h, w, 4), dtype=np.float64)
    img[..., 0] = base; img[..., 1] = base * 0.95; img[..., 2] = base * 1.05
    s = 0.6 + 0.4 * mag
    img[..., 0] = np.where(star, s * (0.85 + 0.15 * tint), img[..., 0])
    img[..., 1] = np.where(star, s * 0.9, img[..., 1])
    img[..., 2] = np.where(star, s * (1.0 - 0.15 * tint), img[..., 2])
    img[..., 3] = 1.0
    return np.ascontiguousarray(np.round(np.clip(img, 0, 1) * 255.0).astype(np.uint8))


def sphere_mesh_obj(n_lat: int = 24, n_lon: int = 32, radius: float = 8.0, bump: float = 0.15,
                    seed: int = 3, with_normals: bool = True) -> str:
    """OBJ text of a lat-long sphere with seeded radial noise; 2*n_lon*(n_lat-1) triangles.
    (n_lat=320, n_lon=320 gives 204 160 triangles — the bench mesh.)"""
    lat = np.linspace(0.0, np.pi, n_lat + 1)
    lon = np.linspace(0.0, 2 * np.pi, n_lon, endpoint=False)
    verts = [(0.0, radius, 0.0)]
    ii, jj = np.mgrid[1:n_lat, 0:n_lon]
    hv = _hash_u32(ii.astype(np.uint32) * np.uint32(2654435761) ^ jj.astype(np.uint32) * np.uint32(40503) ^ np.uint32(seed))
    rr = radius * (1.0 + bump * ((hv & np.uint32(0xFFFF)).astype(np.float64) / 65535.0 - 0.5))
    x = rr * np.sin(lat[ii]) * np.cos(lon[jj]); y = rr * np.cos(lat[ii]); z = rr * np.sin(lat[ii]) * np.sin(lon[jj])
    verts += list(zip(x.ravel().tolist(), y.ravel().tolist(), z.ravel().tolist()))
    verts.append((0.0, -radius, 0.0))
    V = np.array(verts)
    lines = ["# synthetic seeded sphere", "o sphere"]
    lines += ["v %.6f %.6f %.6f" % tuple(v) for v in V]
    if with_normals:
        N = V / np.linalg.norm(V, axis=1, keepdims=True)
        lines += ["vn %.6f %.6f %.6f" % tuple(n) for n in N]

    def vid(i, j):  # 1-based OBJ index of ring i (1..n_lat-1), column j
        return 2 + (i - 1) * n_lon + (j % n_lon)

    south = len(V)
    faces = []
    for j in range(n_lon):
        faces.append((1, vid(1, j + 1), vid(1, j)))
        faces.append((south, vid(n_lat - 1, j), vid(n_lat - 1, j + 1)))
    for i in range(1, n_lat - 1):
        for j in range(n_lon):
            a, b, c, d = vid(i, j), vid(i, j + 1), vid(i + 1, j), vid(i + 1, j + 1)
            faces.append((a, b, d)); faces.append((a, d, c))
    if with_normals:
        lines += ["f %d//%d %d//%d %d//%d" % (a, a, b, b, c, c) for a, b, c in faces]
    else:
        lines += ["f %d %d %d" % f for f in faces]
    return "\n".join(lines) + "\n"


def icosphere_mesh_obj(level: int = 6, radius: float = 8.0, bump: float = 0.15, seed: int = 3, with_normals: bool = True) -> str:
    """OBJ text of a subdivided icosahedron (20 * 4**level triangles, near-uniform like a scanned mesh) with seeded
    radial noise.  level 6 = 81 920 triangles, level 7 = 327 680 (within the reference's 524 288-entry capacities)."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    V = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    V /= np.linalg.norm(V, axis=1, keepdims=True)
    F = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
                  [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    for _ in range(level):
        e = np.concatenate([F[:, [0, 1]], F[:, [1, 2]], F[:, [2, 0]]])
        es = np.sort(e, axis=1)
        key = es[:, 0] * (len(V) + 1) + es[:, 1]
        uniq, inv = np.unique(key, return_inverse=True)
        a, b = uniq // (len(V) + 1), uniq % (len(V) + 1)
        mid = V[a] + V[b]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        base = len(V)
        V = np.concatenate([V, mid])
        n = len(F)
        m01, m12, m20 = base + inv[:n], base + inv[n:2 * n], base + inv[2 * n:]
        F = np.concatenate([np.stack([F[:, 0], m01, m20], 1), np.stack([F[:, 1], m12, m01], 1),
                            np.stack([F[:, 2], m20, m12], 1), np.stack([m01, m12, m20], 1)])
    # smooth seeded relief (a scanned statue is smooth at the triangle scale; per-vertex white noise makes spikes
    # that defeat the reference's midpoint-of-bounds BVH split and leaves hundreds of triangles per leaf)
    ph = [((_hash_u32(np.array([seed * 131 + k], dtype=np.uint32))[0] & 0xFFFF) / 65535.0) * 6.283185307 for k in range(6)]
    disp = (np.sin(3.0 * V[:, 0] + ph[0]) * np.sin(4.0 * V[:, 1] + ph[1]) + 0.5 * np.sin(7.0 * V[:, 2] + ph[2]) * np.sin(5.0 * V[:, 0] + ph[3])
            + 0.25 * np.sin(13.0 * V[:, 1] + ph[4]) * np.sin(11.0 * V[:, 2] + ph[5])) / 1.75
    rr = radius * (1.0 + bump * disp)
    P = V * rr[:, None]
    lines = ["# synthetic seeded icosphere", "o icosphere"]
    lines += ["v %.6f %.6f %.6f" % tuple(v) for v in P]
    if with_normals:
        lines += ["vn %.6f %.6f %.6f" % tuple(n) for n in V]
        lines += ["f %d//%d %d//%d %d//%d" % (a, a, b, b, c, c) for a, b, c in (F + 1)]
    else:
        lines += ["f %d %d %d" % tuple(f) for f in (F + 1)]
    return "\n".join(lines) + "\n"


def reference_disk_texture(size: int = 1000) -> np.ndarray:
    """The reference's own disk texture, regenerated: its asset tool perlin/src/main.rs restated in C++ behind the C ABI
    (bhray_generate_disk_texture).  size=1000 reproduces src/renderer/textures/disk.png up to libm rounding."""
    from ._lib import lib
    from .layouts import check
    out = np.zeros((size, size, 4), dtype=np.uint8)
    check(lib().bhray_generate_disk_texture(size, out.ctypes.data))
    return out
