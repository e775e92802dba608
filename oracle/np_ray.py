"""NumPy binary32 restatement of the reference ray shader — the SECOND, independent oracle.

TEST INFRASTRUCTURE ONLY.  Written directly from /root/reference/src/renderer/shaders/ray.wgsl
(line references below), not from oracle/ray_oracle.c, as a vectorised masked state machine over
all pixels of a level.  Its jobs: (1) cross-check the C oracle (tests/test_oracle_cross.py requires
bit-identical direction pixels and classes), (2) generate the committed fixtures under tests/golden/
(tests/golden/make_golden.py).  Pinned to the reference's shader TEXT (oracle/wgsl_exec.py executes ray.wgsl; this module's literal
mode reproduces its classes and direction pixels bit for bit: tests/test_wgsl_pin.py), not to a driver run (no tests, cannot be built).

All arithmetic is numpy.float32 array arithmetic: every operator is one IEEE binary32 operation (no
fused multiply-add, no wider intermediates).  The numerics contract N1..N6 of DESIGN.md applies:
dot = (x*x + y*y) + z*z, vector/scalar = vector * (1/scalar), integer powers by multiplication,
portable polynomial forms for pow(.,-0.001), acos, atan2, sin, cos; only pow(.,1.3) uses libm.
"""
from __future__ import annotations

import struct

import numpy as np

f32 = np.float32
PI = f32(3.1415926)                                   # ray.wgsl:131


# ------------------------------------------------------------------ vector helpers (N,3) arrays
def vdot(a, b):
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def vlen(a):
    return np.sqrt(vdot(a, a))


def vdivs(a, s):
    r = f32(1.0) / s
    return a * (r[..., None] if isinstance(r, np.ndarray) else r)


def vnorm(a):
    return vdivs(a, vlen(a))


def vcross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                     a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], axis=-1)


# ------------------------------------------------------------------ N7: fused multiply-add for the integrator
def fma32(a, b, c):
    """round_to_binary32(a*b + c) with ONE rounding, for binary32 inputs.  a*b is exact in binary64 (48-bit significand);
    the binary64 sum is rounded once more when cast to binary32, which can only go wrong when the binary64 sum sits exactly
    on a binary32 midpoint while the exact sum does not: TwoSum recovers the lost part and the value is nudged off the tie."""
    a64 = np.asarray(a, dtype=np.float32).astype(np.float64)
    b64 = np.asarray(b, dtype=np.float32).astype(np.float64)
    c64 = np.asarray(c, dtype=np.float32).astype(np.float64)
    with np.errstate(invalid="ignore", over="ignore"):
        p = a64 * b64
        s = p + c64
        bb = s - p
        err = (p - (s - bb)) + (c64 - bb)                     # exact: p + c = s + err
        a64, b64, c64 = np.broadcast_arrays(a64, b64, c64)
        s = np.array(s, dtype=np.float64)
        bits = s.view(np.uint64)
        tie = ((bits & np.uint64(0x1FFFFFFF)) == np.uint64(0x10000000)) & (err != 0) & np.isfinite(s)
        if np.any(tie):
            s = np.where(tie, np.nextafter(s, np.where(err > 0, np.inf, -np.inf)), s)
        return s.astype(np.float32)


def fdot(a, b):
    return fma32(a[..., 2], b[..., 2], fma32(a[..., 1], b[..., 1], a[..., 0] * b[..., 0]))


def flen(a):
    return np.sqrt(fdot(a, a))


def fnorm(a):
    return vdivs(a, flen(a))


def fcross(a, b):
    return np.stack([fma32(a[..., 1], b[..., 2], -(a[..., 2] * b[..., 1])),
                     fma32(a[..., 2], b[..., 0], -(a[..., 0] * b[..., 2])),
                     fma32(a[..., 0], b[..., 1], -(a[..., 1] * b[..., 0]))], axis=-1)


def fmadd3(w, s_, v):
    """v + w*s per component, fused; s_ scalar or per-ray array"""
    s_ = np.asarray(s_, dtype=np.float32)
    if s_.ndim == 1:
        s_ = s_[:, None]
    return fma32(w, s_, v)


def fmin(a, b):
    return np.where(b < a, b, a)


def fmax(a, b):
    return np.where(a < b, b, a)


def clamp(x, lo, hi):
    return fmin(fmax(x, f32(lo)), f32(hi))


def mix(a, b, t):
    return a * (f32(1.0) - t) + b * t


# ------------------------------------------------------------------ portable transcendental forms (N4)
def bh_pow_m001(x):
    x = np.asarray(x, dtype=np.float32).copy()
    out = np.empty_like(x)
    nan = ~(x == x) | (x < 0)
    zero = x == 0
    inf = np.isinf(x) & (x > 0)
    ok = ~(nan | zero | inf)
    out[nan] = np.nan; out[zero] = np.inf; out[inf] = 0.0
    xs = x[ok]
    u = xs.view(np.uint32)
    sub = (u >> 23) == 0
    xs = np.where(sub, xs * f32(8388608.0), xs)
    u = xs.view(np.uint32)
    e = (u >> 23).astype(np.int32) - 127 - np.where(sub, 23, 0).astype(np.int32)
    m = ((u & np.uint32(0x007FFFFF)) | np.uint32(0x3F800000)).view(np.float32)
    big = m > f32(1.41421354)
    m = np.where(big, m * f32(0.5), m)
    e = e + big.astype(np.int32)
    s = (m - f32(1.0)) / (m + f32(1.0))
    s2 = s * s
    p = np.full_like(s, f32(0.111111112))
    p = p * s2 + f32(0.142857149)
    p = p * s2 + f32(0.2)
    p = p * s2 + f32(0.333333343)
    p = p * s2 + f32(1.0)
    lnm = (f32(2.0) * s) * p
    lnx = e.astype(np.float32) * f32(0.693147182) + lnm
    t = f32(-0.001) * lnx
    q = np.full_like(t, f32(0.00138888892))
    for c in (0.00833333377, 0.0416666679, 0.166666672, 0.5, 1.0, 1.0):
        q = q * t + f32(c)
    out[ok] = q
    return out


def _asin_kernel(z):
    z2 = z * z
    p = np.full_like(z, f32(4.2163199048e-2))
    for c in (2.4181311049e-2, 4.5470025998e-2, 7.4953002686e-2, 1.6666752422e-1):
        p = p * z2 + f32(c)
    return z + (z * z2) * p


def bh_acos(x):
    x = np.asarray(x, dtype=np.float32)
    with np.errstate(invalid="ignore"):
        bad = ~(x == x) | (x > 1) | (x < -1)
        hi = x > f32(0.5)
        lo = x < f32(-0.5)
        zh = np.sqrt(np.where(hi, (f32(1.0) - x) * f32(0.5), f32(0.0)))
        zl = np.sqrt(np.where(lo, (f32(1.0) + x) * f32(0.5), f32(0.0)))
        r = f32(1.57079637) - _asin_kernel(np.where(bad, f32(0.0), x))
        r = np.where(hi, f32(2.0) * _asin_kernel(zh), r)
        r = np.where(lo, f32(3.14159274) - f32(2.0) * _asin_kernel(zl), r)
    return np.where(bad, f32(np.nan), r).astype(np.float32)


def bh_atan2(y, x):
    y = np.asarray(y, dtype=np.float32); x = np.asarray(x, dtype=np.float32)
    ax, ay = np.abs(x), np.abs(y)
    mx = np.where(ax < ay, ay, ax); mn = np.where(ax < ay, ax, ay)
    with np.errstate(invalid="ignore", divide="ignore"):
        a = np.where(mx == 0, f32(0.0), mn / mx)
        red = a > f32(0.414213568)
        t = np.where(red, (a - f32(1.0)) / (a + f32(1.0)), a)
    base = np.where(red, f32(0.785398185), f32(0.0))
    z = t * t
    p = np.full_like(z, f32(8.05374449538e-2))
    p = p * z - f32(1.38776856032e-1)
    p = p * z + f32(1.99777106478e-1)
    p = p * z - f32(3.33329491539e-1)
    r = base + ((p * z) * t + t)
    r = np.where(ay > ax, f32(1.57079637) - r, r)
    r = np.where(x < 0, f32(3.14159274) - r, r)
    return np.where(np.signbit(y), -r, r).astype(np.float32)


def _sincos(xin, kind):
    xin = np.asarray(xin, dtype=np.float32)
    x = np.abs(xin)
    sign = np.signbit(xin) if kind == 0 else np.zeros(x.shape, dtype=bool)
    bad = ~(x <= f32(3.0e9))
    xs = np.where(bad, f32(0.0), x)
    j = (xs * f32(1.27323954)).astype(np.uint32)
    j = j + (j & np.uint32(1))
    y = j.astype(np.float32)
    xr = ((xs - y * f32(0.78515625)) - y * f32(2.4187564849853515625e-4)) - y * f32(3.77489497744594108e-8)
    j = j & np.uint32(7)
    flip = j > 3
    sign = sign ^ flip
    j = np.where(flip, j - np.uint32(4), j)
    if kind == 1:
        sign = sign ^ (j > 1)
    z = xr * xr
    mid = (j == 1) | (j == 2)
    use_cos = mid if kind == 0 else ~mid
    pc = np.full_like(z, f32(2.443315711809948e-5))
    pc = pc * z - f32(1.388731625493765e-3)
    pc = pc * z + f32(4.166664568298827e-2)
    rc = ((pc * z) * z - f32(0.5) * z) + f32(1.0)
    ps = np.full_like(z, f32(-1.9515295891e-4))
    ps = ps * z + f32(8.3321608736e-3)
    ps = ps * z - f32(1.6666654611e-1)
    rs = (ps * z) * xr + xr
    r = np.where(use_cos, rc, rs)
    r = np.where(sign, -r, r)
    return np.where(bad, f32(np.nan), r).astype(np.float32)


def bh_sin(x):
    return _sincos(x, 0)


def bh_cos(x):
    return _sincos(x, 1)


# ------------------------------------------------------------------ scene bytes -> fields
class Scene:
    def __init__(self, camera: bytes, black_hole: bytes, details: bytes, t_temp, t_disk, t_sky, models=()):
        c = struct.unpack("<3fI3ff", bytes(camera))
        self.cam_pos = np.array(c[0:3], dtype=np.float32); self.cam_fwd = np.array(c[4:7], dtype=np.float32)
        self.fov = f32(c[7])
        b = struct.unpack("<4f3fi3fi12ff8i", bytes(black_hole))
        self.inner, self.outer, self.rot_speed, self.R = (f32(v) for v in b[0:4])
        self.bh_pos = np.array(b[4:7], dtype=np.float32); self.show_tex = b[7]
        self.bh_normal = np.array(b[8:11], dtype=np.float32); self.show_shift = b[11]
        m = b[12:24]
        self.M = [np.array(m[0:3], dtype=np.float32), np.array(m[4:7], dtype=np.float32), np.array(m[8:11], dtype=np.float32)]
        self.feather = f32(b[24])
        d = struct.unpack("<iifififi", bytes(details))
        self.model_count = d[1]; self.time = f32(d[2]); self.method = d[3]; self.step_size = f32(d[4])
        self.max_iter = d[5]; self.thr = f32(d[6])
        self.t_temp, self.t_disk, self.t_sky = t_temp, t_disk, t_sky
        self.models = list(models)


# ------------------------------------------------------------------ texture.rs:16-69 / textureSampleLevel
def sample_bilinear(tex, u, v):
    h, w = tex.shape[0], tex.shape[1]

    def coord(t, n):
        with np.errstate(invalid="ignore"):
            x = t * f32(n) - f32(0.5)
            x = np.where(~(x >= f32(-1.0)), f32(-1.0), x)
            x = np.where(x > f32(n), f32(n), x)
        fl = np.floor(x)
        a = fl.astype(np.int64); b = a + 1
        return np.clip(a, 0, n - 1), np.clip(b, 0, n - 1), (x - fl).astype(np.float32)

    x0, x1, fx = coord(np.asarray(u, dtype=np.float32), w)
    y0, y1, fy = coord(np.asarray(v, dtype=np.float32), h)
    T = tex.astype(np.float32) / f32(255.0)
    a, b, c, d = T[y0, x0], T[y0, x1], T[y1, x0], T[y1, x1]
    top = mix(a, b, fx[..., None]); bot = mix(c, d, fx[..., None])
    return mix(top, bot, fy[..., None])


# ------------------------------------------------------------------ ray.wgsl:269-285
def create_rays(S: Scene, px, py, sw, sh):
    sm = min(sw - 1, sh - 1)
    inc = f32(1.0) / f32(sm)
    posx = (f32(2.0) * (px.astype(np.float32) - f32(sw - 1) * f32(0.5))) * inc
    posy = (f32(2.0) * (py.astype(np.float32) - f32(sh - 1) * f32(0.5))) * inc
    plane_up = np.array([0.0, -1.0, 0.0], dtype=np.float32)
    right = vnorm(vcross(S.cam_fwd, plane_up))
    up = vnorm(vcross(S.cam_fwd, right))
    half = np.asarray([S.fov / f32(2.0)], dtype=np.float32)
    ff = f32(1.0) / (bh_sin(half) / bh_cos(half))[0]
    d = (posx[:, None] * right[None, :] + posy[:, None] * up[None, :]) + (S.cam_fwd * ff)[None, :]
    d = vnorm(d)
    o = np.broadcast_to(S.cam_pos, d.shape).astype(np.float32).copy()
    return o, d


# ------------------------------------------------------------------ ray.wgsl:725-766
def hit_sphere(pos, dirn, radius, center, t_min, t_max):
    """returns (hit mask, t); t = t_max where not hit (t_max scalar or array)."""
    n = pos.shape[0]
    tmax = np.broadcast_to(np.asarray(t_max, dtype=np.float32), (n,)).astype(np.float32)
    oc = pos - center
    a = vdot(dirn, dirn)
    b = f32(2.0) * vdot(oc, dirn)
    c = vdot(oc, oc) - radius * radius
    disc = b * b - f32(4.0) * a * c
    with np.errstate(invalid="ignore", divide="ignore"):
        sq = np.sqrt(np.where(disc > 0, disc, f32(0.0)))
        t1 = (-b - sq) / (f32(2.0) * a)
        t2 = (-b + sq) / (f32(2.0) * a)
        tc = tmax.copy()
        c1 = (t1 > t_min) & (t1 < tmax)
        tc = np.where(c1, t1, tc)
        c2 = (t2 > t_min) & (t2 < tmax) & (t2 < tc)
        tc = np.where(c2, t2, tc)
        ok = (disc > 0) & (tc < tmax) & (tc > t_min)
    return ok, np.where(ok, tc, tmax).astype(np.float32)


# ------------------------------------------------------------------ ray.wgsl:668-701
def hit_torus2d(pos, dirn, inner, outer, tpos, normal, t_min, t_max):
    n = pos.shape[0]
    tmax = np.broadcast_to(np.asarray(t_max, dtype=np.float32), (n,)).astype(np.float32)
    with np.errstate(invalid="ignore", divide="ignore"):
        denom = vdot(np.broadcast_to(normal, dirn.shape), dirn)
        dist = tpos - pos
        t = vdot(dist, np.broadcast_to(normal, dirn.shape)) / denom
        inrange = (t < tmax) & (t > t_min)
        ip = pos + dirn * t[:, None]
        dc = vlen(tpos - ip)
        ok = inrange & (dc >= inner) & (dc <= outer)
    return ok, np.where(ok, t, tmax).astype(np.float32)


# ------------------------------------------------------------------ ray.wgsl:598-666
def hit_black_hole(S: Scene, pos, dirn, t_min, t_max, total_distance):
    """returns hit, t, color(n,3), opacity for each ray."""
    n = pos.shape[0]
    hs, ts = hit_sphere(pos, dirn, f32(1.0), S.bh_pos, t_min, t_max)
    hd, td = hit_torus2d(pos, dirn, S.inner, S.outer, S.bh_pos, S.bh_normal, t_min, t_max)
    hit = hs.copy(); t = ts.copy()
    color = np.zeros((n, 3), dtype=np.float32)
    opacity = np.where(hs, f32(1.0), f32(0.0)).astype(np.float32)
    disk = hd & (td < ts)
    k = np.nonzero(disk)[0]
    if k.size:
        with np.errstate(invalid="ignore", divide="ignore"):
            p, d, tt = pos[k], dirn[k], td[k]
            total_distance = np.broadcast_to(np.asarray(total_distance, dtype=np.float32), (n,))[k]
            ip = p + d * tt[:, None]
            dist = vlen(S.bh_pos - ip)
            density = f32(1.0) - vlen(vdivs(ip, S.outer))
            e0, e1 = S.inner, S.inner + f32(1.0)
            s = clamp((dist - e0) / (e1 - e0), 0.0, 1.0)
            density = density * (s * s * (f32(3.0) - f32(2.0) * s))
            density = density * (f32(1.0) / np.sqrt(dist))
            od = np.power(f32(30.0) * density, f32(1.3)).astype(np.float32)
            op = clamp(od * f32(0.2), 0.0, 1.0)
            col = np.stack([od, od, od], axis=-1)
            if S.show_tex != 0:
                r = (dist - S.inner) / (S.outer - S.inner)
                rel = vdivs(ip - S.bh_pos, S.outer)
                rot = (S.M[0][None, :] * rel[:, 0:1] + S.M[1][None, :] * rel[:, 1:2]) + S.M[2][None, :] * rel[:, 2:3]
                angle = -bh_atan2(rot[:, 2], rot[:, 0])
                ph = angle + S.time * S.rot_speed
                u = bh_sin(ph) * r; v = bh_cos(ph) * r
                u = (u + f32(1.0)) * f32(0.5); v = (v + f32(1.0)) * f32(0.5)
                dc = sample_bilinear(S.t_disk, u, v)
                op = op * clamp(f32(0.7) + dc[:, 3] * f32(0.5), 0.0, 1.0)
                col = col * (dc[:, 0:3] * dc[:, 3:4])
            if S.show_shift != 0:
                y = f32(1.0) - (f32(15000.0) - f32(10000.0)) / (f32(100000.0) - f32(10000.0))
                up = vnorm(np.array([0.0, -1.0, 0.0], dtype=np.float32))
                sv = vcross(vnorm(ip), np.broadcast_to(up, ip.shape)) * f32(0.6)
                vel = vdot(d, sv)
                dop = np.sqrt((f32(1.0) - vel) / (f32(1.0) + vel))
                grav = np.sqrt((f32(1.0) - f32(2.0) / dist) / (f32(1.0) - f32(2.0) / total_distance))
                sh = clamp(grav * dop, 0.0, 1.0)
                sc = sample_bilinear(S.t_temp, sh * sh, np.full_like(sh, y))
                col = col * sc[:, 0:3]
        hit[k] = True; t[k] = tt; color[k] = col; opacity[k] = op
    return hit, t, color, opacity


# ------------------------------------------------------------------ mesh: ray.wgsl:287-363, 703-723, 768-847 (scalar)
def _s3(a):
    return (f32(a[0]), f32(a[1]), f32(a[2]))


def _det3(c0, c1, c2):
    return (c0[0] * (c1[1] * c2[2] - c2[1] * c1[2]) - c1[0] * (c0[1] * c2[2] - c2[1] * c0[2])) + c2[0] * (c0[1] * c1[2] - c1[1] * c0[2])


def _sub(a, b):
    return (a[0] - b[0], a[1] - b[1], a[2] - b[2])


def _hit_aabb(pos, dirn, node, off):
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = (f32(1.0) / dirn[0], f32(1.0) / dirn[1], f32(1.0) / dirn[2])
        tmn, tmx = [], []
        for a in range(3):
            lo = f32(node["min_corner"][a]) + off[a]; hi = f32(node["max_corner"][a]) + off[a]
            t1 = (lo - pos[a]) * inv[a]; t2 = (hi - pos[a]) * inv[a]
            tmn.append(t2 if t2 < t1 else t1); tmx.append(t2 if t1 < t2 else t1)
        def mx(a, b): return b if a < b else a
        def mn(a, b): return b if b < a else a
        tmin_axis = mx(mx(tmn[0], tmn[1]), tmn[2]); tmax_axis = mn(mn(tmx[0], tmx[1]), tmx[2])
        if tmin_axis > tmax_axis or tmax_axis < 0:
            return f32(1e8)
        return tmin_axis


def _hit_triangle(pos, dirn, t_min, t_max, A, B, Cc, n1, n2, n3):
    with np.errstate(divide="ignore", invalid="ignore"):
        ab, ac = _sub(B, A), _sub(Cc, A)
        cr = (ab[1] * ac[2] - ab[2] * ac[1], ab[2] * ac[0] - ab[0] * ac[2], ab[0] * ac[1] - ab[1] * ac[0])
        ln = np.sqrt((cr[0] * cr[0] + cr[1] * cr[1]) + cr[2] * cr[2]); rl = f32(1.0) / ln
        n = (cr[0] * rl, cr[1] * rl, cr[2] * rl)
        rdt = (dirn[0] * n[0] + dirn[1] * n[1]) + dirn[2] * n[2]
        if rdt > 0:
            rdt = rdt * f32(-1.0); n = (n[0] * f32(-1.0), n[1] * f32(-1.0), n[2] * f32(-1.0))
        if abs(rdt) < f32(0.00001):
            return None
        den = _det3(dirn, _sub(A, B), _sub(A, Cc))
        if abs(den) < f32(0.00001):
            return None
        u = _det3(dirn, _sub(A, pos), _sub(A, Cc)) / den
        if u < 0 or u > 1:
            return None
        v = _det3(dirn, _sub(A, B), _sub(A, pos)) / den
        if v < 0 or u + v > 1:
            return None
        t = _det3(_sub(A, pos), _sub(A, B), _sub(A, Cc)) / den
        if t > t_min and t < t_max:
            w = (f32(1.0) - u) - v
            nm = tuple((n1[a] * w + n2[a] * u) + n3[a] * v for a in range(3))
            col = tuple(-nm[a] * f32(0.5) + f32(0.5) for a in range(3))
            return t, col, n
        return None


def trace_ray_model(m, pos, dirn, t_min, t_max):
    nodes, lookup, tris, pts, nrm = m["nodes"], m["bvh_lookup"], m["triangles"], m["points"], m["normals"]
    off = _s3(m["position"])
    pos = _s3(pos); dirn = _s3(dirn)
    best = None; best_t = f32(t_max)
    node = 0; stack = []
    while True:
        N = nodes[node]
        cnt, contents = int(N["obj_count"]), int(N["left_child"])
        if cnt == 0:
            c1, c2 = contents, contents + 1
            d1 = _hit_aabb(pos, dirn, nodes[c1], off); d2 = _hit_aabb(pos, dirn, nodes[c2], off)
            if d1 > d2:
                d1, d2, c1, c2 = d2, d1, c2, c1
            if d1 > best_t:
                if not stack:
                    break
                node = stack.pop()
            else:
                node = c1
                if d2 < best_t:
                    stack.append(c2)
        else:
            for i in range(cnt):
                ti = tris[int(lookup[contents + i])]
                P = [tuple(f32(pts[int(ti[k])][a]) + off[a] for a in range(3)) for k in range(3)]
                Nn = [_s3(nrm[int(ti[3 + k])]) for k in range(3)]
                r = _hit_triangle(pos, dirn, f32(t_min), f32(t_max), P[0], P[1], P[2], Nn[0], Nn[1], Nn[2])
                if r is not None and r[0] < best_t:
                    best, best_t = r, r[0]
            if not stack:
                break
            node = stack.pop()
    return best


def hit_models(S: Scene, pos, dirn, t_min, t_max):
    """hit_ray(.., render_triangles=true, render_black_hole=false), ray.wgsl:365-393."""
    n = pos.shape[0]
    hit = np.zeros(n, dtype=bool); t = np.full(n, f32(t_max), dtype=np.float32)
    color = np.zeros((n, 3), dtype=np.float32); opacity = np.zeros(n, dtype=np.float32)
    if S.model_count <= 0:
        return hit, t, color, opacity
    l0 = (f32(0.2), f32(0.2), f32(-1.0))
    ll = np.sqrt((l0[0] * l0[0] + l0[1] * l0[1]) + l0[2] * l0[2]); lr = f32(1.0) / ll
    light = (l0[0] * lr, l0[1] * lr, l0[2] * lr)
    for r in range(n):
        for mi in range(S.model_count):
            m = S.models[mi]
            if int(m.get("visible", 1)) == 0:
                continue
            res = trace_ray_model(m, pos[r], dirn[r], t_min, t_max)
            if res is not None and res[0] < t[r]:
                tt, col, nrm = res
                diffuse = (nrm[0] * light[0] + nrm[1] * light[1]) + nrm[2] * light[2]
                hit[r] = True; t[r] = tt; opacity[r] = 1.0
                color[r] = [col[0] * diffuse, col[1] * diffuse, col[2] * diffuse]
    return hit, t, color, opacity


# ------------------------------------------------------------------ ray.wgsl:133-165, 401-480
def _K(x):
    return f32(x)


A21 = _K(1.0 / 5.0)
A31, A32 = _K(3.0 / 40.0), _K(9.0 / 40.0)
A41, A42, A43 = _K(3.0 / 10.0), _K(-9.0 / 10.0), _K(6.0 / 5.0)
A51, A52, A53, A54 = _K(-11.0 / 54.0), _K(5.0 / 2.0), _K(-70.0 / 27.0), _K(35.0 / 27.0)
A61, A62, A63, A64, A65 = _K(1631.0 / 55296.0), _K(175.0 / 512.0), _K(575.0 / 13824.0), _K(44275.0 / 110592.0), _K(253.0 / 4096.0)
B1, B2, B3, B4, B5, B6 = 37.0 / 378.0, 0.0, 250.0 / 621.0, 125.0 / 594.0, 0.0, 512.0 / 1771.0
BA1, BA2, BA3, BA4, BA5, BA6 = 2825.0 / 27648.0, 0.0, 18575.0 / 48384.0, 13525.0 / 55296.0, 277.0 / 14336.0, 1.0 / 4.0
DB = [_K(b - ba) for b, ba in ((B1, BA1), (B2, BA2), (B3, BA3), (B4, BA4), (B5, BA5), (B6, BA6))]
BA = [_K(v) for v in (BA1, BA2, BA3, BA4, BA5, BA6)]


def f_scale(h2, dist):
    """N9: the scalar of f (ray.wgsl:401-403) for one step, s = (-1.5*h2) * (1/dist^5)"""
    d2 = dist * dist
    d5 = (d2 * d2) * dist
    return ((f32(-1.5) * h2) * (f32(1.0) / d5)).astype(np.float32)


# ---- the LITERAL reading of the integrator: ray.wgsl:401-480 operator by operator under N0-N2 (one binary32 operation per
# WGSL operator, no fused multiply-add, no reassociation; pow(d,5) = ((d*d)*(d*d))*d and pow(l,2) = l*l).  It exists to PIN the
# distance between the numerics contract (N3/N7/N9/N10: what the HIP kernel computes by default) and the shader text:
# tests/golden/frames_literal.npz is generated from it and is never regenerated when the contract changes.
LITERAL = False
# A third evaluation (set_eval(2); C oracle: oracle_set_eval(2); kernel: BHRAY_F_EVAL_FMA): the literal expression tree with
# fused multiply-add contraction ONLY - every `x*y + z` of ray.wgsl:401-480 whose product is a direct operand of the addition is
# one fma, the first product of a sum of products stays rounded - and none of the contract's reassociations (N9 / N10).
FMA_ONLY = False


def set_literal(on: bool) -> None:
    global LITERAL, FMA_ONLY
    LITERAL = bool(on)
    FMA_ONLY = False


def set_eval(mode: int) -> None:
    """0 the numerics contract, 1 the literal text, 2 fused multiply-add contraction only"""
    global LITERAL, FMA_ONLY
    LITERAL, FMA_ONLY = mode == 1, mode == 2


def _fsum(terms):
    """k_1*c_1 + k_2*c_2 + ... with the contraction a compiler applies: first product rounded, every further term one fma"""
    k, c = terms[0]
    acc = (k * c).astype(np.float32)
    for k, c in terms[1:]:
        acc = fma32(k, c, acc)
    return acc


def next_ray_euler_fma(S, pos, dirn, step):
    lc = flen(fcross(pos, dirn)); h2 = lc * lc
    dist = flen((pos - S.bh_pos).astype(np.float32))
    nd = fnorm(fmadd3(f_literal(S, pos, h2, dist), step, dirn))
    npos = fmadd3(nd, step, pos)
    return npos.astype(np.float32), nd.astype(np.float32)


def next_ray_rk_fma(S, pos, dirn, h):
    dist = flen((pos - S.bh_pos).astype(np.float32))
    lc = flen(fcross(pos, dirn)); h2 = lc * lc
    k1 = f_literal(S, pos, h2, dist)
    k2 = f_literal(S, fmadd3((k1 * A21).astype(np.float32), h, pos), h2, dist)
    k3 = f_literal(S, fmadd3(_fsum([(k1, A31), (k2, A32)]), h, pos), h2, dist)
    k4 = f_literal(S, fmadd3(_fsum([(k1, A41), (k2, A42), (k2, A43)]), h, pos), h2, dist)                 # a_43*k_2 (sic)
    k5 = f_literal(S, fmadd3(_fsum([(k1, A51), (k2, A52), (k3, A53), (k4, A54)]), h, pos), h2, dist)
    k6 = f_literal(S, fmadd3(_fsum([(k1, A61), (k2, A62), (k3, A63), (k4, A64), (k5, A65)]), h, pos), h2, dist)
    ks = (k1, k2, k3, k4, k5, k6)
    e = (_fsum(list(zip(ks, DB))) * h[:, None]).astype(np.float32)
    ea = np.abs(e)
    e_max = fmax(fmax(ea[:, 0], ea[:, 1]), ea[:, 2])
    nd = fnorm(fmadd3(_fsum(list(zip(ks, BA))), h, dirn))
    npos = fmadd3(dirn, h, pos)                                                                           # old direction
    with np.errstate(invalid="ignore"):
        grow = e_max > f32(0.00002)
    nh = np.where(grow, h * (f32(0.9) * bh_pow_m001(np.where(grow, e_max, f32(1.0)))), h * f32(1.0001)).astype(np.float32)
    return npos.astype(np.float32), nd.astype(np.float32), nh


def f_literal(S, p, h2, dist):
    """fn f, ray.wgsl:401-403: -1.5 * h2 * (rayPos - black_hole.position) / pow(dist, 5.0)"""
    c = (f32(-1.5) * h2).astype(np.float32)
    num = ((p - S.bh_pos).astype(np.float32) * c[:, None]).astype(np.float32)
    d2 = dist * dist
    d5 = (d2 * d2) * dist
    return vdivs(num, d5).astype(np.float32)


def next_ray_euler_literal(S, pos, dirn, step):
    """ray.wgsl:467-480"""
    lc = vlen(vcross(pos, dirn)); h2 = lc * lc
    dist = vlen((pos - S.bh_pos).astype(np.float32))
    nd = vnorm((dirn + f_literal(S, pos, h2, dist) * step[:, None]).astype(np.float32))
    npos = (pos + nd * step[:, None]).astype(np.float32)
    return npos, nd.astype(np.float32)


def next_ray_rk_literal(S, pos, dirn, h):
    """ray.wgsl:405-465; the retry loop (425-451) cannot change h and is run once (D1)"""
    dist = vlen((pos - S.bh_pos).astype(np.float32))
    lc = vlen(vcross(pos, dirn)); h2 = lc * lc
    hh = h[:, None]

    def wsum(terms):                                            # a*k_a + b*k_b + ... left to right
        k, c = terms[0]
        acc = k * c
        for k, c in terms[1:]:
            acc = acc + k * c
        return acc.astype(np.float32)

    k1 = f_literal(S, pos, h2, dist)
    k2 = f_literal(S, pos + (k1 * A21) * hh, h2, dist)
    k3 = f_literal(S, pos + wsum([(k1, A31), (k2, A32)]) * hh, h2, dist)
    k4 = f_literal(S, pos + wsum([(k1, A41), (k2, A42), (k2, A43)]) * hh, h2, dist)                      # a_43*k_2 (sic)
    k5 = f_literal(S, pos + wsum([(k1, A51), (k2, A52), (k3, A53), (k4, A54)]) * hh, h2, dist)
    k6 = f_literal(S, pos + wsum([(k1, A61), (k2, A62), (k3, A63), (k4, A64), (k5, A65)]) * hh, h2, dist)
    ks = (k1, k2, k3, k4, k5, k6)
    e = wsum(list(zip(ks, DB))) * hh
    ea = np.abs(e)
    e_max = fmax(fmax(ea[:, 0], ea[:, 1]), ea[:, 2])
    nd = vnorm((dirn + wsum(list(zip(ks, BA))) * hh).astype(np.float32))
    npos = (pos + dirn * hh).astype(np.float32)                                                           # old direction
    with np.errstate(invalid="ignore"):
        grow = e_max > f32(0.00002)
    nh = np.where(grow, h * (f32(0.9) * bh_pow_m001(np.where(grow, e_max, f32(1.0)))), h * f32(1.0001)).astype(np.float32)
    return npos, nd.astype(np.float32), nh


def next_ray_euler(S, pos, dirn, step):
    if LITERAL:
        return next_ray_euler_literal(S, pos, dirn, step)
    if FMA_ONLY:
        return next_ray_euler_fma(S, pos, dirn, step)
    cr = fcross(pos, dirn); h2 = fdot(cr, cr)               # N3: pow(length(v), 2.0) = dot(v, v)
    q0 = (pos - S.bh_pos).astype(np.float32)                # N9: position relative to the hole
    dist = flen(q0)
    nd = fnorm(fmadd3(q0, (f_scale(h2, dist) * step).astype(np.float32), dirn))          # N9, N10
    npos = fmadd3(nd, step, pos)
    return npos, nd


def _lin(terms):
    """sum of k_i * c_i, left to right: first product rounded, the rest fused (N7)"""
    k, c = terms[0]
    acc = k * c
    for k, c in terms[1:]:
        acc = fma32(k, c, acc)
    return acc


def next_ray_rk(S, pos, dirn, h):
    if LITERAL:
        return next_ray_rk_literal(S, pos, dirn, h)
    if FMA_ONLY:
        return next_ray_rk_fma(S, pos, dirn, h)
    q0 = (pos - S.bh_pos).astype(np.float32)                # N9: position relative to the hole, once per step
    dist = flen(q0)
    cr = fcross(pos, dirn); h2 = fdot(cr, cr)               # N3: pow(length(v), 2.0) = dot(v, v)
    s = f_scale(h2, dist)[:, None]                          # N9: f(p) = (p - bh) * s

    # N10: K_i = h*k_i = (q0 + sum a_ij K_j) * (s*h); zero-coefficient terms (b_2, b*_2) dropped
    sh = (s * h[:, None]).astype(np.float32)

    def stage(terms):
        acc = q0
        for K, c in terms:
            acc = fma32(K, c, acc)
        return (acc * sh).astype(np.float32)

    K1 = (q0 * sh).astype(np.float32)
    K2 = stage([(K1, A21)])
    K3 = stage([(K1, A31), (K2, A32)])
    K4 = stage([(K1, A41), (K2, A42), (K2, A43)])                                            # a_43*k_2 (sic, ray.wgsl:431)
    K5 = stage([(K1, A51), (K2, A52), (K3, A53), (K4, A54)])
    K6 = stage([(K1, A61), (K2, A62), (K3, A63), (K4, A64), (K5, A65)])
    e = (K1 * DB[0]).astype(np.float32)
    for K, c in ((K3, DB[2]), (K4, DB[3]), (K5, DB[4]), (K6, DB[5])):
        e = fma32(K, c, e)
    ea = np.abs(e)
    e_max = fmax(fmax(ea[:, 0], ea[:, 1]), ea[:, 2])
    # retry loop (ray.wgsl:425-451) cannot change h: run once
    ds = (K1 * BA[0]).astype(np.float32)                     # small terms first, then ONE addition to the unit-length direction
    for K, c in ((K3, BA[2]), (K4, BA[3]), (K5, BA[4]), (K6, BA[5])):
        ds = fma32(K, c, ds)
    nd = fnorm((dirn + ds).astype(np.float32))
    npos = fmadd3(dirn, h, pos)                              # old direction (ray.wgsl:456)
    with np.errstate(invalid="ignore"):
        grow = e_max > f32(0.00002)
    nh = np.where(grow, h * (f32(0.9) * bh_pow_m001(np.where(grow, e_max, f32(1.0)))), h * f32(1.0001)).astype(np.float32)
    return npos.astype(np.float32), nd.astype(np.float32), nh


# ------------------------------------------------------------------ ray.wgsl:482-596
def trace_rays(S: Scene, origin, direction, stats=None):
    n = origin.shape[0]
    t_max, t_min = f32(1e5), f32(1e-8)
    cpos, cdir = origin.copy(), direction.copy()
    ppos, pdir = origin.copy(), direction.copy()
    rkpos, rkdir = origin.copy(), direction.copy()
    rkh = np.full(n, S.step_size, dtype=np.float32)
    ray_distance = vlen(origin - S.bh_pos)
    rel = ray_distance < S.R
    amount = np.ones(n, dtype=np.float32)
    color = np.zeros((n, 3), dtype=np.float32)
    step = np.full(n, S.step_size, dtype=np.float32)
    hit = np.zeros(n, dtype=bool)
    closest = vlen(cpos - S.bh_pos)
    alive = np.ones(n, dtype=bool)
    i_final = np.full(n, S.max_iter, dtype=np.int64)
    steps = 0
    for it in range(S.max_iter):
        if not alive.any():
            break
        c_hit = np.zeros(n, dtype=bool); c_t = np.zeros(n, dtype=np.float32)
        c_col = np.zeros((n, 3), dtype=np.float32); c_op = np.zeros(n, dtype=np.float32)
        kr = np.nonzero(alive & rel)[0]
        kf = np.nonzero(alive & ~rel)[0]
        if kr.size:
            steps += kr.size
            ppos[kr] = cpos[kr]; pdir[kr] = cdir[kr]
            if S.method == 0:
                np_, nd_ = next_ray_euler(S, cpos[kr], cdir[kr], step[kr])
                cpos[kr] = np_; cdir[kr] = nd_
            else:
                np_, nd_, nh_ = next_ray_rk(S, rkpos[kr], rkdir[kr], rkh[kr])
                rkpos[kr] = np_; rkdir[kr] = nd_; rkh[kr] = nh_
                cpos[kr] = np_; cdir[kr] = nd_; step[kr] = nh_
            cd = vlen(cpos[kr] - S.bh_pos) if LITERAL else flen(cpos[kr] - S.bh_pos)     # N7: the integrator's distance
            closest[kr] = np.where(cd < closest[kr], cd, closest[kr])
            pdir[kr] = cdir[kr]
            h_, t_, col_, op_ = hit_black_hole(S, ppos[kr], pdir[kr], t_min, step[kr], ray_distance[kr])
            c_hit[kr] = h_; c_t[kr] = t_; c_col[kr] = col_; c_op[kr] = op_
            out = cd > S.R
            ko = kr[out]
            if ko.size:
                rel[ko] = False
                fw = S.R * S.feather
                fs = S.R - fw
                lin = clamp((closest[ko] - fs) / fw, 0.0, 1.0)
                m = lin * lin
                cdir[ko] = mix(cdir[ko], direction[ko], m[:, None])
        if kf.size:
            h_, t_, col_, op_ = hit_models(S, cpos[kf], cdir[kf], t_min, t_max)
            hs, ts = hit_sphere(ppos[kf], pdir[kf], S.R, S.bh_pos, t_min, t_max)
            none = ~hs & ~h_
            alive[kf[none]] = False
            i_final[kf[none]] = it
            enter = ~none & hs & (ts < t_)
            ke = kf[enter]
            cpos[ke] = cpos[ke] + cdir[ke] * ts[enter][:, None]
            rel[ke] = True
            take = ~none & ~enter
            kt = kf[take]
            c_hit[kt] = h_[take]; c_t[kt] = t_[take]; c_col[kt] = col_[take]; c_op[kt] = op_[take]
        ka = np.nonzero(alive & c_hit)[0]
        if ka.size:
            cpos[ka] = cpos[ka] + pdir[ka] * c_t[ka][:, None]
            cc = clamp(c_col[ka], 0.0, 1.0)
            color[ka] = color[ka] + cc * (amount[ka] * c_op[ka])[:, None]
            amount[ka] = amount[ka] * (f32(1.0) - c_op[ka])
            hit[ka] = True
        done = alive & (amount < f32(0.005))
        i_final[done] = it
        alive[done] = False
    out = np.zeros((n, 4), dtype=np.float32)
    colour_px = hit | (i_final <= 5)
    ks = np.nonzero(colour_px & (amount > f32(0.001)))[0]
    if ks.size:
        d = cdir[ks]
        theta = bh_atan2(np.sqrt(d[:, 0] * d[:, 0] + d[:, 2] * d[:, 2]), d[:, 1])
        phi = bh_atan2(d[:, 2], d[:, 0])
        u = (phi + f32(2.6) * PI) / (f32(2.0) * PI)
        v = (PI - theta) / PI
        u = u - np.trunc(u); v = v - np.trunc(v)
        sc = sample_bilinear(S.t_sky, u, v)[:, 0:3]
        miss = (sc * sc) * (sc * sc)
        color[ks] = color[ks] + miss * amount[ks][:, None]
    out[colour_px, 0:3] = color[colour_px]; out[colour_px, 3] = 1.0
    out[~colour_px, 0:3] = cdir[~colour_px]; out[~colour_px, 3] = 0.0
    if stats is not None:
        stats["steps"] = stats.get("steps", 0) + int(steps); stats["traced"] = stats.get("traced", 0) + n
        stats["sky_samples"] = stats.get("sky_samples", 0) + int(ks.size)
    return out


# ------------------------------------------------------------------ ray.wgsl:167-243
def angle_between(a, b):
    with np.errstate(invalid="ignore", divide="ignore"):
        d = vdot(a, b)
        c = d / (vlen(a) * vlen(b))
    return bh_acos(c)


def render_level(S: Scene, size, prev=None, stats=None):
    sw, sh = size
    ys, xs = np.mgrid[0:sh, 0:sw]
    px = xs.ravel().astype(np.int64); py = ys.ravel().astype(np.int64)
    out = np.zeros((sh * sw, 4), dtype=np.float32)
    if prev is None:
        o, d = create_rays(S, px, py, sw, sh)
        out[:] = trace_rays(S, o, d, stats)
        return out.reshape(sh, sw, 4)
    ph, pw = prev.shape[0], prev.shape[1]
    sfx, sfy = (sw - 1) // (pw - 1), (sh - 1) // (ph - 1)
    rx = f32(pw) / f32(sw + (sfx - 1)); ry = f32(ph) / f32(sh + (sfy - 1))
    ppx = px.astype(np.float32) * rx; ppy = py.astype(np.float32) * ry
    tlx = np.floor(ppx); tly = np.floor(ppy)

    def load(ix, iy):
        return prev[np.clip(iy.astype(np.int64), 0, ph - 1), np.clip(ix.astype(np.int64), 0, pw - 1)]

    c_tl = load(tlx, tly)
    copy = (np.abs(tlx - ppx) < f32(0.001)) & (np.abs(tly - ppy) < f32(0.001))
    c_bl = load(tlx, tly + f32(1.0)); c_tr = load(tlx + f32(1.0), tly); c_br = load(tlx + f32(1.0), tly + f32(1.0))
    a0 = angle_between(c_bl[:, :3], c_tl[:, :3]); a1 = angle_between(c_br[:, :3], c_tr[:, :3])
    a2 = angle_between(c_tl[:, :3], c_tr[:, :3]); a3 = angle_between(c_bl[:, :3], c_br[:, :3])
    with np.errstate(invalid="ignore"):
        alphas0 = (c_tl[:, 3] == 0) & (c_tr[:, 3] == 0) & (c_bl[:, 3] == 0) & (c_br[:, 3] == 0)
        small = (a0 < S.thr) & (a1 < S.thr) & (a2 < S.thr) & (a3 < S.thr)
    interp = ~copy & alphas0 & small
    trace = ~copy & ~interp
    out[copy] = c_tl[copy]
    tx = (ppx - tlx)[:, None]; ty = (ppy - tly)[:, None]
    top = mix(c_tl[:, :3], c_tr[:, :3], tx); bot = mix(c_bl[:, :3], c_br[:, :3], tx)
    p = mix(top, bot, ty)
    out[interp, 0:3] = p[interp]; out[interp, 3] = 0.0
    kt = np.nonzero(trace)[0]
    if kt.size:
        o, d = create_rays(S, px[kt], py[kt], sw, sh)
        out[kt] = trace_rays(S, o, d, stats)
    if stats is not None:
        stats["copied"] = stats.get("copied", 0) + int(copy.sum()); stats["interpolated"] = stats.get("interpolated", 0) + int(interp.sum())
    return out.reshape(sh, sw, 4)


def render_ladder(S: Scene, sizes, stats=None):
    imgs, prev = [], None
    for sz in sizes:
        img = render_level(S, sz, prev, stats)
        imgs.append(img); prev = img
    return imgs


# ------------------------------------------------------------------ sky.wgsl:1-38 (resolve pass)
def sky_resolve(prev, t_sky):
    """alpha == 0: direction -> uv -> sky^4, alpha 1 (sky.wgsl:19-26); else pass through.  Stored as rgba16float
    (sky.wgsl:1): numpy's float32 -> float16 cast rounds to nearest even."""
    p = np.asarray(prev, dtype=np.float32)
    h, w = p.shape[:2]
    flat = p.reshape(-1, 4)
    out = flat.copy()
    with np.errstate(invalid="ignore"):
        k = np.nonzero(flat[:, 3] == 0)[0]
    if k.size:
        d = flat[k, :3]
        # cartesian_to_spherical(p.xzy): theta = atan2(length(v.xy), v.z), phi = atan2(v.y, v.x), v = (d.x, d.z, d.y)
        theta = bh_atan2(np.sqrt(d[:, 0] * d[:, 0] + d[:, 2] * d[:, 2]), d[:, 1])
        phi = bh_atan2(d[:, 2], d[:, 0])
        u = (phi + f32(2.6) * PI) / (f32(2.0) * PI)
        v = (PI - theta) / PI
        u = u - np.trunc(u); v = v - np.trunc(v)
        sc = sample_bilinear(t_sky, u, v)[:, 0:3]
        out[k, 0:3] = (sc * sc) * (sc * sc)
        out[k, 3] = 1.0
    with np.errstate(over="ignore"):
        return out.reshape(h, w, 4).astype(np.float16)
