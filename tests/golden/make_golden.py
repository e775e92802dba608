#!/usr/bin/env python3
"""Generates tests/golden/*.npz with the INDEPENDENT NumPy restatement (oracle/np_ray.py) and the
Python restatement of the host code (oracle/host_oracle.py).  Run in the build container:

    python tests/golden/make_golden.py

The reference itself cannot produce vectors (Rust + WGSL, no toolchain here; no tests upstream), so
these fixtures pin the C oracle and the HIP path against a second implementation written from
/root/reference/src/renderer/shaders/ray.wgsl.  Inputs (uniform bytes, small textures, mesh arrays)
are stored next to the expected outputs, so the fixtures are self-contained data.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from bhusie_amd import assets  # noqa: E402
from oracle import host_oracle as H  # noqa: E402
from oracle import np_ray as N  # noqa: E402


def u8(b):
    return np.frombuffer(bytes(b), dtype=np.uint8).copy()


def frames():
    tex = (assets.temp_lut(32), assets.disk_texture(96, seed=11), assets.sky_texture(128, 64, seed=12))
    out = dict(t_temp=tex[0], t_disk=tex[1], t_sky=tex[2])
    cases = {
        # name: (camera kwargs, bh kwargs, details kwargs, base, levels)
        "euler_l0": (dict(), dict(), dict(integration_method=0), (72, 41), 1),
        "rk_l0": (dict(), dict(), dict(integration_method=1), (72, 41), 1),
        "rk_ladder": (dict(), dict(), dict(integration_method=1), (24, 14), 3),
        "euler_ladder": (dict(), dict(), dict(integration_method=0, time=2.5), (24, 14), 3),
        "rk_outside": (dict(position=(0.0, 3.0, -45.0), forward=(0.0, -0.06651901, 0.99778515)), dict(),
                       dict(integration_method=1), (40, 24), 2),
        "euler_flags_off": (dict(fov=0.8), dict(show_disk_texture=0, show_red_shift=0, feather_amount=0.5),
                            dict(integration_method=0, step_size=0.25, max_iterations=600), (32, 18), 2),
    }
    for name, (ck, bk, dk, base, levels) in cases.items():
        cam, bh, det = H.camera_uniform(**ck), H.black_hole_uniform(**bk), H.ray_details(**dk)
        sizes = [(base[0], base[1])]
        for _ in range(levels - 1):
            sizes.append((sizes[-1][0] * 3 - 2, sizes[-1][1] * 3 - 2))
        stats = {}
        imgs = N.render_ladder(N.Scene(cam, bh, det, *tex), sizes, stats)
        out[f"{name}.camera"] = u8(cam); out[f"{name}.black_hole"] = u8(bh); out[f"{name}.details"] = u8(det)
        out[f"{name}.sizes"] = np.array(sizes, dtype=np.int32)
        for l, im in enumerate(imgs):
            out[f"{name}.level{l}"] = im
        out[f"{name}.stats"] = np.array([stats.get(k, 0) for k in ("traced", "steps", "copied", "interpolated", "sky_samples")], dtype=np.int64)
        if name in ("rk_ladder", "rk_outside"):          # sky resolve pass (sky.wgsl) of the final level, rgba16float
            out[f"{name}.sky"] = N.sky_resolve(imgs[-1], tex[2]).view(np.uint16)
        print(name, sizes, stats)
    np.savez_compressed(os.path.join(HERE, "frames.npz"), **out)


def mesh():
    obj = assets.sphere_mesh_obj(8, 10, radius=8.0, bump=0.2, seed=21, with_normals=True)
    obj_flat = assets.sphere_mesh_obj(6, 8, radius=6.0, bump=0.3, seed=22, with_normals=False)
    out = dict(obj=np.frombuffer(obj.encode(), dtype=np.uint8), obj_flat=np.frombuffer(obj_flat.encode(), dtype=np.uint8))
    for tag, text in (("m", obj), ("f", obj_flat)):
        m = H.load_model(text)
        d = m.as_oracle_dict()
        for k in ("points", "normals", "triangles", "bvh_lookup"):
            out[f"{tag}.{k}"] = d[k]
        out[f"{tag}.nodes"] = np.frombuffer(d["nodes"].tobytes(), dtype=np.uint8)
        out[f"{tag}.max_depth"] = np.array([m.max_depth()], dtype=np.int32)
    # a frame with the mesh in view (camera outside the sphere), both integrators
    tex = (assets.temp_lut(32), assets.disk_texture(96, seed=11), assets.sky_texture(128, 64, seed=12))
    m = H.load_model(obj)
    md = m.as_oracle_dict()
    cam = H.camera_uniform(position=(0.0, 0.0, -40.0), forward=(-0.11914522, 0.0, 0.99287683), fov=1.2)
    bh = H.black_hole_uniform()
    out["camera"] = u8(cam); out["black_hole"] = u8(bh)
    for method in (0, 1):
        det = H.ray_details(integration_method=method, model_count=1)
        img = N.render_level(N.Scene(cam, bh, det, *tex, models=[md]), (40, 24), None)
        out[f"details{method}"] = u8(det); out[f"frame{method}"] = img
        print("mesh frame", method, float((img[..., 3] == 1).mean()))
    np.savez_compressed(os.path.join(HERE, "mesh.npz"), **out)


def functions():
    rng = np.random.default_rng(1234)
    out = {}
    # portable transcendental forms
    x = np.concatenate([rng.uniform(-1, 1, 2000), [1.0, -1.0, 0.5, -0.5, 0.0, 0.99999994, 1.5, np.nan]]).astype(np.float32)
    out["acos.x"] = x; out["acos.y"] = N.bh_acos(x)
    e = np.concatenate([10.0 ** rng.uniform(-12, 3, 2000), [2e-5, 1.0, 0.0, np.inf, 1e-45]]).astype(np.float32)
    out["powm001.x"] = e; out["powm001.y"] = N.bh_pow_m001(e)
    yy = np.concatenate([rng.normal(size=2000), [0.0, -0.0, 1.0, 0.0]]).astype(np.float32)
    xx = np.concatenate([rng.normal(size=2000), [-1.0, -1.0, 0.0, 0.0]]).astype(np.float32)
    out["atan2.y"] = yy; out["atan2.x"] = xx; out["atan2.r"] = N.bh_atan2(yy, xx)
    a = np.concatenate([rng.uniform(-50, 50, 2000), [0.0, -0.0, 3.14159274, 7000.0]]).astype(np.float32)
    out["sincos.x"] = a; out["sin.y"] = N.bh_sin(a); out["cos.y"] = N.bh_cos(a)
    # integrator: 40 steps from 64 rays, both methods
    S = N.Scene(H.camera_uniform(), H.black_hole_uniform(), H.ray_details(), assets.temp_lut(8), assets.disk_texture(8), assets.sky_texture(8, 4))
    pos = (rng.normal(size=(64, 3)) * 6 + np.array([0, 0, -12.0])).astype(np.float32)
    d = rng.normal(size=(64, 3)).astype(np.float32); d = N.vnorm(d)
    out["integ.pos"] = pos; out["integ.dir"] = d
    p, q = pos.copy(), d.copy()
    step = np.full(64, np.float32(0.15), dtype=np.float32)
    eul = []
    for _ in range(40):
        p, q = N.next_ray_euler(S, p, q, step)
        eul.append(np.concatenate([p, q], axis=1))
    out["integ.euler"] = np.stack(eul)
    p, q, h = pos.copy(), d.copy(), step.copy()
    rk = []
    for _ in range(40):
        p, q, h = N.next_ray_rk(S, p, q, h)
        rk.append(np.concatenate([p, q, h[:, None]], axis=1))
    out["integ.rk"] = np.stack(rk)
    # intersections
    n = 400
    rp = (rng.normal(size=(n, 3)) * 3).astype(np.float32); rd = N.vnorm(rng.normal(size=(n, 3)).astype(np.float32))
    out["hit.pos"] = rp; out["hit.dir"] = rd
    hs, ts = N.hit_sphere(rp, rd, np.float32(2.0), np.array([0.5, -0.25, 1.0], dtype=np.float32), np.float32(1e-8), np.float32(1e5))
    out["sphere.hit"] = hs; out["sphere.t"] = ts
    nrm = N.vnorm(np.array([0.2, -0.9, 0.1], dtype=np.float32))
    hd, td = N.hit_torus2d(rp, rd, np.float32(1.0), np.float32(4.0), np.array([0.0, 0.5, 0.0], dtype=np.float32), nrm, np.float32(1e-8), np.float32(1e5))
    out["torus.normal"] = nrm; out["torus.hit"] = hd; out["torus.t"] = td
    node = np.zeros(1, dtype=H.NODE_DTYPE)[0]
    node["min_corner"] = (-1.0, -0.5, 0.25); node["max_corner"] = (0.5, 1.5, 2.0)
    off = (np.float32(0.25), np.float32(-0.5), np.float32(0.0))
    out["aabb.t"] = np.array([N._hit_aabb(N._s3(rp[i]), N._s3(rd[i]), node, off) for i in range(n)], dtype=np.float32)
    tri = [N._s3(v) for v in ((-1.0, -1.0, 2.0), (2.0, -0.5, 2.5), (0.0, 2.0, 1.5))]
    nn = [N._s3(v) for v in ((0.0, 0.0, -1.0), (0.6, 0.0, -0.8), (0.0, 0.6, -0.8))]
    tt = np.full(n, 1e5, dtype=np.float32); th = np.zeros(n, dtype=bool); tc = np.zeros((n, 3), dtype=np.float32)
    for i in range(n):
        r = N._hit_triangle(N._s3(rp[i]), N._s3(rd[i]), np.float32(1e-8), np.float32(1e5), tri[0], tri[1], tri[2], nn[0], nn[1], nn[2])
        if r is not None:
            th[i] = True; tt[i] = r[0]; tc[i] = r[1]
    out["tri.hit"] = th; out["tri.t"] = tt; out["tri.color"] = tc
    # bilinear sampler
    tex = assets.disk_texture(16, seed=3)
    uv = rng.uniform(-0.2, 1.2, size=(300, 2)).astype(np.float32)
    out["sample.tex"] = tex; out["sample.uv"] = uv; out["sample.rgba"] = N.sample_bilinear(tex, uv[:, 0], uv[:, 1])
    # host uniforms
    out["bh_uniform.default"] = u8(H.black_hole_uniform())
    out["bh_uniform.rot"] = u8(H.black_hole_uniform(accretion_disk_rotation=(1.1, -0.4, 2.0), position=(1.0, 2.0, 3.0)))
    np.savez_compressed(os.path.join(HERE, "functions.npz"), **out)


if __name__ == "__main__":
    frames(); mesh(); functions()
    for f in ("frames.npz", "mesh.npz", "functions.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)))
