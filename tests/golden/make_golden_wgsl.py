#!/usr/bin/env python3
"""Generates tests/golden/wgsl_exec.npz: frames produced by EXECUTING THE REFERENCE'S OWN SHADER TEXT
(/root/reference/src/renderer/shaders/ray.wgsl, parsed and run by oracle/wgsl_exec.py) - not by any restatement of it.

Runs only in the build container (the reference is not on the GPU box); what is committed is data: the uniform bytes, the small
textures / mesh arrays the frames were made from (the same synthetic assets as the other fixtures) and the frames.  The semantics
WGSL leaves open are the literal evaluation's (wgsl_exec.py's header), so the frames are comparable bit for bit with
oracle/ray_oracle.c under oracle_set_eval(1), oracle/np_ray.py under set_literal(True) and the BHRAY_F_LITERAL kernel.

FROZEN FIXTURE: regenerate only if the reference's shader or the literal conventions change, and say so in the commit.

    python tests/golden/make_golden_wgsl.py [--processes 8]
"""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from bhusie_amd import assets  # noqa: E402
from oracle import host_oracle as H  # noqa: E402
from oracle import wgsl_exec as W  # noqa: E402


def u8(b):
    return np.frombuffer(bytes(b), dtype=np.uint8).copy()


def ladder(base, levels):
    sizes = [tuple(base)]
    for _ in range(levels - 1):
        sizes.append((sizes[-1][0] * 3 - 2, sizes[-1][1] * 3 - 2))
    return sizes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--processes", type=int, default=os.cpu_count() or 8)
    ap.add_argument("--only", default="")
    ap.add_argument("--sky-only", action="store_true", help="add the sky.wgsl frames to an existing file from its stored last levels")
    args = ap.parse_args()
    if args.sky_only:
        dst = os.path.join(HERE, "wgsl_exec.npz")
        g = dict(np.load(dst))
        for name in ("rk_ladder", "rk_outside", "euler_mesh_near"):
            last = len(g[f"{name}.sizes"]) - 1
            g[f"{name}.sky"] = W.render_sky(g[f"{name}.level{last}"], g["t_sky"]).view(np.uint16)
            print(name, "sky", g[f"{name}.sky"].shape, flush=True)
        np.savez_compressed(dst, **g)
        return
    W.compile_shader()                                        # parse once, before the pool forks
    tex = (assets.temp_lut(32), assets.disk_texture(96, seed=11), assets.sky_texture(128, 64, seed=12))
    out = dict(t_temp=tex[0], t_disk=tex[1], t_sky=tex[2])
    obj = assets.sphere_mesh_obj(8, 10, radius=8.0, bump=0.2, seed=21, with_normals=True)       # (the mesh of tests/golden/mesh.npz)
    mesh = H.load_model(obj).as_oracle_dict()
    out["mesh.obj"] = np.frombuffer(obj.encode(), dtype=np.uint8)
    for k in ("points", "normals", "triangles", "bvh_lookup"):
        out[f"mesh.{k}"] = mesh[k]
    out["mesh.nodes"] = np.frombuffer(mesh["nodes"].tobytes(), dtype=np.uint8)
    out["mesh.position"] = np.asarray(mesh["position"], dtype=np.float32)
    mesh_cam = dict(position=(0.0, 0.0, -40.0), forward=(-0.11914522, 0.0, 0.99287683), fov=1.2)
    cases = {                                                # the first six: the scenes of frames_literal.npz
        "euler_l0": (dict(), dict(), dict(integration_method=0), (72, 41), 1, False),
        "rk_l0": (dict(), dict(), dict(integration_method=1), (72, 41), 1, False),
        "rk_ladder": (dict(), dict(), dict(integration_method=1), (24, 14), 3, False),
        "euler_ladder": (dict(), dict(), dict(integration_method=0, time=2.5), (24, 14), 3, False),
        "rk_outside": (dict(position=(0.0, 3.0, -45.0), forward=(0.0, -0.06651901, 0.99778515)), dict(), dict(integration_method=1), (40, 24), 2, False),
        "rk_off_origin": (dict(position=(3.0, 1.0, -19.0)), dict(position=(3.0, 1.0, 0.5), show_disk_texture=0),
                          dict(integration_method=1, step_size=0.1), (48, 27), 2, False),
        "rk_highlight": (dict(), dict(show_red_shift=0), dict(integration_method=1, highlight_interpolation=1, angle_division_threshold=0.05), (16, 9), 3, False),
        "euler_tight": (dict(fov=0.9), dict(feather_amount=0.3), dict(integration_method=0, step_size=0.4, max_iterations=150), (20, 12), 2, False),
        "rk_mesh": (mesh_cam, dict(), dict(integration_method=1, model_count=1), (40, 24), 1, True),
        "euler_mesh": (mesh_cam, dict(), dict(integration_method=0, model_count=1), (14, 8), 2, True),
        # the camera 15 units from the mesh (which stands outside the relativity sphere), the hole behind it: a quarter of the frame is mesh
        "rk_mesh_near": (dict(position=(-10.0, 0.0, 45.0), forward=(0.19611613, 0.0, -0.98058068)), dict(), dict(integration_method=1, model_count=1), (30, 18), 1, True),
        "euler_mesh_near": (dict(position=(-9.0, 1.0, 44.0), forward=(0.19611613, 0.0, -0.98058068), fov=0.8), dict(), dict(integration_method=0, model_count=1), (12, 7), 2, True),
    }
    # 24 seeded random points of the uniform space the reference's UI exposes (ranges from src/ui/*_settings.rs; the sweep of
    # tests/test_gpu_edge_cases.py with another seed), both integrators, two-level ladders
    rng = np.random.default_rng(20260929)
    for k in range(24):
        pos = rng.normal(size=3) * np.array([6.0, 4.0, 6.0]) + np.array([0.0, 0.0, -16.0])
        fwd = -pos + rng.normal(size=3) * 4.0
        fwd = fwd / np.linalg.norm(fwd)
        inner = float(rng.uniform(1.2, 4.0))
        ck = dict(position=tuple(float(v) for v in pos), forward=tuple(float(v) for v in fwd), fov=float(rng.uniform(0.3, 2.2)))
        bk = dict(accretion_disk_rotation=tuple(float(v) for v in rng.uniform(-1.5, 1.5, size=3)), accretion_disk_inner=inner,
                  accretion_disk_outer=inner + float(rng.uniform(1.0, 12.0)), rotation_speed=float(rng.uniform(0, 10)),
                  relativity_sphere_radius=float(rng.uniform(8.0, 40.0)), show_disk_texture=int(rng.integers(0, 2)),
                  show_red_shift=int(rng.integers(0, 2)), feather_amount=float(rng.uniform(0.05, 1.0)))
        if k % 6 == 5:
            bk["position"] = tuple(float(v) for v in rng.normal(size=3) * 3.0)
        dk = dict(integration_method=int(rng.integers(0, 2)), step_size=float(rng.uniform(0.05, 0.6)), max_iterations=int(rng.integers(50, 900)),
                  angle_division_threshold=float(rng.uniform(0.0, 0.2)), time=float(rng.uniform(0, 100)))
        cases[f"fuzz{k:02d}"] = (ck, bk, dk, (10, 6), 2, False)
    for name, (ck, bk, dk, base, levels, with_mesh) in cases.items():
        if args.only and name not in args.only.split(",") and not (args.only == "fuzz" and name.startswith("fuzz")):
            continue
        cam, bh, det = H.camera_uniform(**ck), H.black_hole_uniform(**bk), H.ray_details(**dk)
        sizes = ladder(base, levels)
        t0 = time.time()
        imgs = W.render_ladder(cam, bh, det, tex, sizes, models=[mesh] if with_mesh else (), processes=args.processes)
        out[f"{name}.camera"] = u8(cam); out[f"{name}.black_hole"] = u8(bh); out[f"{name}.details"] = u8(det)
        out[f"{name}.sizes"] = np.array(sizes, dtype=np.int32)
        out[f"{name}.mesh"] = np.array([int(with_mesh)], dtype=np.int32)
        for l, im in enumerate(imgs):
            assert not np.isnan(im[..., 3]).any() or name.startswith("fuzz"), "a pixel was not stored"
            out[f"{name}.level{l}"] = im
        if name in ("rk_ladder", "rk_outside", "euler_mesh_near"):          # sky.wgsl executed over the last level (rgba16float)
            out[f"{name}.sky"] = W.render_sky(imgs[-1], tex[2]).view(np.uint16)
        print(f"{name}: {sizes} {time.time() - t0:.0f} s", flush=True)
    dst = os.path.join(HERE, "wgsl_exec.npz")
    if args.only and os.path.exists(dst):
        old = dict(np.load(dst)); old.update(out); out = old
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
