#!/usr/bin/env python3
"""Generates tests/golden/wgsl_exec_native_samples.npz: the executed-shader pin AT THE METRIC'S OWN FRAME.

The other executed-shader fixture (wgsl_exec.npz) holds small frames (the interpreter runs ~10 traced pixels per second per core).  This
one samples the LAST level of the reference-native ladder 72x41 -> 214x121 -> 640x361 -> 1918x1081 (mod.rs:177-205), adaptive RK,
default camera and hole: a few thousand seeded pixels of every kind - copied, interpolated, traced and escaped, traced and on the disk,
traced and captured by the hole, and the clamped border columns / rows - each produced by running the reference's own `main`
(/root/reference/src/renderer/shaders/ray.wgsl, oracle/wgsl_exec.py) on that pixel with the coarser level (640x361) supplied by the C
oracle in literal mode, which the executed shader reproduces word for word on every frame of wgsl_exec.npz (tests/test_wgsl_pin.py)
and which is re-checked here on a seeded sample of the coarse level itself.

Runs only in the build container.  Committed: uniform bytes, the three small textures, pixel coordinates, their kind, the executed values.
FROZEN FIXTURE: regenerate only if the reference's shader or the literal conventions change, and say so in the commit.

    python tests/golden/make_golden_wgsl_native.py [--processes 8] [--per-class 420]
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
from oracle import oracle as O  # noqa: E402
from oracle import wgsl_exec as W  # noqa: E402

KINDS = ("copied", "interpolated", "traced_escaped", "traced_disk", "traced_captured", "border")


def _job(job):
    cam, bh, det, tex, size, prev, pix = job
    ns = W.compile_shader()
    W.bind_scene(ns, cam, bh, det, *tex)
    return W.render_pixels(ns, size, prev, pix)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--processes", type=int, default=os.cpu_count() or 8)
    ap.add_argument("--per-class", type=int, default=420)
    args = ap.parse_args()
    W.compile_shader()
    tex = (assets.temp_lut(32), assets.disk_texture(96, seed=11), assets.sky_texture(128, 64, seed=12))     # (the textures of wgsl_exec.npz)
    cam, bh, det = H.camera_uniform(), H.black_hole_uniform(), H.ray_details(integration_method=1)
    sizes = [(72, 41), (214, 121), (640, 361), (1918, 1081)]
    sc = O.OracleScene(cam, bh, det, *tex)
    O.set_literal(True)
    try:
        imgs = O.render_ladder(sc, sizes)
        kinds = O.classify_level(sc, sizes[3], imgs[2])
    finally:
        O.set_literal(False)
    last = imgs[3]
    w, h = sizes[3]
    rng = np.random.default_rng(20260930)
    alpha = last[..., 3]
    black = (last[..., :3] == 0).all(axis=-1)
    border = np.zeros((h, w), dtype=bool); border[:, w - 3:] = True; border[h - 3:, :] = True
    sets = {"copied": (kinds == 0) & ~border, "interpolated": (kinds == 1) & ~border, "traced_escaped": (kinds == 2) & (alpha == 0) & ~border,
            "traced_disk": (kinds == 2) & (alpha == 1) & ~black & ~border, "traced_captured": (kinds == 2) & (alpha == 1) & black & ~border,
            "border": border & (kinds != 0)}
    pix, kind_id = [], []
    for k, name in enumerate(KINDS):
        ys, xs = np.nonzero(sets[name])
        take = rng.choice(len(ys), size=min(args.per_class, len(ys)), replace=False)
        pix += [(int(xs[i]), int(ys[i])) for i in take]
        kind_id += [k] * len(take)
        print(name, len(ys), "->", len(take), flush=True)
    # a seeded sample of the COARSE level too: the executed shader must agree with the oracle that supplies it
    cys, cxs = rng.integers(0, sizes[2][1], size=160), rng.integers(0, sizes[2][0], size=160)
    import multiprocessing as mp
    t0 = time.time()
    n = args.processes * 6
    chunks = [pix[i::n] for i in range(n)]
    with mp.get_context("fork").Pool(args.processes) as pool:
        parts = pool.map(_job, [(cam, bh, det, tex, sizes[3], imgs[2], c) for c in chunks])
        coarse = pool.map(_job, [(cam, bh, det, tex, sizes[2], imgs[1], list(zip(cxs[i::n].tolist(), cys[i::n].tolist()))) for i in range(n)])
    vals = np.zeros((len(pix), 4), dtype=np.float32)
    for i, p in enumerate(parts):
        vals[i::n] = p
    cvals = np.zeros((len(cxs), 4), dtype=np.float32)
    for i, p in enumerate(coarse):
        cvals[i::n] = p
    print("executed %d + %d pixels in %.0f s" % (len(pix), len(cxs), time.time() - t0), flush=True)
    want_c = imgs[2][cys, cxs]
    d = want_c[:, 3] == 0                                     # direction pixels: every word; colour pixels: class (their colour goes through powf)
    assert np.array_equal(cvals[:, 3], want_c[:, 3]) and np.array_equal(cvals[d].view(np.uint32), want_c[d].view(np.uint32)), "the executed shader and the literal oracle differ on the coarse level"
    out = dict(t_temp=tex[0], t_disk=tex[1], t_sky=tex[2], camera=np.frombuffer(bytes(cam), dtype=np.uint8).copy(),
               black_hole=np.frombuffer(bytes(bh), dtype=np.uint8).copy(), details=np.frombuffer(bytes(det), dtype=np.uint8).copy(),
               sizes=np.array(sizes, dtype=np.int32), pixels=np.array(pix, dtype=np.int32), kind=np.array(kind_id, dtype=np.uint8),
               kinds=np.array([k.encode() for k in KINDS]), values=vals, coarse_pixels=np.stack([cxs, cys], axis=1).astype(np.int32), coarse_values=cvals)
    dst = os.path.join(HERE, "wgsl_exec_native_samples.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
