#!/usr/bin/env python3
"""Generates tests/golden/frames_fma.npz: frames of the THIRD evaluation of ray.wgsl's integrator - the literal expression tree
with fused multiply-add contraction only, none of the numerics contract's reassociations (oracle/np_ray.py with set_eval(2)) -
written by the NumPy restatement, independent of the C oracle and of the kernel (BHRAY_F_EVAL_FMA), which are both held to it.

FROZEN like frames_literal.npz: it does not follow the numerics contract.

    python tests/golden/make_golden_fma.py
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


def main():
    tex = (assets.temp_lut(32), assets.disk_texture(96, seed=11), assets.sky_texture(128, 64, seed=12))
    out = dict(t_temp=tex[0], t_disk=tex[1], t_sky=tex[2])
    cases = {
        "euler_l0": (dict(), dict(), dict(integration_method=0), (72, 41), 1),
        "rk_l0": (dict(), dict(), dict(integration_method=1), (72, 41), 1),
        "rk_ladder": (dict(), dict(), dict(integration_method=1), (24, 14), 3),
        "rk_outside": (dict(position=(0.0, 3.0, -45.0), forward=(0.0, -0.06651901, 0.99778515)), dict(),
                       dict(integration_method=1), (40, 24), 2),
    }
    N.set_eval(2)
    try:
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
            print(name, sizes, stats)
    finally:
        N.set_eval(0)
    np.savez_compressed(os.path.join(HERE, "frames_fma.npz"), **out)


if __name__ == "__main__":
    main()
