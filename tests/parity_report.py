#!/usr/bin/env python3
"""Runs on an MI355X: renders a set of configurations through the C ABI and through the CPU oracle and writes the parity
statistics the tests assert on (classes, bit-identical fraction, max / p99.9 relative error per channel) as JSON.
    python tests/parity_report.py > profiles/r01_parity.json
This is test infrastructure (it imports oracle/), not part of the product."""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bhusie_amd as B  # noqa: E402
from bhusie_amd import assets  # noqa: E402
from oracle import oracle as O  # noqa: E402

tex = (assets.temp_lut(256), assets.reference_disk_texture(1000), assets.sky_texture(4096, 2048, seed=2))


def run(name, cfg, cam, bh, det, model=None, **kw):
    rp = B.RayPass(cfg, device=0, counters=True, **kw)
    rp.set_textures(*tex)
    if model is not None:
        rp.upload_model(model)
    rp.set_uniforms(cam.uniform(), bh.uniform(), det.uniform())
    rp.render()
    got = rp.read_hdr()
    cnt = O.Counters()
    models = [model.arrays()] if model is not None else []
    want = O.render_ladder(O.OracleScene(cam.uniform(), bh.uniform(), det.uniform(), *tex, models), cfg.sizes(), cnt)[-1]
    cx, cy = int(cfg.crop_x), int(cfg.crop_y)
    want = want[cy:cy + int(cfg.frame_h), cx:cx + int(cfg.frame_w)]
    nan_g, nan_w = np.isnan(got).any(axis=-1), np.isnan(want).any(axis=-1)
    ok = ~nan_w
    with np.errstate(invalid="ignore"):
        e = np.abs(got[ok] - want[ok]) / np.maximum(np.abs(want[ok]), 1e-3)
    d = ok & (want[..., 3] == 0)
    uncropped = int(cfg.crop_x) == 0 and int(cfg.crop_y) == 0 and (int(cfg.frame_w), int(cfg.frame_h)) == cfg.sizes()[-1]
    return {"config": name, "frame": [int(cfg.frame_w), int(cfg.frame_h)], "ladder": [list(s) for s in cfg.sizes()],
            "pixels": int(want.shape[0] * want.shape[1]),
            "nan_pixels_oracle": int(nan_w.sum()), "nan_pixels_same_positions": bool(np.array_equal(nan_g, nan_w)),
            "class_mismatches": int((got[..., 3][ok] != want[..., 3][ok]).sum()),
            "direction_pixels": int(d.sum()), "direction_pixels_bit_identical": bool(np.array_equal(got[d], want[d])),
            "colour_pixels": int((ok & ~d).sum()), "bit_identical_fraction": float((got[ok] == want[ok]).all(axis=-1).mean()),
            "max_rel_err": float(e.max()), "p999_rel_err": float(np.quantile(e.max(axis=-1), 0.999)),
            "tolerance": 1e-4,
            "counters_equal_oracle": (rp.counters() == cnt.as_dict()) if uncropped else "n/a (the GPU renders the window only, the oracle the whole lattice)",
            "counters": rp.counters()}


def run_vs_literal(name, cfg, cam, bh, det):
    """HIP path against the oracle's LITERAL evaluation of the integrator (operator by operator, no N3/N7/N9/N10): the
    distance any two conforming WGSL implementations may show.  Informative; the parity bar applies to the contract oracle."""
    rp = B.RayPass(cfg, device=0)
    rp.set_textures(*tex); rp.set_uniforms(cam.uniform(), bh.uniform(), det.uniform()); rp.render()
    got = rp.read_hdr()
    try:
        O.set_literal(True)
        want = O.render_ladder(O.OracleScene(cam.uniform(), bh.uniform(), det.uniform(), *tex, []), cfg.sizes())[-1]
    finally:
        O.set_literal(False)
    cx, cy = int(cfg.crop_x), int(cfg.crop_y)
    want = want[cy:cy + int(cfg.frame_h), cx:cx + int(cfg.frame_w)]
    same = got[..., 3] == want[..., 3]
    with np.errstate(invalid="ignore"):
        e = (np.abs(got - want) / np.maximum(np.abs(want), 1e-3))[same].max(axis=-1)
    return {"config": name + " — vs the LITERAL evaluation (informative)", "pixels": int(same.size), "class_differences": int((~same).sum()),
            "median_rel_err": float(np.median(e)), "p99_rel_err": float(np.quantile(e, 0.99)), "fraction_within_1e-4": float((e <= 1e-4).mean()),
            "max_rel_err": float(e.max())}


def main():
    out = []
    cam, bh = B.Camera(), B.BlackHole()
    out.append(run("reference-native 72x41 x3 x4 -> 1918x1081, adaptive RK", B.ladder_from_base((72, 41), 3, 4), cam, bh, B.RayDetails(integration_method=1)))
    out.append(run("reference-native 1918x1081, Euler (the reference default)", B.ladder_from_base((72, 41), 3, 4), cam, bh, B.RayDetails(integration_method=0)))
    out.append(run("configs[1]: 1920x1080 window, adaptive RK", B.ladder_for_frame((1920, 1080), 3, 4), cam, bh, B.RayDetails(integration_method=1)))
    with tempfile.NamedTemporaryFile("w", suffix=".obj", delete=False) as f:
        f.write(assets.icosphere_mesh_obj(7, radius=8.0, bump=0.15, seed=3))
    model = B.load_model(f.name)
    os.unlink(f.name)
    out.append(run("configs[2]: 1920x1080 + 327680-triangle mesh at (-10,0,30)", B.ladder_for_frame((1920, 1080), 3, 4), cam, bh,
                   B.RayDetails(integration_method=1, model_count=1), model=model))
    cam2 = B.Camera(position=(0.0, 3.0, -45.0), forward=(0.0, -0.06651901, 0.99778515))
    out.append(run("camera outside the sphere (0,3,-45), 960x540, adaptive RK", B.ladder_for_frame((960, 540), 3, 4), cam2, bh, B.RayDetails(integration_method=1)))
    out.append(run("960x540, disk texture + red shift off, step 0.25", B.ladder_for_frame((960, 540), 3, 4), cam,
                   B.BlackHole(show_disk_texture=0, show_red_shift=0), B.RayDetails(integration_method=0, step_size=0.25)))
    # configs[3] / configs[4] frame sizes on one GPU (the 8-GPU runs render row tiles of exactly these frames)
    out.append(run("configs[3] frame: 3840x2160, adaptive RK", B.ladder_for_frame((3840, 2160), 3, 4), cam, bh, B.RayDetails(integration_method=1),
                   speculative_levels=2, frames_per_batch=2))
    out.append(run("configs[4] frame: 7680x4320, adaptive RK, 2048 max integrator steps", B.ladder_for_frame((7680, 4320), 3, 4), cam, bh,
                   B.RayDetails(integration_method=1, max_iterations=2048), speculative_levels=2))
    out.append(run_vs_literal("configs[1]: 1920x1080, adaptive RK", B.ladder_for_frame((1920, 1080), 3, 4), cam, bh, B.RayDetails(integration_method=1)))
    out.append(run_vs_literal("1920x1080, Euler", B.ladder_for_frame((1920, 1080), 3, 4), cam, bh, B.RayDetails(integration_method=0)))
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
