#!/usr/bin/env python3
"""Evidence behind the latency work of round 2 (run on the GPU box from the repo root; writes gpurun_out/latency_experiments.json,
copied to profiles/r02_latency_experiments.json).

 1. lone_wave_phases   where one wave's time goes in a latency-bound launch (level 0 of the 1080p ladder alone, 2 993 rays dealt out
                       3 per wave): per-phase and per-iteration clocks written by a timing-only build of the SAME sources
                           make -C bhusie_amd/csrc OUT=../../profiles/variants/libbhray_prof.so  OBJDIR=_obj_prof  EXTRA=-DBHRAY_EXP_PROFILE
                           make -C bhusie_amd/csrc OUT=../../profiles/variants/libbhray_prof2.so OBJDIR=_obj_prof2 EXTRA="-DBHRAY_EXP_PROFILE -DBHRAY_EXP_PROFILE_FINE"
                       (clock64 ticks = shader clocks, 0.42 ns on the box: profiles/ubench/lone_wave.hip; every clock read costs ~70).
 2. temporal_prediction  BHRAY_F_TEMPORAL with a moving camera: rays the per-level fix-up launches had to trace (= what the prediction
                       missed) and the latency of one frame at a time, for the prediction parameters BHRAY_TEMPORAL_MARGIN /
                       BHRAY_TEMPORAL_RADIUS="last[,below]" (the library's defaults are 0.8 and 1,4).
"""
import ctypes as C
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "gpurun_out", "latency_experiments.json")


def lone_wave(fine):
    import numpy as np
    import bhusie_amd as B
    from bhusie_amd import assets
    tex = (assets.temp_lut(256), assets.reference_disk_texture(1000), assets.sky_texture(4096, 2048, seed=2))
    L = C.CDLL(B.LIB_PATH)
    L.bhray_debug_read_profile.argtypes = [C.c_void_p, C.c_size_t]
    cfg = B.ladder_from_base((73, 41), 3, 1)
    u = (B.Camera().uniform(), B.BlackHole().uniform(), B.RayDetails(integration_method=1).uniform())
    rp = B.RayPass(cfg, frames_in_flight=1, timing=True)
    rp.set_textures(*tex); rp.set_uniforms(*u)
    for _ in range(3):
        rp.render()
    rp.sync(); rp.timing()
    rp.render(); rp.sync()
    tm = rp.timing()
    buf = np.zeros(8192 * 16, np.int64)
    L.bhray_debug_read_profile(buf.ctypes.data, buf.size)
    d = buf.reshape(8192, 16)
    act = d[d[:, 2] > 0]
    it = act[:, 2].astype(float)
    out = {"launch_ms": tm.trace_ms / tm.frames, "waves_with_rays": int(len(act)), "rays": 73 * 41,
           "iterations_of_the_longest_wave": int(act[:, 2].max()), "ticks_total_longest_wave": int(act[:, 0].max()),
           "mean_ticks_per_wave": {"refill": float(act[:, 3].mean()), "disk_shading": float(act[:, 4].mean()), "flat_phase": float(act[:, 5].mean()),
                                   "epilogue": float(act[:, 6].mean()), "step_batches": float(act[:, 7].mean())},
           "ticks_per_iteration_step_batches": float((act[:, 7] / it).mean())}
    if fine:
        out["ticks_per_iteration_fine"] = {"between_iterations_and_phases": float((act[:, 8] / it).mean()), "integrator_step": float((act[:, 9] / it).mean()),
                                           "distance_and_culls": float((act[:, 10] / it).mean()), "rare_path_branch_and_count": float((act[:, 11] / it).mean()),
                                           "note": "each bucket includes one clock read (~70 ticks)"}
    rp.close()
    return out


def temporal():
    import bhusie_amd as B
    from bhusie_amd import assets
    tex = (assets.temp_lut(256), assets.reference_disk_texture(1000), assets.sky_texture(4096, 2048, seed=2))
    cfg = B.ladder_for_frame((1920, 1080), 3, 4)
    bh, det = B.BlackHole(), B.RayDetails(integration_method=1)

    def path(n, da, dy):
        out = []
        for i in range(n):
            a = da * i
            pos = (19.0 * math.sin(a), dy * i, -19.0 * math.cos(a))
            nn = math.sqrt(sum(v * v for v in pos))
            out.append((B.Camera(position=pos, forward=tuple(-v / nn for v in pos)).uniform(), bh.uniform(), det.uniform()))
        return out
    rows = []
    for name, da, dy in (("static", 0.0, 0.0), ("orbit 0.002 rad + 0.03 up per frame", 0.002, 0.03), ("orbit 0.02 rad + 0.3 up per frame (bench.py's moving camera)", 0.02, 0.3)):
        P = path(16, da, dy)
        for margin, radius in ((1.0, "0"), (0.9, "1,4"), (0.8, "0,4"), (0.8, "1,4"), (0.8, "2,4"), (0.7, "1,2"), (0.7, "1,4"), (0.5, "2,4")):
            os.environ["BHRAY_TEMPORAL_MARGIN"] = str(margin); os.environ["BHRAY_TEMPORAL_RADIUS"] = radius
            row = {"camera": name, "margin": margin, "radius_last_below": radius}
            rp = B.RayPass(cfg, frames_in_flight=1, temporal=True)
            rp.set_textures(*tex)
            ts = []
            for i, u in enumerate(P):
                rp.set_uniforms(*u)
                t0 = time.perf_counter(); rp.render(); rp.sync()
                if i >= 3:
                    ts.append(time.perf_counter() - t0)
            rp.close()
            row["latency_ms_median"] = round(sorted(ts)[len(ts) // 2] * 1e3, 4)
            rp = B.RayPass(cfg, frames_in_flight=1, temporal=True, counters=True)
            rp.set_textures(*tex)
            fix, traced = [], []
            for i, u in enumerate(P[:9]):
                rp.set_uniforms(*u); rp.render(); rp.sync()
                if i >= 5:
                    lc = [rp.level_counters(l)["traced"] for l in range(4)]
                    fix.append(lc[1:]); traced.append(sum(lc))
            rp.close()
            row["fixup_rays_levels_1_2_3_per_frame"] = fix
            row["traced_rays_per_frame"] = traced
            rows.append(row)
            print(row, flush=True)
    return rows


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what == "lone":
        print(json.dumps(lone_wave(bool(os.environ.get("FINE")))))
        sys.exit(0)
    res = {}
    env = dict(os.environ)
    for key, lib, fine in (("lone_wave_phases", "profiles/variants/libbhray_prof.so", ""), ("lone_wave_phases_fine", "profiles/variants/libbhray_prof2.so", "1")):
        if os.path.exists(os.path.join(ROOT, lib)):
            r = subprocess.run([sys.executable, __file__, "lone"], capture_output=True, text=True, env=dict(env, BHRAY_LIB=os.path.join(ROOT, lib), FINE=fine), cwd=ROOT)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            res[key] = json.loads(line[-1]) if line else {"error": r.stderr[-500:]}
    res["temporal_prediction"] = temporal()
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    json.dump(res, open(OUT, "w"), indent=1)
    print("wrote", OUT)
