"""One frame at a time, library A against library B (BHRAY_LIB of two builds of the same sources), alternating, in one process each: wall time per frame
and trace time per level for the bench scenes and ladder modes; every frame of B compared byte for byte with A's.  usage: r6_lat_ab.py [--quick] libA libB ... [rounds]"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, time, argparse, json, hashlib
sys.path.insert(0, %r)
import numpy as np
import bhusie_amd as B
import bench
out = []
for wl, integ, size, spec, temporal in (("disk", "rk", (1920, 1080), 2, False), ("disk", "rk", (1920, 1080), 3, False), ("disk", "euler", (1920, 1080), 2, False), ("disk", "rk", (1920, 1080), 0, True), ("mesh", "rk", (1920, 1080), 2, False), ("disk", "rk", (3840, 2160), 2, False)):
    a = argparse.Namespace(workload=wl, integrator=integ, max_iterations=2000, bvh="reference")
    tex, cam, bh, det, model = bench.build_scene(a)
    cfg = B.ladder_for_frame(size, 3, 4)
    rp = B.RayPass(cfg, device=0, frames_in_flight=1, speculative_levels=spec, timing=True, temporal=temporal)
    rp.set_textures(*tex)
    if model is not None: rp.upload_model(model)
    rp.set_uniforms(cam.uniform(), bh.uniform(), det.uniform())
    for _ in range(4): rp.render(); rp.sync()
    ts = []
    for _ in range(16):
        t0 = time.perf_counter(); rp.render(); rp.sync(); ts.append((time.perf_counter() - t0) * 1e3)
    tm = rp.timing(); n = max(1, tm.frames)
    h = hashlib.sha1(rp.read_hdr().tobytes()).hexdigest()[:12]
    out.append(dict(case="%%s %%s %%dx%%d S=%%d%%s" %% (wl, integ, size[0], size[1], spec, " temporal" if temporal else ""), wall=sorted(ts)[len(ts) // 2], levels=[tm.level_trace_ms[i] / n for i in range(4)], sha=h))
    rp.close()
print(json.dumps(out))
''' % ROOT
args_ = [a for a in sys.argv[1:] if not a.startswith("--")]
libs = [a for a in args_ if a.endswith(".so")]; rounds = int(args_[-1]) if not args_[-1].endswith(".so") else 2
if "--no-timing" in sys.argv: CHILD = CHILD.replace("timing=True", "timing=False").replace("tm = rp.timing(); n = max(1, tm.frames)", "n = 1").replace("[tm.level_trace_ms[i] / n for i in range(4)]", "[0.0] * 4")
if "--mesh" in sys.argv: CHILD = CHILD.replace('(("disk", "rk", (1920, 1080), 2, False), ("disk", "rk", (1920, 1080), 3, False), ("disk", "euler", (1920, 1080), 2, False), ("disk", "rk", (1920, 1080), 0, True), ', '(').replace(', ("disk", "rk", (3840, 2160), 2, False)', ', ("mesh", "rk", (1920, 1080), 3, False), ("mesh", "euler", (1920, 1080), 2, False)')
if "--quick" in sys.argv: CHILD = CHILD.replace(', ("disk", "rk", (1920, 1080), 0, True), ("mesh", "rk", (1920, 1080), 2, False), ("disk", "rk", (3840, 2160), 2, False)', "")
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        p = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=dict(os.environ, BHRAY_LIB=os.path.abspath(l)), timeout=900)
        if p.returncode: print(l, "FAILED", p.stderr[-1500:]); continue
        res[l].append(json.loads(p.stdout.strip().splitlines()[-1]))
for i in range(len(res[libs[0]][0])):
    for l in libs:
        rows = [r[i] for r in res[l]]
        print("%-32s %-40s wall %s  levels %s  %s" % (rows[0]["case"], os.path.basename(l), " ".join("%.3f" % r["wall"] for r in rows), ["%.3f" % v for v in rows[-1]["levels"]], rows[0]["sha"]))
