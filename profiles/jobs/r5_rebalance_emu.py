"""What bhray_rebalance buys while the camera pitches: every rank of an 8-way slab partition of the 1920x1080 frame
rendered in turn on ONE GPU (a ctx of its own per rank and period: row_rank / row_world, no gather), the bounds re-balanced between
periods with the library's own arithmetic (bhray_rebalance_slabs) from the execution spans the trace kernels stamp - exactly what
bhray_rebalance feeds it.  Policies: fixed = bounds balanced on the first period's frames and kept; follow = re-balanced every period;
follow+shift = ... and told how far the hole's projection moved (what bhray_rebalance does).  Reported per period: the slowest rank's
wall time per frame against the undivided frame's on one GPU for the same frames."""
import json, math, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "64")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bhusie_amd as B
from tests import common as T

W, H, N = 1920, 1080, 8
PERIOD, PERIODS = int(os.environ.get("EMU_PERIOD", "200")), int(os.environ.get("EMU_PERIODS", "6"))
FPB = int(os.environ.get("EMU_FPB", "10"))
cfg = B.ladder_for_frame((W, H), 3, 4)
tex = T.textures(small=False)
total = PERIOD * PERIODS
A0, A1 = -0.16, 0.16                                            # pitch range: the hole's projection crosses ~330 rows of the 1080


def cam(i):
    a = A0 + (A1 - A0) * i / (total - 1)
    return B.Camera(position=(0.0, 0.0, -19.0), forward=(0.0, math.sin(a), math.cos(a)))


def hole_row(i):
    a = A0 + (A1 - A0) * i / (total - 1)
    sm = min(cfg.level_w[3] - 1, cfg.level_h[3] - 1)
    return math.tan(-a) / math.tan(0.5) * sm * 0.5 + (cfg.level_h[3] - 1) * 0.5 - cfg.crop_y      # create_ray inverted (fov 1.0)


U = [T.uniforms(integration_method=1, time=i / 60.0, camera=cam(i)) for i in range(total)]


def run(frames, **kw):
    """wall ms per frame and trace-span ms per frame of one block of these frames"""
    rp = B.RayPass(cfg, device=0, timing="sparse", frames_in_flight=22, speculative_levels=2, **kw)
    rp.set_textures(*tex)
    for u in frames[:44]:
        rp.set_uniforms(*u); rp.render()
    rp.sync(); rp.timing()
    best = 1e9
    for rep in range(2):
        rp.sync(); t0 = time.perf_counter()
        for u in frames:
            rp.set_uniforms(*u); rp.render()
        rp.sync(); best = min(best, (time.perf_counter() - t0) / len(frames) * 1e3)
    tm = rp.timing()
    ws, px, _ = rp.work()
    rp.close()
    span = (tm.trace_exec_ms / tm.frames) if tm.frames else 0.0
    return best, span, ws, px


out = {"frame": [W, H], "partitions": N, "period_frames": PERIOD, "periods": PERIODS, "frames_per_batch": FPB,
       "hole_row_first_last": [round(hole_row(0), 1), round(hole_row(total - 1), 1)], "one_gpu_ms_per_frame": [], "policies": {}}
one = []
for k in range(PERIODS):
    w1 = run(U[k * PERIOD:(k + 1) * PERIOD])[0]
    one.append(w1)
out["one_gpu_ms_per_frame"] = [round(v, 5) for v in one]
print("one GPU:", out["one_gpu_ms_per_frame"], flush=True)

KAPPA = float(os.environ.get("EMU_KAPPA", "0.035"))             # classified pixels in wave-steps (bhray_rebalance's price for the RK kernel)


def measure(signal, r):
    wall, span, ws, px = r
    return {"wall": wall, "span": span, "work": ws + KAPPA * px}[signal]


def policy(name, follow, shift, signal):
    w = np.zeros(H)
    b = [H * p // N for p in range(N + 1)]
    # a scene at rest first: three rounds on the first period's frames (what a host does before the camera starts to move)
    for _ in range(3):
        m = [measure(signal, run(U[:PERIOD], row_rank=q, row_world=N, slab_row0=b, frames_per_batch=FPB)) for q in range(N)]
        b, _ = B.rebalance_slabs(H, b, m, w)
    rec = []
    for k in range(PERIODS):
        fr = U[k * PERIOD:(k + 1) * PERIOD]
        res = [run(fr, row_rank=q, row_world=N, slab_row0=b, frames_per_batch=FPB) for q in range(N)]
        walls = [r[0] for r in res]
        rec.append({"period": k, "slab_row0": list(b), "rank_wall_ms": [round(v, 5) for v in walls], "rank_span_ms": [round(r[1], 5) for r in res],
                    "rank_wave_steps": [round(r[2], 1) for r in res], "rank_classify_pixels": [round(r[3], 1) for r in res],
                    "slowest_rank_ms": round(max(walls), 5), "scaling": round(one[k] / max(walls), 3), "wall_imbalance": round(max(walls) / (sum(walls) / N), 3)})
        print(name, rec[-1]["period"], rec[-1]["scaling"], rec[-1]["wall_imbalance"], b, flush=True)
        if follow:
            sh = (hole_row((k + 1) * PERIOD - 1) - hole_row(k * PERIOD)) if shift else 0.0      # one period's displacement, as bhray_rebalance takes it from the uniforms
            b, _ = B.rebalance_slabs(H, b, [measure(signal, r) for r in res], w, shift_rows=sh)
    out["policies"][name] = {"signal": signal, "periods": rec, "scaling_min": min(r["scaling"] for r in rec), "scaling_mean": round(sum(r["scaling"] for r in rec) / len(rec), 3)}

for name, follow, shift, signal in [p_ for p_ in (("fixed", False, False, "work"), ("spans+shift", True, True, "span"), ("work+shift (bhray_rebalance)", True, True, "work"), ("wall+shift (reference: the ranks' own wall times)", True, True, "wall"))
                                    if not os.environ.get("EMU_POLICIES") or p_[0].split()[0] in os.environ["EMU_POLICIES"].split(",")]:
    policy(name, follow, shift, signal)
# what a wave-step and a classified pixel cost: least squares of the ranks' wall times over all periods and policies
A, y = [], []
for pol in out["policies"].values():
    for r in pol["periods"]:
        for q in range(N):
            A.append([r["rank_wave_steps"][q], r["rank_classify_pixels"][q], 1.0]); y.append(r["rank_wall_ms"][q])
coef = np.linalg.lstsq(np.asarray(A), np.asarray(y), rcond=None)[0]
out["fit_wall_ms"] = {"per_wave_step": float(coef[0]), "per_classified_pixel": float(coef[1]), "constant": float(coef[2]), "pixel_in_wave_steps": float(coef[1] / coef[0]) if coef[0] else None}
print("fit", out["fit_wall_ms"])
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r05_rebalance_emulated.json", "w"), indent=1)
print({k: (v["scaling_min"], v["scaling_mean"]) for k, v in out["policies"].items()})
