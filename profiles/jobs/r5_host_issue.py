"""Host time of bhray_render for the 8-partition ctx: what ONE host thread pays per frame and per launched batch
when one bhray_ctx drives 8 partitions.  On a one-GPU box the partitions share device 0 (--devices 0 x8): the host work - 8 engines'
staging, ~9 launches + events per engine and batch, one RCCL group, the de-interleave launch - is what it is on 8 real GPUs.
usage: python profiles/jobs/r5_host_issue.py [out.json]"""
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "64")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bhusie_amd as B
from tests import common as T

cfg = B.ladder_for_frame((1920, 1080), 3, 4)
bounds = [0, 383, 435, 487, 542, 599, 657, 713, 1080]          # profiles/r04_bench_8partitions_one_gpu.json
out = {"frame": [1920, 1080], "partitions": 8, "slab_row0": bounds, "library_threads": os.environ.get("BHRAY_ISSUE_THREADS", "default"), "runs": []}
for world, devs in ((8, [0] * 8), (1, None)):
    for fpb in ((5, 16) if world > 1 else (1,)):
        for fif in ((6,) if world > 1 else (22,)):
            kw = dict(devices=devs, slab_row0=bounds) if devs else dict(device=0)
            rp = B.RayPass(cfg, frames_in_flight=fif, frames_per_batch=fpb, speculative_levels=2, **kw)
            rp.set_textures(*T.textures(small=False))
            u = T.uniforms(integration_method=1)
            rp.set_uniforms(*u)
            L, h = rp._L, rp._h
            n = fpb * min(fif, 4)                                  # never more than the slots hold: no call blocks on a full slot
            for _ in range(3):
                for _ in range(n): L.bhray_render(h)
                rp.sync()
            best, best_sync = 1e9, 1e9
            for rep in range(7):
                rp.sync()
                t0 = time.perf_counter()
                for _ in range(n):
                    L.bhray_set_uniforms(h, u[0], u[1], u[2]); L.bhray_render(h)
                t1 = time.perf_counter()
                rp.sync()
                t2 = time.perf_counter()
                best = min(best, (t1 - t0) / n); best_sync = min(best_sync, (t2 - t0) / n)
            out["runs"].append({"partitions": world, "frames_per_batch": fpb, "frames_in_flight": fif, "frames": n,
                                "host_us_per_render": round(best * 1e6, 2), "host_us_per_batch": round(best * 1e6 * fpb, 1),
                                "wall_us_per_frame_incl_sync": round(best_sync * 1e6, 2)})
            print(out["runs"][-1], flush=True)
            rp.close()
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r05_host_issue_n8.json", "w"), indent=1)
