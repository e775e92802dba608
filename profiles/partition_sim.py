#!/usr/bin/env python3
"""partition_sim.py - where does an N-way row partition of the bench frame lose against N x one GPU?  (CPU only.)

The CPU oracle gives, for every pixel of every ladder level of the bench frame (1920x1080 window of the 73x41 x3 x4 ladder, default
camera / hole, adaptive RK), whether the ladder traces it and how many integrator iterations its ray takes.  A rank's work is the
ray-steps of the rows it renders: its own frame rows at the last level plus, at every coarser level, the rows those depend on
(bhray_api.hip: coarse_rows_needed) - recomputed by every rank that needs them, nothing is exchanged before the gather.  For a
row -> rank assignment the model reports

    redundancy   sum of the ranks' ray-steps / the undivided frame's ray-steps        (coarse rows computed more than once)
    imbalance    slowest rank's ray-steps / mean rank's ray-steps
    bound        N / (redundancy x imbalance): the scaling a perfectly efficient kernel would reach before the gather

Ray-steps are the VALU work (390 flops each); launch tails, occupancy and the chain of dependent levels are NOT in the model - it
ranks partitions, the GPU emulation (profiles/emulate_all_ranks.sh) measures them.

    python profiles/partition_sim.py [--width 1920 --height 1080] [--world 8] [--spec 2] [--out profiles/r04_partition_sim.json]
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def coarse_rows_needed(fine, h, ph):
    """bhray_api.hip: rows of level k-1 that the rows `fine` of level k read (ray.wgsl:185-201), the same binary32 arithmetic."""
    if ph == 1:
        return np.zeros(0, dtype=np.int64)
    sf = (h - 1) // (ph - 1)
    ry = np.float32(ph) / np.float32(h + (sf - 1))
    tl = np.floor(np.asarray(fine, dtype=np.float32) * ry).astype(np.int64)
    need = np.zeros(ph, dtype=bool)
    need[np.clip(tl, 0, ph - 1)] = True
    need[np.clip(tl + 1, 0, ph - 1)] = True
    return np.nonzero(need)[0]


def frame_cost_tables(width, height, levels, spec, max_iter, ray_overhead):
    import bhusie_amd as B
    from bhusie_amd import assets
    from oracle import oracle as O
    cfg = B.ladder_for_frame((width, height), 3, levels)
    sizes = cfg.sizes()
    cam, bh = B.Camera(), B.BlackHole()
    det = B.RayDetails(integration_method=1, step_size=0.15, max_iterations=max_iter, angle_division_threshold=0.02, time=0.0)
    # classification and ray lengths do not depend on the textures' contents: small ones
    sc = O.OracleScene(cam.uniform(), bh.uniform(), det.uniform(), assets.temp_lut(64), assets.disk_texture(128), assets.sky_texture(256, 128))
    imgs = O.render_ladder(sc, sizes)
    row_cost, row_rays = [], []
    for l, (w, h) in enumerate(sizes):
        its = O.render_aux(sc, (w, h))[..., 1].astype(np.float64)
        if l < max(spec, 1):
            traced = np.ones((h, w), dtype=bool)                       # level 0, and the speculative levels: every pixel is traced
        else:
            traced = O.classify_level(sc, (w, h), imgs[l - 1]) == 2
        if l == levels - 1:
            win = np.zeros((h, w), dtype=bool)
            win[:, cfg.crop_x:cfg.crop_x + width] = True
            traced &= win
        row_cost.append(((its + ray_overhead) * traced).sum(axis=1))
        row_rays.append(traced.sum(axis=1))
    return cfg, sizes, row_cost, row_rays


def rank_work(frame_rows, cfg, sizes, row_cost, row_rays):
    """(ray-steps, rays) of a rank that owns `frame_rows` (frame row indices), per level."""
    nl = len(sizes)
    rows = np.asarray(frame_rows, dtype=np.int64) + cfg.crop_y
    steps, rays = [0.0] * nl, [0] * nl
    for l in range(nl - 1, -1, -1):
        steps[l] = float(row_cost[l][rows].sum())
        rays[l] = int(row_rays[l][rows].sum())
        if l > 0:
            rows = coarse_rows_needed(rows, sizes[l][1], sizes[l - 1][1])
    return steps, rays


def assignments(height, world, row_cost_last, crop_y, stripe_work=None, density=None):
    """name -> list of row arrays, one per rank.  stripe_work(rows) = ray-steps of a rank that owned only those rows."""
    out = {}

    def stripes(sr):
        n = (height + sr - 1) // sr
        return [np.arange(s * sr, min(height, (s + 1) * sr)) for s in range(n)]

    for sr in (9, 18, 27, 36, 54, 81):
        st = stripes(sr)
        out[f"round-robin stripes of {sr}"] = [np.concatenate([st[s] for s in range(len(st)) if s % world == r] or [np.zeros(0, dtype=np.int64)]) for r in range(world)]
        snake = lambda s: (s % world) if (s // world) % 2 == 0 else world - 1 - (s % world)
        out[f"snake stripes of {sr}"] = [np.concatenate([st[s] for s in range(len(st)) if snake(s) == r] or [np.zeros(0, dtype=np.int64)]) for r in range(world)]
        # longest-processing-time greedy on what each stripe costs a rank that owns it alone (its rows + the coarse rows under them):
        # what a host could do from the previous frame's per-row counters
        cost = [stripe_work(st[s]) if stripe_work else float(row_cost_last[st[s] + crop_y].sum()) for s in range(len(st))]
        load, own = [0.0] * world, [[] for _ in range(world)]
        for s in sorted(range(len(st)), key=lambda s: -cost[s]):
            r = int(np.argmin(load)); load[r] += cost[s]; own[r].append(s)
        out[f"cost-greedy stripes of {sr}"] = [np.concatenate([st[s] for s in sorted(o)] or [np.zeros(0, dtype=np.int64)]) for o in own]
    # contiguous slabs: equal rows, and balanced by last-level cost
    edges = [round(height * r / world) for r in range(world + 1)]
    out["contiguous slabs, equal rows"] = [np.arange(edges[r], edges[r + 1]) for r in range(world)]
    c = np.cumsum(row_cost_last[crop_y:crop_y + height])
    edges = [0] + [int(np.searchsorted(c, c[-1] * r / world)) for r in range(1, world)] + [height]
    out["contiguous slabs, balanced by last-level cost"] = [np.arange(edges[r], edges[r + 1]) for r in range(world)]
    # equal-cost chunks: the frame cut into K x world contiguous chunks of equal last-level cost (few boundaries = few coarse rows
    # computed twice), dealt to the ranks by the greedy rule on what each chunk costs alone
    if density is not None:
        c = np.cumsum(density)                                     # every level's cost spread over the frame rows above it
    for K in (1, 2, 3, 4, 6):
        n = K * world
        edges = sorted(set([0] + [int(np.searchsorted(c, c[-1] * j / n)) for j in range(1, n)] + [height]))
        ch = [np.arange(edges[j], edges[j + 1]) for j in range(len(edges) - 1)]
        cost = [stripe_work(x) if stripe_work else float(row_cost_last[x + crop_y].sum()) for x in ch]
        load, own = [0.0] * world, [[] for _ in range(world)]
        for j in sorted(range(len(ch)), key=lambda j: -cost[j]):
            r = int(np.argmin(load)); load[r] += cost[j]; own[r].append(j)
        out[f"equal-cost chunks, {K} per rank, greedy"] = [np.concatenate([ch[j] for j in sorted(o)] or [np.zeros(0, dtype=np.int64)]) for o in own]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--spec", type=int, default=2, help="speculative levels (bench.py default 2: levels 0 and 1 are traced whole)")
    ap.add_argument("--max-iterations", type=int, default=2000)
    ap.add_argument("--world", type=int, nargs="*", default=[2, 4, 8])
    ap.add_argument("--ray-overhead", type=float, default=20.0, help="iterations' worth of refill + epilogue + sky sample per ray")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    cfg, sizes, row_cost, row_rays = frame_cost_tables(a.width, a.height, a.levels, a.spec, a.max_iterations, a.ray_overhead)
    whole_steps, whole_rays = rank_work(np.arange(a.height), cfg, sizes, row_cost, row_rays)
    res = {"frame": [a.width, a.height], "ladder": [list(s) for s in sizes], "speculative_levels": a.spec, "ray_overhead_iterations": a.ray_overhead,
           "whole_frame": {"ray_steps_per_level": whole_steps, "rays_per_level": whole_rays}, "worlds": {}}
    for N in a.world:
        rows = {}
        sw = lambda rows: sum(rank_work(rows, cfg, sizes, row_cost, row_rays)[0])
        dens = np.zeros(a.height)
        for r in range(a.height):
            rws = np.array([r + cfg.crop_y])
            for l in range(len(sizes) - 1, -1, -1):
                share = 1.0 if l == len(sizes) - 1 else 1.0 / (3 ** (len(sizes) - 1 - l)) / len(rws)
                dens[r] += float(row_cost[l][rws].sum()) * share
                if l > 0:
                    rws = coarse_rows_needed(rws, sizes[l][1], sizes[l - 1][1])
        for name, parts in assignments(a.height, N, row_cost[-1], cfg.crop_y, sw, dens).items():
            w = [rank_work(p, cfg, sizes, row_cost, row_rays) for p in parts]
            tot = [sum(s) for s, _ in w]
            lvl = [sum(w[r][0][l] for r in range(N)) for l in range(len(sizes))]
            red = sum(tot) / sum(whole_steps)
            imb = max(tot) / (sum(tot) / N)
            rows[name] = {"redundancy": round(red, 4), "imbalance": round(imb, 4), "bound": round(N / (red * imb), 3),
                          "redundant_ray_steps_per_level": [round(lvl[l] / whole_steps[l], 3) if whole_steps[l] else None for l in range(len(sizes))],
                          "rank_ray_steps_rel_mean": [round(t / (sum(tot) / N), 3) for t in tot]}
        res["worlds"][str(N)] = rows
        print(f"N = {N}")
        for name, r in sorted(rows.items(), key=lambda kv: -kv[1]["bound"]):
            print(f"  {name:48s} redundancy {r['redundancy']:.3f}  imbalance {r['imbalance']:.3f}  bound {r['bound']:.2f}x   per level {r['redundant_ray_steps_per_level']}")
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
