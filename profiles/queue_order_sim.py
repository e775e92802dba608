#!/usr/bin/env python3
"""Would sorting a level's ray queue by predicted ray length raise the lane occupancy of the trace kernel's step loop?
A model of the persistent trace kernel's refill policy (waves of 64 lanes, batches of 16 steps, a refill from the shared queue when
>= 16 lanes are empty or nobody marches) run over the LAST level's rays of the 1920x1080 bench frame, with each ray's true length
(iterations, from the CPU oracle's per-ray diagnostics) and the queue in
  (a) the classify kernel's order (blocks of 4x2 tiles of 8x8 pixels, row-major),     (b) plain row-major pixel order,
  (c) (a) bucketed by the PARENT coarse pixel's length into 3 / 8 classes, longest first (what the proposal could know at classify time),
  (d) sorted by the ray's own true length (an oracle no kernel has).
Prints lane-steps / (64 x wave-steps) for each.  CPU only: python profiles/queue_order_sim.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bhusie_amd as B  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests import common as T  # noqa: E402

tex = T.textures(small=False)
u = T.uniforms(integration_method=1)
sc = T.oracle_scene(*u, tex)
cfg = B.ladder_for_frame((1920, 1080), 3, 4)
sizes = cfg.sizes()
imgs = O.render_ladder(sc, sizes)
kind = O.classify_level(sc, sizes[-1], imgs[-2])
aux = O.render_aux(sc, sizes[-1])[..., 1]                # iterations of every pixel's ray at the last level
auxc = O.render_aux(sc, sizes[-2])[..., 1]               # ... and at the level below (the "parent")
W, H = sizes[-1]
cx, cy = int(cfg.crop_x), int(cfg.crop_y)
ys, xs = np.nonzero(kind[cy:cy + 1080, cx:cx + 1920] == 2)
ys += cy; xs += cx
length = aux[ys, xs].astype(np.int64)
parent = auxc[np.minimum((ys / 3.0).astype(int), sizes[-2][1] - 1), np.minimum((xs / 3.0).astype(int), sizes[-2][0] - 1)]
print("rays", len(length), "mean length", length.mean(), "max", length.max())
tx, ty = (xs - cx) // 8, (ys - cy) // 8
block = (ty // 2) * ((1920 // 8 + 3) // 4) + tx // 4
tile_in_block = (ty % 2) * 4 + tx % 4
lane = ((ys - cy) % 8) * 8 + (xs - cx) % 8
order_classify = np.lexsort((lane, tile_in_block, block))


def simulate(L, waves=2048, refill_min=16, batch=16):
    n = len(L); head = 0
    rem = np.zeros((waves, 64), dtype=np.int64)
    wave_steps = 0
    while True:
        empty = rem <= 0
        need = empty.sum(axis=1)
        marching = (~empty).any(axis=1)
        want = (need >= refill_min) | ~marching
        if head < n:
            for w in np.nonzero(want & (need > 0))[0]:
                k = min(need[w], n - head)
                if k <= 0:
                    break
                idx = np.nonzero(empty[w])[0][:k]
                rem[w, idx] = L[head:head + k]; head += k
        mx = rem.max(axis=1)
        if head >= n and (mx <= 0).all():
            break
        st = np.clip(mx, 0, batch)
        wave_steps += int(st.sum())
        rem -= st[:, None]
    return L.sum() / (64.0 * wave_steps)


def bucketed(order, pred, k):
    q = np.quantile(pred, np.linspace(0, 1, k + 1)[1:-1])
    b = np.searchsorted(q, pred[order])
    return order[np.argsort(-b, kind="stable")]


res = {"classify order (shipped)": simulate(length[order_classify]),
       "row-major pixel order": simulate(length[np.lexsort((xs, ys))]),
       "classify order, 3 classes of the parent's length, longest first": simulate(length[bucketed(order_classify, parent, 3)]),
       "classify order, 8 classes of the parent's length, longest first": simulate(length[bucketed(order_classify, parent, 8)]),
       "sorted by the ray's own length (oracle)": simulate(length[np.argsort(-length, kind="stable")])}
for k, v in res.items():
    print("%-70s lane occupancy of the step loop %.4f" % (k, v))
for k in (32, 256):
    print("%-70s lane occupancy of the step loop %.4f" % ("classify order, %d classes of the parent's length, longest first" % k, simulate(length[bucketed(order_classify, parent, k)])))
print("correlation of a ray's length with its parent's: %.4f; mean |difference| %.2f iterations" % (np.corrcoef(length, parent)[0, 1], np.abs(length - parent).mean()))
