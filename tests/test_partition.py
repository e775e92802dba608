"""CPU: the row-partition arithmetic behind bhray_config.partition (stripes and slabs) and the slab balancer (bhray_balance_slabs) -
pure host code of libbhray, no device.  The balancer is held against a brute-force restatement of the work model it optimises:
a partition's work = at every ladder level, the work of the level rows its frame rows depend on (ray.wgsl:185-201)."""
import ctypes as C

import numpy as np
import pytest

import bhusie_amd as B
from bhusie_amd import layouts


def coarse_rows(fine, h, ph):
    """rows of the coarser level that the level rows `fine` read - the shader's own binary32 arithmetic (ray.wgsl:185-201)"""
    sf = (h - 1) // (ph - 1)
    ry = np.float32(ph) / np.float32(h + (sf - 1))
    tl = np.floor(np.asarray(fine, dtype=np.float32) * ry).astype(np.int64)
    need = np.zeros(ph, dtype=bool)
    need[np.clip(tl, 0, ph - 1)] = True
    need[np.clip(tl + 1, 0, ph - 1)] = True
    return np.nonzero(need)[0]


def slab_work(cfg, work, a, b):
    rows = np.arange(a, b) + cfg.crop_y
    tot = 0.0
    for l in range(cfg.levels - 1, -1, -1):
        tot += float(work[l][rows].sum())
        if l > 0:
            rows = coarse_rows(rows, cfg.level_h[l], cfg.level_h[l - 1])
    return tot


def test_stripes_and_slabs_cover_the_frame_exactly_once():
    cfg = B.ladder_for_frame((1920, 1080), 3, 4)
    cfg.row_world, cfg.stripe_rows = 8, 27
    rows = B.config_partition_rows(cfg)
    old = B.partition_rows(1080, 8, 27)
    assert all(np.array_equal(a, b) for a, b in zip(rows, old))
    assert np.array_equal(np.sort(np.concatenate(rows)), np.arange(1080))
    cfg.partition = layouts.PARTITION_SLABS
    bounds = [0, 100, 100, 400, 555, 600, 900, 1079, 1080]          # an empty partition and a one-row partition are legal
    for i, v in enumerate(bounds):
        cfg.slab_row0[i] = v
    rows = B.config_partition_rows(cfg)
    assert [len(r) for r in rows] == [100, 0, 300, 155, 45, 300, 179, 1]
    assert np.array_equal(np.concatenate(rows), np.arange(1080))
    cfg.slab_row0[8] = 1081                                              # does not end at frame_h: no rows for anybody
    assert all(len(r) == 0 for r in B.config_partition_rows(cfg))
    cfg.slab_row0[8] = 1080; cfg.slab_row0[3] = 50                      # decreasing
    assert all(len(r) == 0 for r in B.config_partition_rows(cfg))


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_balanced_slabs_minimise_the_largest_partition(world):
    cfg = B.ladder_for_frame((320, 200), 3, 3)
    rng = np.random.default_rng(world)
    work = []
    for l in range(cfg.levels):
        h = cfg.level_h[l]
        y = np.arange(h) / (h - 1.0)
        w = 200.0 * np.exp(-((y - 0.55) / 0.12) ** 2) + 3.0 + rng.random(h)           # a hole band over a sky floor
        work.append((w * (40 if l < 2 else 10 * (l + 1))).astype(np.uint64))
    bounds = B.balance_slabs(cfg, work, world)
    assert bounds[0] == 0 and bounds[-1] == 200 and all(a <= b for a, b in zip(bounds, bounds[1:]))
    got = max(slab_work(cfg, work, a, b) for a, b in zip(bounds, bounds[1:]) if b > a)
    # brute force over contiguous partitions: dynamic programming on the same work model
    H = 200
    W = np.full((H + 1, H + 1), np.inf)
    for a in range(H):
        for b in range(a + 1, H + 1):
            W[a, b] = slab_work(cfg, work, a, b)
    best = W[0, :].copy()                        # best[b] = min over partitions of [0, b) into k slabs of the largest slab
    for k in range(2, world + 1):
        nxt = np.full(H + 1, np.inf)
        for b in range(1, H + 1):
            nxt[b] = min(max(best[a], W[a, b]) for a in range(0, b)) if b >= 1 else np.inf
            nxt[b] = min(nxt[b], best[b])                                               # a partition may own no rows
        best = nxt
    assert got <= best[H] * (1 + 1e-9), (got, best[H])
    equal = max(slab_work(cfg, work, round(H * p / world), round(H * (p + 1) / world)) for p in range(world))
    assert got < equal                                                                    # and it beats equal rows on a peaked profile


def test_balancer_without_measured_work_gives_equal_rows():
    cfg = B.ladder_for_frame((320, 200), 3, 3)
    work = [np.zeros(cfg.level_h[l], dtype=np.uint64) for l in range(cfg.levels)]
    assert B.balance_slabs(cfg, work, 4) == [0, 50, 100, 150, 200]


def _profile(h, centre, width=0.1):
    rows = np.arange(h)
    return 0.02 + np.exp(-((rows - centre) / (width * h)) ** 2) + 0.3 * np.exp(-((rows - centre) / (0.25 * h)) ** 2)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_rebalance_slabs_converges_on_equal_partition_times(world):
    """bhray_rebalance_slabs (the arithmetic of bhray_rebalance): partitions report what their rows cost; the row weights learn it, the
    bounds equalise it.  A scene at rest: a few rounds to within 3 % of the mean, from equal slabs and knowing nothing."""
    h = 1080
    cost = _profile(h, 450)
    w = np.zeros(h)
    b = [h * p // world for p in range(world + 1)]
    worst = []
    for _ in range(8):
        ms = [float(cost[b[p]:b[p + 1]].sum()) for p in range(world)]
        worst.append(max(ms) / (sum(ms) / world))
        b, pred = B.rebalance_slabs(h, b, ms, w)
        assert b[0] == 0 and b[-1] == h and all(x <= y for x, y in zip(b, b[1:]))
        assert pred >= sum(ms) / world * 0.999                       # nothing is promised below the mean
    assert worst[0] > 1.5 and worst[-1] < 1.03, worst
    assert abs(float(w.sum()) - float(cost.sum())) < 1e-6 * float(cost.sum())      # the weights carry the measured total


def test_rebalance_slabs_follows_a_moving_profile_when_told_the_shift():
    h, world = 1080, 8
    w = np.zeros(h)
    b = [h * p // world for p in range(world + 1)]
    worst = []
    for it in range(14):
        cost = _profile(h, 300 + 30 * it)
        ms = [float(cost[b[p]:b[p + 1]].sum()) for p in range(world)]
        worst.append(max(ms) / (sum(ms) / world))
        b, _ = B.rebalance_slabs(h, b, ms, w, shift_rows=30.0)
    assert max(worst[5:]) < 1.06, worst
    # ... and without the hint it lags behind
    w = np.zeros(h); b = [h * p // world for p in range(world + 1)]; lag = []
    for it in range(14):
        cost = _profile(h, 300 + 30 * it)
        ms = [float(cost[b[p]:b[p + 1]].sum()) for p in range(world)]
        lag.append(max(ms) / (sum(ms) / world))
        b, _ = B.rebalance_slabs(h, b, ms, w)
    assert max(lag[5:]) > max(worst[5:]) + 0.1


def test_rebalance_slabs_gives_the_root_fewer_rows_for_its_extra_work_and_rejects_nonsense():
    h, world = 400, 4
    w = np.zeros(h)
    b = [0, 100, 200, 300, 400]
    extra = [0.0, 0.0, 30.0, 0.0]
    for _ in range(4):
        ms = [float(b[p + 1] - b[p]) for p in range(world)]              # every row costs 1
        b, pred = B.rebalance_slabs(h, b, ms, w, extra_ms=extra)
    sizes = [y - x for x, y in zip(b, b[1:])]
    assert sizes[2] + 30 <= max(sizes) + 1 and abs(sizes[0] - sizes[1]) <= 1 and abs(max(sizes) - (h + 30) / 4) <= 1.5, sizes
    # extra work larger than the balanced load: that partition keeps no rows
    b2, _ = B.rebalance_slabs(h, [0, 100, 200, 300, 400], [100.0] * 4, np.zeros(h), extra_ms=[0.0, 500.0, 0.0, 0.0])
    assert b2[1] == b2[2]
    for bad in (dict(slab_row0=[0, 100, 200, 300, 399]), dict(slab_row0=[0, 200, 100, 300, 400]), dict(part_ms=[1.0, -1.0, 1.0, 1.0]), dict(part_ms=[1.0, float("nan"), 1.0, 1.0])):
        args = dict(slab_row0=[0, 100, 200, 300, 400], part_ms=[1.0] * 4)
        args.update(bad)
        with pytest.raises(B.BhrayError):
            B.rebalance_slabs(h, args["slab_row0"], args["part_ms"], np.zeros(h))


def test_rebalance_slabs_on_random_input_keeps_its_invariants_and_finds_the_best_bounds():
    """Random frames, worlds, bounds (empty slabs included), costs, root extras and shifts: the bounds stay a partition of the frame, the row
    weights carry the measured total, the prediction is the largest (weights of the slab + extra) of the bounds returned - and no other
    contiguous partition of those weights does better (brute force on small frames)."""
    import itertools
    rng = np.random.default_rng(11)
    for trial in range(300):
        small = trial < 120
        h = int(rng.integers(3, 14)) if small else int(rng.integers(20, 1500))
        world = int(rng.integers(2, 5)) if small else int(rng.integers(2, 17))
        cuts = np.sort(rng.integers(0, h + 1, size=world - 1))
        b = [0] + [int(v) for v in cuts] + [h]
        w = np.zeros(h) if trial % 3 == 0 else rng.uniform(0.0, 2.0, size=h)
        ms = [float(rng.uniform(0.1, 5.0)) if b[p + 1] > b[p] or rng.random() < 0.5 else 0.0 for p in range(world)]
        extra = None if trial % 2 else [float(rng.uniform(0.0, 1.5)) if rng.random() < 0.3 else 0.0 for _ in range(world)]
        shift = 0.0 if trial % 4 else float(rng.uniform(-0.3, 0.3) * h)
        prior = w.copy() if w.any() else np.ones(h)                       # (weights that are all zero start as equal rows)
        nb, pred = B.rebalance_slabs(h, b, ms, w, extra_ms=extra, shift_rows=shift)
        assert nb[0] == 0 and nb[-1] == h and all(x <= y for x, y in zip(nb, nb[1:])), (trial, nb)
        assert np.all(w >= 0.0) and np.all(np.isfinite(w))
        if shift == 0.0:      # a slab with rows and a measurement now weighs what it was measured at; every other row keeps its weight
            want = sum(ms[p] if ms[p] > 0.0 else float(prior[b[p]:b[p + 1]].sum()) for p in range(world) if b[p + 1] > b[p])
            assert abs(float(w.sum()) - want) <= 1e-9 * max(1.0, want), (trial, float(w.sum()), want)
        ex = extra if extra is not None else [0.0] * world
        wf = np.maximum(w, 1e-6 * float(w.sum()) / h)                      # a row costs something: the floor the packing works with
        load = [float(wf[nb[p]:nb[p + 1]].sum()) + ex[p] for p in range(world)]
        assert abs(max(load) - pred) <= 1e-6 * max(1.0, pred), (trial, load, pred)
        if small:
            best = min(max(float(wf[c[p]:c[p + 1]].sum()) + ex[p] for p in range(world))
                       for mid in itertools.combinations_with_replacement(range(h + 1), world - 1) for c in [(0,) + mid + (h,)])
            assert pred <= best * (1.0 + 1e-6) + 1e-12, (trial, pred, best, nb)
