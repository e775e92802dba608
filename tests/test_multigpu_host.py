"""CPU: the host-side pieces of the multi-GPU mode that need no device — the row-partition arithmetic the library uses for its
staging tables and de-interleave kernel (bhray_partition_*), and the launcher plumbing of bench.py's one-process-per-GPU mode
(world size 2 over gloo: barrier, reductions, broadcast of the 128-byte communicator id)."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

import bhusie_amd as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("frame_h,world,stripe", [(1080, 8, 27), (1081, 3, 27), (2160, 8, 27), (4320, 8, 27), (7, 4, 3), (100, 16, 1),
                                                  (5, 8, 27), (110, 5, 27), (1080, 1, 27), (1, 2, 1)])
def test_partition_arithmetic_is_a_bijection_onto_the_frame_rows(frame_h, world, stripe):
    rows = B.partition_rows(frame_h, world, stripe)                  # closed forms inside libbhray
    r = np.arange(frame_h)
    owner = (r // stripe) % world                                    # the rule stated in include/bhray.h
    for k in range(world):
        assert np.array_equal(rows[k], r[owner == k]), (frame_h, world, stripe, k)
    allrows = np.concatenate(rows) if rows else np.zeros(0)
    assert sorted(allrows.tolist()) == list(range(frame_h))          # every frame row exactly once: the de-interleave is a permutation
    L = B.lib()
    import ctypes as C
    out = C.c_uint32()
    for k in range(world):                                           # one past the end, bad partition: errors, not wrap-around
        assert L.bhray_partition_row_index(frame_h, world, stripe, k, len(rows[k]), C.byref(out)) != 0
    assert L.bhray_partition_rows(frame_h, world, stripe, world) == 0
    assert L.bhray_partition_row_index(frame_h, world, stripe, world, 0, C.byref(out)) != 0


def test_partition_matches_the_per_device_engine_rule():
    """bhray_config.row_* (one ctx = one partition) and the multi-device ctx use the same rule; the 27-row default stripe is one
    level-0 lattice cell of the 4-level x3 ladder (3^3 final rows)."""
    rows = B.partition_rows(1080, 8, 27)
    assert [len(x) for x in rows] == [135] * 8
    assert rows[3][:3].tolist() == [81, 82, 83] and rows[3][27] == 81 + 8 * 27


def _port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_launcher_plumbing_world2_gloo(tmp_path):
    """bench.py's Launcher under a real 2-process gloo group on CPU: what the one-process-per-GPU mode needs from the launcher."""
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {ROOT!r})
        import bench
        w, r, l = bench.launcher_env()
        L = bench.Launcher(w, r)
        L.barrier()
        assert L.reduce([1.0 + r, 10.0], op="max") == [2.0, 10.0]
        assert L.reduce([1 + r, 5], op="sum", dtype="int64") == [3, 10]
        blob = bytes(range(128)) if r == 0 else None
        got = L.broadcast_bytes(blob, 128)
        assert got == bytes(range(128)), got
        # the balanced partition's calibration as bench.py --gpus N runs it under a launcher: the same work tables on every rank, every
        # rank's OWN probe time summed into one vector, the re-weighted bounds balanced by the library (host arithmetic) and rank 0's shared
        import numpy as np
        import bhusie_amd as B
        cfg = B.ladder_for_frame((320, 200), 3, 3)
        work = []
        for l in range(cfg.levels):
            y = np.arange(cfg.level_h[l]) / (cfg.level_h[l] - 1.0)
            work.append((1000.0 * np.exp(-((y - 0.5) / 0.1) ** 2) + 20.0).astype(np.uint64))
        b = bench.share_bounds(L, B.balance_slabs(cfg, work, 2) if r == 0 else [0, 1, 200], 2)       # rank 1's own (wrong) bounds lose
        assert b[0] == 0 and b[2] == 200 and 80 < b[1] < 120, b
        mine = [0.0, 0.0]; mine[r] = 1.0 + 0.5 * r                                                   # partition 1 measured 50 % slower than the model says
        probe = L.reduce(mine, op="sum")
        assert probe == [1.0, 1.5]
        b2 = bench.share_bounds(L, B.balance_slabs(cfg, bench.reweigh_by_probe(cfg, work, b, probe, 200), 2), 2)
        assert b2[1] > b[1], (b, b2)                                                                 # the slow partition gives rows away
        # ... and the balance of round 5 (--partition library under a launcher): every rank hands in the cost of ITS partition only, the
        # launcher sums them, the library's arithmetic gives both ranks the same bounds; rounds until nothing promises 2 % more
        row_w = np.zeros(200)
        cur = [0, 100, 200]
        prof = 1.0 + 40.0 * np.exp(-((np.arange(200) - 60.0) / 15.0) ** 2)                           # the work sits in partition 0's rows
        for _ in range(5):
            cost = [0.0, 0.0]; cost[r] = float(prof[cur[r]:cur[r + 1]].sum())
            extra = [3.0 if r == 0 else 0.0, 0.0]                                                    # the root's gather, known to the root only
            h = bench.rebalance_over_launcher(L, (cost, extra), cur, row_w, 200)
            assert h["extra_cost"] == [3.0, 0.0] and len(h["part_cost"]) == 2 and min(h["part_cost"]) > 0
            cur = h["slab_row0"]
        assert cur[1] < 80, cur
        a0, a1 = prof[:cur[1]].sum() + 3.0, prof[cur[1]:].sum()
        assert abs(a0 - a1) < 0.1 * max(a0, a1), (cur, a0, a1)
        b2 = b2 + cur
        L.close()
        open(os.path.join({str(tmp_path)!r}, "ok%d" % r), "w").write(",".join(str(v) for v in b2))
    """))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()
    assert (tmp_path / "ok0").read_text() == (tmp_path / "ok1").read_text()          # both ranks ended with the same bounds


def test_frames_per_batch_rule_counts_pixels_and_divides_the_block():
    """bench.py's batch size: about 2.5 frames' worth of 1920x1080 pixels per launch, never more than a quarter of a timed block (several
    batches must overlap: one's coarse launches with another's last level), and a divisor of a short block (batches of equal size) - the
    larger one on a tie."""
    import argparse
    sys.path.insert(0, ROOT)
    import bench

    def fpb(w, h, steps, world):
        return bench.auto_frames_per_batch(argparse.Namespace(width=w, height=h, steps=steps), world)
    assert fpb(1920, 1080, 20, 1) == 1 and fpb(3840, 2160, 20, 1) == 1          # a whole frame per GPU: no batching
    assert fpb(1920, 1080, 20, 8) == 5 and fpb(1920, 1080, 2000, 8) == 16       # capped by a quarter of the block / by 16
    assert fpb(3840, 2160, 20, 8) == 5 and fpb(3840, 2160, 500, 8) == 5
    assert fpb(7680, 4320, 20, 8) == 1
    assert fpb(1920, 1080, 20, 4) == 5 and fpb(1920, 1080, 1000, 4) == 10 and fpb(1920, 1080, 1000, 2) == 5 and fpb(1920, 1080, 20, 2) == 5
    for steps in (7, 20, 21, 64):
        for world in (2, 4, 8):
            b = fpb(1920, 1080, steps, world)
            assert 1 <= b <= max(1, steps // 4) and (steps % b == 0 or b == steps // 4)
