"""-m gpu: a guard on milliseconds per frame.

The trace kernels' speed is a property of the schedule ONE compiler finds for them (106 SGPRs in every instantiation; four instructions
fewer in the step's tail made the kernel 1 % slower: profiles/EXPERIMENTS.md R5.6), and nothing else in the suite would notice a ROCm bump
or an innocent edit losing 10 %.  The bounds are 10 % over the medians of round 5's boxes (BENCH_r05, profiles/r05_bench_*.json) and of
this round's first runs (gpurun_out/r6_base):

    1920x1080 adaptive RK,  the driver's block (--steps 20 --warmup 5, 22 frames in flight)   <= 0.445 ms per frame  (0.401-0.406 since the unified march; 0.412-0.416 before)
    ... one frame at a time (bhray_render + bhray_sync, two speculative levels)              <= 1.19 ms             (1.06-1.08; 1.19-1.21 in round 5)
    1920x1080 Euler, the driver's block                                                       <= 0.27 ms             (0.244-0.248)
    1920x1080 adaptive RK + the 327 680-triangle mesh (configs[2]), the driver's block        <= 0.56 ms             (0.508-0.516)

Measured by `bench.py` itself in a process of its own, and FIRST of the -m gpu files (hence the file's name): a second process that holds
hardware queues on the device - this pytest process once any other GPU test has run: 24 queues - makes the driver time-slice the two
processes' queues, and a frame takes 10-100x as long (measured: 4.8 ms per frame in the block, 150 ms one frame at a time).  Nothing in this
file touches HIP in the pytest process; the guard refuses to judge when it finds it was not first.

A box that runs slow says nothing about the code: two of round 5's boxes ran 11 % slower than the others at the same reported clocks.
The calibration is `bhray_selftest` - a fixed amount of VALU work (2^32 bit patterns through the exact 1/x, sqrt and step-size-power
sequences), timed here on the same device: a box whose selftest takes more than 8 % longer than NOMINAL_SELFTEST_MS is skipped, with the
two numbers in the reason.
"""
from __future__ import annotations

import json
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NOMINAL_SELFTEST_MS = 8.1      # bhray_selftest on this round's boxes: 8.03-8.11 ms (gpurun_out/r6_p/selftest_ms.txt)
SLOW_BOX = 1.08

BOUNDS_MS = {"rk_block": 0.445, "rk_one_frame": 1.19, "euler_block": 0.27, "mesh_block": 0.56}


_SELFTEST = """
import time, bhusie_amd as B
from tests import common as T
rp = B.RayPass(B.ladder_from_base((24, 14), 3, 2), device=0)
rp.set_textures(*T.textures())
rp.selftest()                                   # first launch: code object load, clock ramp
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    assert rp.selftest() == (0, 0, 0)
    ts.append((time.perf_counter() - t0) * 1e3)
rp.close()
print(sorted(ts)[1])
"""


def _selftest_ms():
    r = subprocess.run([sys.executable, "-c", _SELFTEST], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    return float(r.stdout.strip().splitlines()[-1])


@pytest.fixture(scope="module")
def box():
    ms = _selftest_ms()
    if NOMINAL_SELFTEST_MS > 0 and ms > SLOW_BOX * NOMINAL_SELFTEST_MS:
        pytest.skip(f"this box runs slow: bhray_selftest (fixed VALU work) took {ms:.1f} ms against a nominal {NOMINAL_SELFTEST_MS:.1f} ms "
                    f"(more than {int(round((SLOW_BOX - 1) * 100))} % over): its milliseconds per frame say nothing about the code")
    return ms


def _bench(*flags):
    env = dict(os.environ)
    env.pop("GPU_MAX_HW_QUEUES", None)              # bench.py sets what its frames in flight need
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline", *flags],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def _report(name, ms, box_ms):
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "perf_guard.jsonl"), "a") as f:
        f.write(json.dumps({"what": name, "ms": ms, "bound_ms": BOUNDS_MS[name], "selftest_ms": round(box_ms, 2)}) + "\n")


def _alone(d):
    """a frame that takes ten times its bound was not measured on an idle device (another process holds queues on it): nothing to judge"""
    if d["ms_per_step"] > 10 * BOUNDS_MS["rk_block"]:
        pytest.skip(f"{d['ms_per_step']} ms per frame: the device is shared with another process's hardware queues (this file must run first, on an idle GPU)")


def test_rk_block_and_one_frame_at_a_time(box):
    d = _bench()                                    # with the extra legs: the latency leg is one of them
    _alone(d)
    blk, one = d["ms_per_step"], d["latency_ms_one_frame_in_flight"]
    _report("rk_block", blk, box); _report("rk_one_frame", one, box)
    assert d["config"]["toolchain"]["major_minor"] == d["config"]["toolchain"]["pinned_major_minor"], d["config"]["toolchain"]
    assert blk <= BOUNDS_MS["rk_block"], f"1080p adaptive RK, 20-frame block: {blk} ms per frame > {BOUNDS_MS['rk_block']} (selftest {box:.1f} ms)"
    assert one <= BOUNDS_MS["rk_one_frame"], f"1080p adaptive RK, one frame at a time: {one} ms > {BOUNDS_MS['rk_one_frame']} (selftest {box:.1f} ms)"


def test_euler_block(box):
    d = _bench("--integrator", "euler", "--no-extra-legs")
    _alone(d)
    _report("euler_block", d["ms_per_step"], box)
    assert d["ms_per_step"] <= BOUNDS_MS["euler_block"], f"1080p Euler, 20-frame block: {d['ms_per_step']} ms per frame > {BOUNDS_MS['euler_block']} (selftest {box:.1f} ms)"


def test_mesh_block(box):
    d = _bench("--workload", "mesh", "--no-extra-legs")
    _alone(d)
    _report("mesh_block", d["ms_per_step"], box)
    assert d["ms_per_step"] <= BOUNDS_MS["mesh_block"], f"1080p adaptive RK + mesh, 20-frame block: {d['ms_per_step']} ms per frame > {BOUNDS_MS['mesh_block']} (selftest {box:.1f} ms)"
