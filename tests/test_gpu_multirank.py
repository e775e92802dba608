"""-m gpu: the N>1 path of bench.py end to end on ONE GPU — several ranks (processes) render their row tiles on device 0,
the tiles travel through torch.distributed (gloo, staged through host memory, because RCCL refuses two ranks on one device)
and every receiving rank compares the frames it assembled with a frame it renders whole (bench.py --verify).  What this
covers that the CPU gloo test cannot: the real kernels on a row partition, frame batches, the per-slot buffers, the rotating
root, partial batches at the end of a phase.  What it cannot cover: RCCL itself and stream ordering against it (exercised at
world size 1 by `bench.py --force-distributed`)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("world,extra", [(2, []), (4, ["--gather-root", "0", "--frames-per-batch", "3"])])
def test_row_tiled_bench_assembles_the_whole_frame(world, extra):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--backend", "gloo", "--single-device",
           "--verify", "--steps", "29", "--warmup", "5", "--no-cpu-baseline", "--width", "960", "--height", "540", "--frames-in-flight", "4"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["scaling"] == "strong"
    assert d["config"]["verified_frames"] >= 12, d["config"]
