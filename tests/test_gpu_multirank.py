"""-m gpu: bench.py's N>1 paths end to end on ONE GPU.  The gather lives in libbhray (RCCL send/recv enqueued by bhray_render),
so `python bench.py --gpus N` needs no launcher; on a one-GPU box the device list repeats device 0 and the tiles travel as RCCL
send/recv-to-self.  --verify compares the gathered frame with a frame rendered whole, byte for byte.  Also covered: the same
command under torchrun with --process-model single (rank 0 drives the GPUs, the other ranks only keep the launcher's barriers
company).  Not coverable here: one process per GPU over RCCL (RCCL refuses two ranks on one device) and the xGMI hop."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--verify", "--steps", "13", "--warmup", "5", "--min-seconds", "0.05", "--no-cpu-baseline", "--width", "960", "--height", "540", "--frames-in-flight", "4"]


def _port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _line(r):
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = r.stdout.strip().splitlines()
    assert len(out) == 1 and out[0].startswith("{"), "stdout must be exactly the one JSON line: " + r.stdout[-2000:]
    return json.loads(out[0])


@pytest.mark.parametrize("n,extra", [(2, []), (4, ["--gather-root", "3", "--frames-per-batch", "3"]), (8, []), (4, ["--gather-sky", "--gather-root", "1"])])
def test_bench_gpus_n_without_a_launcher(n, extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--devices", ",".join(["0"] * n)] + COMMON + extra
    d = _line(subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT))
    assert d["n_gpus"] == n and d["scaling"] == "strong"
    assert d["config"]["verified_frames"] == 1
    g = d["gather"]
    assert g["batches_gathered"] >= 1 and g["bytes_received_per_frame"] > 0 and "RCCL" in g["transport"]
    if "--gather-sky" in extra:                                   # the verification compared sky images; 8 bytes per pixel travelled
        assert "RGBA16F" in g["gathered_image"] and g["bytes_received_per_frame"] < 960 * 540 * 8


def test_bench_under_torchrun_single_process_model():
    for attempt in range(4):          # (the port is free when _port() looks and may be taken when torchrun binds it: other jobs share the host's ports - seen once in round 6)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
               "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--process-model", "single", "--devices", "0,0"] + COMMON
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
        if r.returncode == 0 or "EADDRINUSE" not in r.stderr + r.stdout:
            break
    d = _line(r)
    assert d["n_gpus"] == 2 and d["config"]["verified_frames"] == 1


def test_a_gather_that_fails_in_the_warm_up_falls_back_to_stripes_and_still_reports():
    """a red collective must not cost the N > 1 run its number (nor hang it).  BENCH_TEST_GATHER_FAILS=1 makes the first block of the
    first ctx (balanced slabs, bhray_rebalance) raise BHRAY_E_COMM; every rank learns it, the ctx is dropped, interleaved stripes take over, the
    gathered frame is still verified and the line says where `value` comes from."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--devices", "0,0,0,0"] + COMMON
    d = _line(subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, BENCH_TEST_GATHER_FAILS="1")))
    assert d["config"]["verified_frames"] == 1
    assert "stripes" in d["config"]["partition"]["mode"] and "balanced slabs" in d["config"]["partition"]["fallback_from"]
    assert len(d["gather"]["fallbacks"]) == 1 and "BENCH_TEST_GATHER_FAILS" in d["gather"]["fallbacks"][0]


def test_a_gather_that_keeps_failing_ends_the_run_quickly_with_a_message():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--devices", "0,0,0,0"] + COMMON
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, BENCH_TEST_GATHER_FAILS="9"))
    assert r.returncode == 5 and "no gather configuration left" in r.stderr and r.stdout.strip() == "", (r.returncode, r.stderr[-800:])
