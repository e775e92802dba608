#!/usr/bin/env python3
"""bench.py — Mrays/s of the geodesic ray-trace pass on N MI355X (BASELINE.json metric).

One step = one frame of the workload through the whole pass (all ladder levels; at N>1 also the
gather of the row tiles to rank 0 and the de-interleave into the frame).  Inputs (uniforms,
textures, mesh) are resident in HBM before the timed region.  value = frame pixels * steps / time.

    python bench.py                       # N=1, configs[1]: 1920x1080 adaptive RK, disk + adaptive grid
    python bench.py --workload mesh       # configs[2]: + BVH mesh (.obj)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N   # row-tiled

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (trace_kernel; averaged over
all its launches, i.e. the 4 ladder levels, so that it matches the per-kernel average of
`rocprofv3 --kernel-trace --stats`), measured with HIP events recorded on the stream the kernels run
on.  `cpu_baseline` is the CPU oracle (a port of the reference WGSL; the reference itself is
Rust+wgpu and cannot run on this box) on a bounded sample, N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# Frames in flight run on separate HIP streams; ROCm maps streams onto 4 hardware queues by default, so 4+ streams
# would serialise pairwise (measured: 8 queues are not enough once the N>1 path adds its side streams and RCCL's).  Must be set before the HIP runtime initialises (libbhray also sets it when it is loaded).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "64")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
VALU_PEAK_TFLOPS = 157.3       # vector FP32 peak
FLOPS_PER_STEP = {0: 125.0, 1: 390.0}     # Euler / Cash–Karp, incl. per-step hit tests
FLOPS_NODE_PAIR, FLOPS_TRIANGLE = 60.0, 120.0


def algorithmic_bytes_trace(c: dict) -> float:
    """HBM bytes the trace kernel must move: per traced ray a 4 B queue entry and a
    16 B RGBA32F store; 32 B per disk shading event (2 bilinear taps x 4 texels x 4 B), 16 B per sky
    sample, 64 B per BVH node pair, 124 B per leaf triangle."""
    return (c["traced"] * 20.0 + c["disk_hits"] * 32.0 + c["sky_samples"] * 16.0
            + c["node_pairs"] * 64.0 + c["triangles"] * 124.0)


def algorithmic_bytes_frame(c: dict) -> float:
    """Whole pass: 16 B store per pixel per level, +16 B per copied pixel, +64 B (4 coarse texels)
    per interpolated/refined pixel, + the trace-side taps."""
    refined = c["traced"]
    return (c["pixels"] * 16.0 + c["copied"] * 16.0 + (c["interpolated"] + refined) * 64.0
            + c["disk_hits"] * 32.0 + c["sky_samples"] * 16.0 + c["node_pairs"] * 64.0 + c["triangles"] * 124.0)


def counters_sky_taps(c: dict, frame_pixels: int) -> float:
    """Direction pixels of the final frame = pixels the sky pass has to sample: everything that is not a colour pixel.
    Upper bound from the counters: frame pixels minus the rays that sampled the sky in-kernel or hit something."""
    return float(max(0, frame_pixels - c["sky_samples"]))


def algorithmic_flops(c: dict, method: int) -> float:
    return c["steps"] * FLOPS_PER_STEP[method] + c["node_pairs"] * FLOPS_NODE_PAIR + c["triangles"] * FLOPS_TRIANGLE


def build_scene(args):
    import bhusie_amd as B
    from bhusie_amd import assets
    tex = (assets.temp_lut(256), assets.disk_texture(1000, seed=1), assets.sky_texture(4096, 2048, seed=2))
    cam, bh = B.Camera(), B.BlackHole()
    det = B.RayDetails(integration_method=1, step_size=0.15, max_iterations=args.max_iterations,
                       angle_division_threshold=0.02, time=0.0)
    model = None
    if args.workload == "mesh":
        import tempfile
        obj = assets.icosphere_mesh_obj(7, radius=8.0, bump=0.15, seed=3)          # 327 680 triangles, leaves <= 2
        with tempfile.NamedTemporaryFile("w", suffix=".obj", delete=False) as f:
            f.write(obj)
            path = f.name
        model = B.load_model(path)
        os.unlink(path)
        det.model_count = 1
    return tex, cam, bh, det, model


def cpu_baseline(args, tex, cam, bh, det, model):
    """Oracle (port of ray.wgsl) on a bounded sample: the same camera/scene at a quarter-size frame."""
    import bhusie_amd as B
    from oracle import oracle as O
    fw, fh = max(64, args.width // 2), max(36, args.height // 2)
    cfg = B.ladder_for_frame((fw, fh), 3, args.levels)
    models = []
    if model is not None:
        models = [model.arrays()]
    sc = O.OracleScene(cam.uniform(), bh.uniform(), det.uniform(), tex[0], tex[1], tex[2], models)
    t0 = time.perf_counter()
    O.render_ladder(sc, cfg.sizes())
    dt = time.perf_counter() - t0
    last = cfg.sizes()[-1]
    return {"value": round(last[0] * last[1] / dt / 1e6, 4), "unit": "Mrays/s", "cores": O.num_threads(), "kind": "port",
            "sample": f"CPU oracle (C restatement of ray.wgsl, OpenMP x{O.num_threads()}), one {last[0]}x{last[1]} frame "
                      f"(ladder {cfg.sizes()[0][0]}x{cfg.sizes()[0][1]} x3 x{args.levels}, same camera/scene, adaptive RK), "
                      f"{dt:.2f} s wall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=["disk", "mesh"], default="disk")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--max-iterations", type=int, default=2000)
    ap.add_argument("--stripe-rows", type=int, default=27)
    ap.add_argument("--frames-in-flight", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-distributed", action="store_true", help="run the N>1 code path (process group, gather) even at world size 1")
    ap.add_argument("--emulate-world", type=int, default=0, help="single process renders only rank 0's row tiles of an N-way partition (sizing experiments)")
    ap.add_argument("--readback", action="store_true", help="also copy every frame to host memory (PCIe-inclusive; never the headline)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    distributed = world > 1 or args.force_distributed

    torch = dist = None
    if distributed:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if "RANK" not in os.environ:             # --force-distributed without a launcher
            os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"))
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import bhusie_amd as B
    from bhusie_amd.rowtile import FrameGather

    tex, cam, bh, det, model = build_scene(args)
    cfg = B.ladder_for_frame((args.width, args.height), 3, args.levels)
    dev = local_rank if distributed else 0

    def make_pass(**kw):
        rp = B.RayPass(cfg, device=dev, row_rank=rank, row_world=(args.emulate_world or world), stripe_rows=args.stripe_rows, **kw)
        rp.set_textures(*tex)
        if model is not None:
            rp.upload_model(model)
        rp.set_uniforms(cam.uniform(), bh.uniform(), det.uniform())
        return rp

    # counters (algorithmic work) from an untimed render of the same frame
    rpc = make_pass(counters=True)
    rpc.render()
    counters = rpc.counters()
    rpc.close()

    rp = make_pass(timing=True, frames_in_flight=args.frames_in_flight)
    gathers = side = None
    nbuf = max(1, args.frames_in_flight)
    if distributed:
        # one gather buffer + one torch side stream per frame in flight: the gather of frame k overlaps
        # the render of frame k+1; buffer reuse is ordered by the side stream, never by the host
        dev_t = torch.device("cuda", local_rank)
        gathers = [FrameGather(args.width, args.height, rank, world, args.stripe_rows, 0, device=dev_t) for _ in range(nbuf)]
        side = [torch.cuda.Stream(device=dev_t) for _ in range(nbuf)]

    pending = [None] * nbuf

    def step(k):
        if not distributed:
            rp.render()
            if args.readback:
                rp.read_hdr()
            return
        b = k % nbuf
        g = gathers[b]
        with torch.cuda.stream(side[b]):
            if pending[b] is not None:          # frame k-nbuf: gathered -> de-interleave on rank 0
                pending[b].wait()
                pending[b] = None
                g.assemble()
            rp.wait_stream(side[b].cuda_stream)             # render k may overwrite the buffer only after that
            rp.bind_output(g.local.data_ptr(), g.local.numel() * 4)
            rp.render()
            rp.signal_stream(side[b].cuda_stream)           # the gather reads the buffer only after render k
            pending[b] = g.gather(async_op=True)

    def drain():
        if not distributed:
            rp.sync()
            return
        for b in range(nbuf):
            with torch.cuda.stream(side[b]):
                if pending[b] is not None:
                    pending[b].wait()
                    pending[b] = None
                    gathers[b].assemble()
        rp.sync()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    drain()
    rp.timing()                                  # reset the event aggregation
    if distributed:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    drain()
    if distributed:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=torch.device("cuda", local_rank))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    tm = rp.timing()
    # the same kernels with nothing else on the device (one frame at a time): clean per-launch durations
    iso = None
    if not distributed:
        rp1 = make_pass(timing=True, frames_in_flight=1)
        for _ in range(3):
            rp1.render()
        rp1.sync(); rp1.timing()
        for _ in range(10):
            rp1.render(); rp1.sync()
        iso = rp1.timing()
        # the pass that follows the ray pass in the reference (sky.wgsl): HBM-bound, reported next to the ray pass
        for _ in range(10):
            rp1.render(); rp1.resolve_sky(); rp1.sync()
        iso_sky = rp1.timing()
        rp1.close()
    if rank == 0:
        pixels = args.width * args.height
        ms_per_step = elapsed / args.steps * 1e3
        value = pixels * args.steps / elapsed / 1e6
        # dominant kernel: trace_kernel (rank 0's launches; at N>1 rank 0 traced only its row tiles, and
        # `counters` are then rank 0's as well, so bytes and time stay consistent)
        launches = max(1, tm.trace_launches)
        avg_ms = tm.trace_ms / launches
        frames = max(1, tm.frames)
        bytes_per_launch = algorithmic_bytes_trace(counters) * frames / launches
        achieved_gbs = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        flops_per_frame = algorithmic_flops(counters, 1)
        # HBM traffic of the same kernel from the committed PMC passes (profiles/collect.sh: rocprofv3 --pmc
        # FETCH_SIZE / WRITE_SIZE, separate passes of this command); null when no summary is present
        traffic, traffic_note = None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath) and args.workload == "disk" and world == 1:
            tj = json.load(open(tpath))
            traffic = round(tj["bytes_per_launch_corrected"], 1)
            traffic_note = (f"PMC (profiles/{tj['tag']}_pmc.json): FETCH_SIZE {tj['FETCH_SIZE_KB_per_launch']:.0f} KB x2 (gfx950 correction) + "
                            f"WRITE_SIZE {tj['WRITE_SIZE_KB_per_launch']:.0f} KB per launch; raw sum {tj['bytes_per_launch_raw']:.0f} B")
        valu_tflops = flops_per_frame / (ms_per_step * 1e-3) / 1e12 * (world if world > 1 else 1) / max(1, world)   # device-level: flops per frame / wall per frame
        iso_d = None
        if iso is not None and iso.trace_launches:
            iso_ms = iso.trace_ms / iso.trace_launches
            iso_gbs = bytes_per_launch / (iso_ms * 1e-3) / 1e9
            iso_d = {"avg_launch_ms": round(iso_ms, 5), "launches": int(iso.trace_launches), "achieved": round(iso_gbs, 4),
                     "frac": round(iso_gbs / HBM_PEAK_GBS, 6),
                     "level_trace_ms": [round(iso.level_trace_ms[i] / max(1, iso.frames), 5) for i in range(args.levels)],
                     "valu_tflops": round(flops_per_frame * iso.frames / (iso.trace_ms * 1e-3) / 1e12, 4),
                     "note": "same kernels, one frame in flight, nothing else on the device (matches rocprofv3 --stats of `bench.py --frames-in-flight 1`)"}
        sky_d = None
        if iso is not None and iso_sky.sky_launches:
            sky_ms = iso_sky.sky_ms / iso_sky.sky_launches
            sky_bytes = pixels * 24.0 + counters_sky_taps(counters, pixels) * 16.0
            sky_d = {"kernel": "sky_kernel (sky.wgsl, rgba32f -> rgba16f)", "avg_launch_ms": round(sky_ms, 5), "launches": int(iso_sky.sky_launches),
                     "algorithmic_bytes_per_launch": sky_bytes, "achieved": round(sky_bytes / (sky_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(sky_bytes / (sky_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "note": "16 B read + 8 B written per pixel + 16 B of sky texels per direction pixel; not part of `value`"}
        out = {
            "metric": "Mrays/sec at 1920x1080 adaptive-RK4; 1/2/4/8 MI355X + % HBM roofline",
            "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": ("configs[2]: " if args.workload == "mesh" else "configs[1]: ")
                + f"{args.width}x{args.height} adaptive RK (Cash-Karp), accretion disk + adaptive background grid"
                + (" + BVH mesh (327680-triangle icosphere OBJ) at (-10,0,30)" if args.workload == "mesh" else ""),
                "ladder": [list(s) for s in cfg.sizes()], "crop": [int(cfg.crop_x), int(cfg.crop_y)],
                "step_size": 0.15, "max_iterations": args.max_iterations, "angle_division_threshold": 0.02,
                "parallelism": f"row-tiled x{world}, stripes of {args.stripe_rows} rows, gather to rank 0" if world > 1 else "single GPU",
                "frames_in_flight": args.frames_in_flight, "readback_to_host": bool(args.readback),
            },
            "roofline": {
                "bound": "hbm", "kernel": "trace_kernel", "achieved": round(achieved_gbs, 4), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved_gbs / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_note": traffic_note,
                "avg_launch_ms": round(avg_ms, 5), "launches": int(tm.trace_launches),
                "algorithmic_bytes_per_launch": round(bytes_per_launch, 1),
                "concurrency": f"{args.frames_in_flight} frames in flight: launch durations in the timed region include the kernels they overlap with",
                "isolated": iso_d,
                "note": "VALU-bound f32 ODE march (SURVEY.md F8): algorithmic HBM traffic is tiny; see `valu` for the binding roofline",
            },
            "valu": {"achieved": round(valu_tflops, 4), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(valu_tflops / VALU_PEAK_TFLOPS, 6),
                     "algorithmic_flops_per_frame": flops_per_frame,
                     "note": "algorithmic flops per frame / wall time per frame (device-level rate over the timed region); peak = FMA rate, "
                             "the numerics contract forbids contraction, so the reachable ceiling for this instruction mix is ~1/2 of it"},
            "pass_ms": {"event_total": round(tm.total_ms / frames, 5), "trace": round(tm.trace_ms / frames, 5),
                        "classify": round(tm.classify_ms / frames, 5),
                        "level_trace": [round(tm.level_trace_ms[i] / frames, 5) for i in range(args.levels)]},
            "sky_resolve": sky_d,
            "counters": counters,
            "algorithmic_bytes_per_frame": algorithmic_bytes_frame(counters),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, tex, cam, bh, det, model)
        print(json.dumps(out), flush=True)
    rp.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
