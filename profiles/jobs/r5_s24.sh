cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s24; O=gpurun_out/s24
export GPU_MAX_HW_QUEUES=64 PYTHONPATH=$GRAFT_REPO_ROOT
PREV=$GRAFT_REPO_ROOT/profiles/variants/libbhray_prev.so
timeout 1500 python -m pytest tests -x -q -m gpu -k "mesh or bvh or config2 or stack or chain or depth or model or fuzz" 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/mesh_tests.txt
one() {  # label lib workload-args
  for cfg in "--steps 20 --warmup 5" "--steps 400 --warmup 32"; do
    BHRAY_LIB=$2 timeout 300 python bench.py $cfg $3 --no-extra-legs --no-cpu-baseline --min-seconds 2 --sustained-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$3', d['steps'], d['value'], d['ms_per_step'])"
  done
}
for r in 1 2 3; do
  one prev $PREV "--workload mesh"
  one dense_mesh "" "--workload mesh"
done 2>&1 | tee $O/ab_mesh_dense.txt
for r in 1 2; do
  one prev $PREV "--workload mesh --integrator euler"
  one dense_mesh "" "--workload mesh --integrator euler"
done 2>&1 | tee -a $O/ab_mesh_dense.txt
for lib in $PREV ""; do
BHRAY_LIB=$lib python - <<'P'
import time, os, argparse, bhusie_amd as B
import bench
a = argparse.Namespace(workload="mesh", integrator="rk", max_iterations=500, bvh="median")
tex, cam, bh, det, model = bench.build_scene(a)
cfg = B.ladder_for_frame((1920, 1080), 3, 4)
for fif in (1, 2):
    rp = B.RayPass(cfg, device=0, frames_in_flight=fif, speculative_levels=2)
    rp.set_textures(*tex); rp.upload_model(model)
    rp.set_uniforms(cam.uniform(), bh.uniform(), det.uniform())
    ts = []
    for i in range(14):
        t0 = time.perf_counter(); rp.render(); rp.sync(); ts.append(time.perf_counter() - t0)
    print(os.environ.get("BHRAY_LIB", "default")[-20:], "mesh rk fif", fif, "one frame at a time %.4f ms" % (sorted(ts[3:])[5] * 1e3))
    rp.close()
P
done 2>&1 | grep -E "one frame|Error|error|Traceback" | tee -a $O/ab_mesh_dense.txt
