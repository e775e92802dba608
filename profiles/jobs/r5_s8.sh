cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s8; O=gpurun_out/s8
export GPU_MAX_HW_QUEUES=64
timeout 900 python -m pytest tests/test_gpu_pair.py tests/test_gpu_repartition.py -x -q -m gpu 2>&1 | tail -15
one() {  # label lib workload-args
  for cfg in "--steps 20 --warmup 5" "--steps 400 --warmup 32"; do
    BHRAY_LIB=$2 timeout 300 python bench.py $cfg $3 --no-extra-legs --no-cpu-baseline --min-seconds 2 --sustained-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$3', d['steps'], d['value'], d['ms_per_step'])"
  done
}
for r in 1 2; do
  one default "" "--integrator euler"
  one pair_euler_s2 $GRAFT_REPO_ROOT/bhusie_amd/libbhray_pair.so "--integrator euler"
  one pair_euler_v2 $GRAFT_REPO_ROOT/profiles/variants/libbhray_pairs2.so "--integrator euler"
  one default "" ""
  one pair_rk_v2 $GRAFT_REPO_ROOT/bhusie_amd/libbhray_pair.so ""
  one pair_rk_s2 $GRAFT_REPO_ROOT/profiles/variants/libbhray_pairs2.so ""
done 2>&1 | tee $O/ab_pair.txt
for lib in "" $GRAFT_REPO_ROOT/bhusie_amd/libbhray_pair.so; do
BHRAY_LIB=$lib python - <<'P'
import time, bhusie_amd as B
from tests import common as T
cfg = B.ladder_for_frame((1920, 1080), 3, 4)
for m in (0, 1):
    rp = B.RayPass(cfg, device=0, frames_in_flight=1, speculative_levels=2)
    rp.set_textures(*T.textures(small=False)); rp.set_uniforms(*T.uniforms(integration_method=m))
    import os
    ts = []
    for i in range(14):
        t0 = time.perf_counter(); rp.render(); rp.sync(); ts.append(time.perf_counter() - t0)
    print(os.environ.get("BHRAY_LIB", "default")[-20:], "method", m, "one frame at a time %.4f ms" % (sorted(ts[3:])[5] * 1e3))
    rp.close()
P
done
