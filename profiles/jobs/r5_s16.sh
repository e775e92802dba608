cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s16; O=gpurun_out/s16
export GPU_MAX_HW_QUEUES=64
EC=$GRAFT_REPO_ROOT/profiles/variants/libbhray_eulerc.so
BHRAY_LIB=$EC timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
one() {  # label lib workload-args
  for cfg in "--steps 20 --warmup 5" "--steps 400 --warmup 32"; do
    BHRAY_LIB=$2 timeout 300 python bench.py $cfg $3 --no-extra-legs --no-cpu-baseline --min-seconds 2 --sustained-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$3', d['steps'], d['value'], d['ms_per_step'])"
  done
}
for r in 1 2 3; do
  one default "" "--integrator euler"
  one euler_const_culls $EC "--integrator euler"
done 2>&1 | tee $O/ab_euler_const_culls.txt
for lib in "" $EC; do
BHRAY_LIB=$lib python - <<'P'
import time, os, bhusie_amd as B
from tests import common as T
cfg = B.ladder_for_frame((1920, 1080), 3, 4)
rp = B.RayPass(cfg, device=0, frames_in_flight=1, speculative_levels=2)
rp.set_textures(*T.textures(small=False)); rp.set_uniforms(*T.uniforms(integration_method=0))
ts = []
for i in range(14):
    t0 = time.perf_counter(); rp.render(); rp.sync(); ts.append(time.perf_counter() - t0)
print(os.environ.get("BHRAY_LIB", "default")[-20:], "euler one frame at a time %.4f ms" % (sorted(ts[3:])[5] * 1e3))
rp.close()
P
done 2>&1 | grep "one frame" | tee -a $O/ab_euler_const_culls.txt
