cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s23; O=gpurun_out/s23
export GPU_MAX_HW_QUEUES=64
V=$GRAFT_REPO_ROOT/profiles/variants
one() {  # label lib workload-args
  for cfg in "--steps 20 --warmup 5" "--steps 400 --warmup 32"; do
    BHRAY_LIB=$2 timeout 300 python bench.py $cfg $3 --no-extra-legs --no-cpu-baseline --min-seconds 2 --sustained-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$3', d['steps'], d['value'], d['ms_per_step'])"
  done
}
for r in 1 2 3; do
  one default "" "--workload mesh"
  one park6 $V/libbhray_m_park6.so "--workload mesh"
  one park6ww $V/libbhray_m_park6ww.so "--workload mesh"
done 2>&1 | tee $O/ab_mesh_park.txt
cat > /tmp/lat.py <<'P'
import time, os, bhusie_amd as B
from tests import common as T
from bhusie_amd import assets
import tempfile
cfg = B.ladder_for_frame((1920, 1080), 3, 4)
obj = assets.icosphere_mesh_obj(7, radius=8.0, bump=0.15, seed=3)
f = tempfile.NamedTemporaryFile("w", suffix=".obj", delete=False); f.write(obj); f.close()
model = B.load_model(f.name); os.unlink(f.name)
rp = B.RayPass(cfg, device=0, frames_in_flight=1, speculative_levels=2)
rp.set_textures(*T.textures(small=False)); rp.upload_model(model); rp.set_uniforms(*T.uniforms(integration_method=1, model_count=1))
ts = []
for i in range(14):
    t0 = time.perf_counter(); rp.render(); rp.sync(); ts.append(time.perf_counter() - t0)
print("LAT", os.path.basename(os.environ.get("BHRAY_LIB", "default")), "mesh one frame at a time %.4f ms" % (sorted(ts[3:])[5] * 1e3), flush=True)
rp.close()
P
for lib in "" $V/libbhray_m_park6.so $V/libbhray_m_park6ww.so; do BHRAY_LIB=$lib python /tmp/lat.py 2>&1 | grep "^LAT"; done | tee -a $O/ab_mesh_park.txt
