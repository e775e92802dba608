cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s27; O=gpurun_out/s27
V=$GRAFT_REPO_ROOT/profiles/variants
for lib in "" $V/libbhray_l_flat1.so $V/libbhray_l_flat16_8.so $V/libbhray_l_park5.so $V/libbhray_l_w4.so $V/libbhray_l_w4park.so; do
  echo "== ${lib##*/}"; BHRAY_LIB=$lib python profiles/jobs/r5_mesh_latency.py 2>&1 | grep -E "^mesh +spec 2|^culled +spec 2"
done 2>&1 | tee $O/latency_variants.txt
echo "== longest traversal (counters.max_ray_iterations repurposed)"; BHRAY_LIB=$V/libbhray_l_longest.so python profiles/jobs/r5_mesh_latency.py 2>&1 | grep -E "level" | head -4 | tee -a $O/latency_variants.txt
