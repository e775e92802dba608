cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s51; O=gpurun_out/s51
V=$GRAFT_REPO_ROOT/profiles/variants
for r in 1 2; do for lib in "" $V/libbhray_w5.so; do
  echo "== ${lib##*/}"; BHRAY_LIB=$lib python profiles/jobs/r5_lat.py 2>&1 | grep wall | grep disk
done; done 2>&1 | tee $O/latency_w5.txt
