cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s42
BHRAY_LIB=$GRAFT_REPO_ROOT/profiles/variants/libbhray_x_flatclock.so python profiles/jobs/r5_flatclock.py 2>&1 | grep -E "level|Error|error" | tee gpurun_out/s42/flatclock.txt
