cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s29; O=gpurun_out/s29
V=$GRAFT_REPO_ROOT/profiles/variants
BHRAY_LIB=$V/libbhray_l_longest.so python profiles/jobs/r5_longest.py 2>&1 | grep -E "longest|level|Error" | tee $O/longest.txt
