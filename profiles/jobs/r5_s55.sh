cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s55
for i in 1 2; do timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -2; done | tee gpurun_out/s55/repeat_full.txt
