cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/s53
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(SQ|SQC|TCP|TCC|TA|TD|GRBM|SPI)_[A-Z0-9_]+" | sort -u > $GRAFT_REPO_ROOT/gpurun_out/s53/counters.txt
wc -l $GRAFT_REPO_ROOT/gpurun_out/s53/counters.txt
