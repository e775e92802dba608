cd $GRAFT_REPO_ROOT
for v in h0t0 h1t0 h0t1; do for bpc in 4 2 1; do
BHRAY_TRACE_BLOCKS_PER_CU=$bpc BHRAY_LIB=$GRAFT_REPO_ROOT/profiles/variants/libbhray_$v.so timeout 200 python profiles/jobs/lat2.py 2>/dev/null
done; done
for bpc in 4 2 1; do BHRAY_TRACE_BLOCKS_PER_CU=$bpc timeout 200 python profiles/jobs/lat2.py 2>/dev/null; done
