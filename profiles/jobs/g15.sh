cd $GRAFT_REPO_ROOT
timeout 200 python profiles/jobs/lat3.py 2>/dev/null | grep '^{'
for v in c2 c4 c4b32 b32; do BHRAY_LIB=$GRAFT_REPO_ROOT/profiles/variants/libbhray_$v.so timeout 200 python profiles/jobs/lat3.py 2>/dev/null | grep '^{'; done
timeout 200 python profiles/jobs/lat3.py 2>/dev/null | grep '^{'
