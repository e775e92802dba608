cd $GRAFT_REPO_ROOT
echo "copy streams"; timeout 300 python profiles/jobs/handoff_diag.py 2>&1 | grep '^{' | head -3
echo "on slot stream"; BHRAY_COPY_ON_SLOT=1 timeout 300 python profiles/jobs/handoff_diag.py 2>&1 | grep '^{'
echo "on slot stream, HSA_ENABLE_SDMA=0"; HSA_ENABLE_SDMA=0 BHRAY_COPY_ON_SLOT=1 timeout 300 python profiles/jobs/handoff_diag.py 2>&1 | grep '^{' | head -3
