cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s3; O=gpurun_out/s3
export GPU_MAX_HW_QUEUES=64 TMPDIR=/tmp
for a in "8 5 6" "4 5 6" "2 5 6" "8 20 6" "8 10 6" "8 2 6" "8 5 2"; do python profiles/jobs/r5_one_block.py $a 2>&1 | grep "^N" | tail -2; done
rocprofv3 --kernel-trace --output-format csv -d $O/prof -o p8 -- python profiles/jobs/r5_one_block.py 8 5 6 > $O/one_block.log 2>&1
f=$(find $O/prof -name '*kernel_trace.csv' | head -1)
python profiles/jobs/r5_block_dump.py $f 70 > $O/last_block.txt
wc -l $O/last_block.txt; tail -3 $O/one_block.log
rm -rf $O/prof
