# extra SQ / SQC counters for the dense trace kernels (PMC passes only, kernel trace only)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
O=$ROOT/gpurun_out/s54; mkdir -p $O
BASE="--steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs"
i=0
for W in "" "--integrator euler"; do
  tag=rk; [ -n "$W" ] && tag=euler
  for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES" \
             "SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES" \
             "SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" \
             "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_WAVE_CYCLES"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o bench -- python $ROOT/bench.py $BASE $W > $O/p$i.log 2>&1
    python - "$O/p$i" "$tag" <<'P' >> $O/extra_counters.txt
import sys, glob, csv, collections
d, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "trace_kernel" not in k: continue
        k = k.split("(")[0].replace("void bhray::", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen: seen.add(key); n[k] += 1
for k in acc:
    if n[k] < 50: continue
    print(tag, k, "launches", n[k], {c: round(v / n[k], 1) for c, v in sorted(acc[k].items())})
P
    rm -rf $O/p$i
  done
done
cat $O/extra_counters.txt
