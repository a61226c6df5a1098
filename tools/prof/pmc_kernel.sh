#!/bin/bash
# Memory-path counters of ONE kernel (regex) of a bench workload, one small counter group per pass (rocprofv3 --pmc, --kernel-trace only):
# address translation (UTCL1), L1 (TCP) accesses / misses / latency, texture addresser stalls, L2 hits.
# (the TA_* counters are left out: a pass with TA_TA_BUSY / TA_*_STALLED_BY_TC hangs until the timeout on this pool)
# usage: tools/prof/pmc_kernel.sh <tag> <kernel regex> <bench args ...>     output: gpurun_out/<tag>/summary.txt
TAG=$1; KERNEL=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-serial-pass --no-self-check $*"
i=0
for GROUP in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
             "TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum" \
             "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES TA_FLAT_READ_WAVEFRONTS_sum"; do
  i=$((i + 1))
  timeout 300 rocprofv3 --kernel-trace --pmc $GROUP --kernel-include-regex "$KERNEL" -d $OUT/g$i -o c -f csv -- $CMD > $OUT/g$i.log 2>&1
  echo "group $i rc $? ($GROUP)" >> $OUT/outcomes.txt
done
python3 - "$OUT" <<'PY' | tee $OUT/summary.txt
import collections, csv, glob, os, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(out, "g*", "*counter_collection.csv")):
    for r in csv.DictReader(open(path)):
        if "smst::" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("smst::", "")
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(acc.items()):
    print(k[:70])
    for n, v in sorted(c.items()):
        print("   %-50s launches %3d  mean %.4g" % (n, len(v), sum(v)/len(v)))
PY
cat $OUT/outcomes.txt
find $OUT -name "*_kernel_trace.csv" -size +8M -delete
