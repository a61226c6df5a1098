#!/bin/bash
# Counter evidence for BASELINE config 5 (8 channels, 96 kHz, presetCheaper): rocprofv3's counter collection crashed inside the profiled
# process on this workload in round 4 (three attempts, any stream count).  This script narrows the collection: ONE counter per pass, only the
# kernels that match a regex (--kernel-include-regex), a subset of the streams, one step -- and records which combinations survive.
# usage: tools/prof/pmc_config5.sh [tag] [streams]      output: gpurun_out/<tag>/
TAG=${1:-pmc5}
STREAMS=${2:-64}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --config 5 --streams $STREAMS --steps 1 --warmup 1 --no-cpu-baseline --no-serial-pass --no-self-check"
echo "streams $STREAMS" > $OUT/outcomes.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -f csv -- $CMD > $OUT/trace.log 2>&1; echo "trace rc $?" >> $OUT/outcomes.txt
for K in kVocoderN kAnalyseFast kSynthFast kEmit kFeedScanA kCarry; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "$K" -d $OUT/${K}_$C -o c -f csv -- $CMD > $OUT/${K}_$C.log 2>&1
    echo "$K $C rc $? $(ls $OUT/${K}_$C/*counter_collection.csv 2>/dev/null | wc -l) csv" >> $OUT/outcomes.txt
  done
done
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-include-regex kVocoderN -d $OUT/kVocoderN_SQ -o c -f csv -- $CMD > $OUT/kVocoderN_SQ.log 2>&1
echo "kVocoderN SQ rc $? $(ls $OUT/kVocoderN_SQ/*counter_collection.csv 2>/dev/null | wc -l) csv" >> $OUT/outcomes.txt
cat $OUT/outcomes.txt
python3 - "$OUT" <<'PY'
import collections, csv, glob, os, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(out, "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(path)):
        if "smst::" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("smst::", "")
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(acc.items()):
    print(k[:60], {n: (len(v), sum(v)/len(v)) for n, v in c.items()})
PY
find $OUT -name "*_kernel_trace.csv" -size +8M -delete
