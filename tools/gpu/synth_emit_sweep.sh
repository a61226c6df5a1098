#!/bin/bash
# where does kSynthEmitTeams pay?  stream sweep and preset table with the one-kernel form forced (2), off (0) and as the launcher decides (1)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/se_sweep
for mode in 0 2 1; do
  SMST_SYNTH_EMIT=$mode timeout 600 python tools/bench_sweep.py --sizes ${SIZES:-64,128,192,256,288,320,384,448,512,640,768,1024} --steps 4 2>/dev/null > gpurun_out/se_sweep/sweep_$mode.json
  python - <<PY
import json
d = json.load(open("gpurun_out/se_sweep/sweep_$mode.json"))
print("SYNTH_EMIT=$mode", " ".join("%d:%.0f" % (r["streams"], r["Msamples_s"]) for r in d["rows"]))
PY
done
bash tools/gpu/presets_ab.sh "two_kernels|SMST_SYNTH_EMIT=0" "auto|SMST_X=0"
