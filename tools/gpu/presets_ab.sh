#!/bin/bash
# the preset table under different engine switches, same box: tools/gpu/presets_ab.sh "name|ENV=.." ...
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for spec in "$@"; do
  name=${spec%%|*}; envs=${spec#*|}
  env $envs SMST_LIBRARY_ALLOW_MISSING=1 timeout 300 python tools/bench_presets.py ${PRESET_ARGS:-} 2>/dev/null > gpurun_out/presets_$name.json
  python - <<PY
import json
for r in json.load(open("gpurun_out/presets_$name.json"))["rows"]:
    print("%-12s %-8s %6d  %6.0f Ms/s  %6.2f ms  %s" % ("$name", r["preset"], r["sample_rate"], r["fast_fft"]["Msamples_s"], r["fast_fft"]["ms_per_step"], r["fast_fft"]["alone_ms"]))
PY
done
