# usage: r6_modes2.sh "<modes>" [bench args]: the headline (or another config) with the experiments variant in the given SMST_DEBUG_MODEs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/modes2
export SMST_LIBRARY_ALLOW_MISSING=1 SMST_LIBRARY=$GRAFT_REPO_ROOT/signalsmith-stretch_amd/variants/experiments.so
MODES=$1; shift
for mode in $MODES; do
  SMST_DEBUG_MODE=$mode timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-self-check --no-other-configs "$@" > gpurun_out/modes2/mode$mode.json 2> gpurun_out/modes2/mode$mode.err
  python -c "
import json
d = json.loads(open('gpurun_out/modes2/mode$mode.json').read().strip().splitlines()[-1]); r = d['roofline']
print('mode $mode: %.3f ms/step  alone %s' % (d['ms_per_step'], {k: v for k, v in r['kernel_ms_per_step_alone'].items() if v > 0.3}))" || tail -2 gpurun_out/modes2/mode$mode.err
done
