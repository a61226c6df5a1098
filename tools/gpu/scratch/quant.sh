R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { env $2 timeout 200 python bench.py --streams $1 --steps 8 --warmup 2 --no-cpu-baseline --no-self-check --no-serial-pass 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('streams $1 [$2]: ms/step', d['ms_per_step'], 'per-stream-rate vs 256:', round(d['value']/$1, 1))"; }
run 256 ""
run 288 ""
run 288 SMST_SUB_STREAMS=144
run 288 SMST_SUB_STREAMS=96
run 320 ""
run 320 SMST_SUB_STREAMS=160
run 384 ""
run 384 SMST_SUB_STREAMS=192
run 384 SMST_SUB_STREAMS=128
run 512 ""
