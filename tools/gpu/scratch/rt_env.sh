R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for round in 1 2; do
for v in "" "SMST_SYNTH_EMIT=2"; do
  for preset in default cheaper; do
    env $v timeout 300 python tools/bench_realtime.py --preset $preset --streams 256 1024 4096 --quanta 500 > /tmp/o.json 2>/tmp/e.log
    python -c "
import json
d = json.loads(open('/tmp/o.json').read().strip().splitlines()[-1])
print('[$v] $preset:', [(r['streams'], r['median_ms'], r['p99_ms']) for r in d['rows']])"
  done
done
done
