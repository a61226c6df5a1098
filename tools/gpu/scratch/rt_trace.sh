cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/rt_trace
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rtprof -o rt -f csv -- python $R/tools/bench_realtime.py --streams 4096 --quanta 220 > $R/gpurun_out/rt_trace/bench.json 2> $R/gpurun_out/rt_trace/err.log
find /tmp/rtprof -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/rt_trace/kernel_stats.csv \;
head -40 $R/gpurun_out/rt_trace/kernel_stats.csv | cut -c1-200
