# usage: r6_tests.sh <tag> [pytest -k expression]: the GPU suite (or a part of it) with timing of the slowest tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$1
if [ -n "$2" ]; then K="-k"; fi
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 $K "$2" > gpurun_out/$1/gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/$1/gpu_tests.log
tail -25 gpurun_out/$1/gpu_tests.log
