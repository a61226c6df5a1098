# usage: r6_run2.sh <tag> [bench args]: the continuous-wavefront tests, then (only if they pass) the same-box A/B of SMST_CONTINUOUS=1 against the default (tile by tile)
cd $GRAFT_REPO_ROOT
TAG=${1:-r6_cont}; shift
mkdir -p gpurun_out/$TAG
timeout 900 python -m pytest tests -m gpu -x -q -k "continuous" > gpurun_out/$TAG/tests_new.log 2>&1; RC=$?; echo "rc $RC" >> gpurun_out/$TAG/tests_new.log; tail -3 gpurun_out/$TAG/tests_new.log
if [ $RC -ne 0 ]; then exit 1; fi
bash tools/gpu/r6_env_ab.sh $TAG "SMST_CONTINUOUS=1" "" "$@"
