cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6_cont
timeout 900 python -m pytest tests -m gpu -x -q -k "continuous or split_mid_interval_flush or split_freq_map" > gpurun_out/r6_cont/tests_new.log 2>&1; echo "rc $?" >> gpurun_out/r6_cont/tests_new.log; tail -5 gpurun_out/r6_cont/tests_new.log
bash tools/gpu/r6_env_ab.sh r6_cont "" "SMST_NO_CONTINUOUS=1"
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r6_cont/tests_all.log 2>&1; echo "rc $?" >> gpurun_out/r6_cont/tests_all.log; tail -5 gpurun_out/r6_cont/tests_all.log
