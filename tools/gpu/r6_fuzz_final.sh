# the dense API fuzz on the round's last library: fresh seed ranges (400 walks in four processes), 12-perturbation classification
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6_fuzz_final
( timeout 1200 python tools/diag/fuzz_dense.py 3000 3100 plain 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r6_fuzz_final/plain_3000_3100.txt ) &
( timeout 1200 python tools/diag/fuzz_dense.py 3100 3200 split 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r6_fuzz_final/split_3100_3200.txt ) &
( timeout 1200 python tools/diag/fuzz_dense.py 3200 3300 cheaper48 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r6_fuzz_final/cheaper48_3200_3300.txt ) &
( timeout 1200 python tools/diag/fuzz_dense.py 3300 3400 plain 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r6_fuzz_final/plain_3300_3400.txt ) &
wait
tail -n 3 gpurun_out/r6_fuzz_final/*.txt
