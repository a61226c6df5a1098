# the dense API fuzz of round 5's campaign ranges (a sample) with the 12-perturbation classification, and the outlier diagnosis of the six walks round 5 left open
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6_fuzz
for spec in "1017 plain" "1017 split" "1001 plain" "1037 plain" "2195 cheaper48" "2066 cheaper48" "2095 cheaper48"; do
  echo "== outlier $spec" >> gpurun_out/r6_fuzz/outliers.txt
  timeout 300 python tools/diag/fuzz_outlier.py $spec 2>&1 | grep -v "amdgpu.ids" >> gpurun_out/r6_fuzz/outliers.txt
done
( timeout 900 python tools/diag/fuzz_dense.py 1000 1150 plain 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r6_fuzz/plain_1000_1150.txt ) &
( timeout 900 python tools/diag/fuzz_dense.py 1000 1150 split 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r6_fuzz/split_1000_1150.txt ) &
( timeout 900 python tools/diag/fuzz_dense.py 2000 2100 cheaper48 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r6_fuzz/cheaper48_2000_2100.txt ) &
( timeout 900 python tools/diag/fuzz_dense.py 2150 2250 cheaper48 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r6_fuzz/cheaper48_2150_2250.txt ) &
wait
tail -n 2 gpurun_out/r6_fuzz/*.txt
