"""Experiment: the register-blocked FFT kernels at SIX waves per SIMD (<= 85 VGPRs, six 26-KB workgroups per CU) instead of four."""
import sys
p = sys.argv[1] + '/smst_kernels.hip'
s = open(p).read()
old = "__global__ __launch_bounds__(16*R3 > 256 ? 16*R3 : 256) __attribute__((amdgpu_waves_per_eu(4, 4))) void kAnalyseFast("
assert s.count(old) == 1
s = s.replace(old, "__global__ __launch_bounds__(16*R3 > 256 ? 16*R3 : 256) __attribute__((amdgpu_waves_per_eu(6, 6))) void kAnalyseFast(")
old = "__global__ __launch_bounds__(16*R3 > 256 ? 16*R3 : 256) void kSynthFast("
assert s.count(old) == 1
s = s.replace(old, "__global__ __launch_bounds__(16*R3 > 256 ? 16*R3 : 256) __attribute__((amdgpu_waves_per_eu(6, 6))) void kSynthFast(")
open(p, 'w').write(s)
