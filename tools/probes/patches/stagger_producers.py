"""Experiment on the sliding-window producers (branch exp/sliding-producers): a one-off phase offset between the eight producer waves,
so that their load instructions do not reach the texture-address pipeline in bursts of eight."""
import os
import sys
N = int(os.environ.get("SMST_PATCH_STAGGER", "4"))
p = sys.argv[1] + '/smst_kernels.hip'
s = open(p).read()
old = "	for (int n = 0; n < totalBlocks; n += 2) { // totalBlocks is a multiple of 8\n		blockStep(setA, n);"
assert old in s
s = s.replace(old, "	for (int i = 0; i < (pIndex & 7)*%d; ++i) __builtin_amdgcn_s_sleep(8);\n" % N + old)
open(p, 'w').write(s)
