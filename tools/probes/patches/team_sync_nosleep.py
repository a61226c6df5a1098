"""Timing experiment: the team barrier polls its counter without s_sleep (all team kernels)."""
import sys
p = sys.argv[1] + '/smst_kernels.hip'
s = open(p).read()
old = "< *generation) __builtin_amdgcn_s_sleep(1);"
assert s.count(old) == 1
s = s.replace(old, "< *generation) {}")
open(p, 'w').write(s)
