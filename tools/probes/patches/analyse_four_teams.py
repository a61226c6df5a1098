"""Timing experiment (results wrong by construction): kAnalyseTeams with FOUR teams of 256 threads per workgroup -- the fourth team
shares the first one's transform buffer (there is no LDS for a fourth beside the full tables) -- to see what sixteen waves per CU at a
128-register budget would buy before building the lean-table form that would make it legal.  Run the bench with --no-self-check."""
import sys
p = sys.argv[1] + '/smst_kernels.hip'
s = open(p).read()
def rep(old, new, count=1):
    global s
    assert s.count(old) == count, (s.count(old), old)
    s = s.replace(old, new)
i = s.index('void kAnalyseTeams(')
j = s.index('void kSynthFast(')
k = s[i:j]
old = "float2 *lds = reinterpret_cast<float2 *>(twBLds + 8*R3) + (size_t)team*(H + H/16);"
assert k.count(old) == 1
k = k.replace(old, "float2 *lds = reinterpret_cast<float2 *>(twBLds + 8*R3) + (size_t)(team % 3)*(H + H/16);")
old = "(size_t)TEAMS*(H + H/16));"
assert k.count(old) == 1
k = k.replace(old, "(size_t)3*(H + H/16));")
s = s[:i] + k + s[j:]
rep("hipLaunchKernelGGL((kAnalyseTeams<12, 3, true>), dim3(wgs), dim3(768)", "hipLaunchKernelGGL((kAnalyseTeams<12, 4, true>), dim3(wgs), dim3(1024)")
open(p, 'w').write(s)
