"""Timing experiment (results wrong by construction): the staged producers never issue their window loads -- what kVocoder
costs if the staging traffic were free."""
import sys
p = sys.argv[1] + '/smst_kernels.hip'
s = open(p).read()
for old, new in (("	if (n < totalBlocks) { issue(n); if (it == 0) issueCarried(n); }\n", "	for (int i = 0; i < G::LOADS; ++i) v[i] = make_float4(0.1f, 0.2f, 0.3f, 0.4f);\n"),
                 ("		if (n + NPB < totalBlocks) { issue(n + NPB); if (it == 0) issueCarried(n + NPB); }\n", "")):
    assert old in s
    s = s.replace(old, new)
open(p, 'w').write(s)
