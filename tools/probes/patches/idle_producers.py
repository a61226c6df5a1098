"""Timing experiment (results wrong by construction): the staged producers neither stage nor compute -- they only hand zero records to the
recurrence wave.  What is left is the recurrence wave and the writer: the floor of kVocoder."""
import sys
p = sys.argv[1] + '/smst_kernels.hip'
s = open(p).read()
for old, new in (("	if (n < totalBlocks) { issue(n); if (it == 0) issueCarried(n); }\n", ""),
                 ("		park(n);\n", ""),
                 ("		if (n + NPB < totalBlocks) { issue(n + NPB); if (it == 0) issueCarried(n + NPB); }\n", ""),
                 ("		if (row < nh && b >= 0 && b < M && d.debugMode != 1) {\n			// same arithmetic as computeRecord<CH, true, false, false>", "		if (false) {\n			// same arithmetic as computeRecord<CH, true, false, false>")):
    assert old in s, old[:50]
    s = s.replace(old, new)
open(p, 'w').write(s)
