"""Timing experiment (results wrong by construction): the staged producers of kVocoder share window buffers in pairs (or fours),
so the workgroup asks for 25 (38) KB less LDS -- does an FFT workgroup (26 KB) then run beside it, and what does the step gain?
SMST_PATCH_SBUFS = number of distinct buffers (default 4; the product has 8)."""
import os
import sys
n = int(os.environ.get("SMST_PATCH_SBUFS", "4"))
p = sys.argv[1] + '/smst_kernels.hip'
s = open(p).read()
for old, new in (("(size_t)pIndex*G::ROWS*G::ROWLEN;", "(size_t)(pIndex %% %d)*G::ROWS*G::ROWLEN;" % n),
                 ("(size_t)kVocStagedProducers*G::ROWS*G::ROWLEN*sizeof(float2);", "(size_t)%d*G::ROWS*G::ROWLEN*sizeof(float2);" % n)):
    assert s.count(old) == 1, old
    s = s.replace(old, new)
open(p, 'w').write(s)
