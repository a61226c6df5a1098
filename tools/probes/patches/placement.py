"""Wave placement experiments of the ALIGNED recurrence kernel (results unchanged, only which SIMD a role lands on).
PLACE=p2222: two producers on every SIMD (waves 8 and 12 beside the recurrence wave instead of 9 and 10)
PLACE=first3: producer 0 (the heaviest) on the two-producer SIMD
PLACE=w11:   the writer on wave 11 (SIMD 3, which holds two producers) instead of wave 4 (the recurrence wave's SIMD)"""
import os
import sys
SRC = sys.argv[1]
mode = os.environ["PLACE"]
p = SRC + '/smst_kernels.hip'
s = open(p).read()
old = "		if (STAGED) pIndex = (wave & 3) ? ((wave < 8) ? pIndex : ((wave == 9) ? 6 : ((wave == 10) ? 7 : NP))) : NP;"
assert s.count(old) == 1
if mode == "p2222":
    s = s.replace(old, "		if (STAGED) pIndex = (wave < 8) ? ((wave & 3) ? pIndex : NP) : ((wave == 8) ? 6 : ((wave == 12) ? 7 : NP));")
elif mode == "first3":
    # the producer that owns rows 0..7 (carried taps, FOLD0: the heaviest) on the SIMD that holds only two producers (waves 3, 7)
    s = s.replace(old, old + "\n		if (STAGED && ALIGNED && pIndex < NP) pIndex = (pIndex == 0) ? 2 : ((pIndex == 2) ? 0 : pIndex);")
elif mode == "w11":
    a = s.index("template <int CH, bool PLAIN, int L, bool STAGED, bool ROTL = false, bool ACROSS = false, bool ALIGNED = false>")
    b = s.index("constexpr int kVocNBlockSteps")
    body = s[a:b]
    assert body.count("		if (wave == 4) {") == 1
    body = body.replace("		if (wave == 4) {", "		if (wave == 11) {")
    body = body.replace(old, "		if (STAGED) pIndex = ((wave & 3) && wave != 11) ? ((wave < 8) ? pIndex : ((wave == 9) ? 6 : ((wave == 10) ? 7 : NP))) : NP;")
    s = s[:a] + body + s[b:]
open(p, 'w').write(s)
