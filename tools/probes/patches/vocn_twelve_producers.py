"""Experiment: kVocoderN with 12 producer waves, none on the recurrence wave's SIMD (waves w and w + 4 share a SIMD: waves 8 and 12 sit
idle instead of producing) -- are the passes of the producers beside the recurrence wave the late ones?  Results stay correct."""
import sys
p = sys.argv[1] + '/smst_kernels.hip'
s = open(p).read()
i = s.index('void kVocoderN(')
j = s.index('// K3 for single-hop tiles')
k = s[i:j]
old = "constexpr int NP = kVocWaves - 2;"
assert k.count(old) == 1
k = k.replace(old, "constexpr int NP = 12;")
old = "const int pIndex = wave - 1 - (wave > 4);"
assert k.count(old) == 1
k = k.replace(old, "if ((wave & 3) == 0) return; // (wave 4, the writer, has left above)\n\t\tconst int pIndex = (wave >> 2)*3 + (wave & 3) - 1;")
s = s[:i] + k + s[j:]
open(p, 'w').write(s)
