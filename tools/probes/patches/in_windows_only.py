"""Timing experiment (results wrong by construction): the staged producers fetch only the IN windows from memory -- the PV and ROT
pieces all read one fixed 16-byte location (an L1 hit, one cache line per instruction).  What kVocoder would cost if the previous-hop
spectrum came from the neighbouring row's IN window and the rotation table sat in LDS."""
import sys
p = sys.argv[1] + '/smst_kernels.hip'
s = open(p).read()
for old, new in (("	int pbin[G::LOADS], plds[G::LOADS];\n", "	int pbin[G::LOADS], plds[G::LOADS], pmul[G::LOADS];\n"),
                 ("		psrc[i] = ok ? src : d.rot;\n		pbin[i] = rel - lag*row;\n", "		psrc[i] = (ok && j < CH*G::PIN) ? src : d.rot;\n		pbin[i] = (j < CH*G::PIN) ? rel - lag*row : 0;\n		pmul[i] = (j < CH*G::PIN) ? BS : 0;\n"),
                 ("				const int sb = BS*n + pbin[i];\n				v[i] = *reinterpret_cast<const float4 *>(psrc[i] + sb);", "				const int sb = pmul[i]*n + pbin[i];\n				v[i] = *reinterpret_cast<const float4 *>(psrc[i] + sb);")):
    assert old in s, old[:50]
    s = s.replace(old, new)
open(p, 'w').write(s)
