"""TEST/BENCH INFRASTRUCTURE ONLY (oracle): times the CPU reference (oracle/_ref = the unmodified reference header
+ the L1 restatement, `g++ -O3`) on the same synthetic workload bench.py gives the GPU: stereo 48 kHz streams,
presetDefault, 1.5x.  One process per core; each prints {"streams", "seconds_audio", "process_s"}.
usage: python oracle/cpu_baseline.py <streams> <seconds> <stretch> <first_stream_index>"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))


def main():
    import ref_oracle
    from conftest import synth_input
    streams, seconds, stretch, first = int(sys.argv[1]), float(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
    sr, C = 48000, 2
    n = int(seconds*sr)
    nout = int(round(n*stretch))
    total = 0.0
    for s in range(streams):
        x = synth_input(first + s, C, n, sr)
        r = ref_oracle.RefStretch()
        r.presetDefault(C, sr)
        t0 = time.perf_counter()
        r.process(x, nout)
        total += time.perf_counter() - t0
    print(json.dumps(dict(streams=streams, seconds_audio=seconds, samples=streams*C*(n + nout), process_s=total)))


if __name__ == "__main__":
    main()
