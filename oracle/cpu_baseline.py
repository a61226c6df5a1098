"""TEST/BENCH INFRASTRUCTURE ONLY (oracle): times the CPU reference (oracle/_ref = the unmodified reference header
+ the L1 restatement, `g++ -O3`) on the same synthetic workload bench.py gives the GPU: stereo 48 kHz streams,
presetDefault, 1.5x, 10 s per stream.  One process per core; each renders streams for a fixed time budget and prints
{"samples", "process_s"} (time inside process() only).
usage: python oracle/cpu_baseline.py <budget_seconds> <seconds_per_stream> <stretch> <first_stream_index>"""
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
    budget, seconds, stretch, first = float(sys.argv[1]), float(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
    sr, C = 48000, 2
    n = int(seconds*sr)
    nout = int(round(n*stretch))
    inputs = [synth_input(first + k, C, n, sr) for k in range(3)]  # one stream of each type (sine / chirp / noise)
    total, samples, count = 0.0, 0, 0
    while total < budget or count < 3:
        x = inputs[count % 3]
        r = ref_oracle.RefStretch()
        r.presetDefault(C, sr)
        t0 = time.perf_counter()
        r.process(x, nout)
        total += time.perf_counter() - t0
        samples += C*(n + nout)
        count += 1
    print(json.dumps(dict(streams=count, samples=samples, process_s=total)))


if __name__ == "__main__":
    main()
