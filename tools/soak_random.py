"""Soak run of the seeded random differential test (tests/test_gpu_random.py) over many more seeds than the test suite
takes: python tools/soak_random.py FIRST_SEED N_SEEDS.  Every track of every mixed batch is compared with the oracle
(test infrastructure, as in the test itself).  Prints one line per seed; exits 1 at the first mismatch."""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import headroom_amd as hr  # noqa: E402
import oracle_lib  # noqa: E402
from test_gpu_random import GATES, TPS, make_case  # noqa: E402


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    oracle_lib.build()
    oracle = oracle_lib
    modes = [("ebur128", "swr192k"), ("loudnorm", "swr192k"), ("exact", "bs1770"), ("ebur128", "bs1770"), ("ebur128", "sample")]
    t0 = time.time()
    for seed in range(first, first + count):
        gate, tp = modes[seed % len(modes)]
        rng = np.random.default_rng(seed)
        cases = [make_case(rng) for _ in range(28)]
        m = hr.Meter(device=0, gate_mode=GATES[gate], tp_mode=TPS[tp], engine=(0, 2, 3, 4)[(seed // 5) % 4])  # row lengths of the fused kernel
        try:
            got = m.analyze_batch([hr.TrackSpec(p, fs, **kw) for p, fs, kw, _ in cases])
        finally:
            m.close()
        worst = 0.0
        for (pcm, fs, kw, tag), g in zip(cases, got):
            ref = oracle.measure(pcm, fs, gate_mode=GATES[gate], tp_mode=TPS[tp], **kw)
            ok = g["status"] == 0 and g["n_blocks"] == ref["n_blocks"] and g["n_gated"] == ref["n_gated"]
            if math.isfinite(ref["integrated_lufs"]):
                d = abs(g["integrated_lufs"] - ref["integrated_lufs"])
                worst = max(worst, d)
                ok = ok and d < 1e-6
            else:
                ok = ok and g["integrated_lufs"] == ref["integrated_lufs"]
            ok = ok and g["peak_per_ch"] == ref["peak_per_ch"]
            if not ok:
                print("MISMATCH seed", seed, gate, tp, tag, g, ref, flush=True)
                sys.exit(1)
        print(f"seed {seed} {gate}/{tp}: 28 tracks ok, worst LUFS diff {worst:.2e}, {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
