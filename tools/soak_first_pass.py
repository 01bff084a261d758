"""Soak run for hr_loudnorm_first_pass (k_first_pass_rows + the resampled end of the track): random rates, channel counts,
formats and lengths -- lengths cluster around whole numbers of input hops, where the split between the two paths moves --
against the oracle's hro_loudnorm_measure and the all-in-memory path (engine = 1).
python tools/soak_first_pass.py FIRST_SEED N_SEEDS"""
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
import signals  # noqa: E402

RATES = [44100, 44100, 48000, 48000, 88200, 96000, 32000, 16000, 8000, 176400, 22050, 192000]


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    oracle_lib.build()
    new, old = hr.Meter(device=0), hr.Meter(device=0, engine=1)
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        t0 = time.time()
        cases = []
        for _ in range(10):
            fs = int(rng.choice(RATES))
            hop = fs // 10
            hops = int(rng.integers(0, 90))
            frames = max(1, hops * hop + int(rng.choice([-40, -17, -16, -15, -1, 0, 1, 15, 16, 17, 63, 64, 65, int(rng.integers(0, hop))])))
            ch = int(rng.choice([1, 2, 2, 2, 3, 6]))
            x = signals.seeded_noise(seed * 100 + len(cases), frames, ch, float(rng.uniform(-30.0, -3.0)))
            if rng.random() < 0.3:
                x[: frames // 3] *= 0.01
            if rng.random() < 0.3 and frames > 100:
                x[int(rng.integers(0, frames)), int(rng.integers(0, ch))] = 0.99
            kind = rng.choice(["f32", "s16", "s24", "f64"])
            if kind == "s16":
                cases.append((signals.to_s16(x), fs, {}))
            elif kind == "s24":
                cases.append((signals.to_s24_bytes(x), fs, {"channels": ch, "fmt": hr.FMT_S24}))
            elif kind == "f64":
                cases.append((x.astype(np.float64), fs, {}))
            else:
                cases.append((x, fs, {}))
        specs = [hr.TrackSpec(p, fs, **kw) for p, fs, kw in cases]
        got, was = new.loudnorm_first_pass(specs), old.loudnorm_first_pass(specs)
        for (p, fs, kw), g, w in zip(cases, got, was):
            ref = oracle_lib.loudnorm_measure(p, fs, **kw)
            tag = (seed, fs, kw, getattr(p, "shape", None))
            assert g["status"] == w["status"] == 0, tag
            assert g["n_blocks"] == w["n_blocks"] == ref["n_blocks"], (tag, g, ref)
            assert g["peak_per_ch"] == w["peak_per_ch"] == ref["peak_per_ch"], (tag, g, w, ref)
            if math.isfinite(w["integrated_lufs"]):
                assert abs(g["integrated_lufs"] - w["integrated_lufs"]) < 1e-9, (tag, g, w)
            else:
                assert g["integrated_lufs"] == w["integrated_lufs"], (tag, g, w)
            if math.isfinite(ref["integrated_lufs"]):
                # (libebur128's 0.1 LU histogram makes the result a step function of the block energies: DESIGN.md section 1)
                assert abs(g["integrated_lufs"] - ref["integrated_lufs"]) < 0.11 / max(1, ref["n_blocks"]) + 1e-6, (tag, g, ref)
        print("seed %d: %d tracks ok, %.0f s" % (seed, len(cases), time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
