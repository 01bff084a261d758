"""Soak run for the rows with shifted piece grids (hr_fused_core.hpp, FusedItem::phi): stereo f32 tracks of more than 512 hops
at 44.1 / 88.2 kHz with random lengths, levels, isolated peaks at random positions and random buffer offsets, in mixed batches,
every track against the oracle (test infrastructure) and against engine 7 (rows as they were):
python tools/soak_aligned.py FIRST_SEED N_SEEDS.  One line per seed; exits 1 at the first mismatch."""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import headroom_amd as hr  # noqa: E402
import oracle_lib  # noqa: E402

GATES = {"ebur128": 0, "exact": 1, "loudnorm": 2}
TPS = {"swr192k": 0, "bs1770": 1, "sample": 2}


def make_track(rng):
    fs = int(rng.choice([44100, 44100, 88200]))
    seconds = float(rng.uniform(51.0, 125.0)) if fs == 44100 else float(rng.uniform(51.0, 70.0))
    n = int(fs * seconds) + int(rng.integers(0, 4000))
    x = rng.uniform(-1.0, 1.0, size=(n, 2)).astype(np.float32) * np.float32(10.0 ** (rng.uniform(-40, -8) / 20.0))
    pos = 0
    while pos < n:  # level steps: both gates have something to do
        seg = int(rng.integers(fs, 9 * fs))
        x[pos:pos + seg] *= np.float32(10.0 ** (rng.choice([-50.0, -20.0, -6.0, 0.0]) / 20.0))
        pos += seg
    top = 0.5 + 0.45 * rng.random()
    for _ in range(int(rng.integers(1, 30))):  # isolated samples near the top, anywhere -- hop starts and ends included
        p = int(rng.integers(0, n))
        if rng.random() < 0.4:
            p = min(n - 1, (p // (fs // 10)) * (fs // 10) + int(rng.integers(-2, 3)) % (fs // 10))
        x[p, int(rng.integers(0, 2))] = np.float32(top * rng.uniform(0.9, 1.0) * rng.choice([-1.0, 1.0]))
    return x, fs


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    oracle_lib.build()
    modes = [("ebur128", "swr192k"), ("loudnorm", "swr192k"), ("exact", "bs1770"), ("ebur128", "bs1770"), ("ebur128", "sample")]
    t0 = time.time()
    for seed in range(first, first + count):
        gate, tp = modes[seed % len(modes)]
        rng = np.random.default_rng(seed)
        tracks = [make_track(rng) for _ in range(6)]
        keep, specs = [], []
        for pcm, fs in tracks:
            off = int(rng.choice([0, 0, 0, 128, 16, 64]))  # bytes off the 256-byte boundary torch gives
            flat = torch.zeros(pcm.size + 64, dtype=torch.float32, device="cuda")
            view = flat[off // 4: off // 4 + pcm.size]
            view.copy_(torch.from_numpy(pcm.reshape(-1)))
            keep.append(flat)
            specs.append(hr.TrackSpec(view.data_ptr(), fs, channels=2, fmt=hr.FMT_F32, frames=len(pcm)))
        torch.cuda.synchronize()
        res = []
        for engine in (0, 7):
            m = hr.Meter(device=0, gate_mode=GATES[gate], tp_mode=TPS[tp], engine=engine)
            try:
                res.append(m.analyze_batch(specs))
            finally:
                m.close()
        worst = 0.0
        for (pcm, fs), g, p in zip(tracks, res[0], res[1]):
            ref = oracle_lib.measure(pcm, fs, gate_mode=GATES[gate], tp_mode=TPS[tp])
            ok = g["status"] == 0 and g["n_blocks"] == ref["n_blocks"] and g["n_gated"] == ref["n_gated"] == p["n_gated"]
            if math.isfinite(ref["integrated_lufs"]):
                d = abs(g["integrated_lufs"] - ref["integrated_lufs"])
                worst = max(worst, d)
                ok = ok and d < 1e-6 and abs(g["integrated_lufs"] - p["integrated_lufs"]) < 1e-9
            ok = ok and g["peak_per_ch"] == ref["peak_per_ch"] == p["peak_per_ch"]
            if not ok:
                print("MISMATCH seed", seed, gate, tp, fs, len(pcm), g, p, ref, flush=True)
                sys.exit(1)
        print(f"seed {seed} {gate}/{tp}: 6 tracks ok, worst LUFS diff {worst:.2e}, {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
