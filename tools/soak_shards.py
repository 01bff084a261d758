"""Soak run for the time-shard path and the integer-format path of the fused kernel: awkward signals (tools/soak_long.py)
at random rates and lengths, (a) as one stream over 2..8 in-process time shards, (b) as s16 / s24 stereo and 3..6-channel
tracks, each against the oracle.  python tools/soak_shards.py FIRST_SEED N_SEEDS"""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np  # noqa: E402

import headroom_amd as hr  # noqa: E402
import oracle_lib  # noqa: E402
import signals  # noqa: E402
from soak_long import KINDS, make  # noqa: E402


def same(g, ref, tag, what):
    ok = g["status"] == 0 and g["n_blocks"] == ref["n_blocks"] and g["n_gated"] == ref["n_gated"]
    if math.isfinite(ref["integrated_lufs"]):
        ok = ok and abs(g["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6
    else:
        ok = ok and g["integrated_lufs"] == ref["integrated_lufs"]
    ok = ok and g["peak_per_ch"] == ref["peak_per_ch"]
    if not ok:
        print("MISMATCH", what, tag, g, ref, flush=True)
        sys.exit(1)


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    oracle_lib.build()
    t0 = time.time()
    meters = [hr.Meter(device=0) for _ in range(8)]
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        for kind in KINDS:
            pcm, fs, tag = make(rng, kind)
            ref = oracle_lib.measure(pcm, fs, gate_mode=0, tp_mode=0)
            ns = int(rng.integers(2, 9))
            got = hr.analyze_stream_sharded(meters[:ns], hr.TrackSpec(pcm, fs))
            same(got, ref, tag, f"seed {seed} shards {ns}")
            # integer formats: s16 / s24 stereo (widened on the device), and 3..6 channels of them (plane split)
            x = pcm.astype(np.float64)
            ch = int(rng.choice([2, 2, 3, 5, 6]))
            if ch != 2:
                x = np.concatenate([x, x[:, :1] * 0.5, x[::-1, :2], x[:, 1:2] * -0.7], axis=1)[:, :ch]
            x = x[: min(len(x), fs * 12)]
            if rng.random() < 0.5:
                ipcm, kw = signals.to_s16(x), {}
            else:
                ipcm, kw = signals.to_s24_bytes(x), {"channels": ch, "fmt": hr.FMT_S24}
            ref = oracle_lib.measure(ipcm, fs, gate_mode=0, tp_mode=0, **kw)
            got = meters[0].analyze(ipcm, fs, **kw)
            same(got, ref, tag, f"seed {seed} integer ch {ch} {'s24' if kw else 's16'}")
        print(f"seed {seed}: {len(KINDS)} streams x shards + integer tracks ok, {time.time() - t0:.0f} s", flush=True)
    for m in meters:
        m.close()


if __name__ == "__main__":
    main()
