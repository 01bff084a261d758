#!/usr/bin/env python3
"""Writes tests/golden/first_pass_vectors.json: the loudnorm first pass (hro_loudnorm_measure: resample to 192 kHz with
libswresample's defaults, libebur128's meter there) of tracks from the oracle's own generator.

PARITY UNPINNED, as oracle_vectors.json: FFmpeg is in neither the reference tree nor this image, so these vectors pin
the oracle -- and, through tests/test_gpu_first_pass.py, hr_loudnorm_first_pass -- against drift, not against FFmpeg.
usage: python tests/golden/make_golden_first_pass.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as oracle  # noqa: E402

CASES = [  # (seed, fs, channels, seconds)
    (31, 44100, 2, 8.13), (32, 48000, 2, 7.0), (33, 96000, 2, 3.0), (34, 44100, 1, 4.2), (35, 48000, 6, 3.0),
    (36, 88200, 2, 2.5), (37, 32000, 2, 3.3), (38, 44100, 2, 0.5), (39, 192000, 2, 1.5),
]
KEYS = ("integrated_lufs", "rel_threshold_lufs", "true_peak_dbtp")


def main():
    oracle.build()
    out = {"note": "hro_loudnorm_measure on hro_synth_track(seed, fs, channels, frames); parity unpinned w.r.t. FFmpeg", "cases": []}
    for seed, fs, ch, secs in CASES:
        frames = int(fs * secs)
        r = oracle.loudnorm_measure(oracle.synth_track(seed, fs, ch, frames), fs)
        case = {"seed": seed, "fs": fs, "channels": ch, "frames": frames, "n_blocks": int(r["n_blocks"]), "n_gated": int(r["n_gated"]),
                "peak_per_ch": [float.hex(float(x)) for x in r["peak_per_ch"]]}
        for k in KEYS:
            case[k] = float.hex(float(r[k]))
        out["cases"].append(case)
    with open(os.path.join(HERE, "first_pass_vectors.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote", len(out["cases"]), "cases")


if __name__ == "__main__":
    main()
