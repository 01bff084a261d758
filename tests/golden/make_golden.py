#!/usr/bin/env python3
"""Writes tests/golden/oracle_vectors.json: inputs by generator seed, expected outputs from the CPU oracle.

PARITY UNPINNED: FFmpeg is absent from the reference tree and from this image, so these vectors pin the
oracle (and, through the GPU tests, the HIP path) against drift -- not against FFmpeg.  No reference
source is involved; the inputs come from the oracle's own deterministic generator (hro_synth_track, the
same bits as hr_synth_fill on the device).   usage: python tests/golden/make_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as oracle  # noqa: E402

CASES = [  # (seed, fs, channels, seconds)
    (11, 44100, 2, 12.0), (12, 48000, 2, 9.3), (13, 96000, 2, 4.0), (14, 44100, 1, 6.5), (15, 48000, 6, 5.0),
    (16, 32000, 2, 7.7), (17, 88200, 2, 3.1), (18, 192000, 2, 2.0), (19, 44100, 2, 0.35), (20, 8000, 2, 5.0),
]
GATES = {"ebur128": 0, "exact": 1, "loudnorm": 2}
TPS = {"swr192k": 0, "bs1770": 1, "sample": 2}


def main():
    oracle.build()
    out = {"note": "CPU oracle outputs on hro_synth_track(seed, fs, channels, frames); parity unpinned w.r.t. FFmpeg",
           "cases": []}
    for seed, fs, ch, secs in CASES:
        frames = int(fs * secs)
        pcm = oracle.synth_track(seed, fs, ch, frames)
        H = oracle.hop_sums(pcm, fs)
        P = oracle.block_powers(H, fs)
        case = {"seed": seed, "fs": fs, "channels": ch, "frames": frames,
                "pcm_head": [float(x) for x in pcm[:4].reshape(-1)],
                "block_powers_head": [float.hex(float(x)) for x in P[:32]], "results": {}}
        for gname, g in GATES.items():
            for tname, t in TPS.items():
                if (gname != "ebur128") and (tname != "swr192k"):
                    continue
                r = oracle.measure(pcm, fs, gate_mode=g, tp_mode=t)
                case["results"][f"{gname}/{tname}"] = {
                    "integrated_lufs": float.hex(r["integrated_lufs"]), "rel_threshold_lufs": float.hex(r["rel_threshold_lufs"]),
                    "n_blocks": int(r["n_blocks"]), "n_gated": int(r["n_gated"]),
                    "peak_per_ch": [float.hex(float(x)) for x in r["peak_per_ch"]]}
        out["cases"].append(case)
    with open(os.path.join(HERE, "oracle_vectors.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote", len(out["cases"]), "cases")


if __name__ == "__main__":
    main()
