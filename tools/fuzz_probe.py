"""Fuzz hr_pcm_probe (host code, no GPU): truncations and byte mutations of valid WAV / AIFF / AIFF-C images.  The
probe must return a status, never crash, and whatever it accepts must lie inside the buffer.
python tools/fuzz_probe.py N_CASES [SEED]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import containers  # noqa: E402
import headroom_amd as hr  # noqa: E402


def main():
    n = int(sys.argv[1])
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    pcm = rng.uniform(-0.5, 0.5, size=(37, 2))
    pcm6 = rng.uniform(-0.5, 0.5, size=(19, 6))
    seeds = []
    for kind in ("s16", "s24", "s32", "f32", "f64"):
        seeds.append(containers.wav_bytes(pcm, 44100, kind)[0])
        seeds.append(containers.wav_bytes(pcm6, 96000, kind, extensible_mask=0x3F, extra_chunks=[(b"LIST", b"abcde")])[0])
    for kind, aifc in (("s16", None), ("s24", None), ("s16", b"sowt"), ("f32", b"fl32"), ("f64", b"fl64"), ("s24", b"NONE")):
        seeds.append(containers.aiff_bytes(pcm, 48000, kind, aifc=aifc)[0])
        seeds.append(containers.aiff_bytes(pcm, 48000, kind, aifc=aifc, ssnd_offset=6, ssnd_first=True)[0])
    accepted = 0
    for i in range(n):
        b = bytearray(seeds[int(rng.integers(len(seeds)))])
        op = int(rng.integers(4))
        if op == 0:
            b = b[: int(rng.integers(0, len(b) + 1))]
        elif op == 1:
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(len(b)))] = int(rng.integers(256))
        elif op == 2:  # a 32-bit field takes an extreme value
            at = int(rng.integers(0, max(1, len(b) - 4)))
            b[at:at + 4] = bytes(rng.choice([0, 0xFF, 0x7F, 0x80], size=4).astype(np.uint8))
        else:
            at = int(rng.integers(len(b)))
            b = b[:at] + bytes(rng.integers(0, 256, size=int(rng.integers(1, 9)), dtype=np.uint8)) + b[at:]
        data = bytes(b)
        rc, info = hr.pcm_probe(data)
        if rc == 0:
            accepted += 1
            bps = {hr.FMT_F32: 4, hr.FMT_S16: 2, hr.FMT_S24: 3, hr.FMT_S32: 4, hr.FMT_F64: 8}[info["fmt"]]
            end = info["data_offset"] + info["frames"] * info["channels"] * bps
            assert 1 <= info["channels"] <= 8 and info["sample_rate"] > 0, info
            assert end <= len(data), (info, len(data))
    print(f"{n} mutated images, {accepted} accepted, none crashed or overran")


if __name__ == "__main__":
    main()
