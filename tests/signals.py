"""Synthetic test signals (EBU Tech 3341 meter test set, synthesised from the definitions; the
EBU .wav files themselves are not in the image) and seeded noise helpers."""
import numpy as np


def db(x):
    return 10.0 ** (x / 20.0)


def sine(fs, freq, dbfs, seconds, phase_deg=0.0):
    n = np.arange(int(round(fs * seconds)), dtype=np.float64)
    return db(dbfs) * np.sin(2.0 * np.pi * freq * n / fs + np.deg2rad(phase_deg))


def fade(x, n):
    """raised-cosine fade-in/out over n samples (in place on a copy)"""
    x = np.array(x, dtype=np.float64)
    w = 0.5 - 0.5 * np.cos(np.pi * np.arange(n) / n)
    x[:n] *= w
    x[-n:] *= w[::-1]
    return x


def stereo(x):
    return np.stack([x, x], axis=1).astype(np.float32)


def ebu3341_case(case, fs=48000):
    """Returns (pcm float32 [frames, ch], expected_I_lufs or None, expected_TP_dbtp or None)."""
    seq = {
        1: [(-23.0, 20.0)],
        2: [(-33.0, 20.0)],
        3: [(-36.0, 10.0), (-23.0, 60.0), (-36.0, 10.0)],
        4: [(-72.0, 10.0), (-36.0, 10.0), (-23.0, 60.0), (-36.0, 10.0), (-72.0, 10.0)],
        5: [(-26.0, 20.0), (-20.0, 20.1), (-26.0, 20.0)],
    }
    if case in seq:
        parts = []
        ph = 0
        for lvl, dur in seq[case]:
            n = np.arange(int(round(fs * dur)), dtype=np.float64) + ph
            parts.append(db(lvl) * np.sin(2.0 * np.pi * 1000.0 * n / fs))
            ph += len(n)
        x = np.concatenate(parts)
        return stereo(x), (-23.0 if case != 2 else -33.0), None
    if case == 6:  # 5.1: L R C LFE Ls Rs
        n = int(fs * 20.0)
        s = lambda lvl: sine(fs, 1000.0, lvl, 20.0)
        pcm = np.zeros((n, 6), dtype=np.float32)
        pcm[:, 0] = s(-28.0)
        pcm[:, 1] = s(-28.0)
        pcm[:, 2] = s(-24.0)
        pcm[:, 4] = s(-30.0)
        pcm[:, 5] = s(-30.0)
        return pcm, -23.0, None
    tp = {15: (4, 0.0, 0.5, -6.0), 16: (4, 45.0, 0.5, -6.0), 17: (6, 60.0, 0.5, -6.0),
          18: (8, 67.5, 0.5, -6.0), 19: (4, 45.0, 1.41, 3.0)}
    if case in tp:
        div, ph, amp, exp = tp[case]
        x = sine(fs, fs / div, 20.0 * np.log10(amp), 2.0, ph)
        # 10 ms raised-cosine fades: an abrupt start/stop is itself an inter-sample overshoot
        # (+0.44 dB with the swr bank's start reflection) that is not what these cases probe
        return stereo(fade(x, fs // 100)), None, exp
    raise ValueError(case)


def ebu3342_case(case, fs=48000):
    """EBU Tech 3342 loudness-range test signals 1-4 (stereo 1 kHz sine, 20 s per level), synthesised from their
    definitions.  Returns (pcm, expected LRA in LU); the tolerance of the document is +-1 LU."""
    levels = {1: [-20.0, -30.0], 2: [-20.0, -15.0], 3: [-40.0, -20.0], 4: [-50.0, -35.0, -20.0, -35.0, -50.0]}[case]
    n = np.arange(int(fs * 20.0 * len(levels)), dtype=np.float64)
    amp = np.repeat([db(l) for l in levels], int(fs * 20.0))
    return stereo(amp * np.sin(2.0 * np.pi * 1000.0 * n / fs)), {1: 10.0, 2: 5.0, 3: 20.0, 4: 15.0}[case]


def seeded_noise(seed, frames, channels, dbfs=-20.0, kind="white"):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1.0, 1.0, size=(frames, channels))
    if kind == "pink":  # crude 1/f-ish: cumulative leaky integrator mix
        y = np.zeros_like(x)
        acc = np.zeros(channels)
        for i in range(frames):
            acc = 0.98 * acc + 0.02 * x[i]
            y[i] = 0.5 * x[i] + 4.0 * acc
        x = y / np.max(np.abs(y))
    return (db(dbfs) * x).astype(np.float32)


def to_s16(x):
    return np.clip(np.round(x * 32768.0), -32768, 32767).astype(np.int16)


def to_s24_bytes(x):
    """float [-1,1) -> packed little-endian 24-bit, uint8 array of length frames*ch*3."""
    v = np.clip(np.round(np.asarray(x, dtype=np.float64) * 8388608.0), -8388608, 8388607).astype(np.int32)
    flat = v.reshape(-1)
    out = np.empty((flat.size, 3), dtype=np.uint8)
    out[:, 0] = flat & 0xFF
    out[:, 1] = (flat >> 8) & 0xFF
    out[:, 2] = (flat >> 16) & 0xFF
    return out.reshape(-1)
