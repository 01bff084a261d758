"""hr_analyze_device_noalloc (the entry point that allocates nothing) against hr_analyze_batch, through the C ABI.

Same kernels, same workspace layout, the workspace merely lives in the library's static device array: results must
be those of hr_analyze_batch -- counts and peaks bit-identical, loudness figures equal to 1e-9 LU (the bar of the
other GPU parity tests is 1e-6 LU).  Two calls in a row, the second on a smaller input, because the array keeps the
last call's contents.  Collected last (test_zz_*): the round's end runs pytest -x.
"""
import math

import numpy as np
import pytest

import headroom_amd as hr

pytestmark = pytest.mark.gpu

FLOAT_KEYS = ("integrated_lufs", "true_peak_dbtp", "rel_threshold_lufs", "loudness_range_lu", "lra_low_lufs", "lra_high_lufs",
              "lra_threshold_lufs", "max_momentary_lufs", "max_shortterm_lufs")


def same(a, b):
    assert a["status"] == b["status"] == 0, (a, b)
    assert a["n_blocks"] == b["n_blocks"] and a["n_gated"] == b["n_gated"], (a, b)
    assert a["peak_per_ch"] == b["peak_per_ch"], (a, b)  # bit-identical f64
    for k in FLOAT_KEYS:
        x, y = a[k], b[k]
        if math.isfinite(x) or math.isfinite(y):
            assert abs(x - y) < 1e-9, (k, a, b)
        else:
            assert (math.isnan(x) and math.isnan(y)) or x == y, (k, a, b)


def device_track(m, seed, fs, ch, frames):
    import torch
    t = torch.empty((frames, ch), dtype=torch.float32, device="cuda:0")
    m.synth_fill(t.data_ptr(), seed, fs, ch, frames)
    torch.cuda.synchronize()
    return t


def against_oracle(got, pcm, fs, **kw):
    """The entry point's own result against the CPU oracle, to the bar of tests/test_gpu_parity.py."""
    import oracle_lib
    ref = oracle_lib.measure(pcm, fs, **kw)
    assert got["n_blocks"] == ref["n_blocks"] and got["n_gated"] == ref["n_gated"], (got, ref)
    assert abs(got["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6, (got, ref)
    assert got["peak_per_ch"] == ref["peak_per_ch"], (got, ref)  # bit-identical f64


def test_noalloc_matches_batch_twice_second_smaller():
    import torch
    fs = 44100
    with hr.Meter(device=0) as m:
        big = device_track(m, 9001, fs, 2, fs * 31 + 777)     # 310 hops: five waves of the fused kernel, a ragged tail
        small = device_track(m, 9002, fs, 2, fs * 7 + 123)    # 70 hops: two waves
        for t in (big, small):                                 # the second call finds the first one's contents in the array
            spec = hr.TrackSpec(t, fs)
            ref = m.analyze_batch([spec])[0]
            # the property the entry point exists for: the call takes no device memory (and gives none back)
            torch.cuda.synchronize()
            free_before = torch.cuda.mem_get_info(0)[0]
            rc, got = m.analyze_device_noalloc(spec)
            free_after = torch.cuda.mem_get_info(0)[0]
            assert rc == hr.OK, rc
            assert free_after == free_before, (free_before, free_after)
            same(got, ref)
            against_oracle(got, t.cpu().numpy(), fs)
        # and once more the other way round (a larger input after a smaller one: every buffer is carved again)
        spec = hr.TrackSpec(big, fs)
        rc, got = m.analyze_device_noalloc(spec)
        assert rc == hr.OK
        same(got, m.analyze_batch([spec])[0])


def test_noalloc_other_layouts_and_modes():
    import torch
    # loudnorm gating + BS.1770 peak on 48 kHz stereo; 5.1 (generic kernels in this entry point, stereo planes in
    # hr_analyze_batch: peaks still bit-identical, loudness to 1e-9)
    with hr.Meter(device=0, gate_mode=hr.GATE_LOUDNORM, tp_mode=hr.TP_BS1770) as m:
        t = device_track(m, 9003, 48000, 2, 48000 * 9 + 5)
        spec = hr.TrackSpec(t, 48000)
        ref = m.analyze_batch([spec])[0]
        rc, got = m.analyze_device_noalloc(spec)
        assert rc == hr.OK
        same(got, ref)
    with hr.Meter(device=0) as m:
        t = device_track(m, 9004, 48000, 6, 48000 * 6 + 11)
        w = [1, 1, 1, 0, 1.41, 1.41]
        spec = hr.TrackSpec(t, 48000, weights=w)
        ref = m.analyze_batch([spec])[0]
        rc, got = m.analyze_device_noalloc(spec)
        assert rc == hr.OK
        same(got, ref)
        # a short stereo s16 track after it (direct integer read of the fused kernel)
        q = torch.clamp(torch.round(device_track(m, 9005, 48000, 2, 48000 * 3) * 32768.0), -32768, 32767).to(torch.int16).contiguous()
        spec = hr.TrackSpec(q, 48000)
        ref = m.analyze_batch([spec])[0]
        rc, got = m.analyze_device_noalloc(spec)
        assert rc == hr.OK
        same(got, ref)


def test_noalloc_rejections():
    fs = 44100
    with hr.Meter(device=0) as m:
        t = device_track(m, 9006, fs, 2, fs * 2)
        spec = hr.TrackSpec(t, fs)
        # the rate's tables are not resident before an ordinary call has seen the rate ...
        fresh = hr.Meter(device=0)
        rc, _ = fresh.analyze_device_noalloc(hr.TrackSpec(t.data_ptr(), 32000, channels=2, fmt=hr.FMT_F32, frames=fs * 2))
        assert rc == hr.ERR_NOT_READY
        fresh.close()
        m.analyze_batch([spec])
        # ... host PCM would need a staging buffer
        rc, _ = m.analyze_device_noalloc(hr.TrackSpec(np.zeros((fs, 2), dtype=np.float32), fs))
        assert rc == hr.ERR_INVALID
        # ... and the size limit is checked before anything is touched (the pointer is never dereferenced)
        too_long = hr.TrackSpec(t.data_ptr(), fs, channels=2, fmt=hr.FMT_F32,
                                frames=(hr.NOALLOC_MAX_HOP_CHANNELS // 2 + 1) * (fs // 10))
        rc, r = m.analyze_device_noalloc(too_long)
        assert rc == hr.ERR_TOO_LARGE
        # the largest input it accepts is served (a 54-minute stereo stream of silence: -inf, zero peak)
        import torch
        n = (hr.NOALLOC_MAX_HOP_CHANNELS // 2) * (fs // 10)
        z = torch.zeros((n, 2), dtype=torch.float32, device="cuda:0")
        rc, r = m.analyze_device_noalloc(hr.TrackSpec(z, fs))
        assert rc == hr.OK and r["status"] == 0
        assert r["n_blocks"] == hr.NOALLOC_MAX_HOP_CHANNELS // 2 - 3 and r["integrated_lufs"] == -math.inf
        ref = m.analyze_batch([hr.TrackSpec(z, fs)])[0]
        same(r, ref)
        # the context's own workspace is untouched by all this
        same(m.analyze_batch([spec])[0], m.analyze_device_noalloc(spec)[1])


def test_noalloc_limits_that_grow_with_frames():
    """The work-item lists grow with the frames, not the hops: a high-rate mono track inside the hop limit is rejected up
    front when its estimate passes the array (nothing is queued: the pointer is never read), and a full-scale tone -- every
    fused wave is redone exactly, the worst case of the redo list -- is served."""
    import torch
    with hr.Meter(device=0) as m:
        fs = 768000
        z = torch.zeros((fs, 1), dtype=torch.float32, device="cuda:0")
        m.analyze_batch([hr.TrackSpec(z, fs)])  # (the rate's tables become resident)
        hops = 60000  # inside the hop x channel limit, 4.6e9 frames
        rc, r = m.analyze_device_noalloc(hr.TrackSpec(z.data_ptr(), fs, channels=1, fmt=hr.FMT_F32, frames=hops * (fs // 10)))
        assert rc == hr.ERR_TOO_LARGE and r["status"] == hr.ERR_TOO_LARGE
        rc, r = m.analyze_device_noalloc(hr.TrackSpec(z, fs))
        assert rc == hr.OK
        same(r, m.analyze_batch([hr.TrackSpec(z, fs)])[0])
        fs = 48000
        n = fs * 600
        tt = torch.arange(n, dtype=torch.float64, device="cuda:0")
        tone = torch.sin(tt * (2 * math.pi * 997.0 / fs)).to(torch.float32)
        pcm = torch.stack([tone, 0.5 * tone], dim=1).contiguous()
        spec = hr.TrackSpec(pcm, fs)
        ref = m.analyze_batch([spec])[0]
        rc, r = m.analyze_device_noalloc(spec)
        assert rc == hr.OK
        same(r, ref)
        # the same at 44.1 kHz, where the rows of whole blocks of 512 hops have shifted piece grids and the redo of such a wave
        # is one work item per hop
        fs = 44100
        n = fs * 600
        tt = torch.arange(n, dtype=torch.float64, device="cuda:0")
        tone = torch.sin(tt * (2 * math.pi * 997.0 / fs)).to(torch.float32)
        pcm = torch.stack([tone, 0.5 * tone], dim=1).contiguous()
        spec = hr.TrackSpec(pcm, fs)
        ref = m.analyze_batch([spec])[0]
        rc, r = m.analyze_device_noalloc(spec)
        assert rc == hr.OK
        same(r, ref)
