"""The code BASELINE config 5 (one 10-hour stream, 360 000 blocks) really runs, against the oracle: gating and range
statistics in SEVERAL chunks of kGateChunk blocks (2048; 16384 until round 3: the stream keeps its length, 27 chunks now) (k_block_hist with n_parts > 1, the chunk-order partial sums
of k_gate_final, the chunked k_range_stats / k_range_final), on one whole track and on the same stream as three time
shards of more than one chunk each.  8 kHz keeps 54 000 hops at 43 M frames; HR_TP_SAMPLE keeps the oracle fast (the
peak path does not depend on the chunking)."""
import math

import numpy as np
import pytest

import headroom_amd as hr

pytestmark = pytest.mark.gpu

GATES = {"ebur128": 0, "exact": 1, "loudnorm": 2}
FS = 8000
HOP = FS // 10
CHUNK = 16384  # (what hr_device.hpp kGateChunk was when the stream's length was chosen; every multiple of today's 2048)
RANGE_KEYS = (("loudness_range_lu", "lra"), ("lra_low_lufs", "lra_low"), ("lra_high_lufs", "lra_high"),
              ("lra_threshold_lufs", "lra_threshold"), ("max_momentary_lufs", "max_momentary"),
              ("max_shortterm_lufs", "max_shortterm"))


@pytest.fixture(scope="module")
def stream():
    """Stereo noise whose level steps every 3-11 s between -66 and -8 dBFS, with passages below the absolute gate:
    both gates and the range percentiles have something to do in every chunk."""
    n_hops = 3 * CHUNK + 5000
    frames = n_hops * HOP + 123
    rng = np.random.default_rng(20261111)
    x = rng.uniform(-1.0, 1.0, size=(frames, 2)).astype(np.float32)
    pos = 0
    while pos < frames:
        seg = int(rng.integers(3 * FS, 11 * FS))
        lvl = rng.choice([-96.0, -66.0, -48.0, -36.0, -30.0, -24.0, -20.0, -14.0, -8.0], p=[.08, .07, .1, .15, .15, .15, .1, .1, .1])
        x[pos:pos + seg] *= np.float32(10.0 ** (lvl / 20.0))
        pos += seg
    x[:, 1] *= np.float32(0.7)
    return x


def close(a, b, tol):
    return (math.isnan(a) and math.isnan(b)) or a == b or abs(a - b) < tol


def check_result(got, ref, rng_ref):
    assert got["status"] == 0
    assert got["n_blocks"] == ref["n_blocks"] > 3 * CHUNK
    assert got["n_gated"] == ref["n_gated"] < ref["n_blocks"]          # the absolute gate removed something
    assert abs(got["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6
    assert abs(got["rel_threshold_lufs"] - ref["rel_threshold_lufs"]) < 1e-6
    assert got["peak_per_ch"] == ref["peak_per_ch"]
    for k_got, k_ref in RANGE_KEYS:
        assert close(got[k_got], rng_ref[k_ref], 1e-6), (k_got, got[k_got], rng_ref[k_ref])
    assert got["loudness_range_lu"] > 5.0


@pytest.mark.parametrize("gate", ["ebur128", "loudnorm", "exact"])
def test_many_gating_chunks_whole_and_sharded(oracle, stream, gate):
    ref = oracle.measure(stream, FS, gate_mode=GATES[gate], tp_mode=2)
    rng_ref = oracle.range_stats(stream, FS, GATES[gate])
    with hr.Meter(device=0, gate_mode=GATES[gate], tp_mode=hr.TP_SAMPLE) as m:
        whole = m.analyze(stream, FS)
        check_result(whole, ref, rng_ref)

        # the same stream as three time shards of more than one chunk each (cuts on hop boundaries, off the chunk grid)
        n_hops = stream.shape[0] // HOP
        cuts = [0, (CHUNK + 777) * HOP, (2 * CHUNK + 2222) * HOP, stream.shape[0]]
        assert all((cuts[i + 1] - cuts[i]) // HOP > CHUNK for i in range(3)) and cuts[2] // HOP < n_hops
        shards = [hr.TrackSpec(stream[cuts[i]:cuts[i + 1]], FS) for i in range(3)]
        ms = [hr.Meter(device=0, gate_mode=GATES[gate], tp_mode=hr.TP_SAMPLE) for _ in range(3)]
        try:
            heads = [ms[i].shard_head(shards[i]) for i in range(3)]
            tails = [ms[i].shard_tail(shards[i]) for i in range(2)]
            edges = [ms[i].shard_pass1(shards[i], tails[i - 1] if i else None, heads[i + 1] if i < 2 else None,
                                       first_hop=cuts[i] // HOP) for i in range(3)]
            parts = [ms[i].shard_pass2(edges[i - 1] if i else None) for i in range(3)]
            red = hr.HrShardPartial()
            hist = np.zeros(8001, dtype=np.uint64)
            st_hist = np.zeros(8001, dtype=np.uint64)
            for p in parts:  # the all-reduce of sharded.py: counts and sums add, maxima take the maximum
                hist += np.ctypeslib.as_array(p.hist)
                st_hist += np.ctypeslib.as_array(p.st_hist)
                red.sum_kept += p.sum_kept
                red.n_kept += p.n_kept
                red.n_blocks += p.n_blocks
                red.st_sum_kept += p.st_sum_kept
                red.st_n_kept += p.st_n_kept
                red.n_shortterm += p.n_shortterm
                red.max_m_power = max(red.max_m_power, p.max_m_power)
                red.max_s_power = max(red.max_s_power, p.max_s_power)
                for c in range(2):
                    red.peak_per_ch[c] = max(red.peak_per_ch[c], p.peak_per_ch[c])
            np.ctypeslib.as_array(red.hist)[:] = hist
            np.ctypeslib.as_array(red.st_hist)[:] = st_hist
            red.range_ok = min(p.range_ok for p in parts)
            assert red.range_ok == 1
            if gate == "exact":
                third = [mm.shard_pass3(red) for mm in ms]
                red.sum_gated = sum(p.sum_gated for p in third)
                red.n_gated = sum(p.n_gated for p in third)
            got = ms[0].shard_finish(red, 2)
        finally:
            for mm in ms:
                mm.close()
        check_result(got, ref, rng_ref)
        # sharded and whole agree with each other far below the bar (the sums of kept powers are formed in another
        # order: per shard, then over shards)
        assert abs(got["integrated_lufs"] - whole["integrated_lufs"]) < 1e-9
        assert got["n_gated"] == whole["n_gated"]


@pytest.mark.parametrize("gate", ["ebur128", "loudnorm", "exact"])
def test_short_and_long_tracks_in_one_launch(oracle, stream, gate):
    """One launch that holds tracks of one gating chunk, of several (k_gate_small: one workgroup gates the whole track, the
    histograms in LDS, the chunks' sums added in chunk order) and of more than kGateSmallParts = 8 (the chunked kernels
    through the histograms in global memory, told by a list which tracks are theirs): every one against the oracle, and
    the same bits as when it is analysed alone."""
    cuts = [(0, 700), (1000, 1000 + 5000), (0, stream.shape[0] // HOP), (9000, 9000 + 2048 * 8), (20000, 20000 + 31), (30000, 30000 + 2048 * 8 + 4)]
    specs, pcms = [], []
    for a, b in cuts:
        pcm = stream[a * HOP: b * HOP + 57] if b * HOP + 57 <= stream.shape[0] else stream[a * HOP:]
        pcms.append(pcm)
        specs.append(hr.TrackSpec(pcm, FS))
    with hr.Meter(device=0, gate_mode=GATES[gate], tp_mode=hr.TP_SAMPLE) as m:
        got = m.analyze_batch(specs)
        for g, pcm in zip(got, pcms):
            ref = oracle.measure(pcm, FS, gate_mode=GATES[gate], tp_mode=2)
            rng_ref = oracle.range_stats(pcm, FS, GATES[gate])
            assert g["status"] == 0
            assert g["n_blocks"] == ref["n_blocks"] and g["n_gated"] == ref["n_gated"]
            assert abs(g["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6
            assert g["peak_per_ch"] == ref["peak_per_ch"]
            for k_got, k_ref in RANGE_KEYS:
                assert close(g[k_got], rng_ref[k_ref], 1e-6), (k_got, g[k_got], rng_ref[k_ref])
            alone = m.analyze(pcm, FS)
            for k in ("integrated_lufs", "rel_threshold_lufs", "n_gated", "n_blocks", "peak_per_ch") + tuple(k for k, _ in RANGE_KEYS):
                assert g[k] == alone[k] or (isinstance(g[k], float) and math.isnan(g[k]) and math.isnan(alone[k])), (k, g[k], alone[k])
