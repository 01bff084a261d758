"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same inputs.

Bars (BASELINE.json north_star): block energies within 1e-6 relative of the f64 CPU reference (asserted
here at 1e-9), integrated loudness within 0.01 LU (asserted at 1e-6 LU), true peak within 0.05 dB
(asserted bit-identical: the GPU evaluates the candidates of the maximum in the oracle's own f64 FMA
order).  "Parity unpinned" with respect to FFmpeg itself: see oracle/hr_oracle.h.
"""
import math

import numpy as np
import pytest

import headroom_amd as hr
import signals

pytestmark = pytest.mark.gpu

GATES = {"ebur128": 0, "exact": 1, "loudnorm": 2}
TPS = {"swr192k": 0, "bs1770": 1, "sample": 2}


@pytest.fixture(scope="module")
def meters():
    cache = {}

    def get(gate="ebur128", tp="swr192k", engine=0):
        key = (gate, tp, engine)
        if key not in cache:
            cache[key] = hr.Meter(device=0, gate_mode=GATES[gate], tp_mode=TPS[tp], engine=engine)
        return cache[key]

    yield get
    for m in cache.values():
        m.close()


def check_against_oracle(oracle, m, pcm, fs, gate="ebur128", tp="swr192k", **kw):
    ref = oracle.measure(pcm, fs, gate_mode=GATES[gate], tp_mode=TPS[tp], **kw)
    got = m.analyze(pcm, fs, **kw)
    assert got["status"] == 0
    assert got["n_blocks"] == ref["n_blocks"]
    assert got["n_gated"] == ref["n_gated"]
    if math.isfinite(ref["integrated_lufs"]):
        assert abs(got["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6, (got, ref)
        assert abs(got["rel_threshold_lufs"] - ref["rel_threshold_lufs"]) < 1e-6, (got, ref)
    else:
        assert got["integrated_lufs"] == ref["integrated_lufs"]
    assert got["peak_per_ch"] == ref["peak_per_ch"], (got, ref)  # bit-identical f64
    assert got["true_peak_dbtp"] == ref["true_peak_dbtp"]
    return got, ref


@pytest.mark.parametrize("engine", [0, 1, 2, 3])  # (2 / 3: rows of a whole hop only / of a fraction of a hop)
@pytest.mark.parametrize("fs", [44100, 48000, 96000])
def test_block_energies_match_oracle(oracle, meters, fs, engine):
    pcm = signals.seeded_noise(100 + fs, fs * 5 + 1234, 2, -18.0)
    pcm[: fs] *= 0.01
    m = meters(engine=engine)
    H, P = m.block_energies(hr.TrackSpec(pcm, fs))
    Href = oracle.hop_sums(pcm, fs)
    Pref = oracle.block_powers(Href, fs)
    assert H.shape == Href.shape and P.shape == Pref.shape
    assert np.max(np.abs(H / Href - 1)) < 1e-9
    assert np.max(np.abs(P / Pref - 1)) < 1e-9


@pytest.mark.parametrize("engine", [0, 1, 2, 3])
@pytest.mark.parametrize("gate", ["ebur128", "exact", "loudnorm"])
@pytest.mark.parametrize("fs", [44100, 48000, 96000])
def test_measure_noise(oracle, meters, fs, gate, engine):
    pcm = signals.seeded_noise(7 + fs, fs * 6 + 77, 2, -16.0)
    pcm[: 2 * fs] *= 0.004
    check_against_oracle(oracle, meters(gate, "swr192k", engine), pcm, fs, gate, "swr192k")


@pytest.mark.parametrize("engine", [0, 1])
@pytest.mark.parametrize("tp", ["bs1770", "sample"])
@pytest.mark.parametrize("fs", [44100, 48000, 96000, 192000])
def test_other_true_peak_modes(oracle, meters, fs, tp, engine):
    pcm = signals.seeded_noise(9 + fs, fs * 2 + 5, 2, -10.0)
    check_against_oracle(oracle, meters("ebur128", tp, engine), pcm, fs, "ebur128", tp)


def test_192k_input_true_peak_is_sample_peak(oracle, meters):
    pcm = signals.seeded_noise(5, 192000 * 1, 2, -10.0)
    got, ref = check_against_oracle(oracle, meters(), pcm, 192000)
    assert got["peak_per_ch"][0] == float(np.max(np.abs(pcm[:, 0])))


@pytest.mark.parametrize("case", [1, 3, 4, 5, 6, 16, 19])
def test_ebu3341_cases_on_gpu(oracle, meters, case):
    pcm, exp_i, exp_tp = signals.ebu3341_case(case)
    got, ref = check_against_oracle(oracle, meters(), pcm, 48000)
    if exp_i is not None:
        assert abs(got["integrated_lufs"] - exp_i) <= 0.1
    if exp_tp is not None:
        assert exp_tp - 0.4 <= got["true_peak_dbtp"] <= exp_tp + 0.2


def test_sweep_square_dc(oracle, meters):
    fs = 48000
    n = np.arange(fs * 3)
    sweep = 0.5 * np.sin(2 * np.pi * (20.0 + 3000.0 * n / fs) * n / fs)
    square = 0.999 * np.sign(np.sin(2 * np.pi * 441.0 * n / fs))
    dc = np.full(n.shape, 0.25)
    for x in (sweep, square, dc):
        pcm = signals.stereo(x)
        m = meters()
        H, P = m.block_energies(hr.TrackSpec(pcm, fs))
        Href = oracle.hop_sums(pcm, fs)
        # relative to the loudest hop: a pure DC input decays to ~1e-30 after the high-pass
        assert np.max(np.abs(H - Href)) <= 1e-9 * np.max(Href)
        check_against_oracle(oracle, m, pcm, fs)


@pytest.mark.parametrize("engine", [0, 1])
def test_config4_96k_s24_5_1(oracle, meters, engine):
    """BASELINE config 4: 96 kHz, 24-bit packed, 5.1 (LFE excluded from loudness, true peak on all six)."""
    fs, secs = 96000, 2
    x = signals.seeded_noise(44, fs * secs, 6, -14.0).astype(np.float64)
    x[:, 3] = signals.sine(fs, 50.0, -4.0, secs)  # loud LFE
    raw = signals.to_s24_bytes(x)
    m = meters(engine=engine)
    got, ref = check_against_oracle(oracle, m, raw, fs, channels=6, fmt=hr.FMT_S24)
    assert got["peak_per_ch"][3] == max(got["peak_per_ch"])
    x2 = x.copy()
    x2[:, 3] = 0
    got2 = m.analyze(signals.to_s24_bytes(x2), fs, channels=6, fmt=hr.FMT_S24)
    assert got2["integrated_lufs"] == got["integrated_lufs"]  # LFE does not touch loudness


@pytest.mark.parametrize("engine", [0, 1])
def test_formats_and_channel_counts(oracle, meters, engine):
    fs = 44100
    x = signals.seeded_noise(21, fs * 2 + 11, 2, -9.0)
    m = meters(engine=engine)
    check_against_oracle(oracle, m, signals.to_s16(x), fs)
    check_against_oracle(oracle, m, (x.astype(np.float64) * 2147483648.0).astype(np.int32), fs)
    check_against_oracle(oracle, m, x.astype(np.float64), fs)
    check_against_oracle(oracle, m, x[:, :1].copy(), fs)  # mono
    x8 = signals.seeded_noise(22, fs * 1 + 3, 8, -9.0)
    check_against_oracle(oracle, m, x8, fs)
    w = np.array([1.0, 0.5], dtype=np.float32)
    check_against_oracle(oracle, m, x, fs, weights=w)


@pytest.mark.parametrize("engine", [0, 1])
def test_default_layout_weights_known_answers(oracle, meters, engine):
    """NULL ch_weight = f_ebur128's weights on FFmpeg's default layout of the channel count (2.1, 4.0, 5.0, 5.1, 6.1,
    7.1): a tone in one channel at a time reads its analytic value through the C ABI (LFE: -inf), and equals the oracle."""
    import test_oracle_kat as kat
    m = meters(engine=engine)
    rows = kat.one_channel_answers(lambda pcm, fs: m.analyze(pcm, fs))
    kat.check_one_channel_answers(rows)
    for (ch, c, w, lufs), (_, _, _, ref) in zip(rows, kat.one_channel_answers(oracle.measure)):
        assert lufs == ref or abs(lufs - ref) < 1e-6, (ch, c)


@pytest.mark.parametrize("engine", [0, 1])
def test_edge_cases(oracle, meters, engine):
    m = meters(engine=engine)
    fs = 48000
    r = m.analyze(np.zeros((fs * 2, 2), dtype=np.float32), fs)
    assert r["integrated_lufs"] == -math.inf and r["true_peak_dbtp"] == -math.inf and r["n_gated"] == 0
    assert r["n_blocks"] == 17 and r["status"] == 0
    rc, a = hr.gain_decision(r["integrated_lufs"], r["true_peak_dbtp"])
    assert rc == hr.ERR_INVALID  # the caller's non-finite rule (src/analyzer.rs:284) still fires
    short = signals.stereo(signals.sine(fs, 1000.0, -20.0, 0.39))
    got, ref = check_against_oracle(oracle, m, short, fs)
    assert got["n_blocks"] == 0 and got["integrated_lufs"] == -math.inf
    tiny = np.ones((20, 2), dtype=np.float32)
    check_against_oracle(oracle, m, tiny, fs)
    exactly = signals.seeded_noise(1, 4800 * 4, 2, -20.0)  # exactly one block
    got, ref = check_against_oracle(oracle, m, exactly, fs)
    assert got["n_blocks"] == 1
    r = m.analyze(np.zeros((0, 2), dtype=np.float32), fs)
    assert r["status"] == 0 and r["n_blocks"] == 0 and r["integrated_lufs"] == -math.inf
    ragged = signals.seeded_noise(2, 4800 * 7 + 4799, 2, -20.0)  # trailing partial hop is discarded
    got, ref = check_against_oracle(oracle, m, ragged, fs)
    assert got["n_blocks"] == 4


def test_batch_order_errors_and_device_pcm(oracle, meters):
    import torch
    m = meters()
    fs = 44100
    a = signals.seeded_noise(31, fs * 3, 2, -12.0)
    b = signals.seeded_noise(32, fs * 2 + 9, 2, -25.0)
    c = signals.seeded_noise(33, 48000 * 2, 1, -6.0)
    td = torch.from_numpy(b).cuda()
    specs = [hr.TrackSpec(a, fs), hr.TrackSpec(a, 768001),  # beyond HR_MAX_SAMPLE_RATE
             hr.TrackSpec(td, fs), hr.TrackSpec(c, 48000), hr.TrackSpec(a, fs)]
    bad = hr.TrackSpec(a, fs)
    bad.c.channels = 9
    specs.append(bad)
    res = m.analyze_batch(specs)
    assert [r["status"] for r in res] == [0, hr.ERR_RATE, 0, 0, 0, hr.ERR_FORMAT]
    ra, rb, rc_ = (oracle.measure(x, f) for x, f in ((a, fs), (b, fs), (c, 48000)))
    for got, ref in ((res[0], ra), (res[2], rb), (res[3], rc_), (res[4], ra)):
        assert abs(got["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6
        assert got["peak_per_ch"] == ref["peak_per_ch"]
    assert m.analyze_batch([]) == []


def test_sample_rate_limit_and_no_exception_across_the_abi(meters):
    """fs = 0xFFFFFFFF once made rate_tables() allocate ~40 GB (or throw across the C ABI); now HR_ERR_RATE per track,
    from both batch entry points, without touching the PCM."""
    m = meters()
    x = np.zeros((4800, 2), dtype=np.float32)
    for fs in (0xFFFFFFFF, 768001):
        got = m.analyze_batch([hr.TrackSpec(x, fs)])[0]
        assert got["status"] == hr.ERR_RATE
        assert m.loudnorm_first_pass([hr.TrackSpec(x, fs)])[0]["status"] == hr.ERR_RATE
    assert m.analyze_batch([hr.TrackSpec(x, 48000)])[0]["status"] == 0


def test_small_staging_splits_batch(oracle):
    fs = 48000
    tracks = [signals.seeded_noise(50 + i, fs * 2, 2, -15.0 - i) for i in range(5)]
    with hr.Meter(stage_bytes=1 << 20) as m:  # 1 MiB: every track is its own group
        res = m.analyze_batch([hr.TrackSpec(t, fs) for t in tracks])
    for t, r in zip(tracks, res):
        ref = oracle.measure(t, fs)
        assert abs(r["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6 and r["peak_per_ch"] == ref["peak_per_ch"]


def test_synth_fill_matches_oracle_generator(oracle, meters):
    import torch
    m = meters()
    for fs, ch, frames in ((44100, 2, 44100 * 25), (48000, 1, 48000 * 3), (96000, 6, 9600 * 7)):
        buf = torch.empty((frames, ch), dtype=torch.float32, device="cuda")
        m.synth_fill(buf.data_ptr(), 987654321 + fs, fs, ch, frames)
        torch.cuda.synchronize()
        ref = oracle.synth_track(987654321 + fs, fs, ch, frames)
        assert np.array_equal(buf.cpu().numpy(), ref)


RANGE_KEYS = ("loudness_range_lu", "lra_low_lufs", "lra_high_lufs", "lra_threshold_lufs", "max_momentary_lufs",
              "max_shortterm_lufs")


def same_range(a, b, tol=1e-9):
    return all((math.isnan(a[k]) and math.isnan(b[k])) or a[k] == b[k] or abs(a[k] - b[k]) < tol for k in RANGE_KEYS)


def test_time_shards_equal_whole_stream(oracle, meters):
    """BASELINE config 5 on one device: 3 shards with IIR carry hand-off, boundary blocks, FIR halos."""
    fs = 48000
    pcm = signals.seeded_noise(77, 4800 * 61 + 1000, 2, -17.0)
    pcm[: fs * 2] *= 0.002
    for gate in ("ebur128", "loudnorm", "exact"):
        m = meters(gate)
        whole = m.analyze(pcm, fs)
        ref = oracle.measure(pcm, fs, gate_mode=GATES[gate])
        cuts = [0, 4800 * 20, 4800 * 27, pcm.shape[0]]
        shards = [hr.TrackSpec(pcm[cuts[i]:cuts[i + 1]].copy(), fs) for i in range(3)]
        heads = [m.shard_head(s) for s in shards]
        edges, partials = [], []
        # pass 1 on every shard (on a node: one per rank), edges travel to the right neighbour
        m2 = [hr.Meter(gate_mode=GATES[gate]) for _ in range(3)]
        for i, s in enumerate(shards):
            left_tail = np.ascontiguousarray(np.ctypeslib.as_array(edges[i - 1].tail)) if i else None
            right_head = heads[i + 1] if i < 2 else None
            edges.append(m2[i].shard_pass1(s, left_tail, right_head, first_hop=cuts[i] // 4800))
        for i in range(3):
            partials.append(m2[i].shard_pass2(edges[i - 1] if i else None))
        red = hr.HrShardPartial()
        for p in partials:  # the all-reduce
            for b in range(8001):
                red.hist[b] += p.hist[b]
            red.sum_kept += p.sum_kept
            red.n_kept += p.n_kept
            red.n_blocks += p.n_blocks
            for c in range(2):
                red.peak_per_ch[c] = max(red.peak_per_ch[c], p.peak_per_ch[c])
            for b in range(8001):
                red.st_hist[b] += p.st_hist[b]
            red.st_sum_kept += p.st_sum_kept
            red.st_n_kept += p.st_n_kept
            red.n_shortterm += p.n_shortterm
            red.max_m_power = max(red.max_m_power, p.max_m_power)
            red.max_s_power = max(red.max_s_power, p.max_s_power)
        red.range_ok = min(p.range_ok for p in partials)
        if gate == "exact":
            # BS.1770's text: no histogram, the relative gate needs the reduced sums first -> a second pass and reduction
            assert all(p.hist[b] == 0 for p in partials for b in (0, 4000, 8000))
            with pytest.raises(hr.HeadroomError):
                m2[1].shard_pass2(edges[0])          # pass 2 twice
            third = [mm.shard_pass3(red) for mm in m2]
            red.sum_gated = sum(p.sum_gated for p in third)
            red.n_gated = sum(p.n_gated for p in third)
            with pytest.raises(hr.HeadroomError):
                m2[0].shard_pass3(red)               # the shard is closed now
        else:
            with pytest.raises(hr.HeadroomError) as e:
                m2[0].shard_pass3(red)               # the histogram modes have no third pass
            assert e.value.code == hr.ERR_INVALID
        got = m2[0].shard_finish(red, 2)
        for mm in m2:
            mm.close()
        assert got["n_blocks"] == whole["n_blocks"] == ref["n_blocks"]
        assert got["n_gated"] == whole["n_gated"]
        assert abs(got["integrated_lufs"] - whole["integrated_lufs"]) < 1e-9
        assert abs(got["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6
        assert got["peak_per_ch"] == whole["peak_per_ch"] == ref["peak_per_ch"]
        # the middle shard has 7 hops: too short to hand 29 hop energies on, so the range statistics are withheld
        assert [e.range_ok for e in edges] == [0, 0, 1] and all(math.isnan(got[k]) for k in RANGE_KEYS)


def test_a_call_between_the_shard_passes_closes_the_shard(meters):
    fs = 48000
    pcm = signals.seeded_noise(78, 4800 * 12, 2, -17.0)
    with hr.Meter() as m:
        spec = hr.TrackSpec(pcm, fs)
        m.shard_pass1(spec, None, None)
        m.analyze(pcm[: 4800 * 3], fs)              # reuses the context's workspace
        with pytest.raises(hr.HeadroomError) as e:
            m.shard_pass2(None)
        assert e.value.code == hr.ERR_INVALID
        m.shard_pass1(spec, None, None)             # and the pair still works when nothing comes between
        m.shard_pass2(None)
        with pytest.raises(hr.HeadroomError):       # the same context twice in one sharded call
            hr.analyze_stream_sharded([m, m], spec)


def test_full_size_track_properties(oracle, meters):
    """A full 3-minute 44.1 kHz track (BASELINE config 1/2 shape), generated in HBM: oracle parity on the
    whole track, plus size-independent properties: gain linearity (+6.0206 dB exactly doubles) and
    invariance under channel swap."""
    import torch
    m = meters()
    fs, frames = 44100, 44100 * 180
    buf = torch.empty((frames, 2), dtype=torch.float32, device="cuda")
    m.synth_fill(buf.data_ptr(), 4242, fs, 2, frames)
    torch.cuda.synchronize()
    got = m.analyze(buf, fs)
    assert got["n_blocks"] == 1797
    ref = oracle.measure(buf.cpu().numpy(), fs)
    assert abs(got["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6
    assert got["peak_per_ch"] == ref["peak_per_ch"]
    assert got["n_gated"] == ref["n_gated"]
    half = (buf * 0.5).contiguous()
    g2 = m.analyze(half, fs)
    assert g2["peak_per_ch"] == [p * 0.5 for p in got["peak_per_ch"]]
    assert abs((got["integrated_lufs"] - g2["integrated_lufs"]) - 20 * math.log10(2)) < 0.011  # 0.01 LU bins
    swapped = buf.flip(1).contiguous()
    g3 = m.analyze(swapped, fs)
    assert g3["peak_per_ch"] == got["peak_per_ch"][::-1]
    assert abs(g3["integrated_lufs"] - got["integrated_lufs"]) < 1e-9


def test_mixed_rate_batch_on_fused_kernel(oracle, meters):
    """One batch of stereo f32 tracks at five rates: every kernel specialisation (tile shapes 1x32, 2x32,
    3x32 outputs, with and without the short last block) runs in the same call, every rate's run of wave
    items is padded to whole workgroups, and lengths go from two hops to several waves of rows."""
    m = meters()
    cases = [(44100, 2.05), (44100, 13.7), (48000, 7.3), (48000, 0.25), (96000, 6.61), (88200, 3.3), (32000, 9.9),
             (44100, 0.31), (96000, 0.2), (32000, 0.45)]
    pcms = []
    for i, (fs, secs) in enumerate(cases):
        p = signals.seeded_noise(900 + i, int(fs * secs) + 17 * i, 2, -14.0 - i)
        p[: len(p) // 3] *= 0.05
        pcms.append(p)
    got = m.analyze_batch([hr.TrackSpec(p, fs) for p, (fs, _) in zip(pcms, cases)])
    for g, p, (fs, _) in zip(got, pcms, cases):
        ref = oracle.measure(p, fs)
        assert g["status"] == 0 and g["n_blocks"] == ref["n_blocks"] and g["n_gated"] == ref["n_gated"], (fs, g, ref)
        if math.isfinite(ref["integrated_lufs"]):
            assert abs(g["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6, (fs, g, ref)
        assert g["peak_per_ch"] == ref["peak_per_ch"], (fs, g, ref)
    # the same tracks with sample-peak metering take the K-weighting-only build of the kernel
    ms = meters("ebur128", "sample")
    got_s = ms.analyze_batch([hr.TrackSpec(p, fs) for p, (fs, _) in zip(pcms, cases)])
    for g, gs in zip(got, got_s):
        assert gs["integrated_lufs"] == g["integrated_lufs"] and gs["n_gated"] == g["n_gated"]


@pytest.mark.parametrize("fs", [44100, 48000, 96000])
def test_low_frequency_content_carry_terms(oracle, meters, fs):
    """A 25 Hz tone on a DC offset makes the carry-in terms (2 s.q + s'Gs) as large as the zero-state energy:
    the truncated carry sums of the fused kernel must still give the hop energies of the sequential filter."""
    n = fs * 4 + 321
    t = np.arange(n) / fs
    pcm = np.stack([0.4 * np.sin(2 * np.pi * 25.0 * t) + 0.3, 0.2 * np.cos(2 * np.pi * 31.0 * t) - 0.25], axis=1).astype(np.float32)
    pcm += signals.seeded_noise(5, n, 2, -60.0)
    for engine in (0, 1, 2, 3):
        H, P = meters(engine=engine).block_energies(hr.TrackSpec(pcm, fs))
        Href = oracle.hop_sums(pcm, fs)
        assert np.max(np.abs(H / Href - 1)) < 1e-9
        check_against_oracle(oracle, meters(engine=engine), pcm, fs)


@pytest.mark.parametrize("idx", range(10))
def test_golden_vectors_on_gpu(meters, idx):
    """The committed golden vectors (tests/golden/, inputs regenerated on the device by hr_synth_fill), without
    any oracle call: loudness within 1e-6 LU, block counts equal, true peak bit-identical, in every mode."""
    import json
    import os
    import torch
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_vectors.json")) as fh:
        c = json.load(fh)["cases"][idx]
    fs, ch, frames = c["fs"], c["channels"], c["frames"]
    buf = torch.empty((frames, ch), dtype=torch.float32, device="cuda")
    meters().synth_fill(buf.data_ptr(), c["seed"], fs, ch, frames)
    torch.cuda.synchronize()
    assert [float(x) for x in buf[:4].reshape(-1).cpu()] == c["pcm_head"]
    H, P = meters().block_energies(hr.TrackSpec(buf, fs))
    want = np.array([float.fromhex(x) for x in c["block_powers_head"]])
    assert len(P[:32]) == len(want) and (len(want) == 0 or np.max(np.abs(P[:32] / want - 1)) < 1e-9)
    for key, exp in c["results"].items():
        gname, tname = key.split("/")
        got = meters(gname, tname).analyze(buf, fs)
        assert got["status"] == 0 and got["n_blocks"] == exp["n_blocks"] and got["n_gated"] == exp["n_gated"], (key, got)
        li = float.fromhex(exp["integrated_lufs"])
        assert got["integrated_lufs"] == li or abs(got["integrated_lufs"] - li) < 1e-6, (key, got)
        assert [float.hex(float(x)) for x in got["peak_per_ch"]] == exp["peak_per_ch"], (key, got)


@pytest.mark.parametrize("gate", ["ebur128", "loudnorm", "exact"])
def test_analyze_stream_sharded_in_process(oracle, meters, gate):
    """hr_analyze_stream_sharded: BASELINE config 5's shape in one process, here three contexts on one device
    (host PCM and device PCM), against the unsharded result and the oracle; short streams use fewer shards."""
    import torch
    fs = 48000
    pcm = signals.seeded_noise(314, 4800 * 95 + 2222, 2, -20.0)
    pcm[fs: 3 * fs] *= 0.01
    ms = [hr.Meter(gate_mode=GATES[gate]) for _ in range(3)]
    try:
        whole = meters(gate).analyze(pcm, fs)
        ref = oracle.measure(pcm, fs, gate_mode=GATES[gate])
        dev = torch.from_numpy(pcm).cuda()
        for spec in (hr.TrackSpec(pcm, fs), hr.TrackSpec(dev, fs)):
            got = hr.analyze_stream_sharded(ms, spec)
            assert got["n_blocks"] == whole["n_blocks"] == ref["n_blocks"] and got["n_gated"] == whole["n_gated"]
            assert abs(got["integrated_lufs"] - whole["integrated_lufs"]) < 1e-9
            assert abs(got["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6
            assert got["peak_per_ch"] == whole["peak_per_ch"] == ref["peak_per_ch"]
            # 31 + 32 + 32 hops: every shard can hand its last 29 hop energies on, the 3 s windows match the whole stream's
            assert same_range(got, whole), (got, whole)
            assert got["loudness_range_lu"] > 0.0 and math.isfinite(got["max_shortterm_lufs"])
        short = pcm[: 4800 * 8 + 100].copy()  # eight hops: one shard only
        got = hr.analyze_stream_sharded(ms, hr.TrackSpec(short, fs))
        ref = oracle.measure(short, fs, gate_mode=GATES[gate])
        assert got["n_blocks"] == ref["n_blocks"] and abs(got["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6
        assert got["peak_per_ch"] == ref["peak_per_ch"]
    finally:
        for m in ms:
            m.close()


def test_in_process_shards_through_a_one_rank_rccl_communicator(oracle, monkeypatch):
    """hr_analyze_stream_sharded moves its records with RCCL when its contexts sit on distinct GPUs.  A one-GPU box can
    show only the smallest communicator: HR_SHARD_RCCL=1 sends a single shard's partial through ncclAllGather on a one-rank
    communicator (librccl opened at run time, the collective ordered on the context's stream, the gathered image
    reduced and finished on the device)."""
    fs = 48000
    pcm = signals.seeded_noise(4711, 4800 * 31 + 99, 2, -17.0)
    pcm[: fs] *= 0.01
    ref = oracle.measure(pcm, fs)
    monkeypatch.setenv("HR_SHARD_RCCL", "1")
    with hr.Meter(device=0) as m:
        got = hr.analyze_stream_sharded([m], hr.TrackSpec(pcm, fs))
        whole = m.analyze(pcm, fs)
    assert got["peak_per_ch"] == whole["peak_per_ch"] == ref["peak_per_ch"]
    assert got["n_blocks"] == ref["n_blocks"] and got["n_gated"] == ref["n_gated"]
    assert abs(got["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6 and abs(got["integrated_lufs"] - whole["integrated_lufs"]) < 1e-9
    for k in ("loudness_range_lu", "max_momentary_lufs", "max_shortterm_lufs"):
        assert abs(got[k] - whole[k]) < 1e-9


@pytest.mark.parametrize("fs", [44100, 48000])
def test_integer_stereo_tracks_take_the_fused_kernel(oracle, meters, fs):
    """s24 stereo is widened to f32 on the device (exactly), s16 stereo read directly, and measured by the fused kernel: same numbers
    as the generic kernels and the oracle, from host and device buffers, also when the widened copies do not fit
    one group."""
    import torch
    pcm = signals.seeded_noise(77 + fs, fs * 4 + 1001, 2, -11.0)
    pcm[: fs] *= 0.02
    s16 = signals.to_s16(pcm)
    s24 = signals.to_s24_bytes(pcm)
    ref16 = oracle.measure(s16, fs)
    ref24 = oracle.measure(s24, fs, channels=2, fmt=hr.FMT_S24)
    for engine in (0, 1):
        m = meters(engine=engine)
        got = m.analyze_batch([hr.TrackSpec(s16, fs), hr.TrackSpec(s24, fs, channels=2, fmt=hr.FMT_S24),
                               hr.TrackSpec(torch.from_numpy(s16).cuda(), fs),
                               hr.TrackSpec(torch.from_numpy(s24).cuda(), fs, channels=2, fmt=hr.FMT_S24)])
        for g, ref in zip(got, (ref16, ref24, ref16, ref24)):
            assert g["status"] == 0 and g["n_blocks"] == ref["n_blocks"] and g["n_gated"] == ref["n_gated"]
            assert abs(g["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6
            assert g["peak_per_ch"] == ref["peak_per_ch"]
    with hr.Meter(stage_bytes=1 << 20) as small:  # 16 MiB of widened copies per group: the six tracks need three groups
        got = small.analyze_batch([hr.TrackSpec(torch.from_numpy(s16).cuda(), fs) for _ in range(6)])
        assert all(g["peak_per_ch"] == ref16["peak_per_ch"] and abs(g["integrated_lufs"] - ref16["integrated_lufs"]) < 1e-6
                   for g in got)


# ---- loudness range, maximum momentary / short-term loudness ------------------------------

def check_range(oracle, m, pcm, fs, gate, **kw):
    ref = oracle.range_stats(pcm, fs, GATES[gate], **kw)
    got = m.analyze(pcm, fs, **kw)
    for k_got, k_ref in (("loudness_range_lu", "lra"), ("lra_low_lufs", "lra_low"), ("lra_high_lufs", "lra_high")):
        if math.isnan(ref[k_ref]):
            assert math.isnan(got[k_got]), (k_got, got, ref)
        else:
            assert abs(got[k_got] - ref[k_ref]) < 1e-9, (k_got, got, ref)  # bin positions: integers
    for k_got, k_ref in (("lra_threshold_lufs", "lra_threshold"), ("max_momentary_lufs", "max_momentary"),
                         ("max_shortterm_lufs", "max_shortterm")):
        if math.isnan(ref[k_ref]) or math.isinf(ref[k_ref]):
            assert got[k_got] == ref[k_ref] or (math.isnan(got[k_got]) and math.isnan(ref[k_ref])), (k_got, got, ref)
        else:
            assert abs(got[k_got] - ref[k_ref]) < 1e-6, (k_got, got, ref)
    return got


@pytest.mark.parametrize("gate", ["ebur128", "loudnorm"])
@pytest.mark.parametrize("case", [1, 2, 3, 4])
def test_ebu3342_loudness_range_on_gpu(oracle, meters, case, gate):
    pcm, exp = signals.ebu3342_case(case)
    got = check_range(oracle, meters(gate), pcm, 48000, gate)
    assert abs(got["loudness_range_lu"] - exp) <= 1.0


@pytest.mark.parametrize("engine", [0, 1])
@pytest.mark.parametrize("gate", ["ebur128", "exact", "loudnorm"])
def test_range_stats_noise_multichannel_and_short(oracle, meters, gate, engine):
    m = meters(gate, "swr192k", engine)
    fs = 44100
    rng = np.random.default_rng(5)
    x = signals.seeded_noise(61, fs * 42, 2, -14.0)
    env = np.repeat(10.0 ** (rng.uniform(-30.0, 0.0, size=42) / 20.0), fs).astype(np.float32)  # a new level every second
    check_range(oracle, m, x * env[:, None], fs, gate)
    x6 = signals.seeded_noise(62, 48000 * 8, 6, -20.0)
    x6[: 48000 * 3] *= 0.05
    check_range(oracle, m, x6, 48000, gate)
    check_range(oracle, m, x[: fs * 2], fs, gate)            # momentary blocks, no 3 s window
    check_range(oracle, m, x[: fs // 4], fs, gate)           # no block at all
    check_range(oracle, m, np.zeros((fs * 4, 2), np.float32), fs, gate)  # silence


# ---- the literal af_loudnorm first pass (what src/analyzer.rs:261 asks ffmpeg for) -------------------------------

def test_loudnorm_first_pass_matches_oracle(oracle, meters):
    m = meters()
    cases = []
    for fs in (44100, 48000, 96000, 192000):
        x = signals.seeded_noise(300 + fs, int(fs * 4.3) + 17, 2, -15.0)
        x[: fs] *= 0.01
        cases.append((x, fs, {}))
    cases.append((signals.to_s16(signals.seeded_noise(41, 44100 * 3, 1, -8.0)), 44100, {}))
    x6 = signals.seeded_noise(42, 48000 * 3 + 5, 6, -12.0)
    cases.append((x6, 48000, {}))
    cases.append((signals.seeded_noise(43, 20, 2, -12.0), 48000, {}))          # shorter than the resampler's window
    cases.append((np.zeros((48000, 2), np.float32), 48000, {}))                 # silence
    got = m.loudnorm_first_pass([hr.TrackSpec(p, fs, **kw) for p, fs, kw in cases])
    for (p, fs, kw), g in zip(cases, got):
        ref = oracle.loudnorm_measure(p, fs, **kw)
        assert g["status"] == 0 and g["n_blocks"] == ref["n_blocks"], (fs, g, ref)
        assert g["peak_per_ch"] == ref["peak_per_ch"], (fs, g, ref)             # same f64 FMA chain: bit-identical
        assert g["true_peak_dbtp"] == ref["true_peak_dbtp"]
        if math.isfinite(ref["integrated_lufs"]):
            # the oracle runs libebur128's 4th-order direct form II, the GPU two biquads from zero state per hop:
            # block energies agree to ~1e-12, the 0.1 LU histogram makes the result a step function of them
            assert abs(g["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6, (fs, g, ref)
        else:
            assert g["integrated_lufs"] == ref["integrated_lufs"]
    # the context's own modes are untouched by the call
    pcm, fs, _ = cases[0]
    check_against_oracle(oracle, m, pcm, fs)
    bad = m.loudnorm_first_pass([hr.TrackSpec(signals.seeded_noise(1, 1000, 2), 384000)])
    assert bad[0]["status"] == hr.ERR_RATE
    # a context whose own tables hold the BS.1770 bank still resamples with libswresample's
    mb = meters("ebur128", "bs1770")
    mb.analyze(pcm, fs)                                   # builds the context's tables for this rate first
    g = mb.loudnorm_first_pass([hr.TrackSpec(pcm, fs)])[0]
    ref = oracle.loudnorm_measure(pcm, fs)
    assert g["peak_per_ch"] == ref["peak_per_ch"] and abs(g["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6


def test_concurrent_callers_share_one_context(oracle, meters):
    """The drop-in caller keeps rayon's par_iter shape (src/cli.rs:407-415): many host threads, one context.  Calls are
    serialised inside the library; every thread must get its own track's numbers back."""
    from concurrent.futures import ThreadPoolExecutor
    m = meters()
    fs = 48000
    tracks = [signals.seeded_noise(500 + i, fs * 2 + 997 * i, 2 if i % 3 else 6, -10.0 - i) for i in range(12)]
    refs = [oracle.measure(t, fs) for t in tracks]

    def work(i):
        out = []
        for _ in range(3):
            out.append(m.analyze(tracks[i], fs))
        return out

    with ThreadPoolExecutor(max_workers=6) as ex:
        got = list(ex.map(work, range(len(tracks))))
    for i, runs in enumerate(got):
        for g in runs:
            assert g["status"] == 0 and g["n_blocks"] == refs[i]["n_blocks"]
            assert abs(g["integrated_lufs"] - refs[i]["integrated_lufs"]) < 1e-6
            assert g["peak_per_ch"] == refs[i]["peak_per_ch"]


@pytest.mark.parametrize("tp", ["swr192k", "bs1770"])
@pytest.mark.parametrize("fs", [8000, 16000, 32000, 64000, 88200, 176400])
def test_other_sample_rates(oracle, meters, fs, tp):
    """Rates beside the three of the BASELINE configs: other tile shapes of the fused kernel (32 kHz: three full 32-row
    blocks; 88.2 kHz: 35 outputs per tile) and, below 32 kHz in the libswresample mode, the generic kernel."""
    pcm = signals.seeded_noise(900 + fs, int(fs * 3.7) + 13, 2, -11.0)
    pcm[: fs // 2] *= 0.02
    check_against_oracle(oracle, meters("ebur128", tp), pcm, fs, "ebur128", tp)
    x5 = signals.seeded_noise(901 + fs, fs * 2 + 5, 5, -15.0)
    check_against_oracle(oracle, meters("ebur128", tp), signals.to_s16(x5), fs, "ebur128", tp)


@pytest.mark.parametrize("fs", [22050, 11025, 12345, 64000])
def test_rates_that_libswresample_interpolates_between_phases(oracle, meters, fs):
    """22.05 kHz needs 1280 exact phases, 11.025 kHz 2560, 12.345 kHz 128000: more than libswresample's 1024, so it keeps
    1024 phases and interpolates linearly between neighbours (resample_linear).  The reference hands every file to ffmpeg
    whatever its rate (src/analyzer.rs:253-265); so does the meter now (generic true-peak kernel; 64 kHz = 3 exact phases
    rides along as the control).  Peaks bit-identical to the oracle's restatement, both entry points."""
    pcm = signals.seeded_noise(77 + fs, fs * 3 + 17, 2, -9.0)
    pcm[fs // 2: fs // 2 + 40] *= 3.0                      # a passage whose true peak sits between the samples
    for gate in ("ebur128", "loudnorm"):
        got, ref = check_against_oracle(oracle, meters(gate), pcm, fs, gate)
        assert got["true_peak_dbtp"] > 20 * math.log10(np.abs(pcm).max()) - 1e-9
    check_against_oracle(oracle, meters("ebur128", "bs1770"), pcm, fs, "ebur128", "bs1770")
    x3 = signals.to_s16(signals.seeded_noise(78 + fs, fs * 2 + 5, 3, -15.0))
    check_against_oracle(oracle, meters(), x3, fs)
    # the literal first pass of af_loudnorm (resample to 192 kHz, then measure) at the same rate
    g = meters().loudnorm_first_pass([hr.TrackSpec(pcm, fs)])[0]
    ref = oracle.loudnorm_measure(pcm, fs)
    assert g["status"] == 0 and g["peak_per_ch"] == ref["peak_per_ch"]
    assert abs(g["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6


@pytest.mark.parametrize("engine", [0, 1])
@pytest.mark.parametrize("fs", [44100, 88200, 48000])
def test_isolated_peaks_between_output_positions(oracle, meters, fs, engine):
    """At 44.1 kHz only every 147th input frame is itself a 192 kHz output (at 88.2 kHz likewise, at 48 kHz every frame
    is): the largest sample of a sparse signal is then no output at all and the true peak is BELOW the sample peak.
    (Found by tools/soak_long.py: the fused kernel took the sample peak for an output at every rate.)"""
    n = fs * 3 + 77
    pcm = np.zeros((n, 2), np.float32)
    rng = np.random.default_rng(fs)
    idx = rng.integers(200, n - 200, size=40)
    idx = idx[idx % 147 != 0]
    pcm[idx, 0] = rng.uniform(0.2, 0.95, size=idx.size).astype(np.float32)
    pcm[idx[::2] + 50, 1] = -0.9
    got, ref = check_against_oracle(oracle, meters(engine=engine), pcm, fs)
    if fs != 48000:
        assert got["peak_per_ch"][0] < float(np.abs(pcm[:, 0]).max())
    else:
        assert got["peak_per_ch"][0] >= float(np.abs(pcm[:, 0]).max())


@pytest.mark.parametrize("kind", ["s16", "s24"])
@pytest.mark.parametrize("tp", ["swr192k", "bs1770", "sample"])
@pytest.mark.parametrize("fs", [44100, 48000, 96000, 88200])
def test_integer_stereo_is_read_directly_by_the_fused_kernel(oracle, meters, fs, tp, kind, monkeypatch):
    """Stereo s16 / s24 takes the fused kernel's full pass as it is (64- / 96-byte pieces, widened to f32 as they leave
    LDS); with HR_NO_S16_DIRECT=1, and in the sample-peak mode, it is widened into f32 planes first.  Same numbers
    either way, and the oracle's: noise with a quiet start, a full-scale sine (candidate lists overflow), a clipped
    signal, an odd start (device pointer 16 but not 128 bytes aligned), a track of two hops and a bit."""
    import torch
    n = fs * 7 + 555
    noise = signals.seeded_noise(5 + fs, n, 2, -9.0)
    noise[: fs] *= 0.01
    t = np.arange(n) / fs
    sine = np.stack([np.sin(2 * np.pi * 997.0 * t), np.sin(2 * np.pi * 3001.0 * t + 1.0)], axis=1) * 0.999
    clipped = np.clip(signals.seeded_noise(6 + fs, n, 2, -3.0) * 3.0, -1.0, 1.0) * 0.9
    sigs = (noise, sine, clipped, noise[: fs // 5 + 99])
    if kind == "s16":
        cases, kw, fmt = [signals.to_s16(x) for x in sigs], {}, hr.FMT_S16
        dev = torch.from_numpy(np.concatenate([np.zeros((4, 2), np.int16), cases[0]])).cuda()  # 16 bytes in: odd start
    else:
        cases, kw, fmt = [signals.to_s24_bytes(x) for x in sigs], {"channels": 2, "fmt": hr.FMT_S24}, hr.FMT_S24
        dev = torch.from_numpy(np.concatenate([np.zeros(16, np.uint8), cases[0].reshape(-1)])).cuda()
    odd = hr.TrackSpec(dev.data_ptr() + 16, fs, channels=2, fmt=fmt, frames=len(sigs[0]))
    refs = [oracle.measure(x, fs, gate_mode=0, tp_mode=TPS[tp], **kw) for x in cases]
    direct = meters("ebur128", tp).analyze_batch([hr.TrackSpec(x, fs, **kw) for x in cases] + [odd])
    monkeypatch.setenv("HR_NO_S16_DIRECT", "1")
    m = hr.Meter(device=0, tp_mode=TPS[tp])
    try:
        widened = m.analyze_batch([hr.TrackSpec(x, fs, **kw) for x in cases] + [odd])
    finally:
        m.close()
    for g, w, ref in zip(direct, widened, refs + [refs[0]]):
        assert g["status"] == 0 and g["n_blocks"] == ref["n_blocks"] and g["n_gated"] == ref["n_gated"]
        assert g["peak_per_ch"] == w["peak_per_ch"] == ref["peak_per_ch"]
        assert g["integrated_lufs"] == w["integrated_lufs"] or (math.isnan(g["integrated_lufs"]) and math.isnan(w["integrated_lufs"]))
        if math.isfinite(ref["integrated_lufs"]):
            assert abs(g["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6


END_RAGGED = [0, 1, 7, 16, 21, 22, 40, 63, 64, 65, 100, 333, 1001, 3000]


def end_of_track_cases(oracle, m, fs, ragged, tp, kind):
    hop = fs // 10
    frames = 3 * hop + ragged
    base = signals.seeded_noise(4242 + ragged, frames, 2, -40.0)
    for dist, chan in [(1, 0), (5, 1), (16, 0), (17, 1), (18, 0), (33, 1), (49, 0), (ragged + 3, 1), (ragged + 40, 0), (hop + 5, 1)]:
        if dist > frames:
            continue
        pcm = base.copy()
        pcm[frames - dist, chan] = 0.9
        pcm[frames - dist, 1 - chan] = -0.3
        if kind == "s16":
            check_against_oracle(oracle, m, signals.to_s16(pcm), fs, tp=tp)
        elif kind == "s24":
            check_against_oracle(oracle, m, signals.to_s24_bytes(pcm), fs, tp=tp, channels=2, fmt=hr.FMT_S24)
        else:
            check_against_oracle(oracle, m, pcm, fs, tp=tp)


@pytest.mark.parametrize("engine", [2, 3])
@pytest.mark.parametrize("tp", ["swr192k", "bs1770"])
@pytest.mark.parametrize("fs", [44100, 48000])
@pytest.mark.parametrize("ragged", END_RAGGED)
def test_largest_sample_near_the_end_of_the_track(oracle, meters, fs, ragged, tp, engine):
    """libswresample is not flushed: the last 16 positions produce no output, so a sample there is not a true peak
    (BS.1770 bank: every position counts).  The track's end is where the fused kernel's final rows, the zeroed frames
    and the generic tail kernel meet: the largest sample is put at several distances from the end, in and out of the
    dropped zone, for ragged tails of every kind (none, shorter than a DMA chunk, odd, around the limits at which the
    tail becomes a fused row)."""
    end_of_track_cases(oracle, meters(tp=tp, engine=engine), fs, ragged, tp, "f32")


@pytest.mark.parametrize("kind", ["s16", "s24"])
@pytest.mark.parametrize("fs", [44100, 48000])
@pytest.mark.parametrize("ragged", END_RAGGED)
def test_largest_sample_near_the_end_of_an_integer_track(oracle, meters, fs, ragged, kind):
    """The same for stereo s16 / s24 read directly by the fused kernel (four / eight frames per 16-byte chunk: other
    alignment cases at the track's end)."""
    end_of_track_cases(oracle, meters(tp="swr192k"), fs, ragged, "swr192k", kind)


@pytest.mark.parametrize("fs", [88200, 96000, 32000, 8000])
@pytest.mark.parametrize("ragged", [0, 23, 27, 28, 64, 501])
def test_largest_sample_near_the_end_other_rates(oracle, meters, fs, ragged):
    """Hops of other lengths (88.2 kHz: a row fetches 28 frames past its hop; 8 kHz: a hop is 50 pieces)."""
    end_of_track_cases(oracle, meters(tp="swr192k"), fs, ragged, "swr192k", "f32")


@pytest.mark.parametrize("case", ["s16_44k", "s24_44k", "f32_22k", "s16_44k_odd_hops", "s24_44k_ragged8"])
def test_device_tracks_whose_last_row_is_off_the_16_byte_grid(oracle, meters, case):
    """Rows of the fused kernel start on a 16-byte boundary of the track's image only when hop x frame bytes is a multiple of
    16: not at 44.1 kHz s16 (17 640 bytes: every other row starts 8 bytes in), 44.1 kHz s24 (26 460: 12 bytes in), nor at
    odd hop lengths (22.05 kHz f32).  The DMA's 16-byte chunks count from the ROW's start, so a final row may be fused only
    when the track ends on ITS chunk grid (the host tested the track's own grid, and the last chunk of such a row
    read 8 / 12 bytes past the caller's buffer).  Device tracks that end exactly with their allocation, in the geometries
    that decide differently now; the numbers must be the oracle's either way."""
    import torch
    fs, kind, frames = {"s16_44k": (44100, "s16", 4410 * 20),            # even hop count, frames % 4 == 0: row 19 starts 8 bytes in
                        "s24_44k": (44100, "s24", 4410 * 20),            # frames % 8 == 0: row 19 starts 12 bytes in
                        "f32_22k": (22050, "f32", 2205 * 20),            # odd hop length: every other row starts 8 bytes in
                        "s16_44k_odd_hops": (44100, "s16", 4410 * 21 + 2),  # the last row starts on the grid, the track ends off it
                        "s24_44k_ragged8": (44100, "s24", 4410 * 21 + 8)}[case]
    pcm = signals.seeded_noise(77 + frames, frames, 2, -12.0)
    pcm[frames - 3, 0] = 0.97   # the largest sample sits in the last chunk of the track
    pcm[frames - 20, 1] = -0.95
    if kind == "s16":
        host = signals.to_s16(pcm)
        dev = torch.from_numpy(host.reshape(-1)).cuda()
        spec = hr.TrackSpec(dev.data_ptr(), fs, channels=2, fmt=hr.FMT_S16, frames=frames)
        ref = oracle.measure(host, fs)
    elif kind == "s24":
        host = signals.to_s24_bytes(pcm)
        dev = torch.from_numpy(host.reshape(-1)).cuda()
        spec = hr.TrackSpec(dev.data_ptr(), fs, channels=2, fmt=hr.FMT_S24, frames=frames)
        ref = oracle.measure(host, fs, channels=2, fmt=hr.FMT_S24)
    else:
        dev = torch.from_numpy(pcm).cuda()
        spec = hr.TrackSpec(dev.data_ptr(), fs, channels=2, fmt=hr.FMT_F32, frames=frames)
        ref = oracle.measure(pcm, fs)
    for engine in (0, 2, 3):
        got = meters(engine=engine).analyze_batch([spec])[0]
        assert got["status"] == 0 and got["n_blocks"] == ref["n_blocks"] and got["n_gated"] == ref["n_gated"]
        assert abs(got["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6, (engine, got, ref)
        assert got["peak_per_ch"] == ref["peak_per_ch"], (engine, got, ref)


def test_misaligned_device_tracks_are_restaged(oracle, meters):
    """Device PCM at an address the fast kernels cannot take (stereo f32 off the 16-byte grid, integer PCM off the dword
    grid) is copied into the staging buffer instead of falling to the generic kernels: same numbers, and the fused
    kernel's timer shows that it ran."""
    import torch
    fs = 44100
    pcm = signals.seeded_noise(31, fs * 3 + 1235, 2, -14.0)
    s16 = signals.to_s16(pcm)
    s24 = signals.to_s24_bytes(pcm)
    d32 = torch.from_numpy(np.concatenate([np.zeros((1, 2), np.float32), pcm])).cuda()         # + 8 bytes
    d16 = torch.from_numpy(np.concatenate([np.zeros(1, np.int16), s16.reshape(-1)])).cuda()      # + 2 bytes
    d24 = torch.from_numpy(np.concatenate([np.zeros(1, np.uint8), s24.reshape(-1)])).cuda()      # + 1 byte
    specs = [hr.TrackSpec(d32.data_ptr() + 8, fs, channels=2, fmt=hr.FMT_F32, frames=len(pcm)),
             hr.TrackSpec(d16.data_ptr() + 2, fs, channels=2, fmt=hr.FMT_S16, frames=len(pcm)),
             hr.TrackSpec(d24.data_ptr() + 1, fs, channels=2, fmt=hr.FMT_S24, frames=len(pcm))]
    refs = [oracle.measure(pcm, fs), oracle.measure(s16, fs), oracle.measure(s24, fs, channels=2, fmt=hr.FMT_S24)]
    m = meters()
    for spec, ref in zip(specs, refs):
        got = m.analyze_batch([spec])[0]
        assert got["status"] == 0 and got["n_gated"] == ref["n_gated"]
        assert abs(got["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6
        assert got["peak_per_ch"] == ref["peak_per_ch"]
        assert m.last_timing()[1] > 0.0     # the fused kernel's own event pair
    got = m.analyze_batch(specs + [hr.TrackSpec(pcm, fs)])
    assert [g["peak_per_ch"] for g in got] == [r["peak_per_ch"] for r in refs + [refs[0]]]


@pytest.mark.parametrize("layout", ["mono_f32", "5.1_f32", "5.1_s24", "3ch_s16"])
@pytest.mark.parametrize("ragged", [0, 22, 65, 333, 1001])
def test_largest_sample_near_the_end_of_a_track_measured_through_planes(oracle, meters, layout, ragged):
    """Mono and multichannel tracks are split into stereo planes that take the fused kernel: the planes end like tracks."""
    fs = 48000
    ch = {"mono_f32": 1, "5.1_f32": 6, "5.1_s24": 6, "3ch_s16": 3}[layout]
    frames = 3 * 4800 + ragged
    base = signals.seeded_noise(999 + ragged + ch, frames, ch, -40.0)
    m = meters()
    for dist, chan in [(1, 0), (16, ch - 1), (17, 0), (40, ch - 1), (ragged + 20, 0)]:
        pcm = base.copy()
        pcm[frames - dist, chan] = 0.9
        if layout.endswith("s24"):
            check_against_oracle(oracle, m, signals.to_s24_bytes(pcm), fs, channels=ch, fmt=hr.FMT_S24)
        elif layout.endswith("s16"):
            check_against_oracle(oracle, m, signals.to_s16(pcm), fs)
        else:
            check_against_oracle(oracle, m, pcm, fs)


@pytest.mark.parametrize("ragged", [0, 5, 64, 777])
def test_largest_sample_near_the_end_of_a_sharded_stream(oracle, meters, ragged):
    """The last time shard ends the stream: its final rows get the same care (the other shards end in a halo)."""
    fs = 48000
    frames = 4800 * 40 + ragged
    base = signals.seeded_noise(555 + ragged, frames, 2, -35.0)
    ms = [hr.Meter() for _ in range(3)]
    try:
        for dist in (1, 16, 17, 30, ragged + 5):
            pcm = base.copy()
            pcm[frames - dist, 0] = 0.9
            pcm[4800 * 13 - 3, 1] = 0.8      # and one right in front of a cut (hr_analyze_stream_sharded cuts at hop 13 and 26)
            ref = oracle.measure(pcm, fs)
            got = hr.analyze_stream_sharded(ms, hr.TrackSpec(pcm, fs))
            assert got["peak_per_ch"] == ref["peak_per_ch"], (dist, got, ref)
            assert abs(got["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6
    finally:
        for mm in ms:
            mm.close()


@pytest.mark.parametrize("engine", [2, 3, 4])
@pytest.mark.parametrize("fs", [44100, 48000, 96000])
def test_row_lengths_of_the_fused_kernel(oracle, meters, fs, engine):
    """hr_config.engine 2 / 3 / 4: rows of a whole hop, of a fraction of a hop, and both in one track (whole-hop rows for
    the hops that fill waves, shorter ones behind them): the carry crosses the sections' boundary.  Low-frequency content on
    a DC offset makes the carry terms as large as the zero-state energies; tracks of 70 and 133 hops and a bit."""
    for seconds, ragged in ((7.0, 1234), (13.3, 0), (13.3, 4000)):
        n = int(fs * seconds) + ragged
        t = np.arange(n) / fs
        pcm = np.stack([0.4 * np.sin(2 * np.pi * 25.0 * t) + 0.3, 0.2 * np.cos(2 * np.pi * 31.0 * t) - 0.25], axis=1).astype(np.float32)
        pcm += signals.seeded_noise(int(seconds) + ragged, n, 2, -30.0)
        pcm[n - 40, 0] = 0.97          # a peak near the end, in front of the positions without output
        m = meters(engine=engine)
        H, P = m.block_energies(hr.TrackSpec(pcm, fs))
        Href = oracle.hop_sums(pcm, fs)
        assert H.shape == Href.shape and np.max(np.abs(H / Href - 1)) < 1e-9
        check_against_oracle(oracle, m, pcm, fs)
