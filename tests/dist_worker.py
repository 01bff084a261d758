"""Worker for the multi-process tests (launched by torch.distributed.run, world size >= 2, backend gloo).

mode "cpu": the N>1 host logic of headroom_amd.sharded without a GPU -- track partition / result gather,
            neighbour exchange and all-reduce on ctypes structs, and the data flow of a time-sharded
            stream with a recording stand-in for the meter.
mode "gpu": the same drivers over the real C ABI, every rank with its own context on cuda:0, checked
            against the unsharded result and the oracle.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import headroom_amd as hr  # noqa: E402
from headroom_amd import sharded  # noqa: E402


class RecordingMeter:
    """Stand-in with the Meter's shard interface: computes nothing, records what the driver hands it."""

    def __init__(self, rank, gate_mode=hr.GATE_EBUR128):
        self.rank = rank
        self.gate_mode = gate_mode
        self.seen = {}

    def _frames(self, spec, sl):
        out = np.zeros((16, hr.MAX_CH))
        out[:, :spec.pcm.shape[1]] = spec.pcm[sl]
        return out

    def shard_head(self, spec):
        return self._frames(spec, slice(0, 16))

    def shard_tail(self, spec):
        return self._frames(spec, slice(-16, None))

    def shard_pass1(self, spec, left_tail, right_head, first_hop=0):
        self.seen["left_tail"], self.seen["right_head"], self.seen["first_hop"] = left_tail, right_head, first_hop
        e = hr.HrShardEdge()
        e.n_hops = spec.pcm.shape[0] // 4800
        np.ctypeslib.as_array(e.rec)[:] = 1000.0 * self.rank + np.arange(6 * 8 * 9).reshape(6, 8, 9)
        np.ctypeslib.as_array(e.tail)[:] = self.shard_tail(spec)
        return e

    def shard_pass2(self, left_edge):
        self.seen["left_edge"] = left_edge
        p = hr.HrShardPartial()
        p.hist[self.rank] = 1 + self.rank
        p.hist[8000] = 7
        p.sum_kept = 0.5 + self.rank
        p.n_kept = 10 + self.rank
        p.n_blocks = 100 + self.rank
        p.peak_per_ch[0] = 0.25 * (self.rank + 1)
        p.peak_per_ch[1] = 1.0 / (self.rank + 1)
        p.st_hist[100 + self.rank] = 3
        p.st_sum_kept = 0.25 * (self.rank + 1)
        p.st_n_kept = 2 + self.rank
        p.n_shortterm = 50 + self.rank
        p.max_m_power = 1.0 + self.rank
        p.max_s_power = 5.0 - self.rank
        p.range_ok = 0 if self.rank == 1 else 1
        return p

    def shard_pass3(self, red):
        self.seen["reduced_for_pass3"] = (red.sum_kept, red.n_kept)
        p = hr.HrShardPartial()
        p.sum_gated = 0.125 * (self.rank + 1)
        p.n_gated = 3 + self.rank
        return p

    def shard_finish(self, red, channels):
        return red


    # ---- the device-resident protocol: the records are written into / read from the caller's buffers (host memory here) ----
    def shard_pass1_device(self, spec, left_tail, right_head, first_hop, edge_ptr):
        e = self.shard_pass1(spec, left_tail, right_head, first_hop)
        C.memmove(edge_ptr, C.addressof(e), C.sizeof(e))

    def shard_pass2_device(self, left_edge_ptr, partial_ptr):
        left = hr.HrShardEdge.from_address(left_edge_ptr) if left_edge_ptr else None
        p = self.shard_pass2(left)
        C.memmove(partial_ptr, C.addressof(p), C.sizeof(p))

    def shard_finish_gathered(self, gathered_ptr, n, channels):
        parts = (hr.HrShardPartial * n).from_address(gathered_ptr)
        self.seen["finish_calls"] = self.seen.get("finish_calls", 0) + 1
        if any(p.reserved for p in parts):
            return hr.SHARD_AGAIN, None
        return hr.OK, [hr.HrShardPartial.from_buffer_copy(p) for p in parts]

    def shard_redo_device(self, partial_ptr):
        hr.HrShardPartial.from_address(partial_ptr).reserved = 0
        self.seen["redo"] = True


class FakeSpec:
    def __init__(self, pcm):
        self.pcm = pcm


def mode_cpu(rank, world):
    # --- track batches: every track analysed exactly once, results back in input order ------------
    for n in (0, 1, world, 7, 1000):
        lo, hi = sharded.partition_tracks(n, world, rank)
        mine = [("track", i) for i in range(lo, hi)]
        allr = sharded.gather_results(dist, mine, n, world)
        assert allr == [("track", i) for i in range(n)], (n, allr[:4])
        sizes = [b - a for a, b in (sharded.partition_tracks(n, world, r) for r in range(world))]
        assert sum(sizes) == n and max(sizes) - min(sizes) <= 1

    # --- unequal tracks: longest-first greedy by bytes; every track exactly once, loads within one track of each other
    sizes = [((7 * i) % 13 + 1) * 1000 for i in range(41)]
    parts = sharded.partition_tracks_by_bytes(sizes, world)
    assert sorted(i for p in parts for i in p) == list(range(41))
    loads = [sum(sizes[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= max(sizes)

    # --- stream cuts ------------------------------------------------------------------------------
    fs = 48000
    frames = 4800 * 61 + 1000
    cuts = sharded.cut_stream(frames, fs, world)
    assert cuts[0] == 0 and cuts[-1] == frames and len(cuts) == world + 1
    for a, b in zip(cuts[:-2], cuts[1:-1]):
        assert (b - a) % 4800 == 0 and (b - a) // 4800 >= sharded.MIN_SHARD_HOPS
    short = sharded.cut_stream(4800 * 7, fs, world)  # too short to split: one active shard
    assert short == [0] + [4800 * 7] * world
    assert sharded.cut_stream(0, fs, world) == [0] * (world + 1)

    # --- data flow of a time-sharded stream -------------------------------------------------------
    rng = np.random.default_rng(5)
    pcm = rng.standard_normal((frames, 2))
    m = RecordingMeter(rank)
    spec = FakeSpec(pcm[cuts[rank]:cuts[rank + 1]])
    red = sharded.analyze_stream_shard(dist, m, spec, rank, world, 2, first_hop=cuts[rank] // 4800)
    assert m.seen["first_hop"] == cuts[rank] // 4800
    if rank > 0:
        assert np.array_equal(m.seen["left_tail"][:, :2], pcm[cuts[rank] - 16:cuts[rank]])
        le = m.seen["left_edge"]
        assert le.n_hops == (cuts[rank] - cuts[rank - 1]) // 4800
        assert np.ctypeslib.as_array(le.rec)[5, 7, 8] == 1000.0 * (rank - 1) + 431
        assert np.array_equal(np.ctypeslib.as_array(le.tail)[:, :2], pcm[cuts[rank] - 16:cuts[rank]])
    else:
        assert m.seen["left_tail"] is None and m.seen["left_edge"] is None
    if rank + 1 < world:
        assert np.array_equal(m.seen["right_head"][:, :2], pcm[cuts[rank + 1]:cuts[rank + 1] + 16])
    else:
        assert m.seen["right_head"] is None
    # the all-reduce: counts and the power sum add, peaks take the maximum; identical on every rank
    for r in range(world):
        assert red.hist[r] == 1 + r
    assert red.hist[8000] == 7 * world and red.hist[world] == 0
    assert red.sum_kept == sum(0.5 + r for r in range(world))
    assert red.n_kept == sum(10 + r for r in range(world)) and red.n_blocks == sum(100 + r for r in range(world))
    assert red.peak_per_ch[0] == 0.25 * world and red.peak_per_ch[1] == 1.0 and red.peak_per_ch[2] == 0.0
    assert all(red.st_hist[100 + r] == 3 for r in range(world)) and red.st_hist[100 + world] == 0
    assert red.st_sum_kept == sum(0.25 * (r + 1) for r in range(world)) and red.st_n_kept == sum(2 + r for r in range(world))
    assert red.n_shortterm == sum(50 + r for r in range(world))
    assert red.max_m_power == float(world) and red.max_s_power == 5.0 and red.range_ok == 0   # rank 1 vetoed

    assert red.sum_gated == 0.0 and red.n_gated == 0 and "reduced_for_pass3" not in m.seen   # histogram modes: one reduction

    # --- BS.1770's exact gating: a second pass sees the reduced sums, its two numbers are reduced once more ---------
    mx = RecordingMeter(rank, hr.GATE_EXACT)
    redx = sharded.analyze_stream_shard(dist, mx, spec, rank, world, 2, first_hop=cuts[rank] // 4800)
    assert mx.seen["reduced_for_pass3"] == (sum(0.5 + r for r in range(world)), sum(10 + r for r in range(world)))
    assert redx.sum_gated == sum(0.125 * (r + 1) for r in range(world)) and redx.n_gated == sum(3 + r for r in range(world))

    # --- the same flow with the records in buffers of the caller (the device-resident protocol, in host memory here): the
    # edge reaches the right neighbour byte for byte, every rank ends up with every rank's partial in rank order, an idle
    # rank contributes the neutral image, and a pending redo on ONE rank sends every rank round once more
    bufs = sharded.ShardBuffers(world, torch.device("cpu"))
    md = RecordingMeter(rank)
    halos = sharded.shard_halos(dist, md, spec, rank, world)
    parts = sharded.analyze_stream_shard_device(dist, md, spec, rank, world, 2, first_hop=cuts[rank] // 4800, halos=halos, bufs=bufs)
    assert len(parts) == world and [p.n_blocks for p in parts] == [100 + r for r in range(world)]
    assert [p.hist[r] for r, p in enumerate(parts)] == [1 + r for r in range(world)] and md.seen["finish_calls"] == 1
    if rank > 0:
        le = md.seen["left_edge"]
        assert le.n_hops == (cuts[rank] - cuts[rank - 1]) // 4800 and np.ctypeslib.as_array(le.rec)[5, 7, 8] == 1000.0 * (rank - 1) + 431
        assert np.array_equal(md.seen["left_tail"][:, :2], pcm[cuts[rank] - 16:cuts[rank]])
    else:
        assert md.seen["left_edge"] is None and md.seen["left_tail"] is None

    class Redo(RecordingMeter):
        def shard_pass2(self, left_edge):
            p = super().shard_pass2(left_edge)
            p.reserved = 1 if self.rank == world - 1 else 0
            return p
    mr = Redo(rank)
    parts = sharded.analyze_stream_shard_device(dist, mr, spec, rank, world, 2, halos=halos, bufs=bufs)
    assert mr.seen["finish_calls"] == 2 and mr.seen.get("redo") and all(p.reserved == 0 for p in parts)
    mi = RecordingMeter(rank)
    parts = sharded.analyze_stream_shard_device(dist, mi, FakeSpec(pcm[:4800 * 7]) if rank == 0 else None, rank, 1, 2, halos=(None, None), bufs=bufs)
    assert parts[0].n_blocks == 100 and all(p.n_blocks == 0 and p.range_ok == 1 for p in parts[1:]) and (rank == 0) == ("left_edge" in mi.seen)

    # --- a rank with an empty shard sits the passes out but joins the reduction --------------------
    m2 = RecordingMeter(rank)
    red2 = sharded.analyze_stream_shard(dist, m2, FakeSpec(pcm[:4800 * 7]) if rank == 0 else None, rank, 1, 2)
    assert red2.n_blocks == 100 and red2.hist[0] == 1 and (rank == 0) == bool(m2.seen)


def mode_gpu(rank, world):
    import signals
    import oracle_lib as oracle
    torch.cuda.set_device(0)
    if rank == 0:
        oracle.build()
    dist.barrier()
    fs = 48000
    pcm = signals.seeded_noise(91, 4800 * 83 + 777, 2, -19.0)
    pcm[: fs * 2] *= 0.003
    with hr.Meter(device=0) as m:
        # one long stream, cut in time
        cuts = sharded.cut_stream(pcm.shape[0], fs, world)
        spec = hr.TrackSpec(pcm[cuts[rank]:cuts[rank + 1]].copy(), fs)
        got = sharded.analyze_stream_shard(dist, m, spec, rank, world, 2, first_hop=cuts[rank] // 4800)
        whole = m.analyze(pcm, fs)
        ref = oracle.measure(pcm, fs)
        assert got["n_blocks"] == whole["n_blocks"] == ref["n_blocks"], (got, whole)
        assert got["n_gated"] == whole["n_gated"]
        assert abs(got["integrated_lufs"] - whole["integrated_lufs"]) < 1e-9
        assert abs(got["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6
        assert got["peak_per_ch"] == whole["peak_per_ch"] == ref["peak_per_ch"]
        for k in ("loudness_range_lu", "lra_low_lufs", "lra_high_lufs", "max_momentary_lufs", "max_shortterm_lufs"):
            assert abs(got[k] - whole[k]) < 1e-9, (k, got, whole)   # 83 hops over the ranks: every shard has its 32
        # BS.1770's exact gating: two reductions (hr_shard_pass3 between them)
        with hr.Meter(device=0, gate_mode=hr.GATE_EXACT) as mx:
            got_x = sharded.analyze_stream_shard(dist, mx, spec, rank, world, 2, first_hop=cuts[rank] // 4800)
            whole_x = mx.analyze(pcm, fs)
            ref_x = oracle.measure(pcm, fs, gate_mode=1)
            assert got_x["n_gated"] == whole_x["n_gated"] == ref_x["n_gated"], (got_x, whole_x)
            assert abs(got_x["integrated_lufs"] - whole_x["integrated_lufs"]) < 1e-9
            assert abs(got_x["integrated_lufs"] - ref_x["integrated_lufs"]) < 1e-6
            assert got_x["peak_per_ch"] == ref_x["peak_per_ch"]
        # the exchange with the records resident on the device (hr_shard_pass1_device ...): one edge send / receive, one
        # all-gather, staged through the host here because the group is gloo; twice on the same buffers, then a stream of
        # tones, whose candidate lists overflow: the finish answers SHARD_AGAIN and every rank runs its exact redo
        bufs = sharded.ShardBuffers(world, torch.device("cuda", 0))
        halos = sharded.shard_halos(dist, m, spec, rank, world)
        for _ in range(2):
            phases = {}
            got_v = sharded.analyze_stream_shard_device(dist, m, spec, rank, world, 2, first_hop=cuts[rank] // 4800, halos=halos, bufs=bufs,
                                                        timings=phases)
            assert all(got_v[k] == got[k] or (got_v[k] != got_v[k] and got[k] != got[k]) for k in got), (got_v, got)
            assert "halo_exchange" not in phases and "redo" not in phases
        t = np.arange(pcm.shape[0]) / fs
        tones = np.stack([0.999 * np.sin(2 * np.pi * 997.0 * t), 0.5 * np.sin(2 * np.pi * 3001.0 * t + 1.0)], axis=1).astype(np.float32)
        tspec = hr.TrackSpec(tones[cuts[rank]:cuts[rank + 1]].copy(), fs)
        phases = {}
        got_t = sharded.analyze_stream_shard_device(dist, m, tspec, rank, world, 2, first_hop=cuts[rank] // 4800, bufs=bufs, timings=phases)
        ref_t = oracle.measure(tones, fs)
        assert "redo" in phases and "halo_exchange" in phases
        assert got_t["peak_per_ch"] == ref_t["peak_per_ch"], (got_t, ref_t)
        assert got_t["n_gated"] == ref_t["n_gated"] and abs(got_t["integrated_lufs"] - ref_t["integrated_lufs"]) < 1e-6
        # a stream too short to give every rank a shard: the idle ranks contribute the neutral image
        short = pcm[: 4800 * 7 + 33]
        cs = sharded.cut_stream(short.shape[0], fs, world)
        act = sharded.active_shards(cs)
        sspec = hr.TrackSpec(short[cs[rank]:cs[rank + 1]].copy(), fs) if cs[rank + 1] > cs[rank] else None
        got_s = sharded.analyze_stream_shard_device(dist, m, sspec, rank, act, 2, first_hop=cs[rank] // 4800, bufs=bufs)
        ref_s = oracle.measure(short, fs)
        assert act == 1 and got_s["peak_per_ch"] == ref_s["peak_per_ch"] and got_s["n_blocks"] == ref_s["n_blocks"]
        assert abs(got_s["integrated_lufs"] - ref_s["integrated_lufs"]) < 1e-6
        # the same with the shard resident in HBM (device pointers, offset view for the tail)
        dev = torch.from_numpy(pcm[cuts[rank]:cuts[rank + 1]].copy()).cuda()
        got_d = sharded.analyze_stream_shard(dist, m, hr.TrackSpec(dev, fs), rank, world, 2, first_hop=cuts[rank] // 4800)
        # (the time-shard path reports the range statistics as NaN, and NaN != NaN)
        assert all(got_d[k] == got[k] or (got_d[k] != got_d[k] and got[k] != got[k]) for k in got), (got_d, got)
        # a batch of independent tracks, split over the ranks
        tracks = [signals.seeded_noise(200 + i, fs * 3 + 100 * i, 2, -12.0 - i) for i in range(5)]
        lo, hi = sharded.partition_tracks(len(tracks), world, rank)
        mine = m.analyze_batch([hr.TrackSpec(t, fs) for t in tracks[lo:hi]])
        allr = sharded.gather_results(dist, mine, len(tracks), world)
        single = m.analyze_batch([hr.TrackSpec(t, fs) for t in tracks])
        assert allr == single


def main():
    mode = sys.argv[1]
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    try:
        {"cpu": mode_cpu, "gpu": mode_gpu}[mode](rank, world)
        dist.barrier()
        if rank == 0:
            print(f"dist_worker {mode}: ok (world {world})")
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
