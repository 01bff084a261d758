"""Multi-GPU drivers over the C ABI: one process per GPU, `torch.distributed` for the exchange steps.

Two ways the path shards:

* track batches -- independent units, no data-path collective: `partition_tracks` gives every rank its
  slice, `gather_results` puts the per-rank results back in input order (what analyze_files' par_iter
  does across threads, src/cli.rs:403-435);
* one long stream cut in time -- `cut_stream` picks hop-aligned cuts, `analyze_stream_shard` runs
  hr_shard_head / pass1 / pass2 / finish with a neighbour exchange of `hr_shard_edge` and an all-reduce
  of `hr_shard_partial` in between (records in host structs; serves every gating mode);
  `analyze_stream_shard_device` (round 3) is the same exchange with the records resident on the GPUs:
  one neighbour send / receive of the edge and ONE all-gather of the partial, both on device tensors, the
  halos handed in with the shard, one synchronisation at the end.

The backend is whatever the process group was created with: "nccl" (RCCL over xGMI) on the GPUs,
"gloo" in the CPU tests.  Payloads are a few KB; both steps are latency-bound.
"""
import ctypes as C

import numpy as np

from . import GATE_EXACT, SHARD_AGAIN, HrShardEdge, HrShardPartial, MAX_CH

MIN_SHARD_HOPS = 6  # hr_shard_pass1: a shard hands the records of its last 6 hops to the right


def partition_tracks(n_tracks, world, rank):
    """Contiguous, balanced slice [lo, hi) of the track list for `rank` (sizes differ by at most one)."""
    base, extra = divmod(n_tracks, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def partition_tracks_by_bytes(sizes, world):
    """Longest-first greedy assignment: track i of `sizes` bytes goes to the rank with the least
    bytes so far, so that ranks with unequal tracks finish together.  Returns a list of index lists, one per
    rank, each in input order."""
    load = [0] * world
    mine = [[] for _ in range(world)]
    for i in sorted(range(len(sizes)), key=lambda j: (-sizes[j], j)):
        r = min(range(world), key=lambda q: (load[q], q))
        load[r] += sizes[i]
        mine[r].append(i)
    return [sorted(m) for m in mine]


def cut_stream(frames, sample_rate, world):
    """Frame offsets [c_0 = 0, ..., c_world = frames] of `world` time shards: every cut is a multiple of
    the hop, every shard but the last holds at least MIN_SHARD_HOPS whole hops.  Ranks that would get
    less are given empty shards (c_i == c_{i+1}) at the END of the list and sit the passes out."""
    hop = sample_rate // 10
    n_hops = frames // hop
    active = max(1, min(world, n_hops // MIN_SHARD_HOPS))
    cuts = [0]
    for r in range(1, active):
        cuts.append((n_hops * r // active) * hop)
    cuts.append(frames)
    cuts += [frames] * (world - active)
    return cuts


def active_shards(cuts):
    """Number of non-empty shards of a cut list from `cut_stream` (they come first)."""
    return sum(1 for a, b in zip(cuts[:-1], cuts[1:]) if b > a)


def _comm_device(dist):
    import torch
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def _as_bytes(struct):
    return np.frombuffer(memoryview(struct).cast("B"), dtype=np.uint8)


def exchange_edges(dist, edge, rank, active):
    """Neighbour exchange among ranks 0..active-1: every rank sends its hr_shard_edge to rank+1.
    Returns the left neighbour's edge (None on rank 0)."""
    import torch
    dev = _comm_device(dist)
    ops, left_buf = [], None
    if rank + 1 < active:
        ops.append(dist.P2POp(dist.isend, torch.from_numpy(_as_bytes(edge).copy()).to(dev), rank + 1))
    if rank > 0:
        left_buf = torch.empty(C.sizeof(HrShardEdge), dtype=torch.uint8, device=dev)
        ops.append(dist.P2POp(dist.irecv, left_buf, rank - 1))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return None if left_buf is None else HrShardEdge.from_buffer_copy(left_buf.cpu().numpy().tobytes())


def allreduce_partial(dist, partial):
    """All-reduce of hr_shard_partial: histogram counts, block / window counts and the kept-power sums add, the peaks and
    the maximum powers take the maximum, range_ok the minimum."""
    import torch
    dev = _comm_device(dist)
    counts = np.empty(2 * 8001 + 4, dtype=np.int64)
    counts[:8001] = np.ctypeslib.as_array(partial.hist).astype(np.int64)
    counts[8001:16002] = np.ctypeslib.as_array(partial.st_hist).astype(np.int64)
    counts[16002:] = partial.n_kept, partial.n_blocks, partial.st_n_kept, partial.n_shortterm
    tc = torch.from_numpy(counts).to(dev)
    ts = torch.tensor([partial.sum_kept, partial.st_sum_kept], dtype=torch.float64, device=dev)
    mx = np.empty(MAX_CH + 3, dtype=np.float64)
    mx[:MAX_CH] = np.ctypeslib.as_array(partial.peak_per_ch)
    mx[MAX_CH:] = partial.max_m_power, partial.max_s_power, -float(partial.range_ok)  # max of the negative = minimum
    tp = torch.from_numpy(mx).to(dev)
    dist.all_reduce(tc, op=dist.ReduceOp.SUM)
    dist.all_reduce(ts, op=dist.ReduceOp.SUM)
    dist.all_reduce(tp, op=dist.ReduceOp.MAX)
    out = HrShardPartial()
    c = tc.cpu().numpy()
    np.ctypeslib.as_array(out.hist)[:] = c[:8001].astype(np.uint64)
    np.ctypeslib.as_array(out.st_hist)[:] = c[8001:16002].astype(np.uint64)
    out.n_kept, out.n_blocks, out.st_n_kept, out.n_shortterm = (int(v) for v in c[16002:])
    sums = ts.cpu().numpy()
    out.sum_kept, out.st_sum_kept = float(sums[0]), float(sums[1])
    m = tp.cpu().numpy()
    np.ctypeslib.as_array(out.peak_per_ch)[:] = m[:MAX_CH]
    out.max_m_power, out.max_s_power = float(m[MAX_CH]), float(m[MAX_CH + 1])
    out.range_ok = int(round(-m[MAX_CH + 2]))
    return out


def analyze_stream_shard(dist, meter, shard_spec, rank, active, channels, first_hop=0, timings=None):
    """One rank's part of a time-sharded stream.  `shard_spec` is this rank's TrackSpec (None for a rank
    with an empty shard); first_hop = index of its first hop in the whole stream (cut // hop length).  Every rank of
    the group must call this; all get the same result dict.  `timings` (a dict) receives this rank's wall seconds per
    phase: halo exchange, pass 1, edge exchange, pass 2, all-reduce, finish."""
    import time
    t = [time.perf_counter()]

    def lap(name):
        t.append(time.perf_counter())
        if timings is not None:
            timings[name] = timings.get(name, 0.0) + (t[-1] - t[-2])

    part = HrShardPartial()
    part.range_ok = 1  # a rank that sits the passes out does not veto the range statistics
    if shard_spec is not None and rank < active:
        head = meter.shard_head(shard_spec)
        # pass 1 needs the neighbours' 16-frame halos: tails travel right, heads travel left
        left_tail, right_head = _exchange_halos(dist, meter, shard_spec, head, rank, active)
        lap("halo_exchange")
        edge = meter.shard_pass1(shard_spec, left_tail, right_head, first_hop)
        lap("pass1")
        left_edge = exchange_edges(dist, edge, rank, active)
        lap("edge_exchange")
        part = meter.shard_pass2(left_edge)
        lap("pass2")
    red = allreduce_partial(dist, part)
    lap("all_reduce")
    if meter.gate_mode == GATE_EXACT:
        # BS.1770's text has no histogram: the relative gate comes from the reduced sums, so the blocks above it are
        # counted in a second pass over each shard's block powers and reduced once more (two numbers)
        import torch
        p3 = meter.shard_pass3(red) if shard_spec is not None and rank < active else HrShardPartial()
        lap("pass3")
        dev = _comm_device(dist)
        ts = torch.tensor([p3.sum_gated], dtype=torch.float64, device=dev)
        tn = torch.tensor([p3.n_gated], dtype=torch.int64, device=dev)
        dist.all_reduce(ts, op=dist.ReduceOp.SUM)
        dist.all_reduce(tn, op=dist.ReduceOp.SUM)
        red.sum_gated, red.n_gated = float(ts.item()), int(tn.item())
        lap("all_reduce")
    out = meter.shard_finish(red, channels)
    lap("finish")
    return out


class ShardBuffers:
    """The device buffers of analyze_stream_shard_device, allocated once per rank: this shard's right edge, the left
    neighbour's edge, this shard's partial, and the gathered partials of all ranks (images of the C structs)."""

    def __init__(self, world, device):
        import torch
        self.edge = torch.zeros(C.sizeof(HrShardEdge), dtype=torch.uint8, device=device)
        self.left = torch.zeros(C.sizeof(HrShardEdge), dtype=torch.uint8, device=device)
        self.part = torch.zeros(C.sizeof(HrShardPartial), dtype=torch.uint8, device=device)
        self.gathered = torch.zeros(world * C.sizeof(HrShardPartial), dtype=torch.uint8, device=device)
        idle = HrShardPartial()
        idle.range_ok = 1  # a rank that sits the passes out does not veto the range statistics
        self.idle = torch.from_numpy(_as_bytes(idle).copy()).to(device)


def shard_halos(dist, meter, shard_spec, rank, active):
    """(left_tail, right_head) of this rank's shard: the 16 frames either side of it, as hr_shard_pass1 takes them.  Whoever
    cuts the stream can hand them over with the shard; ranks that hold only their own shard fetch them from the
    neighbours once, here (not part of the per-call exchange)."""
    if shard_spec is None or rank >= active:
        return None, None
    return _exchange_halos(dist, meter, shard_spec, meter.shard_head(shard_spec), rank, active)


def analyze_stream_shard_device(dist, meter, shard_spec, rank, active, channels, first_hop=0, halos=None, bufs=None,
                                shared_stream=False, timings=None, align_phases=False):
    """One rank's part of a time-sharded stream with the exchange records resident on the device (hr_shard_pass1_device /
    hr_shard_pass2_device / hr_shard_finish_gathered; every gating mode but GATE_EXACT).  Per call: one neighbour
    send / receive of the edge, one all-gather of the partial, one synchronisation (inside the finish).  Over "nccl" the
    tensors never leave the GPUs; over "gloo" (CPU tests, the one-device rehearsal) each of the two steps is staged through
    host memory.  `halos` = (left_tail, right_head) from shard_halos() or from whoever cut the stream; None: fetched here.
    shared_stream: the meter was created on torch's current stream (Meter(stream=torch.cuda.current_stream().cuda_stream)),
    so the collectives and the library's kernels are ordered on it and nothing waits on the host between the phases;
    otherwise the two are ordered through synchronisations.  align_phases (timing runs on a shared device): a barrier in front
    of each of the two exchange steps, so that `timings` shows what the exchange itself costs and not how long a rank waited
    for a neighbour whose kernels were queued behind its own.  Every rank of the group must call this."""
    import time

    import torch
    t = [time.perf_counter()]

    def lap(name):
        t.append(time.perf_counter())
        if timings is not None:
            timings[name] = timings.get(name, 0.0) + (t[-1] - t[-2])

    world = dist.get_world_size()
    on_gpu = dist.get_backend() == "nccl"
    if bufs is None:
        bufs = ShardBuffers(world, torch.device("cuda", torch.cuda.current_device()))
    mine = shard_spec is not None and rank < active

    def order():  # the library's stream and torch's: the same one, or brought together on the host
        if bufs.edge.is_cuda and not (shared_stream and on_gpu):
            torch.cuda.synchronize()

    if mine:
        if halos is None:
            halos = shard_halos(dist, meter, shard_spec, rank, active)
            lap("halo_exchange")
        meter.shard_pass1_device(shard_spec, halos[0], halos[1], first_hop, bufs.edge.data_ptr())
        order()  # (where the streams are brought together on the host, the kernels' time shows up in this phase)
        lap("pass1")
    if align_phases:
        dist.barrier()
        lap("wait_for_slowest_rank")
    if mine:
        ops = []
        if on_gpu:
            if rank + 1 < active:
                ops.append(dist.P2POp(dist.isend, bufs.edge, rank + 1))
            if rank > 0:
                ops.append(dist.P2POp(dist.irecv, bufs.left, rank - 1))
        else:
            host_left = torch.empty(C.sizeof(HrShardEdge), dtype=torch.uint8) if rank > 0 else None
            if rank + 1 < active:
                ops.append(dist.P2POp(dist.isend, bufs.edge.cpu(), rank + 1))
            if rank > 0:
                ops.append(dist.P2POp(dist.irecv, host_left, rank - 1))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        if not on_gpu and rank > 0:
            bufs.left.copy_(host_left)
        order()
        lap("edge_exchange")
        meter.shard_pass2_device(bufs.left.data_ptr() if rank > 0 else None, bufs.part.data_ptr())
        order()
        lap("pass2")
    else:
        bufs.part.copy_(bufs.idle)
    if align_phases:
        dist.barrier()
        lap("wait_for_slowest_rank")
    while True:
        if on_gpu:
            dist.all_gather_into_tensor(bufs.gathered, bufs.part)
        else:
            host = torch.empty(world * C.sizeof(HrShardPartial), dtype=torch.uint8)
            dist.all_gather_into_tensor(host, bufs.part.cpu())
            bufs.gathered.copy_(host)
        order()
        lap("all_gather")
        rc, out = meter.shard_finish_gathered(bufs.gathered.data_ptr(), world, channels)
        lap("finish")
        if rc != SHARD_AGAIN:
            return out
        if mine:  # some shard's exact true-peak redo was pending: every rank patches its image, then once more
            meter.shard_redo_device(bufs.part.data_ptr())
        order()
        lap("redo")


def _exchange_halos(dist, meter, shard_spec, head, rank, active):
    """left_tail (the 16 frames before the shard) and right_head (the 16 after) for pass 1."""
    import torch
    dev = _comm_device(dist)
    tail = meter.shard_tail(shard_spec)
    ops, lt, rh = [], None, None
    if rank + 1 < active:
        ops.append(dist.P2POp(dist.isend, torch.from_numpy(tail.reshape(-1).copy()).to(dev), rank + 1))
        rh = torch.empty(16 * MAX_CH, dtype=torch.float64, device=dev)
        ops.append(dist.P2POp(dist.irecv, rh, rank + 1))
    if rank > 0:
        ops.append(dist.P2POp(dist.isend, torch.from_numpy(head.reshape(-1).copy()).to(dev), rank - 1))
        lt = torch.empty(16 * MAX_CH, dtype=torch.float64, device=dev)
        ops.append(dist.P2POp(dist.irecv, lt, rank - 1))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    lt = None if lt is None else lt.cpu().numpy().reshape(16, MAX_CH).copy()
    rh = None if rh is None else rh.cpu().numpy().reshape(16, MAX_CH).copy()
    return lt, rh


def gather_results(dist, local_results, n_tracks, world):
    """Per-rank result lists -> one list in input order on every rank (tiny: a few doubles per track)."""
    gathered = [None] * world
    dist.all_gather_object(gathered, local_results)
    out = []
    for r in range(world):
        lo, hi = partition_tracks(n_tracks, world, r)
        assert len(gathered[r]) == hi - lo
        out.extend(gathered[r])
    return out
