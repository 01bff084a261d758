#!/usr/bin/env python3
"""bench.py -- throughput of the loudness + true-peak hot path on synthetic PCM resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): a 1,000-track batch of
3-minute 44.1 kHz stereo f32 PCM per GPU (63.5 GB, generated in HBM by hr_synth_fill).  One step = one
hr_analyze_batch over the whole batch: integrated LUFS (ebur128 gating) + dBTP (192 kHz interpolation)
for every track.  Tracks are independent, so with N GPUs every rank analyses its own 1,000 tracks and no
data-path collective exists ("scaling": "weak").  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

FS = 44100
CH = 2
SECONDS = 180
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def cpu_baseline(n_tracks, threads):
    """The oracle (a scalar f64 restatement of FFmpeg's algorithm: "port") timed on the host cores on a
    bounded sample of the same workload: n_tracks full 3-minute tracks, one thread per track (mirrors one
    ffmpeg process per rayon worker, src/cli.rs:407)."""
    import subprocess
    from concurrent.futures import ThreadPoolExecutor

    import oracle_lib
    oracle_lib.build()
    # the timed leg uses a build of the same source for THIS machine's cores (-O3 -march=native)
    build = "-O2 -mfma (the portable build: gcc or make missing on this machine)"
    try:
        subprocess.check_call(["make", "-C", oracle_lib.ORACLE_DIR, "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        native = os.path.join(oracle_lib.ORACLE_DIR, "_native", "libhr_oracle_native.so")
        if os.path.exists(native) and oracle_lib._lib is None:
            oracle_lib.LIB_PATH = native
            build = "-O3 -march=native"
    except Exception:
        pass
    frames = FS * SECONDS
    tracks = [oracle_lib.synth_track(1000 + i, FS, CH, frames) for i in range(n_tracks)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        res = list(ex.map(lambda t: oracle_lib.measure(t, FS), tracks))
    dt = time.perf_counter() - t0
    hours = n_tracks * SECONDS / 3600.0
    return {"value": hours / dt, "unit": "audio-hours/s", "cores": threads, "kind": "port",
            "host_cores_visible": os.cpu_count(), "cores_note": "cores = threads actually used, one per core of this job's CPU share "
            "(16 per GPU on the box; os.cpu_count() shows the whole machine)", "build": build,
            "sample": f"{n_tracks} x {SECONDS} s 44.1 kHz stereo f32 tracks of the same synthetic generator, "
                      f"one thread per track, oracle mode ebur128 gating + 192 kHz interpolation, {dt:.2f} s wall"}, res, tracks


PROFILES = os.path.join(ROOT, "profiles")


def valu_ceiling(bytes_per_step, kernel_ms):
    """SURVEY 8(d) "min(HBM, VALU)": what the vector ALU lets the fused kernel reach, beside the HBM fraction.  Everything
    here except `frac_of_valu_ceiling` is read from a committed PMC profile of THIS workload (profiles/valu_profile.json:
    SQ_INSTS_VALU per sample, the clock under the kernel from GRBM_GUI_ACTIVE), not measured in this run; the caller emits
    the block only for the configuration that profile was taken on.  A full-rate VALU instruction occupies a SIMD for 4 cycles
    per 64-lane wave (profiles/r1_microbench2.txt: 2.2-2.6 ns per f64 FMA)."""
    path = os.path.join(PROFILES, "valu_profile.json")
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        prof = json.load(fh)
    simds, lanes_per_cycle = 256 * 4, 16
    f64_ops, valu_slots = prof["f64_ops_per_sample"], prof["valu_issue_slots_per_sample"]
    clk_obs, clk_max = prof["clock_observed_ghz"] * 1e9, 2.4e9

    def gbs(ops, clk):
        return simds * lanes_per_cycle * clk / ops * 4.0 / 1e9
    return {"f64_ops_per_sample": f64_ops, "valu_issue_slots_per_sample": valu_slots, "cycles_per_wave_instruction": 4,
            "clock_observed_ghz": clk_obs / 1e9, "clock_max_ghz": clk_max / 1e9,
            "ceiling_gbs_at_observed_clock": gbs(valu_slots, clk_obs), "ceiling_gbs_at_max_clock": gbs(valu_slots, clk_max),
            "f64_only_ceiling_gbs_at_observed_clock": gbs(f64_ops, clk_obs),
            "frac_of_valu_ceiling": (bytes_per_step / (kernel_ms * 1e-3) / 1e9) / gbs(valu_slots, clk_obs) if kernel_ms > 0 else None,
            "measured_in_this_run": ["frac_of_valu_ceiling"],
            "source": "from profile, not this run: profiles/valu_profile.json (" + prof["source"] + ")"}


def side_roofline(nbytes, kernel_ms, traffic_file=None, what="K-weight + true-peak pass (HIP events on the library stream, mean over timed steps)"):
    """The roofline object of a side figure: algorithmic bytes (every sample read once) over the fused kernel's own duration
    of THIS run; `traffic` from a committed PMC pass of the same workload where one exists (labelled), else null."""
    achieved = nbytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    traffic, src = None, None
    if traffic_file and os.path.exists(os.path.join(PROFILES, traffic_file)):
        with open(os.path.join(PROFILES, traffic_file)) as fh:
            t = json.load(fh)
        ratio = t["bytes_per_launch"] / t["algorithmic_bytes_per_launch"] if "bytes_per_launch" in t else t["bytes_per_track"] / t["algorithmic_bytes_per_track"]
        traffic = ratio * nbytes
        src = f"from profile, not this run: profiles/{traffic_file} (rocprofv3 --pmc TCC_EA0_RDREQ/WRREQ pass, gfx950 x2 read correction), scaled by bytes"
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": src, "algorithmic_bytes": nbytes, "kernel": what, "kernel_ms": kernel_ms}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tracks", type=int, default=0, help="tracks per GPU (default: 1000 = BASELINE config 2; 1250 for config3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--engine", type=int, default=0)
    ap.add_argument("--tp-mode", choices=["swr192k", "bs1770", "sample"], default="swr192k",
                    help="true-peak interpolator (swr192k = FFmpeg's, the BASELINE workload; the others are side figures)")
    ap.add_argument("--pcm-format", choices=["f32", "s16", "s24"], default="f32",
                    help="f32 is the BASELINE workload; s16 / s24 measure the integer stereo input of the fused kernel (side figures, labelled as such)")
    ap.add_argument("--workload", choices=["config2", "config3", "config4", "config5"], default="config2",
                    help="config2 = the BASELINE bench workload; config3 = one GPU's share of the 10,000-track 48 kHz batch "
                         "(1,250 tracks, 86.4 GB; a side figure); config4 = 96 kHz 24-bit 5.1 tracks (a side figure: split into "
                         "stereo f32 planes on the device, then the same fused kernel); config5 = one 10-hour 48 kHz stereo stream, "
                         "whole and as 8 time shards on this one device (a side figure)")
    ap.add_argument("--seconds", type=float, default=180.0,
                    help="track length of the config2 / config3 batch (180 = the BASELINE workload; e.g. --seconds 10 --tracks 10000 is the "
                         "short-clip side figure: per-track fixed costs -- the last hop and the ragged tail -- weigh 18 times more)")
    ap.add_argument("--ragged", type=int, default=0, help="extra frames per track beyond a whole number of hops (side figures)")
    ap.add_argument("--hours", type=float, default=10.0, help="config5: length of the stream")
    ap.add_argument("--shard-protocol", choices=["device", "host"], default="device",
                    help="config5 with --gpus N > 1: the exchange records stay on the GPUs (one edge send / receive, one all-gather, one "
                         "synchronisation: hr_shard_*_device) or travel as host structs (rounds 1-2: halo exchange, edge exchange, three "
                         "all-reduces)")
    ap.add_argument("--total-tracks", type=int, default=0,
                    help="config3 with --gpus N > 1 (or this flag): the whole batch is this many tracks (default 10000 = BASELINE "
                         "configs[2]) split over the ranks by bytes, strong scaling")
    ap.add_argument("--first-pass", type=int, default=0, metavar="N",
                    help="also time hr_loudnorm_first_pass (the literal loudnorm first pass at 192 kHz, src/analyzer.rs:261) on the first N "
                         "tracks; reported beside, never as value")
    ap.add_argument("--host-pcm", type=int, default=0, metavar="N",
                    help="also time N tracks handed over as pinned HOST buffers (PCIe-inclusive; reported beside, never as value)")
    args = ap.parse_args()

    import torch

    import headroom_amd as hr

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # HR_BENCH_ONE_DEVICE=1 (rehearsal of the multi-rank path on a one-GPU box): every rank uses cuda:0 and the
    # timing collectives run over gloo
    rehearsal = os.environ.get("HR_BENCH_ONE_DEVICE") == "1"
    if rehearsal:
        local = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    global FS, SECONDS
    if args.workload == "config3":
        FS = 48000
    SECONDS = args.seconds if args.seconds != int(args.seconds) else int(args.seconds)
    n_tracks = args.tracks or (1250 if args.workload == "config3" else 1000)
    meter = hr.Meter(device=local, engine=args.engine, tp_mode={"swr192k": 0, "bs1770": 1, "sample": 2}[args.tp_mode])
    if args.workload == "config4":
        run_config4(args, meter, dev, n_tracks, rank)
        meter.close()
        return
    if args.workload == "config5":
        if world > 1 and args.shard_protocol == "device":
            # the meter works on a torch stream, so that RCCL's collectives and the library's kernels are ordered on it
            meter.close()
            comm_stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(comm_stream):
                meter = hr.Meter(device=local, engine=args.engine, stream=comm_stream.cuda_stream)
                run_config5(args, meter, dev, local, rank, world, rehearsal)
                torch.cuda.synchronize()
        else:
            run_config5(args, meter, dev, local, rank, world, rehearsal)
        meter.close()
        if world > 1:
            dist.destroy_process_group()
        return
    if args.workload == "config3" and (world > 1 or args.total_tracks):
        run_config3_strong(args, meter, dev, rank, world, rehearsal)
        meter.close()
        if world > 1:
            dist.destroy_process_group()
        return
    frames = int(FS * SECONDS) + args.ragged
    # HR_BENCH_OFFSET (experiments): byte offset of every track inside its slot, to see what cache-line alignment is worth
    boff = int(os.environ.get("HR_BENCH_OFFSET", "0"))
    # (every track starts on a 16-byte boundary, as separately allocated buffers would: the stride is an even frame count)
    pcm = torch.empty((n_tracks, (frames + 1) // 2 * 2 + (16 if boff else 0), CH), dtype=torch.float32, device=dev)
    for i in range(n_tracks):
        meter.synth_fill(pcm[i].data_ptr() + boff, 1000 + rank * n_tracks + i, FS, CH, frames)
    torch.cuda.synchronize()
    if args.pcm_format == "s16":
        pcm16 = torch.empty((n_tracks, frames, CH), dtype=torch.int16, device=dev)
        for i in range(n_tracks):
            pcm16[i] = torch.clamp(torch.round(pcm[i][:frames] * 32768.0), -32768, 32767).to(torch.int16)
        torch.cuda.synchronize()
        specs = [hr.TrackSpec(pcm16[i].data_ptr(), FS, channels=CH, fmt=hr.FMT_S16, frames=frames) for i in range(n_tracks)]
        bytes_per_step = n_tracks * frames * CH * 2
    elif args.pcm_format == "s24":
        pcm24 = torch.empty((n_tracks, frames * CH * 3), dtype=torch.uint8, device=dev)
        for i in range(n_tracks):
            q = torch.clamp(torch.round(pcm[i][:frames] * 8388608.0), -8388608, 8388607).to(torch.int32).reshape(-1)
            pcm24[i] = torch.stack([q & 0xFF, (q >> 8) & 0xFF, (q >> 16) & 0xFF], dim=1).to(torch.uint8).reshape(-1)
        torch.cuda.synchronize()
        specs = [hr.TrackSpec(pcm24[i].data_ptr(), FS, channels=CH, fmt=hr.FMT_S24, frames=frames) for i in range(n_tracks)]
        bytes_per_step = n_tracks * frames * CH * 3
    else:
        specs = [hr.TrackSpec(pcm[i].data_ptr() + boff, FS, channels=CH, fmt=hr.FMT_F32, frames=frames) for i in range(n_tracks)]
        bytes_per_step = n_tracks * frames * CH * 4

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the descriptor / result arrays are built once, as a C caller's would be: a step is exactly one hr_analyze_batch
    batch = hr.TrackBatch(specs)
    for _ in range(args.warmup):
        meter.analyze_prepared(batch)
    main_ms = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        meter.analyze_prepared(batch)
        main_ms.append(meter.last_timing()[1])
    barrier()
    elapsed = time.perf_counter() - t0
    res = batch.results()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert all(r["status"] == 0 for r in res)

    if rank == 0:
        hours_per_step = world * n_tracks * SECONDS / 3600.0
        value = hours_per_step * args.steps / elapsed
        kernel_ms = sum(main_ms) / max(len(main_ms), 1)
        achieved = bytes_per_step / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        # HBM bytes per launch of the fused kernel: PMC counters need a profiler pass of their own, so this is the
        # committed measurement (profiles/r3_fused_hbm_traffic.json, bytes per track) scaled to this launch
        traffic, traffic_src = None, None
        tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r3_fused_hbm_traffic.json")
        if (os.path.exists(tpath) and args.workload == "config2" and args.pcm_format == "f32" and SECONDS == 180 and not args.ragged
                and args.tp_mode == "swr192k" and args.engine == 0):
            with open(tpath) as fh:
                traffic = json.load(fh)["bytes_per_track"] * n_tracks
            traffic_src = "profiles/r3_fused_hbm_traffic.json (rocprofv3 --pmc TCC_EA0_RDREQ/WRREQ pass, gfx950 x2 read correction), per track x tracks in the launch"
        line = {
            "metric": "audio-hours analyzed/sec (LUFS+dBTP)", "value": value, "unit": "audio-hours/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{n_tracks}-track batch per GPU of {SECONDS:g} s {FS / 1000:g} kHz stereo {args.pcm_format} PCM resident in HBM "
                                   f"(BASELINE configs[{2 if args.workload == 'config3' else 1}]); per-track integrated LUFS (ebur128 gating) + dBTP ({'192 kHz' if args.tp_mode == 'swr192k' else args.tp_mode})",
                       "tracks_per_gpu": n_tracks, "bytes_per_gpu": bytes_per_step, "parallelism": f"track-sharded x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes": bytes_per_step,
                         "kernel": "K-weight + true-peak pass (HIP events on the library stream, mean over timed steps)",
                         "kernel_ms": kernel_ms},
            "hbm_gbs_whole_step": bytes_per_step * args.steps / elapsed / 1e9,
        }
        # (only for the configuration the PMC profile behind it was taken on: 180 s 44.1 kHz stereo f32, libswresample's bank)
        if traffic is not None and args.tp_mode == "swr192k":
            valu = valu_ceiling(bytes_per_step, kernel_ms)
            if valu:
                line["roofline"]["valu"] = valu
                line["roofline"]["binding_ceiling"] = "valu" if valu["ceiling_gbs_at_observed_clock"] < HBM_PEAK_GBS else "hbm"
        if (world == 1 and not args.no_cpu_baseline and args.pcm_format == "f32" and args.workload == "config2" and args.tp_mode == "swr192k"
                and SECONDS == 180 and not args.ragged):
            # one thread per core of the job's CPU share: the scheduler's affinity mask when it is narrower than the
            # machine, else the box's documented share of 16 cores per GPU
            try:
                share = len(os.sched_getaffinity(0))
            except AttributeError:
                share = os.cpu_count() or 1
            threads = max(1, min(16, share))
            base, ref, cpu_tracks = cpu_baseline(threads, threads)
            line["cpu_baseline"] = base
            # the same tracks (seeds 1000..) are the first tracks of rank 0's batch: parity spot check
            worst_i = max(abs(res[i]["integrated_lufs"] - ref[i]["integrated_lufs"]) for i in range(len(ref)))
            same_tp = all(res[i]["peak_per_ch"] == ref[i]["peak_per_ch"] for i in range(len(ref)))
            line["parity_check"] = {"tracks": len(ref), "max_abs_lufs_diff": worst_i, "true_peak_bit_identical": same_tp}
        if args.first_pass > 0:
            nf = min(args.first_pass, n_tracks)
            fres = meter.loudnorm_first_pass(specs[:nf])
            torch.cuda.synchronize()
            best = None
            for _ in range(3):
                t1 = time.perf_counter()
                fres = meter.loudnorm_first_pass(specs[:nf])
                dt = time.perf_counter() - t1
                best = dt if best is None or dt < best else best
            assert all(r["status"] == 0 for r in fres)
            line["first_pass"] = {"tracks": nf, "audio_hours_per_s": nf * SECONDS / 3600.0 / best, "ms_per_call": best * 1e3,
                                  "max_abs_lufs_diff_to_default_mode": max(abs(fres[i]["integrated_lufs"] - res[i]["integrated_lufs"]) for i in range(nf)),
                                  "true_peak_equal_to_default_mode": all(abs(fres[i]["true_peak_dbtp"] - res[i]["true_peak_dbtp"]) < 0.05 for i in range(nf)),
                                  "note": "hr_loudnorm_first_pass: resample to 192 kHz (kept in registers), K-weighting at 192 kHz, libebur128 gating; ~10x the arithmetic of the default mode per input sample"}
        if args.host_pcm > 0:
            # the boundary also takes host buffers: same tracks from pinned host memory, H2D staging included
            nh = min(args.host_pcm, n_tracks)
            host = torch.empty((nh, frames, CH), dtype=torch.float32, pin_memory=True)
            host.copy_(pcm[:nh, :frames])
            torch.cuda.synchronize()
            hspecs = [hr.TrackSpec(host[i], FS) for i in range(nh)]
            hres = meter.analyze_batch(hspecs)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                hres = meter.analyze_batch(hspecs)
            dt = (time.perf_counter() - t1) / reps
            assert all(hres[i]["integrated_lufs"] == res[i]["integrated_lufs"] and hres[i]["peak_per_ch"] == res[i]["peak_per_ch"]
                       for i in range(nh))
            line["host_pcm"] = {"tracks": nh, "audio_hours_per_s": nh * SECONDS / 3600.0 / dt,
                                "gb_per_s": nh * frames * CH * 4 / dt / 1e9, "note": "pinned host PCM, staging copies included"}
        print(json.dumps(line), flush=True)
    meter.close()
    if world > 1:
        dist.destroy_process_group()


def run_config4(args, meter, dev, n_tracks, rank):
    """BASELINE configs[3]: 180 s 96 kHz 5.1 tracks, 24-bit packed (311 MB each), weights {1,1,1,0,1.41,1.41}."""
    import torch

    import headroom_amd as hr
    fs, ch = 96000, 6
    frames = fs * SECONDS
    raw = torch.empty((n_tracks, frames * ch * 3), dtype=torch.uint8, device=dev)
    tmp = torch.empty((frames, ch), dtype=torch.float32, device=dev)
    for i in range(n_tracks):
        meter.synth_fill(tmp.data_ptr(), 4000 + rank * n_tracks + i, fs, ch, frames)
        q = torch.clamp(torch.round(tmp * 8388608.0), -8388608, 8388607).to(torch.int32).reshape(-1)
        raw[i] = torch.stack((q & 0xff, (q >> 8) & 0xff, (q >> 16) & 0xff), dim=1).to(torch.uint8).reshape(-1)
    del tmp
    torch.cuda.synchronize()
    specs = [hr.TrackSpec(raw[i].data_ptr(), fs, channels=ch, fmt=hr.FMT_S24, frames=frames) for i in range(n_tracks)]
    batch = hr.TrackBatch(specs)
    for _ in range(args.warmup):
        meter.analyze_prepared(batch)
    torch.cuda.synchronize()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        meter.analyze_prepared(batch)
        kernel_ms.append(meter.last_timing()[1])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    res = batch.results()
    assert all(r["status"] == 0 for r in res)
    nbytes = n_tracks * frames * ch * 3
    direct = args.engine != 5
    print(json.dumps({"metric": "audio-hours analyzed/sec (LUFS+dBTP)", "value": n_tracks * SECONDS / 3600.0 / dt,
                      "unit": "audio-hours/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3,
                      "higher_is_better": True, "dtype": "f64", "data": "synthetic",
                      "config": {"workload": f"{n_tracks} x 180 s 96 kHz 5.1 s24 tracks resident in HBM (BASELINE configs[3]); side figure; "
                                             + ("18-byte frames read directly by the fused kernel" if direct else "engine 5: split into stereo planes first"),
                                 "bytes_per_gpu": nbytes},
                      "roofline": side_roofline(nbytes, sum(kernel_ms) / len(kernel_ms), "r3_config4_hbm_traffic.json" if direct else None,
                                                "K-weight + true-peak pass" + ("" if direct else " on the planes (the split kernels are not in this figure)")
                                                + " (HIP events on the library stream, mean over timed steps)"),
                      "hbm_gbs_whole_step": nbytes / dt / 1e9, "sample_result": res[0]}), flush=True)


def _dist_max(value, world, rehearsal, dev):
    """MAX over ranks of a host float (the timing all-reduce: RCCL on the GPUs, gloo in the one-device rehearsal)."""
    if world == 1:
        return value
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if rehearsal else dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _dist_barrier(world):
    import torch
    torch.cuda.synchronize()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def run_config5(args, meter, dev, local, rank=0, world=1, rehearsal=False):
    """BASELINE configs[4]: a single long 48 kHz stereo f32 stream.

    One rank: analysed whole (one track) and as 8 time shards through hr_analyze_stream_sharded (8 contexts on this
    device: carry hand-off, boundary blocks, halos and the histogram reduction are all exercised; only the GPUs are not
    distinct).  N ranks (torch.distributed.run): the stream is cut into N hop-aligned time shards, rank r holds shard r in
    its own HBM and runs headroom_amd.sharded.analyze_stream_shard -- pass 1, halo + edge exchange with the
    neighbours, pass 2, all-reduce of the gating partial, finish -- over RCCL ("nccl"); strong scaling (the stream's
    length is fixed), per-phase wall times reported as the maximum over ranks.  The shards are generated on the device,
    each from its own seed: the stream is their concatenation."""
    import torch

    import headroom_amd as hr
    from headroom_amd import sharded
    fs, ch = 48000, 2
    frames = int(args.hours * 3600) * fs
    if world > 1:
        import torch.distributed as dist
        cuts = sharded.cut_stream(frames, fs, world)
        active = sharded.active_shards(cuts)
        n = cuts[rank + 1] - cuts[rank]
        spec = None
        if n > 0:
            pcm = torch.empty((n, ch), dtype=torch.float32, device=dev)
            meter.synth_fill(pcm.data_ptr(), 77 + rank, fs, ch, n)
            spec = hr.TrackSpec(pcm.data_ptr(), fs, channels=ch, fmt=hr.FMT_F32, frames=n)
        torch.cuda.synchronize()
        phases = {}
        if args.shard_protocol == "device":
            # the 16 frames either side of a shard come with the shard (whoever cuts the stream has them): fetched once here
            bufs = sharded.ShardBuffers(world, dev)
            halos = sharded.shard_halos(dist, meter, spec, rank, active)

            def one_call(timings=None):
                return sharded.analyze_stream_shard_device(dist, meter, spec, rank, active, ch, first_hop=cuts[rank] // (fs // 10), halos=halos,
                                                           bufs=bufs, shared_stream=not rehearsal, timings=timings, align_phases=rehearsal)
        else:
            def one_call(timings=None):
                return sharded.analyze_stream_shard(dist, meter, spec, rank, active, ch, first_hop=cuts[rank] // (fs // 10), timings=timings)
        for _ in range(args.warmup):
            res = one_call()
        _dist_barrier(world)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = one_call(phases)
        _dist_barrier(world)
        elapsed = _dist_max(time.perf_counter() - t0, world, rehearsal, dev)
        phase_ms = {k: _dist_max(v, world, rehearsal, dev) / args.steps * 1e3 for k, v in sorted(phases.items())}
        if rank == 0:
            dt = elapsed / args.steps
            print(json.dumps({
                "metric": "audio-hours analyzed/sec (LUFS+dBTP)", "value": args.hours / dt, "unit": "audio-hours/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": f"one {args.hours:g} h 48 kHz stereo f32 stream time-sharded over {world} ranks, shards resident in "
                                       f"HBM (BASELINE configs[4]); side figure", "bytes_total": frames * ch * 4,
                           "parallelism": f"time-sharded x{world}" + (" (one-device rehearsal over gloo)" if rehearsal else " over RCCL"),
                           "exchange": "records resident on the device: one edge send/receive + one all-gather per call" if args.shard_protocol == "device"
                                       else "host records: halo exchange, edge exchange, three all-reduces per call"},
                "phase_ms_max_over_ranks": phase_ms, "hbm_gbs_whole_step": frames * ch * 4 / dt / 1e9, "result": res}), flush=True)
        return
    pcm = torch.empty((frames, ch), dtype=torch.float32, device=dev)
    meter.synth_fill(pcm.data_ptr(), 77, fs, ch, frames)
    torch.cuda.synchronize()
    spec = hr.TrackSpec(pcm.data_ptr(), fs, channels=ch, fmt=hr.FMT_F32, frames=frames)
    out = {}
    shards = [hr.Meter(device=local, engine=args.engine) for _ in range(8)]
    for name, fn in (("whole", lambda: meter.analyze_batch([spec])[0]), ("sharded8", lambda: hr.analyze_stream_sharded(shards, spec))):
        for _ in range(args.warmup):
            res = fn()
        torch.cuda.synchronize()
        kernel_ms = []
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = fn()
            kernel_ms.append(meter.last_timing()[1])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        out[name] = {"ms": dt * 1e3, "audio_hours_per_s": args.hours / dt, "gb_per_s": frames * ch * 4 / dt / 1e9, "result": res}
        if name == "whole":
            out[name]["roofline"] = side_roofline(frames * ch * 4, sum(kernel_ms) / len(kernel_ms))
    for m in shards:
        m.close()
    a, b = out["whole"]["result"], out["sharded8"]["result"]
    out["agree"] = {"n_blocks": a["n_blocks"] == b["n_blocks"], "lufs_diff": abs(a["integrated_lufs"] - b["integrated_lufs"]),
                    "peaks_identical": a["peak_per_ch"] == b["peak_per_ch"]}
    out["config"] = {"workload": f"one {args.hours:g} h 48 kHz stereo f32 stream resident in HBM (BASELINE configs[4]), 1 device; side figure"}
    print(json.dumps(out), flush=True)


def run_config3_strong(args, meter, dev, rank, world, rehearsal):
    """BASELINE configs[2]: ONE batch of 10,000 x 180 s 48 kHz stereo tracks split file-by-file over the ranks
    (headroom_amd.sharded.partition_tracks_by_bytes: longest first onto the least-loaded rank; no data-path collective,
    results gathered on the host), strong scaling.  A rank's share is analysed in launches of at most --tracks
    (default 1250 = 86.4 GB) tracks; when the share exceeds what is resident, the resident tracks are analysed again
    (the work is done every time; only the synthetic bytes repeat) -- on one GPU the 10,000 tracks are 8 launches."""
    import torch

    import headroom_amd as hr
    from headroom_amd import sharded
    fs, ch = 48000, 2
    frames = fs * SECONDS
    total = args.total_tracks or 10000
    per_launch = args.tracks or 1250
    mine = sharded.partition_tracks_by_bytes([frames * ch * 4] * total, world)[rank]
    n_res = min(len(mine), per_launch)
    if rehearsal:  # every rank on one device: keep the resident pool small
        n_res = min(n_res, max(1, per_launch // world))
    pcm = torch.empty((max(n_res, 1), frames, ch), dtype=torch.float32, device=dev)
    for i in range(n_res):
        meter.synth_fill(pcm[i].data_ptr(), 3000 + mine[i], fs, ch, frames)
    torch.cuda.synchronize()
    launches = []
    left = len(mine)
    while left > 0:
        k = min(left, n_res)
        launches.append(hr.TrackBatch([hr.TrackSpec(pcm[i].data_ptr(), fs, channels=ch, fmt=hr.FMT_F32, frames=frames) for i in range(k)]))
        left -= k
    for _ in range(args.warmup):
        for b in launches[:1]:
            meter.analyze_prepared(b)
    _dist_barrier(world)
    kernel_ms = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for b in launches:
            meter.analyze_prepared(b)
            kernel_ms += meter.last_timing()[1]
    _dist_barrier(world)
    elapsed = _dist_max(time.perf_counter() - t0, world, rehearsal, dev)
    assert all(st == 0 for b in launches for st in b.statuses())
    if world > 1:
        import torch.distributed as dist
        counts = [None] * world
        dist.all_gather_object(counts, len(mine))   # the host-side gather of analyze_files: here only the counts
        assert sum(counts) == total
    if rank == 0:
        dt = elapsed / args.steps
        print(json.dumps({
            "metric": "audio-hours analyzed/sec (LUFS+dBTP)", "value": total * SECONDS / 3600.0 / dt, "unit": "audio-hours/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{total}-track batch of 180 s 48 kHz stereo f32 PCM split per file over {world} ranks (BASELINE "
                                   f"configs[2]); side figure", "tracks_this_rank": len(mine), "launches_this_rank": len(launches),
                       "resident_tracks_this_rank": n_res, "bytes_total": total * frames * ch * 4,
                       "parallelism": f"track-sharded x{world}" + (" (one-device rehearsal over gloo)" if rehearsal else "")},
            "roofline": dict(side_roofline(len(mine) * frames * ch * 4, kernel_ms / args.steps, "r1_fused_hbm_traffic_48k.json"),
                             note="rank 0's share: its bytes over the sum of its launches' fused-kernel time"),
            "hbm_gbs_whole_step": total * frames * ch * 4 / dt / 1e9}), flush=True)


if __name__ == "__main__":
    main()
