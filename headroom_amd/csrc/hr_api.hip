// hr_api.hip -- the C ABI of include/headroom_hip.h: context, device tables, batch driver.
//
// Mirrors, on the reference side: analyze_files (src/cli.rs:403-435: order-preserving batch, per-track
// errors do not fail the batch) and the measurement half of analyze_file_with_target
// (src/analyzer.rs:253-290).  There is no CPU fallback: without a HIP device hr_init fails.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is opened at run time (RcclApi below), never linked

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <new>
#include <limits>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/headroom_hip.h"
#include "hr_device.hpp"
#include "hr_kernels.hpp"
#include "hr_tables.hpp"

// No exception crosses the C ABI: every int-returning entry point is a function-try-block (std::bad_alloc from a host
// container -> HR_ERR_OOM, anything else -> HR_ERR_INVALID).
#ifndef HR_CATCH_ALL
#define HR_CATCH_ALL catch (const std::bad_alloc &) { return HR_ERR_OOM; } catch (...) { return HR_ERR_INVALID; }
#endif

using namespace hr;

namespace {

// A fixed piece of device memory that workspace buffers are carved from instead of hipMalloc
// (hr_analyze_device_noalloc: the library's static scratch array).
struct Arena {
    char *base = nullptr;
    size_t cap = 0, used = 0;
};

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    Arena *arena = nullptr;  // non-null: carve from the arena, never allocate or free
    // grow-only; returns false on allocation failure
    bool ensure(size_t bytes)
    {
        if (bytes <= cap) return true;
        if (arena) {
            const size_t off = (arena->used + 255) / 256 * 256;
            if (off > arena->cap || bytes > arena->cap - off) return false;  // (the old piece, if any, is simply left behind)
            p = arena->base + off;
            cap = bytes;
            arena->used = off + bytes;
            return true;
        }
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 4 + 4096;
        if (hipMalloc(&p, want) != hipSuccess) {
            (void)hipGetLastError();
            want = bytes;
            if (hipMalloc(&p, want) != hipSuccess) { (void)hipGetLastError(); p = nullptr; return false; }
        }
        cap = want;
        return true;
    }
    void release() { if (p && !arena) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

constexpr size_t kMaxRatesPerCtx = 64;  // device tables are kept per distinct rate for the context's lifetime
constexpr uint32_t kPeakSpan = 8192;  // positions per workgroup of the generic true-peak kernels
constexpr uint32_t kTailSpan = 512;   // ... for the tail of a track whose body runs on the fused kernel

// frames of a track (frame bytes fb) after which a 16-byte chunk grid repeats: 16 / gcd(16, fb)
uint64_t chunk_frames(uint32_t fb)
{
    uint32_t g = 16;
    while (fb % g) g /= 2;
    return 16 / g;
}

size_t fmt_bytes(int fmt)
{
    switch (fmt) {
    case HR_FMT_F32: return 4;
    case HR_FMT_S16: return 2;
    case HR_FMT_S24: return 3;
    case HR_FMT_S32: return 4;
    case HR_FMT_F64: return 8;
    default: return 0;
    }
}

}  // namespace

namespace { struct Group; }

struct hr_ctx {
    hr_config cfg{};
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::mutex mu;

    // device tables
    std::vector<uint32_t> rate_fs;
    std::vector<RateDev> rates_h;
    std::vector<void *> rate_allocs;
    std::vector<std::pair<uint32_t, const double *>> swr_banks;  // libswresample banks for hr_loudnorm_first_pass, per rate
    DevBuf rates_d, ebu_energy, ebu_edges, ln_energy, ln_bounds, unit_tap;

    // workspace (grow-only)
    DevBuf stage_b;  // the second staging buffer (hr_analyze_batch alternates between `stage` and this one)
    DevBuf widened, resampled, parents_d, counters, gate_items, gate_parts, st_hist, tracks_d, items_d, comb_items_d, fitems_d, cands, cand_cnt, tpb, overflow, ovf_items, recs, hop_sums, blk_powers, hist, outs, peaks, stage, halos, fp_src_d, fp_items_d;

    // host copies of what is in flight on the stream (pageable sources must outlive the async copies)
    std::vector<WorkItem> h_items, h_comb_items;
    std::vector<FusedItem> h_fitems;
    // the fused launches of the last run_measure, per kernel class: items [item_off, item_off + n_items) of h_fitems, their
    // candidate lists [list_off, list_off + n_items * wpi) (wpi lists per item: 1, or channels / 2 for a multichannel track
    // read directly, list w of an item being channels 2 w and 2 w + 1)
    struct FusedRun { size_t item_off, n_items, list_off; uint32_t wpi; };
    std::vector<FusedRun> h_runs;
    std::vector<uint8_t> h_ftp;       // per track of the last run_measure: its true peak went through the fused kernel
    std::vector<uint64_t> h_fend;     // ... which screened the positions in front of this one (redo_overflowed_peaks)
    std::vector<unsigned> h_ovf, h_cand_cnt;  // redo_overflowed_peaks: the flags of the last launch, read back
    std::vector<WorkItem> h_ovf_items, h_gate_items;
    // hr_loudnorm_first_pass, while it measures a group of 192 kHz pseudo-tracks (empty otherwise): per descriptor, the
    // source track whose leading `ext_hops` hops k_first_pass_rows measures straight from the input PCM -- run_measure
    // launches it and gives the generic kernels only what is behind those hops
    std::vector<FpSrc> fp_srcs;
    std::vector<uint64_t> fp_ext_hops;
    std::vector<FpItem> h_fp_items;
    bool no_s16_direct = false;  // HR_NO_S16_DIRECT=1: stereo s16 is widened into f32 planes first, as before
    // the next three are set only by the experiments library (-DHR_EXPERIMENTS, make exp); always false / 0 in the shipped one
    bool serial_tail = false;  // HR_SERIAL_TAIL=1 (timing experiments): tail kernels before the fused kernel, not beside it
    bool split = false;  // HR_FUSED_SPLIT=1: k_fused_split (K-weighting and true peak in separate waves) instead of k_fused_f32s
    int ablate = 0;  // HR_ABLATE=1|2|3: timing-only builds of the fused kernel (results are then wrong)

    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // the bounds-checked generic kernels that finish the tracks of the fused kernel (last hop, ragged tail) are
    // latency-bound and tiny: they run beside the fused kernel on a second stream
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // staging copies of the next launch group run on a stream of their own, into the other of two staging buffers
    hipStream_t copy = nullptr;
    hipEvent_t ev_staged[2] = {nullptr, nullptr};
    float last_total_ms = 0, last_main_ms = 0;
    // HR_HOST_TIMING=1: where the host's time of an hr_analyze_batch call goes (stderr, one line per call)
    bool host_timing = false;
    std::chrono::steady_clock::time_point ht_last;
    std::vector<std::pair<const char *, double>> ht_marks;
    void ht(const char *what)
    {
        if (!host_timing) return;
        const auto now = std::chrono::steady_clock::now();
        ht_marks.emplace_back(what, std::chrono::duration<double, std::milli>(now - ht_last).count());
        ht_last = now;
    }

    size_t comb_n = 0;  // combine work items of the last run_measure (comb_items_d)

    // state of an open time shard (hr_shard_pass1 .. hr_shard_pass2)
    bool shard_open = false;
    bool shard_pass2_done = false;  // HR_GATE_EXACT keeps the shard open from pass 2 to pass 3
    Group *shard_group = nullptr;

    bool shard_device = false;      // the open shard went through the device-resident entry points (pass 2 / redo must match)
    std::vector<double> h_halo;     // pass 1's packed halos (the source of an asynchronous upload: must outlive the call)
    DevBuf shard_red;               // ShardReduced of hr_shard_finish_gathered
    DevBuf shard_edge_buf, shard_left_buf, shard_part_buf, shard_gather_buf;  // hr_analyze_stream_sharded: exchange buffers of this context
    bool no_alloc = false;  // inside hr_analyze_device_noalloc: workspace comes from the static scratch array, nothing is widened
    // every grow-only workspace buffer (not the device tables)
    std::vector<DevBuf *> workspace()
    {
        return {&widened, &resampled, &parents_d, &counters, &gate_items, &gate_parts, &st_hist, &tracks_d, &items_d, &comb_items_d, &fitems_d,
                &cands, &cand_cnt, &tpb, &overflow, &ovf_items, &recs, &hop_sums, &blk_powers, &hist, &outs, &peaks, &stage, &halos, &fp_src_d, &fp_items_d,
                &shard_red, &stage_b};
    }
};

// ---- workspace of hr_analyze_device_noalloc: one array per GPU that the library runs on, held until the process ends ----
// Host code never uses the array's name as an address: hr_init asks the runtime for the array's address on the current
// device (hipGetSymbolAddress) and keeps it per device; kernels, memsets and copies take that pointer.
constexpr size_t kScratchBytes = 16u << 20;
static __device__ __align__(256) unsigned char scratch[kScratchBytes];
constexpr int kMaxDevices = 64;
static void *g_scratch_ptr[kMaxDevices];      // per device; filled once by hr_init under g_scratch_init_mu
static std::mutex g_scratch_init_mu;
static std::mutex g_scratch_mu[kMaxDevices];  // calls on one GPU share that GPU's array: they are serialised here

#define HIPCHK(expr)                                                                      \
    do {                                                                                  \
        hipError_t e__ = (expr);                                                          \
        if (e__ != hipSuccess) {                                                          \
            std::fprintf(stderr, "headroom_hip: %s failed: %s\n", #expr, hipGetErrorString(e__)); \
            (void)hipGetLastError();                                                      \
            return e__ == hipErrorOutOfMemory ? HR_ERR_OOM : HR_ERR_HIP;                  \
        }                                                                                 \
    } while (0)

namespace {

int upload(DevBuf &b, const void *src, size_t bytes, hipStream_t st)
{
    if (!b.ensure(bytes)) return HR_ERR_OOM;
    HIPCHK(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, st));
    return HR_OK;
}

// device tables of one sample rate; returns the index in ctx->rates_h or a negative error
int rate_index(hr_ctx *ctx, uint32_t fs)
{
    for (size_t i = 0; i < ctx->rate_fs.size(); i++)
        if (ctx->rate_fs[i] == fs) return static_cast<int>(i);
    if (ctx->rate_fs.size() >= kMaxRatesPerCtx) {
        std::fprintf(stderr, "headroom_hip: more than %zu distinct sample rates on one context\n", kMaxRatesPerCtx);
        return HR_ERR_RATE;
    }
    const RateTables *rt = rate_tables(fs);
    if (!rt || rt->hop == 0) return HR_ERR_RATE;
    RateDev rd{};
    rd.k = rt->k;
    std::memcpy(rd.G, rt->G, sizeof rd.G);
    std::memcpy(rd.M1, rt->M[0], sizeof rd.M1);
    std::memcpy(rd.M2, rt->M[1], sizeof rd.M2);
    std::memcpy(rd.Gs, rt->sub_G, sizeof rd.Gs);
    std::memcpy(rd.Ms, rt->sub_M, sizeof rd.Ms);
    rd.sub_parts = static_cast<int32_t>(rt->sub_parts);
    rd.sub_terms = rt->sub_terms;
    rd.hop = rt->hop;
    rd.q_n0 = rt->q_n0;
    std::memcpy(rd.q_ab, rt->q_ab, sizeof rd.q_ab);
    void *g = nullptr;
    // (the fused kernel's 1 KiB slices read past the end, and the first slice of a row with a shifted piece grid starts up to 14
    // entries in front of the table: 512 bytes of zeros there)
    HIPCHK(hipMalloc(&g, rt->g.size() * sizeof(double) + 2048 + 512));
    ctx->rate_allocs.push_back(g);
    HIPCHK(hipMemset(g, 0, rt->g.size() * sizeof(double) + 2048 + 512));
    HIPCHK(hipMemcpy(static_cast<char *>(g) + 512, rt->g.data(), rt->g.size() * sizeof(double), hipMemcpyHostToDevice));
    rd.g = reinterpret_cast<const double *>(static_cast<const char *>(g) + 512);
    // the interpolation bank of the context's true-peak mode
    const bool bs = ctx->cfg.tp_mode == HR_TP_BS1770;
    const int phases = bs ? rt->bs_phases : rt->swr_phases;
    const std::vector<double> &taps = bs ? rt->bs_taps : rt->swr_taps;
    const std::vector<uint16_t> &atab = bs ? rt->bs_atab : rt->tp_atab;
    rd.swr_phases = phases;
    rd.swr_step = bs ? rt->bs_step : rt->swr_step;
    rd.swr_S = bs ? 1 : rt->swr_S;
    rd.swr_frac_scale = bs ? 1u : rt->swr_frac_scale;
    rd.swr_inv_src_incr = bs ? 1.0 : rt->swr_inv_src_incr;
    rd.swr_l1 = bs ? rt->bs_l1 : rt->swr_l1;
    rd.tp_passthru0 = (!bs && rt->swr_step == 1) ? 1 : 0;  // every input frame is a phase-0 output (integer ratio)
    if (phases > 0) {
        void *tp = nullptr;
        HIPCHK(hipMalloc(&tp, taps.size() * sizeof(double)));
        ctx->rate_allocs.push_back(tp);
        HIPCHK(hipMemcpy(tp, taps.data(), taps.size() * sizeof(double), hipMemcpyHostToDevice));
        rd.swr_taps = static_cast<const double *>(tp);
        if (!atab.empty()) {
            void *at = nullptr;
            HIPCHK(hipMalloc(&at, atab.size() * sizeof(uint16_t)));
            ctx->rate_allocs.push_back(at);
            HIPCHK(hipMemcpy(at, atab.data(), atab.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            rd.tp_atab = at;
            rd.tp_np = bs ? rt->bs_np : rt->tp_np;
            rd.tp_mt = bs ? rt->bs_mt : rt->tp_mt;
            rd.tp_last_rows = (bs ? rt->bs_max_out : rt->tp_max_out) - 32 * (rd.tp_mt - 1);
        }
    }
    ctx->rate_fs.push_back(fs);
    ctx->rates_h.push_back(rd);
    if (!ctx->rates_d.ensure(ctx->rates_h.size() * sizeof(RateDev) + 16 * sizeof(RateDev))) return HR_ERR_OOM;
    HIPCHK(hipMemcpy(ctx->rates_d.p, ctx->rates_h.data(), ctx->rates_h.size() * sizeof(RateDev), hipMemcpyHostToDevice));
    return static_cast<int>(ctx->rates_h.size() - 1);
}

int validate_track(const hr_ctx *ctx, const hr_track &t)
{
    if (t.channels < 1 || t.channels > HR_MAX_CH) return HR_ERR_FORMAT;
    if (fmt_bytes(t.fmt) == 0) return HR_ERR_FORMAT;
    // (the upper limit also bounds what rate_tables() builds per rate: O(sample_rate) doubles and work)
    if (t.sample_rate < 10 || t.sample_rate > HR_MAX_SAMPLE_RATE) return HR_ERR_RATE;
    if (t.frames && !t.pcm) return HR_ERR_INVALID;
    if (ctx->cfg.tp_mode == HR_TP_SWR192K) {
        const RateTables *rt = rate_tables(t.sample_rate);
        if (!rt || rt->swr_phases < 0) return HR_ERR_RATE;  // would need libswresample's phase interpolation
    }
    return HR_OK;
}

bool fused_mc(const hr_ctx *ctx, const hr_track &t, const void *dev);  // (below, beside the other routing predicates)
bool fused_mc4(const hr_ctx *ctx, const TrackDesc &d, int tp_class);

// One launch group: descriptors already built in `descs` (host), PCM resident on the device.
struct Group {
    std::vector<TrackDesc> descs;
    std::vector<int> fmts;  // per desc
    uint64_t rec_slots = 0, hop_slots = 0, blk_slots = 0;  // totals: (hop, channel) records / sums, blocks
    uint32_t part_slots = 0;  // gating chunks
    std::vector<uint32_t> parents;  // descriptors with role 1 (measured through stereo planes)
};

void add_desc(Group &g, const hr_track &t, const void *dev_pcm, int rate_idx, bool stream_start, bool stream_end,
              bool has_left, uint64_t stream_frames, int tp_mode)
{
    TrackDesc d{};
    d.pcm = dev_pcm;
    d.frames = t.frames;
    d.fs = t.sample_rate;
    d.hop = t.sample_rate / 10;
    d.row_len = d.hop;
    d.parts = 1;
    d.n_hops = t.frames / d.hop;
    d.sub_from = d.n_hops;
    d.ch = t.channels;
    d.fmt = t.fmt;
    d.rate_idx = rate_idx;
    d.stream_start = stream_start;
    d.stream_end = stream_end;
    d.tp_enabled = tp_mode != HR_TP_SWR192K || stream_frames >= kSwrTaps + 1;
    d.tp_reflect = tp_mode == HR_TP_SWR192K;
    d.tp_drop = tp_mode == HR_TP_SWR192K ? 16 : 0;
    d.blk_first = has_left ? -kPreBlk : 0;
    const int64_t hops_avail = static_cast<int64_t>(d.n_hops) + (has_left ? kPreBlk : 0);
    d.n_blocks = hops_avail >= 4 ? static_cast<uint64_t>(hops_avail - 3) : 0;
    d.rec_base = g.rec_slots;
    d.hop_base = g.hop_slots;
    d.blk_base = g.blk_slots;
    g.rec_slots += (d.n_hops + kPreRec) * d.ch;
    g.hop_slots += (d.n_hops + kPreHop) * d.ch;
    g.blk_slots += d.n_blocks;
    d.part_base = g.part_slots;
    d.n_parts = static_cast<uint32_t>(std::max<uint64_t>(1, (d.n_blocks + kGateChunk - 1) / kGateChunk));
    g.part_slots += d.n_parts;
    if (t.ch_weight) for (int c = 0; c < t.channels; c++) d.w[c] = t.ch_weight[c];
    else default_weights(t.channels, d.w);
    g.descs.push_back(d);
    g.fmts.push_back(t.fmt);
}

// Runs K-weight + combine (+ true peak) for a group.  Leaves hop sums / peaks on the device.
int run_measure(hr_ctx *ctx, Group &g, bool do_peaks, bool do_combine = true)
{
    hipStream_t st = ctx->stream;
    const size_t nt = g.descs.size();
    if (!nt) return HR_OK;
    if (!ctx->hop_sums.ensure((g.hop_slots + 8) * sizeof(double))) return HR_ERR_OOM;
    if (!ctx->blk_powers.ensure(std::max<uint64_t>(g.blk_slots, 1) * sizeof(double))) return HR_ERR_OOM;
    if (!ctx->peaks.ensure(nt * kMaxCh * sizeof(unsigned long long))) return HR_ERR_OOM;
    const RateDev *dr = ctx->rates_d.as<RateDev>();

    // Which tracks take the fused kernel (hr_kernels_fused.hip)?  Stereo f32, 16-byte aligned, at least two
    // hops; its true-peak half additionally needs HR_TP_SWR192K and the rate's f16 tile table.  The fused
    // kernel covers hops 0 .. n_hops-2; the last hop and the ragged tail stay on the generic kernels.
    std::vector<uint8_t> fkw(nt, 0), ftp(nt, 0), fsp(nt, 0);  // fused K-weighting / + true-peak screen / + sample peak
    std::vector<uint8_t> fmc(nt, 0);  // ... as a 5.1 / 7.1 track read directly (fused_mc): waves per item = channels / 2
    for (size_t i = 0; i < nt; i++) {
        const TrackDesc &d = g.descs[i];
        bool ok = d.role != 1 && ctx->cfg.engine != 1 && (d.fmt == HR_FMT_F32 || d.fmt == HR_FMT_S16 || (d.fmt == HR_FMT_S24 && d.hop % 2 == 0)) && d.ch == 2 && d.n_hops >= 2 &&
                  d.hop >= 64 && (reinterpret_cast<uintptr_t>(d.pcm) & 15) == 0 && d.hop * 8ull * 64 < 0xffffffffull;
        if (d.role == 0 && d.ch > 2 && do_peaks) {
            hr_track as_track{};
            as_track.frames = d.frames; as_track.sample_rate = d.fs; as_track.channels = d.ch; as_track.fmt = d.fmt;
            ok = fused_mc(ctx, as_track, d.pcm);
            fmc[i] = ok;
        }
        const bool tp_ok = do_peaks && (ctx->cfg.tp_mode == HR_TP_SWR192K || ctx->cfg.tp_mode == HR_TP_BS1770) &&
                           ctx->rates_h[d.rate_idx].tp_atab != nullptr;
        // stereo s16 is read as it is by the full pass only (K-weighting + true-peak screen); see fused_s16()
        if (d.fmt != HR_FMT_F32 && (!tp_ok || ctx->no_s16_direct)) ok = false;
        if (fmc[i] && !tp_ok) ok = false, fmc[i] = 0;
        fkw[i] = ok;
        ftp[i] = ok && tp_ok;
        // the peak is the plain sample peak: HR_TP_SAMPLE, 192 kHz input (nothing to interpolate)
        const bool plain = ctx->cfg.tp_mode == HR_TP_SAMPLE || ctx->rates_h[d.rate_idx].swr_phases == 0;
        fsp[i] = ok && do_peaks && !ftp[i] && plain;
    }
    // The track's final hop joins the fused rows (instead of the generic one-lane-per-hop kernel) when the stream ends
    // with this track and the fused kernel can keep the row honest (hr_kernels_fused.hip, "FINAL hop"): either the ragged
    // tail is long enough that every byte the row touches exists (the row reads 16 (T + 1) - hop <= 31 frames past the
    // hop), or the track's image ends on a 16-byte boundary, so that a chunk that STARTS inside the track ends inside it.
    // And the ragged tail behind it becomes one more row (row n_hops, for its true peak only: no block ends in it) when it
    // is worth a lane: the row holds the tail's frames up to the last 16-byte boundary of the track's image (fin_valid), its
    // tiles are screened as far as their windows stay inside those frames and in front of the positions that produce no
    // output, and the generic kernel starts at the first tile that is not (fend).
    std::vector<uint8_t> ffin(nt, 0);
    std::vector<uint64_t> &fend = ctx->h_fend;  // fused tracks: first position the fused kernel does not screen
    fend.assign(nt, 0);
    // Rows shorter than a hop (RateTables::sub_parts).  A wave costs its steps whatever the number of its lanes that hold a
    // row, and a launch costs whole rounds of the GPU's 2048 wave slots: short tracks (100 hops are one full wave and one
    // of 36 rows) and small batches (one 3-minute track is 29 waves) do better with rows of a third of a hop -- three
    // times the rows, a third of the steps, at a few steps of fixed cost per row (the start of a row accumulates all
    // four carry products; its last piece is partial).  Whole tracks whose last hop is a fused row only; never time shards.
    const bool sub_ok = !ctx->split && !ctx->no_alloc && ctx->shard_group != &g && ctx->cfg.engine != 2;
    // per track: hops [0, sub_from) are one row each, hops [sub_from, n_hops) are sub_p rows each (sub_from == n_hops: none)
    std::vector<uint32_t> sub_p(nt, 1);
    std::vector<uint64_t> sub_from(nt, 0);
    for (size_t i = 0; i < nt; i++) sub_from[i] = g.descs[i].n_hops;
    {
        auto final_row_ok = [&](const TrackDesc &d, uint64_t len) {  // (see ffin below)
            const uint32_t fb = fmt_bytes(d.fmt) * d.ch;
            const uint64_t T = (len + 15) / 16, row0 = d.n_hops * d.hop - len;
            return d.stream_end && !d.right_head && (d.frames - row0 >= 16 * (T + 1) || (d.frames - row0) % chunk_frames(fb) == 0);
        };
        // waves and wave-steps of a track as whole hops (0), as shorter rows throughout (1), and with only the hops that
        // do not fill a wave of whole-hop rows as shorter rows (2: those waves run a third of the steps)
        struct Cost { uint64_t waves[3] = {0, 0, 0}, steps[3] = {0, 0, 0}; bool can = false; };
        std::vector<Cost> cost(nt);
        for (size_t i = 0; i < nt && sub_ok; i++) {
            const TrackDesc &d = g.descs[i];
            if (!fkw[i]) continue;
            const RateDev &rd = ctx->rates_h[d.rate_idx];
            Cost &c = cost[i];
            const uint64_t t1 = (d.hop + 15) / 16 + 1;
            c.waves[0] = (d.n_hops + 63) / 64;
            c.steps[0] = c.waves[0] * t1;
            const uint64_t P = static_cast<uint64_t>(rd.sub_parts);
            const uint64_t len = P > 1 ? d.hop / P : d.hop;
            c.can = P > 1 && !d.left_tail && !d.right_head && d.blk_first == 0 && d.stream_start &&
                    (len * fmt_bytes(d.fmt) * d.ch) % 16 == 0 && final_row_ok(d, len);
            for (int k = 1; k < 3; k++) { c.waves[k] = c.waves[0]; c.steps[k] = c.steps[0]; }
            if (!c.can) continue;
            const uint64_t tP = (len + 15) / 16 + 1;
            c.waves[1] = (d.n_hops * P + 63) / 64;
            c.steps[1] = c.waves[1] * tP;
            const uint64_t left = d.n_hops % 64, full = d.n_hops / 64;
            if (left && full) {
                const uint64_t wl = (left * P + 63) / 64;
                c.waves[2] = full + wl;
                c.steps[2] = full * t1 + wl * tP;
            }
        }
        // (an item of a multichannel track read directly is a workgroup of its own, and a CU holds two of them, as it holds two
        // workgroups of four stereo waves: four of the 2048 slots, whether the workgroup is three waves or four)
        auto wpi_of = [&](size_t i) { return fmc[i] ? uint64_t{4} : uint64_t{1}; };
        // the track's own best: fuller waves must pay for the fixed cost of more rows (3 %); shorter LAST waves cost nothing
        auto own_choice = [&](size_t i) {
            const Cost &c = cost[i];
            if (!c.can) return 0;
            int best = 0;
            if (c.steps[2] * 1000 < c.steps[0] * 995) best = 2;
            if (c.steps[1] * 100 < c.steps[best] * 97) best = 1;
            return best;
        };
        // What a launch costs, in steps: its waves run in rounds of the GPU's wave slots, a last partial round costs half
        // a round plus its share (measured: profiles/README.md, round 2), one round or less costs one wave's steps.
        auto estimate = [&](auto choice) {
            double waves = 0, steps = 0;
            for (size_t i = 0; i < nt; i++) {
                if (!cost[i].waves[0]) continue;
                const int k = choice(i);
                waves += static_cast<double>(cost[i].waves[k] * wpi_of(i));
                steps += static_cast<double>(cost[i].steps[k] * wpi_of(i));
            }
            if (waves == 0) return 0.0;
            const double rounds = waves / 2048.0, whole = std::floor(rounds), frac = rounds - whole;
            const double eff = rounds <= 1.0 ? 1.0 : whole + (frac > 0 ? 0.5 + 0.5 * frac : 0.0);
            return eff * steps / waves;
        };
        auto all_sub = [&](size_t i) { return cost[i].can ? 1 : 0; };
        const double est_b = estimate(own_choice), est_c = estimate(all_sub);
        double waves_a = 0;
        for (size_t i = 0; i < nt; i++) waves_a += static_cast<double>(cost[i].waves[0] * wpi_of(i));
        // (measured: over several rounds shorter rows everywhere gain nothing -- 1000 three-minute tracks 19.8 against
        // 19.65 ms, and their combine kernel costs 0.4 ms more -- so that is for launches of at most one round)
        const bool everywhere = ctx->cfg.engine == 3 || (waves_a <= 2048.0 && est_c < 0.985 * est_b);
        for (size_t i = 0; i < nt && sub_ok; i++) {
            if (!cost[i].can) continue;
            int k = everywhere ? 1 : own_choice(i);
            if (ctx->cfg.engine == 4) k = cost[i].steps[2] != cost[i].steps[0] ? 2 : 0;
            if (k == 0) continue;
            sub_p[i] = static_cast<uint32_t>(ctx->rates_h[g.descs[i].rate_idx].sub_parts);
            sub_from[i] = k == 1 ? 0 : g.descs[i].n_hops / 64 * 64;
            // (the pinned mode of the tests also takes one more wave's hops, as the loop below does in large launches)
            if (ctx->cfg.engine == 4 && sub_from[i] >= 128) sub_from[i] -= 64;
        }
        // Whole rounds of whole-hop waves: the waves beyond the last full round of the 2048 slots would run as a partial
        // round of long waves; their hops become shorter rows instead (64 hops = one wave less per step of this loop),
        // which fill the last rounds in finer grain (measured: 300 tracks 6.88 -> 6.32 ms, 500 and 1000 tracks -0.7 %).
        if (!everywhere && (ctx->cfg.engine == 0 || ctx->cfg.engine >= 5)) {  // (5, 6, 7: as 0 but for one choice each)
            uint64_t whole = 0;
            for (size_t i = 0; i < nt; i++)
                if (cost[i].waves[0]) whole += (sub_from[i] + 63) / 64 * wpi_of(i);
            uint64_t excess = whole > 2048 ? whole % 2048 : 0;
            for (int pass = 0; pass < 64 && excess; pass++)
                for (size_t i = 0; i < nt && excess; i++) {
                    if (!cost[i].can || sub_from[i] < 64 || sub_from[i] % 64) continue;
                    sub_p[i] = static_cast<uint32_t>(ctx->rates_h[g.descs[i].rate_idx].sub_parts);
                    sub_from[i] -= 64;
                    excess -= std::min(excess, wpi_of(i));
                }
        }
    }
    g.rec_slots = 0;
    for (size_t i = 0; i < nt; i++) {
        TrackDesc &d = g.descs[i];
        d.parts = sub_p[i];
        d.sub_from = sub_from[i];
        d.row_len = d.hop / d.parts;
        d.rec_base = g.rec_slots;
        g.rec_slots += (sub_from[i] + (d.n_hops - sub_from[i]) * d.parts + kPreRec) * d.ch;
    }
    if (!ctx->recs.ensure((g.rec_slots + 8) * sizeof(HopRec))) return HR_ERR_OOM;
    std::vector<uint64_t> tail_rows_of(nt, 0);
    for (size_t i = 0; i < nt; i++) {
        TrackDesc &d = g.descs[i];
        d.fin_tail = 0;
        d.fin_valid = 0;
        if (!fkw[i]) continue;
        const uint64_t len = sub_from[i] < d.n_hops ? d.row_len : d.hop;  // rows of the track's last section
        const uint64_t last_hop0 = d.n_hops * d.hop - len;  // (start of the last row that belongs to a hop)
        fend[i] = last_hop0;
        // The same care serves a time shard that has a right neighbour (round 3): its last hop ends with the shard's buffer,
        // the frames behind it live in the halo.  The row K-weights the whole hop and screens the tiles whose windows stay
        // inside the shard; the generic kernel, which reads the halo, takes the rest from fend on.  (Before, that hop went
        // to the one-lane-per-channel kernel: 0.5 - 1 ms of latency beside a fused kernel of 0.5 ms per eighth of a stream.)
        const bool track_end = d.stream_end && !d.right_head, shard_mid = !d.stream_end && d.right_head != nullptr;
        if (!(track_end || shard_mid) || ctx->split) continue;  // (the experiments' split-wave kernel has no final-row care)
        const uint64_t ragged = d.frames - d.n_hops * d.hop;
        const uint32_t fb = fmt_bytes(d.fmt) * d.ch;
        const uint64_t T = (len + 15) / 16;
        // The DMA moves 16-byte chunks counted from the ROW's first byte (rows need not start on a 16-byte boundary of the
        // track's image: 44.1 kHz s16 / s24, odd hop lengths), and a chunk is fetched when it starts inside the track.  So a row
        // that reaches the end of the track must see that end on its own chunk grid: a whole number of `cf` frames from its start.
        const uint64_t cf = chunk_frames(fb);
        // (a row fetches pieces 0 .. T: 16 (T + 1) frames from its start)
        ffin[i] = d.frames - last_hop0 >= 16 * (T + 1) || (d.frames - last_hop0) % cf == 0;
        if (!ffin[i]) continue;
        // rows behind the last hop: the tail's whole rows (only when a row is less than a hop) and, if what is left is
        // worth a lane, one partial row; all of them for their peak only.  They start at n_hops * hop + j * len: one row
        // (len == hop), or rows whose length is a whole number of chunks (run_measure's `can`), so every one of them sees
        // the same chunk grid and `usable` ends on it.
        const uint64_t usable = (d.frames - d.n_hops * d.hop) / cf * cf;
        uint64_t tail_rows = 0;
        if (ftp[i] && ragged >= 64 && usable >= 48) {
            tail_rows = usable / len;
            if (usable - tail_rows * len >= 48) tail_rows++;
        }
        tail_rows_of[i] = tail_rows;
        if (tail_rows) {
            d.fin_tail = static_cast<uint8_t>(tail_rows);
            const uint64_t l0 = d.n_hops * d.hop + (tail_rows - 1) * len;
            d.fin_valid = static_cast<uint32_t>(d.n_hops * d.hop + usable - l0);
            const uint64_t keep = d.fin_valid > d.tp_drop ? d.fin_valid - d.tp_drop : 0;
            fend[i] = l0 + 16 * std::min<uint64_t>(keep / 16, T);
        } else {
            const uint64_t left = d.frames - last_hop0;
            d.fin_valid = static_cast<uint32_t>(std::min<uint64_t>(left, 1u << 30));
            const uint64_t keep = d.fin_valid > d.tp_drop ? d.fin_valid - d.tp_drop : 0;
            // (the sample-peak build takes no part in this: its generic kernel starts where it always did)
            fend[i] = last_hop0 + 16 * (ftp[i] ? std::min<uint64_t>(keep / 16, T) : T - 2);
        }
    }
    ctx->ht("plan rows");
    int rc = upload(ctx->tracks_d, g.descs.data(), nt * sizeof(TrackDesc), st);
    if (rc) return rc;
    ctx->ht("upload descs");
    const TrackDesc *dt = ctx->tracks_d.as<TrackDesc>();
    std::vector<FusedItem> &fitems = ctx->h_fitems;
    fitems.clear();
    // items sorted by kernel specialisation: classes 0..5 = K-weight + true peak (tile shape of the rate),
    // class 6 = K-weight only, class 7 = K-weight + sample peak, classes 8..13 / 14..19 = classes 0..5 reading stereo s16 / s24,
    // classes 20 + 6 (3 [7.1 instead of 5.1] + sample format) + (0..5) = a 5.1 / 7.1 track read directly (fused_mc),
    // classes 56 + 6 (sample format) + (0..5) = 5.1 read directly by the four-wave workgroup (fused_mc4: three candidate lists per item)
    constexpr int kStereoClasses = 20, kFourWaveClasses = kStereoClasses + 36, kFusedClasses = kFourWaveClasses + 18;
    size_t class_off[kFusedClasses + 1];
    auto class_wpi = [&](int cls) -> uint32_t { return cls < kStereoClasses ? 1u : cls >= kFourWaveClasses ? 3u : ((cls - kStereoClasses) / 18 ? 4u : 3u); };
    // The four waves of a workgroup share the rate's tables in LDS and run in lock step, so a workgroup never
    // mixes rates or row lengths: every (class, rate, row length) run is padded to a multiple of 4 with n_rows == 0
    // copies of its last item.
    // (the fused tracks of every (class, rate), in track order: the loops below visit each track once per row length
    // instead of once per class, rate and row length -- a launch of 10 000 short clips spent a millisecond there)
    const size_t n_rates = ctx->rates_h.size();
    std::vector<std::vector<uint32_t>> by_class_rate(static_cast<size_t>(kFusedClasses) * n_rates);
    for (size_t i = 0; i < nt; i++) {
        if (!fkw[i]) continue;
        const RateDev &rd = ctx->rates_h[g.descs[i].rate_idx];
        const int tpc = fused_tp_class(rd.tp_mt, rd.tp_last_rows);
        const int mine = fmc[i] ? (fused_mc4(ctx, g.descs[i], tpc) ? kFourWaveClasses + 6 * g.fmts[i] + tpc : kStereoClasses + 6 * ((g.descs[i].ch == 8 ? 3 : 0) + g.fmts[i]) + tpc)
                       : ftp[i] ? fused_tp_class(rd.tp_mt, rd.tp_last_rows) + (g.fmts[i] == HR_FMT_S16 ? 8 : g.fmts[i] == HR_FMT_S24 ? 14 : 0) : fsp[i] ? 7 : 6;
        by_class_rate[static_cast<size_t>(mine) * n_rates + g.descs[i].rate_idx].push_back(static_cast<uint32_t>(i));
    }
    // Line-aligned rows (hr_fused_core.hpp, FusedItem::phi): stereo f32 whose hops do not start on cache lines (44.1 kHz: a hop is
    // 35 280 bytes) but whose buffer does.  Hops h and h + 8 start at the same offset from the 128-byte grid, so the whole-hop
    // rows of a track are taken in blocks of 512 hops as eight items of 64 rows, hops 512 b + j + 8 r, whose piece grids start
    // phi(j) frames early, on a line; what is left of the track behind its last whole block stays as it was.  The four waves of
    // a workgroup share the coefficient tiles and the carry-table slices, so every phi is a run of its own.
    // hr_config.engine = 7 keeps all rows as they were (A/B and tests).
#ifdef HR_EXPERIMENTS  // HR_ALIGN_MODE=1: strided items whose grids are NOT shifted (what the stride alone costs: +4.4 % kernel time)
    static const int align_mode = std::getenv("HR_ALIGN_MODE") ? std::atoi(std::getenv("HR_ALIGN_MODE")) : 0;
    static const size_t align_super = std::getenv("HR_ALIGN_SUPER") ? static_cast<size_t>(std::atol(std::getenv("HR_ALIGN_SUPER"))) : 8;
    // HR_ALIGN_STRIDE (with HR_ALIGN_MODE=1 only: the offsets repeat every 8 hops): hops between the rows of a strided item
    static const uint64_t kAlignedStride = align_mode == 1 && std::getenv("HR_ALIGN_STRIDE") ? static_cast<uint64_t>(std::atol(std::getenv("HR_ALIGN_STRIDE"))) : 8;
#else
    constexpr int align_mode = 0;
    constexpr size_t align_super = 8;
    constexpr uint64_t kAlignedStride = 8;
#endif
    const uint64_t kAlignedBlock = 64 * kAlignedStride;
    auto phi_of = [&](const TrackDesc &d, uint64_t j) { return align_mode == 1 ? 0u : static_cast<uint32_t>(((j * d.hop * 8ull) % 128) / 8); };
    // per rate and phi / 2: the tile pattern a grid that starts phi frames before a hop begins with (16 pat0 + phi is a whole
    // number of input steps); -1: there is none
    std::vector<int> pat0_tab(n_rates * 8, 0);
    for (size_t ri = 0; ri < n_rates; ri++) {
        const RateDev &rd = ctx->rates_h[ri];
        for (uint32_t q = 0; q < 8; q++) {
            int found = rd.tp_np <= 1 || rd.swr_step <= 1 ? 0 : -1;
            for (int p = 0; p < rd.tp_np && found < 0; p++)
                if ((16ll * p + 2 * q) % rd.swr_step == 0) found = p;
            pat0_tab[ri * 8 + q] = found;
        }
    }
    // per track: whole blocks of 512 hops that take shifted grids (in front of everything else of the track)
    std::vector<uint64_t> al_blocks(nt, 0);
    for (size_t i = 0; i < nt; i++) {
        const TrackDesc &d = g.descs[i];
        if (!fkw[i] || fmc[i] || g.fmts[i] != HR_FMT_F32 || d.ch != 2 || ctx->cfg.engine == 7 || ctx->split || (d.hop * 8ull) % 128 == 0 ||
            (reinterpret_cast<uintptr_t>(d.pcm) & 127))
            continue;
        bool ok = true;
        for (uint64_t j = 0; j < kAlignedStride && ok; j++) ok = phi_of(d, j) % 2 == 0 && pat0_tab[d.rate_idx * 8 + phi_of(d, j) / 2] >= 0;
        const uint64_t limit = sub_from[i] < d.n_hops ? sub_from[i] : d.n_hops - 1;  // (never the track's final row)
        if (ok) al_blocks[i] = limit / kAlignedBlock;
    }
    for (int cls = 0; cls < kFusedClasses; cls++) {
        class_off[cls] = fitems.size();
        // (rate, rows with shifted grids), (rate, whole-hop rows), (rate, shorter rows): the long items first -- workgroups start
        // in launch order, and the short items are there to fill the launch's last round in finer grain (with the shifted-grid
        // items behind them the last round was all long items again: +1.6 % kernel time)
        for (size_t rp = 0; rp < 3 * n_rates; rp++) {
            const size_t ri = rp / 3;
            const bool shorter = rp % 3 == 2;
            const size_t run_begin = fitems.size();
            if (rp % 3 == 0) {
                // The (track, block) pairs of the rate four at a time: for each of the eight offsets a workgroup of the four
                // pairs' items with that offset.  Eight consecutive workgroups then read the same four blocks completely --
                // rows 8 hops apart leave seven eighths of a block to the other seven items, and with a run per offset those
                // came a whole sweep over the launch's tracks later (the strided items alone cost 4.4 % that way).
                std::vector<std::pair<uint32_t, uint64_t>> pairs;
                for (const uint32_t i : by_class_rate[static_cast<size_t>(cls) * n_rates + ri])
                    for (uint64_t b = 0; b < al_blocks[i]; b++) pairs.emplace_back(i, b);
                // (Workgroups go to the XCDs in turn, so the eight workgroups of four pairs are put eight apart: they meet in one
                // L2, where the line that ends hop h and starts hop h + 1 is fetched once for both.)
                const size_t n_chunks = (pairs.size() + kFusedWaves - 1) / kFusedWaves;
                for (size_t sc = 0; sc < n_chunks; sc += align_super)
                    for (uint64_t j = 0; j < kAlignedStride; j++)
                        for (size_t c = sc; c < std::min(sc + align_super, n_chunks); c++)
                            for (size_t w = 0; w < static_cast<size_t>(kFusedWaves); w++) {
                                const size_t pi = c * kFusedWaves + w;
                                const bool real = pi < pairs.size();
                                const auto &pr = pairs[real ? pi : pairs.size() - 1];  // (padding: a copy of the last pair's item, no rows)
                                const TrackDesc &d = g.descs[pr.first];
                                const uint32_t phi = phi_of(d, j);
                                FusedItem it{};
                                it.track = pr.first;
                                it.n_rows = real ? 64 : 0;
                                it.first_hop = pr.second * kAlignedBlock + j;
                                it.frame0 = it.first_hop * d.hop;
                                it.row_len = d.hop;
                                it.row_stride = static_cast<uint16_t>(kAlignedStride);
                                it.phi = static_cast<uint16_t>(phi);
                                it.pat0 = static_cast<uint32_t>(pat0_tab[d.rate_idx * 8 + phi / 2]);
                                fitems.push_back(it);
                            }
                continue;
            }
            for (const uint32_t i : by_class_rate[static_cast<size_t>(cls) * n_rates + ri]) {
                const TrackDesc &d = g.descs[i];
                const uint64_t nb = al_blocks[i];
                // this section's rows: record rows [r0, r1) from frame f0 on, `len` frames each; the track's last section
                // also holds the rows behind the last hop, and loses the final hop's row when that one stays generic
                const bool has_sub = sub_from[i] < d.n_hops;
                uint64_t r0, r1, f0, len;
                if (!shorter) {
                    r0 = nb * kAlignedBlock; r1 = sub_from[i]; f0 = r0 * d.hop; len = d.hop;  // (behind the aligned blocks)
                    if (!has_sub) r1 = r1 - (ffin[i] ? 0 : 1) + tail_rows_of[i];
                } else {
                    if (!has_sub) continue;
                    r0 = sub_from[i]; r1 = sub_from[i] + (d.n_hops - sub_from[i]) * d.parts + tail_rows_of[i];
                    f0 = sub_from[i] * d.hop; len = d.row_len;
                }
                const bool last_section = shorter == has_sub;
                for (uint64_t r = r0; r < r1; r += 64) {
                    FusedItem it{};
                    it.track = static_cast<uint32_t>(i);
                    it.n_rows = static_cast<uint32_t>(std::min<uint64_t>(64, r1 - r));
                    it.first_hop = r;
                    it.frame0 = f0 + (r - r0) * len;
                    it.row_len = static_cast<uint32_t>(len);
                    it.last = last_section && ffin[i] && r + 64 >= r1;
                    it.row_stride = 1;
                    fitems.push_back(it);
                }
            }
            if (fitems.size() > run_begin && cls < kStereoClasses) {  // (a multichannel item is a workgroup of its own: nothing to pad)
                FusedItem pad = fitems.back();
                pad.last = pad.last && pad.n_rows == 1;  // (a padding wave works on row 0 of the wave it copies)
                pad.n_rows = 0;
                while ((fitems.size() - run_begin) % kFusedWaves) fitems.push_back(pad);
            }
        }
    }
    class_off[kFusedClasses] = fitems.size();
    ctx->ht("fused items");
    if (!fitems.empty()) {
        rc = upload(ctx->fitems_d, fitems.data(), fitems.size() * sizeof(FusedItem), st);
        if (rc) return rc;
    }
    ctx->ht("upload fused items");

    // candidate lists: one per wave of the launch
    ctx->h_runs.clear();
    size_t n_lists = 0;
    for (int cls = 0; cls < kFusedClasses; cls++) {
        const size_t n = class_off[cls + 1] - class_off[cls];
        ctx->h_runs.push_back({class_off[cls], n, n_lists, class_wpi(cls)});
        n_lists += n * class_wpi(cls);
    }
    FusedScratch fs{};
    if (!fitems.empty()) {
        if (!ctx->cands.ensure(fused_cand_bytes(static_cast<uint32_t>(n_lists))) ||
            !ctx->cand_cnt.ensure(n_lists * sizeof(unsigned)) || !ctx->tpb.ensure(nt * kMaxCh * 2 * sizeof(unsigned)) ||
            !ctx->overflow.ensure(nt * sizeof(unsigned)))
            return HR_ERR_OOM;
        fs.cands = ctx->cands.p;
        fs.cand_cnt = ctx->cand_cnt.as<unsigned>();
        fs.tpb = ctx->tpb.as<unsigned>();
        fs.overflow = ctx->overflow.as<unsigned>();
        HIPCHK(hipMemsetAsync(ctx->tpb.p, 0, nt * kMaxCh * 2 * sizeof(unsigned), st));
        HIPCHK(hipMemsetAsync(ctx->overflow.p, 0, nt * sizeof(unsigned), st));
    }
    auto ext_hops = [&](size_t i) { return i < ctx->fp_ext_hops.size() ? std::min<uint64_t>(ctx->fp_ext_hops[i], g.descs[i].n_hops) : 0; };
    // work items of the generic kernels that run beside the fused kernel: [fmt] hop items, [fmt][kind] peak items
    std::vector<WorkItem> &items = ctx->h_items;
    items.clear();
    struct Span { size_t off, n; };
    Span hop_span[5], pk_span[5][2];
    for (int f = 0; f < 5; f++) {
        hop_span[f].off = items.size();
        for (size_t i = 0; i < nt; i++) {
            if (g.fmts[i] != f || g.descs[i].role == 1) continue;
            if (fkw[i]) continue;  // fused tracks: only the last hop is left, see tail_span
            // (first pass: the hops k_first_pass_rows measures are not the generic kernel's)
            for (uint64_t h = ext_hops(i); h < g.descs[i].n_hops; h += 256)
                items.push_back({static_cast<uint32_t>(i), 0, h});
        }
        hop_span[f].n = items.size() - hop_span[f].off;
    }
    // last hop of the fused tracks (all f32): one lane per (track, channel)
    Span tail_span[3];  // f32, s16, s24
    for (int f = 0; f < 3; f++) {
        tail_span[f].off = items.size();
        for (size_t i = 0; i < nt; i++)
            if (fkw[i] && !ffin[i] && g.fmts[i] == f)
                for (uint32_t c = 0; c < g.descs[i].ch; c++) items.push_back({static_cast<uint32_t>(i), c, g.descs[i].n_hops - 1});
        tail_span[f].n = items.size() - tail_span[f].off;
    }
    // kind 0 = the configured interpolator, kind 1 = sample peak (192 kHz input: libswresample does not
    // resample, so f_ebur128's "true peak" is the sample peak; BS.1770 asks for no oversampling there)
    for (int f = 0; f < 5; f++)
        for (int kind = 0; kind < 2; kind++) {
            pk_span[f][kind].off = items.size();
            if (do_peaks)
                for (size_t i = 0; i < nt; i++) {
                    if (g.fmts[i] != f || g.descs[i].role == 1) continue;
                    const bool plain = ctx->cfg.tp_mode != HR_TP_SAMPLE && ctx->rates_h[g.descs[i].rate_idx].swr_phases == 0;
                    if (plain != (kind == 1)) continue;
                    // fused tracks: the generic kernel finishes the track -- from the last hop on, or, when the last rows
                    // are fused too, from the first position of the first tile that is not screened there
                    const uint64_t n0 = !(ftp[i] || fsp[i]) ? ext_hops(i) * g.descs[i].hop : fend[i];
                    // (the tail of a fused track is cut into short items: many brief workgroups beside the fused kernel
                    // instead of one long-running one)
                    const uint32_t span = ftp[i] && kind == 0 ? kTailSpan : 0;
                    for (uint64_t n = n0; n < g.descs[i].frames; n += span ? span : kPeakSpan)
                        items.push_back({static_cast<uint32_t>(i), span, n});
                }
            pk_span[f][kind].n = items.size() - pk_span[f][kind].off;
        }
    if (!items.empty()) {
        rc = upload(ctx->items_d, items.data(), items.size() * sizeof(WorkItem), st);
        if (rc) return rc;
    }
    const WorkItem *di = ctx->items_d.as<WorkItem>();
    ctx->h_ftp.assign(ftp.begin(), ftp.end());
    ctx->ht("generic items + upload");

    HIPCHK(hipEventRecord(ctx->ev[0], st));
    if (do_peaks) HIPCHK(hipMemsetAsync(ctx->peaks.p, 0, nt * kMaxCh * sizeof(unsigned long long), st));
    // fork point for the generic kernels on the side stream: everything they and the fused kernel share (descriptors,
    // rate tables, the zeroed peak slots) has been issued on st before it
    hipStream_t tail_st = ctx->serial_tail ? st : ctx->side;
    HIPCHK(hipEventRecord(ctx->ev_fork, st));
    if (!ctx->serial_tail) HIPCHK(hipStreamWaitEvent(tail_st, ctx->ev_fork, 0));

    for (int f = 0; f < 5; f++)
        launch_hop_zero_state(f, dt, dr, di + hop_span[f].off, static_cast<uint32_t>(hop_span[f].n), ctx->recs.as<HopRec>(), tail_st);
    for (int f = 0; f < 3; f++)
        launch_hop_zero_state_list(f, dt, dr, di + tail_span[f].off, static_cast<uint32_t>(tail_span[f].n), ctx->recs.as<HopRec>(), tail_st);
    if (do_peaks) {
        for (int f = 0; f < 5; f++)
            for (int kind = 0; kind < 2; kind++) {
                if (!pk_span[f][kind].n) continue;
                // (both interpolators are banks of the same shape: one kernel, the context's bank in the rate tables)
                launch_true_peak_generic(f, kind == 1 || ctx->cfg.tp_mode == HR_TP_SAMPLE ? HR_TP_SAMPLE : HR_TP_SWR192K, dt, dr,
                                         di + pk_span[f][kind].off, static_cast<uint32_t>(pk_span[f][kind].n), kPeakSpan,
                                         ctx->peaks.as<unsigned long long>(), tail_st);
            }
    }
    HIPCHK(hipEventRecord(ctx->ev_join, tail_st));
    HIPCHK(hipEventRecord(ctx->ev[2], st));
    if (!ctx->fp_ext_hops.empty()) {
        // hr_loudnorm_first_pass: the leading hops of the 192 kHz pseudo-tracks straight from their source PCM
        std::vector<FpItem> &fp = ctx->h_fp_items;
        fp.clear();
        for (size_t i = 0; i < nt && i < ctx->fp_ext_hops.size(); i++) {
            const uint64_t eh = ext_hops(i);
            if (!eh) continue;
            FpSrc &src = ctx->fp_srcs[i];
            src.rec_base = g.descs[i].rec_base;
            src.desc = static_cast<uint32_t>(i);
            src.rate192 = g.descs[i].rate_idx;
            for (uint32_t c = 0; c < src.channels; c++)
                for (uint64_t h = 0; h < eh; h += 64)
                    fp.push_back({static_cast<uint32_t>(i), c, static_cast<uint32_t>(h), static_cast<uint32_t>(std::min<uint64_t>(64, eh - h))});
        }
        if (!fp.empty()) {
            rc = upload(ctx->fp_src_d, ctx->fp_srcs.data(), ctx->fp_srcs.size() * sizeof(FpSrc), st);
            if (!rc) rc = upload(ctx->fp_items_d, fp.data(), fp.size() * sizeof(FpItem), st);
            if (rc) return rc;
            launch_first_pass_rows(ctx->fp_src_d.as<FpSrc>(), dr, ctx->fp_items_d.as<FpItem>(), static_cast<uint32_t>(fp.size()),
                                   ctx->recs.as<HopRec>(), ctx->peaks.as<unsigned long long>(), st);
        }
    }
    {
        const FusedItem *fi = ctx->fitems_d.as<FusedItem>();
        // HR_ABLATE (timing experiments): low 2 bits pick the kernel variant, bits 4.. are the kernel's dbg flags
        const int v0 = ctx->ablate & 3, dbg = ctx->ablate >> 4;
        for (int cls = 0; cls < kFusedClasses; cls++) {
            const size_t off = class_off[cls], n = class_off[cls + 1] - off;
            if (!n) continue;
            FusedScratch fsc = fs;
            fsc.cands = static_cast<char *>(fs.cands) + fused_cand_bytes(static_cast<uint32_t>(ctx->h_runs[cls].list_off));
            fsc.cand_cnt = fs.cand_cnt + ctx->h_runs[cls].list_off;
            if (cls >= kStereoClasses) {
                const bool four = cls >= kFourWaveClasses;
                const int sub = cls - (four ? kFourWaveClasses : kStereoClasses);
                launch_fused_mc(sub / 18 ? 8 : 6, (sub / 6) % 3, sub % 6, four, dt, dr, fi + off, static_cast<uint32_t>(n), ctx->recs.as<HopRec>(), fsc,
                                ctx->peaks.as<unsigned long long>(), st);
                continue;
            }
            const bool tp = cls < 6;
            launch_fused_f32s(dt, dr, fi + off, static_cast<uint32_t>(n), ctx->recs.as<HopRec>(), fsc,
                              ctx->peaks.as<unsigned long long>(),
                              cls >= 14 ? 6 : cls >= 8 ? 5 : tp ? (v0 == 0 && dbg == 0 && ctx->split ? 4 : v0) : (v0 == 3 ? 3 : 1),
                              cls >= 14 ? cls - 14 : cls >= 8 ? cls - 8 : cls, cls == 7 ? (dbg | 64) : dbg, st);
        }
    }
    HIPCHK(hipEventRecord(ctx->ev[3], st));
    ctx->ht("launch fused");
    // The fused kernel is on its way: the host builds the combine kernel's work items beside it (one per 256 (hop, channel)
    // pairs: the bulk of the generic items); their upload queues behind the fused kernel, where the combine kernel follows
    std::vector<WorkItem> &citems = ctx->h_comb_items;
    citems.clear();
    for (size_t i = 0; i < nt; i++) {
        const auto &d = g.descs[i];
        if (d.role == 1) continue;
        const uint64_t pairs = (d.n_hops + (d.blk_first < 0 ? kPreBlk : 0)) * d.ch;
        for (uint64_t p = 0; p < pairs; p += 256) citems.push_back({static_cast<uint32_t>(i), 0, p});
    }
    if (!citems.empty()) {
        rc = upload(ctx->comb_items_d, citems.data(), citems.size() * sizeof(WorkItem), st);
        if (rc) return rc;
    }
    HIPCHK(hipStreamWaitEvent(st, ctx->ev_join, 0));
    ctx->comb_n = citems.size();
    if (do_combine)
        launch_hop_combine(dt, dr, ctx->comb_items_d.as<WorkItem>(), static_cast<uint32_t>(citems.size()), ctx->recs.as<HopRec>(),
                           ctx->hop_sums.as<double>(), std::any_of(sub_p.begin(), sub_p.end(), [](uint32_t p) { return p > 1; }), st);
    if (do_combine && !g.parents.empty()) {
        rc = upload(ctx->parents_d, g.parents.data(), g.parents.size() * sizeof(uint32_t), st);
        if (rc) return rc;
        launch_gather_planes(dt, ctx->parents_d.as<uint32_t>(), static_cast<uint32_t>(g.parents.size()), ctx->hop_sums.as<double>(), st);
    }
    HIPCHK(hipGetLastError());
    return HR_OK;
}

// finalize: relative gate, gated mean and the range statistics per track; range_parts (time shards): only the chunks'
// short-term histograms and partial sums, which the caller reduces over the shards
int run_gating(hr_ctx *ctx, Group &g, bool finalize, bool range_parts = false)
{
    hipStream_t st = ctx->stream;
    const uint32_t nt = static_cast<uint32_t>(g.descs.size());
    if (!nt) return HR_OK;
    // Whole tracks of a few chunks (finalize, not a time shard's partial statistics) are gated by one workgroup each with the
    // histograms in LDS (k_gate_small); longer ones and shards chunk by chunk through `hist` / `st_hist` in global memory.
    const bool small_ok = finalize && !range_parts;
    auto is_small = [&](uint32_t i) { return small_ok && g.descs[i].n_parts <= kGateSmallParts; };
    uint32_t n_long = 0;
    for (uint32_t i = 0; i < nt; i++) n_long += is_small(i) ? 0u : 1u;
    const uint32_t n_small = nt - n_long;
    const size_t hist_bytes = n_long ? static_cast<size_t>(nt) * kEbuBins * sizeof(uint32_t) : 0;
    if (!ctx->outs.ensure(static_cast<size_t>(nt) * sizeof(TrackOut)) || !ctx->gate_parts.ensure(static_cast<size_t>(g.part_slots) * sizeof(GatePart)) ||
        (n_long && (!ctx->hist.ensure(hist_bytes) || ((finalize || range_parts) && !ctx->st_hist.ensure(hist_bytes)))))
        return HR_ERR_OOM;
    // one workgroup per chunk of kGateChunk blocks: a 10-hour stream is 176 chunks; behind them the lists of the long and of
    // the small tracks (WorkItem::track)
    std::vector<WorkItem> &items = ctx->h_gate_items;  // (outlives the asynchronous copy)
    items.clear();
    for (uint32_t i = 0; i < nt; i++)
        if (!is_small(i))
            for (uint32_t c = 0; c < g.descs[i].n_parts; c++) items.push_back({i, c, static_cast<uint64_t>(c) * kGateChunk});
    const uint32_t ni = static_cast<uint32_t>(items.size());
    if (n_small)  // (no small tracks: the per-track kernels take tracks 0 .. nt - 1 as they are)
        for (int small = 0; small < 2; small++)
            for (uint32_t i = 0; i < nt; i++)
                if (is_small(i) == (small == 1)) items.push_back({i, 0, 0});
    int rc = upload(ctx->gate_items, items.data(), items.size() * sizeof(WorkItem), st);
    if (rc) return rc;
    const TrackDesc *dt = ctx->tracks_d.as<TrackDesc>();
    const WorkItem *di = ctx->gate_items.as<WorkItem>();
    const WorkItem *long_list = n_small ? di + ni : nullptr, *small_list = di + ni + n_long;
    if (n_long) {
        HIPCHK(hipMemsetAsync(ctx->hist.p, 0, hist_bytes, st));
        launch_block_hist(dt, di, ni, ctx->cfg.gate_mode, ctx->hop_sums.as<double>(), ctx->ln_bounds.as<double>(),
                          ctx->ebu_edges.as<double>(), ctx->blk_powers.as<double>(), ctx->hist.as<uint32_t>(), ctx->gate_parts.as<GatePart>(), st);
        launch_gate_final(dt, n_long, ctx->cfg.gate_mode, ctx->blk_powers.as<double>(), ctx->hist.as<uint32_t>(),
                          ctx->ebu_energy.as<double>(), ctx->ebu_edges.as<double>(), ctx->ln_energy.as<double>(), ctx->ln_bounds.as<double>(),
                          ctx->gate_parts.as<GatePart>(), !finalize, ctx->outs.as<TrackOut>(), st, long_list);
        if (finalize || range_parts) {
            HIPCHK(hipMemsetAsync(ctx->st_hist.p, 0, hist_bytes, st));
            launch_range_stats(dt, n_long, di, ni, ctx->cfg.gate_mode, ctx->hop_sums.as<double>(), ctx->blk_powers.as<double>(),
                               ctx->ln_bounds.as<double>(), ctx->ln_energy.as<double>(), ctx->ebu_edges.as<double>(), ctx->st_hist.as<uint32_t>(),
                               ctx->gate_parts.as<GatePart>(), finalize, ctx->outs.as<TrackOut>(), st, long_list);
        }
    }
    launch_gate_small(dt, small_list, n_small, ctx->cfg.gate_mode, ctx->hop_sums.as<double>(), ctx->blk_powers.as<double>(),
                      ctx->ebu_energy.as<double>(), ctx->ebu_edges.as<double>(), ctx->ln_energy.as<double>(), ctx->ln_bounds.as<double>(),
                      ctx->outs.as<TrackOut>(), st);
    HIPCHK(hipGetLastError());
    return HR_OK;
}

void fill_result(const hr_ctx *ctx, const TrackOut &o, const unsigned long long *pk, int channels, hr_result *r,
                 bool have_range = true)
{
    const double ninf = -std::numeric_limits<double>::infinity();
    const double nan = std::numeric_limits<double>::quiet_NaN();
    r->loudness_range_lu = r->lra_low_lufs = r->lra_high_lufs = r->lra_threshold_lufs = nan;
    r->max_momentary_lufs = r->max_shortterm_lufs = nan;
    if (have_range) {
        const bool ln = ctx->cfg.gate_mode == HR_GATE_LOUDNORM;
        auto loud = [&](double e) { return ln ? 10.0 * (std::log(e) / std::log(10.0)) - 0.691 : -0.691 + 10.0 * std::log10(e); };
        r->loudness_range_lu = 0.0;
        if (o.st_rel_energy > 0.0) r->lra_threshold_lufs = ln ? loud(0.01 * o.st_rel_energy) : loud(o.st_rel_energy) + (-20.0);
        if (o.lra_low_bin >= 0) {
            if (ln) {
                r->lra_low_lufs = loud(loudnorm_bin_energy()[o.lra_low_bin]);
                r->lra_high_lufs = loud(loudnorm_bin_energy()[o.lra_high_bin]);
            } else {
                r->lra_low_lufs = static_cast<double>(o.lra_low_bin) / 100.0 + (-70.0);
                r->lra_high_lufs = static_cast<double>(o.lra_high_bin) / 100.0 + (-70.0);
            }
            r->loudness_range_lu = r->lra_high_lufs - r->lra_low_lufs;
        }
        r->max_momentary_lufs = o.n_blocks ? loud(o.max_m_power) : ninf;
        r->max_shortterm_lufs = o.n_shortterm ? loud(o.max_s_power) : ninf;
    }
    r->n_blocks = o.n_blocks;
    r->n_gated = o.n_kept;
    r->integrated_lufs = ninf;
    r->rel_threshold_lufs = ninf;
    const int gm = ctx->cfg.gate_mode;
    if (o.n_kept) {
        if (gm == HR_GATE_EBUR128) r->rel_threshold_lufs = (-0.691 + 10.0 * std::log10(o.rel_energy)) + (-10.0);
        else if (gm == HR_GATE_LOUDNORM) r->rel_threshold_lufs = 10.0 * (std::log(o.rel_energy * 0.1) / std::log(10.0)) - 0.691;
        else r->rel_threshold_lufs = -0.691 + 10.0 * std::log10(0.1 * o.rel_energy);
        if (o.n_gated_final) {
            if (gm == HR_GATE_LOUDNORM) r->integrated_lufs = 10.0 * (std::log(o.mean_energy) / std::log(10.0)) - 0.691;
            else r->integrated_lufs = -0.691 + 10.0 * std::log10(o.mean_energy);
        }
    }
    double m = 0.0;
    for (int c = 0; c < HR_MAX_CH; c++) r->peak_per_ch[c] = 0.0;
    for (int c = 0; c < channels; c++) {
        double v;
        std::memcpy(&v, &pk[c], sizeof v);
        r->peak_per_ch[c] = v;
        if (v > m) m = v;
    }
    r->true_peak_dbtp = 20.0 * std::log10(m);
    r->status = HR_OK;
}

// A wave of the fused kernel ran out of candidate slots (sustained tones: every crest is a candidate of the f16 screen), or a
// track met samples beyond the f16 range: the positions that wave -- or that track's fused rows -- screened are redone
// exactly by k_true_peak_rows.  Synchronises the stream (the flags are read on the host).  Rare on music (a clipped or
// limited master lists a few dozen groups per wave); it keeps the exactness guarantee unconditional.
// Two halves, so that a caller can read the flags back with its results and synchronise once: overflow_flags_fetch queues
// the copies of the flags, overflow_redo (after the stream has been synchronised) launches the redo where a flag is set
// and says whether it did -- the caller then reads the peaks again.
int overflow_flags_fetch(hr_ctx *ctx, Group &g)
{
    hipStream_t st = ctx->stream;
    const size_t nt = g.descs.size();
    ctx->h_ovf.clear();
    ctx->h_cand_cnt.clear();
    bool any_fused = false;
    for (size_t i = 0; i < nt && i < ctx->h_ftp.size(); i++) any_fused |= ctx->h_ftp[i] != 0;
    if (!any_fused) return HR_OK;
    ctx->h_ovf.resize(nt);
    HIPCHK(hipMemcpyAsync(ctx->h_ovf.data(), ctx->overflow.p, nt * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    return HR_OK;
}

int overflow_redo(hr_ctx *ctx, Group &g, bool *relaunched)
{
    hipStream_t st = ctx->stream;
    const size_t nt = g.descs.size();
    *relaunched = false;
    std::vector<unsigned> &ovf = ctx->h_ovf, &cnt = ctx->h_cand_cnt;
    if (ovf.size() != nt) return HR_OK;  // (no fused track in the launch)
    const bool debug = std::getenv("HR_DEBUG") != nullptr;
    bool any = debug, waves = debug;
    for (unsigned o : ovf) { any |= o != 0; waves |= (o & 2u) != 0; }
    if (!any) return HR_OK;
    if (waves && !ctx->h_fitems.empty()) {  // which waves: their candidate counts carry kCandOverflowBit
        size_t n_lists = 0;
        for (const auto &run : ctx->h_runs) n_lists = std::max(n_lists, run.list_off + run.n_items * run.wpi);
        cnt.resize(n_lists);
        HIPCHK(hipMemcpyAsync(cnt.data(), ctx->cand_cnt.p, cnt.size() * sizeof(unsigned), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    std::vector<WorkItem> &items = ctx->h_ovf_items;
    items.clear();
    // [first, end) of track i in items of one wave each: 64 lanes x redo_lane_frames(step) positions
    // (chan0: the first channel of the pair the redo covers -- k_true_peak_rows takes it from the high half of item.pad)
    auto add_range = [&](size_t i, uint64_t first, uint64_t end, uint32_t chan0) {
        const uint64_t per_wave = 64ull * redo_lane_frames(static_cast<uint32_t>(ctx->rates_h[g.descs[i].rate_idx].swr_step));
        for (uint64_t n = first; n < end; n += per_wave)
            items.push_back({static_cast<uint32_t>(i), static_cast<uint32_t>(std::min(per_wave, end - n)) | chan0 << 16, n});
    };
    size_t fmt_end[3];  // f32 items first, then s16, then s24
    for (int f = 0; f < 3; f++) {
        // whole tracks: everything the fused kernel screened
        for (size_t i = 0; i < nt; i++)
            if (ctx->h_ftp[i] && (ovf[i] & 1u) && g.fmts[i] == f)
                for (uint32_t c0 = 0; c0 + 1 < std::max<uint32_t>(g.descs[i].ch, 2); c0 += 2) add_range(i, 0, ctx->h_fend[i], c0);
        // single waves (their rows start at multiples of the resampler's input step: run_measure)
        for (const auto &run : ctx->h_runs)
            for (size_t w = 0; w < run.n_items * run.wpi && run.list_off + w < cnt.size(); w++) {
                const FusedItem &it = ctx->h_fitems[run.item_off + w / run.wpi];
                if (!(cnt[run.list_off + w] & kCandOverflowBit) || !it.n_rows || !ctx->h_ftp[it.track] || (ovf[it.track] & 1u) || g.fmts[it.track] != f) continue;
                if (it.row_stride > 1) {  // (rows with shifted piece grids: the wave's hops are row_stride apart, each redone on its own)
                    for (uint32_t rr = 0; rr < it.n_rows; rr++) {
                        const uint64_t f0 = it.frame0 + static_cast<uint64_t>(rr) * it.row_stride * it.row_len;
                        add_range(it.track, f0, std::min<uint64_t>(f0 + it.row_len, ctx->h_fend[it.track]), 0);
                    }
                    continue;
                }
                add_range(it.track, it.frame0, std::min<uint64_t>(it.frame0 + static_cast<uint64_t>(it.n_rows) * it.row_len, ctx->h_fend[it.track]),
                          static_cast<uint32_t>(2 * (w % run.wpi)));
            }
        fmt_end[f] = items.size();
    }
    if (std::getenv("HR_DEBUG")) {
        unsigned long long tot = 0, mx = 0, full = 0, novf = 0;
        for (unsigned c : cnt) { const unsigned n = c & ~kCandOverflowBit; tot += n; mx = std::max<unsigned long long>(mx, n); full += (c & kCandOverflowBit) != 0; }
        for (unsigned o : ovf) novf += (o & 1u) != 0;
        std::fprintf(stderr, "headroom_hip: fused waves %zu, candidates total %llu (mean %.1f, max %llu, overflowed lists %llu), tracks beyond the f16 range %llu of %zu, redo items %zu\n",
                     cnt.size(), tot, cnt.empty() ? 0.0 : double(tot) / cnt.size(), mx, full, novf, nt, items.size());
    }
    if (items.empty()) return HR_OK;
    int rc = upload(ctx->ovf_items, items.data(), items.size() * sizeof(WorkItem), st);
    if (rc) return rc;
    for (int f = 0; f < 3; f++) {
        const size_t b = f ? fmt_end[f - 1] : 0;
        launch_true_peak_rows(f, ctx->tracks_d.as<TrackDesc>(), ctx->rates_d.as<RateDev>(), ctx->ovf_items.as<WorkItem>() + b,
                              static_cast<uint32_t>(fmt_end[f] - b), ctx->peaks.as<unsigned long long>(), st);
    }
    HIPCHK(hipGetLastError());
    *relaunched = true;
    return HR_OK;
}

// (both halves at once, for callers that have no results to read in the same breath; synchronises the stream)
int redo_overflowed_peaks(hr_ctx *ctx, Group &g)
{
    int rc = overflow_flags_fetch(ctx, g);
    if (rc) return rc;
    if (ctx->h_ovf.empty()) return HR_OK;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    bool relaunched = false;
    return overflow_redo(ctx, g, &relaunched);
}

// Every f32 / s16 / s24 track that is not already stereo f32 is split on the device into stereo f32 planes (exactly)
// so that it takes the fused kernel: mono, 5.1, 7.1, integer stereo.  The generic kernels, one thread per hop, are two
// orders of magnitude slower.  s32 / f64 samples do not fit f32 and stay generic.
// Stereo s16 / s24 is read directly by the fused kernel's full pass (K-weighting + true-peak screen): needs a true-peak mode
// with a tile table at this rate and a 16-byte aligned buffer.  (Other cases are widened into f32 planes first.)
bool fused_s16(const hr_ctx *ctx, const hr_track &t, const void *dev)
{
    if ((t.fmt != HR_FMT_S16 && t.fmt != HR_FMT_S24) || t.channels != 2 || ctx->no_s16_direct) return false;
    if (t.fmt == HR_FMT_S24 && (t.sample_rate / 10) % 2) return false;  // its rows must start on a dword
    if (ctx->cfg.tp_mode != HR_TP_SWR192K && ctx->cfg.tp_mode != HR_TP_BS1770) return false;
    const RateTables *rt = rate_tables(t.sample_rate);
    if (!rt) return false;
    const bool has_tiles = ctx->cfg.tp_mode == HR_TP_BS1770 ? !rt->bs_atab.empty() : !rt->tp_atab.empty();
    return has_tiles && (reinterpret_cast<uintptr_t>(dev) & 15) == 0;
}

// A 5.1 / 7.1 track of f32 / s16 / s24 samples is read directly by the fused kernel (hr_kernels_fused_mc.hip: a workgroup of
// channels / 2 waves shares the rows' pieces) when its rate's tile shape has a build, its rows start on dwords and its buffer on
// a 16-byte boundary.  Everything else of more than two channels goes through stereo planes (splits()), as all of it did
// until round 3; hr_config.engine = 5 keeps it that way (A/B and tests).
// A 5.1 track that fused_mc() takes goes to the four-wave workgroup (hr_fused_core.hpp, ROLE: three waves screen a channel pair
// each and K-weight its first channel, a fourth wave K-weights two of the second channels) when one of the second channels
// (1, 3, 5) carries no loudness weight -- the LFE of the default layout -- so that nobody has to K-weight it: its records are
// written as zeros, which the weight multiplies either way.  Two such workgroups fill the eight wave slots of a CU, where the
// three-wave workgroup leaves two empty.  hr_config.engine = 6 keeps the three-wave workgroup (A/B and tests).
bool fused_mc4(const hr_ctx *ctx, const TrackDesc &d, int tp_class)
{
    if (ctx->cfg.engine == 6 || d.ch != 6 || !fused_mc4_supported(d.fmt, tp_class)) return false;
    return d.w[1] == 0.0f || d.w[3] == 0.0f || d.w[5] == 0.0f;
}

bool fused_mc(const hr_ctx *ctx, const hr_track &t, const void *dev)
{
    if (t.channels != 6 && t.channels != 8) return false;
    if (t.fmt != HR_FMT_F32 && t.fmt != HR_FMT_S16 && t.fmt != HR_FMT_S24) return false;
    if (ctx->cfg.engine == 1 || ctx->cfg.engine == 5 || ctx->split || (ctx->no_s16_direct && t.fmt != HR_FMT_F32)) return false;
    if (ctx->cfg.tp_mode != HR_TP_SWR192K && ctx->cfg.tp_mode != HR_TP_BS1770) return false;
    const uint64_t hop = t.sample_rate / 10, fb = fmt_bytes(t.fmt) * t.channels;
    if (hop < 64 || t.frames / hop < 2 || hop * fb * 64 >= 0xffffffffull || (hop * fb) % 4) return false;
    const RateTables *rt = rate_tables(t.sample_rate);
    if (!rt) return false;
    const bool bs = ctx->cfg.tp_mode == HR_TP_BS1770;
    if (bs ? rt->bs_atab.empty() : rt->tp_atab.empty()) return false;
    const int mt = bs ? rt->bs_mt : rt->tp_mt, last_rows = (bs ? rt->bs_max_out : rt->tp_max_out) - 32 * (mt - 1);
    return fused_mc_supported(t.channels, t.fmt, fused_tp_class(mt, last_rows)) && (reinterpret_cast<uintptr_t>(dev) & 15) == 0;
}

// A track in device memory whose address keeps it off the fast kernels (stereo f32 needs 16-byte alignment for the
// piece DMA, integer PCM 4-byte alignment for the plane split) is copied, device to device, into the group's staging
// buffer like a host track: one extra read and write of its bytes instead of the generic f64 kernels, which are two
// orders of magnitude slower per sample.
bool restaged(const hr_ctx *ctx, const hr_track &t)
{
    if (!(t.flags & HR_TRACK_DEVICE_PCM) || ctx->no_alloc || ctx->cfg.engine == 1) return false;
    const uint32_t hop = t.sample_rate / 10;
    if (hop < 64 || t.frames / hop < 2 || hop * 8ull * 64 >= 0xffffffffull) return false;
    const uintptr_t a = reinterpret_cast<uintptr_t>(t.pcm);
    if (t.fmt == HR_FMT_F32) return t.channels == 2 && (a & 15) != 0;
    if (t.fmt == HR_FMT_S16 || t.fmt == HR_FMT_S24) return (a & 3) != 0;
    return false;
}
bool on_device(const hr_ctx *ctx, const hr_track &t) { return (t.flags & HR_TRACK_DEVICE_PCM) && !restaged(ctx, t); }

bool splits(const hr_ctx *ctx, const hr_track &t, const void *dev_or_null = nullptr, bool need_peaks = true)
{
    const uint32_t hop = t.sample_rate / 10;
    // (the direct read of 5.1 / 7.1 is a build of the full pass only: hr_block_energies, which asks for no peaks, keeps the planes)
    if (need_peaks && fused_mc(ctx, t, on_device(ctx, t) ? t.pcm : dev_or_null)) return false;
    if (ctx->no_alloc) return false;  // the planes would need a PCM-sized buffer: such tracks take the generic kernels there
    if (ctx->cfg.engine == 1 || hop < 64 || t.frames / hop < 2 || hop * 8ull * 64 >= 0xffffffffull) return false;
    if (t.fmt != HR_FMT_F32 && t.fmt != HR_FMT_S16 && t.fmt != HR_FMT_S24) return false;
    if (t.fmt == HR_FMT_F32 && t.channels == 2) return false;  // the fused kernel reads it as it is
    const void *p = on_device(ctx, t) ? t.pcm : dev_or_null;  // staged copies are 256-byte aligned
    if (fused_s16(ctx, t, p)) return false;                     // likewise
    return (reinterpret_cast<uintptr_t>(p) & 3) == 0;
}
size_t plane_floats(const hr_track &t) { return (t.frames * 2 + 63) / 64 * 64; }
size_t split_bytes(const hr_track &t) { return plane_floats(t) * sizeof(float) * ((t.channels + 1) / 2); }

// Adds one track to the group (PCM at `dev` on the device); a split track becomes a parent descriptor followed by its
// plane descriptors in `plane_mem` (split_bytes(t) of 256-byte aligned device memory).  Returns the index of the
// descriptor that carries the track's results, or a negative error.
int add_track(hr_ctx *ctx, Group &g, const hr_track &t, const void *dev, float *plane_mem, bool do_split)
{
    const int ri = rate_index(ctx, t.sample_rate);
    if (ri < 0) return ri;
    const int own = static_cast<int>(g.descs.size());
    add_desc(g, t, dev, ri, true, true, false, t.frames, ctx->cfg.tp_mode);
    if (!do_split) return own;
    const size_t pf = plane_floats(t);
    // integer tracks keep their sample format in the planes where the fused kernel reads stereo s16 / s24 directly
    hr_track as_stereo = t;
    as_stereo.channels = 2;
    const bool keep = fused_s16(ctx, as_stereo, plane_mem);
    launch_split_planes(t.fmt, dev, plane_mem, t.frames, t.channels, pf, keep, ctx->stream);
    g.descs[own].role = 1;
    g.descs[own].first_plane = static_cast<uint32_t>(own + 1);
    g.parents.push_back(static_cast<uint32_t>(own));
    hr_track pt = t;
    pt.channels = 2;
    pt.fmt = keep ? t.fmt : HR_FMT_F32;
    pt.ch_weight = nullptr;
    for (int p = 0; p < (t.channels + 1) / 2; p++) {
        add_desc(g, pt, plane_mem + p * pf, ri, true, true, false, t.frames, ctx->cfg.tp_mode);
        g.descs.back().role = 2;
    }
    return own;
}

// peak bit patterns of the track whose results live in descriptor `di`
void track_peaks(const Group &g, const std::vector<unsigned long long> &pk, int di, unsigned long long *out)
{
    const TrackDesc &d = g.descs[di];
    for (int c = 0; c < kMaxCh; c++) out[c] = 0;
    for (int c = 0; c < d.ch; c++)
        out[c] = d.role == 1 ? pk[(d.first_plane + c / 2) * kMaxCh + (c & 1)] : pk[static_cast<size_t>(di) * kMaxCh + c];
}

// One launch group of hr_analyze_batch in three phases, so that the staging copies of the NEXT group run beside the kernels of
// this one (pinned H2D on a copy stream, two staging buffers):
//   group_stage   the group's host-resident (and misaligned device) tracks into staging buffer `half`, on the copy stream
//   group_launch  descriptors, plane splits, K-weighting + true peak + gating, and the copies of the results, on the main stream
//                 (which first waits for the staging copies)
//   group_finish  the one synchronisation, the exact redo where the screen asked for it, the results
// The workspace behind a launch is the context's: one group is launched and finished before the next is launched; only the
// staging of the next one overlaps.
struct GroupJob {
    std::vector<size_t> idx;   // the group's tracks (indices into the caller's array)
    std::vector<size_t> off;   // per track: offset in the staging buffer
    size_t stage_need = 0;
    int half = 0;
    Group g;
    std::vector<int> desc_of;
    std::vector<TrackOut> outs;
    std::vector<unsigned long long> pk;
};

DevBuf &stage_buf(hr_ctx *ctx, int half) { return half ? ctx->stage_b : ctx->stage; }

int group_stage(hr_ctx *ctx, const hr_track *tracks, GroupJob &job)
{
    job.off.assign(job.idx.size(), 0);
    job.stage_need = 0;
    for (size_t k = 0; k < job.idx.size(); k++) {
        const hr_track &t = tracks[job.idx[k]];
        if (on_device(ctx, t)) continue;
        job.off[k] = job.stage_need;
        job.stage_need += (t.frames * t.channels * fmt_bytes(t.fmt) + 255) / 256 * 256;
    }
    if (!job.stage_need) return HR_OK;
    DevBuf &sb = stage_buf(ctx, job.half);
    if (!sb.ensure(job.stage_need)) return HR_ERR_OOM;
    hipStream_t cs = ctx->copy ? ctx->copy : ctx->stream;
    for (size_t k = 0; k < job.idx.size(); k++) {
        const hr_track &t = tracks[job.idx[k]];
        if (on_device(ctx, t)) continue;
        const size_t bytes = t.frames * t.channels * fmt_bytes(t.fmt);
        if (bytes)
            HIPCHK(hipMemcpyAsync(static_cast<char *>(sb.p) + job.off[k], t.pcm, bytes,
                                  (t.flags & HR_TRACK_DEVICE_PCM) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, cs));
    }
    if (cs != ctx->stream) HIPCHK(hipEventRecord(ctx->ev_staged[job.half], cs));
    return HR_OK;
}

int group_launch(hr_ctx *ctx, const hr_track *tracks, GroupJob &job)
{
    hipStream_t st = ctx->stream;
    const std::vector<size_t> &idx = job.idx;
    if (job.stage_need && ctx->copy) HIPCHK(hipStreamWaitEvent(st, ctx->ev_staged[job.half], 0));
    DevBuf &sb = stage_buf(ctx, job.half);
    size_t widen_need = 0;
    std::vector<size_t> woff(idx.size(), 0);
    std::vector<uint8_t> sp(idx.size(), 0);
    for (size_t k = 0; k < idx.size(); k++)
        if ((sp[k] = splits(ctx, tracks[idx[k]]))) { woff[k] = widen_need; widen_need += split_bytes(tracks[idx[k]]); }
    if (widen_need && !ctx->widened.ensure(widen_need)) return HR_ERR_OOM;
    job.desc_of.assign(idx.size(), 0);
    for (size_t k = 0; k < idx.size(); k++) {
        const hr_track &t = tracks[idx[k]];
        const void *dev = on_device(ctx, t) ? t.pcm : static_cast<char *>(sb.p) + job.off[k];
        job.desc_of[k] = add_track(ctx, job.g, t, dev, reinterpret_cast<float *>(static_cast<char *>(ctx->widened.p) + woff[k]), sp[k] != 0);
        if (job.desc_of[k] < 0) return job.desc_of[k];
    }
    ctx->ht("add tracks");
    int rc = run_measure(ctx, job.g, true);
    if (rc) return rc;
    ctx->ht("rest of run_measure");
    rc = run_gating(ctx, job.g, true);
    if (rc) return rc;
    ctx->ht("run_gating");
    HIPCHK(hipEventRecord(ctx->ev[1], st));
    // the screen's overflow flags come back with the results: one synchronisation in the common case (no flag set)
    rc = overflow_flags_fetch(ctx, job.g);
    if (rc) return rc;
    const size_t nt = job.g.descs.size();
    job.outs.resize(nt);
    job.pk.resize(nt * kMaxCh);
    HIPCHK(hipMemcpyAsync(job.outs.data(), ctx->outs.p, nt * sizeof(TrackOut), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(job.pk.data(), ctx->peaks.p, nt * kMaxCh * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    ctx->ht("queue result copies");
    return HR_OK;
}

int group_finish(hr_ctx *ctx, const hr_track *tracks, GroupJob &job, hr_result *out)
{
    hipStream_t st = ctx->stream;
    const size_t nt = job.g.descs.size();
    HIPCHK(hipStreamSynchronize(st));
    ctx->ht("wait for the GPU");
    bool relaunched = false;
    const int rc = overflow_redo(ctx, job.g, &relaunched);
    if (rc) return rc;
    if (relaunched) {
        HIPCHK(hipMemcpyAsync(job.pk.data(), ctx->peaks.p, nt * kMaxCh * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    float ms = 0;
    if (hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]) == hipSuccess) ctx->last_total_ms += ms;
    if (hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]) == hipSuccess) ctx->last_main_ms += ms;
    for (size_t k = 0; k < job.idx.size(); k++) {
        unsigned long long tpk[kMaxCh];
        track_peaks(job.g, job.pk, job.desc_of[k], tpk);
        fill_result(ctx, job.outs[job.desc_of[k]], tpk, tracks[job.idx[k]].channels, &out[job.idx[k]]);
    }
    ctx->ht("results");
    return HR_OK;
}

// one group, start to end (callers with a single group: the no-alloc entry point, the loudnorm first pass)
int analyze_group(hr_ctx *ctx, const hr_track *tracks, const std::vector<size_t> &idx, hr_result *out)
{
    GroupJob job;
    job.idx = idx;
    int rc = group_stage(ctx, tracks, job);
    if (!rc) rc = group_launch(ctx, tracks, job);
    if (!rc) rc = group_finish(ctx, tracks, job, out);
    if (rc) {  // (whatever was queued must not outlive the job's host buffers)
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->copy) (void)hipStreamSynchronize(ctx->copy);
    }
    return rc;
}

}  // namespace

extern "C" {

int hr_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int hr_init(const hr_config *cfg, hr_ctx **out)
try {
    if (!out) return HR_ERR_INVALID;
    *out = nullptr;
    hr_config c{};
    if (cfg) c = *cfg;
    if (c.gate_mode < 0 || c.gate_mode > 2 || c.tp_mode < 0 || c.tp_mode > 2 || c.engine < 0 || c.engine > 7) return HR_ERR_INVALID;
    const int n = hr_device_count();
    if (n <= 0) return HR_ERR_NO_DEVICE;
    if (c.device < 0 || c.device >= n) return HR_ERR_INVALID;
    HIPCHK(hipSetDevice(c.device));
    if (c.device < kMaxDevices) {
        // address of this device's copy of the scratch array (each GPU holds its own, at its own address)
        std::lock_guard<std::mutex> lk(g_scratch_init_mu);
        if (!g_scratch_ptr[c.device]) {
            void *sp = nullptr;
            HIPCHK(hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch)));
            g_scratch_ptr[c.device] = sp;
        }
    }
    hr_ctx *ctx = new hr_ctx();
    ctx->cfg = c;
    if (!ctx->cfg.stage_bytes) ctx->cfg.stage_bytes = 1ull << 30;
#ifdef HR_EXPERIMENTS  // timing switches of the experiments library (make exp); the shipped library reads none of them
    if (const char *ab = std::getenv("HR_ABLATE")) ctx->ablate = std::atoi(ab);
    if (const char *e = std::getenv("HR_HOST_TIMING")) ctx->host_timing = std::atoi(e) != 0;
    if (const char *sp = std::getenv("HR_FUSED_SPLIT")) ctx->split = std::atoi(sp) != 0;
    if (const char *se = std::getenv("HR_SERIAL_TAIL")) ctx->serial_tail = std::atoi(se) != 0;
#endif
    if (const char *ns = std::getenv("HR_NO_S16_DIRECT")) ctx->no_s16_direct = std::atoi(ns) != 0;
    if (c.stream) ctx->stream = static_cast<hipStream_t>(c.stream);
    else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return HR_ERR_HIP; }
        ctx->own_stream = true;
    }
    for (auto &e : ctx->ev)
        if (hipEventCreate(&e) != hipSuccess) { hr_destroy(ctx); return HR_ERR_HIP; }
    if (hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->copy, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_staged[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_staged[1], hipEventDisableTiming) != hipSuccess) { hr_destroy(ctx); return HR_ERR_HIP; }
    int rc = upload(ctx->ebu_energy, ebur128_bin_energy().data(), kEbuBins * sizeof(double), ctx->stream);
    if (!rc) rc = upload(ctx->ebu_edges, ebur128_bin_edges().data(), 3 * kEbuBins * sizeof(double), ctx->stream);
    if (!rc) rc = upload(ctx->ln_energy, loudnorm_bin_energy().data(), kLnBins * sizeof(double), ctx->stream);
    if (!rc) rc = upload(ctx->ln_bounds, loudnorm_bin_bounds().data(), (kLnBins + 1) * sizeof(double), ctx->stream);
    double unit[kSwrTaps] = {0};
    unit[15] = 1.0;  // the centre tap: a one-phase "bank" that passes 192 kHz input through
    if (!rc) rc = upload(ctx->unit_tap, unit, sizeof unit, ctx->stream);
    if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = HR_ERR_HIP;
    if (rc) { hr_destroy(ctx); return rc; }
    *out = ctx;
    return HR_OK;
} HR_CATCH_ALL

void hr_destroy(hr_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->cfg.device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (DevBuf *b : {&ctx->rates_d, &ctx->ebu_energy, &ctx->ebu_edges, &ctx->ln_energy, &ctx->ln_bounds, &ctx->unit_tap, &ctx->tracks_d,
                      &ctx->widened, &ctx->resampled, &ctx->parents_d, &ctx->counters, &ctx->gate_items, &ctx->gate_parts, &ctx->st_hist, &ctx->items_d, &ctx->comb_items_d, &ctx->fitems_d, &ctx->cands, &ctx->cand_cnt, &ctx->tpb, &ctx->overflow, &ctx->ovf_items, &ctx->recs, &ctx->hop_sums, &ctx->blk_powers, &ctx->hist, &ctx->outs, &ctx->peaks,
                      &ctx->stage, &ctx->halos, &ctx->fp_src_d, &ctx->fp_items_d, &ctx->shard_red, &ctx->shard_edge_buf, &ctx->shard_left_buf,
                      &ctx->shard_part_buf, &ctx->shard_gather_buf, &ctx->stage_b})
        b->release();
    for (void *p : ctx->rate_allocs) (void)hipFree(p);
    delete ctx->shard_group;
    for (auto &e : ctx->ev) if (e) (void)hipEventDestroy(e);
    if (ctx->side) { (void)hipStreamSynchronize(ctx->side); (void)hipStreamDestroy(ctx->side); }
    if (ctx->copy) { (void)hipStreamSynchronize(ctx->copy); (void)hipStreamDestroy(ctx->copy); }
    for (auto &e : ctx->ev_staged) if (e) (void)hipEventDestroy(e);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

void *hr_alloc_pinned(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}

void hr_free_pinned(void *p) { if (p) (void)hipHostFree(p); }

int hr_analyze_batch(hr_ctx *ctx, const hr_track *tracks, size_t n, hr_result *out)
try {
    if (!ctx || (n && (!tracks || !out))) return HR_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->shard_open = false;  // the workspace an open time shard left behind is about to be reused: its pass 2 must fail
    HIPCHK(hipSetDevice(ctx->cfg.device));
    ctx->last_total_ms = ctx->last_main_ms = 0;
    ctx->ht_marks.clear();
    ctx->ht_last = std::chrono::steady_clock::now();
    const double ninf = -std::numeric_limits<double>::infinity();
    std::vector<size_t> cur;
    size_t cur_bytes = 0, cur_widened = 0;
    // stereo planes of split tracks per group: 16 GiB by default, and up to a third of the device memory that is free (at
    // most 96 GiB) -- a launch that fills several rounds of the GPU's wave slots wastes less on its last, partial round
    size_t widen_limit = 16 * ctx->cfg.stage_bytes;
    bool widen_probed = false;
    std::vector<std::vector<size_t>> groups;
    auto flush = [&]() {
        cur_widened = 0;
        if (!cur.empty()) groups.push_back(cur);
        cur.clear();
        cur_bytes = 0;
    };
    for (size_t i = 0; i < n; i++) {
        std::memset(&out[i], 0, sizeof out[i]);
        out[i].integrated_lufs = out[i].true_peak_dbtp = out[i].rel_threshold_lufs = ninf;
        out[i].loudness_range_lu = out[i].lra_low_lufs = out[i].lra_high_lufs = out[i].lra_threshold_lufs =
            out[i].max_momentary_lufs = out[i].max_shortterm_lufs = std::numeric_limits<double>::quiet_NaN();
        const int v = validate_track(ctx, tracks[i]);
        if (v) { out[i].status = v; continue; }
        const size_t bytes = on_device(ctx, tracks[i])
                                 ? 0 : tracks[i].frames * tracks[i].channels * fmt_bytes(tracks[i].fmt) + 256;
        const size_t wbytes = splits(ctx, tracks[i]) ? split_bytes(tracks[i]) : 0;
        if (wbytes && !widen_probed) {
            widen_probed = true;
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
                widen_limit = std::max(widen_limit, std::min<size_t>(96ull << 30, (free_b + ctx->widened.cap) / 3));
            else (void)hipGetLastError();
        }
        if (!cur.empty() && (cur_bytes + bytes > ctx->cfg.stage_bytes || cur_widened + wbytes > widen_limit)) flush();
        cur.push_back(i);
        cur_bytes += bytes;
        cur_widened += wbytes;
    }
    flush();
    ctx->ht("validate + group");
    // The groups, pipelined: while group k is measured, the staging copies of group k + 1 run on the copy stream into the other
    // staging buffer (group k - 1, which used it, has been finished).  A failing group marks its own tracks; HIP errors are
    // fatal for the call.
    int fatal = HR_OK;
    // (an exception on the way -- std::bad_alloc from a host container -- must not free the jobs' result buffers under copies
    // that are still in flight: the streams are drained before the jobs go)
    struct Drain {
        hr_ctx *c;
        bool armed = true;
        ~Drain()
        {
            if (!armed) return;
            (void)hipStreamSynchronize(c->stream);
            if (c->copy) (void)hipStreamSynchronize(c->copy);
        }
    } drain{ctx};
    std::vector<GroupJob> jobs(groups.size());
    std::vector<int> staged(groups.size(), HR_OK);
    auto fail = [&](size_t k, int rc) {
        for (size_t i : groups[k]) out[i].status = rc;
        if (rc == HR_ERR_HIP) fatal = rc;
        (void)hipStreamSynchronize(ctx->stream);  // (nothing queued for this group may outlive its host buffers)
        if (ctx->copy) (void)hipStreamSynchronize(ctx->copy);
    };
    for (size_t k = 0; k < groups.size(); k++) { jobs[k].idx = groups[k]; jobs[k].half = static_cast<int>(k & 1); }
    if (!groups.empty()) staged[0] = group_stage(ctx, tracks, jobs[0]);
    size_t k_done = 0;  // groups that have had their turn
    for (size_t k = 0; k < groups.size() && !fatal; k++, k_done++) {
        int rc = staged[k];
        if (!rc) rc = group_launch(ctx, tracks, jobs[k]);
        if (k + 1 < groups.size()) staged[k + 1] = group_stage(ctx, tracks, jobs[k + 1]);
        if (!rc) rc = group_finish(ctx, tracks, jobs[k], out);
        if (rc) fail(k, rc);
        jobs[k] = GroupJob{};  // (its descriptors and result copies are not needed any more)
    }
    for (size_t k = k_done; k < groups.size(); k++)  // (after a fatal error: the groups that never ran)
        for (size_t i : groups[k]) out[i].status = fatal;
    drain.armed = fatal != HR_OK;
    if (ctx->host_timing) {
        std::fprintf(stderr, "headroom_hip host timing (ms), %zu tracks:", n);
        for (const auto &m : ctx->ht_marks) std::fprintf(stderr, " [%s %.3f]", m.first, m.second);
        std::fprintf(stderr, "\n");
    }
    return fatal;
} HR_CATCH_ALL

int hr_analyze_device_noalloc(hr_ctx *ctx, const hr_track *track, hr_result *out)
try {
    if (!ctx || !track || !out) return HR_ERR_INVALID;
    std::memset(out, 0, sizeof *out);
    out->integrated_lufs = out->true_peak_dbtp = out->rel_threshold_lufs = -std::numeric_limits<double>::infinity();
    out->loudness_range_lu = out->lra_low_lufs = out->lra_high_lufs = out->lra_threshold_lufs = out->max_momentary_lufs =
        out->max_shortterm_lufs = std::numeric_limits<double>::quiet_NaN();
    auto fail = [&](int rc) { out->status = rc; return rc; };  // (a rejected call says so in the result record too)
    if (!(track->flags & HR_TRACK_DEVICE_PCM)) return fail(HR_ERR_INVALID);  // a host buffer would need a staging allocation
    std::lock_guard<std::mutex> lk(ctx->mu);
    const int v = validate_track(ctx, *track);
    if (v) return fail(v);
    const uint32_t hop = track->sample_rate / 10;
    const uint64_t hops = track->frames / hop;
    if (hops * track->channels > HR_NOALLOC_MAX_HOP_CHANNELS) {
        std::fprintf(stderr, "headroom_hip: hr_analyze_device_noalloc: %llu hops x %u channels exceed the limit of %u (the %zu-byte "
                             "static workspace); use hr_analyze_batch\n",
                     static_cast<unsigned long long>(hops), static_cast<unsigned>(track->channels),
                     static_cast<unsigned>(HR_NOALLOC_MAX_HOP_CHANNELS), kScratchBytes);
        return fail(HR_ERR_TOO_LARGE);
    }
    // Everything the call will carve from the array, bounded from above BEFORE anything is queued: records, hop sums, block
    // powers, the two histograms, candidate lists (4 KiB per fused wave), and the work-item lists, which grow with the
    // FRAMES (one item per 8192 positions for the generic peak kernels and, worst case, for the exact redo of every fused
    // wave), not with the hops -- a mono or multichannel track at 384 / 768 kHz inside the hop limit is far longer in frames.
    {
        // (waves: the items of rows with shifted piece grids come in workgroups of four per offset, up to 24 of them padding;
        // per_span: the exact redo of such an item is one work item per hop)
        const uint64_t ch = track->channels, waves = (hops / 64 + 8 + 32) * std::max<uint64_t>(1, ch / 2),
                       per_span = track->frames / kPeakSpan + hops / 256 + 16 + hops / 2;
        const uint64_t need = (hops + kPreRec + 8) * ch * sizeof(HopRec) + (hops + kPreHop + 8) * ch * sizeof(double) + (hops + 1) * sizeof(double) +
                              2 * kEbuBins * sizeof(uint32_t) + waves * (fused_cand_bytes(1) + sizeof(FusedItem) + sizeof(unsigned)) +
                              (2 * per_span + 2 * ch) * sizeof(WorkItem) + (hops * ch / 256 + 2) * sizeof(WorkItem) +
                              (hops / kGateChunk + 2) * (sizeof(WorkItem) + sizeof(GatePart)) + sizeof(TrackDesc) + sizeof(TrackOut) + 4096 +
                              32 * 256 /* every buffer starts on a 256-byte boundary */;
        if (need > kScratchBytes) {
            std::fprintf(stderr, "headroom_hip: hr_analyze_device_noalloc: this track would need %llu bytes of workspace, the static array "
                                 "holds %zu (%llu frames x %u channels: the work-item lists grow with the frames); use hr_analyze_batch\n",
                         static_cast<unsigned long long>(need), kScratchBytes, static_cast<unsigned long long>(track->frames),
                         static_cast<unsigned>(track->channels));
            return fail(HR_ERR_TOO_LARGE);
        }
    }
    // the rate's device tables are allocated when a rate is first seen: that must have happened in an earlier, ordinary call
    bool resident = false;
    for (uint32_t fs : ctx->rate_fs) resident |= fs == track->sample_rate;
    if (!resident) return fail(HR_ERR_NOT_READY);
    const int dev = ctx->cfg.device;
    if (dev >= kMaxDevices || !g_scratch_ptr[dev]) return fail(HR_ERR_HIP);
    HIPCHK(hipSetDevice(dev));
    std::lock_guard<std::mutex> lk_dev(g_scratch_mu[dev]);
    Arena arena;
    arena.base = static_cast<char *>(g_scratch_ptr[dev]);
    arena.cap = kScratchBytes;
    // the context's own workspace steps aside for the call; every buffer is carved from the array instead
    struct Swap {  // (restores the context's buffers on every way out, an exception included)
        hr_ctx *ctx;
        std::vector<DevBuf *> ws;
        std::vector<DevBuf> saved;
        Swap(hr_ctx *c, Arena *a) : ctx(c), ws(c->workspace())
        {
            saved.reserve(ws.size());
            for (DevBuf *b : ws) saved.push_back(*b);
            for (DevBuf *b : ws) { *b = DevBuf{}; b->arena = a; }
            ctx->no_alloc = true;
        }
        ~Swap()
        {
            // Whatever way the call is left (an error after kernels were queued included), nothing may still be reading or
            // writing the array when the per-GPU lock is released: the next caller carves the same bytes.
            (void)hipStreamSynchronize(ctx->stream);
            if (ctx->side) (void)hipStreamSynchronize(ctx->side);
            ctx->no_alloc = false;
            for (size_t i = 0; i < ws.size(); i++) *ws[i] = saved[i];
        }
    } swap(ctx, &arena);
    ctx->shard_open = false;  // an open time shard's workspace is not the one pass 2 would find
    ctx->last_total_ms = ctx->last_main_ms = 0;
    int rc = analyze_group(ctx, track, std::vector<size_t>{0}, out);
    if (rc == HR_ERR_OOM) rc = HR_ERR_TOO_LARGE;  // the array was too small after all (the limit above is meant to exclude this)
    if (rc) out->status = rc;
    return rc;
} HR_CATCH_ALL

int hr_block_energies(hr_ctx *ctx, const hr_track *track, double *hop_sums, uint64_t cap_hops, double *block_powers,
                      uint64_t cap_blocks, uint64_t *n_hops, uint64_t *n_blocks)
try {
    if (!ctx || !track) return HR_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->shard_open = false;  // the workspace an open time shard left behind is about to be reused: its pass 2 must fail
    HIPCHK(hipSetDevice(ctx->cfg.device));
    const int v = validate_track(ctx, *track);
    if (v) return v;
    hipStream_t st = ctx->stream;
    Group g;
    const void *dev = track->pcm;
    const size_t bytes = track->frames * track->channels * fmt_bytes(track->fmt);
    if (!(track->flags & HR_TRACK_DEVICE_PCM)) {
        if (!ctx->stage.ensure(bytes + 256)) return HR_ERR_OOM;
        dev = ctx->stage.p;
        if (bytes) HIPCHK(hipMemcpyAsync(ctx->stage.p, track->pcm, bytes, hipMemcpyHostToDevice, st));
    }
    const bool sp = splits(ctx, *track, nullptr, false);
    if (sp && !ctx->widened.ensure(split_bytes(*track))) return HR_ERR_OOM;
    const int own = add_track(ctx, g, *track, dev, ctx->widened.as<float>(), sp);
    if (own < 0) return own;
    int rc = run_measure(ctx, g, false);
    if (rc) return rc;
    rc = run_gating(ctx, g, false);
    if (rc) return rc;
    const TrackDesc &d = g.descs[own];
    if (n_hops) *n_hops = d.n_hops;
    if (n_blocks) *n_blocks = d.n_blocks;
    if (hop_sums) {
        const uint64_t nh = std::min<uint64_t>(d.n_hops, cap_hops);
        if (nh) HIPCHK(hipMemcpyAsync(hop_sums, ctx->hop_sums.as<double>() + d.hop_base + kPreHop * d.ch,
                                      nh * d.ch * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    if (block_powers) {
        const uint64_t nb = std::min<uint64_t>(d.n_blocks, cap_blocks);
        if (nb) HIPCHK(hipMemcpyAsync(block_powers, ctx->blk_powers.as<double>() + d.blk_base, nb * sizeof(double),
                                      hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipStreamSynchronize(st));
    return HR_OK;
} HR_CATCH_ALL

static double raw_to_double(const unsigned char *p, int fmt)
{
    double x = 0;
    switch (fmt) {
    case HR_FMT_F32: { float f; std::memcpy(&f, p, 4); x = f; } break;
    case HR_FMT_S16: { int16_t q; std::memcpy(&q, p, 2); x = q * (1.0 / 32768.0); } break;
    case HR_FMT_S24: {
        const int32_t q = static_cast<int32_t>(static_cast<uint32_t>(p[0]) << 8 | static_cast<uint32_t>(p[1]) << 16 |
                                               static_cast<uint32_t>(p[2]) << 24);
        x = q * (1.0 / 2147483648.0);
    } break;
    case HR_FMT_S32: { int32_t q; std::memcpy(&q, p, 4); x = q * (1.0 / 2147483648.0); } break;
    default: std::memcpy(&x, p, 8); break;
    }
    return x;
}

int hr_shard_head(hr_ctx *ctx, const hr_track *shard, double *head)
try {
    if (!ctx || !shard || !head) return HR_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->cfg.device));
    const int v = validate_track(ctx, *shard);
    if (v) return v;
    const int C = shard->channels;
    const size_t fb = fmt_bytes(shard->fmt);
    const uint64_t tf = std::min<uint64_t>(shard->frames, kHalo);
    std::vector<unsigned char> raw(tf * C * fb + 8);
    if (tf) {
        if (shard->flags & HR_TRACK_DEVICE_PCM) HIPCHK(hipMemcpy(raw.data(), shard->pcm, tf * C * fb, hipMemcpyDeviceToHost));
        else std::memcpy(raw.data(), shard->pcm, tf * C * fb);
    }
    for (int k = 0; k < kHalo; k++)
        for (int c = 0; c < HR_MAX_CH; c++)
            head[k * HR_MAX_CH + c] = (static_cast<uint64_t>(k) < tf && c < C) ? raw_to_double(raw.data() + (k * C + c) * fb, shard->fmt) : 0.0;
    return HR_OK;
} HR_CATCH_ALL

// Pass 1 up to the point where the shard's records are on their way (nothing synchronised): staging, halos, descriptor,
// K-weighting of every hop from zero state + true peak + a first combine (the hop energies handed to the right neighbour
// do not depend on the left edge from hop 3 on).  Caller holds ctx->mu.  *dev_out: the shard's PCM on the device.
static int shard_pass1_core(hr_ctx *ctx, const hr_track *shard, const double *left_tail, const double *right_head, int is_first,
                            int is_last, uint64_t first_hop, const void **dev_out)
{
    ctx->shard_open = false;  // (set again by the caller, once the new shard's workspace is complete)
    ctx->shard_pass2_done = false;
    ctx->shard_device = false;
    HIPCHK(hipSetDevice(ctx->cfg.device));
    const int v = validate_track(ctx, *shard);
    if (v) return v;
    const uint32_t hop = shard->sample_rate / 10;
    const uint64_t nh = shard->frames / hop;
    if (!is_last && (shard->frames % hop || nh < kPreRec)) return HR_ERR_INVALID;
    hipStream_t st = ctx->stream;
    const int C = shard->channels;
    const size_t fb = fmt_bytes(shard->fmt);
    const size_t bytes = shard->frames * C * fb;
    const void *dev = shard->pcm;
    if (!(shard->flags & HR_TRACK_DEVICE_PCM)) {
        if (!ctx->stage.ensure(bytes + 256)) return HR_ERR_OOM;
        dev = ctx->stage.p;
        if (bytes) HIPCHK(hipMemcpyAsync(ctx->stage.p, shard->pcm, bytes, hipMemcpyHostToDevice, st));
    }
    if (!ctx->halos.ensure(2 * kHalo * kMaxCh * sizeof(double))) return HR_ERR_OOM;
    double *halo_l = ctx->halos.as<double>(), *halo_r = halo_l + kHalo * kMaxCh;
    // the ABI hands halos over as [16][HR_MAX_CH]; the kernels read them packed [16][channels]
    // (the packed copy lives in the context: the upload is asynchronous and the device-resident entry point does not wait)
    HIPCHK(hipStreamSynchronize(st));  // (nothing of an earlier call may still be reading h_halo; a no-op on an idle stream)
    std::vector<double> &packed = ctx->h_halo;
    packed.assign(2 * kHalo * C, 0.0);
    for (int k = 0; k < kHalo; k++)
        for (int c = 0; c < C; c++) {
            if (!is_first) packed[k * C + c] = left_tail[k * HR_MAX_CH + c];
            if (!is_last) packed[(kHalo + k) * C + c] = right_head[k * HR_MAX_CH + c];
        }
    HIPCHK(hipMemcpyAsync(halo_l, packed.data(), kHalo * C * sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(halo_r, packed.data() + kHalo * C, kHalo * C * sizeof(double), hipMemcpyHostToDevice, st));
    const int ri = rate_index(ctx, shard->sample_rate);
    if (ri < 0) return ri;
    delete ctx->shard_group;
    ctx->shard_group = new Group();
    Group &g = *ctx->shard_group;
    add_desc(g, *shard, dev, ri, is_first != 0, is_last != 0, !is_first, ~0ull, ctx->cfg.tp_mode);
    g.descs[0].left_tail = is_first ? nullptr : halo_l;
    g.descs[0].right_head = is_last ? nullptr : halo_r;
    g.descs[0].first_hop = first_hop;
    ctx->last_total_ms = ctx->last_main_ms = 0;
    // (the combine of pass 1 is for the hop energies handed to the right neighbour: from hop 3 on they do not depend
    // on the left edge, which only arrives for pass 2; pass 2 combines again)
    const int rc = run_measure(ctx, g, true, true);
    if (rc) return rc;
    *dev_out = dev;
    return HR_OK;
}

int hr_shard_pass1(hr_ctx *ctx, const hr_track *shard, const double *left_tail, const double *right_head, int is_first,
                   int is_last, uint64_t first_hop, hr_shard_edge *edge)
try {
    if (!ctx || !shard || !edge) return HR_ERR_INVALID;
    if ((!is_first && !left_tail) || (!is_last && !right_head)) return HR_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->mu);
    const void *dev = nullptr;
    int rc = shard_pass1_core(ctx, shard, left_tail, right_head, is_first, is_last, first_hop, &dev);
    if (rc) return rc;
    hipStream_t st = ctx->stream;
    const Group &g = *ctx->shard_group;
    const int C = shard->channels;
    const size_t fb = fmt_bytes(shard->fmt);
    const uint64_t nh = shard->frames / (shard->sample_rate / 10);
    // right edge: last 6 hop records and the last 16 frames
    std::memset(edge, 0, sizeof *edge);
    edge->n_hops = nh;
    const TrackDesc &d = g.descs[0];
    const uint64_t take = std::min<uint64_t>(nh, kPreRec);
    std::vector<HopRec> tmp(take * C);
    if (take)
        HIPCHK(hipMemcpyAsync(tmp.data(), ctx->recs.as<HopRec>() + d.rec_base + (kPreRec + nh - take) * C,
                              take * C * sizeof(HopRec), hipMemcpyDeviceToHost, st));
    const uint64_t tf = std::min<uint64_t>(shard->frames, kHalo);
    std::vector<unsigned char> raw(tf * C * fb + 8);
    if (tf)
        HIPCHK(hipMemcpyAsync(raw.data(), static_cast<const char *>(dev) + (shard->frames - tf) * C * fb, tf * C * fb,
                              hipMemcpyDeviceToHost, st));
    // the energies of the last 29 hops, for the right neighbour's first 3 s windows
    edge->range_ok = nh >= static_cast<uint64_t>(kPreHop + (is_first ? 0 : kPreBlk));
    std::vector<double> ht(static_cast<size_t>(kPreHop) * C, 0.0);
    if (edge->range_ok)
        HIPCHK(hipMemcpyAsync(ht.data(), ctx->hop_sums.as<double>() + d.hop_base + (kPreHop + nh - kPreHop) * C,
                              ht.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int k = 0; k < kPreHop; k++)
        for (int c = 0; c < C; c++) edge->hop_tail[k][c] = ht[static_cast<size_t>(k) * C + c];
    for (uint64_t k = 0; k < take; k++)
        for (int c = 0; c < C; c++) {
            const HopRec &r = tmp[k * C + c];
            double *o = edge->rec[kPreRec - take + k][c];
            o[0] = r.e0;
            for (int j = 0; j < 4; j++) { o[1 + j] = r.q[j]; o[5 + j] = r.s[j]; }
        }
    for (uint64_t k = 0; k < tf; k++)
        for (int c = 0; c < C; c++) {
            const double x = raw_to_double(raw.data() + (k * C + c) * fb, shard->fmt);
            edge->tail[kHalo - tf + k][c] = x;
        }
    ctx->shard_open = true;
    return HR_OK;
} HR_CATCH_ALL

// ---- the same exchange with the records resident on the device (round 3) ---------------------------------------------------
int hr_shard_pass1_device(hr_ctx *ctx, const hr_track *shard, const double *left_tail, const double *right_head, int is_first,
                          int is_last, uint64_t first_hop, void *edge_dev)
try {
    if (!ctx || !shard || !edge_dev) return HR_ERR_INVALID;
    if ((!is_first && !left_tail) || (!is_last && !right_head)) return HR_ERR_INVALID;
    if (ctx->cfg.gate_mode == HR_GATE_EXACT) return HR_ERR_INVALID;  // (two dependent reductions: the host-record entry points serve it)
    std::lock_guard<std::mutex> lk(ctx->mu);
    const void *dev = nullptr;
    const int rc = shard_pass1_core(ctx, shard, left_tail, right_head, is_first, is_last, first_hop, &dev);
    if (rc) return rc;
    launch_pack_edge(ctx->tracks_d.as<TrackDesc>(), ctx->recs.as<HopRec>(), ctx->hop_sums.as<double>(), is_first != 0, edge_dev, ctx->stream);
    HIPCHK(hipGetLastError());
    ctx->shard_open = true;
    ctx->shard_device = true;
    return HR_OK;
} HR_CATCH_ALL

int hr_shard_pass2_device(hr_ctx *ctx, const void *left_edge_dev, void *partial_dev)
try {
    if (!ctx || !partial_dev) return HR_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!ctx->shard_open || !ctx->shard_device || ctx->shard_pass2_done || !ctx->shard_group) return HR_ERR_INVALID;
    HIPCHK(hipSetDevice(ctx->cfg.device));
    hipStream_t st = ctx->stream;
    Group &g = *ctx->shard_group;
    const TrackDesc &d = g.descs[0];
    if ((d.blk_first < 0) != (left_edge_dev != nullptr)) return HR_ERR_INVALID;
    if (left_edge_dev)
        launch_unpack_edge(ctx->tracks_d.as<TrackDesc>(), ctx->recs.as<HopRec>(), ctx->hop_sums.as<double>(), left_edge_dev, st);
    launch_hop_combine(ctx->tracks_d.as<TrackDesc>(), ctx->rates_d.as<RateDev>(), ctx->comb_items_d.as<WorkItem>(),
                       static_cast<uint32_t>(ctx->comb_n), ctx->recs.as<HopRec>(), ctx->hop_sums.as<double>(), false, st);
    int rc = run_gating(ctx, g, false, true);
    if (rc) return rc;
    // the screen's flags travel with the image (its `reserved`): a shard whose exact redo is pending says so to every rank,
    // and hr_shard_finish_gathered tells the callers to run hr_shard_redo_device.  The host's copy of the flags is on its
    // way too; it is read only after a later synchronisation.
    rc = overflow_flags_fetch(ctx, g);
    if (rc) return rc;
    const int nb = ctx->cfg.gate_mode == HR_GATE_LOUDNORM ? kLnBins : kEbuBins;
    launch_pack_partial(ctx->tracks_d.as<TrackDesc>(), ctx->hist.as<uint32_t>(), ctx->st_hist.as<uint32_t>(), ctx->gate_parts.as<GatePart>(),
                        ctx->outs.as<TrackOut>(), ctx->peaks.as<unsigned long long>(), ctx->h_ovf.empty() ? nullptr : ctx->overflow.as<unsigned>(),
                        left_edge_dev, nb, partial_dev, st);
    HIPCHK(hipGetLastError());
    ctx->shard_pass2_done = true;  // (the shard stays open: hr_shard_redo_device may still need its workspace)
    return HR_OK;
} HR_CATCH_ALL

int hr_shard_redo_device(hr_ctx *ctx, void *partial_dev)
try {
    if (!ctx || !partial_dev) return HR_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!ctx->shard_open || !ctx->shard_device || !ctx->shard_pass2_done || !ctx->shard_group) return HR_ERR_INVALID;
    HIPCHK(hipSetDevice(ctx->cfg.device));
    hipStream_t st = ctx->stream;
    Group &g = *ctx->shard_group;
    // (hr_shard_finish_gathered put its stand-in track where the shard's descriptor was)
    int rc = upload(ctx->tracks_d, g.descs.data(), g.descs.size() * sizeof(TrackDesc), st);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(st));  // the flags pass 2 asked for are in h_ovf now
    bool relaunched = false;
    rc = overflow_redo(ctx, g, &relaunched);
    if (rc) return rc;
    launch_patch_partial_peaks(ctx->tracks_d.as<TrackDesc>(), ctx->peaks.as<unsigned long long>(), partial_dev, st);
    HIPCHK(hipGetLastError());
    ctx->h_ovf.clear();  // (a second redo finds nothing to do)
    return HR_OK;
} HR_CATCH_ALL

int hr_shard_finish_gathered(hr_ctx *ctx, const void *partials_dev, int n, int channels, hr_result *out)
try {
    if (!ctx || !partials_dev || n < 1 || !out || channels < 1 || channels > HR_MAX_CH) return HR_ERR_INVALID;
    if (ctx->cfg.gate_mode == HR_GATE_EXACT) return HR_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->cfg.device));
    hipStream_t st = ctx->stream;
    const size_t hist_bytes = kEbuBins * sizeof(uint32_t);
    if (!ctx->hist.ensure(hist_bytes) || !ctx->st_hist.ensure(hist_bytes) || !ctx->gate_parts.ensure(sizeof(GatePart)) ||
        !ctx->outs.ensure(sizeof(TrackOut)) || !ctx->tracks_d.ensure(sizeof(TrackDesc)) || !ctx->blk_powers.ensure(sizeof(double)) ||
        !ctx->hop_sums.ensure(sizeof(double)) || !ctx->shard_red.ensure(sizeof(ShardReduced)))
        return HR_ERR_OOM;
    const int nb = ctx->cfg.gate_mode == HR_GATE_LOUDNORM ? kLnBins : kEbuBins;
    launch_reduce_partials(partials_dev, n, nb, ctx->hist.as<uint32_t>(), ctx->st_hist.as<uint32_t>(), ctx->gate_parts.as<GatePart>(),
                           ctx->tracks_d.as<TrackDesc>(), ctx->shard_red.as<ShardReduced>(), st);
    HIPCHK(hipMemsetAsync(ctx->outs.p, 0, sizeof(TrackOut), st));
    launch_gate_final(ctx->tracks_d.as<TrackDesc>(), 1, ctx->cfg.gate_mode, ctx->blk_powers.as<double>(), ctx->hist.as<uint32_t>(),
                      ctx->ebu_energy.as<double>(), ctx->ebu_edges.as<double>(), ctx->ln_energy.as<double>(), ctx->ln_bounds.as<double>(),
                      ctx->gate_parts.as<GatePart>(), false, ctx->outs.as<TrackOut>(), st);
    launch_range_stats(ctx->tracks_d.as<TrackDesc>(), 1, nullptr, 0, ctx->cfg.gate_mode, ctx->hop_sums.as<double>(),
                       ctx->blk_powers.as<double>(), ctx->ln_bounds.as<double>(), ctx->ln_energy.as<double>(), ctx->ebu_edges.as<double>(),
                       ctx->st_hist.as<uint32_t>(), ctx->gate_parts.as<GatePart>(), true, ctx->outs.as<TrackOut>(), st);
    HIPCHK(hipGetLastError());
    TrackOut o{};
    ShardReduced red{};
    HIPCHK(hipMemcpyAsync(&o, ctx->outs.p, sizeof o, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&red, ctx->shard_red.p, sizeof red, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (red.too_large) return HR_ERR_TOO_LARGE;
    unsigned long long pk[kMaxCh] = {0};
    for (int c = 0; c < channels; c++) std::memcpy(&pk[c], &red.peak_per_ch[c], sizeof(double));
    std::memset(out, 0, sizeof *out);
    fill_result(ctx, o, pk, channels, out, red.range_ok != 0);
    if (red.redo) return HR_SHARD_AGAIN;  // (out holds lower bounds of the peaks until the redo has run on every rank)
    ctx->shard_open = false;
    ctx->shard_device = false;
    return HR_OK;
} HR_CATCH_ALL

int hr_shard_pass2(hr_ctx *ctx, const hr_shard_edge *left, hr_shard_partial *partial)
try {
    if (!ctx || !partial) return HR_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!ctx->shard_open || ctx->shard_device || ctx->shard_pass2_done || !ctx->shard_group) return HR_ERR_INVALID;
    HIPCHK(hipSetDevice(ctx->cfg.device));
    hipStream_t st = ctx->stream;
    Group &g = *ctx->shard_group;
    TrackDesc &d = g.descs[0];
    const int C = d.ch;
    const bool has_left = d.blk_first < 0;
    if (has_left != (left != nullptr)) return HR_ERR_INVALID;
    std::vector<HopRec> pre(kPreRec * C);
    if (left) {
        for (int k = 0; k < kPreRec; k++)
            for (int c = 0; c < C; c++) {
                HopRec &r = pre[k * C + c];
                const double *o = left->rec[k][c];
                r.e0 = o[0];
                for (int j = 0; j < 4; j++) { r.q[j] = o[1 + j]; r.s[j] = o[5 + j]; }
            }
        HIPCHK(hipMemcpyAsync(ctx->recs.as<HopRec>() + d.rec_base, pre.data(), pre.size() * sizeof(HopRec),
                              hipMemcpyHostToDevice, st));
    }
    // the left neighbour's last 29 hop energies go in front of hop 0 (the combine below rewrites the last three of them
    // from the records, to the same values): the 3 s windows that end in this shard are then all its own
    const bool range_ok = !left || left->range_ok;
    if (left && left->range_ok) {
        std::vector<double> ht(static_cast<size_t>(kPreHop) * C);
        for (int k = 0; k < kPreHop; k++)
            for (int c = 0; c < C; c++) ht[static_cast<size_t>(k) * C + c] = left->hop_tail[k][c];
        HIPCHK(hipMemcpyAsync(ctx->hop_sums.as<double>() + d.hop_base, ht.data(), ht.size() * sizeof(double), hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));  // `ht` is a local
        d.st_pre = kPreHop;
    }
    int rc = upload(ctx->tracks_d, g.descs.data(), g.descs.size() * sizeof(TrackDesc), st);  // (st_pre may have changed)
    if (rc) return rc;
    launch_hop_combine(ctx->tracks_d.as<TrackDesc>(), ctx->rates_d.as<RateDev>(), ctx->comb_items_d.as<WorkItem>(),
                       static_cast<uint32_t>(ctx->comb_n), ctx->recs.as<HopRec>(), ctx->hop_sums.as<double>(), false, st);
    rc = run_gating(ctx, g, false, true);
    if (rc) return rc;
    rc = redo_overflowed_peaks(ctx, g);
    if (rc) return rc;
    std::vector<uint32_t> hist(kEbuBins), st_hist(kEbuBins);
    std::vector<GatePart> parts(d.n_parts);
    TrackOut o{};
    unsigned long long pk[kMaxCh];
    HIPCHK(hipMemcpyAsync(st_hist.data(), ctx->st_hist.p, kEbuBins * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(parts.data(), ctx->gate_parts.as<GatePart>() + d.part_base, parts.size() * sizeof(GatePart),
                          hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(hist.data(), ctx->hist.p, kEbuBins * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&o, ctx->outs.p, sizeof o, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(pk, ctx->peaks.p, sizeof pk, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    std::memset(partial, 0, sizeof *partial);
    const int nb = ctx->cfg.gate_mode == HR_GATE_LOUDNORM ? kLnBins : kEbuBins;
    for (int i = 0; i < nb; i++) partial->hist[i] = hist[i];
    partial->sum_kept = o.sum_kept;
    partial->n_kept = o.n_kept;
    partial->n_blocks = o.n_blocks;
    for (int c = 0; c < C; c++) std::memcpy(&partial->peak_per_ch[c], &pk[c], sizeof(double));
    for (int i = 0; i < nb; i++) partial->st_hist[i] = st_hist[i];
    for (const GatePart &p : parts) {  // chunk order, as k_range_final sums them
        partial->st_sum_kept += p.st_sum_kept;
        partial->st_n_kept += p.st_n_kept;
        partial->max_m_power = std::max(partial->max_m_power, p.max_m_power);
        partial->max_s_power = std::max(partial->max_s_power, p.max_s_power);
    }
    partial->n_shortterm = d.n_hops + d.st_pre >= 30 ? d.n_hops + d.st_pre - 29 : 0;
    partial->range_ok = range_ok ? 1 : 0;
    ctx->shard_open = ctx->cfg.gate_mode == HR_GATE_EXACT;  // (its block powers are read once more: hr_shard_pass3)
    ctx->shard_pass2_done = true;
    return HR_OK;
} HR_CATCH_ALL

int hr_shard_pass3(hr_ctx *ctx, const hr_shard_partial *red, hr_shard_partial *partial)
try {
    if (!ctx || !red || !partial) return HR_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (ctx->cfg.gate_mode != HR_GATE_EXACT || !ctx->shard_open || !ctx->shard_pass2_done || !ctx->shard_group) return HR_ERR_INVALID;
    HIPCHK(hipSetDevice(ctx->cfg.device));
    hipStream_t st = ctx->stream;
    Group &g = *ctx->shard_group;
    // the shard's own track (descriptor 0) with the REDUCED abs-gated sums in its output slot: k_gate_final then takes
    // the relative gate from them (n_parts == 0: "sums already reduced") and sums this shard's blocks above both gates
    TrackDesc d = g.descs[0];
    d.n_parts = 0;
    TrackOut o{};
    o.sum_kept = red->sum_kept;
    o.n_kept = red->n_kept;
    o.n_blocks = d.n_blocks;
    HIPCHK(hipMemcpyAsync(ctx->tracks_d.p, &d, sizeof d, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ctx->outs.p, &o, sizeof o, hipMemcpyHostToDevice, st));
    launch_gate_final(ctx->tracks_d.as<TrackDesc>(), 1, HR_GATE_EXACT, ctx->blk_powers.as<double>(), ctx->hist.as<uint32_t>(),
                      ctx->ebu_energy.as<double>(), ctx->ebu_edges.as<double>(), ctx->ln_energy.as<double>(),
                      ctx->ln_bounds.as<double>(), ctx->gate_parts.as<GatePart>(), false, ctx->outs.as<TrackOut>(), st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&o, ctx->outs.p, sizeof o, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    std::memset(partial, 0, sizeof *partial);
    partial->sum_gated = o.sum_gated;
    partial->n_gated = o.n_gated_final;
    ctx->shard_open = false;
    return HR_OK;
} HR_CATCH_ALL

int hr_shard_finish(hr_ctx *ctx, const hr_shard_partial *red, int channels, hr_result *out)
try {
    if (!ctx || !red || !out || channels < 1 || channels > HR_MAX_CH) return HR_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->shard_open = false;  // the workspace an open time shard left behind is about to be reused: its pass 2 must fail
    HIPCHK(hipSetDevice(ctx->cfg.device));
    hipStream_t st = ctx->stream;
    std::vector<uint32_t> hist(2 * kEbuBins);  // gating histogram, then the short-term one
    for (int i = 0; i < kEbuBins; i++) {
        if (red->hist[i] > 0xffffffffull || red->st_hist[i] > 0xffffffffull) return HR_ERR_TOO_LARGE;
        hist[i] = static_cast<uint32_t>(red->hist[i]);
        hist[kEbuBins + i] = static_cast<uint32_t>(red->st_hist[i]);
    }
    // the reduced sums enter as the single "chunk" of a stand-in track
    GatePart gp{};
    gp.sum_kept = red->sum_kept;
    gp.n_kept = red->n_kept;
    gp.st_sum_kept = red->st_sum_kept;
    gp.st_n_kept = red->st_n_kept;
    gp.max_m_power = red->max_m_power;
    gp.max_s_power = red->max_s_power;
    TrackDesc d{};
    d.n_blocks = red->n_blocks;
    d.n_hops = red->n_shortterm ? red->n_shortterm + 29 : 0;
    d.part_base = 0;
    d.n_parts = 1;
    TrackOut o{};
    const bool exact = ctx->cfg.gate_mode == HR_GATE_EXACT;
    if (exact) {  // both reductions are done: what k_gate_final's exact branch would leave (hr_kernels_generic.hip)
        o.sum_kept = red->sum_kept;
        o.n_kept = red->n_kept;
        o.n_blocks = red->n_blocks;
        if (red->n_kept) {
            o.rel_energy = red->sum_kept / static_cast<double>(red->n_kept);
            o.n_gated_final = red->n_gated;
            o.sum_gated = red->sum_gated;
            o.mean_energy = red->n_gated ? red->sum_gated / static_cast<double>(red->n_gated) : 0.0;
        }
    }
    int rc = upload(ctx->hist, hist.data(), kEbuBins * sizeof(uint32_t), st);
    if (!rc) rc = upload(ctx->st_hist, hist.data() + kEbuBins, kEbuBins * sizeof(uint32_t), st);
    if (!rc) rc = upload(ctx->gate_parts, &gp, sizeof gp, st);
    if (!rc) rc = upload(ctx->outs, &o, sizeof o, st);
    if (!rc) rc = upload(ctx->tracks_d, &d, sizeof d, st);
    if (rc) return rc;
    if (!ctx->blk_powers.ensure(sizeof(double))) return HR_ERR_OOM;
    if (!exact)
        launch_gate_final(ctx->tracks_d.as<TrackDesc>(), 1, ctx->cfg.gate_mode, ctx->blk_powers.as<double>(),
                          ctx->hist.as<uint32_t>(), ctx->ebu_energy.as<double>(), ctx->ebu_edges.as<double>(), ctx->ln_energy.as<double>(),
                          ctx->ln_bounds.as<double>(), ctx->gate_parts.as<GatePart>(), false, ctx->outs.as<TrackOut>(), st);
    launch_range_stats(ctx->tracks_d.as<TrackDesc>(), 1, nullptr, 0, ctx->cfg.gate_mode, ctx->hop_sums.as<double>(),
                       ctx->blk_powers.as<double>(), ctx->ln_bounds.as<double>(), ctx->ln_energy.as<double>(),
                       ctx->ebu_edges.as<double>(), ctx->st_hist.as<uint32_t>(), ctx->gate_parts.as<GatePart>(), true, ctx->outs.as<TrackOut>(), st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&o, ctx->outs.p, sizeof o, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    unsigned long long pk[kMaxCh] = {0};
    for (int c = 0; c < channels; c++) std::memcpy(&pk[c], &red->peak_per_ch[c], sizeof(double));
    std::memset(out, 0, sizeof *out);
    fill_result(ctx, o, pk, channels, out, red->range_ok != 0);
    return HR_OK;
} HR_CATCH_ALL

}  // extern "C"

namespace {

// RCCL, opened at run time: the meter must load (and serve every other entry point) on machines without it, and in a process
// where PyTorch has already brought its own copy the loader hands back that one (same SONAME) instead of a second runtime.
struct RcclApi {
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};

const RcclApi &rccl()
{
    static const RcclApi api = [] {
        RcclApi a;
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return a;
        a.CommInitAll = reinterpret_cast<decltype(a.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(h, "ncclAllGather"));
        a.Send = reinterpret_cast<decltype(a.Send)>(dlsym(h, "ncclSend"));
        a.Recv = reinterpret_cast<decltype(a.Recv)>(dlsym(h, "ncclRecv"));
        a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(dlsym(h, "ncclGroupStart"));
        a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        a.ok = a.CommInitAll && a.CommDestroy && a.AllGather && a.Send && a.Recv && a.GroupStart && a.GroupEnd && a.GetErrorString;
        return a;
    }();
    return api;
}

// One communicator set per ordered list of distinct devices, kept for the life of the process (creating one costs tens of
// milliseconds; a stream is analysed in a few).
std::mutex g_comm_mu;
std::map<std::vector<int>, std::vector<ncclComm_t>> g_comms;

const std::vector<ncclComm_t> *comms_for(const std::vector<int> &devs)
{
    std::lock_guard<std::mutex> lk(g_comm_mu);
    auto it = g_comms.find(devs);
    if (it != g_comms.end()) return &it->second;
    std::vector<ncclComm_t> comms(devs.size());
    const ncclResult_t rc = rccl().CommInitAll(comms.data(), static_cast<int>(devs.size()), devs.data());
    if (rc != ncclSuccess) {
        std::fprintf(stderr, "headroom_hip: ncclCommInitAll failed: %s\n", rccl().GetErrorString(rc));
        return nullptr;
    }
    return &g_comms.emplace(devs, std::move(comms)).first->second;
}

#define NCCLCHK(expr)                                                                                      \
    do {                                                                                                   \
        const ncclResult_t r__ = (expr);                                                                   \
        if (r__ != ncclSuccess) {                                                                          \
            std::fprintf(stderr, "headroom_hip: %s failed: %s\n", #expr, rccl().GetErrorString(r__));      \
            return HR_ERR_HIP;                                                                             \
        }                                                                                                  \
    } while (0)

// The device-resident exchange of hr_analyze_stream_sharded (every gating mode but HR_GATE_EXACT): pass 1 leaves each
// shard's right edge in its context's buffer, the edges move one context to the right, pass 2 leaves the partial images,
// they are all-gathered, and the first context reduces them in shard order and finishes.  Contexts on DISTINCT devices
// exchange through RCCL (ncclSend / ncclRecv for the edges, one ncclAllGather for the partials: stream-ordered on each
// context's stream, over xGMI, no host synchronisation between the phases); contexts that share a device -- the tests and
// the one-GPU rehearsal: a communicator cannot hold one device twice -- take stream-ordered device copies instead.
int stream_sharded_device(hr_ctx *const *ctxs, int active, const std::vector<hr_track> &shard, const std::vector<uint64_t> &cut, uint32_t hop,
                          const std::vector<std::vector<double>> &head, const std::vector<std::vector<double>> &tail, int channels,
                          hr_result *out)
{
    std::vector<int> devs(active);
    bool distinct = true;
    for (int r = 0; r < active; r++) {
        devs[r] = ctxs[r]->cfg.device;
        for (int q = 0; q < r; q++) distinct &= devs[q] != devs[r];
    }
    // HR_SHARD_RCCL=0 keeps the copies; =1 also sends a single shard through a one-rank communicator (the test on a one-GPU box)
    const char *env = std::getenv("HR_SHARD_RCCL");
    const bool want = env ? std::atoi(env) != 0 : active > 1;
    const std::vector<ncclComm_t> *comms = nullptr;
    if (want && distinct && (active > 1 || env) && rccl().ok) comms = comms_for(devs);
    for (int r = 0; r < active; r++) {
        hr_ctx *c = ctxs[r];
        std::lock_guard<std::mutex> lk(c->mu);
        if (hipSetDevice(c->cfg.device) != hipSuccess) return HR_ERR_HIP;
        if (!c->shard_edge_buf.ensure(sizeof(hr_shard_edge)) || !c->shard_left_buf.ensure(sizeof(hr_shard_edge)) ||
            !c->shard_part_buf.ensure(sizeof(hr_shard_partial)) ||
            !c->shard_gather_buf.ensure(static_cast<size_t>(active) * sizeof(hr_shard_partial)))
            return HR_ERR_OOM;
    }
    std::vector<int> rcs(active, HR_OK);
    auto on_all = [&](auto fn) {  // one host thread per context: the GPUs work at the same time
        std::vector<std::thread> th;
        for (int r = 1; r < active; r++) th.emplace_back([&, r] { rcs[r] = fn(r); });
        rcs[0] = fn(0);
        for (auto &t : th) t.join();
        for (int r = 0; r < active; r++)
            if (rcs[r]) return rcs[r];
        return static_cast<int>(HR_OK);
    };
    // edges: context r's right edge becomes context r + 1's left edge
    auto move_edges = [&](int r) -> int {
        hr_ctx *c = ctxs[r];
        if (hipSetDevice(c->cfg.device) != hipSuccess) return HR_ERR_HIP;
        if (comms) {
            NCCLCHK(rccl().GroupStart());
            if (r + 1 < active) NCCLCHK(rccl().Send(c->shard_edge_buf.p, sizeof(hr_shard_edge), ncclUint8, r + 1, (*comms)[r], c->stream));
            if (r > 0) NCCLCHK(rccl().Recv(c->shard_left_buf.p, sizeof(hr_shard_edge), ncclUint8, r - 1, (*comms)[r], c->stream));
            NCCLCHK(rccl().GroupEnd());
        } else if (r > 0) {
            // (the left context's stream has been synchronised by the caller of this phase)
            HIPCHK(hipMemcpyPeerAsync(c->shard_left_buf.p, c->cfg.device, ctxs[r - 1]->shard_edge_buf.p, ctxs[r - 1]->cfg.device,
                                      sizeof(hr_shard_edge), c->stream));
        }
        return HR_OK;
    };
    auto gather = [&](int r) -> int {
        hr_ctx *c = ctxs[r];
        if (hipSetDevice(c->cfg.device) != hipSuccess) return HR_ERR_HIP;
        if (comms) {
            NCCLCHK(rccl().AllGather(c->shard_part_buf.p, c->shard_gather_buf.p, sizeof(hr_shard_partial), ncclUint8, (*comms)[r], c->stream));
        } else if (r == 0) {
            for (int q = 0; q < active; q++)
                HIPCHK(hipMemcpyPeerAsync(static_cast<char *>(c->shard_gather_buf.p) + static_cast<size_t>(q) * sizeof(hr_shard_partial), c->cfg.device,
                                          ctxs[q]->shard_part_buf.p, ctxs[q]->cfg.device, sizeof(hr_shard_partial), c->stream));
        }
        return HR_OK;
    };
    auto sync_all = [&]() -> int {  // (the copy path only: the collectives are ordered on the streams)
        for (int r = 0; r < active; r++) {
            if (hipSetDevice(ctxs[r]->cfg.device) != hipSuccess || hipStreamSynchronize(ctxs[r]->stream) != hipSuccess) return HR_ERR_HIP;
        }
        return HR_OK;
    };
    int rc = on_all([&](int r) {
        return hr_shard_pass1_device(ctxs[r], &shard[r], r ? tail[r - 1].data() : nullptr, r + 1 < active ? head[r + 1].data() : nullptr,
                                     r == 0, r + 1 == active, cut[r] / hop, ctxs[r]->shard_edge_buf.p);
    });
    if (rc) return rc;
    if (!comms && (rc = sync_all())) return rc;
    if ((rc = on_all(move_edges))) return rc;
    rc = on_all([&](int r) { return hr_shard_pass2_device(ctxs[r], r ? ctxs[r]->shard_left_buf.p : nullptr, ctxs[r]->shard_part_buf.p); });
    if (rc) return rc;
    for (int round = 0; round < 2; round++) {
        if (!comms && (rc = sync_all())) return rc;
        if ((rc = on_all(gather))) return rc;
        rc = hr_shard_finish_gathered(ctxs[0], ctxs[0]->shard_gather_buf.p, active, channels, out);
        if (rc != HR_SHARD_AGAIN) break;
        // some shard's true-peak screen left work for the exact redo (tones): every context patches its image, then once more
        if ((rc = on_all([&](int r) { return hr_shard_redo_device(ctxs[r], ctxs[r]->shard_part_buf.p); }))) return rc;
        rc = HR_ERR_HIP;  // (a second HR_SHARD_AGAIN cannot happen: the redo clears the flag)
    }
    // the other contexts' shards are done too
    for (int r = 1; r < active; r++) {
        std::lock_guard<std::mutex> lk(ctxs[r]->mu);
        if (comms && (hipSetDevice(ctxs[r]->cfg.device) != hipSuccess || hipStreamSynchronize(ctxs[r]->stream) != hipSuccess)) return HR_ERR_HIP;
        ctxs[r]->shard_open = false;
        ctxs[r]->shard_device = false;
    }
    return rc;
}

}  // namespace

extern "C" {

// Config 5 inside one process: one context per GPU.  HR_GATE_EXACT: the exchange steps on the host (a few KB); the other
// gating modes: records resident on the devices, RCCL between distinct devices (stream_sharded_device above).
int hr_analyze_stream_sharded(hr_ctx *const *ctxs, int n_ctx, const hr_track *track, hr_result *out)
try {
    if (!ctxs || n_ctx < 1 || !track || !out) return HR_ERR_INVALID;
    for (int i = 0; i < n_ctx; i++)
        if (!ctxs[i] || ctxs[i]->cfg.gate_mode != ctxs[0]->cfg.gate_mode || ctxs[i]->cfg.tp_mode != ctxs[0]->cfg.tp_mode)
            return HR_ERR_INVALID;
    for (int i = 0; i < n_ctx; i++)
        for (int j = 0; j < i; j++)
            if (ctxs[i] == ctxs[j]) return HR_ERR_INVALID;  // one context holds ONE open shard between pass 1 and pass 2
    const int v = validate_track(ctxs[0], *track);
    if (v) return v;
    if (track->flags & HR_TRACK_DEVICE_PCM)  // a device pointer belongs to one GPU: all contexts must sit on it
        for (int i = 1; i < n_ctx; i++)
            if (ctxs[i]->cfg.device != ctxs[0]->cfg.device) return HR_ERR_INVALID;
    const uint32_t hop = track->sample_rate / 10;
    const uint64_t n_hops = hop ? track->frames / hop : 0;
    const int active = static_cast<int>(std::max<uint64_t>(1, std::min<uint64_t>(n_ctx, n_hops / kPreRec)));
    std::vector<uint64_t> cut(active + 1, 0);
    for (int r = 1; r < active; r++) cut[r] = (n_hops * r / active) * hop;
    cut[active] = track->frames;
    const size_t frame_bytes = fmt_bytes(track->fmt) * track->channels;
    std::vector<hr_track> shard(active, *track);
    for (int r = 0; r < active; r++) {
        shard[r].pcm = static_cast<const char *>(track->pcm) + cut[r] * frame_bytes;
        shard[r].frames = cut[r + 1] - cut[r];
    }
    // halos: the 16 frames either side of every cut (host copies / small D2H reads)
    std::vector<std::vector<double>> head(active, std::vector<double>(kHalo * HR_MAX_CH)), tail(active, std::vector<double>(kHalo * HR_MAX_CH));
    for (int r = 0; r < active; r++) {
        int rc = hr_shard_head(ctxs[r], &shard[r], head[r].data());
        if (rc) return rc;
        if (r + 1 < active) {  // every shard but the last holds at least six hops, so it has 16 frames
            hr_track last16 = shard[r];
            last16.pcm = static_cast<const char *>(shard[r].pcm) + (shard[r].frames - kHalo) * frame_bytes;
            last16.frames = kHalo;
            rc = hr_shard_head(ctxs[r], &last16, tail[r].data());
            if (rc) return rc;
        }
    }
    if (ctxs[0]->cfg.gate_mode != HR_GATE_EXACT) return stream_sharded_device(ctxs, active, shard, cut, hop, head, tail, track->channels, out);
    std::vector<hr_shard_edge> edge(active);
    std::vector<hr_shard_partial> part(active);
    std::vector<int> rcs(active, HR_OK);
    auto on_all = [&](auto fn) {  // one host thread per context: the GPUs work at the same time
        std::vector<std::thread> th;
        for (int r = 1; r < active; r++) th.emplace_back([&, r] { rcs[r] = fn(r); });
        rcs[0] = fn(0);
        for (auto &t : th) t.join();
        for (int r = 0; r < active; r++)
            if (rcs[r]) return rcs[r];
        return static_cast<int>(HR_OK);
    };
    int rc = on_all([&](int r) {
        return hr_shard_pass1(ctxs[r], &shard[r], r ? tail[r - 1].data() : nullptr, r + 1 < active ? head[r + 1].data() : nullptr,
                              r == 0, r + 1 == active, cut[r] / hop, &edge[r]);
    });
    if (rc) return rc;
    rc = on_all([&](int r) { return hr_shard_pass2(ctxs[r], r ? &edge[r - 1] : nullptr, &part[r]); });
    if (rc) return rc;
    hr_shard_partial red{};
    for (int r = 0; r < active; r++) {  // the "all-reduce": counts and the kept-power sum add, peaks take the maximum
        for (int b = 0; b < kEbuBins; b++) red.hist[b] += part[r].hist[b];
        red.sum_kept += part[r].sum_kept;
        red.n_kept += part[r].n_kept;
        red.n_blocks += part[r].n_blocks;
        for (int c = 0; c < HR_MAX_CH; c++) red.peak_per_ch[c] = std::max(red.peak_per_ch[c], part[r].peak_per_ch[c]);
        for (int b = 0; b < kEbuBins; b++) red.st_hist[b] += part[r].st_hist[b];
        red.st_sum_kept += part[r].st_sum_kept;
        red.st_n_kept += part[r].st_n_kept;
        red.n_shortterm += part[r].n_shortterm;
        red.max_m_power = std::max(red.max_m_power, part[r].max_m_power);
        red.max_s_power = std::max(red.max_s_power, part[r].max_s_power);
        red.range_ok = r == 0 ? part[r].range_ok : std::min(red.range_ok, part[r].range_ok);
    }
    if (ctxs[0]->cfg.gate_mode == HR_GATE_EXACT) {  // the second reduction: blocks above the relative gate the first one set
        std::vector<hr_shard_partial> part3(active);
        rc = on_all([&](int r) { return hr_shard_pass3(ctxs[r], &red, &part3[r]); });
        if (rc) return rc;
        for (int r = 0; r < active; r++) {
            red.sum_gated += part3[r].sum_gated;
            red.n_gated += part3[r].n_gated;
        }
    }
    return hr_shard_finish(ctxs[0], &red, track->channels, out);
} HR_CATCH_ALL

int hr_loudnorm_first_pass(hr_ctx *ctx, const hr_track *tracks, size_t n, hr_result *out)
try {
    if (!ctx || (n && (!tracks || !out))) return HR_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->shard_open = false;  // the workspace an open time shard left behind is about to be reused: its pass 2 must fail
    HIPCHK(hipSetDevice(ctx->cfg.device));
    ctx->last_total_ms = ctx->last_main_ms = 0;
    hipStream_t st = ctx->stream;
    // libavfilter/ebur128.c channel map by index (af_loudnorm sets no layout-specific map): L R C LFE Ls Rs
    static const float wln[HR_MAX_CH + 1][HR_MAX_CH] = {
        {0}, {1}, {1, 1}, {1, 1, 1}, {1, 1, 1.41f, 1.41f}, {1, 1, 1, 1.41f, 1.41f},
        {1, 1, 1, 0, 1.41f, 1.41f}, {1, 1, 1, 0, 1.41f, 1.41f, 0}, {1, 1, 1, 0, 1.41f, 1.41f, 0, 0}};
    // (restored, and the first-pass hand-over to run_measure withdrawn, however this function is left: an exception on
    // the way must not leave the context in loudnorm's modes)
    struct Restore {
        hr_ctx *c;
        hr_config saved;
        ~Restore() { c->cfg = saved; c->fp_ext_hops.clear(); c->fp_srcs.clear(); }
    } restore{ctx, ctx->cfg};
    ctx->cfg.gate_mode = HR_GATE_LOUDNORM;
    ctx->cfg.tp_mode = HR_TP_SAMPLE;
    const double ninf = -std::numeric_limits<double>::infinity(), nan = std::numeric_limits<double>::quiet_NaN();
    // resampled f64 PCM per group: the more tracks share a launch of the one-thread-per-hop kernels the better they fill
    // the GPU, so up to 32 GiB where a quarter of the free memory allows it (the buffer stays with the context)
    size_t limit = 8ull << 30;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) limit = std::min<size_t>(32ull << 30, std::max<size_t>(2ull << 30, free_b / 4));
    }
    int fatal = HR_OK;
    size_t i = 0;
    while (i < n && !fatal) {
        // tracks [i, j): as many as fit the resampled-PCM budget (at least one)
        std::vector<hr_track> pseudo;
        std::vector<size_t> src_idx, off;
        struct Ext { uint64_t hops, n_start, pos_begin; };  // see below: what k_first_pass_rows takes, where the resampled rest starts
        std::vector<Ext> ext;
        size_t need = 0, staged = 0, j = i;
        for (; j < n; j++) {
            const hr_track &t = tracks[j];
            std::memset(&out[j], 0, sizeof out[j]);
            out[j].integrated_lufs = out[j].true_peak_dbtp = out[j].rel_threshold_lufs = ninf;
            out[j].loudness_range_lu = out[j].lra_low_lufs = out[j].lra_high_lufs = out[j].lra_threshold_lufs =
                out[j].max_momentary_lufs = out[j].max_shortterm_lufs = nan;
            int v = HR_OK;
            const RateTables *rt = nullptr;
            if (t.channels < 1 || t.channels > HR_MAX_CH || fmt_bytes(t.fmt) == 0) v = HR_ERR_FORMAT;
            else if (t.sample_rate < 10 || t.sample_rate > HR_MAX_SAMPLE_RATE || !(rt = rate_tables(t.sample_rate)) || rt->swr_phases < 0 ||
                     (rt->swr_phases == 0 && t.sample_rate != 192000)) v = HR_ERR_RATE;
            else if (t.frames && !t.pcm) v = HR_ERR_INVALID;
            if (v) { out[j].status = v; continue; }
            uint64_t m = t.frames;  // frames at 192 kHz
            if (rt->swr_phases > 0)
                m = t.frames >= kSwrTaps + 1
                        ? (t.frames * rt->swr_phases * static_cast<uint64_t>(rt->swr_S) + rt->swr_step - 1) / rt->swr_step : 0;
            // Whole 192 kHz hops whose input windows stay clear of the end of the track (frames up to position + 16) are
            // measured straight from the input PCM by k_first_pass_rows: exact phase counts, an input hop that is a whole
            // number of resampler cycles (then it yields exactly 19 200 outputs).  Only the rest -- from three input positions
            // in front of the first hop that is left, for its two samples of filter history -- is resampled into memory.
            uint64_t eh = 0, n_start = 0, pos_begin = 0;
            const uint64_t hop_in = t.sample_rate / 10;
            if (ctx->cfg.engine != 1 && m && rt->swr_phases > 0 && rt->swr_S == 1 && rt->swr_step >= 1 && hop_in * 10 == t.sample_rate &&
                hop_in % static_cast<uint64_t>(rt->swr_step) == 0 && hop_in * rt->swr_phases / rt->swr_step == 19200 &&
                hop_in <= (1u << 20) && t.frames >= 2 * hop_in + 64) {
                eh = std::min<uint64_t>((t.frames - 16) / hop_in, 0xffffff00u);
                // (every input position holds at least one output -- the rate is at most 192 kHz -- so three positions in
                // front of the hop hold the two outputs its filter history needs, whatever the ratio)
                pos_begin = eh * hop_in - 3;
                n_start = (pos_begin * rt->swr_phases + rt->swr_step - 1) / rt->swr_step;
            }
            const size_t bytes = ((m - n_start) * t.channels * sizeof(double) + 255) / 256 * 256;
            // (host tracks of a group are staged side by side: their bytes count against the same budget)
            const size_t in_stage = (t.flags & HR_TRACK_DEVICE_PCM) ? 0 : (t.frames * t.channels * fmt_bytes(t.fmt) + 255) / 256 * 256;
            if (!pseudo.empty() && need + staged + bytes + in_stage > limit) break;
            staged += in_stage;
            hr_track p = t;
            p.frames = m;
            p.sample_rate = 192000;
            p.fmt = HR_FMT_F64;
            p.flags = HR_TRACK_DEVICE_PCM;
            if (!p.ch_weight) p.ch_weight = wln[t.channels];
            pseudo.push_back(p);
            src_idx.push_back(j);
            off.push_back(need);
            ext.push_back({eh, n_start, pos_begin});
            need += bytes;
        }
        if (!pseudo.empty()) {
            int rc = ctx->resampled.ensure(need + 256) ? HR_OK : HR_ERR_OOM;
            // host tracks are staged side by side (k_first_pass_rows reads them while the group is measured)
            size_t stage_need = 0;
            std::vector<size_t> soff(pseudo.size(), 0);
            for (size_t k = 0; k < pseudo.size(); k++) {
                const hr_track &t = tracks[src_idx[k]];
                if (t.flags & HR_TRACK_DEVICE_PCM) continue;
                soff[k] = stage_need;
                stage_need += (t.frames * t.channels * fmt_bytes(t.fmt) + 255) / 256 * 256;
            }
            if (!rc && stage_need && !ctx->stage.ensure(stage_need + 256)) rc = HR_ERR_OOM;
            ctx->fp_srcs.assign(pseudo.size(), FpSrc{});
            ctx->fp_ext_hops.assign(pseudo.size(), 0);
            // The resampled stream is read only by the generic kernels, which run_measure puts on the side stream: it is
            // written there too, in front of them, and so beside k_first_pass_rows instead of in front of it.  (The side
            // stream first waits for whatever the main stream still does with the buffers: the previous group's kernels.)
            hipStream_t rs = ctx->serial_tail ? st : ctx->side;
            if (!rc && rs != st && (hipEventRecord(ctx->ev_fork, st) != hipSuccess || hipStreamWaitEvent(rs, ctx->ev_fork, 0) != hipSuccess)) rc = HR_ERR_HIP;
            for (size_t k = 0; k < pseudo.size() && !rc; k++) {
                const hr_track &t = tracks[src_idx[k]];
                // (the pseudo-track's frame n_start sits at the start of its share of the buffer: nothing in front of it is read)
                double *dst = reinterpret_cast<double *>(static_cast<char *>(ctx->resampled.p) + off[k]) - ext[k].n_start * t.channels;
                pseudo[k].pcm = dst;
                const size_t in_bytes = t.frames * t.channels * fmt_bytes(t.fmt);
                if (!pseudo[k].frames || !in_bytes) continue;
                const void *src = t.pcm;
                if (!(t.flags & HR_TRACK_DEVICE_PCM)) {
                    src = static_cast<char *>(ctx->stage.p) + soff[k];
                    if (hipMemcpyAsync(const_cast<void *>(src), t.pcm, in_bytes, hipMemcpyHostToDevice, st) != hipSuccess) { rc = HR_ERR_HIP; break; }
                    if (rs != st && (hipEventRecord(ctx->ev_fork, st) != hipSuccess || hipStreamWaitEvent(rs, ctx->ev_fork, 0) != hipSuccess)) { rc = HR_ERR_HIP; break; }
                }
                // always libswresample's bank, whatever interpolator the context's own tables hold
                const RateTables *rt = rate_tables(t.sample_rate);
                const double *bank = nullptr;
                for (const auto &e : ctx->swr_banks)
                    if (e.first == t.sample_rate) bank = e.second;
                if (!bank && rt->swr_phases > 0) {
                    void *p = nullptr;
                    if (hipMalloc(&p, rt->swr_taps.size() * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); rc = HR_ERR_OOM; break; }
                    ctx->rate_allocs.push_back(p);
                    if (hipMemcpy(p, rt->swr_taps.data(), rt->swr_taps.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { rc = HR_ERR_HIP; break; }
                    bank = static_cast<const double *>(p);
                    ctx->swr_banks.emplace_back(t.sample_rate, bank);
                }
                if (rt->swr_phases > 0)
                    launch_resample_swr(t.fmt, src, t.frames, t.channels, rt->swr_phases, rt->swr_step, rt->swr_S, rt->swr_frac_scale,
                                        rt->swr_inv_src_incr, bank, dst, ext[k].pos_begin, rs);
                else  // 192 kHz input: the resampler passes it through; the f64 copy keeps one code path
                    launch_resample_swr(t.fmt, src, t.frames, t.channels, 1, 1, 1, 1u, 1.0, ctx->unit_tap.as<double>(), dst, 0, rs);
                if (ext[k].hops) {
                    FpSrc &fs = ctx->fp_srcs[k];
                    fs.pcm = src;
                    fs.frames = t.frames;
                    fs.taps = bank;
                    fs.channels = t.channels;
                    fs.fmt = static_cast<uint32_t>(t.fmt);
                    fs.hop_in = t.sample_rate / 10;
                    fs.phases = rt->swr_phases;
                    fs.step = rt->swr_step;
                    ctx->fp_ext_hops[k] = ext[k].hops;
                }
            }
            if (!rc && hipGetLastError() != hipSuccess) rc = HR_ERR_HIP;
            std::vector<hr_result> res(pseudo.size());
            if (!rc) {
                std::vector<size_t> idx(pseudo.size());
                for (size_t k = 0; k < idx.size(); k++) idx[k] = k;
                rc = analyze_group(ctx, pseudo.data(), idx, res.data());
            }
            ctx->fp_ext_hops.clear();
            ctx->fp_srcs.clear();
            for (size_t k = 0; k < pseudo.size(); k++) {
                if (rc) out[src_idx[k]].status = rc;
                else out[src_idx[k]] = res[k];
            }
            if (rc == HR_ERR_HIP) fatal = rc;
        }
        i = j;
    }
    return fatal;
} HR_CATCH_ALL

int hr_apply_gain(hr_ctx *ctx, const hr_track *track, double gain_db, int out_big_endian, void *out, uint64_t *clipped)
try {
    if (!ctx || !track || (track->frames && (!track->pcm || !out)) || !std::isfinite(gain_db)) return HR_ERR_INVALID;
    if (track->channels < 1 || track->channels > HR_MAX_CH || fmt_bytes(track->fmt) == 0) return HR_ERR_FORMAT;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->shard_open = false;  // the workspace an open time shard left behind is about to be reused: its pass 2 must fail
    HIPCHK(hipSetDevice(ctx->cfg.device));
    hipStream_t st = ctx->stream;
    const uint64_t n = track->frames * track->channels;
    const size_t in_bytes = n * fmt_bytes(track->fmt), out_bytes = n * 3;
    const bool on_dev = (track->flags & HR_TRACK_DEVICE_PCM) != 0;
    if (on_dev && (reinterpret_cast<uintptr_t>(out) & 3)) return HR_ERR_INVALID;
    const void *src = track->pcm;
    void *dst = out;
    const size_t in_pad = (in_bytes + 255) / 256 * 256;
    if (!ctx->counters.ensure(256)) return HR_ERR_OOM;
    unsigned long long *cnt = ctx->counters.as<unsigned long long>();
    HIPCHK(hipMemsetAsync(cnt, 0, sizeof *cnt, st));
    if (!on_dev) {
        if (!ctx->stage.ensure(in_pad + out_bytes + 256)) return HR_ERR_OOM;
        if (in_bytes) HIPCHK(hipMemcpyAsync(ctx->stage.p, track->pcm, in_bytes, hipMemcpyHostToDevice, st));
        src = ctx->stage.p;
        dst = static_cast<char *>(ctx->stage.p) + in_pad;
    }
    // volume=XdB: the expression evaluator turns "XdB" into 10^(X/20) (double); the float path multiplies by (float) of it
    const float vol = static_cast<float>(std::pow(10.0, gain_db / 20.0));
    launch_apply_gain(track->fmt, out_big_endian != 0, src, dst, n, vol, cnt, st);
    HIPCHK(hipGetLastError());
    unsigned long long c = 0;
    if (!on_dev && out_bytes) HIPCHK(hipMemcpyAsync(out, dst, out_bytes, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&c, cnt, sizeof c, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (clipped) *clipped = c;
    return HR_OK;
} HR_CATCH_ALL

int hr_synth_fill(hr_ctx *ctx, void *device_pcm, uint64_t seed, uint32_t sample_rate, uint32_t channels, uint64_t frames)
try {
    if (!ctx || !device_pcm || channels < 1 || channels > HR_MAX_CH || sample_rate < 10) return HR_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->cfg.device));
    SynthPar p;
    synth_params(seed, sample_rate, frames, &p);
    launch_synth(static_cast<float *>(device_pcm), seed, p, channels, frames, ctx->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return HR_OK;
} HR_CATCH_ALL

int hr_last_timing(hr_ctx *ctx, float *total_ms, float *main_kernel_ms)
try {
    if (!ctx) return HR_ERR_INVALID;
    if (total_ms) *total_ms = ctx->last_total_ms;
    if (main_kernel_ms) *main_kernel_ms = ctx->last_main_ms;
    return HR_OK;
} HR_CATCH_ALL

const char *hr_strerror(int code)
{
    switch (code) {
    case HR_OK: return "ok";
    case HR_ERR_INVALID: return "invalid argument";
    case HR_ERR_NO_DEVICE: return "no HIP device found (the meter has no CPU fallback)";
    case HR_ERR_HIP: return "HIP runtime error";
    case HR_ERR_OOM: return "out of device or pinned memory";
    case HR_ERR_RATE: return "sample rate not supported by the selected true-peak mode";
    case HR_ERR_FORMAT: return "unsupported sample format or channel count";
    case HR_ERR_TOO_LARGE: return "input exceeds the entry point's limit";
    case HR_ERR_NOT_READY: return "device tables for this sample rate are not resident yet (make one ordinary call at this rate first)";
    default: return "unknown error";
    }
}

}  // extern "C"
