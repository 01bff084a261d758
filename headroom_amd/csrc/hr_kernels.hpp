// hr_kernels.hpp -- launchers implemented in the .hip files.
#pragma once

#include <hip/hip_runtime.h>

#include "hr_device.hpp"

namespace hr {

// hr_kernels_generic.hip
void launch_hop_zero_state(int fmt, const TrackDesc *tracks, const RateDev *rates, const WorkItem *items,
                           uint32_t n_items, HopRec *recs, hipStream_t st);
// one lane per item; item.first = hop, item.pad = channel
void launch_hop_zero_state_list(int fmt, const TrackDesc *tracks, const RateDev *rates, const WorkItem *items,
                                uint32_t n_items, HopRec *recs, hipStream_t st);
void launch_hop_combine(const TrackDesc *tracks, const RateDev *rates, const WorkItem *items, uint32_t n_items,
                        const HopRec *recs, double *hop_sums, bool any_rows, hipStream_t st);
// ebu_edges: hr_tables.hpp ebur128_bin_edges() on the device (3 x kEbuBins powers): f_ebur128.c's bin decisions without a
// device logarithm.
// Gating runs in chunks of kGateChunk blocks, one workgroup each (items: track, pad = chunk index, first = first block);
// `hist` / `st_hist` ([n_tracks][kEbuBins]) must be zero before launch_block_hist / launch_range_stats.
void launch_block_hist(const TrackDesc *tracks, const WorkItem *items, uint32_t n_items, int gate_mode, const double *hop_sums,
                       const double *ln_bounds, const double *ebu_edges, double *block_powers, uint32_t *hist, GatePart *parts,
                       hipStream_t st);
// per track: sums the chunks' partial sums into `outs` (tracks with n_parts == 0 bring reduced sums in `outs` already);
// unless reduce_only, also the relative gate and the gated mean.  list (here and in launch_range_stats): the n_tracks tracks
// the launch is for, by WorkItem::track; nullptr = tracks 0 .. n_tracks - 1
void launch_gate_final(const TrackDesc *tracks, uint32_t n_tracks, int gate_mode, const double *block_powers,
                       const uint32_t *hist, const double *ebu_energy, const double *ebu_edges, const double *ln_energy,
                       const double *ln_bounds, const GatePart *parts, bool reduce_only, TrackOut *outs, hipStream_t st,
                       const WorkItem *list = nullptr);
// Tracks of at most kGateSmallParts chunks: everything launch_block_hist, launch_gate_final and launch_range_stats (finalize)
// produce for them, by one workgroup per track with the histograms in LDS -- same results bit for bit, no `hist` / `st_hist`
// / GatePart traffic (k_gate_small)
void launch_gate_small(const TrackDesc *tracks, const WorkItem *list, uint32_t n_list, int gate_mode, const double *hop_sums,
                       double *block_powers, const double *ebu_energy, const double *ebu_edges, const double *ln_energy,
                       const double *ln_bounds, TrackOut *outs, hipStream_t st);
// loudness range + maximum 400 ms / 3 s powers (after launch_block_hist: reads the block powers): the chunks' histograms
// and partial sums, and with `finalize` the per-track result (n_items == 0: only that, from reduced partials)
void launch_range_stats(const TrackDesc *tracks, uint32_t n_tracks, const WorkItem *items, uint32_t n_items, int gate_mode,
                        const double *hop_sums, const double *block_powers, const double *ln_bounds, const double *ln_energy,
                        const double *ebu_edges, uint32_t *st_hist, GatePart *parts, bool finalize, TrackOut *outs, hipStream_t st,
                        const WorkItem *list = nullptr);
// hop sums of the stereo planes of a split track -> the track's own (hop, channel) slots; one workgroup per parent
void launch_gather_planes(const TrackDesc *tracks, const uint32_t *parents, uint32_t n_parents, double *hop_sums, hipStream_t st);
// libswresample-default resampling to 192 kHz with end flush: dst holds ceil(frames*phases*S/step) interleaved f64 frames
// (S, frac_scale, inv_src_incr: RateTables::swr_S / swr_frac_scale / swr_inv_src_incr; 1, 1, 1.0 for exact phase counts)
// pos0: first input position whose outputs are written (dst still addresses output n at dst[n * channels + c])
void launch_resample_swr(int fmt, const void *src, uint64_t frames, int channels, int phases, int step, int S, uint32_t frac_scale,
                         double inv_src_incr, const double *taps, double *dst, uint64_t pos0, hipStream_t st);
// tp_kind: 0 = the context's interpolation bank (libswresample's or BS.1770's, RateDev), 2 = sample peak; every item
// covers `span` (multiple of 256) positions, or item.pad positions when that is non-zero (bank kernel only)
void launch_true_peak_generic(int fmt, int tp_kind, const TrackDesc *tracks, const RateDev *rates,
                              const WorkItem *items, uint32_t n_items, uint32_t span,
                              unsigned long long *peaks, hipStream_t st);

// hr_kernels_fused.hip: stereo f32, swr true peak.  variant 0 = full; 1 = K-weight only, 2 = true peak only,
// 3 = data movement only (ablation builds for the roofline analysis); dbg: timing-only switches, 0 in production.
struct FusedScratch {
    void *cands;         // fused_cand_bytes(n_items): per-wave candidate lists
    unsigned *cand_cnt;  // [n_items]
    unsigned *tpb;       // [n_tracks][kMaxCh][2]: final lower bound / sample peak per channel (float bits), zeroed per call
    unsigned *overflow;  // [n_tracks], zeroed per call: bit 0 = samples beyond the f16 range (the track's fused positions are redone
                         // exactly), bit 1 = some wave's list overflowed (those waves are: kCandOverflowBit in their counts)
};
size_t fused_cand_bytes(uint32_t n_items);
// Exact true peak (the context's bank, exact phase counts only) of ranges of stereo f32 / s16 / s24 tracks: an item is one
// wave, item.first its first position (a multiple of the resampler's input step), the low 16 bits of item.pad its positions
// (<= 64 * redo_lane_frames(step)), the high 16 bits the first channel of the pair it covers (0 for a stereo track; tracks of
// more channels go frame by frame).  The redo of fused waves whose candidate lists overflowed.
void launch_true_peak_rows(int fmt, const TrackDesc *tracks, const RateDev *rates, const WorkItem *items, uint32_t n_items,
                           unsigned long long *peaks, hipStream_t st);
// exact evaluation of the candidate lists of a fused launch: n_lists lists, wpi of them per item (1: stereo; channels / 2: a
// multichannel track read directly, list w of an item being channels 2 w and 2 w + 1)
void launch_refine_cands(const TrackDesc *tracks, const RateDev *rates, const FusedItem *items, uint32_t n_lists, uint32_t wpi,
                         const FusedScratch &fs, unsigned long long *peaks, hipStream_t st);
// hr_kernels_fused_mc.hip: the fused pass on interleaved 5.1 / 7.1 tracks read directly (one item = 64 rows of ALL channels, a
// workgroup of channels / 2 waves; fs.cands / fs.cand_cnt hold channels / 2 lists per item).  fused_mc_supported: is there a
// build for this channel count, sample format (HR_FMT_F32 / S16 / S24) and true-peak class?
bool fused_mc_supported(int channels, int fmt, int tp_class);
bool fused_mc4_supported(int fmt, int tp_class);
// hr_kernels_fused_mc4.hip: the four-wave 5.1 kernel alone (launch_fused_mc() follows it with the refine kernel)
void launch_fused_mc4(int fmt, int tp_class, const TrackDesc *tracks, const RateDev *rates, const FusedItem *items, uint32_t n_items,
                      HopRec *recs, const FusedScratch &fs, unsigned long long *peaks, hipStream_t st);
// four_waves (5.1 only, where fused_mc4_supported()): the workgroup is three screening waves + one K-weighting wave (hr_fused_core.hpp, ROLE) instead of three
// waves that do both; same lists (three per item), same records, same results
void launch_fused_mc(int channels, int fmt, int tp_class, bool four_waves, const TrackDesc *tracks, const RateDev *rates, const FusedItem *items,
                     uint32_t n_items, HopRec *recs, const FusedScratch &fs, unsigned long long *peaks, hipStream_t st);
// All items of one launch share a true-peak class (the kernel is specialised on the tile shape of the rate).
int fused_tp_class(int tp_mt, int tp_last_rows);
void launch_fused_f32s(const TrackDesc *tracks, const RateDev *rates, const FusedItem *items, uint32_t n_items,
                       HopRec *recs, const FusedScratch &fs, unsigned long long *peaks, int variant, int tp_class, int dbg,
                       hipStream_t st);

// hr_kernels_firstpass.hip: the loudnorm first pass of whole 192 kHz hops straight from the input PCM (no resampled stream)
struct FpSrc {            // one source track
    const void *pcm;      // device, interleaved, at the input rate
    uint64_t frames;
    uint64_t rec_base;    // of its 192 kHz descriptor: (hop h, ch c) is rec_base + (kPreRec + h)*channels + c
    const double *taps;   // libswresample's bank for the input rate, [phases][32]
    uint32_t channels, fmt;
    uint32_t hop_in;      // input frames per 192 kHz hop (fs / 10; a multiple of `step`)
    int32_t phases, step;
    uint32_t desc;        // index of the 192 kHz descriptor (its peak slots)
    int32_t rate192;      // index of the 192 kHz rate tables
    int32_t pad_;
};
struct FpItem { uint32_t src, ch, first_hop, n_rows; };  // one wave: hops first_hop .. first_hop + n_rows - 1 (<= 64) of channel ch
void launch_first_pass_rows(const FpSrc *srcs, const RateDev *rates, const FpItem *items, uint32_t n_items, HopRec *recs,
                            unsigned long long *peaks, hipStream_t st);

// hr_kernels_shard.hip: the exchange records of a time-sharded stream on the device, in the layouts of headroom_hip.h
// (edge / partial / partials point at hr_shard_edge / hr_shard_partial images in device memory)
struct ShardReduced {  // what the host still needs of the reduced partials
    double peak_per_ch[kMaxCh];
    int32_t range_ok;   // minimum over the shards
    int32_t redo;       // some shard's image carries peaks that its exact redo has not confirmed yet
    int32_t too_large;  // a reduced histogram count passed 32 bits
    int32_t pad_;
};
void launch_pack_edge(const TrackDesc *tracks, const HopRec *recs, const double *hop_sums, bool is_first, void *edge, hipStream_t st);
void launch_unpack_edge(TrackDesc *tracks, HopRec *recs, double *hop_sums, const void *left_edge, hipStream_t st);
void launch_pack_partial(const TrackDesc *tracks, const uint32_t *hist, const uint32_t *st_hist, const GatePart *parts, const TrackOut *outs,
                         const unsigned long long *peaks, const unsigned *overflow, const void *left_edge, int nb, void *partial, hipStream_t st);
void launch_patch_partial_peaks(const TrackDesc *tracks, const unsigned long long *peaks, void *partial, hipStream_t st);
void launch_reduce_partials(const void *partials, int n, int nb, uint32_t *hist, uint32_t *st_hist, GatePart *gp, TrackDesc *tracks,
                            ShardReduced *red, hipStream_t st);

// hr_gain.hip: interleaved PCM of format fmt times vol (f32) -> packed s24 (dst dword aligned); *clipped += samples
// that saturated
void launch_apply_gain(int fmt, bool big_endian, const void *src, void *dst, uint64_t n_samples, float vol,
                       unsigned long long *clipped, hipStream_t st);

// hr_synth.hip
struct SynthPar {
    float a_noise, a_sine, burst;
    uint32_t sine_step;
    uint64_t quiet_len;
    uint64_t burst_pos[3];
    uint32_t burst_len;
};
void synth_params(uint64_t seed, uint32_t fs, uint64_t frames, SynthPar *p);
// interleaved f32 / s16 / s24 (fmt 0 / 1 / 2; src dword aligned) of any channel count -> (channels+1)/2 stereo f32
// planes, plane p at dst + p*plane_floats (16-byte aligned), exact
void launch_split_planes(int fmt, const void *src, float *dst, uint64_t frames, uint32_t channels, uint64_t plane_floats,
                         bool keep_format, hipStream_t st);
void launch_synth(float *out, uint64_t seed, const SynthPar &p, uint32_t channels, uint64_t frames, hipStream_t st);

}  // namespace hr
