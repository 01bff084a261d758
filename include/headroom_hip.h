/*
 * headroom_hip.h -- C ABI of the MI355X-native loudness / true-peak meter.
 *
 * Drop-in boundary for the measurement step of M-Igashi/headroom's analyzer.  The reference has
 * no FFI for this path: src/analyzer.rs:253-267 spawns `ffmpeg -af loudnorm=print_format=json`
 * and src/analyzer.rs:273-280 parses `input_i` / `input_tp` out of its stderr.  A maintainer
 * replaces that process boundary with the entry points below (the Rust `extern "C"` block is
 * shown in INTEGRATION.md).  Plain pointers and sizes only; no exceptions cross this boundary.
 *
 * Library: headroom_amd/libheadroom_hip.so (hipcc --offload-arch=gfx950).
 */
#ifndef HEADROOM_HIP_H
#define HEADROOM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HR_MAX_CH 8
#define HR_MAX_SAMPLE_RATE 768000u /* higher rates: HR_ERR_RATE (per-rate tables are O(rate); a crafted header cannot ask for more) */

/* error codes (0 = ok, < 0 = error); also used in hr_result.status */
enum {
    HR_OK = 0,
    HR_ERR_INVALID = -1,       /* NULL / malformed argument                                  */
    HR_ERR_NO_DEVICE = -2,     /* no usable gfx950 device (analogue of check_ffmpeg failing) */
    HR_ERR_HIP = -3,           /* a HIP runtime call failed                                  */
    HR_ERR_OOM = -4,           /* device or pinned allocation failed                         */
    HR_ERR_RATE = -5,          /* sample rate not supported by the selected true-peak mode   */
    HR_ERR_FORMAT = -6,        /* unknown sample format / channel count                      */
    HR_ERR_TOO_LARGE = -7,     /* input exceeds a stated limit of the entry point            */
    HR_ERR_NOT_READY = -8      /* hr_analyze_device_noalloc: the rate's device tables are not resident yet */
};

/* interleaved PCM sample formats (what ffmpeg would decode `-map 0:a:0` to) */
enum { HR_FMT_F32 = 0, HR_FMT_S16 = 1, HR_FMT_S24 = 2 /* packed 3-byte LE */, HR_FMT_S32 = 3, HR_FMT_F64 = 4 };

/* gating: which FFmpeg code path is reproduced */
enum {
    HR_GATE_EBUR128 = 0,  /* libavfilter/f_ebur128.c: 8001 x 0.01 LU histogram  (default)     */
    HR_GATE_EXACT = 1,    /* ITU-R BS.1770-4 text, no histogram                               */
    HR_GATE_LOUDNORM = 2  /* libavfilter/ebur128.c (af_loudnorm): 1000 x 0.1 LU histogram     */
};

/* Known divergence from f_ebur128 at sample rates that are not a multiple of 10 Hz (11 025 Hz; none of the usual
 * rates otherwise): FFmpeg ticks every fs/10 samples but its 400 ms / 3 s rings hold fs*4/10 / fs*3 samples, i.e. four
 * hops plus fs*4/10 - 4*(fs/10) <= 3 samples, and its first block appears one tick later.  The meter takes a block as
 * exactly four hops (normalised by fs*4/10): one block more, each block 2 samples of 4410 short at 11 025 Hz (< 0.005 LU
 * on steady material).  tests/test_oracle_kat.py::test_block_count_at_rates_not_divisible_by_ten pins this. */

/* true peak: which interpolator */
enum {
    HR_TP_SWR192K = 0, /* libswresample defaults to 192 kHz, as f_ebur128 peak=true (default) */
    HR_TP_BS1770 = 1,  /* ITU-R BS.1770-4 Annex 2, 4 x 12-tap polyphase FIR                   */
    HR_TP_SAMPLE = 2   /* sample peak only                                                    */
};

/* hr_track.flags */
enum {
    HR_TRACK_DEVICE_PCM = 1u /* pcm is a device pointer on the context's GPU (already in HBM) */
};

typedef struct hr_ctx hr_ctx;

typedef struct hr_config {
    int32_t device;     /* HIP device ordinal                                               */
    int32_t gate_mode;  /* HR_GATE_*                                                        */
    int32_t tp_mode;    /* HR_TP_*                                                          */
    int32_t engine;     /* 0 = default (fastest available), 1 = force the generic kernels;
                           2 / 3 / 4 = the fast kernels with rows of a whole hop only / of a fraction of a hop wherever
                           a track allows it / of a fraction of a hop for the hops that do not fill a wave (0 chooses
                           per track and launch; same results, for tests and timing); 5 = as 0, but 5.1 / 7.1 tracks are
                           split into stereo planes first instead of being read directly (the path of rounds 1-2; same
                           results bit for bit, for tests and A/B timing); 6 = as 0, but a 5.1 track read directly keeps
                           the three-wave workgroup where 0 takes the four-wave one (same results bit for bit); 7 = as 0, but no
                           row of a stereo f32 track gets a piece grid shifted onto cache lines (44.1 kHz and its multiples;
                           same counts and peaks, loudness equal to the 14th digit) */
    void *stream;       /* hipStream_t to run on, or NULL for a stream owned by the context */
    uint64_t stage_bytes; /* device staging for host-resident PCM per launch group; 0 = default (1 GiB).  A batch that
                             needs more is measured group by group, the copies of the next group (on a stream of
                             their own, into a second buffer of this size) running beside the kernels of this one */
} hr_config;

/* One decoded audio stream (the first audio stream of a file: `-map 0:a:0`, analyzer.rs:258). */
typedef struct hr_track {
    const void *pcm;        /* interleaved samples, host (pageable or pinned) or device     */
    uint64_t frames;        /* samples per channel                                          */
    uint32_t sample_rate;   /* Hz                                                           */
    uint16_t channels;      /* 1..HR_MAX_CH                                                 */
    uint16_t fmt;           /* HR_FMT_*                                                     */
    const float *ch_weight; /* host pointer, len = channels; 0 excludes a channel from loudness
                               (never from true peak); NULL = FFmpeg's default layout weights */
    uint32_t flags;         /* HR_TRACK_*                                                   */
    uint32_t reserved;
} hr_track;

/* What analyzer.rs:273-280 reads (`input_i`, `input_tp`) plus the meter's side outputs. */
typedef struct hr_result {
    double integrated_lufs;    /* -INFINITY when no block passes the -70 LUFS gate (silence);
                                  the caller applies analyzer.rs:284 to reject it             */
    double true_peak_dbtp;     /* 20*log10(max over channels); -INFINITY for digital silence  */
    double rel_threshold_lufs; /* relative gate                                               */
    uint64_t n_blocks;         /* 400 ms blocks                                               */
    uint64_t n_gated;          /* blocks above the absolute gate                              */
    double peak_per_ch[HR_MAX_CH]; /* linear true peak per channel (all channels, LFE too)    */
    int32_t status;            /* HR_OK or the error that made this track fail                */
    int32_t reserved;
    /* The other numbers FFmpeg prints for a file (loudnorm JSON `input_lra`, `input_thresh` = rel_threshold_lufs;
     * the ebur128 summary's "Loudness range" block), which src/analyzer.rs:102-106 leaves unread: SURVEY.md 8f.
     * hr_analyze_batch and the time-shard entry points fill them (NaN from the latter when a shard other than the last
     * is shorter than 32 hops and cannot hand the energies of 29 hops to its neighbour). */
    double loudness_range_lu;  /* LRA; 0 when no short-term block passes the gates                     */
    double lra_low_lufs;       /* loudness at the 10 % point (NaN without a block)                     */
    double lra_high_lufs;      /* loudness at the 95 % point                                           */
    double lra_threshold_lufs; /* relative gate of the range: short-term mean - 20 LU                  */
    double max_momentary_lufs; /* maximum 400 ms loudness (-INFINITY without a block)                  */
    double max_shortterm_lufs; /* maximum 3 s loudness                                                 */
} hr_result;

/* Replaces check_ffmpeg() (src/analyzer.rs:348-354, called at src/cli.rs:30): number of usable
 * devices, 0 when there is none. */
int hr_device_count(void);

/* Creates the per-GPU context (device tables, workspace, stream).  HR_ERR_NO_DEVICE when the HIP
 * runtime finds no device: the library never falls back to a CPU path. */
int hr_init(const hr_config *cfg, hr_ctx **out);
void hr_destroy(hr_ctx *ctx);

/* Pinned host memory for decoded PCM (fast H2D). */
void *hr_alloc_pinned(size_t bytes);
void hr_free_pinned(void *p);

/* Replaces the body of analyze_file_with_target up to the two parsed numbers
 * (src/analyzer.rs:253-280) for a whole batch, i.e. the work of analyze_files' par_iter
 * (src/cli.rs:403-435): results are in input order, a bad track sets out[i].status and does not
 * fail the batch (src/cli.rs:419-430).  Thread-safe per context (serialised internally). */
int hr_analyze_batch(hr_ctx *ctx, const hr_track *tracks, size_t n, hr_result *out);

/* hr_analyze_batch for ONE track that is already in HBM, without any device or pinned allocation.  No counterpart in the
 * reference: it is for callers that cannot allocate device memory -- tight serving loops, harnesses that forbid hipMalloc,
 * processes whose memory is fully budgeted.  The entry point calls no hipMalloc / hipFree / hipHostMalloc and no library
 * routine that does: all device workspace (zero-state records, hop sums, block powers, histograms, candidate lists, work
 * items) is carved from ONE static device array of 16 MiB inside the library, which every GPU the library runs on holds
 * until the process ends.  It is NOT a graph-capturable call: like hr_analyze_batch it copies its few hundred bytes of
 * results back to the host, synchronises the context's stream before it returns, and uses ordinary host containers
 * (std::vector) for its launch descriptors.
 * Requirements: track->flags has HR_TRACK_DEVICE_PCM (else HR_ERR_INVALID); hops x channels, hops = frames /
 * (sample_rate / 10), at most HR_NOALLOC_MAX_HOP_CHANNELS (stereo: 54 minutes), and a workspace estimate -- which also
 * grows with frames / 8192, so very high sample rates reach it first -- that fits the array, both checked before anything
 * is queued (else HR_ERR_TOO_LARGE); the context has already seen this sample rate in an ordinary call, which is when the
 * rate's tables are allocated (else HR_ERR_NOT_READY).  Tracks that hr_analyze_batch would first split into stereo planes
 * (mono, 5.1, 7.1) take the generic kernels here: same results, slower.  Calls on one GPU share that GPU's array (it keeps
 * the last call's contents; everything that must start at zero is cleared per call) and are serialised by a per-GPU lock
 * inside the library, which is released only once the context's streams have drained; results are those of
 * hr_analyze_batch. */
#define HR_NOALLOC_MAX_HOP_CHANNELS 65536u
int hr_analyze_device_noalloc(hr_ctx *ctx, const hr_track *track, hr_result *out);

/* Debug / parity entry: per-hop K-weighted energies H[hop*channels + ch] and 400 ms block powers
 * (f_ebur128.c power_400, 1e-12 floor included) of one track.  Either output may be NULL.
 * No counterpart in the reference; exists for the 1e-6 block-energy parity check. */
int hr_block_energies(hr_ctx *ctx, const hr_track *track, double *hop_sums, uint64_t cap_hops,
                      double *block_powers, uint64_t cap_blocks, uint64_t *n_hops, uint64_t *n_blocks);

/* One time shard of a long stream (BASELINE config 5: one stream time-sharded over the GPUs of a
 * node, one process per GPU).  The stream is cut at multiples of the hop (sample_rate/10); every
 * shard but the last holds a whole number of hops and at least 6 of them.  Each rank runs pass1
 * on its shard, hands its hr_shard_edge to the right neighbour, runs pass2 with the edge it got
 * from the left, and the hr_shard_partial records are all-reduced (sum; max for the peaks) before
 * hr_shard_finish.  HR_GATE_EXACT has no histogram to reduce: its relative gate needs the all-reduced sums first, so
 * there the ranks run hr_shard_pass3 with the reduced record, all-reduce (sum) its sum_gated / n_gated into the
 * reduced record, and then call hr_shard_finish (two small all-reduces).
 * A context holds ONE open shard: pass 2 reads the workspace pass 1 left in the context, so any other call on that
 * context in between (hr_analyze_batch, hr_block_energies, hr_apply_gain, hr_loudnorm_first_pass, another pass 1, ...)
 * closes the shard and its hr_shard_pass2 then returns HR_ERR_INVALID.
 * No counterpart in the reference: ffmpeg handles a file strictly sequentially. */
typedef struct hr_shard_edge {
    /* zero-state records of the shard's last 6 hops, oldest first, per channel:
       {energy, 4 correction dot products, 4 end-state values} */
    double rec[6][HR_MAX_CH][9];
    double tail[16][HR_MAX_CH]; /* last 16 frames: FIR halo and filter input history */
    uint64_t n_hops;
    /* energies of the shard's last 29 hops, oldest first: the right neighbour's first 3 s windows reach back that far */
    double hop_tail[29][HR_MAX_CH];
    int32_t range_ok;           /* hop_tail is complete (the shard has the 32 hops that takes; 29 for the first) */
    int32_t reserved;
} hr_shard_edge;

typedef struct hr_shard_partial {
    uint64_t hist[8001];   /* gating histogram counts of this shard's blocks  */
    double sum_kept;       /* sum of block powers above the absolute gate     */
    uint64_t n_kept;
    uint64_t n_blocks;
    double peak_per_ch[HR_MAX_CH];
    /* loudness range and maxima: the 3 s windows that END in this shard (sum / max like the fields above) */
    uint64_t st_hist[8001];
    double st_sum_kept;
    uint64_t st_n_kept;
    uint64_t n_shortterm;
    double max_m_power, max_s_power; /* reduce with max */
    int32_t range_ok;                /* reduce with min: 0 when a shard was too short to hand its 29 hop energies on */
    int32_t reserved;
    /* HR_GATE_EXACT only, written by hr_shard_pass3: this shard's blocks above both gates (reduce with sum) */
    double sum_gated;
    uint64_t n_gated;
} hr_shard_partial;

/* The first 16 frames of a shard as doubles, head[16][HR_MAX_CH]: what its left neighbour needs as
 * right_head. */
int hr_shard_head(hr_ctx *ctx, const hr_track *shard, double *head);
/* Phase 1: K-weighting of every hop from zero state + true peak.  left_tail / right_head are the 16
 * frames before / after the shard ([16][HR_MAX_CH] doubles: the left neighbour's hr_shard_edge.tail
 * and the right neighbour's hr_shard_head; NULL at the stream's ends, is_first / is_last).  Writes
 * this shard's right edge.  first_hop: index of the shard's first hop in the whole stream (first frame / hop length;
 * only HR_GATE_LOUDNORM's loudness range reads it: libebur128 takes its range blocks on a 1 s grid). */
int hr_shard_pass1(hr_ctx *ctx, const hr_track *shard, const double *left_tail, const double *right_head,
                   int is_first, int is_last, uint64_t first_hop, hr_shard_edge *right_edge_out);
/* Phase 2: IIR carry hand-off from the left neighbour's edge (NULL for the first shard), the three
 * boundary blocks, and this shard's gating partial. */
int hr_shard_pass2(hr_ctx *ctx, const hr_shard_edge *left_edge, hr_shard_partial *partial_out);
/* HR_GATE_EXACT only (HR_ERR_INVALID in the other modes), between pass 2 and finish: `reduced` carries the all-reduced
 * sum_kept / n_kept of pass 2, which set the relative gate (-10 LU below their mean); writes this shard's sum_gated /
 * n_gated into partial_out (all its other fields zero).  The shard stays open from pass 2 until this call. */
int hr_shard_pass3(hr_ctx *ctx, const hr_shard_partial *reduced, hr_shard_partial *partial_out);
/* Gating from the all-reduced partial. */
int hr_shard_finish(hr_ctx *ctx, const hr_shard_partial *reduced, int channels, hr_result *out);

/* The same exchange with the records RESIDENT ON THE DEVICE (round 3; every gating mode but HR_GATE_EXACT, whose two dependent
 * reductions stay with the entry points above: HR_ERR_INVALID here).  The edge and the partial are written into / read from
 * device buffers of the caller -- images of hr_shard_edge / hr_shard_partial, sizeof() bytes each -- so that the caller's
 * communicator moves them GPU to GPU with no host round trip: ncclSend / ncclRecv of the edge to the right neighbour, ONE
 * ncclAllGather of the partial (n images back to back), then hr_shard_finish_gathered, which reduces the images in rank
 * order on the device (reproducible sums) and finishes.  A rank that sits the passes out contributes an all-zero image with
 * range_ok = 1.  hr_shard_pass1_device / hr_shard_pass2_device / hr_shard_redo_device only QUEUE work on the context's stream
 * (hr_config.stream, or the context's own) and return without waiting for it: issue the collectives on that stream, or
 * order yours behind it.  hr_shard_finish_gathered synchronises the stream once and returns
 *   HR_OK           out is final;
 *   HR_SHARD_AGAIN  some shard's true-peak screen left work for its exact redo (sustained tones; the image's `reserved`
 *                   says so, which is how every rank learns it from the same gathered bytes): every rank calls
 *                   hr_shard_redo_device on its partial buffer, the partials are gathered again, and
 *                   hr_shard_finish_gathered is called again -- it returns HR_OK then.
 * The shard's PCM must stay valid until the final hr_shard_finish_gathered.  INTEGRATION.md section 4 shows the sequence with
 * ncclAllGather; headroom_amd/sharded.py analyze_stream_shard_device runs it over torch.distributed. */
#define HR_SHARD_AGAIN 1
int hr_shard_pass1_device(hr_ctx *ctx, const hr_track *shard, const double *left_tail, const double *right_head,
                          int is_first, int is_last, uint64_t first_hop, void *edge_dev_out);
int hr_shard_pass2_device(hr_ctx *ctx, const void *left_edge_dev, void *partial_dev_out);
int hr_shard_redo_device(hr_ctx *ctx, void *partial_dev_inout);
int hr_shard_finish_gathered(hr_ctx *ctx, const void *partials_dev, int n, int channels, hr_result *out);

/* The same inside ONE process: n_ctx contexts (one per GPU; same modes in all of them) each take one time shard of `track`.  Host PCM, or
 * device PCM when every context sits on the GPU that holds it.  Streams too short to cut (fewer than six
 * hops per shard) use fewer contexts.  Contexts on distinct GPUs exchange their records through RCCL inside the library
 * (ncclSend / ncclRecv of the edges, one ncclAllGather of the gating partials, over xGMI; librccl is opened at run time);
 * contexts that share a GPU, and HR_GATE_EXACT, take device copies / the host records instead.  No counterpart in the
 * reference. */
int hr_analyze_stream_sharded(hr_ctx *const *ctxs, int n_ctx, const hr_track *track, hr_result *out);

/* Deterministic synthetic PCM written straight into device memory (bench / tests):
 * f32 interleaved, `frames` x `channels`.  Same bits as the oracle's generator. */
int hr_synth_fill(hr_ctx *ctx, void *device_pcm, uint64_t seed, uint32_t sample_rate,
                  uint32_t channels, uint64_t frames);

/* Device time of the last hr_analyze_batch on this context, in milliseconds, measured with HIP
 * events on the library's own stream: total of the compute kernels, and of the dominant (fused
 * K-weight + true-peak) kernel alone. */
int hr_last_timing(hr_ctx *ctx, float *total_ms, float *main_kernel_ms);

const char *hr_strerror(int code);

/* ---- PCM container ingest: WAV and AIFF need no decoder -------------------------------
 * The reference hands every file to `ffmpeg -i <path> -map 0:a:0` (src/analyzer.rs:253-259) and lists wav, aiff
 * and aif among its lossless extensions (src/scanner.rs:6).  For these containers the first audio stream is
 * plain PCM: RIFF/WAVE (PCM, IEEE float, WAVE_FORMAT_EXTENSIBLE with its channel mask) and AIFF / AIFF-C
 * (NONE, twos, sowt, fl32, fl64).  Anything compressed returns HR_ERR_FORMAT and stays with the caller's decoder. */
typedef struct hr_pcm_info {
    uint64_t frames;       /* samples per channel                                   */
    uint64_t data_offset;  /* byte offset of the first sample in the file           */
    uint32_t sample_rate;
    uint32_t channel_mask; /* WAVE_FORMAT_EXTENSIBLE dwChannelMask, 0 when absent   */
    uint16_t channels;
    uint16_t fmt;          /* HR_FMT_*                                              */
    int32_t big_endian;    /* samples need a byte swap (AIFF) before the meter      */
} hr_pcm_info;

/* Parses the container header found in file_bytes[0..n) (the whole file or a prefix that holds the header
 * chunks).  Host-only: needs no GPU. */
int hr_pcm_probe(const void *file_bytes, size_t n, hr_pcm_info *info);
/* Reads `path` into pinned host memory (little endian, ready for hr_analyze_batch), fills `track` (its ch_weight
 * points at `weights`, HR_MAX_CH floats owned by the caller: FFmpeg's layout weights, from the channel mask when
 * the file has one).  Free *pcm_out with hr_free_pinned. */
int hr_pcm_load_file(const char *path, void **pcm_out, hr_track *track, float *weights);

/* ---- host-side mirror of the analyzer's scalar logic (src/analyzer.rs:157-187, :304-326) ---- */

enum { /* GainMethod, src/analyzer.rs:36-50 */
    HR_GAIN_FFMPEG_LOSSLESS = 0, HR_GAIN_MP3_LOSSLESS = 1, HR_GAIN_MP3_REENCODE = 2,
    HR_GAIN_AAC_LOSSLESS = 3, HR_GAIN_AAC_REENCODE = 4, HR_GAIN_NONE = 5
};
enum { HR_KIND_LOSSLESS = 0, HR_KIND_MP3 = 1, HR_KIND_AAC = 2 }; /* scanner::is_mp3 / is_aac */

typedef struct hr_tp_target { /* TpTargetMode, src/analyzer.rs:157-164 */
    int32_t split;  /* 0 = Uniform(high), 1 = SplitBitrate(high, low) */
    double high, low;
} hr_tp_target;

typedef struct hr_analysis { /* AudioAnalysis minus the path strings, src/analyzer.rs:52-65 */
    double input_i, input_tp;
    int32_t has_bitrate; uint32_t bitrate_kbps;
    double target_tp, headroom;
    int32_t gain_method;
    double effective_gain;
    int32_t lossless_gain_steps;
    int32_t status; /* HR_OK, or HR_ERR_INVALID for a non-finite measurement (analyzer.rs:284) */
} hr_analysis;

/* src/analyzer.rs:284-326 applied to one measurement, on exactly the two numbers it is given. */
int hr_gain_decision(double input_i, double input_tp, int kind, int has_bitrate, uint32_t bitrate_kbps,
                     const hr_tp_target *mode, hr_analysis *out);

/* The two numbers as the reference holds them after src/analyzer.rs:273-280: loudnorm prints `input_i` / `input_tp`
 * with "%.2f" (src/analyzer.rs:102-106 receives them as strings) and the analyzer parses that text, so its headroom,
 * its MIN_EFFECTIVE_GAIN (0.05 dB, :307) and GAIN_STEP (1.5 dB, :313) comparisons and its report all see the
 * measurement rounded to two decimals.  A drop-in caller must feed the decision THESE values, not the full-precision
 * fields of hr_result: tp = -0.5451 dBTP reads "-0.55" there (headroom 0.05: processed) but 0.0451 unrounded (skipped).
 * "-inf" (silence) stays -inf and is then rejected by the finite check.  Either output may be NULL. */
int hr_loudnorm_text_values(const hr_result *r, double *input_i, double *input_tp);

/* hr_loudnorm_text_values followed by hr_gain_decision: the whole of src/analyzer.rs:273-326 for one measured track.
 * A track whose measurement failed (r->status != HR_OK) yields that status (the Err arm, src/cli.rs:419-430). */
int hr_gain_decision_for_result(const hr_result *r, int kind, int has_bitrate, uint32_t bitrate_kbps,
                                const hr_tp_target *mode, hr_analysis *out);

/* The measurement the reference literally asks FFmpeg for: the first pass of `-af loudnorm=print_format=json`
 * (src/analyzer.rs:253-267; `input_i`, `input_tp`, `input_lra`, `input_thresh` of its JSON).  af_loudnorm negotiates
 * 192 kHz f64 input, so every track is first resampled to 192 kHz with libswresample's defaults (flushed at the
 * end) and then measured AT 192 kHz: K-weighting for fs = 192000, libavfilter/ebur128.c's 1000 x 0.1 LU histogram
 * gating with its channel map by index (L R C LFE Ls Rs: LFE and channels beyond six unweighted), and the sample
 * peak of the resampled signal as `input_tp`.  The context's gate / true-peak modes are not used.  Same batch
 * semantics as hr_analyze_batch.  Rates above 192 kHz: HR_ERR_RATE.  Ten times the arithmetic of hr_analyze_batch per
 * input sample (every 192 kHz output is a 32-tap f64 dot product and a K-weighting step): about a tenth of its rate.
 * hr_config.engine = 1 takes the older path (the whole resampled stream in device memory, generic kernels). */
int hr_loudnorm_first_pass(hr_ctx *ctx, const hr_track *tracks, size_t n, hr_result *out);

/* Lossless gain for PCM files, replacing `ffmpeg -i in -af volume={gain}dB -c:a pcm_s24le|pcm_s24be out`
 * (apply_gain_ffmpeg, src/processor.rs:53-87; WAV -> pcm_s24le, AIFF -> pcm_s24be): every sample of `track` times
 * 10^(gain_db/20) in the arithmetic of that FFmpeg chain (f32 volume filter, round-to-nearest-even to 32 bits,
 * saturating, upper three bytes kept), written to `out` as packed 24-bit (frames * channels * 3 bytes; big endian
 * when out_big_endian != 0).  `out` is a device pointer when the track carries HR_TRACK_DEVICE_PCM, else a host
 * pointer.  *clipped (may be NULL) receives the number of samples that saturated. */
int hr_apply_gain(hr_ctx *ctx, const hr_track *track, double gain_db, int out_big_endian, void *out, uint64_t *clipped);

/* The same for a WAV / AIFF / AIFF-C file, in place, as apply_gain_ffmpeg does (src/processor.rs:53-87): the file is
 * rewritten as 24-bit PCM (WAV little endian, AIFF big endian) into `<stem>.tmp.<ext>` beside it and renamed over
 * the original; every chunk other than the format / sound chunks (bext, LIST, ID3, markers ...) is copied through in
 * place.  Files whose result would pass 4 GiB: HR_ERR_TOO_LARGE (no RF64).  Compressed files: HR_ERR_FORMAT. */
int hr_pcm_gain_file(hr_ctx *ctx, const char *path, double gain_db, uint64_t *clipped);

/* One line of the CSV report, src/report.rs:29-59: the header record, and the record of one analysed file
 * (labels: GainMethod::format_label / method_label, src/analyzer.rs:80-100; numbers with one decimal, headroom and
 * gain signed; fields quoted as the csv crate does by default).  Return the length written (without the NUL), or
 * HR_ERR_TOO_LARGE when `cap` is too small. */
int hr_report_csv_header(char *buf, size_t cap);
int hr_report_csv_row(const char *filename, const hr_analysis *a, char *buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
