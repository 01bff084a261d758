/*
 * hr_oracle.h -- CPU oracle for the headroom loudness / true-peak measurement path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker.  The product (headroom_amd/libheadroom_hip.so) never links or calls it.
 *
 * PARITY UNPINNED.  The reference (M-Igashi/headroom, src/analyzer.rs:253-267) holds no
 * loudness arithmetic: it spawns `ffmpeg ... -af loudnorm=print_format=json` and parses two
 * strings.  The arithmetic lives in FFmpeg (libavfilter/f_ebur128.c, libavfilter/ebur128.c,
 * libavfilter/af_loudnorm.c, libswresample/resample.c), a third-party dependency that is
 * absent from /root/reference, unpinned by the reference (any `ffmpeg` on PATH), and not
 * installed in this image.  The reference's own tests (src/analyzer.rs:356-459) pin no
 * numeric result.  This file is therefore a restatement of the *published* algorithms
 * (ITU-R BS.1770-4, EBU Tech 3341, and FFmpeg's documented choices: see SURVEY.md App. A)
 * pinned only by analytic known-answer tests (tests/test_oracle_kat.py), never by FFmpeg
 * output.
 *
 * Arithmetic convention (so that a second implementation can be bit-identical): every
 * recurrence and dot product below is evaluated in IEEE-754 binary64 with fused
 * multiply-add in exactly the order written in hr_oracle.c; the library is compiled with
 * -ffp-contract=off so the compiler adds no contraction of its own.
 */
#ifndef HR_ORACLE_H
#define HR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* PCM sample formats (interleaved). Same numbering as include/headroom_hip.h. */
enum { HRO_FMT_F32 = 0, HRO_FMT_S16 = 1, HRO_FMT_S24 = 2, HRO_FMT_S32 = 3, HRO_FMT_F64 = 4 };

/* Gating modes. */
enum {
    HRO_GATE_EBUR128 = 0,  /* f_ebur128.c: 8001 x 0.01 LU histogram, exact rel-threshold sum  */
    HRO_GATE_EXACT = 1,    /* BS.1770-4 text: two passes over the exact block powers          */
    HRO_GATE_LOUDNORM = 2  /* libavfilter/ebur128.c: 1000 x 0.1 LU histogram only             */
};

/* True-peak modes. */
enum {
    HRO_TP_SWR192K = 0,  /* libswresample defaults to 192 kHz (what f_ebur128 peak=true does) */
    HRO_TP_BS1770 = 1,   /* BS.1770-4 Annex 2 48-tap / 4-phase FIR                            */
    HRO_TP_SAMPLE = 2    /* plain max |x| (no oversampling)                                   */
};

#define HRO_MAX_CH 8
#define HRO_SWR_TAPS 32

typedef struct {
    double pre_b[3], pre_a[3]; /* a[0] unused (=1) */
    double rlb_b[3], rlb_a[3];
} hro_kcoef;

typedef struct {
    const void *pcm;       /* interleaved samples */
    uint64_t frames;
    uint32_t sample_rate;
    uint16_t channels;
    uint16_t fmt;
    const float *ch_weight; /* len = channels; NULL => hro_default_weights() */
} hro_track;

typedef struct {
    double integrated_lufs;    /* -inf when no block passes the absolute gate */
    double true_peak_dbtp;     /* -inf for digital silence */
    double rel_threshold_lufs; /* relative gate, LUFS */
    uint64_t n_blocks;         /* 400 ms blocks produced */
    uint64_t n_gated;          /* blocks above the absolute gate */
    double peak_per_ch[HRO_MAX_CH]; /* linear true peak per channel */
    int status;
} hro_result;

/* K-weighting coefficients: FFmpeg f_ebur128.c config_audio_input / libebur128 ebur128_init_filter. */
void hro_kweight_coeffs(double fs, hro_kcoef *k);

/* BS.1770 channel weights for FFmpeg's default layouts (1, 2 ch: all 1.0; 6 ch 5.1: L R C LFE Ls Rs
 * = 1,1,1,0,1.41,1.41).  Other counts: all 1.0. */
void hro_default_weights(int channels, float *w);

/* Sample as double, FFmpeg conversion rules (s16/32768, s24/2^23, s32/2^31, float as is). */
double hro_sample(const void *pcm, int fmt, uint64_t idx);

/* Hop length in frames: fs/10 (f_ebur128.c) */
uint32_t hro_hop(uint32_t fs);

/* Per-hop, per-channel sums of squared K-weighted samples: H[k*channels + c], k < frames/hop.
 * Returns the number of hops. */
uint64_t hro_hop_sums(const hro_track *t, double *H, uint64_t cap_hops);

/* 400 ms block powers from hop sums: P[j] = (1e-12 + sum_c w_c*(H[j]+H[j+1]+H[j+2]+H[j+3]))/(fs*4/10).
 * Returns n_blocks = max(0, n_hops-3). */
uint64_t hro_block_powers(const double *H, uint64_t n_hops, int channels, const float *w,
                          uint32_t fs, double *P);

/* Gating.  hist may be NULL; otherwise receives the histogram counts (8001 or 1000 entries). */
void hro_gate(const double *P, uint64_t n_blocks, int mode, hro_result *r, uint64_t *hist);

/* Loudness range and the maxima of the momentary (400 ms) and short-term (3 s) loudness.
 *   gate modes EBUR128 / EXACT: f_ebur128.c -- a short-term power every hop once 3 s are filled,
 *     (1e-12 + sum_c w_c * sum of 30 hops) / (3 fs); those >= -70 LUFS enter the i3000 histogram through
 *     gate_update(.., LRA_GATE_THRES = -20); LRA = loudness of the bin holding the 95 % point minus that
 *     of the 10 % point among the bins at or above the relative gate (the filter's own counting rule).
 *   gate mode LOUDNORM: libavfilter/ebur128.c -- ff_ebur128_loudness_range on the 1000-bin histogram of
 *     the short-term energies taken at 3 s and then every 1 s (what loudnorm prints as input_lra).
 * lra is 0 when no short-term block passes; lra_low / lra_high / lra_threshold are NaN then.
 * max_momentary / max_shortterm: LUFS, -inf without a block.  (Neither filter prints them; EBU Tech 3341
 * defines them for a meter.  They use the mode's block power formula at every hop.) */
typedef struct {
    double lra, lra_low, lra_high, lra_threshold;
    double max_momentary, max_shortterm;
    uint64_t n_shortterm; /* short-term values produced (one per hop from the 30th) */
} hro_range;
void hro_range_stats(const double *H, uint64_t n_hops, int channels, const float *w, uint32_t fs,
                     int gate_mode, hro_range *out);

/* libswresample filter bank (filter_size 32, kaiser beta 9, exact_rational): returns phase_count
 * (0 if the rate pair is unsupported or needs no resampling) and the per-output input step
 * (dst_incr_div); taps must hold (phase_count+1)*32 doubles when not NULL. */
/* libswresample/resample.c resample_init(): what the resampling loop needs besides the bank */
typedef struct hro_swr_plan_t {
    int phases;           /* filter phases per input sample: out/gcd(in,out) when <= 1024 (exact_rational), else 1024 */
    int linear;           /* the loop interpolates between neighbouring phases (dst_incr % src_incr != 0) */
    uint32_t d0, s0;      /* dst_incr / src_incr in lowest terms: output n sits at floor(n*d0/s0) phase units */
    uint32_t frac_scale;  /* both are doubled until one reaches 1 << 20: frac = ((n*d0) % s0) * frac_scale */
    double inv_src_incr;  /* 1.0 / (s0 * frac_scale), as resample_linear computes it */
} hro_swr_plan_t;
int hro_swr_plan(uint32_t in_rate, uint32_t out_rate, hro_swr_plan_t *plan);
int hro_swr_bank(uint32_t in_rate, uint32_t out_rate, int *in_step, double *taps);

/* BS.1770-4 Annex 2 coefficients: 4 phases x 12 taps. */
const double *hro_bs1770_taps(void);

/* True peak per channel (linear). flush_end: 0 = f_ebur128 behaviour (no flush), 1 = aresample flush. */
void hro_true_peak(const hro_track *t, int tp_mode, int flush_end, double *peak_per_ch);

/* Whole measurement in the hop-sum formulation (what the GPU path is compared against). */
int hro_measure(const hro_track *t, int gate_mode, int tp_mode, hro_result *r);

/* Literal f_ebur128.c restatement (per-sample ring cache + running sum, gate_update at every hop).
 * Used only to cross-check the hop-sum identity. block_powers may be NULL. */
int hro_ebur128_literal(const hro_track *t, hro_result *r, double *block_powers, uint64_t cap);

/* af_loudnorm first-pass measurement: aresample to 192 kHz (start reflection, end flush), then
 * libavfilter/ebur128.c (4th-order DF2 filter, 1000-bin histogram), input_tp = sample peak of the
 * resampled signal. */
int hro_loudnorm_measure(const hro_track *t, hro_result *r);

/* Lossless gain as `ffmpeg -af volume={g}dB -c:a pcm_s24le|pcm_s24be` computes it (src/processor.rs:53-87):
 * sample -> f32 (libswresample audioconvert), times (float)10^(g/20) in f32 (af_volume.c, precision float),
 * av_clipl_int32(llrintf(y * 2^31)), upper three bytes written.  out: frames*channels*3 bytes.  Returns the
 * number of samples that saturated. */
uint64_t hro_apply_gain(const hro_track *t, double gain_db, int big_endian, uint8_t *out);

/* Deterministic synthetic PCM (f32 interleaved), SURVEY.md 8(d): noise + sine + quiet intro/outro +
 * inter-sample-peak bursts.  Same generator as the device kernel hr_synth_fill. */
void hro_synth_track(uint64_t seed, uint32_t fs, int channels, uint64_t frames, float *out);

#ifdef __cplusplus
}
#endif
#endif
