/*
 * hr_oracle.c -- CPU oracle (test infrastructure; see hr_oracle.h for the rules of use and the
 * "PARITY UNPINNED" statement).
 *
 * What each function restates, with the reference call site that makes it part of the hot path:
 *   reference: src/analyzer.rs:253-267 spawns ffmpeg; :273-290 parses input_i / input_tp.
 *   FFmpeg (not in /root/reference; public source, restated from the published algorithm):
 *     hro_kweight_coeffs   <- libavfilter/f_ebur128.c config_audio_input(), ebur128.c ebur128_init_filter()
 *     hro_hop_sums         <- f_ebur128.c filter_frame() FILTER() macro, i400 accumulation
 *     hro_block_powers     <- f_ebur128.c COMPUTE_LOUDNESS(); ebur128.c ebur128_calc_gating_block()
 *     hro_gate             <- f_ebur128.c gate_update() + integration loop; ebur128.c ebur128_gated_loudness()
 *     hro_swr_bank         <- libswresample/resample.c build_filter()/resample_init(), defaults of options.c
 *     hro_true_peak        <- f_ebur128.c PEAK_MODE_TRUE_PEAKS (swr_convert to 192 kHz + fabs max);
 *                             ITU-R BS.1770-4 Annex 2 for HRO_TP_BS1770
 *     hro_loudnorm_measure <- libavfilter/af_loudnorm.c first pass (what analyzer.rs:261 really asks for)
 *
 * Compile: gcc -O2 -mfma -ffp-contract=off -fPIC -shared (see oracle/Makefile).
 */
#include "hr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------------------------ */
/* K-weighting                                                                                */
/* ------------------------------------------------------------------------------------------ */

void hro_kweight_coeffs(double fs, hro_kcoef *k)
{
    double f0 = 1681.974450955533;
    double G = 3.999843853973347;
    double Q = 0.7071752369554196;
    double K = tan(M_PI * f0 / fs);
    double Vh = pow(10.0, G / 20.0);
    double Vb = pow(Vh, 0.4996667741545416);
    double a0 = 1.0 + K / Q + K * K;

    k->pre_b[0] = (Vh + Vb * K / Q + K * K) / a0;
    k->pre_b[1] = 2.0 * (K * K - Vh) / a0;
    k->pre_b[2] = (Vh - Vb * K / Q + K * K) / a0;
    k->pre_a[0] = 1.0;
    k->pre_a[1] = 2.0 * (K * K - 1.0) / a0;
    k->pre_a[2] = (1.0 - K / Q + K * K) / a0;

    f0 = 38.13547087602444;
    Q = 0.5003270373238773;
    K = tan(M_PI * f0 / fs);
    k->rlb_b[0] = 1.0;
    k->rlb_b[1] = -2.0;
    k->rlb_b[2] = 1.0;
    k->rlb_a[0] = 1.0;
    k->rlb_a[1] = 2.0 * (K * K - 1.0) / (1.0 + K / Q + K * K);
    k->rlb_a[2] = (1.0 - K / Q + K * K) / (1.0 + K / Q + K * K);
}

void hro_default_weights(int channels, float *w)
{
    /* f_ebur128.c: LFE -> 0, back / side speakers -> 1.41, others 1.0, on FFmpeg's default layout of the channel
     * count: 2.1 = FL FR LFE; 4.0 = FL FR FC BC; 5.0 = FL FR FC BL BR; 5.1 = FL FR FC LFE BL BR;
     * 6.1 = FL FR FC LFE BC SL SR; 7.1 = FL FR FC LFE BL BR SL SR */
    static const float tab[9][8] = {
        {0},
        {1.0f},
        {1.0f, 1.0f},
        {1.0f, 1.0f, 0.0f},
        {1.0f, 1.0f, 1.0f, 1.41f},
        {1.0f, 1.0f, 1.0f, 1.41f, 1.41f},
        {1.0f, 1.0f, 1.0f, 0.0f, 1.41f, 1.41f},
        {1.0f, 1.0f, 1.0f, 0.0f, 1.41f, 1.41f, 1.41f},
        {1.0f, 1.0f, 1.0f, 0.0f, 1.41f, 1.41f, 1.41f, 1.41f},
    };
    for (int c = 0; c < channels; c++) w[c] = (channels >= 1 && channels <= 8) ? tab[channels][c] : 1.0f;
}

double hro_sample(const void *pcm, int fmt, uint64_t idx)
{
    switch (fmt) {
    case HRO_FMT_F32:
        return (double)((const float *)pcm)[idx];
    case HRO_FMT_S16:
        return (double)((const int16_t *)pcm)[idx] * (1.0 / 32768.0);
    case HRO_FMT_S24: {
        const uint8_t *p = (const uint8_t *)pcm + 3 * idx;
        int32_t v = (int32_t)((uint32_t)p[0] << 8 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 24);
        return (double)v * (1.0 / 2147483648.0); /* pcm_s24le -> s32 (<<8) -> /2^31 */
    }
    case HRO_FMT_S32:
        return (double)((const int32_t *)pcm)[idx] * (1.0 / 2147483648.0);
    case HRO_FMT_F64:
        return ((const double *)pcm)[idx];
    default:
        return 0.0;
    }
}

uint32_t hro_hop(uint32_t fs) { return fs / 10; }

uint64_t hro_hop_sums(const hro_track *t, double *H, uint64_t cap_hops)
{
    hro_kcoef k;
    const uint32_t hop = hro_hop(t->sample_rate);
    const int C = t->channels;
    uint64_t nh = hop ? t->frames / hop : 0;
    if (nh > cap_hops) nh = cap_hops;
    hro_kweight_coeffs((double)t->sample_rate, &k);
    const double pb0 = k.pre_b[0], pb1 = k.pre_b[1], pb2 = k.pre_b[2];
    const double pa1 = k.pre_a[1], pa2 = k.pre_a[2];
    const double ra1 = k.rlb_a[1], ra2 = k.rlb_a[2];

    for (int c = 0; c < C; c++) {
        double x1 = 0, x2 = 0, y1 = 0, y2 = 0, z1 = 0, z2 = 0;
        uint64_t n = 0;
        for (uint64_t h = 0; h < nh; h++) {
            double sum = 0.0;
            for (uint32_t i = 0; i < hop; i++, n++) {
                const double x0 = hro_sample(t->pcm, t->fmt, n * C + c);
                /* pre-filter (shelf), direct form I; the newest feedback term is applied last */
                double tt = pb0 * x0;
                tt = fma(pb1, x1, tt);
                tt = fma(pb2, x2, tt);
                tt = fma(-pa2, y2, tt);
                const double y0 = fma(-pa1, y1, tt);
                /* RLB high-pass, b = {1,-2,1} */
                double v = fma(-2.0, y1, y0);
                v = v + y2;
                v = fma(-ra2, z2, v);
                const double z0 = fma(-ra1, z1, v);
                sum = fma(z0, z0, sum);
                x2 = x1; x1 = x0;
                y2 = y1; y1 = y0;
                z2 = z1; z1 = z0;
            }
            H[h * C + c] = sum;
        }
    }
    return nh;
}

static uint64_t block_powers_style(const double *H, uint64_t n_hops, int C, const float *w,
                                   uint32_t fs, double *P, int style)
{
    if (n_hops < 4) return 0;
    const uint64_t nb = n_hops - 3;
    const double bins = style == 0 ? (double)((uint64_t)fs * 4 / 10)
                                   : (double)(4 * (((uint64_t)fs + 5) / 10));
    for (uint64_t j = 0; j < nb; j++) {
        double power = style == 0 ? 1e-12 : 0.0;
        for (int c = 0; c < C; c++) {
            if (w[c] == 0.0f) continue;
            const double s = ((H[j * C + c] + H[(j + 1) * C + c]) + H[(j + 2) * C + c]) +
                             H[(j + 3) * C + c];
            power += (double)w[c] * s;
        }
        P[j] = power / bins;
    }
    return nb;
}

uint64_t hro_block_powers(const double *H, uint64_t n_hops, int channels, const float *w,
                          uint32_t fs, double *P)
{
    return block_powers_style(H, n_hops, channels, w, fs, P, 0);
}

/* ------------------------------------------------------------------------------------------ */
/* Gating                                                                                     */
/* ------------------------------------------------------------------------------------------ */

#define ABS_THRES (-70.0)
#define EB_HIST_GRAIN 100
#define EB_HIST_SIZE 8001
#define LOUDNESS(energy) (-0.691 + 10.0 * log10(energy))
#define ENERGY(loudness) (pow(10.0, ((loudness) + 0.691) / 10.0))

static int clipi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static void gate_ebur128(const double *P, uint64_t nb, hro_result *r, uint64_t *hist_out)
{
    uint64_t *hist = (uint64_t *)calloc(EB_HIST_SIZE, sizeof(uint64_t));
    double sum_kept = 0.0;
    uint64_t n_kept = 0;
    double rel_thr = ABS_THRES; /* never read while n_kept == 0 */
    int gate_pos = 0;

    for (uint64_t j = 0; j < nb; j++) {
        const double loud = LOUDNESS(P[j]);
        if (loud >= ABS_THRES) {
            /* gate_update() */
            const int ipower = clipi((int)((loud - ABS_THRES) * EB_HIST_GRAIN), 0, EB_HIST_SIZE - 1);
            hist[ipower]++;
            sum_kept += P[j];
            n_kept++;
            double rt = sum_kept / (double)n_kept;
            if (!rt) rt = 1e-12;
            rel_thr = LOUDNESS(rt) + (-10.0);
            gate_pos = clipi((int)((rel_thr - ABS_THRES) * EB_HIST_GRAIN), 0, EB_HIST_SIZE - 1);
        }
    }
    r->n_blocks = nb;
    r->n_gated = n_kept;
    r->rel_threshold_lufs = n_kept ? rel_thr : -INFINITY;
    r->integrated_lufs = -INFINITY; /* f_ebur128 itself keeps its initial -70.0 here */
    if (n_kept) {
        double isum = 0.0;
        uint64_t nint = 0;
        for (int i = gate_pos; i < EB_HIST_SIZE; i++) {
            const double loud_i = (double)i / (double)EB_HIST_GRAIN + ABS_THRES;
            nint += hist[i];
            isum += (double)hist[i] * ENERGY(loud_i);
        }
        if (nint) r->integrated_lufs = LOUDNESS(isum / (double)nint);
    }
    if (hist_out) memcpy(hist_out, hist, EB_HIST_SIZE * sizeof(uint64_t));
    free(hist);
}

static void gate_exact(const double *P, uint64_t nb, hro_result *r)
{
    const double abs_e = ENERGY(ABS_THRES);
    double s = 0.0;
    uint64_t n = 0;
    for (uint64_t j = 0; j < nb; j++)
        if (P[j] > abs_e) { s += P[j]; n++; }
    r->n_blocks = nb;
    r->n_gated = n;
    r->integrated_lufs = -INFINITY;
    r->rel_threshold_lufs = -INFINITY;
    if (!n) return;
    const double rel_e = 0.1 * (s / (double)n);
    r->rel_threshold_lufs = LOUDNESS(rel_e);
    double s2 = 0.0;
    uint64_t n2 = 0;
    for (uint64_t j = 0; j < nb; j++)
        if (P[j] > abs_e && P[j] > rel_e) { s2 += P[j]; n2++; }
    if (n2) r->integrated_lufs = LOUDNESS(s2 / (double)n2);
}

/* libavfilter/ebur128.c tables */
static double ln_energies[1000];
static double ln_bounds[1001];
static int ln_init_done = 0;

static void ln_init(void)
{
    if (ln_init_done) return;
    ln_bounds[0] = pow(10.0, (-70.0 + 0.691) / 10.0);
    for (int i = 0; i < 1000; ++i) ln_energies[i] = pow(10.0, ((double)i / 10.0 - 69.95 + 0.691) / 10.0);
    for (int i = 1; i < 1001; ++i) ln_bounds[i] = pow(10.0, ((double)i / 10.0 - 70.0 + 0.691) / 10.0);
    ln_init_done = 1;
}

static size_t ln_find_index(double energy)
{
    size_t index_min = 0, index_max = 1000, index_mid;
    do {
        index_mid = (index_min + index_max) / 2;
        if (energy >= ln_bounds[index_mid]) index_min = index_mid;
        else index_max = index_mid;
    } while (index_max - index_min != 1);
    return index_min;
}

static void gate_loudnorm(const double *P, uint64_t nb, hro_result *r, uint64_t *hist_out)
{
    uint64_t hist[1000];
    memset(hist, 0, sizeof hist);
    ln_init();
    uint64_t n_kept = 0;
    for (uint64_t j = 0; j < nb; j++)
        if (P[j] >= ln_bounds[0]) { ++hist[ln_find_index(P[j])]; n_kept++; }
    r->n_blocks = nb;
    r->n_gated = n_kept;
    r->integrated_lufs = -INFINITY; /* -HUGE_VAL in libebur128 */
    r->rel_threshold_lufs = -INFINITY;
    if (hist_out) memcpy(hist_out, hist, sizeof hist);

    double gated = 0.0;
    uint64_t above = 0;
    for (int j = 0; j < 1000; ++j) { gated += (double)hist[j] * ln_energies[j]; above += hist[j]; }
    if (!above) return;
    gated /= (double)above;
    const double rel = gated * 0.1; /* RELATIVE_GATE_FACTOR = 10^(-10/10) */
    r->rel_threshold_lufs = 10.0 * (log(rel) / log(10.0)) - 0.691;
    size_t start;
    if (rel < ln_bounds[0]) start = 0;
    else {
        start = ln_find_index(rel);
        if (rel > ln_energies[start]) ++start;
    }
    gated = 0.0;
    above = 0;
    for (size_t j = start; j < 1000; ++j) { gated += (double)hist[j] * ln_energies[j]; above += hist[j]; }
    if (!above) return;
    gated /= (double)above;
    r->integrated_lufs = 10.0 * (log(gated) / log(10.0)) - 0.691;
}

void hro_gate(const double *P, uint64_t n_blocks, int mode, hro_result *r, uint64_t *hist)
{
    switch (mode) {
    case HRO_GATE_EXACT: gate_exact(P, n_blocks, r); break;
    case HRO_GATE_LOUDNORM: gate_loudnorm(P, n_blocks, r, hist); break;
    default: gate_ebur128(P, n_blocks, r, hist); break;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Lossless gain (volume filter + pcm_s24 encoder)                                            */
/* ------------------------------------------------------------------------------------------ */

static float sample_as_flt(const void *pcm, int fmt, uint64_t idx)
{
    switch (fmt) { /* libswresample/audioconvert.c CONV_FUNC(..FLT..) */
    case HRO_FMT_F32: return ((const float *)pcm)[idx];
    case HRO_FMT_S16: return (float)((const int16_t *)pcm)[idx] * (1.0f / (1 << 15));
    case HRO_FMT_S24: { /* the pcm_s24le decoder hands over s32 = v << 8 */
        const uint8_t *p = (const uint8_t *)pcm + 3 * idx;
        const int32_t v = (int32_t)((uint32_t)p[0] << 8 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 24);
        return (float)v * (1.0f / (1U << 31));
    }
    case HRO_FMT_S32: return (float)((const int32_t *)pcm)[idx] * (1.0f / (1U << 31));
    default: return (float)((const double *)pcm)[idx];
    }
}

uint64_t hro_apply_gain(const hro_track *t, double gain_db, int big_endian, uint8_t *out)
{
    const float vol = (float)pow(10.0, gain_db / 20.0);
    const uint64_t n = t->frames * t->channels;
    uint64_t clipped = 0;
    for (uint64_t i = 0; i < n; i++) {
        const float y = sample_as_flt(t->pcm, t->fmt, i) * vol; /* vector_fmul_scalar */
        const long long q = llrintf(y * (float)(1U << 31));
        int32_t s;
        if (y != y) { s = INT32_MIN; clipped++; } /* x86 llrintf(NaN) = LLONG_MIN */
        else if (q > INT32_MAX) { s = INT32_MAX; clipped++; }
        else if (q < INT32_MIN) { s = INT32_MIN; clipped++; }
        else s = (int32_t)q;
        const uint32_t u = (uint32_t)s;
        out[3 * i + (big_endian ? 2 : 0)] = (uint8_t)(u >> 8);
        out[3 * i + 1] = (uint8_t)(u >> 16);
        out[3 * i + (big_endian ? 0 : 2)] = (uint8_t)(u >> 24);
    }
    return clipped;
}

/* ------------------------------------------------------------------------------------------ */
/* Loudness range, maximum momentary / short-term loudness                                    */
/* ------------------------------------------------------------------------------------------ */

/* power of the window of `len` hops that starts at hop j: the block formula of the mode */
static double window_power(const double *H, uint64_t j, int len, int C, const float *w, uint32_t fs, int style)
{
    const double bins = style == 0 ? (double)((uint64_t)fs * (uint64_t)len / 10)
                                   : (double)((uint64_t)len * (((uint64_t)fs + 5) / 10));
    double power = style == 0 ? 1e-12 : 0.0;
    for (int c = 0; c < C; c++) {
        if (w[c] == 0.0f) continue;
        double s = H[j * C + c];
        for (int k = 1; k < len; k++) s += H[(j + k) * C + c];
        power += (double)w[c] * s;
    }
    return power / bins;
}

void hro_range_stats(const double *H, uint64_t n_hops, int C, const float *w, uint32_t fs, int gate_mode,
                     hro_range *o)
{
    const int style = gate_mode == HRO_GATE_LOUDNORM ? 1 : 0;
    const uint64_t n_m = n_hops >= 4 ? n_hops - 3 : 0, n_s = n_hops >= 30 ? n_hops - 29 : 0;
    double max_m = 0.0, max_s = 0.0;
    o->lra = 0.0;
    o->lra_low = o->lra_high = o->lra_threshold = NAN;
    o->n_shortterm = n_s;
    for (uint64_t j = 0; j < n_m; j++) {
        const double p = window_power(H, j, 4, C, w, fs, style);
        if (p > max_m) max_m = p;
    }
    if (style == 0) {
        uint64_t *hist = (uint64_t *)calloc(EB_HIST_SIZE, sizeof(uint64_t));
        double sum_kept = 0.0;
        uint64_t n_kept = 0;
        for (uint64_t j = 0; j < n_s; j++) {
            const double p = window_power(H, j, 30, C, w, fs, 0);
            if (p > max_s) max_s = p;
            const double loud = LOUDNESS(p);
            if (loud >= ABS_THRES) {
                hist[clipi((int)((loud - ABS_THRES) * EB_HIST_GRAIN), 0, EB_HIST_SIZE - 1)]++;
                sum_kept += p;
                n_kept++;
            }
        }
        if (n_kept) {
            double rt = sum_kept / (double)n_kept;
            if (!rt) rt = 1e-12;
            const double rel_thr = LOUDNESS(rt) + (-20.0);
            const int gp = clipi((int)((rel_thr - ABS_THRES) * EB_HIST_GRAIN), 0, EB_HIST_SIZE - 1);
            uint64_t nb_powers = 0;
            for (int i = gp; i < EB_HIST_SIZE; i++) nb_powers += hist[i];
            o->lra_threshold = rel_thr;
            if (nb_powers) {
                uint64_t n = 0, nb_pow = (uint64_t)(10 * (double)nb_powers * 0.01 + 0.5);
                for (int i = gp; i < EB_HIST_SIZE; i++) {
                    n += hist[i];
                    if (n >= nb_pow) { o->lra_low = (double)i / (double)EB_HIST_GRAIN + ABS_THRES; break; }
                }
                n = nb_powers;
                nb_pow = (uint64_t)(95 * (double)nb_powers * 0.01 + 0.5);
                for (int i = EB_HIST_SIZE - 1; i >= 0; i--) {
                    n -= n < hist[i] ? n : hist[i];
                    if (n < nb_pow) { o->lra_high = (double)i / (double)EB_HIST_GRAIN + ABS_THRES; break; }
                }
                o->lra = o->lra_high - o->lra_low;
            }
        }
        free(hist);
        o->max_momentary = n_m ? LOUDNESS(max_m) : -INFINITY;
        o->max_shortterm = n_s ? LOUDNESS(max_s) : -INFINITY;
    } else {
        uint64_t hist[1000];
        memset(hist, 0, sizeof hist);
        ln_init();
        for (uint64_t j = 0; j < n_s; j++) {
            const double p = window_power(H, j, 30, C, w, fs, 1);
            if (p > max_s) max_s = p;
            if (j % 10 == 0 && p >= ln_bounds[0]) ++hist[ln_find_index(p)];
        }
        /* ff_ebur128_loudness_range (histogram form) */
        uint64_t stl_size = 0;
        double stl_power = 0.0;
        for (int j = 0; j < 1000; ++j) { stl_size += hist[j]; stl_power += (double)hist[j] * ln_energies[j]; }
        if (stl_size) {
            stl_power /= (double)stl_size;
            const double stl_integrated = 0.01 * stl_power; /* minus_twenty_decibels */
            size_t index;
            if (stl_integrated < ln_bounds[0]) index = 0;
            else {
                index = ln_find_index(stl_integrated);
                if (stl_integrated > ln_energies[index]) ++index;
            }
            o->lra_threshold = 10.0 * (log(stl_integrated) / log(10.0)) - 0.691;
            stl_size = 0;
            for (size_t j = index; j < 1000; ++j) stl_size += hist[j];
            if (stl_size) {
                const uint64_t pl = (uint64_t)((double)(stl_size - 1) * 0.1 + 0.5);
                const uint64_t ph = (uint64_t)((double)(stl_size - 1) * 0.95 + 0.5);
                size_t j = index;
                stl_size = 0;
                while (stl_size <= pl) stl_size += hist[j++];
                const double l_en = ln_energies[j - 1];
                while (stl_size <= ph) stl_size += hist[j++];
                const double h_en = ln_energies[j - 1];
                o->lra_low = 10.0 * (log(l_en) / log(10.0)) - 0.691;
                o->lra_high = 10.0 * (log(h_en) / log(10.0)) - 0.691;
                o->lra = o->lra_high - o->lra_low;
            }
        }
        o->max_momentary = n_m ? 10.0 * (log(max_m) / log(10.0)) - 0.691 : -INFINITY;
        o->max_shortterm = n_s ? 10.0 * (log(max_s) / log(10.0)) - 0.691 : -INFINITY;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* libswresample filter bank                                                                  */
/* ------------------------------------------------------------------------------------------ */

/* Modified Bessel I0 by its power series, run to convergence in binary64 (the series form of
 * libswresample's bessel()). */
static double bessel_i0(double x)
{
    double v = 1.0, lastv = 0.0, t = 1.0;
    const double q = x * x / 4.0;
    for (int i = 1; v != lastv && i < 1000; i++) {
        lastv = v;
        t *= q / ((double)i * (double)i);
        v += t;
    }
    return v;
}

static uint64_t gcd_u64(uint64_t a, uint64_t b)
{
    while (b) { uint64_t t = a % b; a = b; b = t; }
    return a;
}

static double tp_src(const hro_track *t, int c, int64_t i);

/* resample_init(): phase count, and the dst_incr / src_incr bookkeeping of the resampling loop.  exact_rational: the
 * phase count is out/gcd(in, out) when that is <= 1 << phase_shift (1024); else it stays 1024 and the loop
 * interpolates linearly between neighbouring phases (linear_interp = 1: resample_linear is taken whenever frac or
 * dst_incr_mod is non-zero, swri_resample()).  src_incr / dst_incr = av_reduce(out, in * phase_count), both doubled
 * until one reaches 1 << 20.  Output n sits at idx = floor(n * dst_incr / src_incr) phase units (input position
 * idx / phase_count, phase idx % phase_count) with frac = (n * dst_incr) % src_incr. */
int hro_swr_plan(uint32_t in_rate, uint32_t out_rate, hro_swr_plan_t *pl)
{
    memset(pl, 0, sizeof *pl);
    if (!in_rate || !out_rate || in_rate >= out_rate) return 0; /* no upsampling needed / unsupported */
    const uint64_t g = gcd_u64(in_rate, out_rate);
    uint64_t phase_count = 1024;
    if (out_rate / g <= 1024) phase_count = out_rate / g;
    const uint64_t den = (uint64_t)in_rate * phase_count;
    const uint64_t g2 = gcd_u64(out_rate, den);
    uint64_t src_incr = out_rate / g2, dst_incr = den / g2;
    if (src_incr > 0x3fffffffu || dst_incr > 0x3fffffffu) return 0; /* av_reduce would approximate: not restated */
    pl->phases = (int)phase_count;
    pl->d0 = (uint32_t)dst_incr;
    pl->s0 = (uint32_t)src_incr;
    uint32_t scale = 1;
    while (dst_incr < (1u << 20) && src_incr < (1u << 20)) { dst_incr *= 2; src_incr *= 2; scale *= 2; }
    pl->frac_scale = scale;
    pl->inv_src_incr = 1.0 / (double)src_incr;
    pl->linear = (dst_incr % src_incr) != 0;
    return pl->phases;
}

int hro_swr_bank(uint32_t in_rate, uint32_t out_rate, int *in_step, double *taps)
{
    hro_swr_plan_t pl;
    const int phase_count = hro_swr_plan(in_rate, out_rate, &pl);
    if (!phase_count) return 0;
    if (in_step) *in_step = (int)pl.d0; /* exact phase counts: s0 == 1 and d0 is the input step per output in phases */
    if (!taps) return phase_count;

    const int tap_count = HRO_SWR_TAPS;
    const int center = (tap_count - 1) / 2;
    const double factor = 1.0; /* min(out*cutoff/in, 1) with upsampling */
    const double kaiser_beta = 9.0;
    const int ph_nb = phase_count % 2 ? phase_count : phase_count / 2 + 1;
    double tab[HRO_SWR_TAPS];
    double norm = 0.0;

    for (int ph = 0; ph < ph_nb; ph++) {
        double s = sin(M_PI * ph / phase_count) * (center & 1 ? 1 : -1);
        for (int i = 0; i < tap_count; i++) {
            const double x = M_PI * ((double)(i - center) - (double)ph / phase_count) * factor;
            double y;
            if (x == 0) y = 1.0;
            else y = s / x;
            const double w = 2.0 * x / (factor * tap_count * M_PI);
            y *= bessel_i0(kaiser_beta * sqrt(fmax(1 - w * w, 0)));
            tab[i] = y;
            s = -s;
            if (!ph) norm += y;
        }
        for (int i = 0; i < tap_count; i++) taps[ph * tap_count + i] = tab[i] / norm;
        if (phase_count % 2) continue;
        for (int i = 0; i < tap_count; i++)
            taps[(phase_count - ph) * tap_count + tap_count - 1 - i] = taps[ph * tap_count + i];
    }
    /* entry phase_count = phase 0 delayed by one sample (resample_init's two memcpy's; for an even phase count the
     * mirror image of phase 0 written above is the same thing) */
    for (int i = 0; i < tap_count; i++)
        taps[phase_count * tap_count + i] = i ? taps[i - 1] : taps[tap_count - 1];
    return phase_count;
}

/* One output of the resampling loop (resample_common / resample_linear of resample_template.c, DBL): n-th output of
 * channel c.  *pos_out receives its input position floor(t). */
static double swr_output(const hro_track *t, const hro_swr_plan_t *pl, const double *taps, int c, uint64_t n, int64_t *pos_out)
{
    const uint64_t num = n * pl->d0; /* n < 2^40, d0 < 2^30 would overflow: callers keep n * d0 < 2^63 (hours of audio do) */
    const uint64_t idx = num / pl->s0;
    const uint32_t frac = (uint32_t)(num % pl->s0) * pl->frac_scale;
    const int64_t pos = (int64_t)(idx / (uint64_t)pl->phases);
    const int ph = (int)(idx % (uint64_t)pl->phases);
    const double *f = taps + (size_t)ph * HRO_SWR_TAPS;
    double val = 0.0;
    for (int i = 0; i < HRO_SWR_TAPS; i++) val = fma(tp_src(t, c, pos - 15 + i), f[i], val);
    if (pl->linear) {
        const double *f2 = f + HRO_SWR_TAPS;
        double v2 = 0.0;
        for (int i = 0; i < HRO_SWR_TAPS; i++) v2 = fma(tp_src(t, c, pos - 15 + i), f2[i], v2);
        val += (v2 - val) * pl->inv_src_incr * frac;
    }
    if (pos_out) *pos_out = pos;
    return val;
}

static const double bs1770_taps[4][12] = {
    { 0.0017089843750, 0.0109863281250, -0.0196533203125, 0.0332031250000, -0.0594482421875,
      0.1373291015625, 0.9721679687500, -0.1022949218750, 0.0476074218750, -0.0266113281250,
      0.0148925781250, -0.0083007812500 },
    { -0.0291748046875, 0.0292968750000, -0.0517578125000, 0.0891113281250, -0.1665039062500,
      0.4650878906250, 0.7797851562500, -0.2003173828125, 0.1015625000000, -0.0582275390625,
      0.0330810546875, -0.0189208984375 },
    { -0.0189208984375, 0.0330810546875, -0.0582275390625, 0.1015625000000, -0.2003173828125,
      0.7797851562500, 0.4650878906250, -0.1665039062500, 0.0891113281250, -0.0517578125000,
      0.0292968750000, -0.0291748046875 },
    { -0.0083007812500, 0.0148925781250, -0.0266113281250, 0.0476074218750, -0.1022949218750,
      0.9721679687500, 0.1373291015625, -0.0594482421875, 0.0332031250000, -0.0196533203125,
      0.0109863281250, 0.0017089843750 },
};

const double *hro_bs1770_taps(void) { return &bs1770_taps[0][0]; }

/* sample with swr's start reflection (x[-k] = x[k]) and, when flushing, end reflection
 * (x[N+j] = x[N-1-j]) */
static double tp_src(const hro_track *t, int c, int64_t i)
{
    const int64_t N = (int64_t)t->frames;
    if (i < 0) i = -i;
    if (i >= N) i = 2 * N - 1 - i;
    if (i < 0 || i >= N) return 0.0;
    return hro_sample(t->pcm, t->fmt, (uint64_t)i * t->channels + c);
}

static void true_peak_swr(const hro_track *t, int flush_end, double *peak)
{
    int step = 0;
    const int C = t->channels;
    const int64_t N = (int64_t)t->frames;
    hro_swr_plan_t pl;
    const int P = hro_swr_plan(t->sample_rate, 192000, &pl);
    for (int c = 0; c < C; c++) peak[c] = 0.0;
    if (P == 0) { /* 192 kHz input (no resampling) or unsupported: sample peak */
        for (int c = 0; c < C; c++)
            for (int64_t n = 0; n < N; n++) {
                const double a = fabs(hro_sample(t->pcm, t->fmt, (uint64_t)n * C + c));
                if (a > peak[c]) peak[c] = a;
            }
        return;
    }
    if (N < HRO_SWR_TAPS + 1) return; /* invert_initial_buffer never completes: no output */
    double *taps = (double *)malloc((size_t)(P + 1) * HRO_SWR_TAPS * sizeof(double));
    hro_swr_bank(t->sample_rate, 192000, &step, taps);
    /* output exists while floor(t)+16 <= last available index */
    const int64_t last_pos = flush_end ? N - 1 : N - 17;
    for (int c = 0; c < C; c++) {
        double pk = 0.0;
        for (uint64_t n = 0;; n++) {
            int64_t pos;
            const double val = swr_output(t, &pl, taps, c, n, &pos);
            if (pos > last_pos) break;
            const double a = fabs(val);
            if (a > pk) pk = a;
        }
        peak[c] = pk;
    }
    free(taps);
}

static void true_peak_bs1770(const hro_track *t, double *peak)
{
    const int C = t->channels;
    const int64_t N = (int64_t)t->frames;
    const int factor = t->sample_rate < 96000 ? 4 : (t->sample_rate < 192000 ? 2 : 1);
    for (int c = 0; c < C; c++) {
        double pk = 0.0;
        for (int64_t n = 0; n < N; n++) {
            if (factor == 1) {
                const double a = fabs(hro_sample(t->pcm, t->fmt, (uint64_t)n * C + c));
                if (a > pk) pk = a;
                continue;
            }
            for (int p = 0; p < 4; p += 4 / factor) {
                double val = 0.0;
                for (int k = 11; k >= 0; k--) { /* oldest sample first (the order is not part of the spec; the GPU's
                                                    bank form of this filter sums in this order) */
                    const int64_t i = n - k;
                    const double x = i >= 0 ? hro_sample(t->pcm, t->fmt, (uint64_t)i * C + c) : 0.0;
                    val = fma(x, bs1770_taps[p][k], val);
                }
                const double a = fabs(val);
                if (a > pk) pk = a;
            }
        }
        peak[c] = pk;
    }
}

void hro_true_peak(const hro_track *t, int tp_mode, int flush_end, double *peak_per_ch)
{
    if (tp_mode == HRO_TP_BS1770) true_peak_bs1770(t, peak_per_ch);
    else if (tp_mode == HRO_TP_SAMPLE) {
        const int C = t->channels;
        for (int c = 0; c < C; c++) {
            double pk = 0.0;
            for (uint64_t n = 0; n < t->frames; n++) {
                const double a = fabs(hro_sample(t->pcm, t->fmt, n * C + c));
                if (a > pk) pk = a;
            }
            peak_per_ch[c] = pk;
        }
    } else true_peak_swr(t, flush_end, peak_per_ch);
}

/* ------------------------------------------------------------------------------------------ */
/* Whole measurement, hop-sum formulation                                                     */
/* ------------------------------------------------------------------------------------------ */

int hro_measure(const hro_track *t, int gate_mode, int tp_mode, hro_result *r)
{
    memset(r, 0, sizeof *r);
    if (!t->pcm || t->channels < 1 || t->channels > HRO_MAX_CH || t->sample_rate < 10) {
        r->status = -1;
        return -1;
    }
    const int C = t->channels;
    float wdef[HRO_MAX_CH];
    const float *w = t->ch_weight;
    if (!w) { hro_default_weights(C, wdef); w = wdef; }

    const uint32_t hop = hro_hop(t->sample_rate);
    const uint64_t nh = t->frames / hop;
    double *H = (double *)malloc((size_t)(nh ? nh : 1) * C * sizeof(double));
    double *P = (double *)malloc((size_t)(nh ? nh : 1) * sizeof(double));
    hro_hop_sums(t, H, nh);
    const uint64_t nb = block_powers_style(H, nh, C, w, t->sample_rate, P,
                                           gate_mode == HRO_GATE_LOUDNORM ? 1 : 0);
    hro_gate(P, nb, gate_mode, r, NULL);
    free(H);
    free(P);

    hro_true_peak(t, tp_mode, 0, r->peak_per_ch);
    double m = 0.0;
    for (int c = 0; c < C; c++) if (r->peak_per_ch[c] > m) m = r->peak_per_ch[c];
    r->true_peak_dbtp = 20.0 * log10(m); /* DBFS(); log10(0) = -inf */
    r->status = 0;
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Literal f_ebur128.c filter_frame() restatement (per-sample ring cache, running sums)       */
/* ------------------------------------------------------------------------------------------ */

int hro_ebur128_literal(const hro_track *t, hro_result *r, double *block_powers, uint64_t cap)
{
    memset(r, 0, sizeof *r);
    const int C = t->channels;
    if (C < 1 || C > HRO_MAX_CH) return -1;
    float wdef[HRO_MAX_CH];
    const float *w = t->ch_weight;
    if (!w) { hro_default_weights(C, wdef); w = wdef; }
    hro_kcoef k;
    hro_kweight_coeffs((double)t->sample_rate, &k);

    const int cache_size = (int)((uint64_t)t->sample_rate * 4 / 10); /* I400_BINS */
    const uint32_t hop = t->sample_rate / 10;
    double *cache = (double *)calloc((size_t)cache_size * C, sizeof(double));
    double sum[HRO_MAX_CH] = { 0 };
    double x[HRO_MAX_CH * 3] = { 0 }, y[HRO_MAX_CH * 3] = { 0 }, z[HRO_MAX_CH * 3] = { 0 };
    uint64_t *hist = (uint64_t *)calloc(EB_HIST_SIZE, sizeof(uint64_t));
    int cache_pos = 0, filled = 0;
    uint32_t sample_count = 0;
    double sum_kept = 0.0, rel_thr = ABS_THRES, integrated = -INFINITY;
    uint64_t n_kept = 0, nb = 0;

    for (uint64_t n = 0; n < t->frames; n++) {
        const int bin_id = cache_pos;
        cache_pos++;
        if (cache_pos == cache_size) { filled = 1; cache_pos = 0; }
        for (int ch = 0; ch < C; ch++) {
            x[ch * 3] = hro_sample(t->pcm, t->fmt, n * C + ch);
            if (!w[ch]) continue;
            double *dst = y + ch * 3, *src = x + ch * 3;
            dst[2] = dst[1]; dst[1] = dst[0];
            dst[0] = src[0] * k.pre_b[0] + src[1] * k.pre_b[1] + src[2] * k.pre_b[2]
                     - dst[1] * k.pre_a[1] - dst[2] * k.pre_a[2];
            x[ch * 3 + 2] = x[ch * 3 + 1];
            x[ch * 3 + 1] = x[ch * 3];
            dst = z + ch * 3; src = y + ch * 3;
            dst[2] = dst[1]; dst[1] = dst[0];
            dst[0] = src[0] * k.rlb_b[0] + src[1] * k.rlb_b[1] + src[2] * k.rlb_b[2]
                     - dst[1] * k.rlb_a[1] - dst[2] * k.rlb_a[2];
            const double bin = z[ch * 3] * z[ch * 3];
            sum[ch] = sum[ch] + bin - cache[(size_t)ch * cache_size + bin_id];
            cache[(size_t)ch * cache_size + bin_id] = bin;
        }
        if (++sample_count == hop) {
            double power_400 = 1e-12;
            sample_count = 0;
            if (filled) {
                for (int ch = 0; ch < C; ch++) power_400 += (double)w[ch] * sum[ch];
                power_400 /= (double)cache_size;
                if (block_powers && nb < cap) block_powers[nb] = power_400;
                nb++;
            }
            const double loud = LOUDNESS(power_400);
            if (loud >= ABS_THRES) {
                const int ip = clipi((int)((loud - ABS_THRES) * EB_HIST_GRAIN), 0, EB_HIST_SIZE - 1);
                hist[ip]++;
                sum_kept += power_400;
                n_kept++;
                double rt = sum_kept / (double)n_kept;
                if (!rt) rt = 1e-12;
                rel_thr = LOUDNESS(rt) + (-10.0);
                const int gp = clipi((int)((rel_thr - ABS_THRES) * EB_HIST_GRAIN), 0, EB_HIST_SIZE - 1);
                double isum = 0.0;
                uint64_t nint = 0;
                for (int i = gp; i < EB_HIST_SIZE; i++) {
                    nint += hist[i];
                    isum += (double)hist[i] * ENERGY((double)i / (double)EB_HIST_GRAIN + ABS_THRES);
                }
                if (nint) integrated = LOUDNESS(isum / (double)nint);
            }
        }
    }
    r->n_blocks = nb;
    r->n_gated = n_kept;
    r->integrated_lufs = integrated;
    r->rel_threshold_lufs = n_kept ? rel_thr : -INFINITY;
    free(cache);
    free(hist);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* af_loudnorm first pass: aresample -> 192 kHz, libavfilter/ebur128.c measurement            */
/* ------------------------------------------------------------------------------------------ */

int hro_loudnorm_measure(const hro_track *t, hro_result *r)
{
    memset(r, 0, sizeof *r);
    const int C = t->channels;
    if (C < 1 || C > HRO_MAX_CH) return -1;
    const int64_t N = (int64_t)t->frames;
    int step = 0;
    hro_swr_plan_t pl;
    const int P = hro_swr_plan(t->sample_rate, 192000, &pl);
    const uint32_t rate = P ? 192000u : t->sample_rate;

    /* 1. resample every channel to 192 kHz (planar f64) */
    uint64_t M = 0;
    double *res = NULL;
    if (P && N >= HRO_SWR_TAPS + 1) {
        double *taps = (double *)malloc((size_t)(P + 1) * HRO_SWR_TAPS * sizeof(double));
        hro_swr_bank(t->sample_rate, 192000, &step, taps);
        /* outputs with floor(pos) <= N-1: n * d0 / s0 < N * P */
        M = ((uint64_t)N * P * pl.s0 + pl.d0 - 1) / pl.d0;
        res = (double *)malloc((size_t)M * C * sizeof(double));
        for (int c = 0; c < C; c++)
            for (uint64_t m = 0; m < M; m++) res[(size_t)c * M + m] = swr_output(t, &pl, taps, c, m, NULL);
        free(taps);
    } else if (!P) {
        M = (uint64_t)N;
        res = (double *)malloc((size_t)(M ? M : 1) * C * sizeof(double));
        for (int c = 0; c < C; c++)
            for (uint64_t m = 0; m < M; m++) res[(size_t)c * M + m] = hro_sample(t->pcm, t->fmt, m * C + c);
    }

    /* 2. libebur128: 4th-order direct form II, channel map by index */
    hro_kcoef k;
    hro_kweight_coeffs((double)rate, &k);
    double b[5], a[5];
    const double *pb = k.pre_b, *pa = k.pre_a, *rb = k.rlb_b, *ra = k.rlb_a;
    b[0] = pb[0] * rb[0];
    b[1] = pb[0] * rb[1] + pb[1] * rb[0];
    b[2] = pb[0] * rb[2] + pb[1] * rb[1] + pb[2] * rb[0];
    b[3] = pb[1] * rb[2] + pb[2] * rb[1];
    b[4] = pb[2] * rb[2];
    a[0] = pa[0] * ra[0];
    a[1] = pa[0] * ra[1] + pa[1] * ra[0];
    a[2] = pa[0] * ra[2] + pa[1] * ra[1] + pa[2] * ra[0];
    a[3] = pa[1] * ra[2] + pa[2] * ra[1];
    a[4] = pa[2] * ra[2];

    float wln[HRO_MAX_CH];
    for (int c = 0; c < C; c++) wln[c] = 1.0f;
    if (C >= 6) { wln[3] = 0.0f; wln[4] = 1.41f; wln[5] = 1.41f; for (int c = 6; c < C; c++) wln[c] = 0.0f; }
    else if (C == 5) { wln[3] = 1.41f; wln[4] = 1.41f; }
    else if (C == 4) { wln[2] = 1.41f; wln[3] = 1.41f; }

    const uint64_t hop = ((uint64_t)rate + 5) / 10;
    const uint64_t nh = M / hop;
    double *H = (double *)calloc((size_t)(nh ? nh : 1) * C, sizeof(double));
    double *Pw = (double *)malloc((size_t)(nh ? nh : 1) * sizeof(double));
    double pkmax = 0.0;
    for (int c = 0; c < C; c++) {
        double v1 = 0, v2 = 0, v3 = 0, v4 = 0, pk = 0.0;
        for (uint64_t m = 0; m < M; m++) {
            const double s = res[(size_t)c * M + m];
            const double as = fabs(s);
            if (as > pk) pk = as;
            if (m >= nh * hop) continue;
            double v0 = fma(-a[4], v4, s);
            v0 = fma(-a[3], v3, v0);
            v0 = fma(-a[2], v2, v0);
            v0 = fma(-a[1], v1, v0);
            double o = b[4] * v4;
            o = fma(b[3], v3, o);
            o = fma(b[2], v2, o);
            o = fma(b[1], v1, o);
            o = fma(b[0], v0, o);
            H[(m / hop) * C + c] = fma(o, o, H[(m / hop) * C + c]);
            v4 = v3; v3 = v2; v2 = v1; v1 = v0;
        }
        r->peak_per_ch[c] = pk;
        if (pk > pkmax) pkmax = pk;
    }
    const uint64_t nb = block_powers_style(H, nh, C, wln, rate, Pw, 1);
    gate_loudnorm(Pw, nb, r, NULL);
    r->true_peak_dbtp = 20.0 * log10(pkmax);
    r->status = 0;
    free(H);
    free(Pw);
    free(res);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Synthetic PCM.  Integer hash + explicitly rounded f32 ops only, so the      */
/* device generator (csrc/synth.hip) produces the same bits.                                  */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
    float a_noise, a_sine, burst; /* burst = B/sqrt(2) sample magnitude */
    uint32_t sine_step;           /* phase increment per frame, 2^32 = one cycle */
    uint64_t quiet_len;           /* frames of -40 dB intro and outro */
    uint64_t burst_pos[3];
    uint32_t burst_len;
} synth_par;

static uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static double u01(uint64_t seed, uint64_t k) { return (double)(mix64(seed + 0x9E3779B97F4A7C15ull * (k + 1)) >> 11) * (1.0 / 9007199254740992.0); }

static void synth_params(uint64_t seed, uint32_t fs, uint64_t frames, synth_par *p)
{
    const uint64_t ps = mix64(seed ^ 0x48454144ull); /* "HEAD" */
    const double level_db = -30.0 + 24.0 * u01(ps, 0);
    const double an = pow(10.0, level_db / 20.0);
    const double as = 0.5 * an * (0.5 + u01(ps, 1));
    const double f = 40.0 * pow(300.0, u01(ps, 2));
    p->a_noise = (float)an;
    p->a_sine = (float)as;
    p->sine_step = (uint32_t)llround(f / (double)fs * 4294967296.0);
    uint64_t q = 10ull * fs;
    if (q > frames / 8) q = frames / 8;
    p->quiet_len = q;
    double B = 1.35 * (an + as);
    if (B > 0.98) B = 0.98;
    p->burst = (float)(B * 0.70710678118654752440);
    p->burst_len = 32;
    const uint64_t span = frames > 2 * q + 64 ? frames - 2 * q - 64 : 0;
    for (int k = 0; k < 3; k++)
        p->burst_pos[k] = span ? q + (uint64_t)(u01(ps, 3 + k) * (double)span) : UINT64_MAX;
}

void hro_synth_track(uint64_t seed, uint32_t fs, int channels, uint64_t frames, float *out)
{
    synth_par p;
    synth_params(seed, fs, frames, &p);
    for (uint64_t n = 0; n < frames; n++) {
        const float g = (n < p.quiet_len || n >= frames - p.quiet_len) ? 0.01f : 1.0f;
        int in_burst = -1;
        for (int k = 0; k < 3; k++)
            if (n >= p.burst_pos[k] && n - p.burst_pos[k] < p.burst_len) in_burst = (int)(n - p.burst_pos[k]);
        for (int c = 0; c < channels; c++) {
            const uint64_t idx = n * (uint64_t)channels + c;
            float xv;
            if (in_burst >= 0) {
                xv = ((in_burst + c) & 2) ? -p.burst : p.burst;
            } else {
                const uint64_t z = mix64(seed + 0x9E3779B97F4A7C15ull * (idx + 1));
                const float u = (float)((int32_t)(z >> 40) - 8388608) * (1.0f / 8388608.0f);
                const uint32_t ph = (uint32_t)(n * (uint64_t)p.sine_step) + (uint32_t)c * 0x40000000u;
                const float tt = (float)(int32_t)ph * (1.0f / 2147483648.0f);
                const float a1 = 1.0f - fabsf(tt);
                const float y = 4.0f * (tt * a1);
                const float d = y * fabsf(y) - y;
                const float s = fmaf(0.225f, d, y);
                const float v = fmaf(p.a_noise, u, p.a_sine * s);
                xv = g * v;
            }
            out[idx] = xv;
        }
    }
}
