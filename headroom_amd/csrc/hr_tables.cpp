// hr_tables.cpp -- see hr_tables.hpp.
//
// Follows (published algorithm, FFmpeg is not in /root/reference):
//   kweight_coeffs      libavfilter/f_ebur128.c config_audio_input(); reference call site
//                       src/analyzer.rs:253-267
//   swr bank            libswresample/resample.c build_filter() with the defaults of options.c
//                       (filter_size 32, kaiser beta 9, exact_rational)
//   histogram energies  f_ebur128.c init(); libavfilter/ebur128.c histogram_init()
#include "hr_tables.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>

namespace hr {

KCoef kweight_coeffs(double fs)
{
    KCoef c;
    double f0 = 1681.974450955533;
    const double G = 3.999843853973347;
    double Q = 0.7071752369554196;
    double K = std::tan(M_PI * f0 / fs);
    const double Vh = std::pow(10.0, G / 20.0);
    const double Vb = std::pow(Vh, 0.4996667741545416);
    const double a0 = 1.0 + K / Q + K * K;
    c.pb0 = (Vh + Vb * K / Q + K * K) / a0;
    c.pb1 = 2.0 * (K * K - Vh) / a0;
    c.pb2 = (Vh - Vb * K / Q + K * K) / a0;
    c.pa1 = 2.0 * (K * K - 1.0) / a0;
    c.pa2 = (1.0 - K / Q + K * K) / a0;
    f0 = 38.13547087602444;
    Q = 0.5003270373238773;
    K = std::tan(M_PI * f0 / fs);
    c.ra1 = 2.0 * (K * K - 1.0) / (1.0 + K / Q + K * K);
    c.ra2 = (1.0 - K / Q + K * K) / (1.0 + K / Q + K * K);
    return c;
}

// f_ebur128.c config_audio_output: a channel's loudness weight follows its speaker position -- LFE 0 (skipped), every
// back / side / top-back speaker (its BACK_MASK: BL BR BC SL SR TBL TBC TBR) 1.41, the rest 1.0.  Bits as in
// WAVE_FORMAT_EXTENSIBLE's dwChannelMask = FFmpeg's AV_CH_* (bit 0 = FL ... bit 17 = TBR).
void weights_from_mask(uint32_t mask, int channels, float *w)
{
    constexpr uint32_t kLfe = 0x8;
    constexpr uint32_t kBackMask = 0x10 | 0x20 | 0x100 | 0x200 | 0x400 | 0x8000 | 0x10000 | 0x20000;
    int c = 0;
    for (int bit = 0; bit < 32 && c < channels; bit++) {
        const uint32_t sp = 1u << bit;
        if (!(mask & sp)) continue;
        w[c++] = sp == kLfe ? 0.0f : (sp & kBackMask) ? 1.41f : 1.0f;
    }
    for (; c < channels; c++) w[c] = 1.0f;
}

// No layout given: FFmpeg assigns the default layout of the channel count (av_channel_layout_default): mono, stereo,
// 2.1 (FL FR LFE), 4.0 (FL FR FC BC), 5.0 (FL FR FC BL BR), 5.1 (FL FR FC LFE BL BR), 6.1 (FL FR FC LFE BC SL SR),
// 7.1 (FL FR FC LFE BL BR SL SR).
void default_weights(int channels, float *w)
{
    static const uint32_t kDefaultMask[9] = {0, 0x4, 0x3, 0x3 | 0x8, 0x7 | 0x100, 0x7 | 0x30, 0xF | 0x30, 0xF | 0x100 | 0x600, 0xF | 0x30 | 0x600};
    weights_from_mask(channels >= 1 && channels <= 8 ? kDefaultMask[channels] : 0, channels, w);
}

static double bessel_i0(double x)
{
    double v = 1.0, lastv = 0.0, t = 1.0;
    const double q = x * x / 4.0;
    for (int i = 1; v != lastv && i < 1000; i++) {
        lastv = v;
        t *= q / (static_cast<double>(i) * static_cast<double>(i));
        v += t;
    }
    return v;
}

static uint64_t gcd_u64(uint64_t a, uint64_t b)
{
    while (b) { const uint64_t t = a % b; a = b; b = t; }
    return a;
}

static void build_swr(RateTables &rt)
{
    const uint32_t out_rate = 192000;
    if (rt.fs >= out_rate) return;
    const uint64_t g = gcd_u64(rt.fs, out_rate);
    // resample_init: exact_rational takes out/gcd phases when that fits 1 << phase_shift = 1024; else 1024 phases and
    // linear interpolation between them.  dst_incr / src_incr = in * phases / out in lowest terms (av_reduce), both
    // doubled until one reaches 1 << 20.
    const int P = out_rate / g <= 1024 ? static_cast<int>(out_rate / g) : 1024;
    const uint64_t den = static_cast<uint64_t>(rt.fs) * P, g2 = gcd_u64(out_rate, den);
    uint64_t src_incr = out_rate / g2, dst_incr = den / g2;
    if (src_incr > 0x3fffffffu || dst_incr > 0x3fffffffu) { rt.swr_phases = -1; return; }  // av_reduce would approximate
    rt.swr_phases = P;
    rt.swr_step = static_cast<int>(dst_incr);
    rt.swr_S = static_cast<int>(src_incr);
    uint32_t scale = 1;
    while (dst_incr < (1u << 20) && src_incr < (1u << 20)) { dst_incr *= 2; src_incr *= 2; scale *= 2; }
    rt.swr_frac_scale = scale;
    rt.swr_inv_src_incr = 1.0 / static_cast<double>(src_incr);
    rt.swr_taps.assign(static_cast<size_t>(P + 1) * kSwrTaps, 0.0);
    const int center = (kSwrTaps - 1) / 2;
    const int ph_nb = P % 2 ? P : P / 2 + 1;
    double norm = 0.0, tab[kSwrTaps];
    double *taps = rt.swr_taps.data();
    for (int ph = 0; ph < ph_nb; ph++) {
        double s = std::sin(M_PI * ph / P) * (center & 1 ? 1 : -1);
        for (int i = 0; i < kSwrTaps; i++) {
            const double x = M_PI * (static_cast<double>(i - center) - static_cast<double>(ph) / P);
            double y = x == 0 ? 1.0 : s / x;
            const double w = 2.0 * x / (kSwrTaps * M_PI);
            y *= bessel_i0(9.0 * std::sqrt(std::fmax(1 - w * w, 0)));
            tab[i] = y;
            s = -s;
            if (!ph) norm += y;
        }
        for (int i = 0; i < kSwrTaps; i++) taps[ph * kSwrTaps + i] = tab[i] / norm;
        if (P % 2) continue;
        for (int i = 0; i < kSwrTaps; i++)
            taps[(P - ph) * kSwrTaps + kSwrTaps - 1 - i] = taps[ph * kSwrTaps + i];
    }
    // entry P = phase 0 delayed by one sample (resample_init's two memcpy's; the same as the mirror image written above
    // when P is even)
    for (int i = 0; i < kSwrTaps; i++) taps[P * kSwrTaps + i] = i ? taps[i - 1] : taps[kSwrTaps - 1];
    for (int ph = 0; ph < P; ph++) {
        double l1 = 0;
        for (int i = 0; i < kSwrTaps; i++) l1 += std::fabs(taps[ph * kSwrTaps + i]);
        if (l1 > rt.swr_l1) rt.swr_l1 = l1;
    }
}

uint16_t f32_to_f16_rn(float f)
{
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t absx = x & 0x7fffffffu;
    if (absx >= 0x7f800000u) return static_cast<uint16_t>(sign | 0x7c00u | ((absx > 0x7f800000u) ? 0x200u : 0));
    if (absx >= 0x477ff000u) return static_cast<uint16_t>(sign | 0x7c00u);  // rounds to infinity
    if (absx < 0x38800000u) {  // subnormal half (or zero): value = m * 2^-24
        if (absx < 0x33000000u) return static_cast<uint16_t>(sign);  // < 2^-25: rounds to zero
        const int e = static_cast<int>(absx >> 23);                  // biased exponent, 102..112
        const uint32_t mant = (absx & 0x7fffffu) | 0x800000u;        // 24-bit significand
        const int shift = 126 - e;                                   // 14..24 bits to drop
        uint32_t m = mant >> shift;
        const uint32_t rem = mant & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (m & 1))) m++;
        return static_cast<uint16_t>(sign | m);
    }
    uint32_t m = absx - 0x38000000u;  // rebias exponent by 112
    const uint32_t rem = m & 0x1fffu;
    m >>= 13;
    if (rem > 0x1000u || (rem == 0x1000u && (m & 1))) m++;
    return static_cast<uint16_t>(sign | m);
}

// passthru0: phase 0 of the bank is a unit impulse whose outputs the kernel takes from the sample peak
static void build_tp_tiles(int P, int step, const std::vector<double> &taps, bool passthru0, std::vector<uint16_t> &atab,
                           int &tp_np, int &tp_mt, int &tp_max_out)
{
    if (P <= 0) return;
    int g = 16, b = step;
    while (b) { const int tmp = g % b; g = b; b = tmp; }
    const int NP = step / g;
    // outputs per 16-frame tile, largest over the NP tile patterns
    long max_out = 0;
    for (int p = 0; p < NP; p++)
        max_out = std::max(max_out, (16L * (p + 1) * P + step - 1) / step - (16L * p * P + step - 1) / step);
    const int MT = static_cast<int>((max_out + 31) / 32);
    if (MT > 3) return;  // low sample rates (many phases per input frame) stay on the generic kernel
    tp_np = NP;
    tp_mt = MT;
    tp_max_out = static_cast<int>(max_out);
    atab.assign(static_cast<size_t>(NP) * MT * 3 * 64 * 8, 0);
    for (int p = 0; p < NP; p++) {
        const long m_lo = (16L * p * P + step - 1) / step;
        const long m_hi = (16L * (p + 1) * P + step - 1) / step;
        for (int mt = 0; mt < MT; mt++)
            for (int ks = 0; ks < 3; ks++)
                for (int lane = 0; lane < 64; lane++)
                    for (int i = 0; i < 8; i++) {
                        const long m = m_lo + 32 * mt + (lane & 31);
                        const int kappa = 16 * ks + 8 * (lane >> 5) + i;  // window frame index, window = tile frames -16 .. +31
                        double v = 0.0;
                        if (m < m_hi) {
                            const long u = m * step / P - 16L * p;      // position inside the tile, 0..15
                            const int ph = static_cast<int>(m * step % P);
                            const int tap = kappa - 1 - static_cast<int>(u);  // output uses frames pos-15 .. pos+16
                            // phase 0 is a unit impulse (the output IS the input sample): the kernel takes those
                            // outputs exactly from the sample peak, so their rows stay zero and are never candidates
                            if ((ph != 0 || !passthru0) && tap >= 0 && tap < kSwrTaps) v = taps[static_cast<size_t>(ph) * kSwrTaps + tap];
                        }
                        atab[((((static_cast<size_t>(p) * MT + mt) * 3 + ks) * 64) + lane) * 8 + i] =
                            f32_to_f16_rn(static_cast<float>(v));
                    }
    }
}

static void build_bs1770(RateTables &rt)
{
    const int P = rt.fs < 96000 ? 4 : (rt.fs < 192000 ? 2 : 0);
    if (!P) return;
    rt.bs_phases = P;
    rt.bs_step = 1;
    rt.bs_taps.assign(static_cast<size_t>(P + 1) * kSwrTaps, 0.0);
    const double *h = bs1770_taps();
    for (int ph = 0; ph < P; ph++) {
        const int src = ph * (4 / P);
        double l1 = 0;
        for (int k = 0; k < 12; k++) {
            rt.bs_taps[static_cast<size_t>(ph) * kSwrTaps + 15 - k] = h[src * 12 + k];
            l1 += std::fabs(h[src * 12 + k]);
        }
        if (l1 > rt.bs_l1) rt.bs_l1 = l1;
    }
}

static void matmul4(const long double *a, const long double *b, long double *c)
{
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            long double s = 0;
            for (int k = 0; k < 4; k++) s += a[i * 4 + k] * b[k * 4 + j];
            c[i * 4 + j] = s;
        }
}

static void build_carry(RateTables &rt)
{
    const uint32_t L = rt.hop;
    const KCoef &k = rt.k;
    rt.g.assign(static_cast<size_t>(L) * 4, 0.0);
    long double G[16] = {0}, M[16];
    std::vector<long double> gl(static_cast<size_t>(L) * 4);
    for (int j = 0; j < 4; j++) {
        // unit component j of the hop-start state s = (y[-1], y[-1]-y[-2], z[-1], z[-1]-z[-2]) (value and
        // backward difference: far better conditioned than raw histories, whose responses nearly cancel);
        // input history zero
        long double y1 = j == 0, y2 = (j == 0) - (j == 1), z1 = j == 2, z2 = (j == 2) - (j == 3);
        for (uint32_t n = 0; n < L; n++) {
            const long double y0 = -static_cast<long double>(k.pa1) * y1 - static_cast<long double>(k.pa2) * y2;
            const long double z0 = (y0 - 2.0L * y1 + y2) - static_cast<long double>(k.ra1) * z1 -
                                   static_cast<long double>(k.ra2) * z2;
            gl[static_cast<size_t>(n) * 4 + j] = z0;
            y2 = y1; y1 = y0;
            z2 = z1; z1 = z0;
        }
        M[0 * 4 + j] = y1; M[1 * 4 + j] = y1 - y2; M[2 * 4 + j] = z1; M[3 * 4 + j] = z1 - z2;
    }
    for (uint32_t n = 0; n < L; n++)
        for (int a = 0; a < 4; a++) {
            rt.g[static_cast<size_t>(n) * 4 + a] = static_cast<double>(gl[static_cast<size_t>(n) * 4 + a]);
            for (int b = 0; b < 4; b++) G[a * 4 + b] += gl[static_cast<size_t>(n) * 4 + a] * gl[static_cast<size_t>(n) * 4 + b];
        }
    for (int i = 0; i < 16; i++) rt.G[i] = static_cast<double>(G[i]);
    long double Mp[16], Mn[16];
    for (int i = 0; i < 16; i++) { Mp[i] = M[i]; rt.M[0][i] = static_cast<double>(M[i]); }
    for (int t = 1; t < kCarryTerms; t++) {
        matmul4(Mp, M, Mn);
        for (int i = 0; i < 16; i++) {
            Mp[i] = Mn[i];
            if (t < kCarryTerms - 1) rt.M[t][i] = static_cast<double>(Mn[i]);
        }
    }
    long double r = 0;
    for (int i = 0; i < 16; i++) r = std::fmax(r, std::fabs(Mp[i]));
    rt.carry_residual = static_cast<double>(r);
    rt.gl_keep = gl;  // (build_sub_rows, once the resampler's step is known)

    // q truncation point: fit g_j (j = 0, 1) on [n0, L) by the two RLB free modes g_2, g_3 (2x2 normal
    // equations in long double) and take the first multiple of 48 (three 16-frame steps of the fused kernel) where the mismatch is below 1e-15.
    rt.q_n0 = L;
    long double norm_j[2] = {0, 0};
    for (uint32_t n = 0; n < L; n++)
        for (int j = 0; j < 2; j++) norm_j[j] += gl[static_cast<size_t>(n) * 4 + j] * gl[static_cast<size_t>(n) * 4 + j];
    for (uint32_t n0 = 48; n0 + 64 <= L; n0 += 48) {
        long double a22 = 0, a23 = 0, a33 = 0, b2[2] = {0, 0}, b3[2] = {0, 0};
        for (uint32_t n = n0; n < L; n++) {
            const long double *gn = &gl[static_cast<size_t>(n) * 4];
            a22 += gn[2] * gn[2]; a23 += gn[2] * gn[3]; a33 += gn[3] * gn[3];
            for (int j = 0; j < 2; j++) { b2[j] += gn[2] * gn[j]; b3[j] += gn[3] * gn[j]; }
        }
        const long double det = a22 * a33 - a23 * a23;
        if (det == 0) break;
        long double ab[2][2], worst = 0;
        for (int j = 0; j < 2; j++) {
            ab[j][0] = (b2[j] * a33 - b3[j] * a23) / det;
            ab[j][1] = (b3[j] * a22 - b2[j] * a23) / det;
            long double res = 0;
            for (uint32_t n = n0; n < L; n++) {
                const long double *gn = &gl[static_cast<size_t>(n) * 4];
                const long double d = gn[j] - ab[j][0] * gn[2] - ab[j][1] * gn[3];
                res += d * d;
            }
            worst = std::fmax(worst, std::sqrt(res / norm_j[j]));
        }
        if (worst < 1e-15L) {
            rt.q_n0 = n0;
            for (int j = 0; j < 2; j++) { rt.q_ab[j][0] = static_cast<double>(ab[j][0]); rt.q_ab[j][1] = static_cast<double>(ab[j][1]); }
            rt.q_residual = static_cast<double>(worst);
            break;
        }
    }
}

// Rows of hop / p frames (RateTables::sub_parts): the smallest p >= 2 whose rows are at most 1600 frames, even (16-byte
// rows of stereo f32), whole multiples of the resampler's input step (every row starts at phase 0 of both banks; the
// BS.1770 bank's step is 1), and long enough for the q truncation and a few steps behind it.
static void build_sub_rows(RateTables &rt)
{
    const std::vector<long double> gl = std::move(rt.gl_keep);
    rt.gl_keep.clear();
    const uint32_t step = rt.swr_S == 1 && rt.swr_phases > 0 ? static_cast<uint32_t>(rt.swr_step) : 1u;
    for (uint32_t p = 2; p <= 16; p++) {
        if (rt.hop % p) continue;
        const uint32_t len = rt.hop / p;
        if (len > 1600) continue;
        if (len % 2 || len % step || len < 512 || len < rt.q_n0 + 96) break;  // (shorter rows only get worse)
        long double G[16] = {0}, M[16], Mp[16], Mn[16];
        for (uint32_t n = 0; n < len; n++)
            for (int a = 0; a < 4; a++)
                for (int b = 0; b < 4; b++) G[a * 4 + b] += gl[static_cast<size_t>(n) * 4 + a] * gl[static_cast<size_t>(n) * 4 + b];
        // the state after len frames of the zero-input response to unit component j: the cascade again, to len
        const KCoef &k = rt.k;
        for (int j = 0; j < 4; j++) {
            long double y1 = j == 0, y2 = (j == 0) - (j == 1), z1 = j == 2, z2 = (j == 2) - (j == 3);
            for (uint32_t n = 0; n < len; n++) {
                const long double y0 = -static_cast<long double>(k.pa1) * y1 - static_cast<long double>(k.pa2) * y2;
                const long double z0 = (y0 - 2.0L * y1 + y2) - static_cast<long double>(k.ra1) * z1 - static_cast<long double>(k.ra2) * z2;
                y2 = y1; y1 = y0;
                z2 = z1; z1 = z0;
            }
            M[0 * 4 + j] = y1; M[1 * 4 + j] = y1 - y2; M[2 * 4 + j] = z1; M[3 * 4 + j] = z1 - z2;
        }
        for (int i = 0; i < 16; i++) Mp[i] = M[i];
        int terms = 1;
        long double res = 1;
        for (; terms < 64; terms++) {  // Mp = M^terms
            res = 0;
            for (int i = 0; i < 16; i++) res = std::fmax(res, std::fabs(Mp[i]));
            if (res < 1e-20L) break;
            matmul4(Mp, M, Mn);
            for (int i = 0; i < 16; i++) Mp[i] = Mn[i];
        }
        if (res >= 1e-20L) break;
        rt.sub_parts = p;
        rt.sub_len = len;
        rt.sub_terms = terms;
        rt.sub_residual = static_cast<double>(res);
        for (int i = 0; i < 16; i++) { rt.sub_G[i] = static_cast<double>(G[i]); rt.sub_M[i] = static_cast<double>(M[i]); }
        return;
    }
}

const RateTables *rate_tables(uint32_t fs)
{
    static std::mutex mu;
    static std::map<uint32_t, std::unique_ptr<RateTables>> cache;
    if (fs < 10) return nullptr;
    // (a batch asks for the same rate several times per track: the last answer of this thread needs no lock; tables are
    // never freed)
    static thread_local uint32_t last_fs = 0;
    static thread_local const RateTables *last = nullptr;
    if (last && last_fs == fs) return last;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(fs);
    if (it != cache.end()) {
        last_fs = fs;
        last = it->second.get();
        return last;
    }
    auto rt = std::make_unique<RateTables>();
    rt->fs = fs;
    rt->hop = fs / 10;
    rt->k = kweight_coeffs(static_cast<double>(fs));
    build_carry(*rt);
    build_swr(*rt);
    build_sub_rows(*rt);
    // (the sample peak stands in for phase 0 only where EVERY input frame is a phase-0 output, i.e. at integer ratios:
    // at 44.1 kHz only every 147th frame is one, and an isolated peak between two of them is not an output at all)
    // (phase-interpolated rates -- swr_S > 1 -- have no tile table: their true peak runs on the generic kernel)
    if (rt->swr_S == 1)
        build_tp_tiles(rt->swr_phases, rt->swr_step, rt->swr_taps, rt->swr_step == 1, rt->tp_atab, rt->tp_np, rt->tp_mt, rt->tp_max_out);
    build_bs1770(*rt);
    build_tp_tiles(rt->bs_phases, rt->bs_step, rt->bs_taps, false, rt->bs_atab, rt->bs_np, rt->bs_mt, rt->bs_max_out);
    const RateTables *p = rt.get();
    cache[fs] = std::move(rt);
    return p;
}

const double *bs1770_taps()
{
    static const double t[4][12] = {
        {0.0017089843750, 0.0109863281250, -0.0196533203125, 0.0332031250000, -0.0594482421875,
         0.1373291015625, 0.9721679687500, -0.1022949218750, 0.0476074218750, -0.0266113281250,
         0.0148925781250, -0.0083007812500},
        {-0.0291748046875, 0.0292968750000, -0.0517578125000, 0.0891113281250, -0.1665039062500,
         0.4650878906250, 0.7797851562500, -0.2003173828125, 0.1015625000000, -0.0582275390625,
         0.0330810546875, -0.0189208984375},
        {-0.0189208984375, 0.0330810546875, -0.0582275390625, 0.1015625000000, -0.2003173828125,
         0.7797851562500, 0.4650878906250, -0.1665039062500, 0.0891113281250, -0.0517578125000,
         0.0292968750000, -0.0291748046875},
        {-0.0083007812500, 0.0148925781250, -0.0266113281250, 0.0476074218750, -0.1022949218750,
         0.9721679687500, 0.1373291015625, -0.0594482421875, 0.0332031250000, -0.0196533203125,
         0.0109863281250, 0.0017089843750},
    };
    return &t[0][0];
}

const std::vector<double> &ebur128_bin_energy()
{
    static const std::vector<double> v = [] {
        std::vector<double> e(8001);
        for (int i = 0; i < 8001; i++) {
            const double loud = static_cast<double>(i) / 100.0 + (-70.0);
            e[i] = std::pow(10.0, (loud + 0.691) / 10.0);
        }
        return e;
    }();
    return v;
}

// f_ebur128.c decides a histogram bin from a loudness it gets through log10():
//     loud = -0.691 + 10*log10(power);  bin = clip((int)((loud - (-70)) * 100), 0, 8000)        (gate_update / HIST_POS)
// and the two relative gates likewise from (loud + offset), offset = -10 LU (integrated) or -20 LU (range).  A device
// log10 that differs from the host libm's by one ulp would move a value that sits on a bin edge into the neighbouring
// bin, so the kernels never take a logarithm: these tables hold, for every bin b, the SMALLEST power whose bin (computed
// with the host libm, the expression order above, contraction off) is >= b, and the kernels find the bin by comparing
// powers.  Table 0: bins of block / window powers (entry 0 is the absolute gate: smallest power with loud >= -70);
// tables 1, 2: first bin of the relative gates (entry 0 is 0: every mean power has a first bin).
static int ebu_bin_of(double p, double offset, bool gate)
{
    const double loud = (-0.691 + 10.0 * std::log10(p)) + offset;
    if (gate && !(loud >= -70.0)) return -1;
    int b = static_cast<int>((loud - (-70.0)) * 100);
    return b < 0 ? 0 : (b > 8000 ? 8000 : b);
}

const std::vector<double> &ebur128_bin_edges()
{
    static const std::vector<double> v = [] {
        std::vector<double> e(3 * 8001);
        const double offs[3] = {0.0, -10.0, -20.0};
        for (int k = 0; k < 3; k++) {
            const bool gate = k == 0;
            for (int b = 0; b < 8001; b++) {
                if (!gate && b == 0) { e[k * 8001 + b] = 0.0; continue; }
                // the edge sits next to 10^((b/100 - 70 - offset + 0.691)/10): bracket it, then bisect on the doubles in between
                const double guess = std::pow(10.0, (static_cast<double>(b) / 100.0 - 70.0 - offs[k] + 0.691) / 10.0);
                double lo = guess * (1.0 - 1e-9), hi = guess * (1.0 + 1e-9);
                while (ebu_bin_of(lo, offs[k], gate) >= b) lo *= 0.5;  // (never taken unless libm is far off)
                while (ebu_bin_of(hi, offs[k], gate) < b) hi *= 2.0;
                for (;;) {
                    const double mid = lo + (hi - lo) * 0.5;
                    if (!(mid > lo && mid < hi)) break;  // lo and hi are neighbouring doubles
                    if (ebu_bin_of(mid, offs[k], gate) >= b) hi = mid; else lo = mid;
                }
                e[k * 8001 + b] = hi;
            }
        }
        return e;
    }();
    return v;
}

const std::vector<double> &loudnorm_bin_energy()
{
    static const std::vector<double> v = [] {
        std::vector<double> e(1000);
        for (int i = 0; i < 1000; ++i) e[i] = std::pow(10.0, (static_cast<double>(i) / 10.0 - 69.95 + 0.691) / 10.0);
        return e;
    }();
    return v;
}

const std::vector<double> &loudnorm_bin_bounds()
{
    static const std::vector<double> v = [] {
        std::vector<double> e(1001);
        e[0] = std::pow(10.0, (-70.0 + 0.691) / 10.0);
        for (int i = 1; i < 1001; ++i) e[i] = std::pow(10.0, (static_cast<double>(i) / 10.0 - 70.0 + 0.691) / 10.0);
        return e;
    }();
    return v;
}

}  // namespace hr
