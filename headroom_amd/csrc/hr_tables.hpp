// hr_tables.hpp -- host-side constant tables of the meter (K-weighting coefficients, the
// zero-input-response tables that let every hop be filtered independently, resampler banks,
// gating histogram energies).  Product code: shares nothing with oracle/.
#pragma once

#include <cstdint>
#include <vector>

namespace hr {

constexpr int kMaxCh = 8;
constexpr int kSwrTaps = 32;
constexpr int kCarryTerms = 3;  // s_in = zend[h-1] + M zend[h-2] + M^2 zend[h-3]

// K-weighting, FFmpeg f_ebur128.c config_audio_input parametrisation
struct KCoef {
    double pb0, pb1, pb2, pa1, pa2;  // pre-filter (high shelf)
    double ra1, ra2;                 // RLB high-pass, b = {1,-2,1}
};
KCoef kweight_coeffs(double fs);

// Everything a kernel needs to know about one sample rate.
struct RateTables {
    uint32_t fs = 0;
    uint32_t hop = 0;  // fs/10
    KCoef k{};
    // zero-input response of the cascade: the K-weighted output at offset n of a hop is
    //   z[n] = z0[n] + sum_j s[j]*g[n][j],  s = (y[-1], y[-1]-y[-2], z[-1], z[-1]-z[-2]) at the hop start
    std::vector<double> g;   // [hop][4]
    double G[16];            // sum_n g[n][j]*g[n][k]
    double M[kCarryTerms - 1][16];  // M^1, M^2 (row-major): state transfer over one hop
    double carry_residual = 0;      // |M^kCarryTerms|_max, must be negligible
    // Rows shorter than a hop (hr_kernels_fused.hip: short tracks and small batches fill more lanes, and finish sooner,
    // when a hop is sub_parts rows of sub_len = hop / sub_parts frames; 1: this rate has no such division).  A row still
    // starts where the resampler's phase pattern starts (sub_len is a multiple of swr_step) and on a 16-byte boundary of
    // stereo f32.  sub_G / sub_M: G and M over sub_len frames; the state at a row's start needs the end states of the
    // sub_terms rows before it (|sub_M^sub_terms|_max = sub_residual, negligible).
    std::vector<long double> gl_keep;  // (scratch of the builder)
    uint32_t sub_parts = 1, sub_len = 0;
    double sub_G[16] = {0}, sub_M[16] = {0};
    int sub_terms = 0;
    double sub_residual = 0;
    // Once the pre-filter's own transient has died (n >= q_n0, a multiple of 48), the responses to the two
    // pre-filter state components are free modes of the RLB stage, i.e. combinations of g[.][2], g[.][3]:
    //   g[n][j] = q_ab[j][0]*g[n][2] + q_ab[j][1]*g[n][3]   (j = 0, 1; n >= q_n0)
    // so a kernel may stop accumulating q[0], q[1] there and rebuild them from q[2], q[3].
    // q_n0 == hop: no truncation.  q_residual: l2 norm of the mismatch over n >= q_n0 relative to |g[.][j]|.
    uint32_t q_n0 = 0;
    double q_ab[2][2] = {{0, 0}, {0, 0}};
    double q_residual = 0;
    // libswresample bank to 192 kHz (phase_count == 0: no resampling, true peak = sample peak)
    // Output n of the resampler sits at idx = floor(n * swr_step / swr_S) phase units: input position idx / swr_phases, phase
    // idx % swr_phases.  Exact phase counts (out/gcd <= 1024): swr_S == 1.  Otherwise (22.05 kHz, 11.025 kHz, ...) libswresample
    // keeps 1024 phases and interpolates linearly between neighbouring phases with weight frac / src_incr, frac =
    // ((n * swr_step) % swr_S) * swr_frac_scale, 1 / src_incr = swr_inv_src_incr (resample_init / resample_linear).
    int swr_phases = 0, swr_step = 0;
    int swr_S = 1;
    uint32_t swr_frac_scale = 1;
    double swr_inv_src_incr = 1.0;
    std::vector<double> swr_taps;  // [(phases+1)][32]
    double swr_l1 = 0;             // max over phases of sum |tap|
    // f16 coefficient tiles of the MFMA true-peak screen (hr_kernels_fused.hip): one tile of 16 input
    // positions produces the outputs m with floor(m*step/P) in the tile from a 48-frame window; the
    // output pattern repeats every tp_np tiles.  Layout [tp_np][tp_mt][3 k-steps][64 lanes][8 halfs]:
    // the A fragments of v_mfma_f32_32x32x16_f16 (lane l: row 32*mt + (l&31), k = 16*ks + 8*(l>>5) + i).
    std::vector<uint16_t> tp_atab;
    int tp_np = 0, tp_mt = 0;
    int tp_max_out = 0;  // most outputs any tile has
    // ITU-R BS.1770-4 Annex 2 oversampler written as a bank of the same shape, so that the same kernels serve it:
    // bs_phases = 4 below 96 kHz, 2 below 192 kHz (phases 0 and 2 of the table), 0 at and above (sample peak);
    // bs_step = 1; output n*P + p = sum_k x[n-k]*h[p][k] sits in tap slot 15-k of phase p (slots 16..31 are zero: the
    // filter is causal).  Phase 0 of this bank is a real filter, not a pass-through.
    int bs_phases = 0, bs_step = 0;
    std::vector<double> bs_taps;   // [(phases+1)][32]
    double bs_l1 = 0;
    std::vector<uint16_t> bs_atab;
    int bs_np = 0, bs_mt = 0, bs_max_out = 0;
};

// Builds (and caches per process) the tables of one rate.  Returns nullptr for fs < 10.
const RateTables *rate_tables(uint32_t fs);

// BS.1770-4 Annex 2 taps, [4][12]
const double *bs1770_taps();

// f_ebur128.c histogram bin energies: ENERGY(i/100 - 70), i < 8001
const std::vector<double> &ebur128_bin_energy();
// f_ebur128.c bin decisions without a device logarithm: three tables of 8001 powers (bins of a power; first bin of the
// -10 LU gate; first bin of the -20 LU gate), entry b = smallest power that the host libm puts into bin >= b
const std::vector<double> &ebur128_bin_edges();
// libavfilter/ebur128.c tables: bin-centre energies [1000] and boundaries [1001]
const std::vector<double> &loudnorm_bin_energy();
const std::vector<double> &loudnorm_bin_bounds();

// loudness weight per channel from a speaker mask (WAVE dwChannelMask / AV_CH_* bits), and for FFmpeg's default layout of
// a channel count
void weights_from_mask(uint32_t mask, int channels, float *w);
void default_weights(int channels, float *w);

uint16_t f32_to_f16_rn(float f);

}  // namespace hr
