// Test-only host check of the f16 coefficient tiles of the MFMA true-peak screen (hr_tables.cpp
// build_tp_tiles): for random data, (tile x window) must reproduce every interpolated output of the tile
// within the screen's error bound, and f32_to_f16_rn must round like IEEE binary16.
//   usage: cpu_tp_tiles_check <fs> [bs]     (bs: the BS.1770 Annex 2 bank instead of libswresample's)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../headroom_amd/csrc/hr_tables.hpp"

static float h2f(uint16_t h)
{
    const int s = h >> 15, e = (h >> 10) & 31, m = h & 1023;
    float v;
    if (e == 0) v = std::ldexp(static_cast<float>(m), -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = std::ldexp(static_cast<float>(m + 1024), e - 25);
    return s ? -v : v;
}

static uint16_t f2h_rtz(float f)  // what v_cvt_pkrtz_f16_f32 does for finite in-range values
{
    uint16_t h = hr::f32_to_f16_rn(f);
    if (std::fabs(h2f(h)) > std::fabs(f)) h--;  // step toward zero (same sign bit, smaller magnitude)
    return h;
}

int main(int argc, char **argv)
{
    const uint32_t fs = argc > 1 ? std::atoi(argv[1]) : 44100;
    // 1. rounding: every half value round-trips, midpoints go to even
    for (uint32_t h = 0; h < 0x7c00; h++) {
        if (hr::f32_to_f16_rn(h2f(h)) != h) { std::printf("roundtrip fails at %x\n", h); return 1; }
        const float a = h2f(h), b = h2f(h + 1);
        if (h + 1 < 0x7c00) {
            const float mid = 0.5f * (a + b);
            const uint16_t r = hr::f32_to_f16_rn(mid);
            const uint16_t want = (h & 1) ? h + 1 : h;
            if (r != want) { std::printf("midpoint fails at %x: got %x want %x\n", h, r, want); return 1; }
            if (hr::f32_to_f16_rn(std::nextafterf(mid, b)) != h + 1 || hr::f32_to_f16_rn(std::nextafterf(mid, a)) != h) {
                std::printf("neighbour of midpoint fails at %x\n", h);
                return 1;
            }
        }
    }
    const hr::RateTables *rt = hr::rate_tables(fs);
    const bool bs = argc > 2 && !std::strcmp(argv[2], "bs");
    const std::vector<uint16_t> &atab = bs ? rt->bs_atab : rt->tp_atab;
    const std::vector<double> &taps = bs ? rt->bs_taps : rt->swr_taps;
    if (atab.empty()) { std::printf("no tiles for %u\n", fs); return 0; }
    const int P = bs ? rt->bs_phases : rt->swr_phases, step = bs ? rt->bs_step : rt->swr_step;
    const int NP = bs ? rt->bs_np : rt->tp_np, MT = bs ? rt->bs_mt : rt->tp_mt;
    const uint32_t hop = rt->hop;
    const int T = (hop + 15) / 16;
    // a "row": hop frames plus 16 before and 48 after
    std::vector<float> x(hop + 16 + 64);
    uint64_t seed = 12345;
    for (auto &v : x) {
        seed = seed * 6364136223846793005ull + 1442695040888963407ull;
        v = static_cast<float>(static_cast<double>(static_cast<int64_t>(seed >> 11)) / 9007199254740992.0 * 2 - 1) * 0.8f;
    }
    auto X = [&](long u) { return x[u + 16]; };  // row-relative frame u >= -16
    double S = 0;
    for (auto v : x) S = std::fmax(S, std::fabs(v));
    const double err = 4.1e-3 * S + 4e-7;
    double worst = 0;
    long n_out = 0, next_m = 0;
    int tq = 0, tr = 0, pat = 0;
    const int q16 = (16 * P) / step, r16 = (16 * P) % step;
    for (int t = 0; t < T; t++) {
        const int m_lo = tq + (tr > 0);
        int nq = tq + q16, nr = tr + r16;
        if (nr >= step) { nr -= step; nq++; }
        const int count = (nq + (nr > 0)) - m_lo;
        if (m_lo != next_m) { std::printf("tile %d: outputs not contiguous (%d vs %ld)\n", t, m_lo, next_m); return 1; }
        next_m = m_lo + count;
        if (count > 32 * MT) { std::printf("tile %d: %d outputs do not fit %d m-tiles\n", t, count, MT); return 1; }
        for (int row = 0; row < 32 * MT; row++) {
            const int mt = row / 32;
            double approx = 0;
            for (int kappa = 0; kappa < 48; kappa++) {
                const int k3 = kappa / 16, half = (kappa % 16) / 8, i = kappa % 8;
                const int lane = (row & 31) + 32 * half;
                const uint16_t a = atab[((((static_cast<size_t>(pat) * MT + mt) * 3 + k3) * 64) + lane) * 8 + i];
                const float xv = h2f(f2h_rtz(X(16L * t - 16 + kappa)));
                approx += static_cast<double>(h2f(a)) * xv;
            }
            if (row >= count) {
                if (approx != 0) { std::printf("tile %d row %d beyond count is not zero\n", t, row); return 1; }
                continue;
            }
            const long m = m_lo + row;
            const long u = m * step / P;
            const int ph = static_cast<int>(m * step % P);
            if (u < 16L * t || u >= 16L * t + 16) { std::printf("tile %d row %d: position %ld outside tile\n", t, row, u); return 1; }
            if (ph == 0 && !bs && step == 1) {  // pass-through outputs (integer ratios: every frame is one) are not screened
                if (approx != 0) { std::printf("tile %d row %d: phase-0 row is not zero\n", t, row); return 1; }
                continue;
            }
            double exact = 0;
            for (int i = 0; i < 32; i++) exact = std::fma(static_cast<double>(X(u - 15 + i)), taps[static_cast<size_t>(ph) * 32 + i], exact);
            worst = std::fmax(worst, std::fabs(exact - approx));
            n_out++;
        }
        tq = nq; tr = nr;
        pat = pat + 1 == NP ? 0 : pat + 1;
    }
    std::printf("fs=%u P=%d step=%d NP=%d MT=%d tiles=%d outputs=%ld worst |exact-approx| = %.3e (bound %.3e)\n", fs, P, step, NP, MT, T,
                n_out, worst, err);
    return worst <= err ? 0 : 1;
}
