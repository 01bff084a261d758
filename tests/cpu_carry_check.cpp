// Test-only host check of the "every hop from zero state + carry correction" algebra that the HIP kernels
// k_hop_zero_state / k_hop_combine implement, using the product's own tables (hr_tables.cpp) and compared
// with the oracle's sequential hop sums.  Not part of the product; never linked into the library.
//   usage: cpu_carry_check <fs> <seconds> <seed> <kind: 0 noise, 1 noise+dc, 2 low sine>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../headroom_amd/csrc/hr_tables.hpp"
#include "../oracle/hr_oracle.h"

int main(int argc, char **argv)
{
    const uint32_t fs = argc > 1 ? std::atoi(argv[1]) : 48000;
    const double secs = argc > 2 ? std::atof(argv[2]) : 3.0;
    uint64_t seed = argc > 3 ? std::atoll(argv[3]) : 1;
    const int kind = argc > 4 ? std::atoi(argv[4]) : 0;
    const uint64_t N = static_cast<uint64_t>(fs * secs);
    std::vector<float> x(N);
    for (uint64_t n = 0; n < N; n++) {
        seed = seed * 6364136223846793005ull + 1442695040888963407ull;
        double u = static_cast<double>(static_cast<int64_t>(seed >> 11)) / 9007199254740992.0 * 2 - 1;
        double v = 0.1 * u;
        if (kind == 1) v += 0.5;
        if (kind == 2) v = 0.001 * u + 0.7 * std::sin(2 * M_PI * 30.0 * n / fs);
        x[n] = static_cast<float>(v);
    }
    hro_track t{x.data(), N, fs, 1, HRO_FMT_F32, nullptr};
    const uint32_t L = fs / 10;
    const uint64_t nh = N / L;
    std::vector<double> Href(nh);
    hro_hop_sums(&t, Href.data(), nh);

    const hr::RateTables *rt = hr::rate_tables(fs);
    const hr::KCoef k = rt->k;
    struct Rec { double e0, q[4], s[4]; };
    std::vector<Rec> rec(nh);
    for (uint64_t h = 0; h < nh; h++) {
        const int64_t n0 = h * L;
        double x1 = n0 >= 1 ? x[n0 - 1] : 0.0, x2 = n0 >= 2 ? x[n0 - 2] : 0.0;
        double y1 = 0, y2 = 0, z1 = 0, z2 = 0, e0 = 0, q[4] = {0, 0, 0, 0};
        for (uint32_t i = 0; i < L; i++) {
            const double x0 = x[n0 + i];
            double tt = k.pb0 * x0;
            tt = std::fma(k.pb1, x1, tt);
            tt = std::fma(k.pb2, x2, tt);
            tt = std::fma(-k.pa2, y2, tt);
            const double y0 = std::fma(-k.pa1, y1, tt);
            double v = std::fma(-2.0, y1, y0);
            v = v + y2;
            v = std::fma(-k.ra2, z2, v);
            const double z0 = std::fma(-k.ra1, z1, v);
            e0 = std::fma(z0, z0, e0);
            for (int j = 0; j < 4; j++) q[j] = std::fma(z0, rt->g[i * 4 + j], q[j]);
            x2 = x1; x1 = x0; y2 = y1; y1 = y0; z2 = z1; z1 = z0;
        }
        rec[h] = {e0, {q[0], q[1], q[2], q[3]}, {y1, y1 - y2, z1, z1 - z2}};
    }
    double worst = 0, worst_abs = 0;
    for (uint64_t h = 0; h < nh; h++) {
        const double zero4[4] = {0, 0, 0, 0};
        const double *p1 = h >= 1 ? rec[h - 1].s : zero4, *p2 = h >= 2 ? rec[h - 2].s : zero4, *p3 = h >= 3 ? rec[h - 3].s : zero4;
        double s[4];
        for (int i = 0; i < 4; i++) {
            double a = p1[i];
            for (int j = 0; j < 4; j++) a = std::fma(rt->M[0][i * 4 + j], p2[j], a);
            for (int j = 0; j < 4; j++) a = std::fma(rt->M[1][i * 4 + j], p3[j], a);
            s[i] = a;
        }
        double lin = 0, quad = 0;
        for (int i = 0; i < 4; i++) {
            lin = std::fma(s[i], rec[h].q[i], lin);
            double gs = 0;
            for (int j = 0; j < 4; j++) gs = std::fma(rt->G[i * 4 + j], s[j], gs);
            quad = std::fma(s[i], gs, quad);
        }
        const double H = rec[h].e0 + (2.0 * lin + quad);
        const double rel = std::fabs(H - Href[h]) / Href[h];
        if (rel > worst) worst = rel;
        if (std::fabs(H - Href[h]) > worst_abs) worst_abs = std::fabs(H - Href[h]);
    }
    std::printf("fs=%u hops=%llu worst_rel=%.3e worst_abs=%.3e Href0=%.6e Hreflast=%.6e\n", fs, (unsigned long long)nh, worst,
                worst_abs, Href[0], Href[nh - 1]);
    return worst < 1e-9 ? 0 : 1;
}
