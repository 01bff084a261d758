// Test-only: prints the product's constant tables for one sample rate so the tests can compare them with
// the oracle's independent build of the same published formulas.
//   output: pb0 pb1 pb2 pa1 pa2 ra1 ra2 phases step carry_residual taps[(phases+1)*32]
#include <cstdio>
#include <cstdlib>

#include "../headroom_amd/csrc/hr_tables.hpp"

int main(int argc, char **argv)
{
    if (argc > 1 && argv[1][0] == 'e') {  // "edges": the three f_ebur128 bin-edge tables, 3 x 8001 powers
        for (double v : hr::ebur128_bin_edges()) std::printf("%.17g\n", v);
        return 0;
    }
    const uint32_t fs = argc > 1 ? std::atoi(argv[1]) : 48000;
    const hr::RateTables *rt = hr::rate_tables(fs);
    const hr::KCoef &k = rt->k;
    std::printf("%.17g %.17g %.17g %.17g %.17g %.17g %.17g %d %d %.17g\n", k.pb0, k.pb1, k.pb2, k.pa1, k.pa2, k.ra1, k.ra2,
                rt->swr_phases, rt->swr_step, rt->carry_residual);
    for (double v : rt->swr_taps) std::printf("%.17g\n", v);
    return 0;
}
