// hr_kernels_fused_mc4.hip -- the fused K-weighting + true-peak pass for interleaved 5.1 read directly, as a workgroup of FOUR
// waves (hr_fused_core.hpp: k_fused_51x4, ROLE 1 / ROLE 2) instead of the three of hr_kernels_fused_mc.hip.
//
// The three-wave workgroup keeps ~200 VGPRs per lane, so a CU holds two of them: six of its eight wave slots.  Here the
// K-weighting of the pairs' second channels moves to a fourth wave (which screens nothing): two workgroups fill all eight slots,
// and the screening waves get shorter.  The default 5.1 layout gives the LFE no loudness weight, so of the three second channels
// only two need K-weighting at all: the fourth wave's f64 work (two channels) matches a screening wave's (one channel + screen).
// Per channel the arithmetic and its order are those of the three-wave kernel: records of weighted channels, candidate lists and
// peaks are bit-identical to it (tests/test_gpu_multichannel.py); the unweighted channel's records are zeros.
//
// Replaces ffmpeg's single pass over a 5.1 file (src/analyzer.rs:253-265).  A separate file so that it compiles beside the others.
#include "hr_fused_core.hpp"

namespace hr {
namespace {

// 5.1, four waves per workgroup (hr_fused_core.hpp, ROLE): three waves screen a pair each and K-weight its first channel, the
// fourth K-weights the two second channels that carry loudness weight
template <int INF>
void launch_51x4_class(int tp_class, const TrackDesc *tracks, const RateDev *rates, const FusedItem *items, uint32_t n_items,
                       HopRec *recs, const FusedScratch &fs, unsigned long long *peaks, hipStream_t st)
{
    constexpr int NCK = (INF == 1 ? 2 : INF == 2 ? 3 : 4) * 6;
    constexpr size_t lds = 2 * 64 * (NCK | 1) * 16 + kCoefArea + 2 * kCarryLds;
    const dim3 grid(n_items), block(256);
    uint2 *cands = static_cast<uint2 *>(fs.cands);
#define HR_51X4_CASE(MT, SMALL)                                                                                        \
    do {                                                                                                               \
        auto kern = &k_fused_51x4<MT, SMALL, INF>;                                                                     \
        if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)); \
        kern<<<grid, block, lds, st>>>(tracks, rates, items, n_items, recs, cands, fs.cand_cnt, fs.tpb, fs.overflow, peaks, 0); \
    } while (0)
    switch (tp_class) {
    case 0: HR_51X4_CASE(1, false); break;
    case 2: HR_51X4_CASE(2, false); break;
    case 5: HR_51X4_CASE(3, true); break;
    default: break;
    }
#undef HR_51X4_CASE
}

}  // namespace

bool fused_mc4_supported(int fmt, int tp_class)
{
    return (fmt == kFmtF32 || fmt == kFmtS16 || fmt == kFmtS24) && (tp_class == 0 || tp_class == 2 || tp_class == 5);
}

void launch_fused_mc4(int fmt, int tp_class, const TrackDesc *tracks, const RateDev *rates, const FusedItem *items, uint32_t n_items,
                      HopRec *recs, const FusedScratch &fs, unsigned long long *peaks, hipStream_t st)
{
    if (!n_items || !fused_mc4_supported(fmt, tp_class)) return;
    if (fmt == kFmtS24) launch_51x4_class<2>(tp_class, tracks, rates, items, n_items, recs, fs, peaks, st);
    else if (fmt == kFmtS16) launch_51x4_class<1>(tp_class, tracks, rates, items, n_items, recs, fs, peaks, st);
    else launch_51x4_class<0>(tp_class, tracks, rates, items, n_items, recs, fs, peaks, st);
}

}  // namespace hr
