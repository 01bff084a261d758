// hr_kernels_fused_mc.hip -- the fused K-weighting + true-peak pass (hr_fused_core.hpp, k_fused_f32s) instantiated for
// interleaved 5.1 / 7.1 tracks READ DIRECTLY: one read of the 18-byte (5.1 s24) ... 32-byte (7.1 f32) frames, no stereo planes.
//
// BASELINE configs[3] (96 kHz 24-bit 5.1): until round 3 such a track was first rewritten into three stereo planes
// (k_split_planes_wide: every byte read, written and read again -- 3.0 x the algorithmic traffic, 0.17 of the HBM roofline,
// the pre-pass cost more than the measurement).  Here a workgroup of channels / 2 waves owns 64 rows (hops) of the track
// itself: the waves DMA the rows' 16-frame piece into LDS once, a share of the instructions each, and wave p takes channels
// 2 p and 2 p + 1 out of it.  Same arithmetic in the same order as the stereo kernel on a plane of those two channels, so hop
// records, candidate decisions and peaks are bit-identical to the plane path (tests/test_gpu_multichannel.py), which stays
// available (hr_config.engine = 5) and serves every layout this file has no build for.
//
// Replaces ffmpeg's single pass over a file of any channel layout (src/analyzer.rs:253-265).
#include "hr_fused_core.hpp"

namespace hr {

namespace {

template <int INF, int NCH>
void launch_mc_class(int tp_class, const TrackDesc *tracks, const RateDev *rates, const FusedItem *items, uint32_t n_items,
                     HopRec *recs, const FusedScratch &fs, unsigned long long *peaks, hipStream_t st)
{
    constexpr int WGW = NCH / 2;
    constexpr int NCK = (INF == 1 ? 2 : INF == 2 ? 3 : 4) * NCH;        // 16-byte chunks per 16-frame piece
    constexpr size_t lds = 2 * 64 * (NCK | 1) * 16 + kCoefArea + 2 * kCarryLds;
    const dim3 grid(n_items), block(64 * WGW);
    uint2 *cands = static_cast<uint2 *>(fs.cands);
#define HR_MC_CASE(MT, SMALL)                                                                                          \
    do {                                                                                                               \
        auto kern = &k_fused_f32s<true, true, MT, SMALL, INF, NCH>;                                                    \
        if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)); \
        kern<<<grid, block, lds, st>>>(tracks, rates, items, n_items, recs, cands, fs.cand_cnt, fs.tpb, fs.overflow, peaks, 0); \
    } while (0)
    // the tile shapes of the usual rates: 96 / 88.2 kHz and up (one 32-output block), 48 kHz and the BS.1770 bank (two),
    // 44.1 kHz (three, the last one small)
    switch (tp_class) {
    case 0: HR_MC_CASE(1, false); break;
    case 2: HR_MC_CASE(2, false); break;
    case 5: HR_MC_CASE(3, true); break;
    default: break;  // (fused_mc_supported said no: the host never gets here)
    }
#undef HR_MC_CASE
}

}  // namespace

bool fused_mc_supported(int channels, int fmt, int tp_class)
{
    if (channels != 6 && channels != 8) return false;
    if (fmt != kFmtF32 && fmt != kFmtS16 && fmt != kFmtS24) return false;
    return tp_class == 0 || tp_class == 2 || tp_class == 5;
}

void launch_fused_mc(int channels, int fmt, int tp_class, bool four_waves, const TrackDesc *tracks, const RateDev *rates, const FusedItem *items,
                     uint32_t n_items, HopRec *recs, const FusedScratch &fs, unsigned long long *peaks, hipStream_t st)
{
    if (!n_items || !fused_mc_supported(channels, fmt, tp_class)) return;
    if (four_waves && channels == 6) {
        launch_fused_mc4(fmt, tp_class, tracks, rates, items, n_items, recs, fs, peaks, st);
        launch_refine_cands(tracks, rates, items, n_items * 3u, 3u, fs, peaks, st);
        return;
    }
#define HR_MC_FMT(NCH)                                                                                                  \
    do {                                                                                                                \
        if (fmt == kFmtS24) launch_mc_class<2, NCH>(tp_class, tracks, rates, items, n_items, recs, fs, peaks, st);       \
        else if (fmt == kFmtS16) launch_mc_class<1, NCH>(tp_class, tracks, rates, items, n_items, recs, fs, peaks, st);  \
        else launch_mc_class<0, NCH>(tp_class, tracks, rates, items, n_items, recs, fs, peaks, st);                      \
    } while (0)
    if (channels == 6) HR_MC_FMT(6);
    else HR_MC_FMT(8);
#undef HR_MC_FMT
    launch_refine_cands(tracks, rates, items, n_items * static_cast<uint32_t>(channels / 2), static_cast<uint32_t>(channels / 2), fs, peaks, st);
}

}  // namespace hr
