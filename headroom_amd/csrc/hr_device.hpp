// hr_device.hpp -- structures shared by the host driver (hr_api.hip) and the kernels.
#pragma once

#include <cstdint>

#include "hr_tables.hpp"

namespace hr {

constexpr int kPreRec = 6;   // hop records kept in front of hop 0 (left shard edge; zeros otherwise)
constexpr int kPreHop = 29;  // hop-sum slots in front of hop 0: a shard's left neighbour fills them (3 s windows span 30 hops)
constexpr int kPreBlk = 3;   // of those, the last 3 are recomputed from the neighbour's records and complete the 3 boundary blocks
constexpr int kHalo = 16;    // frames of FIR halo on each side of a shard
constexpr int kEbuBins = 8001;
constexpr int kLnBins = 1000;
// waves per workgroup of the fused kernel (they share the rate's coefficient tiles and carry table in LDS); the host pads
// every run of same-rate items to a multiple of it
#ifndef HR_WG_WAVES
#define HR_WG_WAVES 4
#endif
constexpr int kFusedWaves = HR_WG_WAVES;
// k_true_peak_rows (the exact redo of what a fused wave could not list): a lane walks this many consecutive positions, a
// whole number of resampler cycles (`step` input frames), so that every lane of a wave is at the same phase of the bank
#ifdef __HIP__
__host__ __device__
#endif
inline uint32_t redo_lane_frames(uint32_t step) { return step * ((128u + step - 1u) / step); }
constexpr uint32_t kCandOverflowBit = 0x80000000u;  // in a wave's candidate count: its list ran out of slots
// 400 ms blocks (and 3 s windows) per gating workgroup: a 3-minute track (1797 blocks) is one chunk, a 10-hour stream 176 --
// with 16 384 (rounds 1, 2) that stream kept 22 of the 256 CUs busy for 1.5 ms (profiles/r3_config5_kernel_stats.md)
constexpr uint32_t kGateChunk = 2048;
constexpr uint32_t kGateSmallParts = 8;  // tracks of up to this many chunks (27 minutes) are gated by one workgroup each (k_gate_small)

// what one gating chunk hands to the per-track finalisation (summed there in chunk order: reproducible)
struct GatePart {
    double sum_kept;             // power of the blocks above the absolute gate
    unsigned long long n_kept;
    double st_sum_kept;          // the same for the 3 s windows (loudness range)
    unsigned long long st_n_kept;
    double max_m_power, max_s_power;
};

// per sample rate, device resident
struct RateDev {
    KCoef k;
    double G[16];
    double M1[16], M2[16];
    // rows of hop / sub_parts frames (RateTables::sub_*): G and M over such a row, rows behind a row that set its start state
    double Gs[16], Ms[16];
    int32_t sub_parts, sub_terms;
    const double *g;         // [hop][4]
    const double *swr_taps;  // [(phases+1)][32] f64
    const void *tp_atab;     // f16 A fragments of the MFMA screen, [tp_np][tp_mt][3][64][8 halfs]; nullptr: no fused path
    int32_t tp_np, tp_mt;
    int32_t tp_last_rows;    // outputs in the last 32-row block of the largest tile (max outputs - 32*(tp_mt-1))
    uint32_t q_n0;           // RateTables::q_n0 / q_ab: where q[0], q[1] start riding on q[2], q[3]
    double q_ab[2][2];
    uint32_t hop;
    int32_t swr_phases, swr_step;
    double swr_l1;
    // phase-interpolated resampling (RateTables::swr_S > 1: 22.05 kHz, 11.025 kHz, ...): see hr_tables.hpp
    int32_t swr_S;
    uint32_t swr_frac_scale;
    double swr_inv_src_incr;
    // The bank above is libswresample's (HR_TP_SWR192K) or the BS.1770 Annex 2 oversampler in the same shape
    // (HR_TP_BS1770), chosen when the context builds its tables.  tp_passthru0: phase 0 is a unit impulse (swr), so
    // its outputs are the samples themselves and the sample peak is a lower bound of the true peak.
    int32_t tp_passthru0;
    int32_t pad_;
};

// zero-state record of one (hop, channel)
struct HopRec {
    double e0;    // sum z0^2
    double q[4];  // sum z0[n]*g[n][j]
    double s[4];  // zero-state end state (y1, y1-y2, z1, z1-z2)
};

struct TrackDesc {
    const void *pcm;  // device pointer, interleaved
    uint64_t frames;
    uint64_t n_hops;
    uint64_t rec_base;  // HopRec index of this track's first slot; (hop h, ch c) is rec_base + (kPreRec + h)*ch + c
    uint64_t hop_base;  // hop-sum index of the first slot; (hop h, ch c) is hop_base + (kPreHop + h)*ch + c
    uint64_t blk_base;  // first block-power slot
    int64_t blk_first;  // first block index: 0, or -3 for a shard that has a left neighbour
    uint64_t n_blocks;
    const double *left_tail;   // [kHalo][ch] frames before frame 0 (shards), or nullptr
    const double *right_head;  // [kHalo][ch] frames after the last frame (shards), or nullptr
    uint32_t fs, hop;
    int32_t rate_idx;
    uint16_t ch, fmt;
    uint8_t stream_start;  // frame 0 is the start of the stream (swr start reflection / zero history)
    uint8_t stream_end;    // last frame is the end of the stream (no resampler flush: last 16 positions dropped)
    uint8_t tp_enabled;    // resampler produces output at all (swr: stream >= 33 frames)
    uint8_t tp_reflect;    // frames in front of the stream are its mirror image (libswresample) instead of silence
    uint8_t tp_drop;       // positions at the stream end that produce no output (swr: 16, no flush; BS.1770: 0)
    uint8_t fin_tail;      // 1: the ragged tail behind the last whole hop is a row of the fused kernel too (row n_hops; peak only)
    uint8_t role;          // 0 ordinary track; 1 parent of stereo planes (measured through them: no work items of its
                           // own, only hop-sum slots and gating); 2 stereo plane of a parent
    uint32_t first_plane;  // role 1: index of the first of its (ch+1)/2 consecutive plane descriptors
    int32_t st_pre;        // hop sums in front of hop 0 that are valid for the 3 s windows: 0, or kPreHop for a shard whose
                           // left neighbour handed over its last 29
    uint32_t fin_valid;    // fused tracks whose last rows are fused: frames of the LAST fused row, from its start, that exist and may
                           // be fetched (16-byte chunks that start inside them end inside the track)
    uint64_t first_hop;    // index of hop 0 in the whole stream (libebur128 takes its range blocks on a 1 s grid)
    uint32_t part_base, n_parts;  // gating chunks of kGateChunk blocks (one workgroup each): slots of their partial sums
    // rows of the fused kernel (run_measure): hops [0, sub_from) are one row each, hops [sub_from, n_hops) are `parts` rows
    // of row_len = hop / parts frames each (sub_from == n_hops, parts == 1: whole hops throughout).  Zero-state records
    // are per row, numbered in that order: (row r, ch c) is rec_base + (kPreRec + r)*ch + c
    uint32_t row_len, parts;
    uint64_t sub_from;
    float w[kMaxCh];
};

// one workgroup's share of a launch
struct WorkItem {
    uint32_t track;
    uint32_t pad;
    uint64_t first;  // first hop (K-weight kernels) or first frame (true-peak kernels)
};

// one wave's share of the fused kernel: up to 64 consecutive interior hops of one track
struct FusedItem {
    uint32_t track;
    uint32_t n_rows;     // 0: a padding wave (works on row 0 of the item it copies, reports nothing)
    uint64_t first_hop;  // record row of the wave's first row: (row r, ch c) is rec_base + (kPreRec + r)*ch + c
    uint64_t frame0;     // frame at which that row starts
    uint32_t row_len;    // frames per row: a hop, or a whole fraction of one
    uint32_t last;       // the wave's last row is the track's last fused row (TrackDesc::fin_valid)
    // line-aligned rows (stereo f32 at rates whose hops do not start on cache lines; hr_fused_core.hpp): the item's rows are
    // hops first_hop, first_hop + row_stride, ... (they share their offset from the 128-byte grid), every row's 16-frame
    // pieces start phi frames before the row, on a line, and the tile patterns start at pat0.  1, 0, 0: rows as they were.
    uint16_t row_stride;
    uint16_t phi;
    uint32_t pat0;
};

struct TrackOut {  // what the finalize kernel hands to the host
    double mean_energy;  // gated mean block energy (0 when nothing passed)
    double rel_energy;   // mean energy of the abs-gated blocks (before the -10 LU)
    double sum_kept;
    uint64_t n_kept;
    uint64_t n_gated_final;  // blocks in the final integration
    uint64_t n_blocks;
    // k_range_stats (whole tracks only): loudness range and the maxima of the 400 ms / 3 s powers
    double st_rel_energy;    // mean power of the short-term blocks behind the LRA gate (before the -20 LU); 0: none
    double max_m_power, max_s_power;  // 0: no block
    uint64_t n_shortterm;
    int32_t lra_low_bin, lra_high_bin;  // histogram bins of the 10 % / 95 % points; -1: no block passed
    double sum_gated;        // HR_GATE_EXACT: sum of the block powers in the final integration (mean_energy * n_gated_final)
};

}  // namespace hr
