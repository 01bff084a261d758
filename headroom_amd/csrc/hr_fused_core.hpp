// hr_fused_core.hpp -- the fused K-weighting + true-peak kernel template (k_fused_f32s) and its device helpers, shared by the
// two translation units that instantiate it: hr_kernels_fused.hip (stereo f32 / s16 / s24) and hr_kernels_fused_mc.hip (5.1 /
// 7.1 interleaved tracks read directly, round 3).  See hr_kernels_fused.hip for the description of the pass.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "hr_device.hpp"
#include "hr_kernels.hpp"

// Timing experiments (ablation bits of `dbg`, the HR_EXP_* switches, the split-wave kernel, environment switches of the
// launchers) exist only in builds with -DHR_EXPERIMENTS (make exp -> libheadroom_hip_exp.so): several of them make the
// results wrong on purpose, so the shipped library does not contain them.
#ifdef HR_EXPERIMENTS
#define HR_DBG(dbg, bit) (((dbg) & (bit)) != 0)
#else
#define HR_DBG(dbg, bit) false
#if defined(HR_EXP_NO_FRAGS) || defined(HR_EXP_NO_MFMA) || defined(HR_EXP_NO_KW) || defined(HR_EXP_NO_REDUCE) || \
    defined(HR_EXP_NO_AF) || \
    defined(HR_EXP_NO_Q) || defined(HR_EXP_NO_COEF_DMA) || defined(HR_EXP_NO_CAND) || defined(HR_EXP_NO_BOUNDS) || defined(HR_EXP_NO_MFMA2) || defined(HR_EXP_NO_RESIDENT)
#error "HR_EXP_* switches need -DHR_EXPERIMENTS"
#endif
#endif

namespace hr {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

// Tables that no kernel writes are read through the constant address space, so that wave-uniform reads
// become scalar loads (s_load) instead of 64 identical vector loads.
typedef const __attribute__((address_space(4))) double *cdouble_p;
typedef const __attribute__((address_space(4))) uint4v *cuint4v_p;

// LDS of one workgroup (4 waves that run the same steps of same-rate tracks in lock step):
//   4 x 8 KiB   one PCM piece buffer per wave
//   2 x 9 KiB   coefficient tiles of the tile being screened / the next one, shared by the four waves
//   2 x 1 KiB   carry-table slice of this step / the next one, shared
constexpr int kPieceLds = 8192, kCoefLds = 9 * 1024, kCarryLds = 1024;
// (the coefficient area of k_fused_f32s: the two halves above, or -- rates with ONE tile pattern of at most two blocks: 48,
// 96, 32 kHz ..., the BS.1770 bank -- the tile's three rotations, resident for the whole kernel, and a block of zeros)
constexpr int kCoefArea = 24 * 1024;
constexpr int kGroupLds = kFusedWaves * kPieceLds + kCoefArea + 2 * kCarryLds;
constexpr int kFmtF32 = 0, kFmtS16 = 1, kFmtS24 = 2;  // HR_FMT_F32, HR_FMT_S16, HR_FMT_S24 of include/headroom_hip.h
constexpr int kCandCap = 512;              // candidate slots per wave

constexpr float kScreenRel = 4.1e-3f;  // >= (2^-10 [pkrtz x] + 2^-11 [rn h]) * L1(2.3522) + f32 accumulation
constexpr float kScreenAbs = 4.0e-7f;  // f16 subnormal inputs / coefficients
constexpr float kF16Safe = 60000.0f;   // above this |x| does not fit f16: everything becomes a candidate

// Wave-wide maximum with DPP cross-lane moves only (no LDS round trips): quads, half rows, rows, then the
// two row broadcasts; lane 63 ends up with the maximum of all 64 lanes.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_max(float v)
{
    const int o = __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return fmaxf(v, __int_as_float(o));
}

__device__ __forceinline__ float wave_max_f32(float v)
{
    v = dpp_max<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v = dpp_max<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v = dpp_max<0x141, 0xf>(v);  // row_half_mirror
    v = dpp_max<0x140, 0xf>(v);  // row_mirror
    v = dpp_max<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_max<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ double wave_max_f64(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

__device__ __forceinline__ half8 as_half8(uint4v u)
{
    union { uint4v u; half8 h; } c;
    c.u = u;
    return c.h;
}

__device__ __forceinline__ unsigned pk_f16(float a, float b)
{
    const auto h = __builtin_amdgcn_cvt_pkrtz(a, b);  // round toward zero: |x - f16(x)| <= 2^-10 |x|
    unsigned u;
    __builtin_memcpy(&u, &h, 4);
    return u;
}

// frame i (relative to the track start) of channel c of an interleaved f32 / s16 / s24 track of t.ch channels, with start
// reflection / shard halos; exact path only
__device__ __forceinline__ double fetch_f32s(const TrackDesc &t, int c, int64_t i)
{
    const int64_t N = static_cast<int64_t>(t.frames);
    const int C = t.ch;
    if (i < 0) {
        if (t.left_tail) return (i >= -kHalo) ? t.left_tail[(kHalo + i) * C + c] : 0.0;
        if (!t.tp_reflect) return 0.0;  // BS.1770: silence in front of the stream
        i = -i;
    }
    if (i >= N) {
        if (t.right_head && i - N < kHalo) return t.right_head[(i - N) * C + c];
        return 0.0;
    }
    if (t.fmt == kFmtS16) return static_cast<double>(static_cast<const int16_t *>(t.pcm)[i * C + c]) * (1.0 / 32768.0);
    if (t.fmt == kFmtS24) {
        const uint8_t *b = static_cast<const uint8_t *>(t.pcm) + (i * C + c) * 3;
        const int32_t x = static_cast<int32_t>(static_cast<uint32_t>(b[0]) << 8 | static_cast<uint32_t>(b[1]) << 16 | static_cast<uint32_t>(b[2]) << 24) >> 8;
        return static_cast<double>(x) * (1.0 / 8388608.0);
    }
    return static_cast<double>(static_cast<const float *>(t.pcm)[i * C + c]);
}

// One interpolated output, exactly: f64 FMA chain over the 32 taps in the oracle's order.
__device__ __forceinline__ double exact_output(const TrackDesc &t, const double *__restrict__ taps, int P, int step,
                                               int64_t row_frame0, int c, int m)
{
    const int u = (m * step) / P;
    const int ph = m * step - u * P;
    const int64_t pos = row_frame0 + u;
    const double *__restrict__ f = taps + static_cast<size_t>(ph) * kSwrTaps;
    double x[kSwrTaps];
    if (t.fmt == kFmtF32 && pos >= 15 && pos + 16 < static_cast<int64_t>(t.frames)) {
        const int C = t.ch;
        const float *p = static_cast<const float *>(t.pcm) + (pos - 15) * C + c;
#pragma unroll
        for (int k = 0; k < kSwrTaps; k++) x[k] = static_cast<double>(p[C * k]);
    } else {
#pragma unroll
        for (int k = 0; k < kSwrTaps; k++) x[k] = fetch_f32s(t, c, pos - 15 + k);
    }
    double val = 0.0;
#pragma unroll
    for (int k = 0; k < kSwrTaps; k++) val = fma(x[k], f[k], val);
    return fabs(val);
}

struct KState {
    double x1, x2, y1, y2, z1, z2, e0, q0, q1, q2, q3;
};

struct G4 {
    double g0, g1, g2, g3;
};

// one frame's four carry-table values from the wave's LDS slice (same address in every lane: a broadcast read)
__device__ __forceinline__ G4 lds_g4(const char *p)
{
    const double2 a = *reinterpret_cast<const double2 *>(p);
    const double2 b = *reinterpret_cast<const double2 *>(p + 16);
    return G4{a.x, a.y, b.x, b.y};
}

__device__ __forceinline__ void kw_sample(KState &s, const KCoef &k, float xf, const G4 &g)
{
    const double x0 = static_cast<double>(xf);
    double tt = k.pb0 * x0;
    tt = fma(k.pb1, s.x1, tt);
    tt = fma(k.pb2, s.x2, tt);
    tt = fma(-k.pa2, s.y2, tt);
    const double y0 = fma(-k.pa1, s.y1, tt);
    double v = fma(-2.0, s.y1, y0);
    v = v + s.y2;
    v = fma(-k.ra2, s.z2, v);
    const double z0 = fma(-k.ra1, s.z1, v);
    s.e0 = fma(z0, z0, s.e0);
    s.q0 = fma(z0, g.g0, s.q0);
    s.q1 = fma(z0, g.g1, s.q1);
    s.q2 = fma(z0, g.g2, s.q2);
    s.q3 = fma(z0, g.g3, s.q3);
    s.x2 = s.x1; s.x1 = x0;
    s.y2 = s.y1; s.y1 = y0;
    s.z2 = s.z1; s.z1 = z0;
}

// the same without q[0], q[1]: past RateDev::q_n0 they are rebuilt from q[2], q[3]
__device__ __forceinline__ void kw_sample_short(KState &s, const KCoef &k, float xf, double g2, double g3)
{
    const double x0 = static_cast<double>(xf);
    double tt = k.pb0 * x0;
    tt = fma(k.pb1, s.x1, tt);
    tt = fma(k.pb2, s.x2, tt);
    tt = fma(-k.pa2, s.y2, tt);
    const double y0 = fma(-k.pa1, s.y1, tt);
    double v = fma(-2.0, s.y1, y0);
    v = v + s.y2;
    v = fma(-k.ra2, s.z2, v);
    const double z0 = fma(-k.ra1, s.z1, v);
    s.e0 = fma(z0, z0, s.e0);
#ifndef HR_EXP_NO_Q
    s.q2 = fma(z0, g2, s.q2);
    s.q3 = fma(z0, g3, s.q3);
#else
    (void)g2; (void)g3;
#endif
    s.x2 = s.x1; s.x1 = x0;
    s.y2 = s.y1; s.y1 = y0;
    s.z2 = s.z1; s.z1 = z0;
}

// LDS-DMA of 16 bytes per lane: global (uniform base + 32-bit lane offset) -> LDS (uniform base + lane*16).
// The offset passes through an empty asm so that its zero-extension stays next to the add and the
// instruction is selected in its scalar-base form (no 64-bit vector address arithmetic per DMA).
template <int AUX = 0>
__device__ __forceinline__ void dma16(const char *base, unsigned lane_off, char *lds)
{
    asm volatile("" : "+v"(lane_off));
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + lane_off),
                                     (__attribute__((address_space(3))) void *)lds, 16, 0, AUX);
}
#if defined(HR_FAST_BUILD) && !defined(HR_FAST_MT)
#define HR_FAST_MT 3
#define HR_FAST_SMALL true
#endif
#ifdef HR_FAST_MT
#define HR_FAST_MT_OR_3 HR_FAST_MT
#define HR_FAST_SMALL_OR_TRUE HR_FAST_SMALL
#else
#define HR_FAST_MT_OR_3 3
#define HR_FAST_SMALL_OR_TRUE true
#endif

// MT: 32-output blocks per tile (1..3); SMALL: the last block holds at most 8 outputs (44.1 kHz: outputs 64..69
// of 70), so only the first four accumulator registers of its groups can be non-zero.
// LDS reads the compiler must not see as LDS reads.  After a global_load_lds the compiler cannot tell which
// part of LDS the DMA will write, so it puts s_waitcnt vmcnt(0) in front of every later ds_read -- here that
// would stall each step on the PCM piece it has just requested.  The buffers these helpers read (coefficient
// tiles, carry-table slices) were requested one step earlier and are covered by the step's opening
// s_waitcnt vmcnt(0).  The asm hides the access; lds_wait() is the matching lgkmcnt wait and ties the
// registers to it so that no use is scheduled in front of it; its memory clobber keeps a later DMA into the
// buffer just read from being moved in front of the reads.
__device__ __forceinline__ unsigned lds_addr(const void *p)
{
    return static_cast<unsigned>(reinterpret_cast<size_t>((const __attribute__((address_space(3))) char *)p));
}
__device__ __forceinline__ uint4v lds_ld128(unsigned addr, int off)
{
    uint4v r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(off));
    return r;
}
__device__ __forceinline__ void lds_wait(uint4v &a, uint4v &b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b) : : "memory"); }
__device__ __forceinline__ void lds_wait(uint4v &a, uint4v &b, uint4v &c) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c) : : "memory"); }
__device__ __forceinline__ void lds_wait(uint4v &a, uint4v &b, uint4v &c, uint4v &d)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "memory");
}
__device__ __forceinline__ double2 as_double2(uint4v u)
{
    union { uint4v u; double2 d; } c;
    c.u = u;
    return c.d;
}

// INF: input format, 0 = stereo f32 (8-byte frames, a 128-byte piece per row and step), 1 = stereo s16 (4-byte frames,
// a 64-byte piece: four lanes per piece in the DMA, four LDS reads per lane, the samples widened to f32 exactly --
// x * 2^-15 -- as they come out of LDS; everything behind that is the same code).
// two s16 frames (one dword each: left in the low half) as the float4 the f32 path holds
__device__ __forceinline__ float4 widen_s16x2(unsigned a, unsigned b)
{
    constexpr float sc = 1.0f / 32768.0f;
    return make_float4(static_cast<float>(static_cast<int>(a << 16) >> 16) * sc, static_cast<float>(static_cast<int>(a) >> 16) * sc,
                       static_cast<float>(static_cast<int>(b << 16) >> 16) * sc, static_cast<float>(static_cast<int>(b) >> 16) * sc);
}

// INF 2 = stereo s24 (packed, 6-byte frames, a 96-byte piece: six lanes per piece in the DMA, six LDS reads per lane,
// every sample cut out of the lane's 24 dwords by compile-time shifts and widened exactly, x * 2^-23).
// one packed s24 stereo frame from memory (prologue only)
__device__ __forceinline__ float2 s24_frame(const uint8_t *p)
{
    constexpr float sc = 1.0f / 8388608.0f;
    const int32_t l = static_cast<int32_t>(static_cast<uint32_t>(p[0]) << 8 | static_cast<uint32_t>(p[1]) << 16 | static_cast<uint32_t>(p[2]) << 24) >> 8;
    const int32_t r = static_cast<int32_t>(static_cast<uint32_t>(p[3]) << 8 | static_cast<uint32_t>(p[4]) << 16 | static_cast<uint32_t>(p[5]) << 24) >> 8;
    return make_float2(static_cast<float>(l) * sc, static_cast<float>(r) * sc);
}
// sample s (0..31: frame s/2, channel s&1) of a 16-frame s24 piece held as 24 dwords (+1 so that w[d+1] exists)
template <int S>
__device__ __forceinline__ float s24_sample(const unsigned (&w)[25])
{
    constexpr int d = (3 * S) >> 2, sh = ((3 * S) & 3) * 8;
    const unsigned bits = sh <= 8 ? w[d] >> sh : __funnelshift_r(w[d], w[d + 1], sh);
    return static_cast<float>(static_cast<int>(bits << 8) >> 8) * (1.0f / 8388608.0f);
}

// ---- NCH > 2: interleaved 5.1 / 7.1 tracks read directly (round 3) ------------------------------------------------------------
// A workgroup is NCH / 2 waves that work on the SAME 64 rows, wave p on channels 2p and 2p + 1.  The 16-frame piece of a row is
// 16 * NCH * (sample bytes) bytes = NCK 16-byte chunks; the waves fetch it ONCE, a share of the DMA instructions each, into an
// LDS image of NS = NCK | 1 slots per row (an odd row stride: the lanes' reads of one frame land in different banks without
// any swizzle; slot NCK of a row, where there is one, is never written).  DMA lane positions count through that image -- position
// 64 j + lane is (row, chunk) = (pos / NS, pos % NS) -- so adjacent lanes still fetch adjacent chunks of one row.  The image is
// double-buffered (piece k in buffer k & 1): a wave issues the DMA of piece k + 1 while the other waves may still be reading piece
// k.  Everything behind the piece's arrival in registers -- K-weighting, fragments, screen, candidate lists -- is the stereo code,
// on the wave's own channel pair.
// one frame's channel pair (2p, 2p + 1) from memory (prologue: the 16 frames in front of a row), fp = the frame's first byte
template <int INF>
__device__ __forceinline__ float2 mc_frame_pair(const char *fp, int pair)
{
    if constexpr (INF == 2) {
        return s24_frame(reinterpret_cast<const uint8_t *>(fp) + 6 * pair);
    } else if constexpr (INF == 1) {
        const float4 w = widen_s16x2(*reinterpret_cast<const unsigned *>(fp + 4 * pair), 0u);
        return make_float2(w.x, w.y);
    } else {
        return *reinterpret_cast<const float2 *>(fp + 8 * pair);
    }
}

// one channel of one frame from memory (prologue of the K-weighting wave of a four-wave 5.1 workgroup), fp = the frame's first byte
template <int INF>
__device__ __forceinline__ float mc_frame_chan(const char *fp, int c)
{
    if constexpr (INF == 2) {
        const uint8_t *b = reinterpret_cast<const uint8_t *>(fp) + 3 * c;
        const int32_t x = static_cast<int32_t>(static_cast<uint32_t>(b[0]) << 8 | static_cast<uint32_t>(b[1]) << 16 | static_cast<uint32_t>(b[2]) << 24) >> 8;
        return static_cast<float>(x) * (1.0f / 8388608.0f);
    } else if constexpr (INF == 1) {
        return static_cast<float>(*reinterpret_cast<const int16_t *>(fp + 2 * c)) * (1.0f / 32768.0f);
    } else {
        return *reinterpret_cast<const float *>(fp + 4 * c);
    }
}

// ROLE (5.1 only, round 3): 0 = every wave of the workgroup does the same work on its own channel pair (the stereo kernel; 5.1 /
// 7.1 with NCH / 2 waves).  The FOUR-wave 5.1 workgroup splits the work by role instead, so that a CU holds eight balanced
// waves where three-wave workgroups left two of its eight wave slots empty: ROLE 1 = one of the three screening waves --
// true-peak screen of its pair, K-weighting of the pair's FIRST channel only; ROLE 2 = the fourth wave -- K-weighting of the
// second channels (R and Rs of a 5.1 track: the two odd channels that carry loudness weight; the LFE, weight 0, is K-weighted
// by nobody and gets zero records), no screen.  Same arithmetic per channel in the same order: records, candidates and peaks
// are those of ROLE 0.
template <bool DO_KW, bool DO_TP, int MT, bool SMALL, int INF, int NCH, int ROLE>
__device__ __forceinline__ void fused_body(char *smem, const TrackDesc *__restrict__ tracks,
                                           const RateDev *__restrict__ rates,
                                           const FusedItem *__restrict__ items, uint32_t n_items,
                                           HopRec *__restrict__ recs, uint2 *__restrict__ cands,
                                           unsigned *__restrict__ cand_cnt, unsigned *__restrict__ tpb,
                                           unsigned *__restrict__ overflow,
                                           unsigned long long *__restrict__ peaks, int dbg)
{
    // dbg bit 6 (64, a production switch of the K-weighting-only build): also track the sample peak of the frames read
    // (HR_TP_SAMPLE, and 192 kHz input whose "true peak" is the sample peak)
    // other dbg bits (HR_EXPERIMENTS builds only; results are wrong when set): 1 = never list candidates,
    // 16 = no PCM / carry-table DMA after the first piece, 32 = every piece DMA re-reads piece 0, 128 = row bases rounded down to a
    // cache line (stereo)
    constexpr bool MC = NCH > 2;                       // 5.1 / 7.1 read directly: the workgroup's waves share ONE item
    constexpr int WGW = ROLE ? 4 : MC ? NCH / 2 : kFusedWaves;    // waves per workgroup
    constexpr int NSCR = ROLE ? 3 : WGW;               // of them: waves that screen (share the coefficient DMA, own a candidate list)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t item_idx = MC ? blockIdx.x : blockIdx.x * kFusedWaves + wave;
    const uint32_t slot_idx = MC ? blockIdx.x * NSCR + wave : item_idx;  // this wave's candidate list
    const int chan0 = MC ? 2 * wave : 0;               // the wave's channels: chan0, chan0 + 1 (ROLE 2: chA, chB below)
    // (stereo: the host pads every rate's item list to a multiple of 4 with n_rows == 0 items: the four waves of a
    // workgroup share tables and a barrier per step, so all of them run all steps)
    const FusedItem it = items[item_idx];
    const TrackDesc &t = tracks[it.track];
    const RateDev &r = rates[t.rate_idx];
    const bool passthru0 = r.tp_passthru0 != 0;  // else (BS.1770 bank) the sample peak bounds nothing and phase 0 is screened
    const KCoef kc = r.k;
    // ROLE 2: the two second channels this wave K-weights (those with loudness weight first), and the one it does not
    int chA = chan0, chB = chan0 + 1, chZ = -1;
    if constexpr (ROLE == 2) {
        int pick[3], n = 0;
        for (int c = 1; c < NCH; c += 2) if (t.w[c] != 0.0f) pick[n++] = c;
        for (int c = 1; c < NCH; c += 2) if (t.w[c] == 0.0f) pick[n++] = c;
        chA = pick[0]; chB = pick[1]; chZ = pick[2];
    }
    // (Measured and dropped, round 3: not K-weighting a channel without loudness weight -- the LFE of 5.1 / 7.1.  The waves of
    // a workgroup step in lock step, so the wave that holds it only waits longer at the barrier, and the wave-uniform branch
    // around every sample's K-weighting cost all waves 6 %: 355 against 378 audio-hours/s on BASELINE configs[3].)
    const uint32_t hop = it.row_len;  // frames per row: a hop, or a whole fraction of one
    // Line-aligned rows (stereo f32; host: run_measure, FusedItem::phi / row_stride / pat0).  At 44.1 kHz a hop is 35 280 bytes:
    // the 128-byte pieces of a row that starts at a hop straddle cache lines, every line is requested twice, a step apart, and a
    // third of the second requests miss the L2 (1.39 x the algorithmic traffic).  Hops h, h + 8, h + 16 ... start at the same
    // offset from the line grid, so an item may hold 64 such hops (row_stride 8) and start every row's PIECE GRID phi frames
    // early, on a line: the K-weighting then starts phi frames into piece 0 and ends phi frames later, the carry-table slices
    // and the q[0..1] hand-over shift by phi, and the tile patterns are the same sequence started at pat0 (a grid that starts
    // phi frames before a hop looks to the resampler like one that starts 16 pat0 frames after one).  Such items never hold
    // a track's last row.
    constexpr bool ALIGNED_OK = !MC && INF == 0;
    const unsigned phi = ALIGNED_OK ? it.phi : 0u;                // frames between the grid's start and the row's (wave-uniform)
    const unsigned rstride = ALIGNED_OK ? it.row_stride : 1u;     // hops between the item's rows
    const uint32_t hopg = hop + phi;                              // frames from the grid's start to the row's end
    const int T = static_cast<int>((hopg + 15) / 16);  // 16-frame pieces per row (the last may be partial)
    const int n_rows = static_cast<int>(it.n_rows);  // 0: a padding item (computes on row 0, reports nothing)
    const int last_row = n_rows > 0 ? n_rows - 1 : 0;
    const int row = lane < n_rows ? lane : last_row;  // idle lanes shadow the last row
    const uint64_t h = it.first_hop + static_cast<uint64_t>(row) * rstride;
    constexpr unsigned FB = (INF == 1 ? 2 : INF == 2 ? 3 : 4) * NCH;  // frame bytes
    constexpr unsigned PB = 16 * FB;            // piece bytes
    constexpr int NCK = PB / 16;                // 16-byte chunks per piece
    constexpr int NS = MC ? (NCK | 1) : NCK;    // LDS slots per row (MC: odd)
    constexpr int NDMA = MC ? (NS + WGW - 1) / WGW : NCK;  // DMA instructions per wave and piece set (64 rows); stereo: 8, 4 or 6
    constexpr int LPP = PB / 16;                // lanes per piece (stereo)
    constexpr int kMcPiece = 64 * NS * 16;      // MC: one piece buffer
    const char *wave_base = static_cast<const char *>(t.pcm) + (it.frame0 - phi) * static_cast<unsigned long long>(FB);  // (the grid's start)
    char *lbuf = MC ? smem : smem + wave * kPieceLds;  // MC: [2][kMcPiece], piece k in half k & 1
    char *abuf = smem + (MC ? 2 * kMcPiece : kFusedWaves * kPieceLds);    // [2][9 KiB]: step k reads half k&1 (or the resident rotations, below)
    char *gbuf = abuf + kCoefArea;        // [2][1 KiB]: step k reads half k&1
    // One tile pattern (every tile of the rate has the same coefficients): nothing to stream.  The tile's three rotations
    // (step k reads rotation k mod 3: chunk ks sits where the fragment slot of piece k-2+ks expects it) are loaded once,
    // and step 0, which screens a tile that does not exist, reads a block of zeros behind them.
    constexpr int RS = 3 * MT * 1024;  // one rotation
#ifdef HR_EXP_NO_RESIDENT
    const bool resident = false;
#else
    const bool resident = DO_TP && MT <= 2 && r.tp_np == 1;
#endif
    // 512 B (16 frames x 4 doubles) per step; grid frame i of step k is frame 16 k + i - phi of the row (the table has 16 entries of
    // padding in front: the slice of step 0 starts phi entries before it)
    const char *gsrc = reinterpret_cast<const char *>(r.g) - phi * 32u;
    const unsigned lane16 = static_cast<unsigned>(lane) * 16u;
    const int T_full = static_cast<int>(hopg / 16), rem = static_cast<int>(hopg % 16);
    // steps >= k_short accumulate only q[2], q[3] (RateDev::q_n0; with a shifted grid the first step that lies wholly behind
    // q_n0 is one later, and the phases of the loop start at multiples of 3)
    const int k_short = r.q_n0 < hop ? static_cast<int>(r.q_n0 / 16) + (phi ? 3 : 0) : T + 1;

    // ---- DMA geometry: instruction j moves LDS slots 8j..8j+7 (one 128-byte piece each, 8 lanes per piece)
    // (s16: instruction j moves slots 16j..16j+15, four lanes per 64-byte piece; position p of row l holds source chunk
    // p ^ ((l >> 2) & 3), which makes the four reads of sixteen neighbouring lanes cover all banks)
    // (MC: this wave's instruction j is instruction j * WGW + wave of the piece set; lane positions count through the padded image;
    // a lane whose position is a row's padding slot, or lies behind the image, fetches nothing: offset ~0)
    unsigned dma_off[NDMA];
#pragma unroll
    for (int j = 0; j < NDMA; j++) {
        if constexpr (MC) {
            const int pos = 64 * (j * WGW + wave) + lane;
            const int slot = pos / NS, chunk = pos - slot * NS;
            const int srow = slot < n_rows ? slot : last_row;
            dma_off[j] = (slot < 64 && chunk < NCK) ? static_cast<unsigned>(srow) * hop * FB + static_cast<unsigned>(chunk) * 16u : ~0u;  // (row_stride 1)
        } else {
            // (s24: 384 lane slots = 64 rows x 6 chunks, rows do not end with the instruction; position p of row l holds
            // source chunk (p - ((l >> 3) & 1)) mod 6: rows l and l + 8, whose 96-byte images start in the same bank, are a
            // chunk apart)
            const int gi = 64 * j + lane;
            const int slot = INF == 2 ? gi / 6 : (64 / NDMA) * j + lane / LPP;
            const int chunk = INF == 2 ? (gi % 6 + 6 - ((slot >> 3) & 1)) % 6
                            : INF == 1 ? ((lane & 3) ^ ((slot >> 2) & 3)) : ((lane & 7) ^ ((slot >> 1) & 7));
            const int srow = slot < n_rows ? slot : last_row;
            dma_off[j] = static_cast<unsigned>(srow) * rstride * hop * FB + static_cast<unsigned>(chunk) * 16u;
            // dbg 128 (experiments, results wrong): every row's pieces start on a 128-byte line -- what line-aligned pieces would buy
            if (HR_DBG(dbg, 128)) dma_off[j] = ((static_cast<unsigned>(srow) * hop * FB) & ~127u) + static_cast<unsigned>(chunk) * 16u;
        }
    }
    const unsigned rd_swz = INF == 2 ? (lane >> 3) & 1 : INF == 1 ? (lane >> 2) & 3 : (lane >> 1) & 7;  // s24: a rotation, not an xor (stereo only)
    // stereo: this lane's row in the wave's piece buffer; MC: this lane's row in buffer 0, at the first byte of the wave's channel pair
    // (ROLE 2: at the row's first byte)
    const char *my_lds = ROLE == 2 ? lbuf + lane * (NS * 16) : MC ? lbuf + lane * (NS * 16) + wave * (FB / NCH * 2) : lbuf + lane * PB;
    // MC, one piece set: this wave's share of the DMA instructions into buffer `half` (lanes with nothing to fetch stay out)
    auto mc_issue = [&](const char *src, int half) {
#pragma unroll
        for (int j = 0; j < NDMA; j++)
            if (dma_off[j] != ~0u) dma16(src, dma_off[j], lbuf + half * kMcPiece + (j * WGW + wave) * 1024);
    };

    // ---- the track's LAST rows (host: run_measure).  When the stream ends with this track its final hop is a row too, and
    // so is the ragged tail behind it when it is long enough (TrackDesc::fin_tail: row n_hops, there for its peak only).
    // The last fused row reaches past the frames that exist: TrackDesc::fin_valid = frames of that row, from its start, that
    // exist and may be fetched (the host makes a 16-byte chunk that starts inside them end inside the track);
    // fin_keep = fin_valid - tp_drop = its frames that are positions of an output (libswresample is not flushed: the last 16
    // are not).  Behind wave-uniform branches, in the last steps of that one wave:
    //   * DMA lanes whose 16-byte chunk starts at or beyond fin_valid are switched off (their LDS slots keep stale bytes);
    //   * frames >= fin_valid are zeroed as they leave LDS (they never reach the sample peak or the f16 fragments);
    //   * frames in [fin_keep, fin_valid) are real and bound the screen's error (S_x) but are no sample peak;
    //   * tiles from fin_keep / 16 on, whose positions produce no output or whose windows reach past fin_valid, are not
    //     screened for this row: the bounds-checked generic kernel starts at the first of them.
    // (a padding item, n_rows == 0, works on row 0 of the item it copies: when that is a one-row wave holding a last
    // row, the padding wave needs the same care -- it reads the same bytes)
    // One number stays live for all of this: k_fin, the first step that needs any care (the piece that holds frame
    // fin_keep; never, in every other wave).  The rare steps fetch the rest from the descriptor again.
    const bool fin_wave = it.last != 0;
    const int k_fin = fin_wave ? (static_cast<int>(t.fin_valid) > t.tp_drop ? (static_cast<int>(t.fin_valid) - t.tp_drop) / 16 : 0) : (1 << 30);
    auto fin_valid_now = [&]() {
        const TrackDesc *tp = &t;
        asm volatile("" : "+s"(tp));  // (not hoisted out of the rare branch)
        return static_cast<int>(tp->fin_valid);
    };
    auto fin_keep_now = [&]() {
        const TrackDesc *tp = &t;
        asm volatile("" : "+s"(tp));
        const int fv = static_cast<int>(tp->fin_valid), drop = tp->tp_drop;
        return fv > drop ? fv - drop : 0;
    };
    // (nothing per-lane is kept for this: the rare steps recompute what they need from the lane id)

    // the coefficient area starts as zeros: step 0 screens a tile that does not exist and must find nothing
    if constexpr (DO_TP) {
#pragma unroll
        for (int i = 0; i < 3 * MT; i++) *reinterpret_cast<uint4v *>(abuf + (resident ? 3 * RS : 0) + i * 1024 + lane * 16) = uint4v{0, 0, 0, 0};  // (all four waves write the same zeros)
        if constexpr (MT <= 2) {
            if (resident) {
                const char *asrc = reinterpret_cast<const char *>(r.tp_atab);
#pragma unroll
                for (int V = 0; V < 3; V++)
#pragma unroll
                    for (int mt = 0; mt < MT; mt++)
#pragma unroll
                        for (int ksx = 0; ksx < 3; ksx++)
                            if (((V * 3 * MT + mt * 3 + ksx) % NSCR) == wave)
                                dma16(asrc + (mt * 3 + ksx) * 1024, lane16, abuf + V * RS + (mt * 3 + (V + 1 + ksx) % 3) * 1024);
            }
        }
    }

    // ---- prologue: the 16 frames in front of the row (FIR window of tile 0, filter input history)
    float4 v[8];
    KState ks[2];
#pragma unroll
    for (int c = 0; c < 2; c++) ks[c] = KState{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    {
        const float *rowp = reinterpret_cast<const float *>(wave_base + static_cast<size_t>(row) * rstride * hop * FB);  // (the grid's first frame)
        const unsigned *rowq = reinterpret_cast<const unsigned *>(rowp);  // s16: one dword per frame
        const uint8_t *rowb = reinterpret_cast<const uint8_t *>(rowp);    // s24: six bytes per frame
        const char *rowc = reinterpret_cast<const char *>(rowp);          // MC: FB bytes per frame
        if (it.frame0 > 0 || row > 0) {  // (not the track's first row)
#pragma unroll
            for (int c = 0; c < 8; c++) {
                if constexpr (ROLE == 2) {
                    const char *f0 = rowc + (-16 + 2 * c) * static_cast<int>(FB), *f1 = rowc + (-15 + 2 * c) * static_cast<int>(FB);
                    v[c] = make_float4(mc_frame_chan<INF>(f0, chA), mc_frame_chan<INF>(f0, chB), mc_frame_chan<INF>(f1, chA), mc_frame_chan<INF>(f1, chB));
                } else if constexpr (MC) {
                    const float2 a = mc_frame_pair<INF>(rowc + (-16 + 2 * c) * static_cast<int>(FB), wave);
                    const float2 b = mc_frame_pair<INF>(rowc + (-15 + 2 * c) * static_cast<int>(FB), wave);
                    v[c] = make_float4(a.x, a.y, b.x, b.y);
                } else if constexpr (INF == 2) {
                    const float2 a = s24_frame(rowb + (-16 + 2 * c) * 6), b = s24_frame(rowb + (-15 + 2 * c) * 6);
                    v[c] = make_float4(a.x, a.y, b.x, b.y);
                } else if constexpr (INF == 1) {
                    v[c] = widen_s16x2(rowq[-16 + 2 * c], rowq[-15 + 2 * c]);
                } else {
                    v[c] = *reinterpret_cast<const float4 *>(rowp - 32 + 4 * c);
                }
            }
            ks[0].x1 = v[7].z; ks[1].x1 = v[7].w;
            ks[0].x2 = v[7].x; ks[1].x2 = v[7].y;
        } else if (t.left_tail) {
#pragma unroll
            for (int c = 0; c < 8; c++)
                v[c] = make_float4(static_cast<float>(t.left_tail[(2 * c) * NCH + chA]), static_cast<float>(t.left_tail[(2 * c) * NCH + chB]),
                                   static_cast<float>(t.left_tail[(2 * c + 1) * NCH + chA]), static_cast<float>(t.left_tail[(2 * c + 1) * NCH + chB]));
            ks[0].x1 = v[7].z; ks[1].x1 = v[7].w;
            ks[0].x2 = v[7].x; ks[1].x2 = v[7].y;
        } else if (t.tp_reflect) {
            // start of the stream: libswresample reflects, x[-j] = x[j]; the K-weighting starts from silence
#pragma unroll
            for (int c = 0; c < 8; c++) {
                if constexpr (ROLE == 2) {
                    // (the reflected frames only matter to the screen; this wave K-weights from silence and needs none of them)
                    v[c] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                } else if constexpr (MC) {
                    const float2 a = mc_frame_pair<INF>(rowc + (16 - 2 * c) * static_cast<int>(FB), wave);
                    const float2 b = mc_frame_pair<INF>(rowc + (15 - 2 * c) * static_cast<int>(FB), wave);
                    v[c] = make_float4(a.x, a.y, b.x, b.y);
                } else if constexpr (INF == 2) {
                    const float2 a = s24_frame(rowb + (16 - 2 * c) * 6), b = s24_frame(rowb + (15 - 2 * c) * 6);
                    v[c] = make_float4(a.x, a.y, b.x, b.y);
                } else if constexpr (INF == 1) {
                    v[c] = widen_s16x2(rowq[16 - 2 * c], rowq[15 - 2 * c]);
                } else {
                    const float2 a = *reinterpret_cast<const float2 *>(rowp + 2 * (16 - 2 * c));
                    const float2 b = *reinterpret_cast<const float2 *>(rowp + 2 * (15 - 2 * c));
                    v[c] = make_float4(a.x, a.y, b.x, b.y);
                }
            }
        } else {
            // BS.1770: silence in front of the stream
#pragma unroll
            for (int c = 0; c < 8; c++) v[c] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    }

    // Signal fragments of the three newest pieces: [slot][channel][n-tile]; piece p lives in slot p mod 3.
    // Nothing is ever moved between slots: the coefficient chunk that belongs to a slot's piece is put
    // where that slot expects it when the tile's coefficients are DMA'd into LDS (scalar address math).
    uint4v hist[3][2][2];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int n = 0; n < 2; n++) hist[a][c][n] = uint4v{0, 0, 0, 0};

    // per channel: the peak is reported per channel, so every bound is per channel too
    float s_lane[2] = {0.0f, 0.0f};  // max |x| this lane has seen (exact)
    float alb[2] = {0.0f, 0.0f};     // max |approx output| this lane has seen
    // wave-uniform, refreshed only when a lane outgrows them (rare after the first steps):
    float S_u[2] = {-1.0f, -1.0f};   // wave maximum of s_lane at the last refresh
    float A_u[2] = {0.0f, 0.0f};     // wave maximum of alb at the last refresh
    float lb[2] = {0.0f, 0.0f};      // proven lower bound of the channel's true peak
    float S_x[2] = {0.0f, 0.0f};     // wave-uniform: max |x| of the final row's frames that are read but are no output samples
    float thr_l[2][2];               // [ch][n-tile] candidate threshold of this lane's column; +inf past n_rows
#pragma unroll
    for (int c = 0; c < 2; c++) thr_l[c][0] = thr_l[c][1] = __builtin_inff();
    unsigned n_cand = 0;             // wave-uniform: candidates appended so far
    uint2 *__restrict__ my_cands = cands + static_cast<size_t>(slot_idx) * kCandCap;

    // wave-wide bounds and this lane's candidate thresholds from the per-lane maxima seen so far
    auto refresh_bounds = [&]() {
#pragma unroll
        for (int ch = 0; ch < 2; ch++) {
            S_u[ch] = wave_max_f32(s_lane[ch]);
            A_u[ch] = wave_max_f32(alb[ch]);
            const float err = fmaf(kScreenRel, fmaxf(S_u[ch], S_x[ch]), kScreenAbs);
            lb[ch] = fmaxf(lb[ch], fmaxf(passthru0 ? S_u[ch] : 0.0f, A_u[ch] - err));
            const float thr = fmaxf(lb[ch] - err, 1e-30f);  // silence is never a candidate; |x| beyond f16 -> overflow flag
            thr_l[ch][0] = (lane & 31) < n_rows ? thr : __builtin_inff();
            thr_l[ch][1] = 32 + (lane & 31) < n_rows ? thr : __builtin_inff();
        }
    };
    auto make_frags = [&](auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
#ifdef HR_EXP_NO_FRAGS
        if constexpr (DO_TP) { hist[slot][0][0][0] ^= __float_as_uint(v[0].x); return; }
#endif
        if constexpr (DO_TP) {
#pragma unroll
            for (int c = 0; c < 8; c++) {
                s_lane[0] = fmaxf(fmaxf(fabsf(v[c].x), fabsf(v[c].z)), s_lane[0]);
                s_lane[1] = fmaxf(fmaxf(fabsf(v[c].y), fabsf(v[c].w)), s_lane[1]);
            }
            unsigned f[2][8];
#pragma unroll
            for (int c = 0; c < 8; c++) {
                f[0][c] = pk_f16(v[c].x, v[c].z);
                f[1][c] = pk_f16(v[c].y, v[c].w);
            }
#pragma unroll
            for (int ch = 0; ch < 2; ch++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(f[ch][j], f[ch][j + 4], false, false);
                    hist[slot][ch][0][j] = sw[0];
                    hist[slot][ch][1][j] = sw[1];
                }
        }
    };
    make_frags(std::integral_constant<int, 2>{});  // the prologue piece is "piece -1"

    // tile bookkeeping (wave-uniform): the coefficient pattern of tile t is t mod NP
    const int NP = r.tp_np;
    int pat = ALIGNED_OK ? it.pat0 : 0;  // pattern of the next tile whose coefficients are fetched (a shifted grid starts inside the sequence)

    // coefficient chunk ks of the tile screened at step k+1 pairs with piece k-1+ks, slot (k+2+ks) mod 3
    // This wave's share (chunks i with i mod 4 == wave) of the coefficients of the tile that step k+1 screens,
    // into the half of the coefficient area that step k does not read.
    auto issue_coeff = [&](int pattern, int k, int kmod3) {
        const char *asrc = reinterpret_cast<const char *>(r.tp_atab) + static_cast<size_t>(pattern) * MT * 3 * 1024;
        char *dst = abuf + ((k + 1) & 1) * kCoefLds;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int ksx = 0; ksx < 3; ksx++)
                if (((mt * 3 + ksx) % NSCR) == wave)
                    dma16(asrc + (mt * 3 + ksx) * 1024, lane16, dst + (mt * 3 + (kmod3 + 2 + ksx) % 3) * 1024);
    };

    // first piece
    if constexpr (MC) {
        mc_issue(wave_base, 0);
    } else {
#pragma unroll
        for (int j = 0; j < NDMA; j++) dma16(wave_base, dma_off[j], lbuf + j * 1024);
    }
    // carry-table slice of step 0 (1 KiB moved, the first 512 B are the step's; the table is padded)
    if constexpr (DO_KW) {
        if (wave == 0) dma16(gsrc, lane16, gbuf);
    }

    // One step of the row loop.  SHORT: the step accumulates only q[2], q[3] (k >= k_short); the loop is split in
    // two so that each half has one K-weighting code path and the filter state stays put in its registers.
    auto step = [&](int k, auto short_c, auto slot_c) {
        constexpr bool SHORT = decltype(short_c)::value;
        constexpr int KK = decltype(slot_c)::value;  // k mod 3
        // ---- piece k has landed?  (everything this wave has in flight: the DMA and the A fragments)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        // every wave's share of the shared tables has landed, and nobody still reads the halves that this
        // step's DMA is about to overwrite
        __builtin_amdgcn_s_barrier();
        // MC: this lane's row, at the wave's channel pair, in the buffer that holds piece k
        const char *ml = MC ? my_lds + (k & 1) * kMcPiece : my_lds;
        if constexpr (ROLE == 2 && INF == 2) {
            // One 24-bit sample of channel c of frame f starts at byte FB f + 3 c of the row: inside the two dwords from the dword
            // boundary below it, s = (FB f + 3 c) & 3 bytes in.  c is wave-uniform, the parity of FB f compile time: per channel
            // two bases (even / odd frames), immediate offsets, a wave-uniform shift each.
            auto sample = [&](auto fc, int c) {
                constexpr int f = decltype(fc)::value;
                constexpr unsigned e = (FB * f) & 3u;  // 0 or 2
                const unsigned sft = (e + 3u * static_cast<unsigned>(c)) & 3u;
                const unsigned *q = reinterpret_cast<const unsigned *>(ml + 3 * c - static_cast<int>(sft) + static_cast<int>(FB * f));
                const unsigned bits = __funnelshift_r(q[0], q[1], 8u * sft);
                return static_cast<float>(static_cast<int>(bits << 8) >> 8) * (1.0f / 8388608.0f);
            };
#define HR_MC_CH(c) { using F0 = std::integral_constant<int, 2 * (c)>; using F1 = std::integral_constant<int, 2 * (c) + 1>; \
                      v[c] = make_float4(sample(F0{}, chA), sample(F0{}, chB), sample(F1{}, chA), sample(F1{}, chB)); }
            HR_MC_CH(0) HR_MC_CH(1) HR_MC_CH(2) HR_MC_CH(3) HR_MC_CH(4) HR_MC_CH(5) HR_MC_CH(6) HR_MC_CH(7)
#undef HR_MC_CH
        } else if constexpr (ROLE == 2) {
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const char *f0 = ml + FB * (2 * c), *f1 = ml + FB * (2 * c + 1);
                if constexpr (INF == 1) {
                    constexpr float sc = 1.0f / 32768.0f;
                    v[c] = make_float4(static_cast<float>(*reinterpret_cast<const int16_t *>(f0 + 2 * chA)) * sc, static_cast<float>(*reinterpret_cast<const int16_t *>(f0 + 2 * chB)) * sc,
                                       static_cast<float>(*reinterpret_cast<const int16_t *>(f1 + 2 * chA)) * sc, static_cast<float>(*reinterpret_cast<const int16_t *>(f1 + 2 * chB)) * sc);
                } else {
                    v[c] = make_float4(*reinterpret_cast<const float *>(f0 + 4 * chA), *reinterpret_cast<const float *>(f0 + 4 * chB),
                                       *reinterpret_cast<const float *>(f1 + 4 * chA), *reinterpret_cast<const float *>(f1 + 4 * chB));
                }
            }
        } else if constexpr (MC && INF == 2) {
            // Packed s24: the pair's six bytes of frame f start at ml + FB f, on an even byte.  They lie inside the two dwords
            // from the dword boundary below them; which boundary that is depends on the parity of the pair (6 p mod 4 = 0 or
            // 2, wave-uniform) and, for 5.1 (FB = 18), of the frame (compile time): two bases, immediate offsets, and a
            // wave-uniform shift per frame parity bring the 48 bits down to bit 0.
            const unsigned pm = (6u * static_cast<unsigned>(wave)) & 3u;
            const char *b0 = ml - pm, *b2 = ml + pm;
            const unsigned sh0 = 8u * pm, sh2 = 8u * (2u - pm);
            auto frame = [&](auto fc) {
                constexpr int f = decltype(fc)::value;
                constexpr bool odd = ((FB * f) & 3u) == 2u;
                const unsigned *q = reinterpret_cast<const unsigned *>(odd ? b2 + (FB * f - 2) : b0 + FB * f);
                const unsigned d0 = q[0], d1 = q[1], sh = odd ? sh2 : sh0;
                const unsigned lo = __funnelshift_r(d0, d1, sh), hi = d1 >> sh;
                constexpr float sc = 1.0f / 8388608.0f;
                return make_float2(static_cast<float>(static_cast<int>(lo << 8) >> 8) * sc,
                                   static_cast<float>(static_cast<int>(__funnelshift_r(lo, hi, 24) << 8) >> 8) * sc);
            };
#define HR_MC_PAIR(c) { const float2 a = frame(std::integral_constant<int, 2 * (c)>{}), b = frame(std::integral_constant<int, 2 * (c) + 1>{}); \
                        v[c] = make_float4(a.x, a.y, b.x, b.y); }
            HR_MC_PAIR(0) HR_MC_PAIR(1) HR_MC_PAIR(2) HR_MC_PAIR(3) HR_MC_PAIR(4) HR_MC_PAIR(5) HR_MC_PAIR(6) HR_MC_PAIR(7)
#undef HR_MC_PAIR
        } else if constexpr (MC && INF == 1) {
#pragma unroll
            for (int c = 0; c < 8; c++)
                v[c] = widen_s16x2(*reinterpret_cast<const unsigned *>(ml + FB * (2 * c)), *reinterpret_cast<const unsigned *>(ml + FB * (2 * c + 1)));
        } else if constexpr (MC) {
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const float2 a = *reinterpret_cast<const float2 *>(ml + FB * (2 * c)), b = *reinterpret_cast<const float2 *>(ml + FB * (2 * c + 1));
                v[c] = make_float4(a.x, a.y, b.x, b.y);
            }
        } else if constexpr (INF == 2) {
            unsigned w[25];
            const char *rot = my_lds + rd_swz * 16;  // source chunk q sits at position (q + rd_swz) mod 6
#pragma unroll
            for (int q = 0; q < 6; q++) {
                const uint4v u = *reinterpret_cast<const uint4v *>(q < 5 ? rot + q * 16 : rot + 80 - rd_swz * 96);
                w[4 * q] = u[0]; w[4 * q + 1] = u[1]; w[4 * q + 2] = u[2]; w[4 * q + 3] = u[3];
            }
            w[24] = 0;
            v[0] = make_float4(s24_sample<0>(w), s24_sample<1>(w), s24_sample<2>(w), s24_sample<3>(w));
            v[1] = make_float4(s24_sample<4>(w), s24_sample<5>(w), s24_sample<6>(w), s24_sample<7>(w));
            v[2] = make_float4(s24_sample<8>(w), s24_sample<9>(w), s24_sample<10>(w), s24_sample<11>(w));
            v[3] = make_float4(s24_sample<12>(w), s24_sample<13>(w), s24_sample<14>(w), s24_sample<15>(w));
            v[4] = make_float4(s24_sample<16>(w), s24_sample<17>(w), s24_sample<18>(w), s24_sample<19>(w));
            v[5] = make_float4(s24_sample<20>(w), s24_sample<21>(w), s24_sample<22>(w), s24_sample<23>(w));
            v[6] = make_float4(s24_sample<24>(w), s24_sample<25>(w), s24_sample<26>(w), s24_sample<27>(w));
            v[7] = make_float4(s24_sample<28>(w), s24_sample<29>(w), s24_sample<30>(w), s24_sample<31>(w));
        } else if constexpr (INF == 1) {
            uint4v u[4];
#pragma unroll
            for (int q = 0; q < 4; q++) u[q] = *reinterpret_cast<const uint4v *>(my_lds + ((q ^ rd_swz) * 16));
#pragma unroll
            for (int c = 0; c < 8; c++) v[c] = widen_s16x2(u[c >> 1][(c & 1) * 2], u[c >> 1][(c & 1) * 2 + 1]);
        } else {
#pragma unroll
            for (int c = 0; c < 8; c++) v[c] = *reinterpret_cast<const float4 *>(my_lds + ((c ^ rd_swz) * 16));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (k >= k_fin) {  // (wave-uniform; the last steps of the wave that holds a track's last fused row)
            const int lim = lane >= last_row ? fin_valid_now() - 16 * k : (1 << 30);  // frames of this piece that exist (idle lanes shadow the last row)
#pragma unroll
            for (int c = 0; c < 8; c++) {
                v[c].x = 2 * c < lim ? v[c].x : 0.0f;
                v[c].y = 2 * c < lim ? v[c].y : 0.0f;
                v[c].z = 2 * c + 1 < lim ? v[c].z : 0.0f;
                v[c].w = 2 * c + 1 < lim ? v[c].w : 0.0f;
            }
        }
        // a shifted grid: the row starts phi frames into piece 0.  Its K-weighting starts there, from the two frames in front of
        // it, frame by frame straight from LDS (the fused block below takes whole pieces: this step only screens)
        if constexpr (ALIGNED_OK && DO_KW && !SHORT) {
            if (phi && k == 0) {
                auto frame = [&](unsigned i) { return *reinterpret_cast<const float2 *>(my_lds + (((i >> 1) ^ rd_swz) * 16) + (i & 1) * 8); };
                const float2 xm1 = frame(phi - 1), xm2 = frame(phi - 2);  // (phi is even and not 0 here)
                ks[0].x1 = static_cast<double>(xm1.x); ks[1].x1 = static_cast<double>(xm1.y);
                ks[0].x2 = static_cast<double>(xm2.x); ks[1].x2 = static_cast<double>(xm2.y);
                for (unsigned i = phi; i < 16; i++) {
                    const float2 x = frame(i);
                    const G4 g4 = lds_g4(gbuf + i * 32);
                    kw_sample(ks[0], kc, x.x, g4);
                    kw_sample(ks[1], kc, x.y, g4);
                }
            }
        }
        // the last piece of a hop may be partial (44.1 kHz: 4410 = 275*16 + 10): K-weight its frames one at a
        // time straight from LDS, before the buffer is handed back to the DMA
        if (DO_KW && rem && k == T_full) {
            const char *gk = gbuf + (k & 1) * kCarryLds;
            for (int i = 0; i < rem; i++) {
                float2 x;
                if constexpr (ROLE == 2) {
                    x = make_float2(mc_frame_chan<INF>(ml + FB * i, chA), mc_frame_chan<INF>(ml + FB * i, chB));
                } else if constexpr (MC && INF == 2) {
                    uint8_t fb[6];
#pragma unroll
                    for (int bb = 0; bb < 6; bb++) fb[bb] = *reinterpret_cast<const uint8_t *>(ml + FB * i + bb);
                    x = s24_frame(fb);
                } else if constexpr (MC && INF == 1) {
                    const float4 w4 = widen_s16x2(*reinterpret_cast<const unsigned *>(ml + FB * i), 0u);
                    x = make_float2(w4.x, w4.y);
                } else if constexpr (MC) {
                    x = *reinterpret_cast<const float2 *>(ml + FB * i);
                } else if constexpr (INF == 2) {
                    uint8_t fb[6];
#pragma unroll
                    for (int bb = 0; bb < 6; bb++) {
                        const unsigned byte = 6u * i + bb, pos = ((byte >> 4) + rd_swz) % 6u;
                        fb[bb] = *reinterpret_cast<const uint8_t *>(my_lds + pos * 16 + (byte & 15));
                    }
                    x = s24_frame(fb);
                } else if constexpr (INF == 1) {
                    const float4 w4 = widen_s16x2(*reinterpret_cast<const unsigned *>(my_lds + (((i >> 2) ^ rd_swz) * 16) + (i & 3) * 4), 0u);
                    x = make_float2(w4.x, w4.y);
                } else {
                    x = *reinterpret_cast<const float2 *>(my_lds + (((i >> 1) ^ rd_swz) * 16) + (i & 1) * 8);
                }
                if constexpr (!SHORT) {
                    const G4 g4 = lds_g4(gk + i * 32);
                    kw_sample(ks[0], kc, x.x, g4);
                    if constexpr (ROLE != 1) kw_sample(ks[1], kc, x.y, g4);
                } else {
                    const double2 g = *reinterpret_cast<const double2 *>(gk + i * 32 + 16);
                    kw_sample_short(ks[0], kc, x.x, g.x, g.y);
                    if constexpr (ROLE != 1) kw_sample_short(ks[1], kc, x.y, g.x, g.y);
                }
            }
        }
        // the piece now lives in registers: hand the LDS buffer back to the DMA for piece k+1
        if (k < T && !HR_DBG(dbg, 16)) {
            const char *src = wave_base + (HR_DBG(dbg, 32) ? 0 : static_cast<size_t>(k + 1) * PB);  // dbg 32 (experiments): re-read piece 0 (cache hits)
            if (k + 1 >= k_fin) {
                // piece k+1 of the final row reaches past the track: chunks that start beyond its end are not fetched
                const int end_bytes = fin_valid_now() * static_cast<int>(FB) - (k + 1) * static_cast<int>(PB);  // bytes of this piece that exist
#pragma unroll
                for (int j = 0; j < NDMA; j++) {
                    // (the DMA geometry above: which row and which source chunk this lane serves in instruction j)
                    if constexpr (MC) {
                        const int pos = 64 * (j * WGW + wave) + lane;
                        const int slot = pos / NS, chunk = pos - slot * NS;
                        const bool oob = slot >= last_row && chunk * 16 >= end_bytes;
                        if (dma_off[j] != ~0u && !oob) dma16(src, dma_off[j], lbuf + ((k + 1) & 1) * kMcPiece + (j * WGW + wave) * 1024);
                        continue;
                    }
                    const int gi = 64 * j + lane;
                    const int slot = INF == 2 ? gi / 6 : (64 / NDMA) * j + lane / LPP;
                    const int chunk = INF == 2 ? (gi % 6 + 6 - ((slot >> 3) & 1)) % 6
                                    : INF == 1 ? ((lane & 3) ^ ((slot >> 2) & 3)) : ((lane & 7) ^ ((slot >> 1) & 7));
                    const bool oob = slot >= last_row && chunk * 16 >= end_bytes;
                    if (!oob) dma16(src, dma_off[j], lbuf + j * 1024);  // (inactive lanes transfer nothing: their LDS slots keep stale bytes)
                }
            } else if constexpr (MC) {
                mc_issue(src, (k + 1) & 1);
            } else {
#pragma unroll
                for (int j = 0; j < NDMA; j++) dma16(src, dma_off[j], lbuf + j * 1024);
            }
            if constexpr (DO_KW) {
                if (wave == 0) dma16(gsrc + static_cast<size_t>(k + 1) * 512, lane16, gbuf + ((k + 1) & 1) * kCarryLds);
            }
            if constexpr (DO_TP) {
#ifndef HR_EXP_NO_COEF_DMA
                if (!resident) issue_coeff(pat, k, KK);
#endif
                pat = pat + 1 == NP ? 0 : pat + 1;
            }
        }
        // fragments of piece k go into slot k mod 3.  In the last steps of the wave that holds the track's last fused row
        // (wave-uniform, rare) the frames at or beyond fin_keep are no samples of the output: the lane's sample peak is put
        // back to what it was plus the frames that count, the others go into S_x, and the bounds are refreshed below so
        // that the thresholds see S_x before this step's tile is screened.
        float force_refresh = 0.0f;
        const float keep0 = s_lane[0], keep1 = s_lane[1];
        make_frags(slot_c);
        if (DO_TP && k >= k_fin) {
            const int keep = lane >= last_row ? fin_keep_now() - 16 * k : 16;
            float in0 = keep0, in1 = keep1, ex0 = 0.0f, ex1 = 0.0f;
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const bool k0 = 2 * c < keep, k1 = 2 * c + 1 < keep;
                in0 = fmaxf(fmaxf(k0 ? fabsf(v[c].x) : 0.0f, k1 ? fabsf(v[c].z) : 0.0f), in0);
                in1 = fmaxf(fmaxf(k0 ? fabsf(v[c].y) : 0.0f, k1 ? fabsf(v[c].w) : 0.0f), in1);
                ex0 = fmaxf(fmaxf(k0 ? 0.0f : fabsf(v[c].x), k1 ? 0.0f : fabsf(v[c].z)), ex0);
                ex1 = fmaxf(fmaxf(k0 ? 0.0f : fabsf(v[c].y), k1 ? 0.0f : fabsf(v[c].w)), ex1);
            }
            s_lane[0] = in0;
            s_lane[1] = in1;
            S_x[0] = fmaxf(S_x[0], wave_max_f32(ex0));
            S_x[1] = fmaxf(S_x[1], wave_max_f32(ex1));
            force_refresh = 1.0f;
        }
        if constexpr (!DO_TP) {
            if (dbg & 64) {
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    s_lane[0] = fmaxf(fmaxf(fabsf(v[c].x), fabsf(v[c].z)), s_lane[0]);
                    s_lane[1] = fmaxf(fmaxf(fabsf(v[c].y), fabsf(v[c].w)), s_lane[1]);
                }
            }
        }

        if constexpr (DO_TP) {
            // Bounds, per channel.  S_u >= every |x| of the window (it includes piece k) so err bounds
            // |approx - exact| of everything screened so far; lb is a proven lower bound of the true peak
            // (S itself is one: phase 0 of the bank passes the samples through), so an output below lb - err
            // cannot be the maximum.  The wave-wide values are recomputed only when some lane has outgrown them.
            // (one compare: a VALU compare that feeds a scalar branch costs as much as half a dozen f32 operations)
            const float over = fmaxf(fmaxf(fmaxf(s_lane[0] - S_u[0], s_lane[1] - S_u[1]), fmaxf(alb[0] - A_u[0], alb[1] - A_u[1])), force_refresh);
            const bool stale = over > 0.0f;
#ifdef HR_EXP_NO_BOUNDS
            asm volatile("" :: "v"(over));
            if (k == 0) refresh_bounds();
#else
            if (__ballot(stale) != 0ull) refresh_bounds();
#endif
        }

        // ---- the hot block, branch-free so that the scheduler can run the f64 K-weighting of piece k under
        // the MFMAs of tile k-1 (pieces k-2, k-1, k): K3 screen ...
        // where this step's coefficient tile sits in the coefficient area
        const unsigned coef_off = resident ? static_cast<unsigned>(k == 0 ? 3 : KK) * RS : static_cast<unsigned>(k & 1) * kCoefLds;
        float m0[MT][2][2];
        auto screen = [&]() {
            if constexpr (DO_TP) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    uint4v af[3];
#pragma unroll
                    for (int k3 = 0; k3 < 3; k3++) af[k3] = *reinterpret_cast<const uint4v *>(abuf + coef_off + (mt * 3 + k3) * 1024 + lane * 16);
#pragma unroll
                    for (int ch = 0; ch < 2; ch++) {
#pragma unroll
                        for (int nt = 0; nt < 2; nt++) {
                            floatx16 acc = {0};
#pragma unroll
                            for (int k3 = 0; k3 < 3; k3++)
                                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_half8(af[k3]), as_half8(hist[k3][ch][nt]), acc, 0, 0, 0);
                            // this lane's 16 values are column (lane&31) of n-tile nt, i.e. wave row 32*nt + (lane&31),
                            // output rows 32*mt + 4*(lane>>5) + {0..3} + 8*{0..3} of the tile
                            float m = fmaxf(fmaxf(fabsf(acc[0]), fabsf(acc[1])), fmaxf(fabsf(acc[2]), fabsf(acc[3])));
                            if (!(SMALL && mt == MT - 1)) {
#pragma unroll
                                for (int i = 4; i < 16; i += 2) m = fmaxf(fmaxf(fabsf(acc[i]), fabsf(acc[i + 1])), m);
                            }
                            m0[mt][ch][nt] = m;
                        }
                        alb[ch] = fmaxf(fmaxf(m0[mt][ch][0], m0[mt][ch][1]), alb[ch]);
                    }
                }
            }
        };
        // Both at once, software-pipelined by hand: each accumulator's three MFMAs are issued, then a slice of the
        // f64 K-weighting (which does not depend on them) keeps the VALU busy while the matrix pipe works, then
        // the accumulator is reduced.  The fences pin that order; the carry-table values of the next slice are
        // fetched one slice ahead.  (Without DO_TP this is just the K-weighting in eight slices.)
        auto fused_block = [&]() {
            const unsigned g_at = lds_addr(gbuf) + static_cast<unsigned>(k & 1) * kCarryLds;  // wave-uniform: broadcast reads
            const unsigned a_at = lds_addr(abuf) + coef_off + lane16;
            struct GC { uint4v r[4]; };  // rows 2c, 2c+1 of the slice: {g0 g1}, {g2 g3} each (SHORT: only {g2 g3})
            auto load_gc = [&](int c) {
                GC o;
                if constexpr (!SHORT) {
                    o.r[0] = lds_ld128(g_at, (2 * c) * 32);
                    o.r[1] = lds_ld128(g_at, (2 * c) * 32 + 16);
                    o.r[2] = lds_ld128(g_at, (2 * c + 1) * 32);
                    o.r[3] = lds_ld128(g_at, (2 * c + 1) * 32 + 16);
                } else {
#ifndef HR_EXP_NO_Q
                    o.r[1] = lds_ld128(g_at, (2 * c) * 32 + 16);
                    o.r[3] = lds_ld128(g_at, (2 * c + 1) * 32 + 16);
#else
                    o.r[1] = o.r[3] = uint4v{0, 0, 0, 0};
#endif
                    o.r[0] = o.r[2] = uint4v{0, 0, 0, 0};
                }
                return o;
            };
            auto wait_gc = [&](GC &g) {
                if constexpr (!SHORT) lds_wait(g.r[0], g.r[1], g.r[2], g.r[3]);
#ifndef HR_EXP_NO_Q
                else lds_wait(g.r[1], g.r[3]);
#endif
            };
            // sample i of slice c: i = 0, 1 are the two channels of frame 2c, i = 2, 3 of frame 2c+1
            auto ksample = [&](int c, int i, const GC &g) {
                if (ROLE == 1 && (i & 1)) return;  // (a screening wave of the four-wave workgroup K-weights its pair's first channel only)
                const float x = i == 0 ? v[c].x : i == 1 ? v[c].y : i == 2 ? v[c].z : v[c].w;
                const double2 g23 = as_double2(g.r[i < 2 ? 1 : 3]);
                if constexpr (!SHORT) {
                    const double2 g01 = as_double2(g.r[i < 2 ? 0 : 2]);
                    kw_sample(ks[i & 1], kc, x, G4{g01.x, g01.y, g23.x, g23.y});
                } else {
                    kw_sample_short(ks[i & 1], kc, x, g23.x, g23.y);
                }
            };
            constexpr int NG = DO_TP ? 4 * MT : 8;     // accumulator groups of the tile
            constexpr int CPG = NG >= 8 ? 1 : 8 / NG;  // K slices (of 2 frames x 2 channels) per group
            uint4v af[3];
            if constexpr (DO_TP) {
#pragma unroll
                for (int k3 = 0; k3 < 3; k3++) af[k3] = lds_ld128(a_at, k3 * 1024);
            }
            GC gcur = load_gc(0);
            if constexpr (DO_TP) lds_wait(af[0], af[1], af[2]);
#pragma unroll
            for (int g = 0; g < NG; g++) {
                const int mt = g >> 2, ch = (g >> 1) & 1, nt = g & 1;
                // MFMA 1, sample, MFMA 2 (needs 1's result: the sample in between covers its latency), sample,
                // MFMA 3, the slice's other samples
                floatx16 acc = {0};
                const int c0 = g * CPG;
                if (c0 < 8) wait_gc(gcur);
                GC gnext = gcur;
                if (c0 + 1 < 8) gnext = load_gc(c0 + 1);
#pragma unroll
                for (int k3 = 0; k3 < 3; k3++) {
                    if constexpr (DO_TP) {
#if defined(HR_EXP_NO_MFMA2)
                        asm volatile("" : "+v"(acc) : "v"(af[k3]), "v"(hist[k3][ch][nt]));
#elif !defined(HR_EXP_NO_MFMA)
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_half8(af[k3]), as_half8(hist[k3][ch][nt]), acc, 0, 0, 0);
#else
                        acc[k3] += __uint_as_float(af[k3][0] ^ hist[k3][ch][nt][0]);
#endif
                        __builtin_amdgcn_sched_barrier(0);
                    }
#ifndef HR_EXP_NO_KW
                    if (c0 < 8) {
                        ksample(c0, k3, gcur);
                        if (k3 == 2) ksample(c0, 3, gcur);
                    }
#endif
                    if constexpr (DO_TP) __builtin_amdgcn_sched_barrier(0);
                }
                gcur = gnext;
#pragma unroll
                for (int j = 1; j < CPG; j++) {
                    const int c = c0 + j;
                    if (c < 8) {
                        wait_gc(gcur);
                        gnext = gcur;
                        if (c + 1 < 8) gnext = load_gc(c + 1);
#pragma unroll
                        for (int i = 0; i < 4; i++) ksample(c, i, gcur);
                        gcur = gnext;
                    }
                }
                if constexpr (DO_TP) {
                    __builtin_amdgcn_sched_barrier(0);
                    // all sixteen accumulator registers stay reserved until the MFMAs have retired
                    asm volatile("" : "+v"(acc));
                    // this lane's 16 values are column (lane&31) of n-tile nt, i.e. wave row 32*nt + (lane&31),
                    // output rows 32*mt + 4*(lane>>5) + {0..3} + 8*{0..3} of the tile
                    float m = fmaxf(fmaxf(fabsf(acc[0]), fabsf(acc[1])), fmaxf(fabsf(acc[2]), fabsf(acc[3])));
#ifndef HR_EXP_NO_REDUCE
                    if (!(SMALL && mt == MT - 1)) {
#pragma unroll
                        for (int i = 4; i < 16; i += 2) m = fmaxf(fmaxf(fabsf(acc[i]), fabsf(acc[i + 1])), m);
                    }
#endif
                    m0[mt][ch][nt] = m;
                    if (nt == 1) alb[ch] = fmaxf(fmaxf(m0[mt][ch][0], m0[mt][ch][1]), alb[ch]);
#ifndef HR_EXP_NO_AF
                    if ((g & 3) == 3 && mt + 1 < MT) {  // the block's MFMAs are done: fetch the next block's fragments
#pragma unroll
                        for (int k3 = 0; k3 < 3; k3++) af[k3] = lds_ld128(a_at, ((mt + 1) * 3 + k3) * 1024);
                        lds_wait(af[0], af[1], af[2]);
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        // Tiles T-2 and T-1 (screened at steps T-1 and T) of a final row reach into frames that do not exist: that row's
        // columns of the signal fragments become zeros, so its outputs of these two tiles are exact zeros -- no bound, no
        // candidate.  (A lane's fragments of n-tile nt belong to wave row 32*nt + (lane & 31); the row's remaining
        // positions stay with the generic kernel.)
        if (DO_TP && k > k_fin) {  // (tile k - 1 >= fin_keep / 16)
#pragma unroll
            for (int nt = 0; nt < 2; nt++) {
                const bool drop = 32 * nt + (lane & 31) >= last_row;
#pragma unroll
                for (int a = 0; a < 3; a++)
#pragma unroll
                    for (int ch = 0; ch < 2; ch++)
#pragma unroll
                        for (int q = 0; q < 4; q++) hist[a][ch][nt][q] = drop ? 0u : hist[a][ch][nt][q];
            }
        }
        const bool fused_step = DO_KW && k < T_full && !(phi && k == 0);
        if (fused_step) {
            fused_block();
        } else {
            screen();
        }
        if constexpr (DO_TP) {
            // Without a pass-through phase nothing bounds the peak from below until a tile has been screened: the
            // first tile with signal refreshes the bounds BEFORE its groups are listed (else every group would be one).
            if (!passthru0 && (lb[0] == 0.0f || lb[1] == 0.0f)) refresh_bounds();
            // groups that can still hold the maximum go on the wave's list (columns past n_rows have an infinite
            // threshold; step 0 screens zeros against an infinite threshold)
            float margin = -__builtin_inff();  // max over the lane's groups of (group maximum - its threshold)
#pragma unroll
            for (int ch = 0; ch < 2; ch++)
#pragma unroll
                for (int nt = 0; nt < 2; nt++) {
                    float mm = m0[0][ch][nt];
#pragma unroll
                    for (int mt = 1; mt < MT; mt++) mm = fmaxf(mm, m0[mt][ch][nt]);
                    margin = fmaxf(margin, mm - thr_l[ch][nt]);
                }
#ifdef HR_EXP_NO_CAND
            asm volatile("" :: "v"(margin));
            const unsigned long long any = 0ull;
#else
            const unsigned long long any = __ballot(margin >= 0.0f);
#endif
            if (any != 0ull && !HR_DBG(dbg, 1)) {  // wave-uniform, rare
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int ch = 0; ch < 2; ch++)
#pragma unroll
                        for (int nt = 0; nt < 2; nt++) {
                            const bool flag = m0[mt][ch][nt] >= thr_l[ch][nt];
                            const unsigned long long mask = __ballot(flag);
                            if (mask == 0ull) continue;
                            const unsigned slot = n_cand + __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(mask >> 32),
                                                                                      __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(mask), 0));
                            if (flag && slot < kCandCap)
                                my_cands[slot] = make_uint2(static_cast<unsigned>(ch) | static_cast<unsigned>(32 * nt + (lane & 31)) << 1 |
                                                                static_cast<unsigned>(mt) << 7 | static_cast<unsigned>(lane >> 5) << 9 |
                                                                static_cast<unsigned>(k - 1) << 10,
                                                            __float_as_uint(m0[mt][ch][nt]));
                            n_cand += static_cast<unsigned>(__popcll(mask));
                        }
            }
        }
    };

    // The loop runs in triples so that k mod 3 (the fragment slot of piece k) is a compile-time constant;
    // both phases start at a multiple of 3 (q_n0 is a multiple of 48 frames).
    auto run = [&](int k_begin, int k_end, auto short_c) {
        for (int k = k_begin; k < k_end; k += 3) {
            step(k, short_c, std::integral_constant<int, 0>{});
            if (k + 1 < k_end) step(k + 1, short_c, std::integral_constant<int, 1>{});
            if (k + 2 < k_end) step(k + 2, short_c, std::integral_constant<int, 2>{});
        }
    };
    {
        const int k_split = k_short < T + 1 ? k_short : T + 1;
        run(0, k_split, std::false_type{});
        if (DO_KW && k_split <= T) {
            // from here on q[0], q[1] ride on q[2], q[3]: g_j = ab[j][0] g_2 + ab[j][1] g_3 for the rest of the hop
#pragma unroll
            for (int c = 0; c < 2; c++) {
                ks[c].q0 -= fma(r.q_ab[0][0], ks[c].q2, r.q_ab[0][1] * ks[c].q3);
                ks[c].q1 -= fma(r.q_ab[1][0], ks[c].q2, r.q_ab[1][1] * ks[c].q3);
            }
        }
        run(k_split, T + 1, std::true_type{});
    }

    if (DO_KW && lane < n_rows && h < t.sub_from + (t.n_hops - t.sub_from) * t.parts) {  // (rows behind the last hop are there for their peak only)
#pragma unroll
        for (int c = 0; c < (ROLE == 1 ? 1 : 2); c++) {
            if (k_short <= T) {
                ks[c].q0 += fma(r.q_ab[0][0], ks[c].q2, r.q_ab[0][1] * ks[c].q3);
                ks[c].q1 += fma(r.q_ab[1][0], ks[c].q2, r.q_ab[1][1] * ks[c].q3);
            }
            HopRec &o = recs[t.rec_base + (kPreRec + h) * NCH + (c ? chB : chA)];
            o.e0 = ks[c].e0;
            o.q[0] = ks[c].q0; o.q[1] = ks[c].q1; o.q[2] = ks[c].q2; o.q[3] = ks[c].q3;
            o.s[0] = ks[c].y1; o.s[1] = ks[c].y1 - ks[c].y2; o.s[2] = ks[c].z1; o.s[3] = ks[c].z1 - ks[c].z2;
        }
        if constexpr (ROLE == 2) {  // the second channel nobody K-weights (no loudness weight: the LFE): zero records, not stale ones
            HopRec &o = recs[t.rec_base + (kPreRec + h) * NCH + chZ];
            o.e0 = 0.0;
#pragma unroll
            for (int j = 0; j < 4; j++) { o.q[j] = 0.0; o.s[j] = 0.0; }
        }
    }
    if constexpr (!DO_TP) {
        if (dbg & 64) {  // every value seen is a frame of this track (or a reflected copy of one): an exact partial maximum
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const float Sf = wave_max_f32(s_lane[c]);
                if (lane == 0 && n_rows > 0 && Sf > 0.0f)
                    atomicMax(&peaks[static_cast<size_t>(it.track) * kMaxCh + chan0 + c],
                              static_cast<unsigned long long>(__double_as_longlong(static_cast<double>(Sf))));
            }
        }
    }
    if constexpr (DO_TP) {
        // hand the wave's list and bounds over to k_refine_cands (next kernel on the stream)
        float Sf[2], Af[2];
#pragma unroll
        for (int c = 0; c < 2; c++) {
            Sf[c] = wave_max_f32(s_lane[c]);
            Af[c] = wave_max_f32(alb[c]);
            lb[c] = fmaxf(lb[c], fmaxf(passthru0 ? Sf[c] : 0.0f, Af[c] - fmaf(kScreenRel, fmaxf(Sf[c], S_x[c]), kScreenAbs)));
        }
        // A list that ran out of slots marks the WAVE (the host has k_true_peak_rows redo this wave's positions exactly, the
        // listed groups are still refined); samples beyond the f16 range void the screen's bounds for the whole track.
        // (the track's flag word tells the host that some wave's count carries the bit: it reads the counts only then)
        if (lane == 0) cand_cnt[slot_idx] = n_cand <= kCandCap ? n_cand : (kCandCap | (n_rows > 0 ? kCandOverflowBit : 0u));
        if (lane == 0 && n_rows > 0) {  // (padding waves list nothing and report nothing)
            if (n_cand > kCandCap) atomicOr(&overflow[it.track], 2u);
            if (fmaxf(Sf[0], S_x[0]) >= kF16Safe || fmaxf(Sf[1], S_x[1]) >= kF16Safe) atomicOr(&overflow[it.track], 1u);
#pragma unroll
            for (int c = 0; c < 2; c++) {
                // non-negative floats order like their bit patterns
                atomicMax(&tpb[(static_cast<size_t>(it.track) * kMaxCh + chan0 + c) * 2 + 0], __float_as_uint(lb[c]));
                atomicMax(&tpb[(static_cast<size_t>(it.track) * kMaxCh + chan0 + c) * 2 + 1], __float_as_uint(fmaxf(Sf[c], S_x[c])));
                // phase-0 outputs are the samples themselves: exact, and not part of the screen
                if (passthru0 && Sf[c] > 0.0f)
                    atomicMax(&peaks[static_cast<size_t>(it.track) * kMaxCh + chan0 + c],
                              static_cast<unsigned long long>(__double_as_longlong(static_cast<double>(Sf[c]))));
            }
        }
    }
}

// every wave of the workgroup the same work on its own item (stereo) or channel pair (5.1 / 7.1, NCH / 2 waves)
template <bool DO_KW, bool DO_TP, int MT, bool SMALL, int INF = 0, int NCH = 2>
__global__ void __launch_bounds__(64 * (NCH > 2 ? NCH / 2 : kFusedWaves), 8 / (NCH > 2 ? NCH / 2 : kFusedWaves)) k_fused_f32s(const TrackDesc *__restrict__ tracks,
                                                       const RateDev *__restrict__ rates,
                                                       const FusedItem *__restrict__ items, uint32_t n_items,
                                                       HopRec *__restrict__ recs, uint2 *__restrict__ cands,
                                                       unsigned *__restrict__ cand_cnt, unsigned *__restrict__ tpb,
                                                       unsigned *__restrict__ overflow,
                                                       unsigned long long *__restrict__ peaks, int dbg)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];  // kGroupLds
    fused_body<DO_KW, DO_TP, MT, SMALL, INF, NCH, 0>(smem, tracks, rates, items, n_items, recs, cands, cand_cnt, tpb, overflow, peaks, dbg);
}

// 5.1 with FOUR waves per workgroup: three screening waves (ROLE 1) and one K-weighting wave (ROLE 2); two workgroups fill
// the eight wave slots of a CU
template <int MT, bool SMALL, int INF>
__global__ void __launch_bounds__(256, 2) k_fused_51x4(const TrackDesc *__restrict__ tracks, const RateDev *__restrict__ rates,
                                                       const FusedItem *__restrict__ items, uint32_t n_items,
                                                       HopRec *__restrict__ recs, uint2 *__restrict__ cands,
                                                       unsigned *__restrict__ cand_cnt, unsigned *__restrict__ tpb,
                                                       unsigned *__restrict__ overflow,
                                                       unsigned long long *__restrict__ peaks, int dbg)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // (measured and dropped: rotating which hardware wave takes ROLE 2 by workgroup index, so that the K-weighting waves of a
    // CU's two workgroups sit on different SIMDs -- 7.25 ms either way on BASELINE configs[3])
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) < 3)
        fused_body<true, true, MT, SMALL, INF, 6, 1>(smem, tracks, rates, items, n_items, recs, cands, cand_cnt, tpb, overflow, peaks, dbg);
    else
        fused_body<true, false, MT, SMALL, INF, 6, 2>(smem, tracks, rates, items, n_items, recs, cands, cand_cnt, tpb, overflow, peaks, dbg);
}

}  // namespace hr
