// hr_kernels_generic.hip -- format-agnostic f64 kernels of the meter (any sample format, 1..8
// channels, any supported rate).  They define the arithmetic of the GPU path; the fused f32
// kernels in hr_kernels_fused.hip are checked against them and against the oracle.
//
//   k_hop_zero_state   K1+K2a: K-weighting of every hop from zero state, in parallel, recording the
//                      energy, the four correction dot products and the end state (HopRec)
//   k_hop_combine      carry hand-off: true hop-start state from the previous hops' end states,
//                      true hop energy  H = e0 + 2 s.q + s'Gs
//   k_block_hist       K2b pass 1: 400 ms block powers, absolute gate, loudness histogram
//   k_gate_final       K2b pass 2: relative gate and gated mean from the histogram
//   k_true_peak_*      K3: polyphase interpolation + max |.| per channel, f64, oracle tap order
//
// Replaces (FFmpeg, reached from src/analyzer.rs:253-267): f_ebur128.c FILTER()/i400/gate_update,
// swr_convert->192 kHz + fabs max; see SURVEY.md Appendix A.
#include <hip/hip_runtime.h>

#include "hr_device.hpp"
#include "hr_kernels.hpp"

namespace hr {

// ---------------------------------------------------------------------------------------------
// sample access
// ---------------------------------------------------------------------------------------------
template <int FMT>
__device__ __forceinline__ double ld_sample(const void *p, uint64_t i)
{
    if constexpr (FMT == 0) return static_cast<double>(static_cast<const float *>(p)[i]);
    else if constexpr (FMT == 1) return static_cast<double>(static_cast<const int16_t *>(p)[i]) * (1.0 / 32768.0);
    else if constexpr (FMT == 2) {
        const uint8_t *b = static_cast<const uint8_t *>(p) + 3 * i;
        const int32_t v = static_cast<int32_t>(static_cast<uint32_t>(b[0]) << 8 | static_cast<uint32_t>(b[1]) << 16 |
                                               static_cast<uint32_t>(b[2]) << 24);
        return static_cast<double>(v) * (1.0 / 2147483648.0);
    } else if constexpr (FMT == 3) return static_cast<double>(static_cast<const int32_t *>(p)[i]) * (1.0 / 2147483648.0);
    else return static_cast<const double *>(p)[i];
}

// frame i of channel c with the shard halos; reflect = libswresample's start reflection x[-k] = x[k]
template <int FMT>
__device__ __forceinline__ double fetch(const TrackDesc &t, int c, int64_t i, bool reflect)
{
    const int64_t N = static_cast<int64_t>(t.frames);
    if (i < 0) {
        if (t.left_tail) return (i >= -kHalo) ? t.left_tail[(kHalo + i) * t.ch + c] : 0.0;
        if (!reflect) return 0.0;
        i = -i;
    }
    if (i >= N) {
        if (t.right_head && i - N < kHalo) return t.right_head[(i - N) * t.ch + c];
        return 0.0;
    }
    return ld_sample<FMT>(t.pcm, static_cast<uint64_t>(i) * t.ch + c);
}

// ---------------------------------------------------------------------------------------------
// K1 + K2a: every hop from zero state
// ---------------------------------------------------------------------------------------------
template <int FMT>
__global__ void __launch_bounds__(256) k_hop_zero_state(const TrackDesc *__restrict__ tracks,
                                                        const RateDev *__restrict__ rates,
                                                        const WorkItem *__restrict__ items,
                                                        HopRec *__restrict__ recs)
{
    const WorkItem it = items[blockIdx.x];
    const TrackDesc &t = tracks[it.track];
    const uint64_t h = it.first + threadIdx.x;
    if (h >= t.n_hops) return;
    const RateDev &r = rates[t.rate_idx];
    const KCoef k = r.k;
    const int C = t.ch;
    const uint32_t L = t.hop;
    const int64_t n0 = static_cast<int64_t>(h) * L;
    const double *__restrict__ g = r.g;

    for (int c = 0; c < C; c++) {
        // input history is data, not state: take it from the samples in front of the hop
        double x1 = (n0 >= 1 || t.left_tail) ? fetch<FMT>(t, c, n0 - 1, false) : 0.0;
        double x2 = (n0 >= 2 || t.left_tail) ? fetch<FMT>(t, c, n0 - 2, false) : 0.0;
        double y1 = 0, y2 = 0, z1 = 0, z2 = 0, e0 = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0;
        const uint64_t base = static_cast<uint64_t>(n0) * C + c;
        for (uint32_t i = 0; i < L; i++) {
            const double x0 = ld_sample<FMT>(t.pcm, base + static_cast<uint64_t>(i) * C);
            double tt = k.pb0 * x0;
            tt = fma(k.pb1, x1, tt);
            tt = fma(k.pb2, x2, tt);
            tt = fma(-k.pa2, y2, tt);
            const double y0 = fma(-k.pa1, y1, tt);
            double v = fma(-2.0, y1, y0);
            v = v + y2;
            v = fma(-k.ra2, z2, v);
            const double z0 = fma(-k.ra1, z1, v);
            e0 = fma(z0, z0, e0);
            q0 = fma(z0, g[i * 4 + 0], q0);
            q1 = fma(z0, g[i * 4 + 1], q1);
            q2 = fma(z0, g[i * 4 + 2], q2);
            q3 = fma(z0, g[i * 4 + 3], q3);
            x2 = x1; x1 = x0;
            y2 = y1; y1 = y0;
            z2 = z1; z1 = z0;
        }
        HopRec &o = recs[t.rec_base + (kPreRec + h) * C + c];
        o.e0 = e0;
        o.q[0] = q0; o.q[1] = q1; o.q[2] = q2; o.q[3] = q3;
        o.s[0] = y1; o.s[1] = y1 - y2; o.s[2] = z1; o.s[3] = z1 - z2;
    }
}

// The same for a short list of single (track, hop, channel) triples, one lane each: the last hop of every track
// that otherwise runs on the fused kernel.  (A workgroup per track would park 200 almost idle workgroups
// beside the fused kernel for milliseconds; this takes a handful of waves.)
template <int FMT>
__global__ void __launch_bounds__(64) k_hop_zero_state_list(const TrackDesc *__restrict__ tracks,
                                                            const RateDev *__restrict__ rates,
                                                            const WorkItem *__restrict__ items, uint32_t n_items,
                                                            HopRec *__restrict__ recs)
{
    const uint32_t gid = blockIdx.x * 64 + threadIdx.x;
    if (gid >= n_items) return;
    const WorkItem it = items[gid];
    const TrackDesc &t = tracks[it.track];
    const uint64_t h = it.first;
    const int c = static_cast<int>(it.pad);
    const RateDev &r = rates[t.rate_idx];
    const KCoef k = r.k;
    const int C = t.ch;
    const uint32_t L = t.hop;
    const int64_t n0 = static_cast<int64_t>(h) * L;
    const double *__restrict__ g = r.g;
    double x1 = (n0 >= 1 || t.left_tail) ? fetch<FMT>(t, c, n0 - 1, false) : 0.0;
    double x2 = (n0 >= 2 || t.left_tail) ? fetch<FMT>(t, c, n0 - 2, false) : 0.0;
    double y1 = 0, y2 = 0, z1 = 0, z2 = 0, e0 = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    const uint64_t base = static_cast<uint64_t>(n0) * C + c;
#pragma unroll 8
    for (uint32_t i = 0; i < L; i++) {
        const double x0 = ld_sample<FMT>(t.pcm, base + static_cast<uint64_t>(i) * C);
        double tt = k.pb0 * x0;
        tt = fma(k.pb1, x1, tt);
        tt = fma(k.pb2, x2, tt);
        tt = fma(-k.pa2, y2, tt);
        const double y0 = fma(-k.pa1, y1, tt);
        double v = fma(-2.0, y1, y0);
        v = v + y2;
        v = fma(-k.ra2, z2, v);
        const double z0 = fma(-k.ra1, z1, v);
        e0 = fma(z0, z0, e0);
        q0 = fma(z0, g[i * 4 + 0], q0);
        q1 = fma(z0, g[i * 4 + 1], q1);
        q2 = fma(z0, g[i * 4 + 2], q2);
        q3 = fma(z0, g[i * 4 + 3], q3);
        x2 = x1; x1 = x0;
        y2 = y1; y1 = y0;
        z2 = z1; z1 = z0;
    }
    HopRec &o = recs[t.rec_base + (kPreRec + h) * C + c];
    o.e0 = e0;
    o.q[0] = q0; o.q[1] = q1; o.q[2] = q2; o.q[3] = q3;
    o.s[0] = y1; o.s[1] = y1 - y2; o.s[2] = z1; o.s[3] = z1 - z2;
}

// ---------------------------------------------------------------------------------------------
// carry hand-off + true hop energies.  One thread per (hop, channel), hops -kPreBlk .. n_hops-1.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_hop_combine(const TrackDesc *__restrict__ tracks,
                                                     const RateDev *__restrict__ rates,
                                                     const WorkItem *__restrict__ items,
                                                     const HopRec *__restrict__ recs, double *__restrict__ hop_sums)
{
    const WorkItem it = items[blockIdx.x];
    const TrackDesc &t = tracks[it.track];
    const int C = t.ch;
    // it.first counts (hop, channel) pairs from the first hop to combine: hop 0, or hop -kPreBlk for a
    // shard with a left neighbour (their sums complete the boundary blocks)
    const int pre = t.blk_first < 0 ? kPreBlk : 0;
    const uint64_t pair = it.first + threadIdx.x;
    const uint64_t total = (t.n_hops + pre) * C;
    if (pair >= total) return;
    const int64_t h = static_cast<int64_t>(pair / C) - pre;
    const int c = static_cast<int>(pair % C);
    const RateDev &r = rates[t.rate_idx];

    const HopRec *base = recs + t.rec_base + kPreRec * C + c;  // hop 0, channel c; stride C
    if (h >= static_cast<int64_t>(t.sub_from)) return;  // (a hop of several rows: k_hop_combine_rows)
    const HopRec &me = base[h * C];
    // records in front of hop 0 exist only for a shard with a left neighbour (blk_first < 0)
    const bool edge = t.blk_first < 0;
    double p1[4], p2[4], p3[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        p1[i] = (h - 1 >= 0 || edge) ? base[(h - 1) * C].s[i] : 0.0;
        p2[i] = (h - 2 >= 0 || edge) ? base[(h - 2) * C].s[i] : 0.0;
        p3[i] = (h - 3 >= 0 || edge) ? base[(h - 3) * C].s[i] : 0.0;
    }
    double s[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        double a = p1[i];
#pragma unroll
        for (int j = 0; j < 4; j++) a = fma(r.M1[i * 4 + j], p2[j], a);
#pragma unroll
        for (int j = 0; j < 4; j++) a = fma(r.M2[i * 4 + j], p3[j], a);
        s[i] = a;
    }
    double lin = 0, quad = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        lin = fma(s[i], me.q[i], lin);
        double gs = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) gs = fma(r.G[i * 4 + j], s[j], gs);
        quad = fma(s[i], gs, quad);
    }
    double H = me.e0 + (2.0 * lin + quad);
    if (H < 0) H = 0;  // cancellation on (near-)DC input can leave a tiny negative
    hop_sums[t.hop_base + (kPreHop + h) * C + c] = H;
}

// The hops that are several rows of the fused kernel (TrackDesc::sub_from .. n_hops - 1, `parts` rows each; whole tracks
// only: nothing in front of row 0).  Every row was filtered from zero state.  State at the hop's first row: when the
// sub_terms rows behind it are all short rows, their end states by Horner (the truncation is negligible: RateTables::
// sub_residual); otherwise the state at hop sub_from -- from the whole-hop records in front of it, as k_hop_combine
// forms it -- carried through the short rows in between by the exact recurrence s' = Ms s + zend.  The same
// recurrence steps through the hop's own rows; the hop's energy is the sum of theirs.
__global__ void __launch_bounds__(256) k_hop_combine_rows(const TrackDesc *__restrict__ tracks,
                                                          const RateDev *__restrict__ rates,
                                                          const WorkItem *__restrict__ items,
                                                          const HopRec *__restrict__ recs, double *__restrict__ hop_sums)
{
    const WorkItem it = items[blockIdx.x];
    const TrackDesc &t = tracks[it.track];
    if (t.parts <= 1) return;
    const int C = t.ch;
    const uint64_t pair = it.first + threadIdx.x;
    if (pair >= t.n_hops * C) return;
    const int64_t h = static_cast<int64_t>(pair / C);
    if (h < static_cast<int64_t>(t.sub_from)) return;
    const int c = static_cast<int>(pair % C);
    const RateDev &r = rates[t.rate_idx];
    const HopRec *base = recs + t.rec_base + kPreRec * C + c;  // row 0, channel c; stride C
    const int P = static_cast<int>(t.parts), K = r.sub_terms;
    const int64_t H1 = static_cast<int64_t>(t.sub_from);
    const int64_t r0 = H1 + (h - H1) * P;  // the hop's first row
    auto step = [&](double (&s)[4], const double *z) {  // s = Ms s + z
        double n[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            double a = z[i];
#pragma unroll
            for (int j = 0; j < 4; j++) a = fma(r.Ms[i * 4 + j], s[j], a);
            n[i] = a;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) s[i] = n[i];
    };
    double s[4] = {0, 0, 0, 0};
    int64_t from = r0 - K;  // first row whose end state enters
    if (from < H1) {
        from = H1;
        if (H1 > 0) {  // the state at hop H1 from the whole-hop records H1-1, H1-2, H1-3 (k_hop_combine's series)
            double p1[4], p2[4], p3[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                p1[i] = base[(H1 - 1) * C].s[i];
                p2[i] = H1 - 2 >= 0 ? base[(H1 - 2) * C].s[i] : 0.0;
                p3[i] = H1 - 3 >= 0 ? base[(H1 - 3) * C].s[i] : 0.0;
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                double a = p1[i];
#pragma unroll
                for (int j = 0; j < 4; j++) a = fma(r.M1[i * 4 + j], p2[j], a);
#pragma unroll
                for (int j = 0; j < 4; j++) a = fma(r.M2[i * 4 + j], p3[j], a);
                s[i] = a;
            }
        }
    }
    for (int64_t rr = from; rr < r0; rr++) step(s, base[rr * C].s);
    double H = 0;
    for (int p = 0; p < P; p++) {
        const HopRec &row = base[(r0 + p) * C];
        double lin = 0, quad = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            lin = fma(s[i], row.q[i], lin);
            double gs = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) gs = fma(r.Gs[i * 4 + j], s[j], gs);
            quad = fma(s[i], gs, quad);
        }
        H += row.e0 + (2.0 * lin + quad);
        step(s, row.s);
    }
    if (H < 0) H = 0;
    hop_sums[t.hop_base + (kPreHop + h) * C + c] = H;
}

// ---------------------------------------------------------------------------------------------
// K2b pass 1: block powers, absolute gate, histogram.  One workgroup per track.
// ---------------------------------------------------------------------------------------------
// Sum over the workgroup's 256 threads in the order of the classic shared-memory tree (sh[tid] += sh[tid + s] for s = 128, 64,
// ... 1), so that results do not depend on who runs it: the two widest stages from LDS, the other six inside wave 0 with
// cross-lane moves (lane l adds lane l + s: the same pairs), three barriers instead of ten.  Every thread gets the sum.
__device__ __forceinline__ double block_reduce_sum(double v, double *sh)
{
    const int tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
    if (tid < 64) {
        double a = (sh[tid] + sh[tid + 128]) + (sh[tid + 64] + sh[tid + 192]);
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) a += __shfl_down(a, s, 64);  // (lanes >= s add garbage to values nobody reads)
        if (tid == 0) sh[256] = a;
    }
    __syncthreads();
    const double r = sh[256];
    __syncthreads();
    return r;
}

__device__ __forceinline__ unsigned long long block_reduce_sum_u64(unsigned long long v, unsigned long long *sh)
{
    const int tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
    if (tid < 64) {
        unsigned long long a = (sh[tid] + sh[tid + 128]) + (sh[tid + 64] + sh[tid + 192]);
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) a += __shfl_down(a, s, 64);
        if (tid == 0) sh[256] = a;
    }
    __syncthreads();
    const unsigned long long r = sh[256];
    __syncthreads();
    return r;
}

// largest bin b with edges[b] <= p in table `table` of ebu_edges (hr_tables.cpp ebur128_bin_edges: 0 = the bins themselves --
// the caller has checked p >= edges[0] --, 1 / 2 = the first bin of the -10 LU / -20 LU relative gate).  The edges decide; the
// device logarithm only says where to look (a bisection is thirteen dependent loads, this is two or three).
__device__ __forceinline__ int ebu_bin_from_edges(const double *__restrict__ ebu_edges, int table, double p)
{
    const double *__restrict__ edges = ebu_edges + table * kEbuBins;
    const double e = ((10.0 * log10(p) - 0.691) + 70.0 - 10.0 * table) * 100.0;
    int b = !(e >= 0.0) ? 0 : e > static_cast<double>(kEbuBins - 1) ? kEbuBins - 1 : static_cast<int>(e);
    while (b > 0 && p < edges[b]) b--;
    while (b + 1 < kEbuBins && p >= edges[b + 1]) b++;
    return b;
}

// One count for bin `bin` (a negative bin: none) into a workgroup's histogram.  Lanes of a wave that hit the same bin are
// combined first -- a long steady stream puts every block into the same two or three bins -- in up to four rounds (the bin of
// the first pending lane each time); lanes still pending after that add their own.  Integer counts: the result does not
// depend on the order.
__device__ __forceinline__ void hist_add_wave(uint32_t *hg, int bin)
{
    bool pending = bin >= 0;
    for (int round = 0; round < 4; round++) {
        const unsigned long long live = __ballot(pending);
        if (live == 0ull) return;
        const int leader = __ffsll(static_cast<long long>(live)) - 1;
        const int b = __shfl(bin, leader, 64);
        const bool mine = pending && bin == b;
        const unsigned long long same = __ballot(mine);
        if (static_cast<int>(threadIdx.x & 63) == leader) atomicAdd(&hg[b], static_cast<uint32_t>(__popcll(same)));
        pending = pending && !mine;
    }
    if (pending) atomicAdd(&hg[bin], 1u);
}

// The gating passes below are written as device functions over ONE workgroup's LDS histogram, so that the chunked kernels (long
// tracks and streams: one workgroup per chunk of kGateChunk blocks, histograms joined in global memory) and k_gate_small (short
// tracks: one workgroup does all of a track's gating, nothing in global memory but the block powers) run the same arithmetic in
// the same order.  PAD: the histogram is stored with one padding slot per 32 bins (k_range_final's layout: its threads walk 32
// consecutive bins each, starting in different banks).
template <bool PAD>
__device__ __forceinline__ int hslot(int i) { return PAD ? i + (i >> 5) : i; }
constexpr int kHistLds = kEbuBins + kEbuBins / 32 + 1;  // slots of a padded histogram

struct BlockSums { double sum_kept; unsigned long long n_kept; };

// blocks [first, jj_end) of track t: block powers (written to block_powers), absolute gate, counts into hh (zeroed by the
// caller, who also synchronises afterwards); every thread returns the chunk's kept sum and count
template <bool PAD>
__device__ __forceinline__ BlockSums block_hist_chunk(const TrackDesc &t, uint64_t first, uint64_t jj_end, int gate_mode,
                                                      const double *__restrict__ hop_sums, const double *__restrict__ ln_bounds,
                                                      const double *__restrict__ ebu_edges, double *__restrict__ block_powers,
                                                      uint32_t *hh, double *shd, unsigned long long *shu)
{
    const int C = t.ch;
    const double bins = gate_mode == 2 ? static_cast<double>(4 * ((static_cast<uint64_t>(t.fs) + 5) / 10))
                                       : static_cast<double>(static_cast<uint64_t>(t.fs) * 4 / 10);
    const double abs_exact = gate_mode == 1 ? exp10((-70.0 + 0.691) / 10.0) : 0.0;  // only for HR_GATE_EXACT
    double sum_kept = 0.0;
    unsigned long long n_kept = 0;
    const double *H = hop_sums + t.hop_base + kPreHop * C;
    // (every lane of the workgroup runs every trip of the loop -- lanes past the chunk's end with nothing to add -- so that
    // the wave-wide combination of the histogram counts sees whole waves)
    for (uint64_t j0 = first; j0 < jj_end; j0 += blockDim.x) {
        const uint64_t jj = j0 + threadIdx.x;
        int bin = -1;
        if (jj < jj_end) {
            const int64_t j = t.blk_first + static_cast<int64_t>(jj);
            double power = gate_mode == 2 ? 0.0 : 1e-12;
            for (int c = 0; c < C; c++) {
                if (t.w[c] == 0.0f) continue;
                const double s = ((H[j * C + c] + H[(j + 1) * C + c]) + H[(j + 2) * C + c]) + H[(j + 3) * C + c];
                power += static_cast<double>(t.w[c]) * s;
            }
            power /= bins;
            block_powers[t.blk_base + jj] = power;
            if (gate_mode == 0) {
                // loud = -0.691 + 10 log10(power) >= -70 and its bin, decided on the power itself (ebur128_bin_edges)
                if (power >= ebu_edges[0]) {
                    bin = hslot<PAD>(ebu_bin_from_edges(ebu_edges, 0, power));
                    sum_kept += power;
                    n_kept++;
                }
            } else if (gate_mode == 2) {
                if (power >= ln_bounds[0]) {
                    int lo = 0, hi = kLnBins;
                    do {
                        const int mid = (lo + hi) / 2;
                        if (power >= ln_bounds[mid]) lo = mid; else hi = mid;
                    } while (hi - lo != 1);
                    bin = hslot<PAD>(lo);
                    sum_kept += power;
                    n_kept++;
                }
            } else {
                if (power > abs_exact) { sum_kept += power; n_kept++; }
            }
        }
        if (gate_mode != 1) hist_add_wave(hh, bin);
    }
    BlockSums r;
    r.sum_kept = block_reduce_sum(sum_kept, shd);
    r.n_kept = block_reduce_sum_u64(n_kept, shu);
    return r;
}

__global__ void __launch_bounds__(256) k_block_hist(const TrackDesc *__restrict__ tracks, const WorkItem *__restrict__ items,
                                                    int gate_mode, const double *__restrict__ hop_sums,
                                                    const double *__restrict__ ln_bounds,
                                                    const double *__restrict__ ebu_edges,
                                                    double *__restrict__ block_powers,
                                                    uint32_t *__restrict__ hist, GatePart *__restrict__ parts)
{
    // The chunk's histogram is built in LDS and joins the track's afterwards, one atomic per bin the chunk touched: every
    // increment of a bin in global memory is a read-modify-write at the L2, and the 176 chunks of a steady 10-hour stream
    // all want the same two or three bins (round 3: 1.15 ms of a 5.5 ms call, profiles/r3_config5_kernel_stats.md).
    __shared__ uint32_t hh[kEbuBins];
    __shared__ double shd[257];
    __shared__ unsigned long long shu[257];
    const WorkItem it = items[blockIdx.x];  // track, chunk index (pad), first block of the chunk
    const TrackDesc &t = tracks[it.track];
    uint32_t *hg = hist + static_cast<size_t>(it.track) * kEbuBins;  // zeroed by the host before the launch
    const int nbins = gate_mode == 2 ? kLnBins : gate_mode == 0 ? kEbuBins : 0;
    for (int i = threadIdx.x; i < nbins; i += blockDim.x) hh[i] = 0;
    __syncthreads();
    const uint64_t jj_end = it.first + kGateChunk < t.n_blocks ? it.first + kGateChunk : t.n_blocks;
    const BlockSums bs = block_hist_chunk<false>(t, it.first, jj_end, gate_mode, hop_sums, ln_bounds, ebu_edges, block_powers, hh, shd, shu);
    // (the reductions inside end with a barrier: every count is in hh)
    for (int i = threadIdx.x; i < nbins; i += blockDim.x)
        if (hh[i]) atomicAdd(&hg[i], hh[i]);
    if (threadIdx.x == 0) {
        GatePart &p = parts[t.part_base + it.pad];
        p.sum_kept = bs.sum_kept;
        p.n_kept = bs.n_kept;
    }
}

// ---------------------------------------------------------------------------------------------
// K2b pass 2: relative gate + gated mean.  One workgroup per track; `outs` carries sum_kept/n_kept
// (of this track, or all-reduced over shards) and `hist` the (reduced) histogram.
// ---------------------------------------------------------------------------------------------
// the relative gate and the gated mean of one track from its kept sum / count and its complete histogram hg (global, or the
// workgroup's LDS); thread 0 writes the results
template <bool PAD>
__device__ __forceinline__ void gate_final_track(const TrackDesc &t, int gate_mode, const double *__restrict__ block_powers,
                                                 const uint32_t *hg, const double *__restrict__ ebu_energy,
                                                 const double *__restrict__ ebu_edges, const double *__restrict__ ln_energy,
                                                 const double *__restrict__ ln_bounds, double sum_kept, unsigned long long n_kept,
                                                 TrackOut &o, double *shd, unsigned long long *shu)
{
    if (n_kept == 0) return;  // uniform across the block
    if (gate_mode == 0) {
        double rt = sum_kept / static_cast<double>(n_kept);
        if (!rt) rt = 1e-12;
        // first bin of the relative gate, (int)(((-0.691 + 10 log10(rt)) - 10 + 70) * 100) clipped: table 1 of the edges
        const int gp = ebu_bin_from_edges(ebu_edges, 1, rt);
        double isum = 0;
        unsigned long long nint = 0;
        for (int i = gp + threadIdx.x; i < kEbuBins; i += blockDim.x) {
            // (an empty bin adds + 0.0: skipping it leaves the sum's bits alone and saves the load of its energy -- all but a few
            // dozen bins of a track are empty, and the loads were what a workgroup of k_gate_small spent its time waiting for)
            const unsigned cnt = hg[hslot<PAD>(i)];
            if (cnt == 0) continue;
            nint += cnt;
            isum += static_cast<double>(cnt) * ebu_energy[i];
        }
        const double is = block_reduce_sum(isum, shd);
        const unsigned long long ni = block_reduce_sum_u64(nint, shu);
        if (threadIdx.x == 0) {
            o.rel_energy = rt;
            o.n_gated_final = ni;
            o.mean_energy = ni ? is / static_cast<double>(ni) : 0.0;
        }
    } else if (gate_mode == 2) {
        double gs = 0;
        unsigned long long above = 0;
        for (int i = threadIdx.x; i < kLnBins; i += blockDim.x) {
            const unsigned cnt = hg[hslot<PAD>(i)];
            if (cnt == 0) continue;
            above += cnt;
            gs += static_cast<double>(cnt) * ln_energy[i];
        }
        const double g1 = block_reduce_sum(gs, shd);
        const unsigned long long a1 = block_reduce_sum_u64(above, shu);
        if (a1 == 0) return;
        const double mean1 = g1 / static_cast<double>(a1);
        const double rel = mean1 * 0.1;
        int start;
        if (rel < ln_bounds[0]) start = 0;
        else {
            int lo = 0, hi = kLnBins;
            do {
                const int mid = (lo + hi) / 2;
                if (rel >= ln_bounds[mid]) lo = mid; else hi = mid;
            } while (hi - lo != 1);
            start = lo;
            if (rel > ln_energy[start]) ++start;
        }
        gs = 0;
        above = 0;
        for (int i = start + threadIdx.x; i < kLnBins; i += blockDim.x) {
            const unsigned cnt = hg[hslot<PAD>(i)];
            if (cnt == 0) continue;
            above += cnt;
            gs += static_cast<double>(cnt) * ln_energy[i];
        }
        const double g2 = block_reduce_sum(gs, shd);
        const unsigned long long a2 = block_reduce_sum_u64(above, shu);
        if (threadIdx.x == 0) {
            o.rel_energy = mean1;
            o.n_gated_final = a2;
            o.mean_energy = a2 ? g2 / static_cast<double>(a2) : 0.0;
        }
    } else {
        const double abs_exact = exp10((-70.0 + 0.691) / 10.0);
        const double mean1 = sum_kept / static_cast<double>(n_kept);
        const double rel = 0.1 * mean1;
        double s2 = 0;
        unsigned long long n2 = 0;
        for (uint64_t jj = threadIdx.x; jj < t.n_blocks; jj += blockDim.x) {
            const double p = block_powers[t.blk_base + jj];
            if (p > abs_exact && p > rel) { s2 += p; n2++; }
        }
        const double ss = block_reduce_sum(s2, shd);
        const unsigned long long nn = block_reduce_sum_u64(n2, shu);
        if (threadIdx.x == 0) {
            o.rel_energy = mean1;
            o.n_gated_final = nn;
            o.sum_gated = ss;
            o.mean_energy = nn ? ss / static_cast<double>(nn) : 0.0;
        }
    }
}

// list: the tracks this launch is for (their indices in WorkItem::track), or nullptr for tracks 0 .. gridDim.x - 1
__global__ void __launch_bounds__(256) k_gate_final(const TrackDesc *__restrict__ tracks, const WorkItem *__restrict__ list, int gate_mode,
                                                    const double *__restrict__ block_powers,
                                                    const uint32_t *__restrict__ hist,
                                                    const double *__restrict__ ebu_energy,
                                                    const double *__restrict__ ebu_edges,
                                                    const double *__restrict__ ln_energy,
                                                    const double *__restrict__ ln_bounds,
                                                    const GatePart *__restrict__ parts, int reduce_only,
                                                    TrackOut *__restrict__ outs)
{
    __shared__ double shd[257];
    __shared__ unsigned long long shu[257];
    const uint32_t ti = list ? list[blockIdx.x].track : blockIdx.x;
    const TrackDesc &t = tracks[ti];
    TrackOut &o = outs[ti];
    const uint32_t *hg = hist + static_cast<size_t>(ti) * kEbuBins;
    // the chunks' partial sums, in chunk order (n_parts == 0: `outs` already carries sums reduced over shards)
    if (t.n_parts) {
        if (threadIdx.x == 0) {
            double sk = 0.0;
            unsigned long long nk = 0;
            for (uint32_t c = 0; c < t.n_parts; c++) { sk += parts[t.part_base + c].sum_kept; nk += parts[t.part_base + c].n_kept; }
            o.sum_kept = sk;
            o.n_kept = nk;
            o.n_blocks = t.n_blocks;
            o.mean_energy = 0;
            o.rel_energy = 0;
            o.n_gated_final = 0;
        }
        __syncthreads();
    }
    if (reduce_only) return;
    gate_final_track<false>(t, gate_mode, block_powers, hg, ebu_energy, ebu_edges, ln_energy, ln_bounds, o.sum_kept, o.n_kept, o, shd, shu);
}

// ---------------------------------------------------------------------------------------------
// Loudness range + maximum momentary / short-term power.  One workgroup per (whole) track; the short-term
// histogram lives in LDS.  gate modes 0, 1: f_ebur128.c (i3000 histogram, gate_update with -20 LU, its 10 % / 95 %
// counting rule); mode 2: libavfilter/ebur128.c ff_ebur128_loudness_range (3 s blocks every second, 1000 bins).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_reduce_max(double v, double *sh)
{
    const int tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
    if (tid < 64) {
        double a = fmax(fmax(sh[tid], sh[tid + 128]), fmax(sh[tid + 64], sh[tid + 192]));
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) a = fmax(a, __shfl_down(a, s, 64));
        if (tid == 0) sh[256] = a;
    }
    __syncthreads();
    const double r = sh[256];
    __syncthreads();
    return r;
}

struct RangeSums { double sum_kept, max_s, max_m; unsigned long long n_kept; };

// short-term windows and blocks [first, first + kGateChunk) of track t: counts into hh (zeroed by the caller), every thread
// returns the chunk's sums and maxima
template <bool PAD>
__device__ __forceinline__ RangeSums range_stats_chunk(const TrackDesc &t, uint64_t first, int gate_mode,
                                                       const double *__restrict__ hop_sums, const double *__restrict__ block_powers,
                                                       const double *__restrict__ ln_bounds, const double *__restrict__ ebu_edges,
                                                       uint32_t *hh, double *shd, unsigned long long *shu)
{
    const int C = t.ch;
    const bool ln = gate_mode == 2;
    // windows are counted by their LAST hop: a shard owns those that end in it, i.e. start at -st_pre or later
    const uint64_t n_s = t.n_hops + t.st_pre >= 30 ? t.n_hops + t.st_pre - 29 : 0;
    const double bins = ln ? static_cast<double>(30 * ((static_cast<uint64_t>(t.fs) + 5) / 10))
                           : static_cast<double>(static_cast<uint64_t>(t.fs) * 30 / 10);
    const double *H = hop_sums + t.hop_base + kPreHop * C;
    double sum_kept = 0.0, max_s = 0.0, max_m = 0.0;
    unsigned long long n_kept = 0;
    const uint64_t j_end = first + kGateChunk < n_s ? first + kGateChunk : n_s;
    for (uint64_t jw = first + threadIdx.x; jw < j_end; jw += blockDim.x) {
        const int64_t j = static_cast<int64_t>(jw) - t.st_pre;  // first hop of the window, relative to this shard
        double power = ln ? 0.0 : 1e-12;
        for (int c = 0; c < C; c++) {
            if (t.w[c] == 0.0f) continue;
            double s = H[j * C + c];
            for (int k = 1; k < 30; k++) s += H[(j + k) * C + c];
            power += static_cast<double>(t.w[c]) * s;
        }
        power /= bins;
        max_s = power > max_s ? power : max_s;
        if (!ln) {
            if (power >= ebu_edges[0]) {  // (as in k_block_hist)
                atomicAdd(&hh[hslot<PAD>(ebu_bin_from_edges(ebu_edges, 0, power))], 1u);
                sum_kept += power;
                n_kept++;
            }
        } else if ((static_cast<int64_t>(t.first_hop) + j) % 10 == 0 && power >= ln_bounds[0]) {
            int lo = 0, hi = kLnBins;
            do {
                const int mid = (lo + hi) / 2;
                if (power >= ln_bounds[mid]) lo = mid; else hi = mid;
            } while (hi - lo != 1);
            atomicAdd(&hh[hslot<PAD>(lo)], 1u);
        }
    }
    const uint64_t b_end = first + kGateChunk < t.n_blocks ? first + kGateChunk : t.n_blocks;
    for (uint64_t jj = first + threadIdx.x; jj < b_end; jj += blockDim.x) {
        const double p = block_powers[t.blk_base + jj];
        max_m = p > max_m ? p : max_m;
    }
    RangeSums r;
    r.max_s = block_reduce_max(max_s, shd);
    r.max_m = block_reduce_max(max_m, shd);
    r.sum_kept = block_reduce_sum(sum_kept, shd);
    r.n_kept = block_reduce_sum_u64(n_kept, shu);
    return r;
}

__global__ void __launch_bounds__(256) k_range_stats(const TrackDesc *__restrict__ tracks, const WorkItem *__restrict__ items,
                                                     int gate_mode, const double *__restrict__ hop_sums,
                                                     const double *__restrict__ block_powers,
                                                     const double *__restrict__ ln_bounds,
                                                     const double *__restrict__ ebu_edges, uint32_t *__restrict__ st_hist,
                                                     GatePart *__restrict__ parts)
{
    __shared__ uint32_t hh[kEbuBins];
    __shared__ double shd[257];
    __shared__ unsigned long long shu[257];
    const WorkItem it = items[blockIdx.x];  // track, chunk index (pad), first window / block of the chunk
    const TrackDesc &t = tracks[it.track];
    const int nbins = gate_mode == 2 ? kLnBins : kEbuBins;
    for (int i = threadIdx.x; i < nbins; i += blockDim.x) hh[i] = 0;
    __syncthreads();
    const RangeSums rs = range_stats_chunk<false>(t, it.first, gate_mode, hop_sums, block_powers, ln_bounds, ebu_edges, hh, shd, shu);
    // the chunk's histogram joins the track's (zeroed by the host before the launch)
    uint32_t *hg = st_hist + static_cast<size_t>(it.track) * kEbuBins;
    for (int i = threadIdx.x; i < nbins; i += blockDim.x)
        if (hh[i]) atomicAdd(&hg[i], hh[i]);
    if (threadIdx.x == 0) {
        GatePart &p = parts[t.part_base + it.pad];
        p.st_sum_kept = rs.sum_kept;
        p.st_n_kept = rs.n_kept;
        p.max_s_power = rs.max_s;
        p.max_m_power = rs.max_m;
    }
}

// the track's loudness range from its complete short-term histogram lh (LDS, padded: hslot<true>) and the sums / maxima of its
// chunks; thread 0 writes the results.  lo_bin / hi_bin: two ints in LDS.
__device__ __forceinline__ void range_final_track(const TrackDesc &t, int gate_mode, const uint32_t *lh, double sum_kept,
                                                  unsigned long long n_kept, double max_s, double max_m,
                                                  const double *__restrict__ ln_bounds, const double *__restrict__ ln_energy,
                                                  const double *__restrict__ ebu_edges, TrackOut &o, double *shd,
                                                  unsigned long long *shu, int *lo_bin_p, int *hi_bin_p)
{
    int &lo_bin = *lo_bin_p, &hi_bin = *hi_bin_p;
    const bool ln = gate_mode == 2;
    const int nbins = ln ? kLnBins : kEbuBins;
    auto hh = [&](int i) { return lh[i + (i >> 5)]; };
    if (threadIdx.x == 0) { lo_bin = nbins; hi_bin = ln ? nbins : -1; }
    __syncthreads();
    // first bin of the range statistics and the mean power the relative gate derives from
    int first = 0;
    double rel_energy = 0.0;
    bool any;
    constexpr int kChunk = 32;  // bins per thread: 256 * 32 >= 8001
    const int c0 = threadIdx.x * kChunk, c1 = c0 + kChunk < nbins ? c0 + kChunk : nbins;
    if (!ln) {
        any = n_kept != 0;
        if (any) {
            double rt = sum_kept / static_cast<double>(n_kept);
            if (!rt) rt = 1e-12;
            rel_energy = rt;
            first = ebu_bin_from_edges(ebu_edges, 2, rt);  // the -20 LU gate's first bin (table 2 of the edges)
        }
    } else {
        double gs = 0;
        unsigned long long cnt = 0;
        for (int i = c0; i < c1; i++)
            if (hh(i)) { cnt += hh(i); gs += static_cast<double>(hh(i)) * ln_energy[i]; }
        const double g1 = block_reduce_sum(gs, shd);
        const unsigned long long a1 = block_reduce_sum_u64(cnt, shu);
        any = a1 != 0;
        if (any) {
            rel_energy = g1 / static_cast<double>(a1);
            const double integ = 0.01 * rel_energy;
            if (integ >= ln_bounds[0]) {
                int lo = 0, hi = kLnBins;
                do {
                    const int mid = (lo + hi) / 2;
                    if (integ >= ln_bounds[mid]) lo = mid; else hi = mid;
                } while (hi - lo != 1);
                first = lo;
                if (integ > ln_energy[first]) ++first;
            }
        }
    }
    if (any) {  // uniform across the workgroup
        unsigned long long mine = 0;
        for (int i = c0 > first ? c0 : first; i < c1; i++) mine += hh(i);
        shu[threadIdx.x] = mine;
        __syncthreads();
        unsigned long long E = 0, total = 0;
        for (int k = 0; k < 256; k++) {
            if (k < static_cast<int>(threadIdx.x)) E += shu[k];
            total += shu[k];
        }
        if (total) {
            unsigned long long need_lo, need_hi;
            if (!ln) {
                need_lo = static_cast<unsigned long long>(10 * static_cast<double>(total) * 0.01 + 0.5);
                need_hi = static_cast<unsigned long long>(95 * static_cast<double>(total) * 0.01 + 0.5);
            } else {
                need_lo = static_cast<unsigned long long>(static_cast<double>(total - 1) * 0.1 + 0.5);
                need_hi = static_cast<unsigned long long>(static_cast<double>(total - 1) * 0.95 + 0.5);
            }
            // (this thread's bins first, then one atomic per thread and bound)
            int my_lo = nbins, my_hi = ln ? nbins : -1;
            for (int i = c0 > first ? c0 : first; i < c1; i++) {
                const unsigned long long incl = E + hh(i);
                if (!ln) {
                    if (incl >= need_lo && i < my_lo) my_lo = i;  // first bin whose running count reaches the 10 % count
                    if (E < need_hi && i > my_hi) my_hi = i;      // last bin with fewer than the 95 % count below it
                } else {
                    if (incl > need_lo && i < my_lo) my_lo = i;
                    if (incl > need_hi && i < my_hi) my_hi = i;
                }
                E = incl;
            }
            if (my_lo < nbins) atomicMin(&lo_bin, my_lo);
            if (!ln) { if (my_hi >= 0) atomicMax(&hi_bin, my_hi); }
            else if (my_hi < nbins) atomicMin(&hi_bin, my_hi);
        }
        __syncthreads();
        any = total != 0;
    }
    if (threadIdx.x == 0) {
        o.st_rel_energy = rel_energy;
        o.max_m_power = max_m;
        o.max_s_power = max_s;
        o.n_shortterm = t.n_hops + t.st_pre >= 30 ? t.n_hops + t.st_pre - 29 : 0;
        o.lra_low_bin = any ? lo_bin : -1;
        o.lra_high_bin = any ? hi_bin : -1;
    }
}

// the track's loudness range from its complete short-term histogram; one workgroup per track (list: as for k_gate_final)
__global__ void __launch_bounds__(256) k_range_final(const TrackDesc *__restrict__ tracks, const WorkItem *__restrict__ list, int gate_mode,
                                                     const uint32_t *__restrict__ st_hist, const GatePart *__restrict__ parts,
                                                     const double *__restrict__ ln_bounds,
                                                     const double *__restrict__ ln_energy,
                                                     const double *__restrict__ ebu_edges, TrackOut *__restrict__ outs)
{
    __shared__ double shd[257];
    __shared__ unsigned long long shu[257];
    __shared__ int lo_bin, hi_bin;
    const uint32_t ti = list ? list[blockIdx.x].track : blockIdx.x;
    const TrackDesc &t = tracks[ti];
    const int nbins = gate_mode == 2 ? kLnBins : kEbuBins;
    // the track's histogram, read once, coalesced, into LDS; every thread then walks its own 32 consecutive bins there
    // (one padding slot per 32 bins: the threads' walks start in different banks)
    __shared__ uint32_t lh[kHistLds];
    const uint32_t *hg = st_hist + static_cast<size_t>(ti) * kEbuBins;
    for (int i = threadIdx.x; i < nbins; i += blockDim.x) lh[i + (i >> 5)] = hg[i];
    double sum_kept = 0.0, max_s = 0.0, max_m = 0.0;
    unsigned long long n_kept = 0;
    for (uint32_t c = 0; c < t.n_parts; c++) {  // chunk order, every thread the same
        const GatePart &p = parts[t.part_base + c];
        sum_kept += p.st_sum_kept;
        n_kept += p.st_n_kept;
        max_s = p.max_s_power > max_s ? p.max_s_power : max_s;
        max_m = p.max_m_power > max_m ? p.max_m_power : max_m;
    }
    range_final_track(t, gate_mode, lh, sum_kept, n_kept, max_s, max_m, ln_bounds, ln_energy, ebu_edges, outs[ti], shd, shu, &lo_bin, &hi_bin);
}

// ---------------------------------------------------------------------------------------------
// All of a SHORT track's gating in one workgroup (tracks of at most kGateSmallParts chunks: 27 minutes): block powers and
// gating histogram, relative gate and gated mean, short-term histogram, loudness range -- what k_block_hist, k_gate_final,
// k_range_stats and k_range_final do for a long track, chunk by chunk in the same order with the same reductions (same result
// bits), but with the two histograms in LDS, one after the other.  A launch of 10 000 ten-second clips otherwise zeroes, fills
// and re-reads 640 MB of histograms in global memory for 97 blocks per track (round 3: 0.83 of a 14.8 ms step -> 0.1 ms).
// Mirrors the many-small-files case of the reference (src/cli.rs:403-435: one ffmpeg run per file, however short).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_gate_small(const TrackDesc *__restrict__ tracks, const WorkItem *__restrict__ list, int gate_mode,
                                                    const double *__restrict__ hop_sums, double *__restrict__ block_powers,
                                                    const double *__restrict__ ebu_energy, const double *__restrict__ ebu_edges,
                                                    const double *__restrict__ ln_energy, const double *__restrict__ ln_bounds,
                                                    TrackOut *__restrict__ outs)
{
    __shared__ uint32_t lh[kHistLds];
    __shared__ double shd[257];
    __shared__ unsigned long long shu[257];
    __shared__ int lo_bin, hi_bin;
    const uint32_t ti = list[blockIdx.x].track;
    const TrackDesc &t = tracks[ti];
    TrackOut &o = outs[ti];
    for (int i = threadIdx.x; i < kHistLds; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    // pass 1: the chunks in order, their sums added in chunk order (what k_gate_final's thread 0 does with the GatePart records)
    double sk = 0.0;
    unsigned long long nk = 0;
    for (uint32_t c = 0; c < t.n_parts; c++) {
        const uint64_t first = static_cast<uint64_t>(c) * kGateChunk;
        const uint64_t jj_end = first + kGateChunk < t.n_blocks ? first + kGateChunk : t.n_blocks;
        const BlockSums bs = block_hist_chunk<true>(t, first, jj_end, gate_mode, hop_sums, ln_bounds, ebu_edges, block_powers, lh, shd, shu);
        sk += bs.sum_kept;
        nk += bs.n_kept;
    }
    if (threadIdx.x == 0) {
        o.sum_kept = sk;
        o.n_kept = nk;
        o.n_blocks = t.n_blocks;
        o.mean_energy = 0;
        o.rel_energy = 0;
        o.n_gated_final = 0;
    }
    // (block powers written above are read below by other threads of this workgroup: HR_GATE_EXACT's second pass, the maxima)
    __threadfence_block();
    __syncthreads();
    gate_final_track<true>(t, gate_mode, block_powers, lh, ebu_energy, ebu_edges, ln_energy, ln_bounds, sk, nk, o, shd, shu);
    __syncthreads();
    // pass 2: the short-term histogram takes the gating histogram's place
    for (int i = threadIdx.x; i < kHistLds; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    double st_sum = 0.0, max_s = 0.0, max_m = 0.0;
    unsigned long long st_n = 0;
    for (uint32_t c = 0; c < t.n_parts; c++) {
        const RangeSums rs = range_stats_chunk<true>(t, static_cast<uint64_t>(c) * kGateChunk, gate_mode, hop_sums, block_powers, ln_bounds,
                                                     ebu_edges, lh, shd, shu);
        st_sum += rs.sum_kept;
        st_n += rs.n_kept;
        max_s = rs.max_s > max_s ? rs.max_s : max_s;
        max_m = rs.max_m > max_m ? rs.max_m : max_m;
    }
    range_final_track(t, gate_mode, lh, st_sum, st_n, max_s, max_m, ln_bounds, ln_energy, ebu_edges, o, shd, shu, &lo_bin, &hi_bin);
}

void launch_gate_small(const TrackDesc *tracks, const WorkItem *list, uint32_t n_list, int gate_mode, const double *hop_sums,
                       double *block_powers, const double *ebu_energy, const double *ebu_edges, const double *ln_energy,
                       const double *ln_bounds, TrackOut *outs, hipStream_t st)
{
    if (!n_list) return;
    hipLaunchKernelGGL(k_gate_small, dim3(n_list), dim3(256), 0, st, tracks, list, gate_mode, hop_sums, block_powers, ebu_energy,
                       ebu_edges, ln_energy, ln_bounds, outs);
}

void launch_range_stats(const TrackDesc *tracks, uint32_t n_tracks, const WorkItem *items, uint32_t n_items, int gate_mode,
                        const double *hop_sums, const double *block_powers, const double *ln_bounds, const double *ln_energy,
                        const double *ebu_edges, uint32_t *st_hist, GatePart *parts, bool finalize, TrackOut *outs, hipStream_t st,
                        const WorkItem *list)
{
    if (!n_tracks) return;
    if (n_items)
        hipLaunchKernelGGL(k_range_stats, dim3(n_items), dim3(256), 0, st, tracks, items, gate_mode, hop_sums, block_powers,
                           ln_bounds, ebu_edges, st_hist, parts);
    if (finalize)
        hipLaunchKernelGGL(k_range_final, dim3(n_tracks), dim3(256), 0, st, tracks, list, gate_mode, st_hist, parts, ln_bounds, ln_energy,
                       ebu_edges, outs);
}

// ---------------------------------------------------------------------------------------------
// K3: true peak, generic f64.  One thread per input position; per-channel max via atomicMax on the
// bit pattern (non-negative doubles order like unsigned integers).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void peak_commit(double m, unsigned long long *slot)
{
    // wave max, then one atomic per wave
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_down(m, off, 64);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0 && m > 0.0) atomicMax(slot, static_cast<unsigned long long>(__double_as_longlong(m)));
}

// Resampler bookkeeping (hr_tables.hpp): output n sits at floor(n * D / S) phase units.  First output at or after q
// phase units: ceil(q * S / D), formed without overflowing 64 bits (q < 2^46, S < 2^30, D < 2^30).
__device__ __forceinline__ int64_t swr_first_output(int64_t q, int64_t D, int64_t S)
{
    return (q / D) * S + ((q % D) * S + D - 1) / D;
}

// Each workgroup covers `span` consecutive positions starting at item.first (positions past the valid
// range are skipped), 256 at a time.
template <int FMT>
__global__ void __launch_bounds__(256) k_true_peak_swr(const TrackDesc *__restrict__ tracks,
                                                       const RateDev *__restrict__ rates,
                                                       const WorkItem *__restrict__ items, uint32_t span,
                                                       unsigned long long *__restrict__ peaks)
{
    const WorkItem it = items[blockIdx.x];
    if (it.pad) span = it.pad;  // a short item (the tail of a track that otherwise runs on the fused kernel)
    const TrackDesc &t = tracks[it.track];
    const RateDev &r = rates[t.rate_idx];
    const int64_t N = static_cast<int64_t>(t.frames);
    const int P = r.swr_phases, step = r.swr_step;
    // an output at position pos needs frames pos-15 .. pos+16
    const int64_t last_pos = t.stream_end ? N - 1 - t.tp_drop : N - 1;
    const int C = t.ch;
    for (int c = 0; c < C; c++) {
        double m = 0.0;
        for (uint32_t off = 0; off < span; off += 256) {
            const int64_t pos = static_cast<int64_t>(it.first) + off + threadIdx.x;
            if (!(t.tp_enabled && pos <= last_pos && pos < N)) continue;
            double x[kSwrTaps];
#pragma unroll
            for (int i = 0; i < kSwrTaps; i++) x[i] = fetch<FMT>(t, c, pos - 15 + i, t.tp_reflect != 0);
            // outputs n with floor(n*step/S) in [pos*P, (pos+1)*P)
            const int64_t S = r.swr_S;
            const int64_t n_lo = swr_first_output(pos * P, step, S);
            const int64_t n_hi = swr_first_output((pos + 1) * P, step, S);
            for (int64_t n = n_lo; n < n_hi; n++) {
                const int64_t num = n * step;
                const int ph = static_cast<int>(num / S - pos * P);
                const double *__restrict__ f = r.swr_taps + static_cast<size_t>(ph) * kSwrTaps;
                double val = 0.0;
#pragma unroll
                for (int i = 0; i < kSwrTaps; i++) val = fma(x[i], f[i], val);
                if (S > 1) {  // resample_linear: between this phase and the next, weight frac / src_incr
                    double v2 = 0.0;
#pragma unroll
                    for (int i = 0; i < kSwrTaps; i++) v2 = fma(x[i], f[kSwrTaps + i], v2);
                    const double frac = static_cast<double>(static_cast<uint32_t>(num % S) * r.swr_frac_scale);
                    val += (v2 - val) * r.swr_inv_src_incr * frac;
                }
                const double a = fabs(val);
                m = a > m ? a : m;
            }
        }
        peak_commit(m, &peaks[static_cast<size_t>(it.track) * kMaxCh + c]);
    }
}

// The whole 192 kHz stream instead of its maximum: what the `aresample` in front of af_loudnorm produces (libswresample
// defaults, f64; start reflected x[-k] = x[k], and, unlike f_ebur128's private resampler, flushed at the end with
// the reflection x[N+j] = x[N-1-j]).  One thread per input position writes the outputs n with floor(n*step/P) == pos.
template <int FMT>
__global__ void __launch_bounds__(256) k_resample_swr(const void *__restrict__ src, int64_t N, int C, int P, int step, int S,
                                                      uint32_t frac_scale, double inv_src_incr,
                                                      const double *__restrict__ taps, double *__restrict__ dst, int64_t pos0)
{
    const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
    for (int64_t pos = pos0 + static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; pos < N; pos += stride) {
        const int64_t n_lo = swr_first_output(pos * P, step, S);
        const int64_t n_hi = swr_first_output((pos + 1) * P, step, S);
        for (int c = 0; c < C; c++) {
            double x[kSwrTaps];
#pragma unroll
            for (int i = 0; i < kSwrTaps; i++) {
                int64_t k = pos - 15 + i;
                if (k < 0) k = -k;
                if (k >= N) k = 2 * N - 1 - k;
                x[i] = (k < 0 || k >= N) ? 0.0 : ld_sample<FMT>(src, static_cast<uint64_t>(k) * C + c);
            }
            for (int64_t n = n_lo; n < n_hi; n++) {
                const int64_t num = n * step;
                const int ph = static_cast<int>(num / S - pos * P);
                const double *__restrict__ f = taps + static_cast<size_t>(ph) * kSwrTaps;
                double val = 0.0;
#pragma unroll
                for (int i = 0; i < kSwrTaps; i++) val = fma(x[i], f[i], val);
                if (S > 1) {  // (as in k_true_peak_swr)
                    double v2 = 0.0;
#pragma unroll
                    for (int i = 0; i < kSwrTaps; i++) v2 = fma(x[i], f[kSwrTaps + i], v2);
                    val += (v2 - val) * inv_src_incr * static_cast<double>(static_cast<uint32_t>(num % S) * frac_scale);
                }
                dst[n * C + c] = val;
            }
        }
    }
}

// The exact-phase case (S == 1: every rate but the phase-interpolated ones) by tiles: a workgroup owns 256 consecutive
// input positions, stages their window (with the reflections) in LDS once per channel, and its threads take the tile's
// outputs in output order -- stores are coalesced, the phase arithmetic is 32-bit and relative to the tile, the 64-bit
// division happens once per workgroup.  Same FMA order as above, so the same bits.
template <int FMT>
__global__ void __launch_bounds__(256) k_resample_swr_tiled(const void *__restrict__ src, int64_t N, int C, int P, int step,
                                                            const double *__restrict__ taps, double *__restrict__ dst)
{
    __shared__ double xs[256 + kSwrTaps];
    const int64_t p0 = static_cast<int64_t>(blockIdx.x) * 256;
    const int np = N - p0 < 256 ? static_cast<int>(N - p0) : 256;
    const int64_t n0 = swr_first_output(p0 * P, step, 1), n1 = swr_first_output((p0 + np) * P, step, 1);
    const uint32_t r0 = static_cast<uint32_t>(n0 * step - p0 * P);  // < step
    const uint32_t count = static_cast<uint32_t>(n1 - n0);
    for (int c = 0; c < C; c++) {
        __syncthreads();
        for (int i = threadIdx.x; i < np + kSwrTaps - 1; i += 256) {
            int64_t k = p0 - 15 + i;
            if (k < 0) k = -k;
            if (k >= N) k = 2 * N - 1 - k;
            xs[i] = (k < 0 || k >= N) ? 0.0 : ld_sample<FMT>(src, static_cast<uint64_t>(k) * C + c);
        }
        __syncthreads();
        for (uint32_t m = threadIdx.x; m < count; m += 256) {
            const uint32_t num = m * static_cast<uint32_t>(step) + r0;
            const uint32_t pr = num / static_cast<uint32_t>(P), ph = num - pr * static_cast<uint32_t>(P);
            const double *__restrict__ f = taps + static_cast<size_t>(ph) * kSwrTaps;
            const double *x = xs + pr;
            double val = 0.0;
#pragma unroll
            for (int i = 0; i < kSwrTaps; i++) val = fma(x[i], f[i], val);
            dst[(n0 + m) * C + c] = val;
        }
    }
}

// The same by phase (exact phases, step > 1: 44.1 kHz and its family).  Outputs P apart share a phase, i.e. their taps,
// and sit `step` input frames apart: a lane takes one such "cycle" (step frames in, P outputs out), the wave walks the
// phases, so the 32 taps of a phase are wave-uniform (scalar loads) instead of 64 different rows of the table per load
// -- the per-thread tap rows from L2 were what the tiled kernel above cost.  The window of 64 cycles (64 * step + 32
// doubles) sits in LDS; lanes read it `step` doubles apart (conflict-free for odd steps).  Same FMA order, same bits.
constexpr int kRsCycles = 64;
template <int FMT>
__global__ void __launch_bounds__(256) k_resample_swr_phase(const void *__restrict__ src, int64_t N, int C, int P, int step,
                                                            const double *__restrict__ taps, double *__restrict__ dst, int64_t n_total)
{
    extern __shared__ double xs_dyn[];  // kRsCycles * step + kSwrTaps
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t cyc0 = static_cast<int64_t>(blockIdx.x) * kRsCycles;
    const int64_t f0 = cyc0 * step;  // first frame of the tile: phase 0 of the pattern
    const int span = kRsCycles * step + kSwrTaps - 1;
    for (int c = 0; c < C; c++) {
        __syncthreads();
        for (int i = threadIdx.x; i < span; i += 256) {
            int64_t k = f0 - 15 + i;
            if (k < 0) k = -k;
            if (k >= N) k = 2 * N - 1 - k;
            xs_dyn[i] = (k < 0 || k >= N) ? 0.0 : ld_sample<FMT>(src, static_cast<uint64_t>(k) * C + c);
        }
        __syncthreads();
        const int64_t nbase = (cyc0 + lane) * P;  // this lane's first output
        const double *x0 = xs_dyn + lane * step;
        for (int ph = wave; ph < P; ph += 4) {    // (wave-uniform)
            const int num = ph * step, u = num / P, tp = num - u * P;  // position within the cycle, phase of the bank
            const double *__restrict__ f = taps + static_cast<size_t>(tp) * kSwrTaps;
            const double *x = x0 + u;
            double val = 0.0;
#pragma unroll
            for (int i = 0; i < kSwrTaps; i++) val = fma(x[i], f[i], val);
            const int64_t n = nbase + ph;
            if (n < n_total) dst[n * C + c] = val;
        }
    }
}

void launch_resample_swr(int fmt, const void *src, uint64_t frames, int channels, int phases, int step, int S, uint32_t frac_scale,
                         double inv_src_incr, const double *taps, double *dst, uint64_t pos0, hipStream_t st)
{
    if (!frames || pos0 >= frames) return;
    if (pos0 == 0 && S == 1 && step > 1 && step <= 160 && phases >= 4) {
        const uint64_t cycles = (frames + static_cast<uint64_t>(step) - 1) / static_cast<uint64_t>(step);
        const uint64_t n_total = (frames * static_cast<uint64_t>(phases) + static_cast<uint64_t>(step) - 1) / static_cast<uint64_t>(step);
        const uint64_t tiles = (cycles + kRsCycles - 1) / kRsCycles;
        const size_t lds = (static_cast<size_t>(kRsCycles) * step + kSwrTaps) * sizeof(double);
        if (tiles < 0x7fffffffull) {
            const dim3 grid(static_cast<uint32_t>(tiles));
#define HR_RP_CASE(F) case F: \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_resample_swr_phase<F>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)); \
            k_resample_swr_phase<F><<<grid, dim3(256), lds, st>>>(src, static_cast<int64_t>(frames), channels, phases, step, taps, dst, static_cast<int64_t>(n_total)); break;
            switch (fmt) { HR_RP_CASE(0) HR_RP_CASE(1) HR_RP_CASE(2) HR_RP_CASE(3) default: HR_RP_CASE(4) }
#undef HR_RP_CASE
            return;
        }
    }
    // (a tile's phase numerators stay below 256 * phases + step)
    if (pos0 == 0 && S == 1 && 256ull * static_cast<uint64_t>(phases) + static_cast<uint64_t>(step) < 0x7fffffffull && (frames + 255) / 256 < 0x7fffffffull) {
        const dim3 grid(static_cast<uint32_t>((frames + 255) / 256));
#define HR_RT_CASE(F) case F: k_resample_swr_tiled<F><<<grid, dim3(256), 0, st>>>(src, static_cast<int64_t>(frames), channels, phases, step, taps, dst); break;
        switch (fmt) { HR_RT_CASE(0) HR_RT_CASE(1) HR_RT_CASE(2) HR_RT_CASE(3) default: HR_RT_CASE(4) }
#undef HR_RT_CASE
        return;
    }
    // (from input position pos0 on: dst still addresses output n at dst[n * channels + c])
    uint64_t blocks = (frames - pos0 + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
#define HR_RS_CASE(F) case F: k_resample_swr<F><<<dim3(static_cast<uint32_t>(blocks)), dim3(256), 0, st>>>(src, static_cast<int64_t>(frames), channels, phases, step, S, frac_scale, inv_src_incr, taps, dst, static_cast<int64_t>(pos0)); break;
    switch (fmt) { HR_RS_CASE(0) HR_RS_CASE(1) HR_RS_CASE(2) HR_RS_CASE(3) default: HR_RS_CASE(4) }
#undef HR_RS_CASE
}

template <int FMT>
__global__ void __launch_bounds__(256) k_sample_peak(const TrackDesc *__restrict__ tracks,
                                                     const WorkItem *__restrict__ items, uint32_t span,
                                                     unsigned long long *__restrict__ peaks)
{
    const WorkItem it = items[blockIdx.x];
    const TrackDesc &t = tracks[it.track];
    const int64_t N = static_cast<int64_t>(t.frames);
    const int C = t.ch;
    for (int c = 0; c < C; c++) {
        double m = 0.0;
        for (uint32_t off = 0; off < span; off += 256) {
            const int64_t n = static_cast<int64_t>(it.first) + off + threadIdx.x;
            if (n >= N) continue;
            const double a = fabs(ld_sample<FMT>(t.pcm, static_cast<uint64_t>(n) * C + c));
            m = a > m ? a : m;
        }
        peak_commit(m, &peaks[static_cast<size_t>(it.track) * kMaxCh + c]);
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
#define HR_FMT_DISPATCH(fmt, CALL)            \
    switch (fmt) {                            \
    case 0: { constexpr int F = 0; CALL; } break; \
    case 1: { constexpr int F = 1; CALL; } break; \
    case 2: { constexpr int F = 2; CALL; } break; \
    case 3: { constexpr int F = 3; CALL; } break; \
    default: { constexpr int F = 4; CALL; } break; \
    }

void launch_hop_zero_state_list(int fmt, const TrackDesc *tracks, const RateDev *rates, const WorkItem *items,
                                uint32_t n_items, HopRec *recs, hipStream_t st)
{
    if (!n_items) return;
    HR_FMT_DISPATCH(fmt, (k_hop_zero_state_list<F><<<dim3((n_items + 63) / 64), dim3(64), 0, st>>>(tracks, rates, items, n_items, recs)));
}

void launch_hop_zero_state(int fmt, const TrackDesc *tracks, const RateDev *rates, const WorkItem *items,
                           uint32_t n_items, HopRec *recs, hipStream_t st)
{
    if (!n_items) return;
    HR_FMT_DISPATCH(fmt, (k_hop_zero_state<F><<<dim3(n_items), dim3(256), 0, st>>>(tracks, rates, items, recs)));
}

void launch_hop_combine(const TrackDesc *tracks, const RateDev *rates, const WorkItem *items, uint32_t n_items,
                        const HopRec *recs, double *hop_sums, bool any_rows, hipStream_t st)
{
    if (!n_items) return;
    hipLaunchKernelGGL(k_hop_combine, dim3(n_items), dim3(256), 0, st, tracks, rates, items, recs, hop_sums);
    // (tracks whose hops are several rows: each kernel skips the other's tracks)
    if (any_rows) hipLaunchKernelGGL(k_hop_combine_rows, dim3(n_items), dim3(256), 0, st, tracks, rates, items, recs, hop_sums);
}

// A track measured through stereo planes (TrackDesc::role 1): its hop sums are those of its planes' channels.
__global__ void __launch_bounds__(256) k_gather_planes(const TrackDesc *__restrict__ tracks, const uint32_t *__restrict__ parents,
                                                       double *__restrict__ hop_sums)
{
    const TrackDesc &p = tracks[parents[blockIdx.x]];
    const uint64_t C = p.ch, total = p.n_hops * C;
    for (uint64_t i = threadIdx.x; i < total; i += 256) {
        const uint64_t h = i / C, c = i % C;
        const TrackDesc &q = tracks[p.first_plane + c / 2];
        hop_sums[p.hop_base + (kPreHop + h) * C + c] = hop_sums[q.hop_base + (kPreHop + h) * 2 + (c & 1)];
    }
}

void launch_gather_planes(const TrackDesc *tracks, const uint32_t *parents, uint32_t n_parents, double *hop_sums, hipStream_t st)
{
    if (!n_parents) return;
    hipLaunchKernelGGL(k_gather_planes, dim3(n_parents), dim3(256), 0, st, tracks, parents, hop_sums);
}

void launch_block_hist(const TrackDesc *tracks, const WorkItem *items, uint32_t n_items, int gate_mode, const double *hop_sums,
                       const double *ln_bounds, const double *ebu_edges, double *block_powers, uint32_t *hist, GatePart *parts,
                       hipStream_t st)
{
    if (!n_items) return;
    hipLaunchKernelGGL(k_block_hist, dim3(n_items), dim3(256), 0, st, tracks, items, gate_mode, hop_sums, ln_bounds, ebu_edges,
                       block_powers, hist, parts);
}

void launch_gate_final(const TrackDesc *tracks, uint32_t n_tracks, int gate_mode, const double *block_powers,
                       const uint32_t *hist, const double *ebu_energy, const double *ebu_edges, const double *ln_energy,
                       const double *ln_bounds, const GatePart *parts, bool reduce_only, TrackOut *outs, hipStream_t st,
                       const WorkItem *list)
{
    if (!n_tracks) return;
    hipLaunchKernelGGL(k_gate_final, dim3(n_tracks), dim3(256), 0, st, tracks, list, gate_mode, block_powers, hist,
                       ebu_energy, ebu_edges, ln_energy, ln_bounds, parts, reduce_only ? 1 : 0, outs);
}

void launch_true_peak_generic(int fmt, int tp_kind, const TrackDesc *tracks, const RateDev *rates,
                              const WorkItem *items, uint32_t n_items, uint32_t span,
                              unsigned long long *peaks, hipStream_t st)
{
    if (!n_items) return;
    if (tp_kind == 0) {
        HR_FMT_DISPATCH(fmt, (k_true_peak_swr<F><<<dim3(n_items), dim3(256), 0, st>>>(tracks, rates, items, span, peaks)));
    } else {
        HR_FMT_DISPATCH(fmt, (k_sample_peak<F><<<dim3(n_items), dim3(256), 0, st>>>(tracks, items, span, peaks)));
    }
}

}  // namespace hr
