// hr_kernels_shard.hip -- the exchange records of a time-sharded stream (BASELINE configs[4]) packed, unpacked and
// reduced ON THE DEVICE, in the layouts of include/headroom_hip.h (hr_shard_edge, hr_shard_partial).
//
// Rounds 1-2 copied every record to the host, converted it there and uploaded it again on the other side: six host round
// trips per call beside a fused kernel of half a millisecond per eighth of a stream.  With these kernels the
// records never leave HBM: the right edge of a shard is written where the caller's send buffer is, the left neighbour's edge
// is consumed from the caller's receive buffer, the gating partial is written into the buffer the caller all-gathers, and the
// gathered images are reduced -- in rank order, so the sums are reproducible -- straight into the stand-in track that the
// ordinary finishing kernels (k_gate_final, k_range_final) read.
//
// No counterpart in the reference: ffmpeg reads a file strictly sequentially (src/analyzer.rs:253-267).
#include <hip/hip_runtime.h>

#include "../../include/headroom_hip.h"
#include "hr_device.hpp"
#include "hr_kernels.hpp"

namespace hr {

namespace {

__device__ __forceinline__ double ld_any(const void *p, int fmt, uint64_t i)
{
    switch (fmt) {
    case HR_FMT_F32: return static_cast<double>(static_cast<const float *>(p)[i]);
    case HR_FMT_S16: return static_cast<double>(static_cast<const int16_t *>(p)[i]) * (1.0 / 32768.0);
    case HR_FMT_S24: {
        const uint8_t *b = static_cast<const uint8_t *>(p) + 3 * i;
        const int32_t v = static_cast<int32_t>(static_cast<uint32_t>(b[0]) << 8 | static_cast<uint32_t>(b[1]) << 16 | static_cast<uint32_t>(b[2]) << 24);
        return static_cast<double>(v) * (1.0 / 2147483648.0);
    }
    case HR_FMT_S32: return static_cast<double>(static_cast<const int32_t *>(p)[i]) * (1.0 / 2147483648.0);
    default: return static_cast<const double *>(p)[i];
    }
}

// The right edge of the shard in descriptor 0: the records of its last six hops, its last 16 frames, the energies of its last
// 29 hops (what hr_shard_pass1 assembles on the host).
__global__ void __launch_bounds__(256) k_pack_edge(const TrackDesc *__restrict__ tracks, const HopRec *__restrict__ recs,
                                                   const double *__restrict__ hop_sums, int is_first, hr_shard_edge *__restrict__ edge)
{
    const TrackDesc &t = tracks[0];
    const int C = t.ch, tid = threadIdx.x;
    const uint64_t nh = t.n_hops;
    unsigned long long *words = reinterpret_cast<unsigned long long *>(edge);
    for (unsigned i = tid; i < sizeof(hr_shard_edge) / 8; i += blockDim.x) words[i] = 0ull;
    __syncthreads();
    const int take = nh < static_cast<uint64_t>(kPreRec) ? static_cast<int>(nh) : kPreRec;
    for (int i = tid; i < take * C * 9; i += blockDim.x) {
        const int k = i / (C * 9), c = (i / 9) % C, j = i % 9;
        const double *r = reinterpret_cast<const double *>(&recs[t.rec_base + (kPreRec + nh - take + k) * C + c]);  // {e0, q[4], s[4]}
        edge->rec[kPreRec - take + k][c][j] = r[j];
    }
    const int tf = t.frames < static_cast<uint64_t>(kHalo) ? static_cast<int>(t.frames) : kHalo;
    for (int i = tid; i < tf * C; i += blockDim.x) {
        const int k = i / C, c = i % C;
        edge->tail[kHalo - tf + k][c] = ld_any(t.pcm, t.fmt, (t.frames - tf + k) * C + c);
    }
    const bool range_ok = nh >= static_cast<uint64_t>(kPreHop + (is_first ? 0 : kPreBlk));
    if (range_ok)
        for (int i = tid; i < kPreHop * C; i += blockDim.x) {
            const int k = i / C, c = i % C;
            edge->hop_tail[k][c] = hop_sums[t.hop_base + (kPreHop + nh - kPreHop + k) * C + c];
        }
    if (tid == 0) {
        edge->n_hops = nh;
        edge->range_ok = range_ok ? 1 : 0;
    }
}

// The left neighbour's edge into the slots in front of hop 0 of the shard in descriptor 0 (what hr_shard_pass2 uploads): its
// six records, and -- when it could hand them over -- its last 29 hop energies, which make the shard's first 3 s windows
// its own (st_pre; the descriptor on the device is patched, the kernels behind this one read it there).
__global__ void __launch_bounds__(256) k_unpack_edge(TrackDesc *__restrict__ tracks, HopRec *__restrict__ recs, double *__restrict__ hop_sums,
                                                     const hr_shard_edge *__restrict__ left)
{
    const TrackDesc &t = tracks[0];
    const int C = t.ch, tid = threadIdx.x;
    for (int i = tid; i < kPreRec * C * 9; i += blockDim.x) {
        const int k = i / (C * 9), c = (i / 9) % C, j = i % 9;
        reinterpret_cast<double *>(&recs[t.rec_base + k * C + c])[j] = left->rec[k][c][j];
    }
    if (left->range_ok) {
        for (int i = tid; i < kPreHop * C; i += blockDim.x) hop_sums[t.hop_base + i] = left->hop_tail[i / C][i % C];
        __syncthreads();
        if (tid == 0) tracks[0].st_pre = kPreHop;
    }
}

// This shard's gating partial (what hr_shard_pass2 assembles on the host) from the workspace the gating kernels left.
// reserved = 1 when the true-peak screen left work for the exact redo (a candidate list overflowed, samples beyond the f16
// range): the peaks in this image are then lower bounds until hr_shard_redo_device has patched them.
__global__ void __launch_bounds__(256) k_pack_partial(const TrackDesc *__restrict__ tracks, const uint32_t *__restrict__ hist,
                                                      const uint32_t *__restrict__ st_hist, const GatePart *__restrict__ parts,
                                                      const TrackOut *__restrict__ outs, const unsigned long long *__restrict__ peaks,
                                                      const unsigned *__restrict__ overflow, const hr_shard_edge *__restrict__ left, int nb,
                                                      hr_shard_partial *__restrict__ out)
{
    const TrackDesc &t = tracks[0];
    const int tid = threadIdx.x;
    unsigned long long *words = reinterpret_cast<unsigned long long *>(out);
    for (unsigned i = tid; i < sizeof(hr_shard_partial) / 8; i += blockDim.x) words[i] = 0ull;
    __syncthreads();
    for (int i = tid; i < nb; i += blockDim.x) {
        out->hist[i] = hist[i];
        out->st_hist[i] = st_hist[i];
    }
    if (tid == 0) {
        out->sum_kept = outs[0].sum_kept;
        out->n_kept = outs[0].n_kept;
        out->n_blocks = outs[0].n_blocks;
        for (int c = 0; c < t.ch; c++) out->peak_per_ch[c] = __longlong_as_double(static_cast<long long>(peaks[c]));
        double ssk = 0.0, mm = 0.0, ms = 0.0;
        unsigned long long snk = 0;
        for (uint32_t c = 0; c < t.n_parts; c++) {  // chunk order, as k_range_final sums them
            const GatePart &p = parts[t.part_base + c];
            ssk += p.st_sum_kept;
            snk += p.st_n_kept;
            mm = p.max_m_power > mm ? p.max_m_power : mm;
            ms = p.max_s_power > ms ? p.max_s_power : ms;
        }
        out->st_sum_kept = ssk;
        out->st_n_kept = snk;
        out->max_m_power = mm;
        out->max_s_power = ms;
        out->n_shortterm = t.n_hops + t.st_pre >= 30 ? t.n_hops + t.st_pre - 29 : 0;
        out->range_ok = left ? left->range_ok : 1;
        out->reserved = overflow && overflow[0] ? 1 : 0;
    }
}

// after the exact redo: the peaks of the image are the final ones
__global__ void k_patch_partial_peaks(const TrackDesc *__restrict__ tracks, const unsigned long long *__restrict__ peaks,
                                      hr_shard_partial *__restrict__ out)
{
    if (threadIdx.x < tracks[0].ch) out->peak_per_ch[threadIdx.x] = __longlong_as_double(static_cast<long long>(peaks[threadIdx.x]));
    if (threadIdx.x == 0) out->reserved = 0;
}

// n gathered partial images -> the stand-in track of the finishing kernels (hr_shard_finish builds it on the host): the two
// histograms, one GatePart, descriptor 0, and the scalars the host still needs (ShardReduced).  Sums run in rank order.
__global__ void __launch_bounds__(256) k_reduce_partials(const hr_shard_partial *__restrict__ in, int n, int nb, uint32_t *__restrict__ hist,
                                                         uint32_t *__restrict__ st_hist, GatePart *__restrict__ gp, TrackDesc *__restrict__ tracks,
                                                         ShardReduced *__restrict__ red)
{
    __shared__ int too_large;
    const int tid = threadIdx.x;
    if (tid == 0) too_large = 0;
    __syncthreads();
    for (int i = blockIdx.x * blockDim.x + tid; i < kEbuBins; i += gridDim.x * blockDim.x) {
        unsigned long long a = 0, b = 0;
        if (i < nb)
            for (int r = 0; r < n; r++) { a += in[r].hist[i]; b += in[r].st_hist[i]; }
        if (a > 0xffffffffull || b > 0xffffffffull) atomicOr(&too_large, 1);
        hist[i] = static_cast<uint32_t>(a);
        st_hist[i] = static_cast<uint32_t>(b);
    }
    __syncthreads();
    if (tid == 0 && too_large) atomicOr(&red->too_large, 1);
    if (blockIdx.x == 0 && tid == 0) {
        GatePart g{};
        unsigned long long n_blocks = 0, n_short = 0;
        int range_ok = 1, redo = 0;
        double pk[kMaxCh] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int r = 0; r < n; r++) {
            const hr_shard_partial &p = in[r];
            g.sum_kept += p.sum_kept;
            g.n_kept += p.n_kept;
            g.st_sum_kept += p.st_sum_kept;
            g.st_n_kept += p.st_n_kept;
            g.max_m_power = p.max_m_power > g.max_m_power ? p.max_m_power : g.max_m_power;
            g.max_s_power = p.max_s_power > g.max_s_power ? p.max_s_power : g.max_s_power;
            n_blocks += p.n_blocks;
            n_short += p.n_shortterm;
            range_ok = p.range_ok < range_ok ? p.range_ok : range_ok;
            redo |= p.reserved != 0;
            for (int c = 0; c < kMaxCh; c++) pk[c] = p.peak_per_ch[c] > pk[c] ? p.peak_per_ch[c] : pk[c];
        }
        gp[0] = g;
        TrackDesc d{};
        d.n_blocks = n_blocks;
        d.n_hops = n_short ? n_short + 29 : 0;
        d.part_base = 0;
        d.n_parts = 1;
        tracks[0] = d;
        for (int c = 0; c < kMaxCh; c++) red->peak_per_ch[c] = pk[c];
        red->range_ok = range_ok;
        red->redo = redo;
    }
}

}  // namespace

void launch_pack_edge(const TrackDesc *tracks, const HopRec *recs, const double *hop_sums, bool is_first, void *edge, hipStream_t st)
{
    k_pack_edge<<<dim3(1), dim3(256), 0, st>>>(tracks, recs, hop_sums, is_first ? 1 : 0, static_cast<hr_shard_edge *>(edge));
}

void launch_unpack_edge(TrackDesc *tracks, HopRec *recs, double *hop_sums, const void *left_edge, hipStream_t st)
{
    k_unpack_edge<<<dim3(1), dim3(256), 0, st>>>(tracks, recs, hop_sums, static_cast<const hr_shard_edge *>(left_edge));
}

void launch_pack_partial(const TrackDesc *tracks, const uint32_t *hist, const uint32_t *st_hist, const GatePart *parts, const TrackOut *outs,
                         const unsigned long long *peaks, const unsigned *overflow, const void *left_edge, int nb, void *partial, hipStream_t st)
{
    k_pack_partial<<<dim3(1), dim3(256), 0, st>>>(tracks, hist, st_hist, parts, outs, peaks, overflow, static_cast<const hr_shard_edge *>(left_edge),
                                                  nb, static_cast<hr_shard_partial *>(partial));
}

void launch_patch_partial_peaks(const TrackDesc *tracks, const unsigned long long *peaks, void *partial, hipStream_t st)
{
    k_patch_partial_peaks<<<dim3(1), dim3(64), 0, st>>>(tracks, peaks, static_cast<hr_shard_partial *>(partial));
}

void launch_reduce_partials(const void *partials, int n, int nb, uint32_t *hist, uint32_t *st_hist, GatePart *gp, TrackDesc *tracks,
                            ShardReduced *red, hipStream_t st)
{
    (void)hipMemsetAsync(red, 0, sizeof(ShardReduced), st);
    k_reduce_partials<<<dim3(32), dim3(256), 0, st>>>(static_cast<const hr_shard_partial *>(partials), n, nb, hist, st_hist, gp, tracks, red);
}

}  // namespace hr
