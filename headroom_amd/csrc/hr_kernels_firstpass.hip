// hr_kernels_firstpass.hip -- the literal loudnorm first pass without the 192 kHz stream in memory.
//
// `ffmpeg -af loudnorm=print_format=json` (src/analyzer.rs:253-267) resamples to 192 kHz (aresample, libswresample
// defaults) and measures THAT stream: libebur128's K-weighting at 192 kHz, gating, and the largest resampled sample as
// "true peak".  hr_loudnorm_first_pass used to write the resampled f64 stream to HBM (553 MB per 3-minute stereo track)
// and measure it with the generic one-lane-per-hop kernels.  k_first_pass_rows produces the stream and consumes it in
// registers: a lane owns one 192 kHz hop (19 200 outputs = fs/10 input positions: whole resampler cycles, so the 64 lanes
// of a wave are at the same phase of the bank at every step and the 32 taps of an output are wave-uniform scalar
// operands), walks its positions with the 32-frame window sliding through registers (static indices: the loop is expanded
// 32 positions deep, as in k_true_peak_rows), and hands every output, in time order, to the hop's zero-state K-weighting
// (operation order of k_hop_zero_state; the carry-table row of an output is wave-uniform too).  It leaves the hop's
// zero-state record and the largest |output|; the combine / gating kernels behind it do not know the difference.
// One wave per (64 hops, channel).  The resampled values are those of k_resample_swr (same FMA chain), the records
// those of k_hop_zero_state<f64> on them.  The last hop(s) of a track, whose windows reach the end reflection and the
// resampler's flush, stay with the old path on a resampled tail of a few thousand frames.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "hr_device.hpp"
#include "hr_kernels.hpp"

namespace hr {

typedef const __attribute__((address_space(4))) double *cdouble_p;

__device__ __forceinline__ double fp_sample(int fmt, const void *p, uint64_t i)
{
    switch (fmt) {
    case 0: return static_cast<double>(static_cast<const float *>(p)[i]);
    case 1: return static_cast<double>(static_cast<const int16_t *>(p)[i]) * (1.0 / 32768.0);
    case 2: {
        const uint8_t *b = static_cast<const uint8_t *>(p) + 3 * i;
        const int32_t v = static_cast<int32_t>(static_cast<uint32_t>(b[0]) << 8 | static_cast<uint32_t>(b[1]) << 16 |
                                               static_cast<uint32_t>(b[2]) << 24);
        return static_cast<double>(v) * (1.0 / 2147483648.0);
    }
    case 3: return static_cast<double>(static_cast<const int32_t *>(p)[i]) * (1.0 / 2147483648.0);
    default: return static_cast<const double *>(p)[i];
    }
}

__device__ __forceinline__ double fp_wave_max(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

__global__ void __launch_bounds__(64) k_first_pass_rows(const FpSrc *__restrict__ srcs, const RateDev *__restrict__ rates,
                                                        const FpItem *__restrict__ items, HopRec *__restrict__ recs,
                                                        unsigned long long *__restrict__ peaks)
{
    const FpItem it = items[blockIdx.x];
    const FpSrc &s = srcs[it.src];
    const RateDev &r = rates[s.rate192];
    const KCoef k = r.k;
    const cdouble_p g = (cdouble_p)r.g;      // [19200][4]
    const cdouble_p taps = (cdouble_p)s.taps;
    const int lane = threadIdx.x;
    const int n_rows = static_cast<int>(it.n_rows);
    const int row = lane < n_rows ? lane : n_rows - 1;  // idle lanes shadow the last row
    const uint64_t h = static_cast<uint64_t>(it.first_hop) + row;
    const int c = static_cast<int>(it.ch), C = static_cast<int>(s.channels), fmt = static_cast<int>(s.fmt);
    const int P = s.phases, step = s.step, K = static_cast<int>(s.hop_in);
    const int64_t N = static_cast<int64_t>(s.frames);
    const int64_t p0 = static_cast<int64_t>(h) * K;
    const void *pcm = s.pcm;
    // frame f of this wave's channel; libswresample's start reflection x[-j] = x[j]; (frames past the end are only ever
    // prefetched, never used: the host keeps the rows' windows inside the track)
    auto frame = [&](int64_t f) {
        f = f < 0 ? -f : f;
        f = f >= N ? N - 1 : f;
        return fp_sample(fmt, pcm, static_cast<uint64_t>(f) * C + c);
    };
    auto load4 = [&](double (&dst)[4], int64_t f0) {
#pragma unroll
        for (int e = 0; e < 4; e++) dst[e] = frame(f0 + e);
    };

    // input history of the K-weighting: the two outputs in front of the hop (the filter state starts from zero: the
    // combine kernel brings the carry)
    double x1 = 0.0, x2 = 0.0;
    if (h > 0) {
        const int64_t n0 = static_cast<int64_t>(h) * (static_cast<int64_t>(K) * P / step);  // the hop's first output
        for (int b = 1; b <= 2; b++) {
            const int64_t num = (n0 - b) * step, pos = num / P;
            const cdouble_p f = taps + static_cast<size_t>(num - pos * P) * kSwrTaps;
            double v = 0.0;
            for (int j = 0; j < kSwrTaps; j++) v = fma(frame(pos - 15 + j), f[j], v);
            if (b == 1) x1 = v; else x2 = v;
        }
    }

    double w[32];
    double ca[4], cb[4];
#pragma unroll
    for (int j = 0; j < 31; j++) w[j] = frame(p0 - 15 + j);
    w[31] = 0.0;
    load4(ca, p0 + 16);  // what positions 0 .. 3 bring into the window
    double y1 = 0, y2 = 0, z1 = 0, z2 = 0, e0 = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0, m = 0.0;
    int d = 0;           // phase of the next output, relative to the current position (wave-uniform)
    uint32_t n = 0;      // outputs of the hop so far (wave-uniform): row of the carry table
    for (int b = 0; b < K; b += 32) {
        auto position = [&](auto uc) {
            constexpr int u = decltype(uc)::value;
            const int i = b + u;
            if (i >= K) return false;
            // (one chunk of 4 frames ahead of its use: A serves u mod 8 = 0..3, B serves 4..7)
            if constexpr ((u & 7) == 0) load4(cb, p0 + i + 20);
            if constexpr ((u & 7) == 4) load4(ca, p0 + i + 20);
            w[(u + 31) & 31] = (u & 4) ? cb[u & 3] : ca[u & 3];  // frame p0 + i + 16
            while (d < P) {  // the outputs that sit at this position: phase d of the bank
                const cdouble_p f = taps + static_cast<size_t>(d) * kSwrTaps;
                double x0 = 0.0;
#pragma unroll
                for (int j = 0; j < kSwrTaps; j++) x0 = fma(w[(u + j) & 31], f[j], x0);
                const cdouble_p gn = g + static_cast<size_t>(n) * 4;
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
                q0 = fma(z0, gn[0], q0);
                q1 = fma(z0, gn[1], q1);
                q2 = fma(z0, gn[2], q2);
                q3 = fma(z0, gn[3], q3);
                x2 = x1; x1 = x0;
                y2 = y1; y1 = y0;
                z2 = z1; z1 = z0;
                const double a = fabs(x0);
                m = a > m ? a : m;
                d += step;
                n++;
            }
            d -= P;
            return true;
        };
#define HR_POS(u) if (!position(std::integral_constant<int, u>{})) break;
#define HR_POS8(u) HR_POS(u) HR_POS(u + 1) HR_POS(u + 2) HR_POS(u + 3) HR_POS(u + 4) HR_POS(u + 5) HR_POS(u + 6) HR_POS(u + 7)
        HR_POS8(0) HR_POS8(8) HR_POS8(16) HR_POS8(24)
#undef HR_POS8
#undef HR_POS
    }
    if (lane < n_rows) {
        HopRec &o = recs[s.rec_base + (kPreRec + h) * C + c];
        o.e0 = e0;
        o.q[0] = q0; o.q[1] = q1; o.q[2] = q2; o.q[3] = q3;
        o.s[0] = y1; o.s[1] = y1 - y2; o.s[2] = z1; o.s[3] = z1 - z2;
    }
    m = fp_wave_max(m);
    if (lane == 0 && m > 0.0)
        atomicMax(&peaks[static_cast<size_t>(s.desc) * kMaxCh + c], static_cast<unsigned long long>(__double_as_longlong(m)));
}

void launch_first_pass_rows(const FpSrc *srcs, const RateDev *rates, const FpItem *items, uint32_t n_items, HopRec *recs,
                            unsigned long long *peaks, hipStream_t st)
{
    if (!n_items) return;
    k_first_pass_rows<<<dim3(n_items), dim3(64), 0, st>>>(srcs, rates, items, recs, peaks);
}

}  // namespace hr
