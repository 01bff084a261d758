// hr_synth.hip -- deterministic synthetic PCM generated in HBM (bench / tests): white
// noise at a per-track level, one tone, -40 dB intro/outro so the gate gates, three fs/4 45-degree
// bursts so true peak != sample peak.  Integer hashing plus explicitly rounded f32 operations only,
// so the CPU generator used by the tests produces the same bits.
#include <hip/hip_runtime.h>

#include <cmath>

#include "hr_kernels.hpp"

namespace hr {

static inline uint64_t mix64_h(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static inline double u01_h(uint64_t seed, uint64_t k)
{
    return static_cast<double>(mix64_h(seed + 0x9E3779B97F4A7C15ull * (k + 1)) >> 11) * (1.0 / 9007199254740992.0);
}

void synth_params(uint64_t seed, uint32_t fs, uint64_t frames, SynthPar *p)
{
    const uint64_t ps = mix64_h(seed ^ 0x48454144ull);
    const double level_db = -30.0 + 24.0 * u01_h(ps, 0);
    const double an = std::pow(10.0, level_db / 20.0);
    const double as = 0.5 * an * (0.5 + u01_h(ps, 1));
    const double f = 40.0 * std::pow(300.0, u01_h(ps, 2));
    p->a_noise = static_cast<float>(an);
    p->a_sine = static_cast<float>(as);
    p->sine_step = static_cast<uint32_t>(std::llround(f / static_cast<double>(fs) * 4294967296.0));
    uint64_t q = 10ull * fs;
    if (q > frames / 8) q = frames / 8;
    p->quiet_len = q;
    double B = 1.35 * (an + as);
    if (B > 0.98) B = 0.98;
    p->burst = static_cast<float>(B * 0.70710678118654752440);
    p->burst_len = 32;
    const uint64_t span = frames > 2 * q + 64 ? frames - 2 * q - 64 : 0;
    for (int k = 0; k < 3; k++)
        p->burst_pos[k] = span ? q + static_cast<uint64_t>(u01_h(ps, 3 + k) * static_cast<double>(span)) : UINT64_MAX;
}

__device__ __forceinline__ uint64_t mix64_d(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void __launch_bounds__(256) k_synth(float *__restrict__ out, uint64_t seed, SynthPar p, uint32_t C,
                                               uint64_t frames)
{
    const uint64_t total = frames * C;
    for (uint64_t idx = blockIdx.x * 256ull + threadIdx.x; idx < total; idx += gridDim.x * 256ull) {
        const uint64_t n = idx / C;
        const uint32_t c = static_cast<uint32_t>(idx % C);
        int in_burst = -1;
#pragma unroll
        for (int k = 0; k < 3; k++)
            if (n >= p.burst_pos[k] && n - p.burst_pos[k] < p.burst_len) in_burst = static_cast<int>(n - p.burst_pos[k]);
        float xv;
        if (in_burst >= 0) {
            xv = ((in_burst + static_cast<int>(c)) & 2) ? -p.burst : p.burst;
        } else {
            const float g = (n < p.quiet_len || n >= frames - p.quiet_len) ? 0.01f : 1.0f;
            const uint64_t z = mix64_d(seed + 0x9E3779B97F4A7C15ull * (idx + 1));
            const float u = __fmul_rn(static_cast<float>(static_cast<int32_t>(z >> 40) - 8388608), 1.0f / 8388608.0f);
            const uint32_t ph = static_cast<uint32_t>(n * static_cast<uint64_t>(p.sine_step)) + c * 0x40000000u;
            const float tt = __fmul_rn(static_cast<float>(static_cast<int32_t>(ph)), 1.0f / 2147483648.0f);
            const float a1 = __fsub_rn(1.0f, fabsf(tt));
            const float y = __fmul_rn(4.0f, __fmul_rn(tt, a1));
            const float d = __fsub_rn(__fmul_rn(y, fabsf(y)), y);
            const float s = __fmaf_rn(0.225f, d, y);
            const float v = __fmaf_rn(p.a_noise, u, __fmul_rn(p.a_sine, s));
            xv = __fmul_rn(g, v);
        }
        out[idx] = xv;
    }
}

void launch_synth(float *out, uint64_t seed, const SynthPar &p, uint32_t channels, uint64_t frames, hipStream_t st)
{
    const uint64_t total = frames * channels;
    if (!total) return;
    uint64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(k_synth, dim3(static_cast<uint32_t>(blocks)), dim3(256), 0, st, out, seed, p, channels, frames);
}

// ---------------------------------------------------------------------------------------------
// Interleaved PCM of any channel count -> stereo f32 planes, exactly.  Plane p holds channels 2p and 2p+1 as
// interleaved f32 pairs (the second of an odd last channel is zero), which is the layout the fused kernel reads: a
// mono, 5.1 or 7.1 track becomes 1, 3 or 4 stereo pseudo-tracks.  A 16- or 24-bit sample times 2^-15 / 2^-23 is
// representable in f32 (24-bit significand) and is the very value the generic kernels compute in f64, so the
// results stay bit-identical.  (s32 and f64 do not fit f32 and stay on the generic kernels.)
// One workgroup moves 256 frames: the interleaved bytes come in as coalesced dwords through LDS (a 5.1 s24 frame is
// 18 bytes: no lane-aligned view of it exists), the planes go out as one float2 per lane.
// ---------------------------------------------------------------------------------------------
// KEEP (s16 / s24 input): the planes keep the input's sample format -- interleaved stereo s16 / packed s24, which the
// fused kernel reads directly -- instead of f32: 4 / 6 bytes written and read again per frame and plane instead of 8.
// The plane stride stays plane_floats * 4 bytes.
template <int FMT, bool KEEP = false>  // 0 f32, 1 s16, 2 s24 (HR_FMT_*)
__global__ void __launch_bounds__(256) k_split_planes(const uint8_t *__restrict__ src, float *__restrict__ dst, uint64_t frames,
                                                      uint32_t channels, uint64_t plane_floats)
{
    constexpr uint32_t B = FMT == 0 ? 4 : FMT == 1 ? 2 : 3;
    __shared__ __attribute__((aligned(16))) uint32_t lds[256 * kMaxCh * 4 / 4];
    const uint32_t fb = channels * B;  // frame bytes
    const uint64_t total_bytes = frames * fb;
    const uint32_t planes = (channels + 1) / 2;
    for (uint64_t f0 = static_cast<uint64_t>(blockIdx.x) * 256; f0 < frames; f0 += static_cast<uint64_t>(gridDim.x) * 256) {
        const uint64_t byte0 = f0 * fb;  // a multiple of 4: the source is dword aligned (checked by the host)
        const uint64_t nbytes = (total_bytes - byte0 < 256ull * fb) ? total_bytes - byte0 : 256ull * fb;
        const uint32_t full = static_cast<uint32_t>(nbytes / 4), tail = static_cast<uint32_t>(nbytes % 4);
        for (uint32_t i = threadIdx.x; i < full; i += 256) lds[i] = reinterpret_cast<const uint32_t *>(src + byte0)[i];
        if (threadIdx.x < tail) reinterpret_cast<uint8_t *>(lds)[full * 4 + threadIdx.x] = src[byte0 + full * 4 + threadIdx.x];
        __syncthreads();
        const uint64_t f = f0 + threadIdx.x;
        if (f < frames) {
            const uint8_t *fp = reinterpret_cast<const uint8_t *>(lds) + threadIdx.x * fb;
            for (uint32_t p = 0; p < planes; p++) {
                if constexpr (KEEP) {
                    uint8_t *o = reinterpret_cast<uint8_t *>(dst + p * plane_floats) + f * (2 * B);
                    for (uint32_t j = 0; j < 2; j++)
                        for (uint32_t k = 0; k < B; k++) o[j * B + k] = (2 * p + j < channels) ? fp[(2 * p + j) * B + k] : uint8_t(0);
                    continue;
                }
                float v[2] = {0.0f, 0.0f};
#pragma unroll
                for (uint32_t j = 0; j < 2; j++) {
                    const uint32_t c = 2 * p + j;
                    if (c >= channels) break;
                    const uint8_t *b = fp + c * B;
                    if constexpr (FMT == 0) {
                        uint32_t u = static_cast<uint32_t>(b[0]) | static_cast<uint32_t>(b[1]) << 8 | static_cast<uint32_t>(b[2]) << 16 |
                                     static_cast<uint32_t>(b[3]) << 24;
                        v[j] = __uint_as_float(u);
                    } else if constexpr (FMT == 1) {
                        const int32_t x = static_cast<int32_t>(static_cast<uint32_t>(b[0]) << 16 | static_cast<uint32_t>(b[1]) << 24) >> 16;
                        v[j] = static_cast<float>(x) * (1.0f / 32768.0f);
                    } else {
                        const int32_t x = static_cast<int32_t>(static_cast<uint32_t>(b[0]) << 8 | static_cast<uint32_t>(b[1]) << 16 |
                                                               static_cast<uint32_t>(b[2]) << 24) >> 8;
                        v[j] = static_cast<float>(x) * (1.0f / 8388608.0f);
                    }
                }
                *reinterpret_cast<float2 *>(dst + p * plane_floats + f * 2) = make_float2(v[0], v[1]);
            }
        }
        __syncthreads();
    }
}

// The same for whole blocks of 1024 frames with the channel count known at compile time: 16 bytes per lane in, four
// frames per lane through LDS (the lane's chunk is CH*B dwords; an even chunk length is padded by one dword so that the
// lanes spread over the banks), every sample cut out of its dword(s) by compile-time shifts, 32 bytes per lane and plane
// out.  Needs a 16-byte aligned source; launch_split_planes sends the ragged end and anything else to the kernel above.
template <int FMT, int CH, bool KEEP = false>
__global__ void __launch_bounds__(256) k_split_planes_wide(const uint8_t *__restrict__ src, float *__restrict__ dst, uint64_t n_blocks,
                                                           uint64_t plane_floats)
{
    constexpr int B = FMT == 0 ? 4 : FMT == 1 ? 2 : 3;
    constexpr int ND = CH * B;                       // dwords of four frames
    constexpr int STRIDE = (ND & 1) ? ND : ND + 1;   // odd: conflict-free reads
    constexpr int PLANES = (CH + 1) / 2;
    __shared__ uint32_t lds[256 * STRIDE];
    typedef unsigned uint4v __attribute__((ext_vector_type(4)));
    for (uint64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const uint4v *in = reinterpret_cast<const uint4v *>(src + blk * (1024ull * CH * B));
#pragma unroll
        for (int j = 0; j < (ND + 3) / 4; j++) {
            const uint32_t q = threadIdx.x + 256u * j;  // 16-byte piece of the block
            if (ND % 4 == 0 || q < 64u * ND) {
                const uint4v u = in[q];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const uint32_t d = q * 4 + e;
                    lds[d + (STRIDE != ND ? d / ND : 0)] = u[e];
                }
            }
        }
        __syncthreads();
        uint32_t w[ND + 1];
#pragma unroll
        for (int d = 0; d < ND; d++) w[d] = lds[threadIdx.x * STRIDE + d];
        w[ND] = 0;
        __syncthreads();
        const uint64_t f = blk * 1024 + threadIdx.x * 4;
#pragma unroll
        for (int p = 0; p < PLANES; p++) {
            if constexpr (KEEP && FMT != 0) {
                // the four frames' two samples as raw 16- / 24-bit fields, packed again: 16 / 24 bytes per lane and plane
                uint32_t q[8];
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        const int c = 2 * p + j;
                        if (c >= CH) { q[2 * i + j] = 0; continue; }
                        const int byte = (i * CH + c) * B, d = byte >> 2, sh = (byte & 3) * 8;
                        if constexpr (FMT == 1) q[2 * i + j] = (w[d] >> sh) & 0xffffu;
                        else q[2 * i + j] = (sh <= 8 ? w[d] >> sh : __funnelshift_r(w[d], w[d + 1], sh)) & 0xffffffu;
                    }
                uint8_t *o = reinterpret_cast<uint8_t *>(dst + p * plane_floats) + f * (2 * B);
                if constexpr (FMT == 1) {
                    *reinterpret_cast<uint4v *>(o) = uint4v{q[0] | q[1] << 16, q[2] | q[3] << 16, q[4] | q[5] << 16, q[6] | q[7] << 16};
                } else {
                    typedef unsigned uint2v __attribute__((ext_vector_type(2)));
                    uint2v *o2 = reinterpret_cast<uint2v *>(o);  // 24 bytes per lane: 8-byte aligned
                    o2[0] = uint2v{q[0] | q[1] << 24, q[1] >> 8 | q[2] << 16};
                    o2[1] = uint2v{q[2] >> 16 | q[3] << 8, q[4] | q[5] << 24};
                    o2[2] = uint2v{q[5] >> 8 | q[6] << 16, q[6] >> 16 | q[7] << 8};
                }
                continue;
            }
            float v[4][2];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int c = 2 * p + j;
                    if (c >= CH) { v[i][j] = 0.0f; continue; }
                    const int byte = (i * CH + c) * B, d = byte >> 2, sh = (byte & 3) * 8;
                    if constexpr (FMT == 0) {
                        v[i][j] = __uint_as_float(w[d]);
                    } else if constexpr (FMT == 1) {
                        v[i][j] = static_cast<float>(static_cast<int32_t>(w[d] << (16 - sh)) >> 16) * (1.0f / 32768.0f);
                    } else {
                        const uint32_t bits = sh <= 8 ? w[d] >> sh : __funnelshift_r(w[d], w[d + 1], sh);
                        v[i][j] = static_cast<float>(static_cast<int32_t>(bits << 8) >> 8) * (1.0f / 8388608.0f);
                    }
                }
            float4 *o = reinterpret_cast<float4 *>(dst + p * plane_floats + f * 2);
            o[0] = make_float4(v[0][0], v[0][1], v[1][0], v[1][1]);
            o[1] = make_float4(v[2][0], v[2][1], v[3][0], v[3][1]);
        }
    }
}

template <int FMT, bool KEEP = false>
static bool launch_split_wide(const uint8_t *s, float *dst, uint64_t n_blocks, uint32_t channels, uint64_t plane_floats, hipStream_t st)
{
    const dim3 grid(static_cast<uint32_t>(n_blocks < 256 * 16 ? n_blocks : 256 * 16)), block(256);
    switch (channels) {
#define HR_SPLIT_CASE(C) case C: k_split_planes_wide<FMT, C, KEEP><<<grid, block, 0, st>>>(s, dst, n_blocks, plane_floats); return true
    HR_SPLIT_CASE(1); HR_SPLIT_CASE(2); HR_SPLIT_CASE(3); HR_SPLIT_CASE(4); HR_SPLIT_CASE(5); HR_SPLIT_CASE(6); HR_SPLIT_CASE(7); HR_SPLIT_CASE(8);
#undef HR_SPLIT_CASE
    default: return false;
    }
}

void launch_split_planes(int fmt, const void *src, float *dst, uint64_t frames, uint32_t channels, uint64_t plane_floats,
                         bool keep_format, hipStream_t st)
{
    if (fmt == 0) keep_format = false;
    if (!frames) return;
    const uint8_t *s = static_cast<const uint8_t *>(src);
    // whole blocks of 1024 frames on the wide kernel (16-byte aligned source and planes), the rest on the generic one
    uint64_t done = 0;
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0 && (plane_floats & 3) == 0 &&
        frames >= 1024) {
        const uint64_t nb = frames / 1024;
        const bool ok = fmt == 0 ? launch_split_wide<0>(s, dst, nb, channels, plane_floats, st)
                      : fmt == 1 ? (keep_format ? launch_split_wide<1, true>(s, dst, nb, channels, plane_floats, st)
                                                : launch_split_wide<1>(s, dst, nb, channels, plane_floats, st))
                                 : (keep_format ? launch_split_wide<2, true>(s, dst, nb, channels, plane_floats, st)
                                                : launch_split_wide<2>(s, dst, nb, channels, plane_floats, st));
        if (ok) done = nb * 1024;
    }
    if (done == frames) return;
    const uint32_t B = fmt == 0 ? 4 : fmt == 1 ? 2 : 3;
    s += done * channels * B;
    // (the generic kernel addresses plane p at dst + p * plane_floats plus its own frame offset)
    dst = keep_format ? reinterpret_cast<float *>(reinterpret_cast<uint8_t *>(dst) + done * 2 * B) : dst + done * 2;
    frames -= done;
    uint64_t blocks = (frames + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    const dim3 grid(static_cast<uint32_t>(blocks)), block(256);
    if (fmt == 0) k_split_planes<0><<<grid, block, 0, st>>>(s, dst, frames, channels, plane_floats);
    else if (fmt == 1 && keep_format) k_split_planes<1, true><<<grid, block, 0, st>>>(s, dst, frames, channels, plane_floats);
    else if (fmt == 1) k_split_planes<1><<<grid, block, 0, st>>>(s, dst, frames, channels, plane_floats);
    else if (keep_format) k_split_planes<2, true><<<grid, block, 0, st>>>(s, dst, frames, channels, plane_floats);
    else k_split_planes<2><<<grid, block, 0, st>>>(s, dst, frames, channels, plane_floats);
}

}  // namespace hr
