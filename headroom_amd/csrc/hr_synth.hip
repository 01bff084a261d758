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
// Integer PCM -> f32, exactly: a 16- or 24-bit sample times 2^-15 / 2^-23 is representable in f32 (24-bit
// significand) and is the very value the generic kernels compute in f64, so a widened s16 / s24 stereo track can
// take the fused f32 kernel with bit-identical results.  (s32 and f64 do not fit f32 and stay generic.)
// ---------------------------------------------------------------------------------------------
template <int FMT>
__global__ void __launch_bounds__(256) k_widen_pcm(const uint8_t *__restrict__ src, float *__restrict__ dst, uint64_t n_samples)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * 256 * 4;
    for (uint64_t i = (static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x) * 4; i < n_samples; i += stride) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint64_t k = i + j < n_samples ? i + j : n_samples - 1;
            if constexpr (FMT == 1) {
                v[j] = static_cast<float>(reinterpret_cast<const int16_t *>(src)[k]) * (1.0f / 32768.0f);
            } else {
                const uint8_t *b = src + 3 * k;
                const int32_t x = static_cast<int32_t>(static_cast<uint32_t>(b[0]) << 8 | static_cast<uint32_t>(b[1]) << 16 |
                                                       static_cast<uint32_t>(b[2]) << 24) >> 8;
                v[j] = static_cast<float>(x) * (1.0f / 8388608.0f);
            }
        }
        if (i + 4 <= n_samples) *reinterpret_cast<float4 *>(dst + i) = make_float4(v[0], v[1], v[2], v[3]);
        else
            for (int j = 0; i + j < n_samples; j++) dst[i + j] = v[j];
    }
}

void launch_widen_pcm(int fmt, const void *src, float *dst, uint64_t n_samples, hipStream_t st)
{
    if (!n_samples) return;
    uint64_t blocks = (n_samples + 1023) / 1024;
    if (blocks > 256 * 64) blocks = 256 * 64;
    if (fmt == 1) k_widen_pcm<1><<<dim3(static_cast<uint32_t>(blocks)), dim3(256), 0, st>>>(static_cast<const uint8_t *>(src), dst, n_samples);
    else k_widen_pcm<2><<<dim3(static_cast<uint32_t>(blocks)), dim3(256), 0, st>>>(static_cast<const uint8_t *>(src), dst, n_samples);
}

}  // namespace hr
