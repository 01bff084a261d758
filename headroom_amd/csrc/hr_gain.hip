// hr_gain.hip -- lossless gain: y = x * 10^(g/20), re-quantised to packed 24-bit PCM.
//
// Replaces `ffmpeg -i in -af volume={gain}dB -c:a pcm_s24le|pcm_s24be out` (src/processor.rs:53-87) for PCM that
// is already decoded.  Arithmetic of that FFmpeg chain (from knowledge of the public FFmpeg source; not checkable in
// this image: "parity unpinned", see oracle/hr_oracle.h): the volume filter runs in its default float precision, so
// the input is converted to f32 (libswresample audioconvert: s16 * 2^-15, s32 * 2^-31 with the int -> float
// rounding, f64 -> f32), multiplied by (float)10^(g/20) in f32 (vector_fmul_scalar), converted to s32 for the
// encoder as av_clipl_int32(llrintf(y * 2^31)) (ties to even, saturating), and pcm_s24* writes the upper three bytes.
// HBM-bound streaming kernel: bytes-per-sample read once + 3 bytes written.
#include <hip/hip_runtime.h>

#include "hr_device.hpp"
#include "hr_kernels.hpp"

namespace hr {

template <int FMT>
__device__ __forceinline__ float ld_as_f32(const void *p, uint64_t i)
{
    if constexpr (FMT == 0) return static_cast<const float *>(p)[i];
    else if constexpr (FMT == 1) return static_cast<float>(static_cast<const int16_t *>(p)[i]) * (1.0f / 32768.0f);
    else if constexpr (FMT == 2) {
        const uint8_t *b = static_cast<const uint8_t *>(p) + 3 * i;
        const int32_t v = static_cast<int32_t>(static_cast<uint32_t>(b[0]) << 8 | static_cast<uint32_t>(b[1]) << 16 |
                                               static_cast<uint32_t>(b[2]) << 24);
        return static_cast<float>(v) * (1.0f / 2147483648.0f);
    } else if constexpr (FMT == 3) return static_cast<float>(static_cast<const int32_t *>(p)[i]) * (1.0f / 2147483648.0f);
    else return static_cast<float>(static_cast<const double *>(p)[i]);
}

__device__ __forceinline__ int32_t to_s32_sat(float y, unsigned &clipped)
{
    const float v = y * 2147483648.0f;
    if (v >= 2147483648.0f) { clipped++; return 2147483647; }
    if (v < -2147483648.0f || v != v) { clipped++; return static_cast<int32_t>(-2147483647 - 1); }  // (x86 llrintf(NaN) is the minimum)
    return static_cast<int32_t>(rintf(v));  // an integer-valued float below 2^31: exact
}

// four consecutive samples starting at sample i (i a multiple of 4, the source 16-byte aligned where it is read wide)
template <int FMT>
__device__ __forceinline__ void ld4_as_f32(const void *p, uint64_t i, bool wide, float (&x)[4])
{
    if constexpr (FMT == 0) {
        if (wide) {
            const float4 v = *reinterpret_cast<const float4 *>(static_cast<const float *>(p) + i);
            x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
            return;
        }
    } else if constexpr (FMT == 1) {
        if (wide) {
            const uint2 v = *reinterpret_cast<const uint2 *>(static_cast<const int16_t *>(p) + i);
            x[0] = static_cast<float>(static_cast<int32_t>(v.x << 16) >> 16) * (1.0f / 32768.0f);
            x[1] = static_cast<float>(static_cast<int32_t>(v.x) >> 16) * (1.0f / 32768.0f);
            x[2] = static_cast<float>(static_cast<int32_t>(v.y << 16) >> 16) * (1.0f / 32768.0f);
            x[3] = static_cast<float>(static_cast<int32_t>(v.y) >> 16) * (1.0f / 32768.0f);
            return;
        }
    } else if constexpr (FMT == 2) {
        if (wide) {  // 12 bytes = 3 dwords
            const uint32_t *q = reinterpret_cast<const uint32_t *>(static_cast<const uint8_t *>(p) + 3 * i);
            const uint32_t a = q[0], b = q[1], c = q[2];
            const int32_t s0 = static_cast<int32_t>(a << 8), s1 = static_cast<int32_t>((a >> 24 | b << 8) << 8),
                          s2 = static_cast<int32_t>((b >> 16 | c << 16) << 8), s3 = static_cast<int32_t>(c & 0xffffff00u);
            x[0] = static_cast<float>(s0) * (1.0f / 2147483648.0f);
            x[1] = static_cast<float>(s1) * (1.0f / 2147483648.0f);
            x[2] = static_cast<float>(s2) * (1.0f / 2147483648.0f);
            x[3] = static_cast<float>(s3) * (1.0f / 2147483648.0f);
            return;
        }
    } else if constexpr (FMT == 3) {
        if (wide) {
            const int4 v = *reinterpret_cast<const int4 *>(static_cast<const int32_t *>(p) + i);
            x[0] = static_cast<float>(v.x) * (1.0f / 2147483648.0f);
            x[1] = static_cast<float>(v.y) * (1.0f / 2147483648.0f);
            x[2] = static_cast<float>(v.z) * (1.0f / 2147483648.0f);
            x[3] = static_cast<float>(v.w) * (1.0f / 2147483648.0f);
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) x[j] = ld_as_f32<FMT>(p, i + j);
}

// upper three bytes of a 32-bit sample as they lie in memory, in the low 24 bits of the result
template <bool BE>
__device__ __forceinline__ uint32_t s24_bytes(int32_t s)
{
    const uint32_t v = static_cast<uint32_t>(s) >> 8;
    if constexpr (!BE) return v;
    else return (v >> 16) | (v & 0xff00u) | (v & 0xffu) << 16;
}

template <int FMT, bool BE>
__global__ void __launch_bounds__(256) k_apply_gain(const void *__restrict__ src, uint8_t *__restrict__ dst, uint64_t n_samples,
                                                    float vol, unsigned long long *__restrict__ clipped_total)
{
    unsigned clipped = 0;
    // wide loads need the natural alignment of the 4-sample group: 16 bytes for 4-byte samples, 8 for s16, 4 for s24
    constexpr uintptr_t kAlign = FMT == 1 ? 7 : FMT == 2 ? 3 : 15;
    const bool wide = FMT != 4 && (reinterpret_cast<uintptr_t>(src) & kAlign) == 0;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * 256 * 4;
    for (uint64_t i = (static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x) * 4; i < n_samples; i += stride) {
        if (i + 4 <= n_samples) {
            float x[4];
            ld4_as_f32<FMT>(src, i, wide, x);
            uint32_t v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = s24_bytes<BE>(to_s32_sat(x[j] * vol, clipped));
            uint32_t *d = reinterpret_cast<uint32_t *>(dst + 3 * i);  // 3*i is a multiple of 12, dst dword aligned
            d[0] = v[0] | v[1] << 24;
            d[1] = v[1] >> 8 | v[2] << 16;
            d[2] = v[2] >> 16 | v[3] << 8;
        } else {
            for (uint64_t k = i; k < n_samples; k++) {
                const uint32_t v = s24_bytes<BE>(to_s32_sat(ld_as_f32<FMT>(src, k) * vol, clipped));
                dst[3 * k] = static_cast<uint8_t>(v);
                dst[3 * k + 1] = static_cast<uint8_t>(v >> 8);
                dst[3 * k + 2] = static_cast<uint8_t>(v >> 16);
            }
        }
    }
    // one atomic per wave
    for (int off = 32; off > 0; off >>= 1) clipped += __shfl_down(clipped, off, 64);
    if ((threadIdx.x & 63) == 0 && clipped) atomicAdd(clipped_total, static_cast<unsigned long long>(clipped));
}

void launch_apply_gain(int fmt, bool big_endian, const void *src, void *dst, uint64_t n_samples, float vol,
                       unsigned long long *clipped, hipStream_t st)
{
    if (!n_samples) return;
    uint64_t blocks = (n_samples + 1023) / 1024;
    if (blocks > 256 * 64) blocks = 256 * 64;
    const dim3 grid(static_cast<uint32_t>(blocks)), block(256);
    uint8_t *d = static_cast<uint8_t *>(dst);
#define HR_GAIN_CASE(F)                                                                                   \
    case F:                                                                                               \
        if (big_endian) k_apply_gain<F, true><<<grid, block, 0, st>>>(src, d, n_samples, vol, clipped);  \
        else k_apply_gain<F, false><<<grid, block, 0, st>>>(src, d, n_samples, vol, clipped);            \
        break;
    switch (fmt) {
        HR_GAIN_CASE(0)
        HR_GAIN_CASE(1)
        HR_GAIN_CASE(2)
        HR_GAIN_CASE(3)
        default:
        HR_GAIN_CASE(4)
    }
#undef HR_GAIN_CASE
}

}  // namespace hr
