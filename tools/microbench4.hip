// What does fetching one 128-byte piece per lane-row cost a wave that is busy with f64 arithmetic?
//   mode 0: arithmetic only;  mode 1: 8 x global_load_lds_dwordx4 (LDS DMA) + 8 x ds_read_b128;
//   mode 2: 8 x global_load_dwordx4 per lane straight into registers (second register set)
//   mode 3: as mode 1 with two piece buffers per wave and the DMA two pieces ahead
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench4 tools/microbench4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)

constexpr unsigned kRowBytes = 35280;  // one 44.1 kHz stereo f32 hop
constexpr int kSteps = 275;

template <int MODE, int NFMA>
__global__ void __launch_bounds__(256, 2) k_fetch(const char *__restrict__ pcm, double *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t wave_idx = blockIdx.x * 4 + wave;
    const char *wave_base = pcm + wave_idx * 64 * kRowBytes;
    char *lbuf = smem + wave * 8192;
    unsigned dma_off[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int slot = 8 * j + (lane >> 3);
        dma_off[j] = slot * kRowBytes + ((lane & 7) ^ ((slot >> 1) & 7)) * 16;
    }
    const unsigned row_off = lane * kRowBytes;
    double c[8];
#pragma unroll
    for (int i = 0; i < 8; i++) c[i] = 1.0 + i + lane * 1e-9;
    float4 v[8], vn[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = vn[i] = make_float4(0, 0, 0, 0);
    auto issue = [&](int k) {
        const char *src = wave_base + static_cast<size_t>(k) * 128;
        if constexpr (MODE == 3) {
            char *dst = smem + wave * 16384 + (k & 1) * 8192;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                unsigned o = dma_off[j];
                asm volatile("" : "+v"(o));
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + o),
                                                 (__attribute__((address_space(3))) void *)(dst + j * 1024), 16, 0, 0);
            }
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                unsigned o = dma_off[j];
                asm volatile("" : "+v"(o));
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + o),
                                                 (__attribute__((address_space(3))) void *)(lbuf + j * 1024), 16, 0, 0);
            }
        } else if constexpr (MODE == 2) {
            unsigned o = row_off;
            asm volatile("" : "+v"(o));
#pragma unroll
            for (int j = 0; j < 8; j++) vn[j] = *reinterpret_cast<const float4 *>(src + o + j * 16);
        }
    };
    issue(0);
    if constexpr (MODE == 3) issue(1);
    for (int k = 0; k < kSteps; k++) {
        if constexpr (MODE == 3) {
            // piece k is the older of the two requests in flight: 8 newer DMA instructions may stay outstanding
            if (k + 1 < kSteps) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const char *bufk = smem + wave * 16384 + (k & 1) * 8192;
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = *reinterpret_cast<const float4 *>(bufk + lane * 128 + ((i ^ ((lane >> 1) & 7)) * 16));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (k + 2 < kSteps) issue(k + 2);
        } else if constexpr (MODE == 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = *reinterpret_cast<const float4 *>(lbuf + lane * 128 + ((i ^ ((lane >> 1) & 7)) * 16));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = vn[i];
        }
        if (MODE != 3 && k + 1 < kSteps) issue(k + 1);
#pragma unroll
        for (int i = 0; i < 8; i++) c[i] += v[i].x;
#pragma unroll
        for (int j = 0; j < NFMA; j++) c[j & 7] = fma(c[j & 7], 1.0000001, 1e-9);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += c[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int NFMA>
static void run(const char *name, const char *pcm, double *out, int groups)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k_fetch<MODE, NFMA><<<groups, 256, 65536>>>(pcm, out);
    CHECK(hipEventRecord(e0));
    k_fetch<MODE, NFMA><<<groups, 256, 65536>>>(pcm, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = double(groups) * 4 * 64 * kSteps * 128;
    std::printf("%-44s %7.3f ms  %6.2f TB/s-equivalent\n", name, ms, bytes / ms / 1e9);
}

int main()
{
    const int groups = 1450;  // 5800 waves, like 200 three-minute tracks
    const size_t bytes = size_t(groups) * 4 * 64 * kRowBytes + 4096;
    char *pcm; double *out;
    CHECK(hipMalloc(&pcm, bytes)); CHECK(hipMemset(pcm, 0, bytes));
    CHECK(hipMalloc(&out, size_t(groups) * 256 * sizeof(double)));
    run<0, 480>("480 f64 fma, no fetch", pcm, out, groups);
    run<1, 480>("480 f64 fma + LDS DMA + ds_read", pcm, out, groups);
    run<2, 480>("480 f64 fma + per-lane loads to registers", pcm, out, groups);
    run<0, 960>("960 f64 fma, no fetch", pcm, out, groups);
    run<1, 960>("960 f64 fma + LDS DMA + ds_read", pcm, out, groups);
    run<2, 960>("960 f64 fma + per-lane loads to registers", pcm, out, groups);
    run<3, 480>("480 f64 fma + LDS DMA two pieces ahead", pcm, out, groups);
    run<3, 960>("960 f64 fma + LDS DMA two pieces ahead", pcm, out, groups);
    run<3, 1200>("1200 f64 fma + LDS DMA two pieces ahead", pcm, out, groups);
    run<1, 1200>("1200 f64 fma + LDS DMA + ds_read", pcm, out, groups);
    run<0, 1200>("1200 f64 fma, no fetch", pcm, out, groups);
    run<3, 0>("no arithmetic, LDS DMA two pieces ahead", pcm, out, groups);
    run<1, 0>("no arithmetic, LDS DMA + ds_read", pcm, out, groups);
    run<2, 0>("no arithmetic, per-lane loads", pcm, out, groups);
    std::printf("microbench4 exit 0\n");
    return 0;
}
