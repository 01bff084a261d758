// Per-instruction issue cost on gfx950 of the f32/f16 helper ops the true-peak screen uses.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench3 tools/microbench3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k_ops(float *out, int iters, float seed)
{
    float a[8], b[8];
    unsigned u[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = seed + i + threadIdx.x; b[i] = seed * 0.5f - i; u[i] = threadIdx.x * 7 + i; }
    unsigned long long acc = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if constexpr (MODE == 0) {  // v_max3_f32 with abs
                    a[i] = fmaxf(fmaxf(fabsf(b[i]), fabsf(b[(i + 1) & 7])), a[i]);
                } else if constexpr (MODE == 1) {  // v_cvt_pkrtz
                    auto h = __builtin_amdgcn_cvt_pkrtz(a[i], b[i]);
                    unsigned w; __builtin_memcpy(&w, &h, 4);
                    u[i] ^= w;   // (+1 v_xor)
                } else if constexpr (MODE == 2) {  // v_permlane32_swap
                    auto sw = __builtin_amdgcn_permlane32_swap(u[i], u[(i + 4) & 7], false, false);
                    u[i] = sw[0]; u[(i + 4) & 7] = sw[1];
                } else if constexpr (MODE == 3) {  // v_cmp + ballot accumulate on SALU
                    acc |= __ballot(a[i] > b[i]);
                    a[i] += 1.0f;  // (+1 v_add)
                } else if constexpr (MODE == 4) {  // plain f32 fma
                    a[i] = fmaf(a[i], 1.0001f, b[i]);
                } else if constexpr (MODE == 5) {  // v_cmp + branch on the ballot
                    if (__ballot(a[i] > 1e30f) != 0ull) a[i] = 0.0f;
                    a[i] += 1.0f;
                }
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i] + b[i] + u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)acc;
}

template <int MODE>
static void run(const char *name, float *out)
{
    for (int wg : {1, 2}) {
        const int iters = 4000;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        k_ops<MODE><<<256 * wg, 256>>>(out, 10, 1.0f);
        CHECK(hipEventRecord(e0));
        k_ops<MODE><<<256 * wg, 256>>>(out, iters, 1.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("%-40s wg/cu %d: %7.3f ms  %6.2f ns per op (SIMD time)\n", name, wg, ms, ms * 1e6 / (double(iters) * 32 * wg));
    }
}

int main()
{
    float *out; CHECK(hipMalloc(&out, 256 * 2 * 256 * sizeof(float)));
    run<4>("v_fma_f32", out);
    run<0>("v_max3_f32 |a|,|b|,c", out);
    run<1>("v_cvt_pkrtz_f16_f32 (+v_xor)", out);
    run<2>("v_permlane32_swap", out);
    run<3>("v_cmp + s_or ballot (+v_add)", out);
    run<5>("v_cmp + s_cbranch on ballot (+v_add)", out);
    std::printf("microbench3 exit 0\n");
    return 0;
}
