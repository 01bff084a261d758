// Issue-rate microbenchmark for gfx950: does f64 VALU work overlap with f16 MFMA work on one SIMD?
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench2 tools/microbench2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)

// MODE 0: f64 FMA only (NF per iteration, 8 independent chains)
// MODE 1: MFMA only (NM per iteration, NACC independent accumulators)
// MODE 2: both, interleaved NF/NM FMAs after every MFMA
// MODE 3: MFMA in dependent chains of 3 on one accumulator, then reduce (like the screen), no FMAs
template <int MODE, int NF, int NM, int NACC>
__global__ void __launch_bounds__(256) k_issue(double *out, float *outf, int iters, double seed)
{
    double c[8];
#pragma unroll
    for (int i = 0; i < 8; i++) c[i] = seed + i + threadIdx.x * 1e-9;
    const double m1 = 1.0000001, m2 = 1e-9;
    floatx16 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; a++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[a][i] = 0.0f;
    half8 A, B;
#pragma unroll
    for (int i = 0; i < 8; i++) { A[i] = (_Float16)(0.001f * (threadIdx.x & 7)); B[i] = (_Float16)(0.002f * i); }
    for (int it = 0; it < iters; it++) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int j = 0; j < NF; j++) c[j & 7] = fma(c[j & 7], m1, m2);
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int j = 0; j < NM; j++) acc[j % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc[j % NACC], 0, 0, 0);
        } else if constexpr (MODE == 2) {
#pragma unroll
            for (int j = 0; j < NM; j++) {
                acc[j % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc[j % NACC], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int f = 0; f < NF / NM; f++) c[f & 7] = fma(c[f & 7], m1, m2);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NM / 3; j++) {
                floatx16 t = {0};
                t = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, t, 0, 0, 0);
                t = __builtin_amdgcn_mfma_f32_32x32x16_f16(B, A, t, 0, 0, 0);
                t = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, A, t, 0, 0, 0);
                float m = 0.0f;
#pragma unroll
                for (int i = 0; i < 16; i += 2) m = fmaxf(fmaxf(fabsf(t[i]), fabsf(t[i + 1])), m);
                acc[0][j & 15] += m;
                A[j & 7] += (_Float16)m;
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += c[i];
    float sf = 0;
#pragma unroll
    for (int a = 0; a < NACC; a++)
#pragma unroll
        for (int i = 0; i < 16; i++) sf += acc[a][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    outf[blockIdx.x * blockDim.x + threadIdx.x] = sf;
}

template <int MODE, int NF, int NM, int NACC>
static void run(const char *name, int wg_per_cu, double *out, float *outf)
{
    const int iters = 2000;
    const int grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    k_issue<MODE, NF, NM, NACC><<<grid, 256>>>(out, outf, 10, 1.0);
    CHECK(hipEventRecord(e0));
    k_issue<MODE, NF, NM, NACC><<<grid, 256>>>(out, outf, iters, 1.0);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    // per SIMD: wg_per_cu waves (256 threads = 4 waves = one per SIMD)
    const double waves_per_simd = wg_per_cu;
    const double fma_per_simd = (MODE == 0 || MODE == 2) ? waves_per_simd * iters * (double)NF : 0;
    const double mfma_per_simd = (MODE != 0) ? waves_per_simd * iters * (double)NM : 0;
    std::printf("%-34s wg/cu %d: %7.3f ms | ns per f64 FMA %6.2f | ns per MFMA %6.2f\n", name, wg_per_cu, ms,
                fma_per_simd ? ms * 1e6 / fma_per_simd : 0.0, mfma_per_simd ? ms * 1e6 / mfma_per_simd : 0.0);
}

int main()
{
    double *out;
    float *outf;
    CHECK(hipMalloc(&out, 256 * 8 * 256 * sizeof(double)));
    CHECK(hipMalloc(&outf, 256 * 8 * 256 * sizeof(float)));
    for (int w : {1, 2, 4}) {
        if (w == 1) { run<0, 96, 12, 4>("f64 fma x96", 1, out, outf); run<1, 96, 12, 4>("mfma x12 (4 acc)", 1, out, outf); run<1, 96, 12, 1>("mfma x12 (1 acc, dependent)", 1, out, outf);
                      run<2, 96, 12, 4>("mfma x12 + fma x96 interleaved", 1, out, outf); run<3, 96, 12, 1>("mfma chains of 3 + reduce", 1, out, outf); }
        if (w == 2) { run<0, 96, 12, 4>("f64 fma x96", 2, out, outf); run<1, 96, 12, 4>("mfma x12 (4 acc)", 2, out, outf); run<1, 96, 12, 1>("mfma x12 (1 acc, dependent)", 2, out, outf);
                      run<2, 96, 12, 4>("mfma x12 + fma x96 interleaved", 2, out, outf); run<3, 96, 12, 1>("mfma chains of 3 + reduce", 2, out, outf); }
        if (w == 4) { run<0, 96, 12, 4>("f64 fma x96", 4, out, outf); run<1, 96, 12, 4>("mfma x12 (4 acc)", 4, out, outf); run<1, 96, 12, 1>("mfma x12 (1 acc, dependent)", 4, out, outf);
                      run<2, 96, 12, 4>("mfma x12 + fma x96 interleaved", 4, out, outf); run<3, 96, 12, 1>("mfma chains of 3 + reduce", 4, out, outf); }
    }
    std::printf("microbench2 exit 0\n");
    return 0;
}
