// Micro-benchmark (tools/exp, not part of the library): do v_mfma_f32_16x16x32_f16 and ordinary VALU instructions of
// ONE wave / of several waves of a SIMD execute concurrently on gfx950?  Three kernels with the same instruction
// counts per iteration: 8 MFMAs (4 independent accumulators), 24 VALU (independent fma chains), and both interleaved
// 1 : 3 -- at 1, 2 and 3 waves per SIMD.  Co-execution shows as t(both) ~ max(t(mfma), t(valu)), none as the sum.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE>   // 1: MFMA only, 2: VALU only, 3: both interleaved
__global__ void __launch_bounds__(256) k(float* out, float w, int iters)
{
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i * 0.01f); }
    f4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = threadIdx.x * 0.002f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (MODE & 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
                if (MODE & 2) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        float& x = v[(r * 4 + i) % 4 * 3 + j];
                        asm volatile("v_fma_f32 %0, %0, %1, 0.5" : "+v"(x) : "v"(w));
                    }
                }
            }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 12; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    float* d; hipMalloc(&d, 256 * 16 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wg_per_cu = 1; wg_per_cu <= 3; ++wg_per_cu) {        // 256 threads = 4 waves = one per SIMD
        const int grid = 256 * wg_per_cu;
        float ms[4];
        for (int mode = 1; mode <= 3; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, d, 0.999f, iters);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, d, 0.999f, iters);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, d, 0.999f, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms[mode], e0, e1);
            }
        }
        const double mf = 8.0 * iters * grid * 4, va = 24.0 * iters * grid * 4;
        printf("%d wave(s)/SIMD: mfma only %.2f ms (%.1f cycles/MFMA/SIMD at 2.4 GHz), valu only %.2f ms (%.2f cycles/VALU), both %.2f ms  -> both / (mfma + valu) = %.2f, both / max = %.2f\n",
               wg_per_cu, ms[1], ms[1] * 1e-3 * 2.4e9 / (mf / 1024), ms[2], ms[2] * 1e-3 * 2.4e9 / (va / 1024), ms[3],
               ms[3] / (ms[1] + ms[2]), ms[3] / (ms[1] > ms[2] ? ms[1] : ms[2]));
    }
    return 0;
}
