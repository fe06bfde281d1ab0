// cycles per MFMA on one SIMD: v_mfma_f32_16x16x32_f16 against the legacy v_mfma_f32_16x16x16_f16 (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int K>
__global__ void __launch_bounds__(64) rate(float* out, int iters, long long* cyc)
{
    h8 a8; h4 a4;
    for (int i = 0; i < 8; ++i) a8[i] = (_Float16)(threadIdx.x * 0.001f + i);
    for (int i = 0; i < 4; ++i) a4[i] = a8[i];
    f4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // (inline assembly: the builtin form made the compiler rotate the accumulators through AGPR copies)
            if constexpr (K == 32) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %1, %0" : "+v"(acc[i]) : "v"(a8));
            else asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %1, %0" : "+v"(acc[i]) : "v"(a4));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main()
{
    {   // chip-wide: 256 CUs x 4 waves (one per SIMD), wall clock
        float* o2; long long* c2; hipMalloc(&o2, 256 * 1024); hipMalloc(&c2, 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int it2 = 20000;
        for (int k = 0; k < 2; ++k) {
            for (int form = 0; form < 2; ++form) {
                hipEventRecord(e0);
                if (form == 0) hipLaunchKernelGGL(rate<32>, dim3(1024), dim3(64), 0, 0, o2, it2, c2);
                else hipLaunchKernelGGL(rate<16>, dim3(1024), dim3(64), 0, 0, o2, it2, c2);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                printf("chip-wide %s: %.3f ms for %d MFMAs per wave -> %.1f ns per MFMA per SIMD\n", form ? "16x16x16" : "16x16x32", ms, it2 * 8, ms * 1e6 / (it2 * 8));
            }
        }
    }
    float* out; long long* cyc;
    hipMalloc(&out, 256); hipMallocManaged(&cyc, 8);
    const int iters = 4000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(rate<32>, dim3(1), dim3(64), 0, 0, out, iters, cyc); hipDeviceSynchronize();
        printf("16x16x32 f16: %.2f clock ticks per MFMA\n", (double)cyc[0] / (iters * 8));
        hipLaunchKernelGGL(rate<16>, dim3(1), dim3(64), 0, 0, out, iters, cyc); hipDeviceSynchronize();
        printf("16x16x16 f16: %.2f clock ticks per MFMA\n", (double)cyc[0] / (iters * 8));
    }
    return 0;
}
