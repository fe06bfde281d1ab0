// Micro-benchmark: v_fma_f32 vs v_pk_fma_f32 issue rate on gfx950 (tools/exp, not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void k_scalar(float* out, float w, int iters)
{
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = fmaf(a[i], w, 0.5f);
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_packed(float* out, float w, int iters)
{
    v2f a[4];
    for (int i = 0; i < 4; ++i) a[i] = (v2f){threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
    const v2f ww = {w, w}, hh = {0.5f, 0.5f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = __builtin_elementwise_fma(a[i], ww, hh);
    }
    float s = 0; for (int i = 0; i < 4; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    float* d; hipMalloc(&d, 256 * 16 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int rep = 0; rep < 2; ++rep) {
        float ms;
        hipEventRecord(e0); hipLaunchKernelGGL(k_scalar, dim3(256 * 16), dim3(256), 0, 0, d, 0.999f, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        double fl = 2.0 * 8 * iters * 256.0 * 16 * 256;
        printf("scalar v_fma_f32   : %.2f ms  %.1f TFLOP/s\n", ms, fl / ms / 1e9);
        hipEventRecord(e0); hipLaunchKernelGGL(k_packed, dim3(256 * 16), dim3(256), 0, 0, d, 0.999f, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("packed v_pk_fma_f32: %.2f ms  %.1f TFLOP/s\n", ms, fl / ms / 1e9);
    }
    return 0;
}
