// Does v_mfma_f32_16x16x32_f16 keep float16 SUBNORMAL inputs?  A[m][k] = bits 0x00bb (b x 2^-24), B = 1024.0:
// exact answer 32 b 2^-14.  Prints the result for a few b.   hipcc --offload-arch=gfx950 -O2 mfma_denorm.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned bits, float* out)
{
    const unsigned d = bits | (bits << 16);
    const u4 a = {d, d, d, d};
    const u4 b = {0x64006400u, 0x64006400u, 0x64006400u, 0x64006400u};   // 1024.0
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
    // and as the B operand
    f4 e = {0.f, 0.f, 0.f, 0.f};
    e = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, b), __builtin_bit_cast(h8, a), e, 0, 0, 0);
    if (threadIdx.x == 0) out[1] = e[0];
}
int main()
{
    float* d; hipMalloc(&d, 8);
    int bad = 0;
    for (unsigned b : {1u, 3u, 37u, 128u, 255u, 0x3ffu}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, b, d);
        float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        const float want = 32.f * b * 1024.f / 16777216.f;
        printf("b=%4u  A-side %.9g  B-side %.9g  want %.9g %s\n", b, h[0], h[1], want, (h[0] == want && h[1] == want) ? "ok" : "FLUSHED/WRONG");
        bad += !(h[0] == want && h[1] == want);
    }
    printf(bad ? "subnormal inputs NOT kept\n" : "subnormal inputs kept exactly\n");
    return 0;
}
