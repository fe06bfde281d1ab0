// Generic (any radius, any extent) separable LoG passes: the slow, always-valid path.
//
// Used when the kernel radius exceeds MMX_MAX_RADIUS_FAST (sigma > 6 px) or when a block is
// thinner than the register-ring kernels need (extent < radius + prefetch).  Same math as
// mmx_colpass.hip / mmx_xpass.hip (scipy/ndimage/_filters.py:644-707 in float32), one thread
// per output voxel, taps read through L1/L2, weights from the kernarg segment (uniform index
// -> scalar loads).  Also the in-library cross-check for the fast kernels (tests).

#include "mmx_common.h"

struct mmx_taps_generic {
    float w0[MMX_MAX_RADIUS_GENERIC + 1];
    float w2[MMX_MAX_RADIUS_GENERIC + 1];
};

namespace {

__device__ __forceinline__ float load_any(const void* base, int dtype, int64_t idx)
{
    switch (dtype) {
        case MMX_U8:  return (float)((const uint8_t*)base)[idx];
        case MMX_U16: return (float)((const uint16_t*)base)[idx];
        case MMX_F32: return ((const float*)base)[idx];
        default:      return (float)((const double*)base)[idx];
    }
}

// idx runs over the pitched block [nz][ny][px]; returns 0 past the end, 1 for a voxel, 2 for a
// pitch column (skipped)
__device__ __forceinline__ int locate(const mmx_block& bd, int idx, int& z, int& y, int& x)
{
    const int plane = bd.ny * bd.px;
    if (idx >= bd.nz * plane) return 0;
    z = idx / plane;
    const int rem = idx - z * plane;
    y = rem / bd.px;
    x = rem - y * bd.px;
    return x < bd.nx ? 1 : 2;
}

__global__ void __launch_bounds__(MMX_WG)
gen_z(mmx_volume vol, const mmx_block* __restrict__ blocks, int64_t slot_elems, int R,
      float* __restrict__ gz, float* __restrict__ gzz, mmx_taps_generic t)
{
    const mmx_block bd = blocks[blockIdx.y];
    for (int idx = blockIdx.x * MMX_WG + threadIdx.x;; idx += gridDim.x * MMX_WG) {
        int z, y, x;
        const int where = locate(bd, idx, z, y, x);
        if (where == 0) return;
        if (where == 2) continue;
        const int64_t col = bd.src_off + (int64_t)y * vol.stride_y + (int64_t)x * vol.stride_x;
        const float c = load_any(vol.d_data, vol.dtype, col + (int64_t)z * vol.stride_z);
        float a0 = c * t.w0[0], a2 = c * t.w2[0];
        for (int k = 1; k <= R; ++k) {
            const float p = load_any(vol.d_data, vol.dtype, col + (int64_t)mmx_reflect(z - k, bd.nz) * vol.stride_z) +
                            load_any(vol.d_data, vol.dtype, col + (int64_t)mmx_reflect(z + k, bd.nz) * vol.stride_z);
            a0 = fmaf(p, t.w0[k], a0);
            a2 = fmaf(p, t.w2[k], a2);
        }
        const int64_t o = (int64_t)bd.slot * slot_elems + idx;
        gz[o] = a0;
        gzz[o] = a2;
    }
}

__global__ void __launch_bounds__(MMX_WG)
gen_y(const mmx_block* __restrict__ blocks, int64_t slot_elems, int R,
      const float* __restrict__ gz, const float* __restrict__ gzz,
      float* __restrict__ oa, float* __restrict__ obc, mmx_taps_generic t)
{
    const mmx_block bd = blocks[blockIdx.y];
    for (int idx = blockIdx.x * MMX_WG + threadIdx.x;; idx += gridDim.x * MMX_WG) {
        int z, y, x;
        const int where = locate(bd, idx, z, y, x);
        if (where == 0) return;
        if (where == 2) continue;
        const int64_t col = (int64_t)bd.slot * slot_elems + (int64_t)z * bd.ny * bd.px + x;
        const float c1 = gz[col + (int64_t)y * bd.px], c2 = gzz[col + (int64_t)y * bd.px];
        float a = c1 * t.w0[0];
        float bc = fmaf(c2, t.w0[0], c1 * t.w2[0]);
        for (int k = 1; k <= R; ++k) {
            const int64_t lo = col + (int64_t)mmx_reflect(y - k, bd.ny) * bd.px;
            const int64_t hi = col + (int64_t)mmx_reflect(y + k, bd.ny) * bd.px;
            const float p1 = gz[lo] + gz[hi];
            const float p2 = gzz[lo] + gzz[hi];
            a = fmaf(p1, t.w0[k], a);
            bc = fmaf(p1, t.w2[k], bc);
            bc = fmaf(p2, t.w0[k], bc);
        }
        const int64_t o = (int64_t)bd.slot * slot_elems + idx;
        oa[o] = a;
        obc[o] = bc;
    }
}

__global__ void __launch_bounds__(MMX_WG)
gen_x(const mmx_block* __restrict__ blocks, int64_t slot_elems, int R,
      const float* __restrict__ ga, const float* __restrict__ gbc, float* __restrict__ out,
      mmx_taps_generic t)
{
    const mmx_block bd = blocks[blockIdx.y];
    for (int idx = blockIdx.x * MMX_WG + threadIdx.x;; idx += gridDim.x * MMX_WG) {
        int z, y, x;
        const int where = locate(bd, idx, z, y, x);
        if (where == 0) return;
        if (where == 2) continue;
        const int64_t row = (int64_t)bd.slot * slot_elems + ((int64_t)z * bd.ny + y) * bd.px;
        float acc = ga[row + x] * t.w2[0];
        for (int k = 1; k <= R; ++k)
            acc = fmaf(ga[row + mmx_reflect(x - k, bd.nx)] + ga[row + mmx_reflect(x + k, bd.nx)], t.w2[k], acc);
        acc = fmaf(gbc[row + x], t.w0[0], acc);
        for (int k = 1; k <= R; ++k)
            acc = fmaf(gbc[row + mmx_reflect(x - k, bd.nx)] + gbc[row + mmx_reflect(x + k, bd.nx)], t.w0[k], acc);
        out[row + x] = acc;
    }
}

}  // namespace

// pass: 0 = z, 1 = y, 2 = x.  Weights already scaled by the caller.
int mmx_launch_generic_pass(int pass, const mmx_volume* vol, const mmx_block* d_blocks, int n_blocks,
                            int max_vox, int64_t slot_elems, const float* w0, const float* w2, int radius,
                            const float* in1, const float* in2, float* out1, float* out2, hipStream_t s)
{
    if (radius < 0 || radius > MMX_MAX_RADIUS_GENERIC) return MMX_ERR_UNSUPPORTED;
    mmx_taps_generic t;
    for (int k = 0; k <= MMX_MAX_RADIUS_GENERIC; ++k) {
        t.w0[k] = k <= radius ? w0[k] : 0.f;
        t.w2[k] = k <= radius ? w2[k] : 0.f;
    }
    int gx = (max_vox + MMX_WG - 1) / MMX_WG;
    if (gx > 8192) gx = 8192;
    if (gx < 1) gx = 1;
    dim3 grid(gx, n_blocks);
    if (pass == 0)
        hipLaunchKernelGGL(gen_z, grid, dim3(MMX_WG), 0, s, *vol, d_blocks, slot_elems, radius, out1, out2, t);
    else if (pass == 1)
        hipLaunchKernelGGL(gen_y, grid, dim3(MMX_WG), 0, s, d_blocks, slot_elems, radius, in1, in2, out1, out2, t);
    else
        hipLaunchKernelGGL(gen_x, grid, dim3(MMX_WG), 0, s, d_blocks, slot_elems, radius, in1, in2, out1, t);
    return hipGetLastError() == hipSuccess ? MMX_OK : MMX_ERR_HIP;
}
