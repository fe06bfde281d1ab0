// Blob-table kernels: sphere-overlap pairs (A5) and cross-block close pairs (A13).
//
// mmx_overlap_pairs  -- skimage/feature/blob.py:84-143 (_blob_overlap, 3-D branch :55-81)
//   evaluated for every pair of one block's blobs; pairs whose overlap fraction exceeds
//   `overlap - band` are returned (those within `band` of the limit are re-evaluated on the
//   host with the reference's exact libm calls).  The *sequential* part of _prune_blobs
//   (blob.py:172-186: zero the smaller sigma pair by pair) stays on the host: it is
//   order-dependent and touches a handful of pairs.
// mmx_close_pairs    -- magmap/cv/detector.py:1000-1006 (_find_close_blobs): integer
//   |dz|<=tz && |dy|<=ty && |dx|<=tx all-pairs test between a master and a check table;
//   returns for each master row the LAST matching check row (NumPy fancy-assignment
//   last-write-wins, detector.py:1077-1083) and a hit flag per check row (deleted rows,
//   detector.py:1069).
// Built with -ffp-contract=off so the float64 expressions round like the Python ones.

#include "mmx_common.h"

namespace {

__global__ void __launch_bounds__(MMX_WG)
overlap_pairs_kernel(const double* __restrict__ blobs, const int32_t* __restrict__ offsets,
                     double overlap, double band, double max_sigma, int32_t* __restrict__ pairs,
                     double* __restrict__ frac, uint32_t cap, uint32_t* __restrict__ count)
{
    const int b0 = offsets[blockIdx.y], b1 = offsets[blockIdx.y + 1];
    const int n = b1 - b0;
    const double root3 = sqrt(3.0);
    const double kPi = 3.141592653589793;  // math.pi
    for (int i = blockIdx.x * MMX_WG + threadIdx.x; i < n; i += gridDim.x * MMX_WG) {
        const double* bi = blobs + (int64_t)(b0 + i) * 4;
        const double zi = bi[0], yi = bi[1], xi = bi[2], si = bi[3];
        // cheap cut first: no overlap beyond sqrt(3) (s_i + s_j) <= 2 sqrt(3) s_max on any one axis
        const float cut = (float)(2.0 * root3 * max_sigma) + 1.0f;
        const float fz = (float)zi, fy = (float)yi, fx = (float)xi;
        for (int j = i + 1; j < n; ++j) {
            const double* bj = blobs + (int64_t)(b0 + j) * 4;
            if (fabsf((float)bj[0] - fz) > cut || fabsf((float)bj[1] - fy) > cut ||
                fabsf((float)bj[2] - fx) > cut)
                continue;
            const double sj = bj[3];
            if (si == 0.0 && sj == 0.0) continue;
            double r1, r2, ms;
            if (si > sj) { ms = si; r1 = 1.0; r2 = sj / si; }
            else         { ms = sj; r2 = 1.0; r1 = si / sj; }
            const double den = ms * root3;
            const double d0 = bj[0] / den - zi / den;
            const double d1 = bj[1] / den - yi / den;
            const double d2 = bj[2] / den - xi / den;
            const double d = sqrt((d0 * d0 + d1 * d1) + d2 * d2);
            if (d > r1 + r2) continue;
            double f;
            if (d <= fabs(r1 - r2)) {
                f = 1.0;
            } else {
                const double rs = r1 + r2;
                const double t = rs - d;
                const double vol = kPi / (12 * d) * (t * t) *
                                   (d * d + 2 * d * rs - 3 * (r1 * r1 + r2 * r2) + 6 * r1 * r2);
                const double rm = r1 < r2 ? r1 : r2;
                f = vol / (4. / 3 * kPi * (rm * rm * rm));
            }
            if (f > overlap - band) {
                const uint32_t pos = atomicAdd(count, 1u);
                if (pos < cap) {
                    pairs[2 * (int64_t)pos] = b0 + i;
                    pairs[2 * (int64_t)pos + 1] = b0 + j;
                    frac[pos] = f;
                }
            }
        }
    }
}

__global__ void __launch_bounds__(MMX_WG)
close_pairs_kernel(const int32_t* __restrict__ master, int n_master,
                   const int32_t* __restrict__ check, int n_check, int tz, int ty, int tx,
                   int32_t* __restrict__ last, uint8_t* __restrict__ hit)
{
    __shared__ int32_t tile[MMX_WG * 3];
    const int m = blockIdx.x * MMX_WG + threadIdx.x;
    int mz = 0, my = 0, mx = 0;
    if (m < n_master) { mz = master[3 * m]; my = master[3 * m + 1]; mx = master[3 * m + 2]; }
    int best = -1;
    for (int c0 = 0; c0 < n_check; c0 += MMX_WG) {
        const int nc = min(MMX_WG, n_check - c0);
        __syncthreads();
        for (int t = threadIdx.x; t < nc * 3; t += MMX_WG) tile[t] = check[3 * (int64_t)c0 + t];
        __syncthreads();
        if (m < n_master) {
            for (int c = 0; c < nc; ++c) {
                const int dz = abs(mz - tile[3 * c]);
                const int dy = abs(my - tile[3 * c + 1]);
                const int dx = abs(mx - tile[3 * c + 2]);
                if (dz <= tz && dy <= ty && dx <= tx) {
                    best = c0 + c;  // ascending c: the last match survives
                    hit[c0 + c] = 1;
                }
            }
        }
    }
    if (m < n_master) last[m] = best;
}

}  // namespace

extern "C" int mmx_overlap_pairs(const double* d_blobs, const int32_t* d_offsets, int n_blocks,
                                 double overlap, double band, double max_sigma, int32_t* d_pairs, double* d_frac,
                                 uint32_t cap, uint32_t* d_count, void* stream)
{
    if (!d_blobs || !d_offsets || !d_pairs || !d_frac || !d_count || n_blocks < 1 || !(max_sigma > 0)) return MMX_ERR_ARG;
    dim3 grid(8, n_blocks);
    mmx_timed_scope ts(MMX_K_PAIRS, (hipStream_t)stream);
    hipLaunchKernelGGL(overlap_pairs_kernel, grid, dim3(MMX_WG), 0, (hipStream_t)stream, d_blobs, d_offsets,
                       overlap, band, max_sigma, d_pairs, d_frac, cap, d_count);
    return hipGetLastError() == hipSuccess ? MMX_OK : MMX_ERR_HIP;
}

extern "C" int mmx_close_pairs(const int32_t* d_master, int n_master, const int32_t* d_check, int n_check,
                               const int32_t tol[3], int32_t* d_last, uint8_t* d_hit, void* stream)
{
    if (!d_master || !d_check || !tol || !d_last || !d_hit || n_master < 0 || n_check < 0) return MMX_ERR_ARG;
    if (n_master == 0) return MMX_OK;
    hipStream_t s = (hipStream_t)stream;
    if (n_check > 0) {
        hipError_t e = hipMemsetAsync(d_hit, 0, (size_t)n_check, s);
        if (e != hipSuccess) return MMX_ERR_HIP;
    }
    dim3 grid((n_master + MMX_WG - 1) / MMX_WG);
    mmx_timed_scope ts(MMX_K_CLOSE, s);
    hipLaunchKernelGGL(close_pairs_kernel, grid, dim3(MMX_WG), 0, s, d_master, n_master, d_check, n_check,
                       tol[0], tol[1], tol[2], d_last, d_hit);
    return hipGetLastError() == hipSuccess ? MMX_OK : MMX_ERR_HIP;
}

// ---- PMC calibration kernels (tools/pmc_calib.py): streams of a known byte count with the
// access shapes the LoG kernels use, to turn rocprofv3 FETCH_SIZE / WRITE_SIZE into bytes.
namespace {
__global__ void __launch_bounds__(MMX_WG) calib_copy_b32(const float* __restrict__ in, float* __restrict__ out, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * MMX_WG + threadIdx.x; i < n; i += (int64_t)gridDim.x * MMX_WG)
        out[i] = in[i] + 1.0f;
}
__global__ void __launch_bounds__(MMX_WG) calib_copy_b128(const float4* __restrict__ in, float4* __restrict__ out, int64_t n4)
{
    for (int64_t i = (int64_t)blockIdx.x * MMX_WG + threadIdx.x; i < n4; i += (int64_t)gridDim.x * MMX_WG) {
        float4 v = in[i];
        v.x += 1.0f;
        out[i] = v;
    }
}
__global__ void __launch_bounds__(MMX_WG) calib_read_u16(const uint16_t* __restrict__ in, float* __restrict__ out, int64_t n)
{
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * MMX_WG + threadIdx.x; i < n; i += (int64_t)gridDim.x * MMX_WG)
        acc += (float)in[i];
    if (acc == 12345.678f) out[0] = acc;   // keep the loads alive, (almost) never store
}
}  // namespace

extern "C" int mmx_calib_stream(int kind, const void* d_in, void* d_out, int64_t n_elems, void* stream)
{
    if (!d_in || !d_out || n_elems < 1) return MMX_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(256 * 8);
    if (kind == 0) hipLaunchKernelGGL(calib_copy_b32, grid, dim3(MMX_WG), 0, s, (const float*)d_in, (float*)d_out, n_elems);
    else if (kind == 1) hipLaunchKernelGGL(calib_copy_b128, grid, dim3(MMX_WG), 0, s, (const float4*)d_in, (float4*)d_out, n_elems / 4);
    else if (kind == 2) hipLaunchKernelGGL(calib_read_u16, grid, dim3(MMX_WG), 0, s, (const uint16_t*)d_in, (float*)d_out, n_elems);
    else return MMX_ERR_ARG;
    return hipGetLastError() == hipSuccess ? MMX_OK : MMX_ERR_HIP;
}
