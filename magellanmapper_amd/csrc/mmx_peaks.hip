// Scale-space local maxima (A4): 3x3x3x3 non-max suppression over (z, y, x, sigma).
//
// Replaces skimage.feature.peak_local_max as blob_log calls it
// (skimage/feature/blob.py:506-512 -> skimage/feature/peak.py:28-50, 114-319):
//   mask = (cube == maximum_filter(cube, footprint=ones(3,3,3,3), mode='constant'))
//          & (cube > threshold)
// i.e. a voxel is a peak when it is >= its 80 neighbours (0 outside the cube, also past
// the first / last sigma) and strictly above the threshold.
//
// The float32 cube cannot decide exact ties, so this kernel emits CANDIDATES: voxels with
// v >= nbr_max - eps and v > thr - eps, flagged "contested" unless they win by more than
// eps on both tests.  Exact float64 values (mmx_rescore.hip) settle the rest on the host.
//
// Design (gfx950): one coalesced read of each sigma plane per voxel (the algorithmic
// 4 B/voxel/sigma); almost every voxel fails `v > thr - eps` and stops there.  The 80
// neighbour reads happen only inside blobs and come from L1/L2.  Hits are appended with
// a wave-aggregated atomic counter.

#include "mmx_common.h"

namespace {

__global__ void __launch_bounds__(MMX_WG)
peaks_kernel(const float* __restrict__ log, int ns, int64_t sigma_stride,
             const mmx_block* __restrict__ blocks, int64_t slot_elems, float thr, float eps,
             mmx_cand* __restrict__ out, uint32_t cap, uint32_t* __restrict__ count)
{
    const mmx_block bd = blocks[blockIdx.y];
    const int nz = bd.nz, ny = bd.ny, nx = bd.nx, px = bd.px;
    const int plane = ny * px;
    const int nvox = nz * plane;          // pitch columns are skipped below
    const float* base = log + (int64_t)bd.slot * slot_elems;
    const float lo = thr - eps;

    for (int idx = blockIdx.x * MMX_WG + threadIdx.x; idx < nvox; idx += gridDim.x * MMX_WG) {
        bool located = false;
        int z = 0, y = 0, x = 0;
        if (px != nx && (idx % px) >= nx) continue;
        for (int s = 0; s < ns; ++s) {
            const float v = base[(int64_t)s * sigma_stride + idx];
            if (!(v > lo)) continue;
            if (!located) {
                z = idx / plane;
                const int rem = idx - z * plane;
                y = rem / px;
                x = rem - y * px;
                located = true;
            }
            const float reject = v + eps;  // a neighbour above this rules the voxel out
            float m = -INFINITY;
            bool border = false, dead = false;
            // same-sigma face neighbours first: they reject nearly every non-peak
            const float* ps = base + (int64_t)s * sigma_stride;
            if (x > 0) m = fmaxf(m, ps[idx - 1]); else border = true;
            if (x + 1 < nx) m = fmaxf(m, ps[idx + 1]); else border = true;
            if (y > 0) m = fmaxf(m, ps[idx - px]); else border = true;
            if (y + 1 < ny) m = fmaxf(m, ps[idx + px]); else border = true;
            if (z > 0) m = fmaxf(m, ps[idx - plane]); else border = true;
            if (z + 1 < nz) m = fmaxf(m, ps[idx + plane]); else border = true;
            if (m > reject) continue;
            for (int ds = -1; ds <= 1 && !dead; ++ds) {
                const int ss = s + ds;
                if (ss < 0 || ss >= ns) { border = true; continue; }
                const float* pss = base + (int64_t)ss * sigma_stride;
                for (int dz = -1; dz <= 1 && !dead; ++dz) {
                    const int zz = z + dz;
                    if (zz < 0 || zz >= nz) continue;  // border already noted above
                    for (int dy = -1; dy <= 1; ++dy) {
                        const int yy = y + dy;
                        if (yy < 0 || yy >= ny) continue;
                        const int row = zz * plane + yy * px;
#pragma unroll
                        for (int dx = -1; dx <= 1; ++dx) {
                            const int xx = x + dx;
                            if (xx < 0 || xx >= nx) continue;
                            if ((ds | dz | dy | dx) == 0) continue;
                            m = fmaxf(m, pss[row + xx]);
                        }
                    }
                    if (m > reject) dead = true;
                }
            }
            if (dead) continue;
            if (border) m = fmaxf(m, 0.0f);  // mode='constant', cval = 0
            if (!(v >= m - eps)) continue;
            const bool contested = !(v > m + eps) || !(v > thr + eps);
            const uint32_t pos = atomicAdd(count, 1u);
            if (pos < cap) {
                mmx_cand c;
                c.slot = bd.slot;
                c.s = s;
                c.z = z;
                c.y = y;
                c.x = x;
                c.flags = contested ? MMX_CAND_CONTESTED : 0u;
                c.v = v;
                c.nbr_max = m;
                c.v64 = __longlong_as_double(0x7ff8000000000000LL);
                c._reserved = 0.0;
                out[pos] = c;
            }
        }
    }
}

}  // namespace

int mmx_launch_peaks(const float* d_log, int n_sigma, int64_t sigma_stride, const mmx_block* d_blocks,
                     int n_blocks, int max_vox, int64_t slot_elems, float thr, float eps,
                     mmx_cand* d_cands, uint32_t cap, uint32_t* d_count, hipStream_t stream)
{
    int gx = (max_vox + MMX_WG - 1) / MMX_WG;
    if (gx > 4096) gx = 4096;
    if (gx < 1) gx = 1;
    dim3 grid(gx, n_blocks);
    hipLaunchKernelGGL(peaks_kernel, grid, dim3(MMX_WG), 0, stream, d_log, n_sigma, sigma_stride,
                       d_blocks, slot_elems, thr, eps, d_cands, cap, d_count);
    return hipGetLastError() == hipSuccess ? MMX_OK : MMX_ERR_HIP;
}
