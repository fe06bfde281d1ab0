// X pass of the separable LoG (the contiguous axis):  LoG = G''(x) A + G(x) BC.
//
// Replaces, in float32, the axis-2 `correlate1d` passes of
// scipy.ndimage.gaussian_laplace (scipy/ndimage/_filters.py:644-707 -> :126-182) and the
// `-LoG * sigma**2` scale normalisation of skimage.feature.blob_log
// (skimage/feature/blob.py:501-502; the factor is folded into the weights by the caller).
//
// Design (gfx950): along x the taps live in neighbouring lanes, so rows are staged in LDS.
// A workgroup takes RG whole rows (block rows are at most a few hundred voxels, so the
// scipy "reflect" halo is applied while staging and no inter-tile halo exists), each
// thread then produces T = 8 consecutive outputs from a register window of T + 2R
// staged values (fully unrolled, SGPR weights).  LDS rows are padded by one float every
// eight so that lanes reading at a stride of eight floats hit distinct banks.  Results
// go back through LDS so that global stores are whole coalesced row segments.
//
// Algorithmic HBM bytes per voxel: read 8 (A, BC) + write 4.

#include "mmx_common.h"

namespace {

constexpr int kT = 8;  // outputs per thread

__device__ __forceinline__ int pad8(int i) { return i + (i >> 3); }

template <int R>
__global__ void __launch_bounds__(MMX_WG)
xpass_kernel(const mmx_block* __restrict__ blocks, int64_t slot_elems, int rg_max,
             const float* __restrict__ ga, const float* __restrict__ gbc,
             float* __restrict__ out, mmx_taps_f32 taps)
{
    extern __shared__ float lds[];
    const mmx_block bd = blocks[blockIdx.y];
    const int W = bd.nx;
    const int rows = bd.nz * bd.ny;
    const int HW = W + 2 * R;          // staged row length (with halo)
    const int PW = pad8(HW) + 1;       // padded LDS row pitch
    const int CH = (W + kT - 1) / kT;  // chunks per row
    int RG = MMX_WG / CH;              // rows per group
    if (RG < 1) RG = 1;
    if (RG > rg_max) RG = rg_max;
    float* la = lds;
    float* lb = lds + RG * PW;
    const int64_t sbase = (int64_t)bd.slot * slot_elems;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n_groups = (rows + RG - 1) / RG;

    for (int g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const int row0 = g * RG;
        const int nr = min(RG, rows - row0);
        // ---- stage A and BC rows with the reflect halo
        for (int r = wave; r < nr; r += MMX_WG / 64) {
            const int64_t gro = sbase + (int64_t)(row0 + r) * bd.px;
            for (int xx = lane; xx < HW; xx += 64) {
                int x = xx - R;
                x = x < 0 ? -1 - x : x;
                x = x >= W ? 2 * W - 1 - x : x;
                x = min(max(x, 0), W - 1);  // rows thinner than R: routed to the generic path by the host
                la[r * PW + pad8(xx)] = ga[gro + x];
                lb[r * PW + pad8(xx)] = gbc[gro + x];
            }
        }
        __syncthreads();
        // ---- compute kT outputs per work item
        float res[kT];
        int my_r = -1, my_c = 0;
        for (int item = threadIdx.x; item < nr * CH; item += MMX_WG) {
            const int r = item / CH;
            const int c = item - r * CH;
            my_r = r;
            my_c = c;
            const float* pa = la + r * PW;
            const float* pb = lb + r * PW;
            float win[kT + 2 * R];
#pragma unroll
            for (int i = 0; i < kT + 2 * R; ++i) win[i] = pa[pad8(c * kT + i)];
#pragma unroll
            for (int o = 0; o < kT; ++o) {
                float acc = win[o + R] * taps.w2[0];
#pragma unroll
                for (int k = 1; k <= R; ++k) acc = fmaf(win[o + R - k] + win[o + R + k], taps.w2[k], acc);
                res[o] = acc;
            }
#pragma unroll
            for (int i = 0; i < kT + 2 * R; ++i) win[i] = pb[pad8(c * kT + i)];
#pragma unroll
            for (int o = 0; o < kT; ++o) {
                float acc = res[o];
                acc = fmaf(win[o + R], taps.w0[0], acc);
#pragma unroll
                for (int k = 1; k <= R; ++k) acc = fmaf(win[o + R - k] + win[o + R + k], taps.w0[k], acc);
                res[o] = acc;
            }
            if (nr * CH > MMX_WG) {
                // more work items than threads (very wide rows): store this item's outputs directly
                const int64_t gro = sbase + (int64_t)(row0 + r) * bd.px + c * kT;
#pragma unroll
                for (int o = 0; o < kT; ++o)
                    if (c * kT + o < W) out[gro + o] = res[o];
            }
        }
        if (nr * CH <= MMX_WG) {
            // ---- results back through LDS (reusing the A tile) for coalesced row stores
            __syncthreads();
            if (my_r >= 0) {
#pragma unroll
                for (int o = 0; o < kT; ++o) la[my_r * PW + pad8(my_c * kT + o)] = res[o];
            }
            __syncthreads();
            for (int r = wave; r < nr; r += MMX_WG / 64) {
                const int64_t gro = sbase + (int64_t)(row0 + r) * bd.px;
                for (int x = lane; x < bd.px; x += 64) out[gro + x] = x < W ? la[r * PW + pad8(x)] : 0.f;
            }
        }
        __syncthreads();
    }
}

template <int R>
int launch_x(const mmx_block* d_blocks, int n_blocks, int max_rows, int max_nx, int64_t slot_elems,
             const mmx_taps_f32& taps, const float* d_a, const float* d_bc, float* d_log, hipStream_t s)
{
    const int HW = max_nx + 2 * R;
    const int PW = HW + (HW >> 3) + 1;
    const int CH = (max_nx + kT - 1) / kT;
    int RG = MMX_WG / CH;
    if (RG < 1) RG = 1;
    // rows with fewer chunks (narrow edge blocks) would pack more rows per group: cap by LDS
    const int lds_budget = 48 * 1024;
    int rg_max = lds_budget / (2 * PW * (int)sizeof(float));
    if (rg_max < 1) return MMX_ERR_UNSUPPORTED;
    if (rg_max > MMX_WG) rg_max = MMX_WG;
    const size_t lds_bytes = (size_t)2 * rg_max * PW * sizeof(float);
    (void)RG;
    int groups = (max_rows + 0) / 1;  // upper bound on row groups: at least one row per group
    int gx = groups < 2048 ? groups : 2048;
    if (gx < 1) gx = 1;
    dim3 grid(gx, n_blocks);
    hipLaunchKernelGGL(xpass_kernel<R>, grid, dim3(MMX_WG), lds_bytes, s, d_blocks, slot_elems, rg_max,
                       d_a, d_bc, d_log, taps);
    return hipGetLastError() == hipSuccess ? MMX_OK : MMX_ERR_HIP;
}

}  // namespace

#define MMX_FOR_EACH_RADIUS(X) \
    X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) \
    X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24)

int mmx_launch_xpass(const mmx_block* d_blocks, int n_blocks, int max_rows, int max_nx,
                     int64_t slot_elems, const mmx_taps_f32& taps, int radius,
                     const float* d_a, const float* d_bc, float* d_log, hipStream_t stream)
{
    switch (radius) {
#define X(R) case R: return launch_x<R>(d_blocks, n_blocks, max_rows, max_nx, slot_elems, taps, d_a, d_bc, d_log, stream);
        MMX_FOR_EACH_RADIUS(X)
#undef X
        default: return MMX_ERR_UNSUPPORTED;
    }
}
