// X pass of the separable LoG (the contiguous axis):  LoG = G''(x) A + G(x) BC.
//
// Replaces, in float32, the axis-2 `correlate1d` passes of
// scipy.ndimage.gaussian_laplace (scipy/ndimage/_filters.py:644-707 -> :126-182) and the
// `-LoG * sigma**2` scale normalisation of skimage.feature.blob_log
// (skimage/feature/blob.py:501-502; the factor is folded into the weights by the caller).
//
// Design (gfx950): along x the taps live in neighbouring lanes, so rows are staged in LDS.
// A workgroup takes RG whole rows (block rows are at most a few hundred voxels, so the
// scipy "reflect" halo is applied while staging and no inter-tile halo exists).  Rows are
// 128-byte aligned (pitch px), so a row group is one contiguous run read with 16-byte
// loads; the NEXT group's loads are issued right after the barrier and stay in flight
// while the current group is computed (software prefetch through registers).  Each thread
// produces T = 8 consecutive outputs from a register window of T + 2R staged values (fully
// unrolled, SGPR weights, ds_read_b64 on a 2-in-8 padded layout = conflict free).  Results
// go back through LDS so that the global stores are whole aligned rows (16 B per lane).
//
// Algorithmic HBM bytes per voxel: read 8 (A, BC) + write 4.

#include "mmx_common.h"

namespace {

constexpr int kT = 8;        // outputs per thread
constexpr int kMaxQuads = 2; // 16-byte loads per thread, array and row group (host sizes RG to fit)
constexpr int kMaxHalo = 2;  // halo loads per thread, array and row group

// LDS row layout: two pad floats after every eight, so that threads reading 8-float-strided
// windows with ds_read_b64 hit 32 distinct bank pairs.
__device__ __forceinline__ int pad2(int i) { return i + 2 * (i >> 3); }

template <int R>
__global__ void __launch_bounds__(MMX_WG)
xpass_kernel(const mmx_block* __restrict__ blocks, int64_t slot_elems, int rg_max,
             const float* __restrict__ ga, const float* __restrict__ gbc,
             float* __restrict__ out, mmx_taps_f32 taps)
{
    constexpr int LEAD = R & 1;                   // odd radius: window starts one float early (8-B aligned)
    constexpr int S = (R + LEAD + 7) & ~7;        // staged position of x = 0
    constexpr int WIN = kT + 2 * R + 2 * LEAD;    // floats read per thread and array (even: read in pairs)
    extern __shared__ float lds[];
    const mmx_block bd = blocks[blockIdx.y];
    const int W = bd.nx, px = bd.px;
    const int rows = bd.nz * bd.ny;
    const int CH = (W + kT - 1) / kT;             // chunks per row
    int RG = MMX_WG / CH;                         // rows per group
    if (RG < 1) RG = 1;
    if (RG > rg_max) RG = rg_max;
    const int PW = (pad2(S + px + R + LEAD) + 3) & ~1;   // LDS row pitch (floats, even)
    float* la = lds;
    float* lb = la + rg_max * PW;
    float* lc = lb + rg_max * PW;                 // results, plain [RG][px] (16-B aligned rows)
    const int64_t sbase = (int64_t)bd.slot * slot_elems;
    const int n_groups = (rows + RG - 1) / RG;
    const int t = threadIdx.x;

    float4 qa[kMaxQuads], qb[kMaxQuads];
    float ha[kMaxHalo], hb[kMaxHalo];

    auto prefetch = [&](int g) {
        const int row0 = g * RG;
        const int nr = min(RG, rows - row0);
        const float4* pa = reinterpret_cast<const float4*>(ga + sbase + (int64_t)row0 * px);
        const float4* pb = reinterpret_cast<const float4*>(gbc + sbase + (int64_t)row0 * px);
        const int nq = nr * (px / 4);
#pragma unroll
        for (int i = 0; i < kMaxQuads; ++i) {
            const int f = t + i * MMX_WG;
            if (f < nq) { qa[i] = pa[f]; qb[i] = pb[f]; }
        }
#pragma unroll
        for (int i = 0; i < kMaxHalo; ++i) {
            const int h = t + i * MMX_WG;
            if (h < nr * 2 * R) {
                const int r = h / (2 * R), k = h - r * 2 * R;
                const int x = k < R ? k : W - 1 - (k - R);             // scipy "reflect": x = -1-k -> k
                ha[i] = ga[sbase + (int64_t)(row0 + r) * px + x];
                hb[i] = gbc[sbase + (int64_t)(row0 + r) * px + x];
            }
        }
    };
    auto stage = [&](int g) {
        const int row0 = g * RG;
        const int nr = min(RG, rows - row0);
        const int nq = nr * (px / 4);
        const int qrow = px / 4;
#pragma unroll
        for (int i = 0; i < kMaxQuads; ++i) {
            const int f = t + i * MMX_WG;
            if (f < nq) {
                const int r = f / qrow;
                const int x = (f - r * qrow) * 4;
                float* da = la + r * PW;
                float* db = lb + r * PW;
                const int p0 = S + x;            // multiple of 4: the quad stays inside one 8-group
                const int q0 = pad2(p0);
                // pitch columns (x >= W) are not staged: those positions belong to the halo
                const float va[4] = {qa[i].x, qa[i].y, qa[i].z, qa[i].w};
                const float vb[4] = {qb[i].x, qb[i].y, qb[i].z, qb[i].w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (x + j < W) { da[q0 + j] = va[j]; db[q0 + j] = vb[j]; }
            }
        }
#pragma unroll
        for (int i = 0; i < kMaxHalo; ++i) {
            const int h = t + i * MMX_WG;
            if (h < nr * 2 * R) {
                const int r = h / (2 * R), k = h - r * 2 * R;
                const int p = k < R ? S - 1 - k : S + W + (k - R);
                la[r * PW + pad2(p)] = ha[i];
                lb[r * PW + pad2(p)] = hb[i];
            }
        }
    };

    int g = blockIdx.x;
    if (g < n_groups) prefetch(g);
    for (; g < n_groups; g += gridDim.x) {
        const int row0 = g * RG;
        const int nr = min(RG, rows - row0);
        stage(g);
        __syncthreads();
        if (g + (int)gridDim.x < n_groups) prefetch(g + gridDim.x);   // in flight during the compute
        const int r = t / CH;
        const int c = t - r * CH;
        if (r < nr) {
            float res[kT];
            float win[WIN];
            const int base = S - R - LEAD + c * kT;                   // even
            const float* pa = la + r * PW;
#pragma unroll
            for (int i = 0; i < WIN; i += 2) {
                const float2 v = *reinterpret_cast<const float2*>(pa + pad2(base + i));
                win[i] = v.x;
                win[i + 1] = v.y;
            }
#pragma unroll
            for (int o = 0; o < kT; ++o) {
                float acc = win[LEAD + o + R] * taps.w2[0];
#pragma unroll
                for (int k = 1; k <= R; ++k)
                    acc = fmaf(win[LEAD + o + R - k] + win[LEAD + o + R + k], taps.w2[k], acc);
                res[o] = acc;
            }
            const float* pb = lb + r * PW;
#pragma unroll
            for (int i = 0; i < WIN; i += 2) {
                const float2 v = *reinterpret_cast<const float2*>(pb + pad2(base + i));
                win[i] = v.x;
                win[i + 1] = v.y;
            }
#pragma unroll
            for (int o = 0; o < kT; ++o) {
                float acc = fmaf(win[LEAD + o + R], taps.w0[0], res[o]);
#pragma unroll
                for (int k = 1; k <= R; ++k)
                    acc = fmaf(win[LEAD + o + R - k] + win[LEAD + o + R + k], taps.w0[k], acc);
                res[o] = acc;
            }
            float4* dst = reinterpret_cast<float4*>(lc + r * px + c * kT);
            dst[0] = make_float4(res[0], res[1], res[2], res[3]);
            dst[1] = make_float4(res[4], res[5], res[6], res[7]);
        }
        __syncthreads();
        // whole, 128-byte aligned rows back to HBM (pitch columns get whatever the last chunk
        // computed from the halo: never read as data)
        {
            float4* po = reinterpret_cast<float4*>(out + sbase + (int64_t)row0 * px);
            const float4* src = reinterpret_cast<const float4*>(lc);
            const int nq = nr * (px / 4);
            for (int f = t; f < nq; f += MMX_WG) po[f] = src[f];
        }
    }
}

template <int R>
int launch_x(const mmx_block* d_blocks, int n_blocks, int max_rows, int max_nx, int64_t slot_elems,
             const mmx_taps_f32& taps, const float* d_a, const float* d_bc, float* d_log, hipStream_t s)
{
    constexpr int LEAD = R & 1;
    constexpr int S = (R + LEAD + 7) & ~7;
    const int px = (max_nx + MMX_ROW_ALIGN - 1) / MMX_ROW_ALIGN * MMX_ROW_ALIGN;
    const int span = S + px + R + LEAD;
    const int PW = ((span + 2 * (span >> 3)) + 3) & ~1;
    const int CH = (max_nx + kT - 1) / kT;
    // rows per group: as many as the threads cover, within the per-thread prefetch registers
    int rg = MMX_WG / CH;
    if (rg < 1) rg = 1;
    while (rg > 1 && (rg * (px / 4) > kMaxQuads * MMX_WG || rg * 2 * R > kMaxHalo * MMX_WG)) --rg;
    if (px / 4 > kMaxQuads * MMX_WG || 2 * R > kMaxHalo * MMX_WG) return MMX_ERR_UNSUPPORTED;
    if (CH > MMX_WG) return MMX_ERR_UNSUPPORTED;   // rows wider than 2048 voxels: generic path
    // narrower blocks of the same batch may pack more rows per group; keep them within rg
    const size_t lds_bytes = ((size_t)2 * rg * PW + (size_t)rg * px) * sizeof(float);
    if (lds_bytes > 64 * 1024) return MMX_ERR_UNSUPPORTED;
    int gx = (max_rows + rg - 1) / rg;
    if (gx > 2048) gx = 2048;
    if (gx < 1) gx = 1;
    dim3 grid(gx, n_blocks);
    hipLaunchKernelGGL(xpass_kernel<R>, grid, dim3(MMX_WG), lds_bytes, s, d_blocks, slot_elems, rg,
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
