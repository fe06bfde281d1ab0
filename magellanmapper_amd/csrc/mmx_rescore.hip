// Exact float64 value of the scale-space cube at given points.
//
// skimage.feature.peak_local_max compares float64 cube values for exact equality and for
// `> threshold` (skimage/feature/peak.py:28-50); the float32 pipeline cannot settle ties or
// the descending-response order that feeds the overlap prune (peak.py:9-25,
// blob.py:146-187).  This kernel recomputes, for a list of (block, sigma, z, y, x) points,
//     -gaussian_laplace(img_as_float(block), sigma)[z, y, x] * mean(sigma)**2
// bit for bit as SciPy does (scipy/ndimage/_filters.py:644-707; C: NI_Correlate1D,
// symmetric branch):  per 1-D pass
//     acc = in[c] * w[0];  for k = R .. 1:  acc += (in[c-k] + in[c+k]) * w[k]
// in double, NO fused multiply-add (this file is built with -ffp-contract=off), results
// rounded to the array dtype after every pass (float64, or float32 when the image is
// float32), three terms (second derivative on axis 0, 1, 2) summed in that order.
// oracle/ndfilters.c states the same arithmetic on the CPU and is pinned to SciPy bit for
// bit; tests/test_gpu_parity.py checks this kernel against it.
//
// Design (gfx950): a 256-thread workgroup per point, a fixed number of workgroups walking the table.  The point needs a (2R+1)^2 window
// of axis-0 sums (two kernels sharing their loads), staged in LDS in column strips, then
// 3(2R+1) axis-1 sums, then 3 axis-2 sums.  Loads are L2/MALL hits (the block's voxels were
// just streamed by the float32 passes); the work is ~1e5 float64 operations per point and
// there are ~1e3 points per block, so this costs a few per cent of the float32 passes.

#include <algorithm>
#include <type_traits>

#include "mmx_common.h"

#define MMX_MAX_SIGMAS 64

struct mmx_rescore_params {
    int32_t radius[MMX_MAX_SIGMAS];
    double norm[MMX_MAX_SIGMAS];
    int32_t strip;      // dx columns staged at once
    int32_t n_blocks;
};

namespace {

template <typename InT> struct in_scale;
template <> struct in_scale<uint8_t>  { static __device__ double get(uint8_t v)  { return (double)v * (1.0 / 255.0); } };
template <> struct in_scale<uint16_t> { static __device__ double get(uint16_t v) { return (double)v * (1.0 / 65535.0); } };
template <> struct in_scale<float>    { static __device__ double get(float v)    { return (double)v; } };
template <> struct in_scale<double>   { static __device__ double get(double v)   { return v; } };

// one symmetric 1-D correlation at the centre of `vals` (length 2R+1, stride `st`), SciPy order
template <typename StoreT>
__device__ __forceinline__ double corr_at(const double* vals, int st, int R, const double* w)
{
    double acc = vals[R * st] * w[0];
    for (int k = R; k >= 1; --k) acc += (vals[(R - k) * st] + vals[(R + k) * st]) * w[k];
    return (double)(StoreT)acc;
}

#ifndef MMX_RESCORE_BATCH
#define MMX_RESCORE_BATCH 8
#endif
constexpr int kB = MMX_RESCORE_BATCH;    // taps whose loads are in flight together

template <typename InT, typename StoreT>
__global__ void __launch_bounds__(MMX_WG)
rescore_kernel(const InT* __restrict__ vol, int64_t sz, int64_t sy, int64_t sx,
               const mmx_block* __restrict__ blocks, mmx_cand* __restrict__ pts, uint32_t cap,
               const uint32_t* __restrict__ count, const double* __restrict__ w0_tab,
               const double* __restrict__ w2_tab, mmx_rescore_params prm)
{
    extern __shared__ double smem[];
    uint32_t n = cap;
    if (count) { const uint32_t c = *count; n = c < cap ? c : cap; }
    // A fixed number of workgroups walks the table (round 4).  One workgroup per table ENTRY used to be launched --
    // the table's capacity, ~10^6, for the ~3 x 10^4 points a batch really has: 97 % of the workgroups read the count and
    // left, which still cost each a wave slot for a few hundred cycles (~0.1 ms per launch).
    for (uint64_t idx64 = blockIdx.x; idx64 < n; idx64 += gridDim.x) {
    const uint32_t idx = (uint32_t)idx64;
    // (the point is the same for every lane: readfirstlane tells the compiler, so that the weight tables -- indexed by
    //  the point's scale -- are read by scalar loads next to their use instead of two vector loads per tap)
    const mmx_cand ptv = pts[idx];
    mmx_cand pt;
    pt.slot = __builtin_amdgcn_readfirstlane(ptv.slot);
    pt.s = __builtin_amdgcn_readfirstlane(ptv.s);
    pt.z = __builtin_amdgcn_readfirstlane(ptv.z);
    pt.y = __builtin_amdgcn_readfirstlane(ptv.y);
    pt.x = __builtin_amdgcn_readfirstlane(ptv.x);
    if (pt.slot < 0 || pt.slot >= prm.n_blocks || pt.s < 0 || pt.s >= MMX_MAX_SIGMAS) continue;     // (whole workgroup)
    const mmx_block bd = blocks[pt.slot];
    const int R = prm.radius[pt.s];
    const int N = 2 * R + 1;
    const int SW = prm.strip < N ? prm.strip : N;
    const double* w0 = w0_tab + (int64_t)pt.s * (MMX_MAX_RADIUS_GENERIC + 1);
    const double* w2 = w2_tab + (int64_t)pt.s * (MMX_MAX_RADIUS_GENERIC + 1);
    const InT* in = vol + bd.src_off;

    double* zp0 = smem;                 // [N][SW]  axis-0 sums, order-0 kernel
    double* zp2 = zp0 + (size_t)N * SW; // [N][SW]  axis-0 sums, order-2 kernel
    double* yp = zp2 + (size_t)N * SW;  // [3][N]   axis-1 sums of the three terms
    // [N] element offsets of the reflected z planes (as 64-bit products once per point: a 64-bit integer
    // multiply per load would cost as much issue time as the float64 arithmetic of the tap)
    int64_t* zr = (int64_t*)(yp + 3 * (size_t)N);

    __syncthreads();                    // (the previous point's last reads of this memory)
    for (int k = threadIdx.x; k < N; k += MMX_WG) zr[k] = (int64_t)mmx_reflect(pt.z + k - R, bd.nz) * sz;
    __syncthreads();
    for (int dx0 = 0; dx0 < N; dx0 += SW) {
        const int sw = (N - dx0) < SW ? (N - dx0) : SW;
        // ---- axis 0 (z): two kernels share every load
        for (int e = threadIdx.x; e < N * sw; e += MMX_WG) {
            const int dyi = e / sw;
            const int dxi = e - dyi * sw;
            const int yy = mmx_reflect(pt.y + dyi - R, bd.ny);
            const int xx = mmx_reflect(pt.x + dx0 + dxi - R, bd.nx);
            const InT* col = in + (int64_t)yy * sy + (int64_t)xx * sx;
            const double c = in_scale<InT>::get(col[zr[R]]);
            double a0 = c * w0[0];
            double a2 = c * w2[0];
            // the taps in SciPy's order (k = R .. 1); the loads of kB taps are issued before the first is
            // used -- a tap-by-tap loop exposes the full memory latency 2R times per column, and that
            // latency, not the arithmetic, was this kernel's time
            int k = R;
            auto batch = [&](auto nb) __attribute__((always_inline)) {
                constexpr int B = decltype(nb)::value;
                for (; k >= B; k -= B) {
                    InT lo[B], hi[B];
#pragma unroll
                    for (int j = 0; j < B; ++j) {
                        lo[j] = col[zr[R - (k - j)]];
                        hi[j] = col[zr[R + (k - j)]];
                    }
#pragma unroll
                    for (int j = 0; j < B; ++j) {
                        const double p = in_scale<InT>::get(lo[j]) + in_scale<InT>::get(hi[j]);
                        a0 += p * w0[k - j];
                        a2 += p * w2[k - j];
                    }
                }
            };
            batch(std::integral_constant<int, kB>{});      // then the remainder in halves: no tap-by-tap tail
            batch(std::integral_constant<int, 4>{});
            batch(std::integral_constant<int, 2>{});
            batch(std::integral_constant<int, 1>{});
            zp0[dyi * SW + dxi] = (double)(StoreT)a0;
            zp2[dyi * SW + dxi] = (double)(StoreT)a2;
        }
        __syncthreads();
        // ---- axis 1 (y): term 0 = G(y) on G''(z);  term 1 = G''(y) on G(z);  term 2 = G(y) on G(z)
        for (int t = threadIdx.x; t < 3 * sw; t += MMX_WG) {
            const int term = t / sw;
            const int dxi = t - term * sw;
            const double* src = (term == 0 ? zp2 : zp0) + dxi;
            yp[term * N + dx0 + dxi] = corr_at<StoreT>(src, SW, R, term == 1 ? w2 : w0);
        }
        __syncthreads();
    }
    // ---- axis 2 (x) and the sum of the three terms, in SciPy's order
    if (threadIdx.x < 3)      // the three terms side by side (each a serial chain of R dependent LDS reads)
        zp0[threadIdx.x] = corr_at<StoreT>(yp + threadIdx.x * N, 1, R, threadIdx.x == 2 ? w2 : w0);
    __syncthreads();
    if (threadIdx.x == 0) {
        const StoreT t0 = (StoreT)zp0[0];
        const StoreT t1 = (StoreT)zp0[1];
        const StoreT t2 = (StoreT)zp0[2];
        StoreT sum = t0;
        sum += t1;
        sum += t2;
        const StoreT cube = (-sum) * (StoreT)prm.norm[pt.s];
        pts[idx].v64 = (double)cube;
    }
    }       // next point
}

template <typename InT, typename StoreT>
int launch(const mmx_volume* vol, const mmx_block* d_blocks, mmx_cand* d_pts, uint32_t cap,
           const uint32_t* d_count, const double* d_w0, const double* d_w2,
           const mmx_rescore_params& prm, size_t lds, hipStream_t s)
{
    // (256 CUs x the workgroups a CU holds at 60 KiB of LDS each, a few times over: the points' costs differ with sigma)
    const uint32_t gx = cap < 8192u ? cap : 8192u;
    hipLaunchKernelGGL((rescore_kernel<InT, StoreT>), dim3(gx), dim3(MMX_WG), lds, s,
                       (const InT*)vol->d_data, vol->stride_z, vol->stride_y, vol->stride_x, d_blocks,
                       d_pts, cap, d_count, d_w0, d_w2, prm);
    return hipGetLastError() == hipSuccess ? MMX_OK : MMX_ERR_HIP;
}

}  // namespace

// ---- probes: the neighbours whose exact values decide a contested candidate, appended to the candidate table
// itself so that ONE re-score launch and ONE copy bring the host everything it needs (the host used to build this
// list from the candidates, upload it and wait for a second re-score: two device round trips per batch).
namespace {
__global__ void __launch_bounds__(MMX_WG)
expand_probes_kernel(mmx_cand* __restrict__ tab, uint32_t cap, uint32_t* __restrict__ count,
                     const uint32_t* __restrict__ n_cands, const mmx_block* __restrict__ blocks, int n_blocks, int ns)
{
    const uint32_t n = *n_cands < cap ? *n_cands : cap;
    for (uint64_t i = (uint64_t)blockIdx.x * MMX_WG + threadIdx.x; i < n; i += (uint64_t)gridDim.x * MMX_WG) {
        const mmx_cand c = tab[i];
        if (!(c.flags & MMX_CAND_CONTESTED) || c.slot < 0 || c.slot >= n_blocks) continue;
        const mmx_block bd = blocks[c.slot];
        const bool banded = (c.flags & MMX_CAND_BAND) != 0;
        int j = 0;
        for (int ds = -1; ds <= 1; ++ds)
            for (int dz = -1; dz <= 1; ++dz)
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) {
                        if ((ds | dz | dy | dx) == 0) continue;
                        const int bit = j++;
                        const int ss = c.s + ds, zz = c.z + dz, yy = c.y + dy, xx = c.x + dx;
                        if (ss < 0 || ss >= ns || zz < 0 || zz >= bd.nz || yy < 0 || yy >= bd.ny || xx < 0 || xx >= bd.nx)
                            continue;
                        if (banded && !(bit < 64 ? (c.band >> bit) & 1ull : (c.flags >> (16 + bit - 64)) & 1u)) continue;
                        const uint32_t pos = atomicAdd(count, 1u);
                        if (pos < cap) {
                            mmx_cand r;
                            r.slot = c.slot; r.s = ss; r.z = zz; r.y = yy; r.x = xx;
                            r.flags = MMX_CAND_PROBE;
                            r.v = 0.f; r.nbr_max = 0.f;
                            r.v64 = __longlong_as_double(0x7ff8000000000000LL);
                            r.band = i;                      // the candidate this neighbour may out-vote
                            tab[pos] = r;
                        }
                    }
    }
}
}  // namespace

extern "C" int mmx_expand_probes(mmx_cand* d_cands, uint32_t cap, uint32_t* d_count, uint32_t* d_n_cands,
                                 const mmx_block* d_blocks, int n_blocks, int n_sigma, void* stream)
{
    if (!d_cands || !d_count || !d_n_cands || !d_blocks || n_blocks < 1 || n_sigma < 1) return MMX_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemcpyAsync(d_n_cands, d_count, sizeof(uint32_t), hipMemcpyDeviceToDevice, s) != hipSuccess) return MMX_ERR_HIP;
    if (cap == 0) return MMX_OK;
    mmx_timed_scope ts(MMX_K_RESCORE, s);
    const uint32_t gx = std::min<uint32_t>(1024u, (cap + MMX_WG - 1) / MMX_WG);
    hipLaunchKernelGGL(expand_probes_kernel, dim3(gx), dim3(MMX_WG), 0, s, d_cands, cap, d_count, d_n_cands, d_blocks,
                       n_blocks, n_sigma);
    return hipGetLastError() == hipSuccess ? MMX_OK : MMX_ERR_HIP;
}

extern "C" int mmx_rescore_f64(const mmx_volume* vol, const mmx_block* d_blocks, int n_blocks,
                               mmx_cand* d_pts, uint32_t cap, const uint32_t* d_count,
                               const double* d_w0, const double* d_w2, const int32_t* h_radius,
                               const double* h_norm, int n_sigma, int store_f32, void* stream)
{
    if (!vol || !vol->d_data || !d_blocks || !d_pts || !d_w0 || !d_w2 || !h_radius || !h_norm)
        return MMX_ERR_ARG;
    if (n_sigma < 1 || n_sigma > MMX_MAX_SIGMAS || n_blocks < 1) return MMX_ERR_ARG;
    if (cap == 0) return MMX_OK;
    mmx_rescore_params prm;
    int rmax = 0;
    for (int i = 0; i < MMX_MAX_SIGMAS; ++i) { prm.radius[i] = 0; prm.norm[i] = 0.0; }
    for (int i = 0; i < n_sigma; ++i) {
        if (h_radius[i] < 0 || h_radius[i] > MMX_MAX_RADIUS_GENERIC) return MMX_ERR_UNSUPPORTED;
        prm.radius[i] = h_radius[i];
        prm.norm[i] = h_norm[i];
        if (h_radius[i] > rmax) rmax = h_radius[i];
    }
    prm.n_blocks = n_blocks;
    const int N = 2 * rmax + 1;
    const size_t budget = 60 * 1024;
    const size_t fixed = (size_t)3 * N * sizeof(double) + (size_t)N * sizeof(int64_t) + 16;
    int strip = (int)((budget - fixed) / ((size_t)2 * N * sizeof(double)));
    if (strip < 1) return MMX_ERR_UNSUPPORTED;
    if (strip > N) strip = N;
    prm.strip = strip;
    const size_t lds = (size_t)2 * N * strip * sizeof(double) + fixed;
    hipStream_t s = (hipStream_t)stream;
    const bool f32s = store_f32 != 0;
    mmx_timed_scope ts(MMX_K_RESCORE, s);
    switch (vol->dtype) {
        case MMX_U8:  return launch<uint8_t, double>(vol, d_blocks, d_pts, cap, d_count, d_w0, d_w2, prm, lds, s);
        case MMX_U16: return launch<uint16_t, double>(vol, d_blocks, d_pts, cap, d_count, d_w0, d_w2, prm, lds, s);
        case MMX_F32: return f32s ? launch<float, float>(vol, d_blocks, d_pts, cap, d_count, d_w0, d_w2, prm, lds, s)
                                  : launch<float, double>(vol, d_blocks, d_pts, cap, d_count, d_w0, d_w2, prm, lds, s);
        case MMX_F64: return launch<double, double>(vol, d_blocks, d_pts, cap, d_count, d_w0, d_w2, prm, lds, s);
        default: return MMX_ERR_ARG;
    }
}
