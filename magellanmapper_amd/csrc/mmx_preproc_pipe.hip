// Per-block preprocessing for tiles of more than 4 096 voxels (the stock 25^3 denoise tile): the work of
// pp_fast_kernel (mmx_preproc.hip: same reference, magmap/plot/plot_3d.py:55-172 through
// magmap/cv/stack_detect.py:122-150, same bit-for-bit arithmetic) cut in two kernels so that the float64 ALU
// is not idle while a tile waits for memory.
//
// Why.  One 25^3 tile is 122 KiB of float64, so a CU holds ONE such tile in LDS whatever the kernel looks
// like.  pp_fast_kernel spends 62 us per tile and CU of which 33.5 are the blur (float64 VALU bound on the
// SIMDs that carry three of its ten waves); the other 29 us -- voxels in, histogram, radix select, stretch,
// write out -- are latency with nothing beside them (profiles/HISTORY.md section 4c).  Here:
//   * pp_retile_kernel: a tile's voxels are 25-voxel row segments 4 KiB apart; loading them tile by tile costs
//     ~1 000 half-used cache-line requests per tile and took 8-11 us per tile and CU whatever the kernel did with
//     them (measured: three different loaders).  So the voxels are first copied, strip of x-adjacent tiles by strip
//     (whole block rows: coalesced reads), into a TILE-MAJOR uint16 copy: every later read of a tile is one
//     contiguous 31 KiB stream.  +2 B/voxel written, all reads coalesced.
//   * pp_stats_kernel: histogram of the high byte -> order statistics -> vmin / vmax / mean / flags per tile.  5 KiB
//     of LDS and 256 lanes per tile, so several tiles per CU overlap their latencies.
//   * pp_blur_kernel: a workgroup walks a run of tiles.  The stretch is folded into the first line pass (its
//     input is clip(stretch(voxel)) computed from the uint16 copy in LDS), the voxels of the NEXT tile are
//     loaded while the unsharp stage of the current one runs and dropped into the LDS words that stage has
//     just finished with, barriers wait for LDS only (stores stay in flight across tiles).
//   * the line pass is specialised for lines of 25 (clamped tap indices are compile-time constants: the pair
//     sums and the both-ends-clamped products are shared by the outputs of a line -- 1 494 instead of 1 700
//     float64 operations per line) and dealt to TWELVE waves so that every SIMD carries 2.5 wave-lines per pass:
//     waves 0..7 take 64 whole lines each, waves 8..11 take the remaining 113 lines as four half passes
//     (outputs 0..12 / 13..24 of 64 / 49 lines), against 3 / 3 / 2 / 2 whole wave-lines before.
// Tiles of <= 4 096 voxels (anisotropic data) keep pp_fast_kernel with 256 lanes (several tiles per CU).

#include <type_traits>

#include "mmx_pp_common.h"

#define PPS_WG 256           // statistics kernel
#define PPS_NB_LOAD 16       // global loads in flight per lane
#define PPB_WG 768           // blur kernel: twelve waves, three per SIMD (<= 168 VGPRs)
#define PPB_L 25             // the line length the pass is specialised for
#define MMX_PP_KNIFE 0x100   // internal: the statistics kernel asks the blur kernel for np.mean's own summation order
#define PPB_NB 7             // voxels per lane and step in the element-wise stages (25^3: 21 rows per lane = 3 x 7)

// -DPP_PROFILE (an experiment build: make EXTRA=-DPP_PROFILE OBJDIR=_obj_prof OUT=../libmmx_prof.so): lane 0 of every
// workgroup adds the 100 MHz ticks between its stamps to pp_prof[kernel][phase]; tools/ppbench.py --profile reads them
#ifdef PP_PROFILE
__device__ unsigned long long pp_prof[2][16];
#define PP_T0() unsigned long long pp_t = __builtin_amdgcn_s_memrealtime()
#define PP_TICK(K, P) do { if (threadIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); \
        atomicAdd(&pp_prof[K][P], t_ - pp_t); pp_t = t_; } } while (0)
extern "C" int mmx_pp_profile_read(unsigned long long* out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pp_prof), sizeof(unsigned long long) * 32) != hipSuccess) return MMX_ERR_HIP;
    if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(pp_prof), z, sizeof z) != hipSuccess) return MMX_ERR_HIP; }
    return MMX_OK;
}
#else
#define PP_T0() do {} while (0)
#define PP_TICK(K, P) do {} while (0)
#endif

namespace {

// LDS-only workgroup barrier: __syncthreads() also drains vmcnt, i.e. would wait for the output stores of
// the previous tile at every pass boundary
__device__ __forceinline__ void pp_lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Raw buffer descriptors for the element-wise stage: a lane that has nothing to load or store passes an offset
// beyond the descriptor's range -- the hardware drops it -- instead of branching around the instruction, so every
// vector-memory instruction of the stage is issued by every wave and the compiler's s_waitcnt vmcnt counts are exact
// (a wait for the next tile's voxels then never waits for this tile's output stores, which were issued later).
using pp_rsrc_t = __amdgpu_buffer_rsrc_t;
typedef unsigned pp_v2u __attribute__((ext_vector_type(2)));
#define PP_OOB 0xffffffffu
__device__ __forceinline__ pp_rsrc_t pp_rsrc(const void* p)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void pp_store_f64(pp_rsrc_t r, unsigned off, double v)
{
    const pp_v2u d = {(unsigned)__double2loint(v), (unsigned)__double2hiint(v)};
    __builtin_amdgcn_raw_buffer_store_b64(d, r, off, 0, 0);
}
__device__ __forceinline__ void pp_store_f32(pp_rsrc_t r, unsigned off, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, off, 0, 0);
}
__device__ __forceinline__ double pp_uniform(double v)          // a wave-uniform double into SGPRs
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// ------------------------------------------------------------------------------------------------------
// tile-major copy.  copy_off[i] = voxels of the tiles before tile i (exclusive prefix over the table order);
// is_head[i] = tile i starts a strip.  A strip = tiles of the table that continue each other along x (same nz, ny;
// src_off advances by the width), all of one width except possibly the last: what one workgroup of the copy kernel
// reads as whole rows.
__device__ __forceinline__ bool pp_continues(const mmx_subblock& a, const mmx_subblock& b, int64_t sx)
{
    return b.nz == a.nz && b.ny == a.ny && b.src_off == a.src_off + (int64_t)a.nx * sx;
}

#define PPO_WG 1024
__global__ void __launch_bounds__(PPO_WG)
pp_offsets1_kernel(const mmx_subblock* __restrict__ subs, int n_subs, int64_t* __restrict__ copy_off,
                   int64_t* __restrict__ wg_sum)
{
    __shared__ int64_t s_scan[PPO_WG];
    const int tid = threadIdx.x, i = (int)blockIdx.x * PPO_WG + tid;
    const int64_t n = i < n_subs ? (int64_t)subs[i].nz * subs[i].ny * subs[i].nx : 0;
    s_scan[tid] = n;
    __syncthreads();
    for (int d = 1; d < PPO_WG; d <<= 1) {
        const int64_t t = tid >= d ? s_scan[tid - d] : 0;
        __syncthreads();
        s_scan[tid] += t;
        __syncthreads();
    }
    if (i < n_subs) copy_off[i] = s_scan[tid] - n;            // exclusive, inside this workgroup's 1 024 tiles
    if (tid == PPO_WG - 1) wg_sum[blockIdx.x] = s_scan[tid];
}

__global__ void __launch_bounds__(PPO_WG)
pp_offsets2_kernel(const mmx_subblock* __restrict__ subs, int n_subs, int64_t sx, int64_t* __restrict__ copy_off,
                   const int64_t* __restrict__ wg_sum, int* __restrict__ is_head)
{
    __shared__ int64_t s_part[PPO_WG / 64];
    __shared__ int64_t s_base;
    const int tid = threadIdx.x, i = (int)blockIdx.x * PPO_WG + tid;
    // voxels of the workgroups before this one (at most 1 024 of them: the launcher checks)
    int64_t v = tid < (int)blockIdx.x ? wg_sum[tid] : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d);
    if ((tid & 63) == 0) s_part[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) {
        int64_t b = 0;
        for (int w = 0; w < PPO_WG / 64; ++w) b += s_part[w];
        s_base = b;
    }
    __syncthreads();
    if (i >= n_subs) return;
    copy_off[i] += s_base;
    // strips: a greedy parse of the chain of continuing tiles this tile belongs to, from the chain's first tile
    int s0 = i;
    while (s0 > 0 && pp_continues(subs[s0 - 1], subs[s0], sx)) --s0;
    int nx0 = subs[s0].nx;
    bool closed = false, head = true;          // closed: the previous tile was the odd-width last tile of its strip
    for (int j = s0 + 1; j <= i; ++j) {
        const int w = subs[j].nx;
        if (closed) { head = true; nx0 = w; closed = false; }
        else { head = false; if (w != nx0) closed = true; }
    }
    is_head[i] = head ? 1 : 0;
}

#define PPR_WG 256
#define PPR_NK 5             // columns per lane and sweep: 5 x 64 = 320 voxels of a strip row
template <typename InT>
__global__ void __launch_bounds__(PPR_WG)
pp_retile_kernel(const InT* __restrict__ vol, int64_t sz, int64_t sy, int64_t sx,
                 const mmx_subblock* __restrict__ subs, int n_subs,
                 const int64_t* __restrict__ copy_off, const int* __restrict__ is_head,
                 uint16_t* __restrict__ copy)
{
    const int head = (int)blockIdx.x;
    if (!is_head[head]) return;
    const mmx_subblock h = subs[head];
    int c = 1;                                             // tiles of the strip
    while (head + c < n_subs && !is_head[head + c]) ++c;
    const int nx0 = h.nx, nxl = subs[head + c - 1].nx, ny = h.ny;
    const int W = (c - 1) * nx0 + nxl, rows = h.nz * h.ny;
    const float inv_nx0 = 1.0f / (float)nx0;
    const int tile_vox0 = rows * nx0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // (a lane with no voxel passes offsets beyond the descriptors' ranges: no branches, loads and stores pipeline)
    const pp_rsrc_t src = pp_rsrc(vol + h.src_off);
    const pp_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(copy + copy_off[head], 0, rows * W * 2, 0x00020000);
    const int esz = (int)sx * (int)sizeof(InT), row_b = (int)sy * (int)sizeof(InT);
    const unsigned plane_fix = (unsigned)((int)sz - ny * (int)sy) * (unsigned)sizeof(InT);
    // a wave takes every fourth row of the strip; its lanes keep their columns: per voxel one load, one store and
    // the store offset's multiply-add
    for (int x0 = 0; x0 < W; x0 += 64 * PPR_NK) {
        unsigned scol[PPR_NK];
        int dcol[PPR_NK], dstep[PPR_NK];
#pragma unroll
        for (int k = 0; k < PPR_NK; ++k) {
            const int x = x0 + lane + 64 * k;
            int t = (int)(((float)x + 0.5f) * inv_nx0);
            t = t < c - 1 ? t : c - 1;
            dstep[k] = t < c - 1 ? nx0 : nxl;
            dcol[k] = x < W ? t * tile_vox0 + (x - t * nx0) : 0x3fffffff;
            scol[k] = x < W ? (unsigned)(x * esz) : PP_OOB;
        }
        int y = wave, z = 0;
        while (y >= ny) { y -= ny; ++z; }
        unsigned soff = (unsigned)z * (unsigned)((int)sz * (int)sizeof(InT)) + (unsigned)(y * row_b);
        for (int row = wave; row < rows; row += PPR_WG / 64) {
            int v[PPR_NK];
#pragma unroll
            for (int k = 0; k < PPR_NK; ++k) {
                if constexpr (sizeof(InT) == 1) v[k] = (int)(unsigned)__builtin_amdgcn_raw_buffer_load_b8(src, scol[k], soff, 0);
                else v[k] = (int)(unsigned)__builtin_amdgcn_raw_buffer_load_b16(src, scol[k], soff, 0);
            }
#pragma unroll
            for (int k = 0; k < PPR_NK; ++k)
                __builtin_amdgcn_raw_buffer_store_b16((unsigned short)v[k], dst, (unsigned)(dcol[k] + row * dstep[k]) << 1, 0, 0);
            y += PPR_WG / 64; soff += (unsigned)(PPR_WG / 64) * (unsigned)row_b;
            while (y >= ny) { y -= ny; soff += plane_fix; }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// statistics: the order statistics np.percentile interpolates between -> vmin / vmax, and np.mean of the stretched
// tile -> the erosion flag.  256 lanes and 5 KiB of histograms per tile, several tiles per CU; the tile is read from
// the tile-major copy three times (high-byte histogram, low-byte histograms of the selected bins, the sums), in
// chunks of 8 loads per lane: the second and third time from L2.  (Measured against holding a lane's 64 voxels in
// registers: the fully unrolled passes then keep 128 compare masks alive in scalar registers and spill them lane by
// lane -- 0.97 against 0.65 ms per 27 blocks.)
// High-byte histogram of a lane's voxels.  A tile is mostly background, whose voxels share two high bytes: LDS
// atomics of many lanes on ONE address serialise (measured: 50 cycles per wave-instruction), a loop over the
// distinct bins of a wave costs a scalar round trip per bin.  So every wave picks the two bins its first 64 voxels
// name first (wave-uniform), lanes count THOSE in registers (two compares and adds per voxel, no LDS), only the
// other voxels -- blobs, spread over many bins -- take an atomic each; two atomics per wave add the private counts.
struct pp_hist2 {
    int d0, d1;
    uint32_t c0, c1;
    __device__ __forceinline__ void pick(int vv)
    {
        const int bin = vv >> 8;
        const unsigned long long act = __ballot(vv >= 0);
        d0 = act ? __builtin_amdgcn_readlane(bin, __ffsll((long long)act) - 1) : -2;
        const unsigned long long rest = __ballot(vv >= 0 && bin != d0);
        d1 = rest ? __builtin_amdgcn_readlane(bin, __ffsll((long long)rest) - 1) : -2;
        c0 = c1 = 0;
    }
    __device__ __forceinline__ void add(uint32_t* hist, int vv)
    {
        const int bin = vv >> 8;                       // (a lane without a voxel holds -1: bin -1 matches nothing)
        c0 += bin == d0 ? 1u : 0u;
        c1 += bin == d1 ? 1u : 0u;
        if (vv >= 0 && bin != d0 && bin != d1) atomicAdd(&hist[bin], 1u);
    }
    __device__ __forceinline__ void flush(uint32_t* hist)
    {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { c0 += __shfl_down(c0, d); c1 += __shfl_down(c1, d); }
        if ((threadIdx.x & 63) == 0) {
            if (d0 >= 0 && c0) atomicAdd(&hist[d0], c0);
            if (d1 >= 0 && c1) atomicAdd(&hist[d1], c1);
        }
    }
};

__global__ void __launch_bounds__(PPS_WG)
pp_stats_kernel(const uint16_t* __restrict__ copy, const int64_t* __restrict__ copy_off,
                const mmx_subblock* __restrict__ subs, int n_subs,
                const mmx_quantile_class* __restrict__ qcs, pp_args A, mmx_subblock_info* __restrict__ info)
{
    __shared__ uint32_t hist[PP_HIST];
    __shared__ int s_bin[4];
    __shared__ uint32_t s_res[4];
    __shared__ int s_val[4];
    __shared__ uint32_t s_sum[PPS_WG / 64][4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (n_subs + 7) >> 3;                 // contiguous runs of tiles per XCD (see pp_fast_kernel)
    const int sub_id = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (sub_id >= n_subs) return;
    const mmx_subblock sb = subs[sub_id];
    const int nz = sb.nz, ny = sb.ny, nx = sb.nx, n = nz * ny * nx;
    // the tile's voxels, contiguous; a load past its end returns 0 and is replaced by -1: no branch, the loads pipeline
    const pp_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(copy + copy_off[sub_id]), 0, n * 2, 0x00020000);
    auto vox = [&](int i) {
        const int r = (int)(unsigned)__builtin_amdgcn_raw_buffer_load_b16(src, (unsigned)i << 1, 0, 0);
        return i < n ? r : -1;
    };

    PP_T0();
    for (int i = tid; i < PP_HIST; i += PPS_WG) hist[i] = 0;
    __syncthreads();

    {
        pp_hist2 h;
        bool picked = false;
        for (int i0 = 0; i0 < n; i0 += 8 * PPS_WG) {           // same trip count in every lane (ballots)
            int w8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) w8[j] = vox(i0 + j * PPS_WG + tid);
            if (!picked) { h.pick(w8[0]); picked = true; }
#pragma unroll
            for (int j = 0; j < 8; ++j) h.add(hist, w8[j]);
        }
        if (picked) h.flush(hist);
    }
    __syncthreads();
    PP_TICK(0, 0);

    const mmx_quantile_class qc = qcs[sb.qclass];
    {
        const uint32_t rank = wave == 0 ? qc.lo_prev : wave == 1 ? qc.lo_next : wave == 2 ? qc.hi_prev : qc.hi_next;
        int b; uint32_t r;
        pp_select(hist, rank, b, r);
        if (lane == 0) { s_bin[wave] = b; s_res[wave] = r; }
    }
    __syncthreads();
    PP_TICK(0, 1);
    {
        // one low-byte histogram per DISTINCT high-byte bin (prev / next ranks usually share theirs)
        const int b0 = s_bin[0], b1 = s_bin[1], b2 = s_bin[2], b3 = s_bin[3];
        const bool u1 = b1 != b0, u2 = b2 != b0 && b2 != b1, u3 = b3 != b0 && b3 != b1 && b3 != b2;
        auto add2 = [&](int vv) {
            if (vv < 0) return;
            const int hi = vv >> 8, lo = vv & 255;
            if (hi == b0) atomicAdd(&hist[256 + lo], 1u);
            if (u1 && hi == b1) atomicAdd(&hist[512 + lo], 1u);
            if (u2 && hi == b2) atomicAdd(&hist[768 + lo], 1u);
            if (u3 && hi == b3) atomicAdd(&hist[1024 + lo], 1u);
        };
        for (int i0 = 0; i0 < n; i0 += 8 * PPS_WG) {
            int w8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) w8[j] = vox(i0 + j * PPS_WG + tid);
#pragma unroll
            for (int j = 0; j < 8; ++j) add2(w8[j]);
        }
    }
    __syncthreads();
    PP_TICK(0, 2);
    {
        int slot = wave;
        for (int w = wave - 1; w >= 0; --w) if (s_bin[w] == s_bin[wave]) slot = w;
        int b; uint32_t r;
        pp_select(hist + 256 * (1 + slot), s_res[wave], b, r);
        if (lane == 0) s_val[wave] = (s_bin[wave] << 8) | b;
    }
    __syncthreads();
    PP_TICK(0, 3);
    // np.mean(saturated) gates the erosion.  saturated = (clip(v, vmin, vmax) - vmin) / span, so its sum is
    // (sum over vmin <= v <= vmax of (v - floor(vmin)) - n_mid frac(vmin) + n_hi span) / span up to roundings of
    // 1e-16 of itself: integer sums.  Within 1e-9 of the threshold the blur kernel, which
    // holds the voxels in LDS, re-sums in NumPy's pairwise order (MMX_PP_KNIFE: internal, cleared there).
    const double vmin = pp_lerp(s_val[0], s_val[1], qc.lo_gamma);
    double vmax = pp_lerp(s_val[2], s_val[3], qc.hi_gamma);
    const bool identity = vmin == vmax;
    if (vmax < A.max_thresh) vmax = A.max_thresh;
    const int v0 = (int)floor(vmin);
    // (integer voxels: v >= vmin <=> v >= ceil(vmin), v <= vmax <=> v <= floor(vmax); vmax may exceed every voxel)
    const int ic_lo = (int)ceil(vmin), ic_hi = vmax >= 65535.0 ? 65535 : (int)floor(vmax);
    uint32_t i_mid = 0, n_mid = 0, n_hi = 0, i_all = 0;
    auto tally = [&](int vv) {
        const bool mid = vv >= ic_lo && vv <= ic_hi;
        i_mid += mid ? (uint32_t)(vv - v0) : 0u;
        n_mid += mid ? 1u : 0u;
        n_hi += vv > ic_hi ? 1u : 0u;
        i_all += vv >= 0 ? (uint32_t)vv : 0u;
    };
    for (int i0 = 0; i0 < n; i0 += 8 * PPS_WG) {
        int w8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w8[j] = vox(i0 + j * PPS_WG + tid);
#pragma unroll
        for (int j = 0; j < 8; ++j) tally(w8[j]);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        i_mid += __shfl_down(i_mid, d); n_mid += __shfl_down(n_mid, d);
        n_hi += __shfl_down(n_hi, d); i_all += __shfl_down(i_all, d);
    }
    if (lane == 0) { s_sum[wave][0] = i_mid; s_sum[wave][1] = n_mid; s_sum[wave][2] = n_hi; s_sum[wave][3] = i_all; }
    __syncthreads();
    if (tid == 0) {
        uint32_t t[4] = {0, 0, 0, 0};
        for (int w = 0; w < PPS_WG / 64; ++w) for (int q = 0; q < 4; ++q) t[q] += s_sum[w][q];
        const double nd = (double)n, span = vmax - vmin;
        double mean;
        if (identity) mean = (double)t[3] / nd;
        else mean = (((double)t[0] - (double)t[1] * (vmin - (double)v0)) + (double)t[2] * span) / span / nd;
        int fl = identity ? MMX_PP_IDENTITY : 0;
        if (A.do_erosion) {
            const double tol = 1e-9 * (fabs(A.ero_thr) > 1. ? fabs(A.ero_thr) : 1.);
            if (!identity && fabs(mean - A.ero_thr) <= tol) fl |= MMX_PP_KNIFE;
            else if (mean > A.ero_thr) fl |= MMX_PP_ERODED;
        }
        mmx_subblock_info o;
        o.vmin = vmin; o.vmax = vmax; o.mean = mean; o.flags = fl; o._pad = 0;
        info[sub_id] = o;
    }
    PP_TICK(0, 4);
}

// ------------------------------------------------------------------------------------------------------
// One line of the sigma-8 Gaussian in registers.  EXACT: the line has exactly NR voxels, the 'nearest'
// extension clamps the tap indices to [0, NR-1] at compile time.  Otherwise NR = 32 registers hold a line of
// L <= 32 voxels with its last voxel replicated (pp_line_pass's scheme).
template <int NR, bool EXACT>
struct pp_line {
    double r[NR];

    template <typename F>
    __device__ __forceinline__ void load(F f, int L)
    {
#pragma unroll
        for (int i = 0; i < NR; ++i) r[i] = f(EXACT ? i : (i < L ? i : L - 1), EXACT || i < L);
    }

    // outputs I0 .. I1-1 of the line -> tile[base + i * stride]: SciPy's correlate1d order (symmetric branch),
    // acc = in[c] w0; for k = R .. 1: acc += (in[c-k] + in[c+k]) w[k]
    // `pre(i)` runs before output i replaces tile[base + i * stride] (the first pass of a tile flushes the previous
    // tile's result from there, see pp_blur_kernel)
    template <int I0, int I1, typename Pre>
    __device__ __forceinline__ void run(double* __restrict__ tile, int base, int stride, int L,
                                        const double (&w)[PP_R + 1], Pre pre) const
    {
#pragma unroll
        for (int i = I0; i < I1; ++i) {
            if (EXACT || i < L) {
                pre(i);
                double acc = r[i] * w[0];
#pragma unroll
                for (int k = PP_R; k >= 1; --k) {
                    const int a = i - k < 0 ? 0 : i - k;
                    const int b = i + k > NR - 1 ? NR - 1 : i + k;
                    acc += (r[a] + r[b]) * w[k];
                }
                tile[base + i * stride] = acc;
            }
        }
    }
};

struct pp_geom {
    int nz, ny, nx, px, n, nrows;
};

// base / stride of line l of the pass along `axis` in the float64 tile (t*) and in the dense uint16 copy (r*)
__device__ __forceinline__ void pp_line_addr(const pp_geom& g, int axis, int l, float inv_nx,
                                             int& tbase, int& tstride, int& rbase, int& rstride)
{
    if (axis == 2) { tbase = l * g.px; tstride = 1; rbase = l * g.nx; rstride = 1; return; }
    const int t = (int)(((float)l + 0.5f) * inv_nx);
    const int x = l - t * g.nx;
    if (axis == 0) { tbase = t * g.px + x; tstride = g.ny * g.px; rbase = t * g.nx + x; rstride = g.ny * g.nx; }
    else { tbase = t * g.ny * g.px + x; tstride = g.px; rbase = t * g.ny * g.nx + x; rstride = g.nx; }
}

__global__ void __launch_bounds__(PPB_WG)
pp_blur_kernel(const uint16_t* __restrict__ copy, const int64_t* __restrict__ copy_off,
               const mmx_subblock* __restrict__ subs, int n_subs, int tiles_per_wg,
               const double* __restrict__ wts, pp_args A, float* __restrict__ out32, double* __restrict__ out64,
               mmx_subblock_info* __restrict__ info, int raw_off)
{
    extern __shared__ double tile[];
    __shared__ int s_flags;
    __shared__ pp_stack s_stack;
    __shared__ double s_par[8];         // clip_min, clip_max, strength; vmin, vmax, span, 1 / span of the current tile
    uint16_t* raw = (uint16_t*)(tile + raw_off);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwg = (int)gridDim.x;                     // a multiple of 8: contiguous runs of tiles per XCD
    const int v_id = (int)(blockIdx.x & 7) * (nwg >> 3) + (int)(blockIdx.x >> 3);
    const int first = v_id * tiles_per_wg;
    if (first >= n_subs) return;
    const int last = first + tiles_per_wg < n_subs ? first + tiles_per_wg : n_subs;

    // the voxels of a tile -> LDS (dense uint16): one contiguous stream of the tile-major copy
    auto load_raw = [&](const mmx_subblock& t, int64_t off) {
        const int n = t.nz * t.ny * t.nx;
        const uint16_t* src = copy + off;
        for (int i0 = tid; i0 < n; i0 += PPB_NB * PPB_WG) {
            int v[PPB_NB];
#pragma unroll
            for (int j = 0; j < PPB_NB; ++j) v[j] = (int)src[i0 + j * PPB_WG < n ? i0 + j * PPB_WG : i0];
#pragma unroll
            for (int j = 0; j < PPB_NB; ++j) if (i0 + j * PPB_WG < n) raw[i0 + j * PPB_WG] = (uint16_t)v[j];
        }
    };

    PP_T0();
    mmx_subblock sb = subs[first];
    load_raw(sb, copy_off[first]);
    pp_lds_barrier();
    PP_TICK(1, 0);

    bool flush_prev = false;            // the previous tile's result is still in the LDS tile (same shape, not eroded):
    int64_t prev_dst = 0;               // this tile's first pass stores it plane by plane before overwriting the plane
#pragma unroll 1
    for (int k = first; k < last; ++k) {
        const mmx_subblock_info inf = info[k];
        mmx_subblock nsb = sb;
        const bool has_next = k + 1 < last;
        if (has_next) nsb = subs[k + 1];

        pp_geom g;
        g.nz = sb.nz; g.ny = sb.ny; g.nx = sb.nx; g.px = sb.nx | 1; g.n = sb.nz * sb.ny * sb.nx; g.nrows = sb.nz * sb.ny;
        const float inv_nx = 1.0f / (float)g.nx, inv_ny = 1.0f / (float)g.ny;
        pp_sat S;
        S.identity = inf.flags & MMX_PP_IDENTITY;
        S.vmin = inf.vmin; S.vmax = inf.vmax; S.span = inf.vmax - inf.vmin;
        S.finish();
        S.vmin = pp_uniform(S.vmin); S.vmax = pp_uniform(S.vmax); S.span = pp_uniform(S.span); S.rcp = pp_uniform(S.rcp);
        if (tid == 0) {                 // (read after the passes' barriers; rewritten after this tile's last barrier)
            s_par[0] = A.clip_min; s_par[1] = A.clip_max; s_par[2] = A.strength;
            s_par[3] = S.vmin; s_par[4] = S.vmax; s_par[5] = S.span; s_par[6] = S.rcp;
        }
        const double cmin = A.clip_min, cmax = A.clip_max;
        // (S.fast holds for every tile of integer voxels: 2^-200 < span <= 65535; the plain division is kept for form)
        auto sat_fast = [&](int rv) { return S((double)rv); };
        auto sat_plain = [&](int rv) { return S.plain((double)rv); };
        PP_TICK(1, 1);

        // ---- Gaussian blur, axis 0, 1, 2 (scipy gaussian_filter order); the first pass reads the voxels
        if (A.do_unsharp) {
            double wl[PP_R + 1];
#pragma unroll
            for (int q = 0; q <= PP_R; ++q) wl[q] = wts[q];
            const pp_rsrc_t p64 = pp_rsrc(out64 + prev_dst), p32 = pp_rsrc(out32 + prev_dst);
            const int f_dsz = (int)A.dst_sz, f_dsy = (int)A.dst_sy;
#pragma unroll 1
            for (int axis = 0; axis < 3; ++axis) {
                const int L = axis == 0 ? g.nz : axis == 1 ? g.ny : g.nx;
                if (axis == 2 && A.rgb_guess && g.nx == 3) break;
                const int nlines = g.n / L;
                const bool from_raw = axis == 0;
                const bool flush = from_raw && flush_prev;
                auto no_pre = [](int) {};
                if (L == PPB_L && nlines > 512 && nlines <= 640) {
                    // twelve waves, 2.5 wave-lines per SIMD: whole lines on waves 0..7, half passes on 8..11
                    const int l = wave < 8 ? tid : (8 + ((wave - 8) >> 1)) * 64 + lane;
                    const int half = wave < 8 ? 0 : 1 + ((wave - 8) & 1);
                    const bool on = l < nlines;
                    int tbase, tstride, rbase, rstride;
                    pp_line_addr(g, axis, on ? l : 0, inv_nx, tbase, tstride, rbase, rstride);
                    pp_line<PPB_L, true> ln;
                    if (from_raw) {
                        auto ld = [&](auto sat) {
                            ln.load([&](int i, bool) { return pp_clip(sat((int)raw[rbase + i * rstride]), cmin, cmax); }, L);
                        };
                        if (S.fast) ld(sat_fast); else ld(sat_plain);
                    } else {
                        ln.load([&](int i, bool) { return tile[tbase + i * tstride]; }, L);
                        // (two waves share the lines of the half passes: every read of an in-place pass before any write)
                        pp_lds_barrier();
                    }
                    // (z lines: line l = y * nx + x, output i is plane z = i -- the previous tile's voxel (i, y, x))
                    const int f_off = rbase / g.nx * f_dsy + rbase % g.nx;
                    auto pre_flush = [&](int i) {
                        const double o = tile[tbase + i * tstride];
                        const unsigned off = (unsigned)(f_off + i * f_dsz);
                        pp_store_f64(p64, off << 3, o);
                        pp_store_f32(p32, off << 2, (float)o);
                    };
                    if (on) {
                        if (flush) {
                            if (half == 0) ln.template run<0, PPB_L>(tile, tbase, tstride, L, wl, pre_flush);
                            else if (half == 1) ln.template run<0, (PPB_L + 1) / 2>(tile, tbase, tstride, L, wl, pre_flush);
                            else ln.template run<(PPB_L + 1) / 2, PPB_L>(tile, tbase, tstride, L, wl, pre_flush);
                        } else {
                            if (half == 0) ln.template run<0, PPB_L>(tile, tbase, tstride, L, wl, no_pre);
                            else if (half == 1) ln.template run<0, (PPB_L + 1) / 2>(tile, tbase, tstride, L, wl, no_pre);
                            else ln.template run<(PPB_L + 1) / 2, PPB_L>(tile, tbase, tstride, L, wl, no_pre);
                        }
                    }
                } else if (L == PPB_L) {
                    for (int l = tid; l < nlines; l += PPB_WG) {
                        int tbase, tstride, rbase, rstride;
                        pp_line_addr(g, axis, l, inv_nx, tbase, tstride, rbase, rstride);
                        pp_line<PPB_L, true> ln;
                        auto ld = [&](auto sat) {
                            ln.load([&](int i, bool) { return pp_clip(sat((int)raw[rbase + i * rstride]), cmin, cmax); }, L);
                        };
                        if (from_raw) { if (S.fast) ld(sat_fast); else ld(sat_plain); }
                        else ln.load([&](int i, bool) { return tile[tbase + i * tstride]; }, L);
                        const int f_off = rbase / g.nx * f_dsy + rbase % g.nx;
                        auto pre_flush = [&](int i) {
                            if (!flush) return;
                            const double o = tile[tbase + i * tstride];
                            const unsigned off = (unsigned)(f_off + i * f_dsz);
                            pp_store_f64(p64, off << 3, o);
                            pp_store_f32(p32, off << 2, (float)o);
                        };
                        ln.template run<0, PPB_L>(tile, tbase, tstride, L, wl, pre_flush);
                    }
                } else {
                    for (int l = tid; l < nlines; l += PPB_WG) {
                        int tbase, tstride, rbase, rstride;
                        pp_line_addr(g, axis, l, inv_nx, tbase, tstride, rbase, rstride);
                        pp_line<PP_MAXL, false> ln;
                        auto ld = [&](auto sat) {
                            ln.load([&](int i, bool) { return pp_clip(sat((int)raw[rbase + i * rstride]), cmin, cmax); }, L);
                        };
                        if (from_raw) { if (S.fast) ld(sat_fast); else ld(sat_plain); }
                        else ln.load([&](int i, bool) { return tile[tbase + i * tstride]; }, L);
                        const int f_off = rbase / g.nx * f_dsy + rbase % g.nx;
                        auto pre_flush = [&](int i) {
                            if (!flush) return;
                            const double o = tile[tbase + i * tstride];
                            const unsigned off = (unsigned)(f_off + i * f_dsz);
                            pp_store_f64(p64, off << 3, o);
                            pp_store_f32(p32, off << 2, (float)o);
                        };
                        ln.template run<0, PP_MAXL>(tile, tbase, tstride, L, wl, pre_flush);
                    }
                }
                pp_lds_barrier();
                PP_TICK(1, 2 + axis);
            }
        }
        if (!A.do_unsharp) pp_lds_barrier();          // (s_par is read below: the passes' barriers order it otherwise)
        // knife edge (|mean - erosion_threshold| <= 1e-9): NumPy's own summation order decides
        int flags = inf.flags;
        if (flags & MMX_PP_KNIFE) {
            if (tid == 0) {
                auto val = [&](int i) { return S.plain((double)(int)raw[i]); };
                const double mean = pp_pairwise(val, g.n, &s_stack) / (double)g.n;
                int fl = (flags & ~MMX_PP_KNIFE) | MMX_PP_EXACT_MEAN;
                if (mean > A.ero_thr) fl |= MMX_PP_ERODED;
                s_flags = fl;
                mmx_subblock_info o = inf;
                o.mean = mean; o.flags = fl;
                info[k] = o;
            }
            pp_lds_barrier();
            flags = s_flags;
            pp_lds_barrier();              // (s_flags is free for the next tile)
        }

        // ---- unsharp mask (+ erosion), write out; the next tile's voxels arrive meanwhile: a lane's loads of the
        // next tile are all issued first and land in the LDS words this stage has finished with.  Element-wise
        // geometry: a lane keeps its x, its rows advance by rpi = qz * ny + qy; every index is 32-bit and incremental.
        // The stage's uniform parameters come from LDS into VGPRs: as scalars they are spilled across the passes and
        // every use would re-read a spill lane.
        const int nrows = g.nrows, nx = g.nx, ny = g.ny, px = g.px;
        const int rpi = PPB_WG / nx;
        const int r_first = (int)(((float)tid + 0.5f) * inv_nx);
        const int x = tid - r_first * nx;
        const int r0 = r_first < rpi ? r_first : nrows;
        const int qz = (int)(((float)rpi + 0.5f) * inv_ny), qy = rpi - qz * ny;
        const int z0 = (int)(((float)r_first + 0.5f) * inv_ny), y0 = r_first - z0 * ny;
        const int dsz = (int)A.dst_sz, dsy = (int)A.dst_sy;
        const int out_step = qz * dsz + qy * dsy, out_wrap = dsz - ny * dsy;
        const int r_step = rpi * nx, t_step = rpi * px;
        struct walk { int row, y, out_off, ri, ti; };
        auto advance = [&](walk& w) {
            w.row += rpi; w.y += qy; w.out_off += out_step; w.ri += r_step; w.ti += t_step;
            if (w.y >= ny) { w.y -= ny; w.out_off += out_wrap; }
        };
        const walk w0 = {r0, y0, z0 * dsz + y0 * dsy + x, r0 * nx + x, r0 * px + x};
        const bool erode = flags & MMX_PP_ERODED;
        const bool fused = has_next && nsb.nz == sb.nz && nsb.ny == sb.ny && nsb.nx == sb.nx &&
                           nrows <= 3 * PPB_NB * rpi;
        // (the next tile's first pass is a blur pass over the same geometry: it can flush this tile's result)
        const bool defer = fused && S.fast && !erode && A.do_unsharp;
        const bool to_tile = erode || defer;
#ifdef PP_NOSTORE      // (timing experiment: zero-record descriptors, every store is issued and dropped)
        const pp_rsrc_t o64 = __builtin_amdgcn_make_buffer_rsrc(out64 + sb.dst_off, 0, 0, 0x00020000);
        const pp_rsrc_t o32 = __builtin_amdgcn_make_buffer_rsrc(out32 + sb.dst_off, 0, 0, 0x00020000);
#else
        const pp_rsrc_t o64 = pp_rsrc(out64 + sb.dst_off), o32 = pp_rsrc(out32 + sb.dst_off);
#endif
        auto put = [&](const walk& w, double o) {         // both output copies of one voxel (dropped past the last row)
            const unsigned off = w.row < nrows ? (unsigned)w.out_off : 0x3fffffffu;    // (x 8 and x 4 both out of range)
            pp_store_f64(o64, off << 3, o);
            pp_store_f32(o32, off << 2, (float)o);
        };
        pp_sat Sv;                                        // the same numbers as S, in vector registers
        Sv.vmin = s_par[3]; Sv.vmax = s_par[4]; Sv.span = s_par[5]; Sv.rcp = s_par[6]; Sv.identity = 0; Sv.fast = 1;
        const double v_cmin = s_par[0], v_cmax = s_par[1], v_str = s_par[2];
        // PPB_NB voxels: all LDS reads first, then the arithmetic, then LDS writes or stores
        auto batch = [&](walk& w, auto unsharp, auto totile) {
            int rv[PPB_NB];
            double bl[PPB_NB], o[PPB_NB];
            walk wr = w;
#pragma unroll
            for (int j = 0; j < PPB_NB; ++j) {
                const bool in = wr.row < nrows;
                rv[j] = (int)raw[in ? wr.ri : 0];
                if (unsharp.value) bl[j] = tile[in ? wr.ti : 0];
                advance(wr);
            }
#pragma unroll
            for (int j = 0; j < PPB_NB; ++j) {
                const double den = pp_clip(Sv((double)rv[j]), v_cmin, v_cmax);
                if (unsharp.value) {
                    const double m = v_str * bl[j];
                    const double hp = den - m;
                    o[j] = den + hp;
                } else {
                    o[j] = den;
                }
            }
#pragma unroll
            for (int j = 0; j < PPB_NB; ++j) {
                if (totile.value) { if (w.row < nrows) tile[w.ti] = o[j]; }
                else put(w, o[j]);
                advance(w);
            }
        };
        auto batches = [&](walk& w, int count) {          // `count` batches, or all rows when count < 0
            using T = std::true_type; using F = std::false_type;
            for (int b = 0; count < 0 ? w.row < nrows : b < count; ++b) {
                if (A.do_unsharp) { if (to_tile) batch(w, T{}, T{}); else batch(w, T{}, F{}); }
                else { if (to_tile) batch(w, F{}, T{}); else batch(w, F{}, F{}); }
            }
        };
        if (!S.fast) {                       // (never with integer voxels) the plain division, one voxel at a time
            for (walk w = w0; w.row < nrows; advance(w)) {
                const double den = pp_clip(sat_plain((int)raw[w.ri]), cmin, cmax);
                double o = den;
                if (A.do_unsharp) {
                    const double m = A.strength * tile[w.ti];
                    const double hp = den - m;
                    o = den + hp;
                }
                if (erode) tile[w.ti] = o;
                else put(w, o);
            }
        } else if (fused) {
            // (a lane's voxel of step j is dense index row * nx + x: consecutive lanes, consecutive voxels of the copy)
            const pp_rsrc_t nsrc = pp_rsrc(copy + copy_off[k + 1]);
            int nv[3 * PPB_NB];
            {
                walk w = w0;
#pragma unroll
                for (int j = 0; j < 3 * PPB_NB; ++j) {
                    nv[j] = (int)(unsigned)__builtin_amdgcn_raw_buffer_load_b16(
                        nsrc, w.row < nrows ? (unsigned)w.ri * 2u : PP_OOB, 0, 0);
                    advance(w);
                }
            }
            PP_TICK(1, 8);
            walk w = w0;
            batches(w, 3);
            PP_TICK(1, 9);
            w = w0;
#pragma unroll
            for (int j = 0; j < 3 * PPB_NB; ++j) {
                if (w.row < nrows) raw[w.ri] = (uint16_t)nv[j];
                advance(w);
            }
        } else {
            walk w = w0;
            batches(w, -1);
        }
        PP_TICK(1, 5);
        if (erode) {
            pp_lds_barrier();
            for (walk w = w0; w.row < nrows; advance(w)) {
                const int pos = w.ti;
                double o = tile[pos];
                if (x > 0) o = fmin(o, tile[pos - 1]);
                if (x < nx - 1) o = fmin(o, tile[pos + 1]);
                if (w.y > 0) o = fmin(o, tile[pos - px]);
                if (w.y < ny - 1) o = fmin(o, tile[pos + px]);
                if (w.row >= ny) o = fmin(o, tile[pos - ny * px]);
                if (w.row < nrows - ny) o = fmin(o, tile[pos + ny * px]);
                put(w, o);
            }
        }
        pp_lds_barrier();                 // the tile and the voxel copy are free / complete
        PP_TICK(1, 6);
        if (has_next && !(fused && S.fast)) {
            load_raw(nsb, copy_off[k + 1]);
            pp_lds_barrier();
            PP_TICK(1, 7);
        }
        flush_prev = defer;
        prev_dst = sb.dst_off;
        sb = nsb;
    }
}

}  // namespace

// Launches the kernels for tiles that qualify for the LDS path (mmx_preprocess_fast_lds != 0).
// d_info: [n_subs] records (written by the statistics kernel, completed by the blur kernel);
// d_work: mmx_preprocess_work_bytes(...) bytes: tile-major voxel copy, copy offsets, strip heads.
int64_t mmx_pp_pipe_work_bytes(const mmx_subblock* h_subs, int n_subs)
{
    int64_t vox = 0;
    for (int i = 0; i < n_subs; ++i) vox += (int64_t)h_subs[i].nz * h_subs[i].ny * h_subs[i].nx;
    return ((vox * 2 + 255) & ~255ll) + (((int64_t)n_subs * 8 + 255) & ~255ll) + (((int64_t)n_subs * 4 + 255) & ~255ll) +
           8 * PPO_WG;       // voxel copy, copy offsets, strip heads, partial sums of the offsets
}

int mmx_launch_pp_pipe(const mmx_volume* vol, const mmx_subblock* d_subs, const mmx_subblock* h_subs, int n_subs,
                       const mmx_quantile_class* d_qclasses, const double* d_weights, const pp_args& A,
                       float* d_out32, double* d_out64, mmx_subblock_info* d_info, void* d_work,
                       int tiles_per_wg, hipStream_t s)
{
    int64_t tile_b = 0, raw_b = 0, vox = 0;
    for (int i = 0; i < n_subs; ++i) {
        const mmx_subblock& b = h_subs[i];
        const int64_t n = (int64_t)b.nz * b.ny * b.nx;
        tile_b = std::max<int64_t>(tile_b, (int64_t)b.nz * b.ny * (b.nx | 1) * (int64_t)sizeof(double));
        raw_b = std::max<int64_t>(raw_b, (n * 2 + 7) & ~7ll);
        vox += n;
        // (32-bit offsets inside a tile's output extent)
        if ((int64_t)b.nz * A.dst_sz >= (1ll << 28)) return MMX_ERR_UNSUPPORTED;
    }
    if (tile_b + raw_b > MMX_PP_MAX_LDS - 1024) return MMX_ERR_UNSUPPORTED;
    if (vol->dtype != MMX_U16 && vol->dtype != MMX_U8) return MMX_ERR_UNSUPPORTED;
    if (tiles_per_wg < 1) tiles_per_wg = 1;
    uint16_t* copy = (uint16_t*)d_work;
    int64_t* copy_off = (int64_t*)((char*)d_work + ((vox * 2 + 255) & ~255ll));
    int* is_head = (int*)((char*)copy_off + (((int64_t)n_subs * 8 + 255) & ~255ll));
    int64_t* wg_sum = (int64_t*)((char*)is_head + (((int64_t)n_subs * 4 + 255) & ~255ll));
    const int n_owg = (n_subs + PPO_WG - 1) / PPO_WG;
    if (n_owg > PPO_WG) return MMX_ERR_UNSUPPORTED;
    // (32-bit byte offsets inside a strip's extent in the volume and in the copy)
    {
        const int64_t esz = vol->dtype == MMX_U16 ? 2 : 1;
        int64_t zmax = 0, ymax = 0;
        for (int i = 0; i < n_subs; ++i) { zmax = std::max<int64_t>(zmax, h_subs[i].nz); ymax = std::max<int64_t>(ymax, h_subs[i].ny); }
        const int64_t ext = (zmax * std::abs(vol->stride_z) + ymax * std::abs(vol->stride_y)) * esz;
        if (ext >= (1ll << 30) || vol->stride_z < 0 || vol->stride_y < 0 || vol->stride_x < 0 ||
            std::abs(vol->stride_z) * esz >= (1ll << 30) || std::abs(vol->stride_x) * esz * 65536 >= (1ll << 30))
            return MMX_ERR_UNSUPPORTED;
    }
    const int grid_a = ((n_subs + 7) / 8) * 8;
    const int runs = (n_subs + tiles_per_wg - 1) / tiles_per_wg;
    const int grid_b = ((runs + 7) / 8) * 8;
    const int raw_off = (int)(tile_b / (int64_t)sizeof(double));
    hipLaunchKernelGGL(pp_offsets1_kernel, dim3(n_owg), dim3(PPO_WG), 0, s, d_subs, n_subs, copy_off, wg_sum);
    hipLaunchKernelGGL(pp_offsets2_kernel, dim3(n_owg), dim3(PPO_WG), 0, s, d_subs, n_subs, vol->stride_x, copy_off,
                       (const int64_t*)wg_sum, is_head);
    if (vol->dtype == MMX_U16)
        hipLaunchKernelGGL(pp_retile_kernel<uint16_t>, dim3(n_subs), dim3(PPR_WG), 0, s, (const uint16_t*)vol->d_data,
                           vol->stride_z, vol->stride_y, vol->stride_x, d_subs, n_subs, copy_off, is_head, copy);
    else
        hipLaunchKernelGGL(pp_retile_kernel<uint8_t>, dim3(n_subs), dim3(PPR_WG), 0, s, (const uint8_t*)vol->d_data,
                           vol->stride_z, vol->stride_y, vol->stride_x, d_subs, n_subs, copy_off, is_head, copy);
    hipLaunchKernelGGL(pp_stats_kernel, dim3(grid_a), dim3(PPS_WG), 0, s, (const uint16_t*)copy,
                       (const int64_t*)copy_off, d_subs, n_subs, d_qclasses, A, d_info);
    auto kb = pp_blur_kernel;
    if (hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(tile_b + raw_b)) != hipSuccess)
        return MMX_ERR_HIP;
    hipLaunchKernelGGL(kb, dim3(grid_b), dim3(PPB_WG), (size_t)(tile_b + raw_b), s, (const uint16_t*)copy,
                       (const int64_t*)copy_off, d_subs, n_subs, tiles_per_wg, d_weights, A, d_out32, d_out64, d_info,
                       raw_off);
    return hipGetLastError() == hipSuccess ? MMX_OK : MMX_ERR_HIP;
}
