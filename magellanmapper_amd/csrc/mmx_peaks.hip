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

constexpr int kSigmaChunk = 8;   // sigma planes whose loads are issued back to back

struct peak_ctx {
    const float* base;      // slot base of sigma 0
    int64_t sigma_stride;
    int ns, nz, ny, nx, px, plane, slot;
    float thr, eps;
    mmx_cand* out;
    uint32_t cap;
    uint32_t* count;
};

// Full 80-neighbour test of one voxel that passed `v > thr - eps` (rare: inside blobs only).
__device__ __forceinline__ void check_voxel(const peak_ctx& c, int s, int idx, float v)
{
    const int z = idx / c.plane;
    const int rem = idx - z * c.plane;
    const int y = rem / c.px;
    const int x = rem - y * c.px;
    const int nz = c.nz, ny = c.ny, nx = c.nx, px = c.px, plane = c.plane;
    const float reject = v + c.eps;  // a neighbour above this rules the voxel out
    float m = -INFINITY;
    bool border = false;
    // same-sigma face neighbours first: they reject nearly every non-peak
    const float* ps = c.base + (int64_t)s * c.sigma_stride;
    if (x > 0) m = fmaxf(m, ps[idx - 1]); else border = true;
    if (x + 1 < nx) m = fmaxf(m, ps[idx + 1]); else border = true;
    if (y > 0) m = fmaxf(m, ps[idx - px]); else border = true;
    if (y + 1 < ny) m = fmaxf(m, ps[idx + px]); else border = true;
    if (z > 0) m = fmaxf(m, ps[idx - plane]); else border = true;
    if (z + 1 < nz) m = fmaxf(m, ps[idx + plane]); else border = true;
    if (m > reject) return;
    for (int ds = -1; ds <= 1; ++ds) {
        const int ss = s + ds;
        if (ss < 0 || ss >= c.ns) { border = true; continue; }
        const float* pss = c.base + (int64_t)ss * c.sigma_stride;
        for (int dz = -1; dz <= 1; ++dz) {
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
            if (m > reject) return;
        }
    }
    if (border) m = fmaxf(m, 0.0f);  // mode='constant', cval = 0
    if (!(v >= m - c.eps)) return;
    const bool contested = !(v > m + c.eps) || !(v > c.thr + c.eps);
    const uint32_t pos = atomicAdd(c.count, 1u);
    if (pos < c.cap) {
        mmx_cand r;
        r.slot = c.slot;
        r.s = s;
        r.z = z;
        r.y = y;
        r.x = x;
        r.flags = contested ? MMX_CAND_CONTESTED : 0u;
        r.v = v;
        r.nbr_max = m;
        r.v64 = __longlong_as_double(0x7ff8000000000000LL);
        r.band = 0;
        c.out[pos] = r;
    }
}

// Thread = 4 consecutive x (one 16-byte load per sigma plane; rows are 128-B aligned), all
// sigma loads of a chunk issued before any is tested.
__global__ void __launch_bounds__(MMX_WG)
peaks_kernel(const float* __restrict__ log, int ns, int64_t sigma_stride,
             const mmx_block* __restrict__ blocks, int64_t slot_elems, float thr, float eps,
             mmx_cand* __restrict__ out, uint32_t cap, uint32_t* __restrict__ count)
{
    const mmx_block bd = blocks[blockIdx.y];
    peak_ctx c;
    c.base = log + (int64_t)bd.slot * slot_elems;
    c.sigma_stride = sigma_stride;
    c.ns = ns; c.nz = bd.nz; c.ny = bd.ny; c.nx = bd.nx; c.px = bd.px;
    c.plane = bd.ny * bd.px; c.slot = bd.slot;
    c.thr = thr; c.eps = eps; c.out = out; c.cap = cap; c.count = count;
    const int nquads = bd.nz * c.plane / 4;
    const float lo = thr - eps;
    const int qrow = bd.px / 4;          // quads per row

    for (int q = blockIdx.x * MMX_WG + threadIdx.x; q < nquads; q += gridDim.x * MMX_WG) {
        const int x0 = (q % qrow) * 4;
        if (x0 >= bd.nx) continue;       // pitch columns
        const float4* p = reinterpret_cast<const float4*>(c.base) + q;
        const int64_t sstride4 = sigma_stride / 4;
        const int lane = threadIdx.x & 63;
        const bool has_left = x0 > 0 && lane > 0;            // lane - 1 holds x0-4 .. x0-1 of this row
        const bool has_right = x0 + 4 < bd.nx && lane < 63 && q + 1 < nquads;  // lane + 1: x0+4 .. x0+7
        const bool ok1 = x0 + 1 < bd.nx, ok2 = x0 + 2 < bd.nx, ok3 = x0 + 3 < bd.nx;
        for (int s0 = 0; s0 < ns; s0 += kSigmaChunk) {
            // e?[k + 1] = element ? of sigma s0 + k; index 0 and kSigmaChunk + 1 are the neighbouring
            // scales (-inf outside the ladder: zero padding can never out-vote a value > thr >= 0).
            // Plain scalar arrays with compile-time indices only: they must stay in registers.
            float e0[kSigmaChunk + 2], e1[kSigmaChunk + 2], e2[kSigmaChunk + 2], e3[kSigmaChunk + 2];
#pragma unroll
            for (int k = 0; k < kSigmaChunk + 2; ++k) {
                const int s = s0 + k - 1;
                float4 v = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
                if (s >= 0 && s < ns) v = p[(int64_t)s * sstride4];
                // pitch columns hold no data: they must never out-vote a real neighbour
                e0[k] = v.x;
                e1[k] = ok1 ? v.y : -INFINITY;
                e2[k] = ok2 ? v.z : -INFINITY;
                e3[k] = ok3 ? v.w : -INFINITY;
            }
            // In-register pre-filter: a 4-D local maximum must beat its two x neighbours and the
            // same voxel one scale up / down (all already in registers or one lane away).  Only
            // survivors pay for the remaining 76 neighbour reads.
            unsigned hits = 0;
#pragma unroll
            for (int k = 1; k <= kSigmaChunk; ++k) {
                if (s0 + k - 1 < ns) {
                    float lft = __shfl_up(e3[k], 1);
                    float rgt = __shfl_down(e0[k], 1);
                    lft = has_left ? lft : -INFINITY;
                    rgt = has_right ? rgt : -INFINITY;
                    const float n0 = fmaxf(fmaxf(lft, e1[k]), fmaxf(e0[k - 1], e0[k + 1]));
                    const float n1 = fmaxf(fmaxf(e0[k], e2[k]), fmaxf(e1[k - 1], e1[k + 1]));
                    const float n2 = fmaxf(fmaxf(e1[k], e3[k]), fmaxf(e2[k - 1], e2[k + 1]));
                    const float n3 = fmaxf(fmaxf(e2[k], rgt), fmaxf(e3[k - 1], e3[k + 1]));
                    const unsigned sh = 4 * (k - 1);
                    hits |= ((e0[k] > lo && !(n0 > e0[k] + eps)) ? 1u : 0u) << sh;
                    hits |= ((e1[k] > lo && !(n1 > e1[k] + eps)) ? 2u : 0u) << sh;
                    hits |= ((e2[k] > lo && !(n2 > e2[k] + eps)) ? 4u : 0u) << sh;
                    hits |= ((e3[k] > lo && !(n3 > e3[k] + eps)) ? 8u : 0u) << sh;
                }
            }
            while (hits) {              // rare: only inside blobs
                const int b = __ffs(hits) - 1;
                hits &= hits - 1;
                const int j = b & 3;
                if (x0 + j >= bd.nx) continue;
                const int s = s0 + (b >> 2);
                const int idx = 4 * q + j;
                check_voxel(c, s, idx, c.base[(int64_t)s * sigma_stride + idx]);
            }
        }
    }
}

// ---- sparse variant: works from the entries the Y pass of the fused path leaves per 64 columns of a row
// (y2_kernel): .x = candidate bits (above thr - eps, not beaten by the y / x neighbours), .y = "above
// thr - eps" bits.  Segments with .y == 0 were not stored: their voxels count as -inf (they can neither
// be peaks nor out-vote a candidate, which is above thr - eps itself).
struct sparse_ctx {
    peak_ctx c;
    const ulonglong2* ent;      // entries of this block, sigma 0
    int64_t ent_sigma_stride;   // entries per sigma
    int nwords;                 // entries per row y
    int quads, ntx;             // entry layout 2 (y6_kernel): one entry per 4 planes x 16 columns, ntx per plane quad
};

// entry holding voxel (z, y, x).  Layout 1 (y2_kernel): 64 consecutive columns of the (z, x) plane of row y;
// layout 2 (y6_kernel): planes 4 (z >> 2) .. + 3 x columns 16 (x >> 4) .. + 15, bit ((z & 3) << 4) | (x & 15)
__device__ __forceinline__ int64_t sparse_entry(const sparse_ctx& k, int z, int y, int x)
{
    return (int64_t)y * k.nwords + (k.quads ? (z >> 2) * k.ntx + (x >> 4) : (z * k.c.px + x) >> 6);
}

__device__ __forceinline__ float sparse_at(const sparse_ctx& k, int s, int z, int y, int x)
{
    const unsigned long long above = k.ent[(int64_t)s * k.ent_sigma_stride + sparse_entry(k, z, y, x)].y;
    if (!above) return -INFINITY;
    return k.c.base[(int64_t)s * k.c.sigma_stride + (int64_t)z * k.c.plane + y * k.c.px + x];
}

// check_voxel on the sparse cube (same decisions: a neighbour that is not stored is below thr - eps < v)
__device__ __forceinline__ void check_voxel_sparse(const sparse_ctx& k, int s, int z, int y, int x, float v)
{
    const peak_ctx& c = k.c;
    const int nz = c.nz, ny = c.ny, nx = c.nx;
    const float reject = v + c.eps;
    const float band_lo = v - c.eps;       // a neighbour below this cannot out-vote the candidate whatever the exact values
    float m = -INFINITY;
    bool border = false;
    // which of the 80 neighbours (C order of (ds, dz, dy, dx), the centre left out) lie in the band: the only ones
    // whose exact values the host needs when the candidate turns out contested
    unsigned long long band = 0;
    unsigned band_hi = 0;
    for (int ds = -1; ds <= 1; ++ds) {
        const int ss = s + ds;
        if (ss < 0 || ss >= c.ns) { border = true; continue; }
        for (int dz = -1; dz <= 1; ++dz) {
            const int zz = z + dz;
            if (zz < 0 || zz >= nz) { border = true; continue; }
            for (int dy = -1; dy <= 1; ++dy) {
                const int yy = y + dy;
                if (yy < 0 || yy >= ny) { border = true; continue; }
                // the three x neighbours share (at most two) entries and one row
                const ulonglong2* er = k.ent + (int64_t)ss * k.ent_sigma_stride;
                const float* row = c.base + (int64_t)ss * c.sigma_stride + (int64_t)zz * c.plane + yy * c.px;
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int xx = x + dx;
                    if (xx < 0 || xx >= nx) { border = true; continue; }
                    if ((ds | dz | dy | dx) == 0) continue;
                    if (er[sparse_entry(k, zz, yy, xx)].y) {
                        const float u = row[xx];
                        m = fmaxf(m, u);
                        if (u >= band_lo) {
                            int j = (((ds + 1) * 3 + (dz + 1)) * 3 + (dy + 1)) * 3 + (dx + 1);
                            j -= j > 40;
                            if (j < 64) band |= 1ull << j;
                            else band_hi |= 1u << (j - 64);
                        }
                    }
                }
            }
            if (m > reject) return;
        }
    }
    if (border) m = fmaxf(m, 0.0f);  // mode='constant', cval = 0
    if (!(v >= m - c.eps)) return;
    const bool contested = !(v > m + c.eps) || !(v > c.thr + c.eps);
    const uint32_t pos = atomicAdd(c.count, 1u);
    if (pos < c.cap) {
        mmx_cand r;
        r.slot = c.slot;
        r.s = s;
        r.z = z;
        r.y = y;
        r.x = x;
        r.flags = (contested ? MMX_CAND_CONTESTED : 0u) | MMX_CAND_BAND | (band_hi << 16);
        r.v = v;
        r.nbr_max = m;
        r.v64 = __longlong_as_double(0x7ff8000000000000LL);
        r.band = band;
        c.out[pos] = r;
    }
}

// One lane per entry and sigma: nearly all candidate words are zero; set bits are tested against z +- 1,
// then sigma +- 1 (most of them are the in-plane maxima of the z-slices and scales of a blob and fall there),
// survivors get the full 80-neighbour test.
__global__ void __launch_bounds__(MMX_WG)
peaks_sparse_kernel(const float* __restrict__ log, const ulonglong2* __restrict__ entries,
                    int ns, int64_t sigma_stride, const mmx_block* __restrict__ blocks, int64_t slot_elems,
                    float thr, float eps, mmx_cand* __restrict__ out, uint32_t cap,
                    uint32_t* __restrict__ count, int quads)
{
    const mmx_block bd = blocks[blockIdx.y];
    sparse_ctx k;
    peak_ctx& c = k.c;
    c.base = log + (int64_t)bd.slot * slot_elems;
    c.sigma_stride = sigma_stride;
    c.ns = ns; c.nz = bd.nz; c.ny = bd.ny; c.nx = bd.nx; c.px = bd.px;
    c.plane = bd.ny * bd.px; c.slot = bd.slot;
    c.thr = thr; c.eps = eps; c.out = out; c.cap = cap; c.count = count;
    const int ncol = bd.nz * bd.px;
    k.quads = quads;
    k.ntx = (bd.nx + 15) >> 4;
    k.nwords = quads ? ((bd.nz + 3) >> 2) * k.ntx : (ncol + 63) >> 6;
    k.ent = entries + ((int64_t)bd.slot * slot_elems >> 5);
    k.ent_sigma_stride = sigma_stride >> 5;
    const int64_t per_sigma = (int64_t)bd.ny * k.nwords;
    const int64_t total = per_sigma * ns;
    // Set bits are rare (a few per cent of the words hold one) and each costs a chain of dependent loads:
    // a lane that walked the bits of its own word would leave the other 63 waiting.  So the workgroup first
    // queues the set bits of 256 words in LDS, then every lane takes one queued bit.
    constexpr int kQ = 1024;
    __shared__ unsigned long long q[kQ];
    __shared__ int qn;
    auto test_bit = [&](int64_t i, int b) __attribute__((always_inline)) {
        const int s = (int)(i / per_sigma);
        const int64_t r = i - (int64_t)s * per_sigma;
        const int y = (int)(r / k.nwords);
        const int w = (int)(r - (int64_t)y * k.nwords);
        int z, x;
        if (k.quads) {
            const int zqi = w / k.ntx;
            z = 4 * zqi + (b >> 4);
            x = 16 * (w - zqi * k.ntx) + (b & 15);
        } else {
            const int col = (w << 6) + b;
            z = col / bd.px;
            x = col - z * bd.px;
        }
        if (x >= bd.nx || z >= bd.nz) return;
        const float v = c.base[(int64_t)s * sigma_stride + (int64_t)z * c.plane + y * bd.px + x];
        // (the y and x neighbours were tested when the bit was set: most set bits are the in-plane maxima of
        // the z-slices and scales of a blob and fall to z +- 1 or sigma +- 1)
        const float n0 = z > 0 ? sparse_at(k, s, z - 1, y, x) : -INFINITY;
        const float n1 = z + 1 < bd.nz ? sparse_at(k, s, z + 1, y, x) : -INFINITY;
        const float n2 = s > 0 ? sparse_at(k, s - 1, z, y, x) : -INFINITY;
        const float n3 = s + 1 < ns ? sparse_at(k, s + 1, z, y, x) : -INFINITY;
        if (fmaxf(fmaxf(n0, n1), fmaxf(n2, n3)) > v + eps) return;
        check_voxel_sparse(k, s, z, y, x, v);
    };
    if (threadIdx.x == 0) qn = 0;
    for (int j = threadIdx.x; j < kQ; j += MMX_WG) q[j] = ~0ull;     // "no item"
    __syncthreads();
    constexpr int kW = 8;                          // words per lane and round
    const int64_t stride = (int64_t)gridDim.x * MMX_WG * kW;
    const int64_t rounds = (total + stride - 1) / stride;
    for (int64_t rd = 0; rd < rounds; ++rd) {
        const int64_t i0 = rd * stride + (int64_t)blockIdx.x * MMX_WG * kW + threadIdx.x;
        unsigned long long mw[kW];
#pragma unroll
        for (int u = 0; u < kW; ++u) {
            const int64_t i = i0 + (int64_t)u * MMX_WG;
            mw[u] = 0;
            if (i < total) {
                const int s = (int)(i / per_sigma);
                mw[u] = k.ent[(int64_t)s * k.ent_sigma_stride + (i - (int64_t)s * per_sigma)].x;
            }
        }
#pragma unroll
        for (int u = 0; u < kW; ++u) {
            unsigned long long m = mw[u];
            if (!m) continue;
            const int64_t i = i0 + (int64_t)u * MMX_WG;
            const int cnt = __popcll(m);
            const int at = atomicAdd(&qn, cnt);
            if (at + cnt <= kQ) {
                int j = at;
                while (m) {
                    const int b = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    q[j++] = ((unsigned long long)i << 6) | (unsigned long long)b;
                }
            } else {                              // queue full (a dense patch): this lane walks its own bits
                while (m) {
                    const int b = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    test_bit(i, b);
                }
            }
        }
        __syncthreads();
        const int nq = qn < kQ ? qn : kQ;     // entries beyond kQ were never queued (handled inline above)
        // (an `at` below kQ with at + cnt above it queued nothing either: its slots stay unused)
        for (int j = threadIdx.x; j < nq; j += MMX_WG) {
            const unsigned long long e = q[j];
            if (e != ~0ull) test_bit((int64_t)(e >> 6), (int)(e & 63));
        }
        __syncthreads();
        if (threadIdx.x == 0) qn = 0;
        for (int j = threadIdx.x; j < nq; j += MMX_WG) q[j] = ~0ull;
        __syncthreads();
    }
}

}  // namespace

int mmx_launch_peaks_sparse(const float* d_log, const unsigned long long* d_mask, int n_sigma,
                            int64_t sigma_stride, const mmx_block* d_blocks, int n_blocks, int max_vox,
                            int64_t slot_elems, float thr, float eps, mmx_cand* d_cands, uint32_t cap,
                            uint32_t* d_count, int quads, hipStream_t stream)
{
    int64_t words = ((int64_t)max_vox / 64 + 1024) * n_sigma;
    int gx = (int)((words + MMX_WG * 8 - 1) / (MMX_WG * 8));      // 8 words per lane and round
    if (gx > 8192) gx = 8192;
    if (gx < 1) gx = 1;
    dim3 grid(gx, n_blocks);
    hipLaunchKernelGGL(peaks_sparse_kernel, grid, dim3(MMX_WG), 0, stream, d_log,
                       reinterpret_cast<const ulonglong2*>(d_mask), n_sigma, sigma_stride,
                       d_blocks, slot_elems, thr, eps, d_cands, cap, d_count, quads);
    return hipGetLastError() == hipSuccess ? MMX_OK : MMX_ERR_HIP;
}

int mmx_launch_peaks(const float* d_log, int n_sigma, int64_t sigma_stride, const mmx_block* d_blocks,
                     int n_blocks, int max_vox, int64_t slot_elems, float thr, float eps,
                     mmx_cand* d_cands, uint32_t cap, uint32_t* d_count, hipStream_t stream)
{
    int gx = (max_vox / 4 + MMX_WG - 1) / MMX_WG;
    if (gx > 8192) gx = 8192;
    if (gx < 1) gx = 1;
    dim3 grid(gx, n_blocks);
    hipLaunchKernelGGL(peaks_kernel, grid, dim3(MMX_WG), 0, stream, d_log, n_sigma, sigma_stride,
                       d_blocks, slot_elems, thr, eps, d_cands, cap, d_count);
    return hipGetLastError() == hipSuccess ? MMX_OK : MMX_ERR_HIP;
}
