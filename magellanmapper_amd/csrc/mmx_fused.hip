// Y pass of the fused LoG path: the same separable LoG as mmx_colpass.inc / mmx_xpass.hip
// (scipy/ndimage/_filters.py:644-707 in float32) on the (P, Q) pair a fused Z+X kernel leaves
// (mmx_fused2.hip: packed VALU math, mmx_fused4.hip: matrix cores):
//
//   y2_kernel :  P = G(x) G(z) I,  Q = G''(x) G(z) I + G(x) G''(z) I   ->   LoG = -s^2 ( G''(y) P + G(y) Q )
//
// plus the NMS pre-filter entries and the sparse store (see the kernel).  Algorithmic HBM bytes per voxel
// and sigma: Z+X 2 + 8, Y 8 + 4 (26 with the NMS read, against 42 for the three separate passes).
// (The first fused Z+X kernel, one set of waves alternating between the z march and the x pass, lived here
// in round 1; it was 1.3-2 x slower than its successor and has been removed.)

#include <type_traits>

#include "mmx_common.h"

namespace {

__device__ __forceinline__ int reflect_once(int i, int n)
{
    i = i < 0 ? -1 - i : i;
    return i >= n ? 2 * n - 1 - i : i;
}
// reflect, then clamp: prefetches run up to 2G planes past the last needed one
__device__ __forceinline__ int reflect_clamped(int i, int n)
{
    i = reflect_once(i, n);
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
template <typename T> struct vox;
template <> struct vox<uint8_t> {
    static __device__ __forceinline__ float load(rsrc_t r, unsigned o) { return __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b8(r, o, 0, 0)); }
    static __device__ __forceinline__ float act(float raw) { return (float)__float_as_uint(raw); }
};
template <> struct vox<uint16_t> {
    static __device__ __forceinline__ float load(rsrc_t r, unsigned o) { return __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b16(r, o, 0, 0)); }
    static __device__ __forceinline__ float act(float raw) { return (float)__float_as_uint(raw); }
};
template <> struct vox<float> {
    static __device__ __forceinline__ float load(rsrc_t r, unsigned o) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, o, 0, 0)); }
    static __device__ __forceinline__ float act(float raw) { return raw; }
};

// ---------------------------------------------------------------- Y pass on (P, Q)
#ifndef Y2_PF
#define Y2_PF 4
#endif
constexpr int kPrefetch = Y2_PF;
typedef float v2f __attribute__((ext_vector_type(2)));
// (G''(y), G(y)) weights as pairs: the ring holds (P, Q) pairs, so one v_pk_add_f32 forms both pair sums of
// a tap and one v_pk_fma_f32 accumulates (G''(y) P, G(y) Q); the two halves are added once per output.
struct y2_taps {
    v2f w[MMX_MAX_RADIUS_FAST + 1];
};

template <int R, bool MASK>
__global__ void __launch_bounds__(MMX_WG)
y2_kernel(const mmx_block* __restrict__ blocks, int64_t slot_elems,
          const float* __restrict__ gp, const float* __restrict__ gq,
          float* __restrict__ out, y2_taps taps,
          unsigned long long* __restrict__ mask, float nms_lo, float nms_eps)
{
    using io = vox<float>;
    constexpr int N = 2 * R + 1;
    constexpr int M = N + kPrefetch;
    const mmx_block bd = blocks[blockIdx.y];
    const int ncol = bd.nz * bd.px;
    const int col = blockIdx.x * MMX_WG + threadIdx.x;
    if (col >= ncol) return;
    const int z = col / bd.px;
    const int x = col - z * bd.px;
    const int n = bd.ny;
    const int nx = bd.px;
    const unsigned voff = (unsigned)(z * bd.ny * nx + x) * 4u;
    const int64_t sbase = (int64_t)bd.slot * slot_elems;
    const float* i1 = gp + sbase;
    const float* i2 = gq + sbase;
    const float* w1 = out + sbase;

    // NMS pre-filter (optional): per 64 columns of a row one entry of two 64-bit words,
    //   .x  one bit per voxel = "above thr - eps and not beaten by more than eps by its y and (same wave) x
    //       neighbours" -- a superset of the local maxima, decided on the very float32 values computed here:
    //       the NMS kernel only visits these bits (mmx_peaks.hip);
    //   .y  one bit per voxel = "above thr - eps".  A 64-voxel segment without such a voxel (three quarters
    //       of them on the benchmark volume) is NOT STORED: a value below the threshold can neither be a peak
    //       nor out-vote a candidate, so the NMS kernel reads a neighbour only where .y != 0.
    const int lane = threadIdx.x & 63;
    const int nwords = (ncol + 63) >> 6;
    ulonglong2* mrow = MASK ? reinterpret_cast<ulonglong2*>(mask) + ((int64_t)bd.slot * slot_elems >> 5) + (col >> 6)
                            : nullptr;      // entry of the row being decided (advanced by nwords per row)
    unsigned long long ab_prev = 0;
    const bool real = x < bd.nx;
    const bool has_l = lane > 0 && x > 0, has_r = lane < 63 && x + 1 < bd.nx && col + 1 < ncol;
    float prev1 = -INFINITY, prev2 = -INFINITY, nbx_prev = -INFINITY;
    int ydone = 0;       // outputs produced so far

    v2f r[M];     // (P, Q) window
#pragma unroll
    for (int j = -R; j < R + kPrefetch; ++j) {
        const int row = reflect_once(j, n) * nx;
        r[(j + M) % M] = (v2f){io::load(make_rsrc(i1 + row), voff), io::load(make_rsrc(i2 + row), voff)};
    }
    // one descriptor per array for the whole march; the row is a scalar byte offset (one s_add per step and
    // array instead of rebuilding a descriptor: at R >= 18 the kernel is bound by instruction issue)
    const rsrc_t rs1 = make_rsrc(i1), rs2 = make_rsrc(i2), rsw = make_rsrc(w1);
    const unsigned row_b = (unsigned)nx * 4u;
    unsigned qoff = (unsigned)(R + kPrefetch) * row_b;      // next row to load (bytes)
    unsigned woff = 0;                                      // row being written
    auto step = [&](int s, auto reflecting, int y) __attribute__((always_inline)) {
        float n1, n2;
        if constexpr (decltype(reflecting)::value) {
            const unsigned rnext = (unsigned)reflect_once(y + R + kPrefetch, n) * row_b;
            n1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs1, voff, rnext, 0));
            n2 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs2, voff, rnext, 0));
        } else {
            n1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs1, voff, qoff, 0));
            n2 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs2, voff, qoff, 0));
        }
        qoff += row_b;
        // two accumulator chains: a dependent v_pk_add -> v_pk_fma pair back to back costs a wait state
        v2f a2 = r[s] * taps.w[0];
        v2f b2 = (r[(s - 1 + M) % M] + r[(s + 1) % M]) * taps.w[1];
#pragma unroll
        for (int k = 2; k <= R; k += 2) {
            const v2f sa = r[(s - k + M) % M] + r[(s + k) % M];
            a2 = __builtin_elementwise_fma(sa, taps.w[k], a2);
            if (k + 1 <= R) {
                const v2f sb = r[(s - k - 1 + M) % M] + r[(s + k + 1) % M];
                b2 = __builtin_elementwise_fma(sb, taps.w[k + 1], b2);
            }
        }
        a2 += b2;
        const float acc = a2.x + a2.y;
        const unsigned long long ab = MASK ? __ballot(real & (acc > nms_lo)) : ~0ull;
        if (ab) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc), rsw, voff, woff, 0);
        woff += row_b;
        r[(s + R + kPrefetch) % M] = (v2f){n1, n2};
        if constexpr (MASK) {
            // x neighbours by DPP wavefront shifts (no LDS crossbar traffic); lanes shifted in from
            // outside the wave keep `acc`, which has_l / has_r discard
            const float l = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(
                (int)__float_as_uint(acc), (int)__float_as_uint(acc), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
            const float r = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(
                (int)__float_as_uint(acc), (int)__float_as_uint(acc), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
            const float nbx = fmaxf(has_l ? l : -INFINITY, has_r ? r : -INFINITY);
            if (ydone > 0) {      // decide row ydone - 1, now that its successor is known
                const bool cand = real & (prev1 > nms_lo) &
                                  !(fmaxf(fmaxf(prev2, acc), nbx_prev) > prev1 + nms_eps);
                const unsigned long long m = __ballot(cand);
                if (lane == 0) *mrow = make_ulonglong2(m, ab_prev);
                mrow += nwords;
            }
            prev2 = prev1; prev1 = acc; nbx_prev = nbx;
            ab_prev = ab;
            ++ydone;
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    int y0 = 0;
#pragma unroll 1
    for (; y0 + M + R + kPrefetch <= n; y0 += M) {
#pragma unroll
        for (int s = 0; s < M; ++s) step(s, std::false_type{}, 0);
    }
#pragma unroll 1
    for (; y0 < n; y0 += M) {
#pragma unroll
        for (int s = 0; s < M; ++s) {
            if (y0 + s >= n) break;
            step(s, std::true_type{}, y0 + s);
        }
    }
    if constexpr (MASK) {     // the last row has no successor
        const bool cand = real & (prev1 > nms_lo) & !(fmaxf(prev2, nbx_prev) > prev1 + nms_eps);
        const unsigned long long m = __ballot(cand);
        if (lane == 0) *mrow = make_ulonglong2(m, ab_prev);
    }
}

template <int R>
int launch_y2(const mmx_block* d_blocks, int n_blocks, int max_cols, int64_t slot_elems,
              const mmx_taps_f32& taps, const float* d_p, const float* d_q, float* d_log,
              unsigned long long* d_mask, float nms_lo, float nms_eps, hipStream_t s)
{
    dim3 grid((max_cols + MMX_WG - 1) / MMX_WG, n_blocks);
    y2_taps pk;
    for (int k = 0; k <= R; ++k) pk.w[k] = (v2f){taps.w2[k], taps.w0[k]};
    if (d_mask)
        hipLaunchKernelGGL((y2_kernel<R, true>), grid, dim3(MMX_WG), 0, s, d_blocks, slot_elems, d_p, d_q, d_log, pk,
                           d_mask, nms_lo, nms_eps);
    else
        hipLaunchKernelGGL((y2_kernel<R, false>), grid, dim3(MMX_WG), 0, s, d_blocks, slot_elems, d_p, d_q, d_log, pk,
                           d_mask, nms_lo, nms_eps);
    return hipGetLastError() == hipSuccess ? MMX_OK : MMX_ERR_HIP;
}

// ---------------------------------------------------------------- Y pass on tiled (P, Q)  (zx_mode 6)
// Same arithmetic, other geometry: P and Q arrive as 16 z x 16 x tiles of 1 KiB in (y, c, U) order
// (mmx_fused4.hip: zx4_kernel).  A workgroup owns one (column tile c, z tile U); each of its four waves
// marches along y over 4 planes x 16 columns of it: one contiguous 256-byte piece per array and step, the next
// one ntx ntz KiB further on -- at any time the workgroups of a block read inside the same few hundred KiB.  The LoG cube stays row-major (the NMS and re-scoring kernels
// read single voxels from it): a wave stores four 64-byte row pieces, and only where something is above the
// threshold.  NMS entries: per row y one entry per (4 planes, 16 columns) = this wave's footprint,
// index y * (nzq * ntx) + (z >> 2) * ntx + (x >> 4), bit ((z & 3) << 4) | (x & 15)  (mmx_peaks.hip: layout 2).
// Q16: one dword per voxel in the tile, P = unorm16 (low half), Q = snorm16 (high half); their scales ride in the
// weights.  Half the bytes and half the load instructions.
template <int R, bool MASK, bool Q16>
__global__ void __launch_bounds__(MMX_WG)
y6_kernel(const mmx_block* __restrict__ blocks, int64_t slot_elems, int64_t tile_stride,
          const float* __restrict__ gp, const float* __restrict__ gq,
          float* __restrict__ out, y2_taps taps,
          unsigned long long* __restrict__ mask, float nms_lo, float nms_eps)
{
    constexpr int N = 2 * R + 1;
    constexpr int M = N + kPrefetch;
    const mmx_block bd = blocks[blockIdx.y];
    const int ntx = (bd.nx + 15) >> 4, ntz = (bd.nz + 15) >> 4;
    const int tile = blockIdx.x;
    if (tile >= ntx * ntz) return;
    const int c = tile / ntz, U = tile - c * ntz;
    const int zq = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (16 * U + 4 * zq >= bd.nz) return;                       // whole wave past the block (no barriers here)
    const int lane = threadIdx.x & 63;
    const int xi = lane & 15;
    const int z = 16 * U + 4 * zq + (lane >> 4), x = 16 * c + xi;
    const int n = bd.ny;
    const bool real = z < bd.nz && x < bd.nx;
    const float* i1 = gp + (int64_t)bd.slot * tile_stride + (int64_t)tile * 256;
    const float* i2 = gq + (int64_t)bd.slot * tile_stride + (int64_t)tile * 256;
    const unsigned trow_b = (unsigned)(ntx * ntz) * 1024u;      // bytes from one y to the next
    float* w1 = out + (int64_t)bd.slot * slot_elems;
    const unsigned voff = (unsigned)(zq * 64 + lane) * 4u;                          // inside a tile
    // (lanes past the block keep an in-range address: they never store)
    const unsigned ooff = (unsigned)((real ? z : 0) * bd.ny * bd.px + (real ? x : 0)) * 4u;   // row 0 of this voxel's column
    const int nent = ((bd.nz + 3) >> 2) * ntx;
    ulonglong2* mrow = MASK ? reinterpret_cast<ulonglong2*>(mask) + ((int64_t)bd.slot * slot_elems >> 5) +
                              (4 * U + zq) * ntx + c
                            : nullptr;
    unsigned long long ab_prev = 0;
    const bool has_l = xi > 0, has_r = xi < 15 && x + 1 < bd.nx;
    float prev1 = -INFINITY, prev2 = -INFINITY, nbx_prev = -INFINITY;
    int ydone = 0;

    const rsrc_t rs1 = make_rsrc(i1), rs2 = make_rsrc(i2), rsw = make_rsrc(w1);
    auto unpack = [](unsigned d) __attribute__((always_inline)) {
        return (v2f){(float)(d & 0xffffu), (float)((int)d >> 16)};
    };
    v2f r[M];     // (P, Q) window
#pragma unroll
    for (int j = -R; j < R + kPrefetch; ++j) {
        const unsigned row = (unsigned)reflect_once(j, n) * trow_b;
        if constexpr (Q16)
            r[(j + M) % M] = unpack(__builtin_amdgcn_raw_buffer_load_b32(rs1, voff, row, 0));
        else
            r[(j + M) % M] = (v2f){__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs1, voff, row, 0)),
                                   __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs2, voff, row, 0))};
    }
    const unsigned row_b = (unsigned)bd.px * 4u;
    unsigned qoff = (unsigned)(R + kPrefetch) * trow_b;     // next tile to load (bytes)
    unsigned woff = 0;                                      // row being written
    auto step = [&](int s, auto reflecting, int y) __attribute__((always_inline)) {
        float n1, n2 = 0.f;
        if constexpr (decltype(reflecting)::value) {
            const unsigned rnext = (unsigned)reflect_once(y + R + kPrefetch, n) * trow_b;
            n1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs1, voff, rnext, 0));
            if constexpr (!Q16) n2 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs2, voff, rnext, 0));
        } else {
            n1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs1, voff, qoff, 0));
            if constexpr (!Q16) n2 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs2, voff, qoff, 0));
        }
        qoff += trow_b;
        v2f a2 = r[s] * taps.w[0];
        v2f b2 = (r[(s - 1 + M) % M] + r[(s + 1) % M]) * taps.w[1];
#pragma unroll
        for (int k = 2; k <= R; k += 2) {
            const v2f sa = r[(s - k + M) % M] + r[(s + k) % M];
            a2 = __builtin_elementwise_fma(sa, taps.w[k], a2);
            if (k + 1 <= R) {
                const v2f sb = r[(s - k - 1 + M) % M] + r[(s + k + 1) % M];
                b2 = __builtin_elementwise_fma(sb, taps.w[k + 1], b2);
            }
        }
        a2 += b2;
        const float acc = a2.x + a2.y;
        const unsigned long long ab = MASK ? __ballot(real & (acc > nms_lo)) : ~0ull;
        if (ab && real) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc), rsw, ooff, woff, 0);
        woff += row_b;
        if constexpr (Q16) r[(s + R + kPrefetch) % M] = unpack(__float_as_uint(n1));
        else r[(s + R + kPrefetch) % M] = (v2f){n1, n2};
        if constexpr (MASK) {
            // x neighbours inside the 16-lane rows (DPP row shifts); lanes with no source keep `acc`, which
            // has_l / has_r discard: the first and last column of a tile are not tested against the next tile
            // (three quarters of the footprints hold nothing above the threshold: the neighbour values of such a row are
            //  never looked at, and a row with nothing above it has no candidates -- wave-uniform branches)
            float nbx = -INFINITY;
            if (ab) {
                const float l = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(
                    (int)__float_as_uint(acc), (int)__float_as_uint(acc), 0x111 /* row_shr:1 */, 0xf, 0xf, false));
                const float rr = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(
                    (int)__float_as_uint(acc), (int)__float_as_uint(acc), 0x101 /* row_shl:1 */, 0xf, 0xf, false));
                nbx = fmaxf(has_l ? l : -INFINITY, has_r ? rr : -INFINITY);
            }
            if (ydone > 0) {      // decide row ydone - 1, now that its successor is known
                unsigned long long m = 0;
                if (ab_prev) {
                    const bool cand = real & (prev1 > nms_lo) &
                                      !(fmaxf(fmaxf(prev2, acc), nbx_prev) > prev1 + nms_eps);
                    m = __ballot(cand);
                }
                if (lane == 0) *mrow = make_ulonglong2(m, ab_prev);
                mrow += nent;
            }
            prev2 = prev1; prev1 = acc; nbx_prev = nbx;
            ab_prev = ab;
            ++ydone;
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    int y0 = 0;
#pragma unroll 1
    for (; y0 + M + R + kPrefetch <= n; y0 += M) {
#pragma unroll
        for (int s = 0; s < M; ++s) step(s, std::false_type{}, 0);
    }
#pragma unroll 1
    for (; y0 < n; y0 += M) {
#pragma unroll
        for (int s = 0; s < M; ++s) {
            if (y0 + s >= n) break;
            step(s, std::true_type{}, y0 + s);
        }
    }
    if constexpr (MASK) {     // the last row has no successor
        const bool cand = real & (prev1 > nms_lo) & !(fmaxf(prev2, nbx_prev) > prev1 + nms_eps);
        const unsigned long long m = __ballot(cand);
        if (lane == 0) *mrow = make_ulonglong2(m, ab_prev);
    }
}

template <int R>
int launch_y6(const mmx_block* d_blocks, int n_blocks, const mmx_zx6_plan& plan, int64_t slot_elems,
              const mmx_taps_f32& taps, const float* d_p, const float* d_q, float cp, float cq, float* d_log,
              unsigned long long* d_mask, float nms_lo, float nms_eps, hipStream_t s)
{
    dim3 grid(plan.max_tiles, n_blocks);
    y2_taps pk;
    const bool q16 = cp > 0.f;
    for (int k = 0; k <= R; ++k) pk.w[k] = q16 ? (v2f){taps.w2[k] * cp, taps.w0[k] * cq} : (v2f){taps.w2[k], taps.w0[k]};
#define MMX_Y6_LAUNCH(MK, QQ) \
    hipLaunchKernelGGL((y6_kernel<R, MK, QQ>), grid, dim3(MMX_WG), 0, s, d_blocks, slot_elems, plan.tile_stride, d_p, d_q, \
                       d_log, pk, d_mask, nms_lo, nms_eps)
    if (d_mask) { if (q16) MMX_Y6_LAUNCH(true, true); else MMX_Y6_LAUNCH(true, false); }
    else        { if (q16) MMX_Y6_LAUNCH(false, true); else MMX_Y6_LAUNCH(false, false); }
#undef MMX_Y6_LAUNCH
    return hipGetLastError() == hipSuccess ? MMX_OK : MMX_ERR_HIP;
}

}  // namespace

#define MMX_FOR_EACH_RADIUS(X) \
    X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) \
    X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24)

int mmx_launch_y2(const mmx_block* d_blocks, int n_blocks, int max_cols, int64_t slot_elems,
                  const mmx_taps_f32& taps, int radius, const float* d_p, const float* d_q,
                  float* d_log, unsigned long long* d_mask, float nms_lo, float nms_eps, hipStream_t stream)
{
    switch (radius) {
#define X(R) case R: return launch_y2<R>(d_blocks, n_blocks, max_cols, slot_elems, taps, d_p, d_q, d_log, d_mask, nms_lo, nms_eps, stream);
        MMX_FOR_EACH_RADIUS(X)
#undef X
        default: return MMX_ERR_UNSUPPORTED;
    }
}

int mmx_launch_y6(const mmx_block* d_blocks, int n_blocks, const mmx_zx6_plan& plan, int64_t slot_elems,
                  const mmx_taps_f32& taps, int radius, const float* d_p, const float* d_q, float cp, float cq,
                  float* d_log, unsigned long long* d_mask, float nms_lo, float nms_eps, hipStream_t stream)
{
    switch (radius) {
#define X(R) case R: return launch_y6<R>(d_blocks, n_blocks, plan, slot_elems, taps, d_p, d_q, cp, cq, d_log, d_mask, nms_lo, nms_eps, stream);
        MMX_FOR_EACH_RADIUS(X)
#undef X
        default: return MMX_ERR_UNSUPPORTED;
    }
}
