// Fused Z+X pass and the matching Y pass: the same separable LoG as mmx_colpass.inc /
// mmx_xpass.hip (scipy/ndimage/_filters.py:644-707 in float32) with the Z-pass results handed
// to the X pass through LDS instead of HBM.
//
//   zx_kernel :  I (u8/u16/f32, read once)  ->  P = G(x) G(z) I
//                                               Q = G''(x) G(z) I + G(x) G''(z) I
//   y2_kernel :  P, Q                       ->  LoG = -s^2 ( G''(y) P + G(y) Q )
//
// Algorithmic HBM bytes per voxel and sigma: zx 2 + 8, y2 8 + 4  (26 with the NMS read,
// against 42 for the three separate passes).
//
// zx design (gfx950).  One workgroup per block row (z-planes x one y): px threads, lane = x.
// Each thread marches along z holding the last 2R+16 inputs of its column in a register
// window; every 8 steps the workgroup has 8 rows (8 consecutive z at this y) of Gz / Gzz in an
// LDS tile (reflect halo written by the border lanes, 2-in-8 padded rows -> conflict-free
// ds_read_b64), and after one barrier every thread takes one (row, 8-wide chunk) item of the
// X pass from a register window of 8 + 2R staged values.  The pair sums of the Gz window are
// shared by its G(x) and G''(x) products.  Results leave as 16-byte stores of whole aligned
// rows.  The loads of the next 8 planes are issued before the 8 steps and land in the tail of
// the register window, which is then shifted down by 8.

#include <type_traits>

#include "mmx_common.h"

namespace {

constexpr int kG = 8;   // z steps (rows) per X batch
constexpr int kT = 8;   // X outputs per thread

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
__device__ __forceinline__ int pad2(int i) { return i + 2 * (i >> 3); }

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

template <int R> struct xgeom {
    static constexpr int LEAD = R & 1;                 // odd radius: window starts one float early
    static constexpr int S = (R + LEAD + 7) & ~7;      // staged position of x = 0
    static constexpr int WIN = kT + 2 * R + 2 * LEAD;  // floats read per thread and array (even: read in pairs)
    // LDS row pitch, a compile-time constant (sized for the widest supported row, px = 512) so that
    // every row offset folds into the ds_* immediate field instead of living in a VGPR
    static constexpr int SPAN = S + 512 + R + LEAD;
    static constexpr int PITCH = ((SPAN + 2 * (SPAN >> 3)) + 3) & ~1;
};

template <int R, typename InT>
__global__ void __launch_bounds__(512)
zx_kernel(const InT* __restrict__ vol, int64_t stride_z, int stride_y, int stride_x,
          const mmx_block* __restrict__ blocks, int64_t slot_elems,
          float* __restrict__ gp, float* __restrict__ gq, mmx_taps_f32 tz, mmx_taps_f32 tx)
{
    using io = vox<InT>;
    using xg = xgeom<R>;
    constexpr int NW = 2 * R + 2 * kG;   // register window: inputs z0-R .. z0+R+2G-1
    extern __shared__ float lds[];
    const mmx_block bd = blocks[blockIdx.y];
    const int y = blockIdx.x;
    if (y >= bd.ny) return;              // whole workgroup
    const int W = bd.nx, px = bd.px, nz = bd.nz;
    const int t = threadIdx.x;
    const bool lane_on = t < px;         // threads beyond the pitch only take part in barriers
    const int xl = t < W ? t : W - 1;    // pitch lanes re-read the last column
    constexpr int PW = xg::PITCH;
    float* la = lds;                     // Gz rows   [kG][PW]
    float* lb = lds + kG * PW;           // Gzz rows  [kG][PW]
    const InT* in = vol + bd.src_off + (int64_t)y * stride_y;
    const unsigned voff = (unsigned)(xl * stride_x) * (unsigned)sizeof(InT);
    const int64_t sbase = (int64_t)bd.slot * slot_elems + (int64_t)y * px;
    const int64_t plane = (int64_t)bd.ny * px;

    float w[NW];
    if (lane_on) {
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const float raw = io::load(make_rsrc(in + (int64_t)reflect_clamped(i - R, nz) * stride_z), voff);
            w[i] = i < 2 * R + kG ? io::act(raw) : raw;   // the tail is "in flight"
        }
    }
    // X-phase item of this thread
    const int CH = px / kT;
    const int xr = t / CH;               // row of the tile (0..kG-1 when lane_on)
    const int xc = t - xr * CH;          // chunk

#pragma unroll 1
    for (int z0 = 0; z0 < nz; z0 += kG) {
        if (lane_on) {
            // ---- Z phase: 8 steps from the register window, rows into the LDS tile
#pragma unroll
            for (int s = 0; s < kG; ++s) {
                const float c = w[R + s];
                float a0 = c * tz.w0[0];
                float a2 = c * tz.w2[0];
#ifndef ZX_SKIP_Z
#pragma unroll
                for (int k = 1; k <= R; ++k) {
                    const float p = w[R + s - k] + w[R + s + k];
                    a0 = fmaf(p, tz.w0[k], a0);
                    a2 = fmaf(p, tz.w2[k], a2);
                }
#endif
                if (t < W) {
                    float* ra = la + s * PW;
                    float* rb = lb + s * PW;
                    const int q = pad2(xg::S + t);
                    ra[q] = a0;
                    rb[q] = a2;
                    if (t < R) {                       // left halo: x = -1-t  <-  x = t
                        const int h = pad2(xg::S - 1 - t);
                        ra[h] = a0;
                        rb[h] = a2;
                    }
                    if (t >= W - R) {                  // right halo: x = W+j  <-  x = W-1-j
                        const int h = pad2(xg::S + W + (W - 1 - t));
                        ra[h] = a0;
                        rb[h] = a2;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        if (lane_on) {
            // ---- shift the window by 8 (the tail holds the loads issued one group ago) and issue
            // the loads of the group after next into the freed tail
            // (in place, ascending: each move reads a register that is overwritten only later;
            // volatile asm pins that order so the allocator does not double-buffer the window)
#pragma unroll
            for (int i = 0; i < 2 * R + kG; ++i) {
                if (i < 2 * R) asm volatile("v_mov_b32 %0, %1" : "=v"(w[i]) : "v"(w[i + kG]));
                else w[i] = io::act(w[i + kG]);
            }
#pragma unroll
            for (int j = 0; j < kG; ++j)
                w[2 * R + kG + j] = io::load(
                    make_rsrc(in + (int64_t)reflect_clamped(z0 + 2 * kG + R + j, nz) * stride_z), voff);
            __builtin_amdgcn_sched_barrier(0);
            // ---- X phase: one (row, chunk) item per thread
#ifndef ZX_SKIP_X
            if (z0 + xr < nz) {
                // Register diet (the z window stays live): P leaves in two halves as soon as it is
                // complete, the Gzz window reuses the Gz window's registers, and scheduling
                // barriers keep the compiler from overlapping the stages.
                float Q[kT];
                float win[xg::WIN];
                const int base = xg::S - R - xg::LEAD + xc * kT;      // even
                const int64_t o = sbase + (int64_t)(z0 + xr) * plane + xc * kT;
                float4* dp = reinterpret_cast<float4*>(gp + o);
                float4* dq = reinterpret_cast<float4*>(gq + o);
                const float* pa = la + xr * PW;
#pragma unroll
                for (int i = 0; i < xg::WIN; i += 2) {
                    const float2 v = *reinterpret_cast<const float2*>(pa + pad2(base + i));
                    win[i] = v.x;
                    win[i + 1] = v.y;
                }
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    float P[kT / 2];
#pragma unroll
                    for (int oo = 0; oo < kT / 2; ++oo) {
                        const int o8 = half * (kT / 2) + oo;
                        const float c = win[xg::LEAD + o8 + R];
                        float p = c * tx.w0[0];
                        float q = c * tx.w2[0];
#pragma unroll
                        for (int k = 1; k <= R; ++k) {
                            const float sm = win[xg::LEAD + o8 + R - k] + win[xg::LEAD + o8 + R + k];
                            p = fmaf(sm, tx.w0[k], p);
                            q = fmaf(sm, tx.w2[k], q);
                        }
                        P[oo] = p;
                        Q[o8] = q;
                    }
#ifdef ZX_NO_STORE
                    asm volatile("" ::"v"(P[0]), "v"(P[1]), "v"(P[2]), "v"(P[3]));
#else
                    dp[half] = make_float4(P[0], P[1], P[2], P[3]);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
                const float* pb = lb + xr * PW;
#pragma unroll
                for (int i = 0; i < xg::WIN; i += 2) {
                    const float2 v = *reinterpret_cast<const float2*>(pb + pad2(base + i));
                    win[i] = v.x;
                    win[i + 1] = v.y;
                }
#pragma unroll
                for (int o8 = 0; o8 < kT; ++o8) {
                    float q = fmaf(win[xg::LEAD + o8 + R], tx.w0[0], Q[o8]);
#pragma unroll
                    for (int k = 1; k <= R; ++k)
                        q = fmaf(win[xg::LEAD + o8 + R - k] + win[xg::LEAD + o8 + R + k], tx.w0[k], q);
                    Q[o8] = q;
                }
#ifdef ZX_NO_STORE
                asm volatile("" ::"v"(Q[0]), "v"(Q[1]), "v"(Q[2]), "v"(Q[3]), "v"(Q[4]), "v"(Q[5]), "v"(Q[6]), "v"(Q[7]));
#else
                dq[0] = make_float4(Q[0], Q[1], Q[2], Q[3]);
                dq[1] = make_float4(Q[4], Q[5], Q[6], Q[7]);
#endif
            }
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
}

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
int launch_zx(const mmx_volume* vol, const mmx_block* d_blocks, int n_blocks, int max_ny, int max_px,
              int64_t slot_elems, const mmx_taps_f32& tz, const mmx_taps_f32& tx, float* d_p, float* d_q,
              hipStream_t s)
{
    const int threads = (max_px + 63) / 64 * 64;
    if (threads > 512) return MMX_ERR_UNSUPPORTED;
    const size_t lds = (size_t)2 * kG * xgeom<R>::PITCH * sizeof(float);
    if (lds > 64 * 1024) return MMX_ERR_UNSUPPORTED;
    dim3 grid(max_ny, n_blocks);
    const int sy = (int)vol->stride_y, sx = (int)vol->stride_x;
    if (vol->dtype == MMX_U16)
        hipLaunchKernelGGL((zx_kernel<R, uint16_t>), grid, dim3(threads), lds, s, (const uint16_t*)vol->d_data,
                           vol->stride_z, sy, sx, d_blocks, slot_elems, d_p, d_q, tz, tx);
    else if (vol->dtype == MMX_F32)
        hipLaunchKernelGGL((zx_kernel<R, float>), grid, dim3(threads), lds, s, (const float*)vol->d_data,
                           vol->stride_z, sy, sx, d_blocks, slot_elems, d_p, d_q, tz, tx);
    else if (vol->dtype == MMX_U8)
        hipLaunchKernelGGL((zx_kernel<R, uint8_t>), grid, dim3(threads), lds, s, (const uint8_t*)vol->d_data,
                           vol->stride_z, sy, sx, d_blocks, slot_elems, d_p, d_q, tz, tx);
    else
        return MMX_ERR_UNSUPPORTED;
    return hipGetLastError() == hipSuccess ? MMX_OK : MMX_ERR_HIP;
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

}  // namespace

#define MMX_FOR_EACH_RADIUS(X) \
    X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) \
    X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24)

int mmx_launch_zx(const mmx_volume* vol, const mmx_block* d_blocks, int n_blocks, int max_ny, int max_px,
                  int64_t slot_elems, const mmx_taps_f32& tz, const mmx_taps_f32& tx, int radius,
                  float* d_p, float* d_q, hipStream_t stream)
{
    switch (radius) {
#define X(R) case R: return launch_zx<R>(vol, d_blocks, n_blocks, max_ny, max_px, slot_elems, tz, tx, d_p, d_q, stream);
        MMX_FOR_EACH_RADIUS(X)
#undef X
        default: return MMX_ERR_UNSUPPORTED;
    }
}

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
