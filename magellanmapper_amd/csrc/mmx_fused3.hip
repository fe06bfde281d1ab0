// Fused Z+X pass, third design: Z on the vector ALUs, X on the matrix cores (gfx950).
//
//   zx3_kernel :  I (u8/u16/f32, read once)  ->  P = G(x) G(z) I
//                                                Q = G''(x) G(z) I + G(x) G''(z) I
// feeding y2_kernel (mmx_fused.hip) exactly as zx2_kernel (mmx_fused2.hip) does.
//
// Why.  zx2_kernel is bound by VALU issue (5R packed instructions per voxel), not by
// HBM, and gfx950 has a second arithmetic pipe that the stencil left idle: v_mfma_f32_16x16x4_f32 multiplies
// exact float32 (a k-ordered fmaf chain, no reduced precision) at the vector rate and issues beside VALU
// work of other waves.  The X pass of a tile of 16 rows is a product with a banded Toeplitz matrix:
//     out[x_out][row] = sum_{x_in} W[x_out][x_in] * in[x_in][row],   W[x_out][x_in] = w[|x_in - x_out|]
// so per 16 x 16 output tile and product the consumers issue K/4 MFMAs with K = 16 + 2R inputs
// (R = 16: 12 for P, 24 for Q; 69 % of the multiplies hit the band, the rest are zeros of the Toeplitz
// corners), while the producer waves keep the Z pass (2R + 1 VALU instructions per voxel, v_pk_fma_f32 for
// both derivative orders) on the vector pipe of the same SIMDs.
//
// Layout.  A workgroup is one block row (all x of one y) marching along z in groups of kG3 = 16 planes:
//   * producer waves (lane = x; a tail wave with lane = (plane, column) for widths just past a multiple
//     of 64, as in zx2_kernel) write Gz and Gzz rows PLANAR into a double-buffered LDS tile
//     [buffer][Gz | Gzz][16 rows][PW], reflect halos included;
//   * consumer wave c takes the 16-column output tiles t = c, c + NC, ...: per k-step one ds_read_b32 per
//     array gives the B operand (lane l: row l & 15, x_in = xo - R + 4s + (l >> 4); PW = 2 mod 32 makes the
//     32 lanes of a read group hit 32 banks), the A operands are the Toeplitz fragments, 2 * NS constant
//     registers built once per wave.  The accumulators come out with the row (z) on lane & 15 and four
//     consecutive x_out in the four registers: one 16-byte store per lane for P and for Q.
//   * three independent accumulators (P, G''(x) Gz, G(x) Gzz) keep dependent MFMAs three issues apart.
// LDS that is never written stays zero (cleared once): zero weights times stale finite values are harmless,
// zero times an uninitialised NaN would not be.

#include <type_traits>

#include "mmx_common.h"

typedef float v2f3 __attribute__((ext_vector_type(2)));
typedef float v4f3 __attribute__((ext_vector_type(4)));

struct mmx_taps_zx3 {
    v2f3 zw[MMX_MAX_RADIUS_FAST + 1];     // (w0z[k], w2z[k])
    float xw0[MMX_MAX_RADIUS_FAST + 1];
    float xw2[MMX_MAX_RADIUS_FAST + 1];
};

namespace {

constexpr int kG3 = 16;       // z planes (tile rows) per group
constexpr int kNC3 = 8;       // consumer waves
constexpr int kMaxPx3 = 320;  // 5 producer waves
#ifndef ZX3_PF
#define ZX3_PF 1
#endif
constexpr int kPF3 = ZX3_PF;  // groups of z planes in flight per producer lane

__device__ __forceinline__ int reflect_once3(int i, int n)
{
    i = i < 0 ? -1 - i : i;
    return i >= n ? 2 * n - 1 - i : i;
}
__device__ __forceinline__ int reflect_clamped3(int i, int n)
{
    i = reflect_once3(i, n);
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

using rsrc3_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc3_t make_rsrc3(const void* p)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
template <typename T> struct vox3;
template <> struct vox3<uint8_t> {
    static __device__ __forceinline__ float load(rsrc3_t r, unsigned o, unsigned so = 0) { return __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b8(r, o, so, 0)); }
    static __device__ __forceinline__ float act(float raw) { return (float)__float_as_uint(raw); }
};
template <> struct vox3<uint16_t> {
    static __device__ __forceinline__ float load(rsrc3_t r, unsigned o, unsigned so = 0) {
#ifdef ZX3_NO_LOAD
        return __uint_as_float(o & 0xffffu);
#endif
        return __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b16(r, o, so, 0)); }
    static __device__ __forceinline__ float act(float raw) { return (float)__float_as_uint(raw); }
};
template <> struct vox3<float> {
    static __device__ __forceinline__ float load(rsrc3_t r, unsigned o, unsigned so = 0) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, o, so, 0)); }
    static __device__ __forceinline__ float act(float raw) { return raw; }
};

template <int R> struct xgeom3 {
    static constexpr int NS = 4 + (R + 1) / 2;                    // k-steps of 4 inputs: x_in = xo - R + 4s + kq
    static constexpr int S = R;                                   // staged column of x = 0
    static constexpr int COLS = kMaxPx3 + 4 * NS;                 // last column a consumer can read, + 1
    static constexpr int PW = ((COLS + 29) / 32) * 32 + 2;        // row pitch in floats, = 2 (mod 32)
    static constexpr int ARR = kG3 * PW;                          // one array (Gz or Gzz) of one buffer
    static constexpr int BUF = 2 * ARR;
    static constexpr size_t LDS_BYTES = (size_t)2 * BUF * sizeof(float);
};

// Producer wave: the Z pass.  TAIL = false: lane = one column x, 16 outputs per group from a register window of
// 2R + 16 inputs.  TAIL = true (block widths just past a multiple of 64): the wave takes the last <= 8 columns
// with lane = (plane mod 8, column): the same window and prefetch ring shifted along z per lane, two outputs
// per lane and group (rows p and p + 8).
template <int R, typename InT, bool TAIL>
__device__ __forceinline__ void zx3_producer(const InT* __restrict__ vol, int64_t stride_z, int stride_y, int stride_x,
                                             const mmx_block& bd, int y, int xlane, float* tile, const mmx_taps_zx3& T)
{
    using io = vox3<InT>;
    using xg = xgeom3<R>;
    constexpr int NA = 2 * R + kG3;      // register window: inputs z0-R .. z0+R+15 (+ zs)
    constexpr int PW = xg::PW;
    constexpr int S0 = TAIL ? 7 : 0;     // window index of the first output (relative to R); descriptor shift
    constexpr int NSO = TAIL ? 2 : kG3;  // outputs per lane and group
    constexpr int OST = TAIL ? 8 : 1;    // window / tile-row stride between a lane's outputs
    constexpr int I0 = TAIL ? 7 : 0;     // first window entry in use
    const int W = bd.nx, px = bd.px, nz = bd.nz;
    const int ngroups = (nz + kG3 - 1) / kG3;
    const int lane = threadIdx.x & 63;
    const int x = TAIL ? (W & ~63) + (lane & 7) : xlane;
    const int row0 = TAIL ? (lane >> 3) : 0;
    const int zs = TAIL ? row0 - 7 : 0;
    const bool lane_on = TAIL ? true : x < px;
    const bool col_real = x < W;
    const int xl = col_real ? x : W - 1;
    const InT* in = vol + bd.src_off + (int64_t)y * stride_y;
    const unsigned zstride_b = (unsigned)(stride_z * (int64_t)sizeof(InT));
    const unsigned voff0 = (unsigned)(xl * stride_x) * (unsigned)sizeof(InT);
    const unsigned voff = voff0 + (unsigned)(zs + S0) * zstride_b;
    const int qmain = xg::S + x;
    const int qleft = xg::S - 1 - x;
    const int qright = xg::S + W + (W - 1 - x);
    const bool do_left = col_real && x < R;
    const bool do_right = col_real && x >= W - R;
    float w[NA];
    float pf[kPF3][kG3];
    auto load_plane = [&](int plane, int pb) __attribute__((always_inline)) {
        if constexpr (TAIL)
            return io::load(make_rsrc3(in + (int64_t)pb * stride_z), voff0 + (unsigned)(plane - pb) * zstride_b);
        else
            return io::load(make_rsrc3(in + (int64_t)plane * stride_z), voff0);
    };
    const int pb_end = nz > 64 ? nz - 64 : 0;
    if (lane_on) {
#pragma unroll
        for (int i = I0; i < NA; ++i)
            w[i] = io::act(load_plane(reflect_clamped3(i - R + zs, nz), 0));
#pragma unroll
        for (int u = 0; u < kPF3; ++u)
#pragma unroll
            for (int j = 0; j < kG3; ++j)
                pf[u][j] = load_plane(reflect_clamped3(u * kG3 + R + kG3 + j + zs, nz), 0);
    }
#pragma unroll 1
    for (int g0 = 0; g0 <= ngroups; g0 += kPF3) {
#pragma unroll
        for (int u = 0; u < kPF3; ++u) {
            const int g = g0 + u;
            if (g > ngroups) break;
            if (g < ngroups && lane_on) {
                float* gz = tile + (g & 1) * xg::BUF + row0 * PW;
                float* gzz = gz + xg::ARR;
                const int z0 = g * kG3;
#pragma unroll
                for (int s = 0; s < NSO; ++s) {
                    const int c = R + S0 + s * OST;
                    const float cv = w[c];
                    v2f3 a = (v2f3){cv, cv} * T.zw[0];
#ifndef ZX3_SKIP_PROD
#pragma unroll
                    for (int k = 1; k <= R; ++k) {
                        const float p = w[c - k] + w[c + k];
                        a = __builtin_elementwise_fma((v2f3){p, p}, T.zw[k], a);
                    }
#endif
                    const int ro = s * OST * PW;
                    if (col_real) { gz[ro + qmain] = a.x; gzz[ro + qmain] = a.y; }
                    // reflect halos of the row: x = -1-t <- x = t ; x = W+j <- x = W-1-j
                    if (do_left) { gz[ro + qleft] = a.x; gzz[ro + qleft] = a.y; }
                    if (do_right) { gz[ro + qright] = a.x; gzz[ro + qright] = a.y; }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // shift the window by 16 planes (ascending, in place), take the planes loaded kPF3 groups ago
                // and reload their registers for group g + kPF3
#pragma unroll
                for (int i = I0; i < 2 * R; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(w[i]) : "v"(w[i + kG3]));
                const int zf = z0 + (kPF3 + 1) * kG3 + R;       // first plane of the group being prefetched
                if (zf + kG3 <= nz) {
                    const rsrc3_t rs = make_rsrc3(in + (int64_t)(zf - S0) * stride_z);
#pragma unroll
                    for (int j = 0; j < kG3; ++j) {
                        w[2 * R + j] = io::act(pf[u][j]);
                        pf[u][j] = io::load(rs, voff, (unsigned)j * zstride_b);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < kG3; ++j) {
                        w[2 * R + j] = io::act(pf[u][j]);
                        pf[u][j] = load_plane(reflect_clamped3(zf + j + zs, nz), pb_end);
                    }
                }
            }
            __syncthreads();
        }
    }
}

template <int R, typename InT>
__global__ void __launch_bounds__(kMaxPx3 + kNC3 * 64)
zx3_kernel(const InT* __restrict__ vol, int64_t stride_z, int stride_y, int stride_x,
           const mmx_block* __restrict__ blocks, int64_t slot_elems,
           float* __restrict__ gp, float* __restrict__ gq, mmx_taps_zx3 T)
{
    using xg = xgeom3<R>;
    constexpr int PW = xg::PW;
    constexpr int NS = xg::NS;
    extern __shared__ float tile3[];        // [2][2][kG3][PW]
    const mmx_block bd = blocks[blockIdx.y];
    const int y = blockIdx.x;
    if (y >= bd.ny) return;                 // whole workgroup
    const int W = bd.nx, px = bd.px, nz = bd.nz;
    const int npw = (int)(blockDim.x >> 6) - kNC3;     // producer wave slots of this launch
    const int npw_b = (px + 63) >> 6;                  // producer waves this block needs
    const int t = threadIdx.x;
    const int wv = t >> 6;
    const int ngroups = (nz + kG3 - 1) / kG3;
    // clear the tile: columns no producer writes must hold finite values (they meet zero weights)
    for (int i = t; i < 2 * xg::BUF; i += (int)blockDim.x) tile3[i] = 0.f;
    __syncthreads();
    const bool tailmode = (W & 63) != 0 && (W & 63) <= 8 && ((W + 63) >> 6) == npw_b;

    if (wv < npw) {
        if (wv >= npw_b) {                   // a narrower block in a batch sized for wider ones: only the barriers
            for (int g = 0; g <= ngroups; ++g) __syncthreads();
        } else if (tailmode && wv == npw_b - 1) {
            zx3_producer<R, InT, true>(vol, stride_z, stride_y, stride_x, bd, y, 0, tile3, T);
        } else {
            zx3_producer<R, InT, false>(vol, stride_z, stride_y, stride_x, bd, y, wv * 64 + (t & 63), tile3, T);
        }
    } else {
        // ------------------------------------------------------------ consumers: X pass on the matrix cores
        const int ord = wv - npw;
        const int lane = t & 63;
        const int li = lane & 15, kq = lane >> 4;
        const int ntiles = (W + 15) >> 4;
        // Toeplitz fragments: A[i = x_out][k = kq] of k-step s is w[|4s + kq - R - i|] (0 outside the band)
        float wa0[NS], wa2[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            int d = 4 * s + kq - R - li;
            d = d < 0 ? -d : d;
            const bool in_band = d <= R;
            d = in_band ? d : 0;
            const float a0 = T.xw0[d], a2 = T.xw2[d];
            wa0[s] = in_band ? a0 : 0.f;
            wa2[s] = in_band ? a2 : 0.f;
        }
        const rsrc3_t rp = make_rsrc3(gp + (int64_t)bd.slot * slot_elems);
        const rsrc3_t rq = make_rsrc3(gq + (int64_t)bd.slot * slot_elems);
        const unsigned row_b = (unsigned)px * 4u;
        const unsigned plane_b = (unsigned)bd.ny * row_b;
        // this lane's output: z = z0 + li, x = 16 t + 4 kq
        unsigned obase = (unsigned)li * plane_b + (unsigned)y * row_b + (unsigned)kq * 16u;
        const int lbase = li * PW + kq;     // + buffer + 16 t + 4 s
#pragma unroll 1
        for (int g = 0; g <= ngroups; ++g) {
            if (g >= 1) {
                const float* bz = tile3 + ((g - 1) & 1) * xg::BUF + lbase;
                const int z0 = (g - 1) * kG3;
                const bool zok = z0 + li < nz;
#pragma unroll 1
                for (int tt = ord; tt < ntiles; tt += kNC3) {
                    const float* pz = bz + tt * 16;
                    const float* pzz = pz + xg::ARR;
                    v4f3 accP = {0.f, 0.f, 0.f, 0.f}, accA = accP, accB = accP;
#ifdef ZX3_SKIP_CONS
                    constexpr int NSX = 1;
#else
                    constexpr int NSX = NS;
#endif
                    // all B operands of the tile first (2 * NS registers): the LDS latency is paid once per
                    // tile, not once per pair of k-steps
                    float dz[NSX], dzz[NSX];
#pragma unroll
                    for (int s = 0; s < NSX; ++s) { dz[s] = pz[4 * s]; dzz[s] = pzz[4 * s]; }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s = 0; s < NSX; ++s) {
                        accP = __builtin_amdgcn_mfma_f32_16x16x4f32(wa0[s], dz[s], accP, 0, 0, 0);
                        accA = __builtin_amdgcn_mfma_f32_16x16x4f32(wa2[s], dz[s], accA, 0, 0, 0);
                        accB = __builtin_amdgcn_mfma_f32_16x16x4f32(wa0[s], dzz[s], accB, 0, 0, 0);
                    }
#ifdef ZX3_NO_STORE
                    asm volatile("" ::"v"(accP), "v"(accA), "v"(accB));
                    if (false) {
#else
                    if (zok) {
#endif
                        const unsigned o = obase + (unsigned)tt * 64u;
                        const v4f3 q = accA + accB;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, accP), rp, o, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, q), rq, o, 0, 0);
                    }
                }
                obase += (unsigned)kG3 * plane_b;
            }
            __syncthreads();
        }
    }
}

template <int R>
int launch_zx3(const mmx_volume* vol, const mmx_block* d_blocks, int n_blocks, int max_ny, int max_px,
               int64_t slot_elems, const mmx_taps_f32& tz, const mmx_taps_f32& tx, float* d_p, float* d_q,
               hipStream_t s)
{
    if (max_px > kMaxPx3) return MMX_ERR_UNSUPPORTED;
    if (vol->stride_z * 16 * (int64_t)sizeof(double) >= (int64_t(1) << 32)) return MMX_ERR_UNSUPPORTED;   // scalar plane offsets
    mmx_taps_zx3 T;
    for (int k = 0; k <= MMX_MAX_RADIUS_FAST; ++k) {
        T.zw[k] = (v2f3){tz.w0[k], tz.w2[k]};
        T.xw0[k] = tx.w0[k];
        T.xw2[k] = tx.w2[k];
    }
    const int np = (max_px + 63) / 64 * 64;
    const int threads = np + kNC3 * 64;
    const size_t lds = xgeom3<R>::LDS_BYTES;
    dim3 grid(max_ny, n_blocks);
    const int sy = (int)vol->stride_y, sx = (int)vol->stride_x;
#define MMX_ZX3_LAUNCH(TT)                                                                                 \
    do {                                                                                                   \
        auto k = zx3_kernel<R, TT>;                                                                        \
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
            return MMX_ERR_HIP;                                                                            \
        hipLaunchKernelGGL(k, grid, dim3(threads), lds, s, (const TT*)vol->d_data, vol->stride_z, sy, sx,   \
                           d_blocks, slot_elems, d_p, d_q, T);                                             \
    } while (0)
    if (vol->dtype == MMX_U16) MMX_ZX3_LAUNCH(uint16_t);
    else if (vol->dtype == MMX_F32) MMX_ZX3_LAUNCH(float);
    else if (vol->dtype == MMX_U8) MMX_ZX3_LAUNCH(uint8_t);
    else return MMX_ERR_UNSUPPORTED;
#undef MMX_ZX3_LAUNCH
    return hipGetLastError() == hipSuccess ? MMX_OK : MMX_ERR_HIP;
}

}  // namespace

#define MMX_FOR_EACH_RADIUS(X) \
    X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) \
    X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24)

int mmx_launch_zx3(const mmx_volume* vol, const mmx_block* d_blocks, int n_blocks, int max_ny, int max_px,
                   int64_t slot_elems, const mmx_taps_f32& tz, const mmx_taps_f32& tx, int radius,
                   float* d_p, float* d_q, hipStream_t stream)
{
    switch (radius) {
#define X(R) case R: return launch_zx3<R>(vol, d_blocks, n_blocks, max_ny, max_px, slot_elems, tz, tx, d_p, d_q, stream);
        MMX_FOR_EACH_RADIUS(X)
#undef X
        default: return MMX_ERR_UNSUPPORTED;
    }
}
