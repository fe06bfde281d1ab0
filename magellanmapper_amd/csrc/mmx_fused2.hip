// Fused Z+X pass, second design: wave-specialised and packed-math (gfx950).
//
//   zx2_kernel :  I (u8/u16/f32, read once)  ->  P = G(x) G(z) I
//                                                Q = G''(x) G(z) I + G(x) G''(z) I
// feeding the same y2_kernel as mmx_fused.hip (LoG = -s^2 (G''(y) P + G(y) Q)); 26 algorithmic HBM
// bytes per voxel and sigma against 42 for the three separate passes.
//
// Why a second design.  Instruction counts say the separate passes are HBM-bound with the scalar-f32
// VALU already 60-100 % busy, so a fused kernel is VALU-bound unless it issues packed math, and the
// first fused kernel (one set of waves alternating between the z march and the x pass) was held at
// 17-40 % of the vector rate by its registers: z window + x window in every wave.  Here
//   * producer waves (lane = x of one block row y) march along z with the register window of
//     mmx_fused.hip and write (Gz, Gzz) PAIRS into an LDS tile of 8 rows (8 consecutive z);
//     one v_pk_fma_f32 per tap makes both from the shared pair sum;
//   * consumer waves take (row, 4-voxel chunk) items of the tile: the window holds (Gz, Gzz) pairs,
//     so per tap one v_pk_add_f32 forms both pair sums, one v_pk_fma_f32 accumulates G(x) Gz and
//     G(x) Gzz and one v_fma_f32 adds the G''(x) Gz term: 3 instructions where the scalar kernel
//     needs 5;
//   * the tile is double buffered: producers fill group g while consumers drain group g - 1, one
//     workgroup barrier per group.  Each wave kind only holds its own window, so the kernel fits
//     128 VGPRs with 14 waves per workgroup.
// v_pk_fma_f32 measured at 134 TFLOP/s against 74 for v_fma_f32 on this part (tools/exp/pkrate.hip).

#include <type_traits>

#include "mmx_common.h"

typedef float v2f __attribute__((ext_vector_type(2)));

struct mmx_taps_zx2 {
    v2f zw[MMX_MAX_RADIUS_FAST + 1];    // (w0z[k], w2z[k])
    v2f xw0[MMX_MAX_RADIUS_FAST + 1];   // (w0x[k], w0x[k])
    float xw2[MMX_MAX_RADIUS_FAST + 1];
    float _pad;
};

namespace {

constexpr int kG = 8;        // z steps (rows) per tile
constexpr int kT = 4;        // X outputs per consumer item
constexpr int kCons = 576;   // consumer lanes (9 waves): one round of (row, chunk) items for px = 288
constexpr int kMaxPx = 320;  // 5 producer waves
#ifndef ZX2_PF
#define ZX2_PF 2
#endif
constexpr int kPF = ZX2_PF;  // groups of z planes in flight per producer lane

__device__ __forceinline__ int reflect_once(int i, int n)
{
    i = i < 0 ? -1 - i : i;
    return i >= n ? 2 * n - 1 - i : i;
}
__device__ __forceinline__ int reflect_clamped(int i, int n)
{
    i = reflect_once(i, n);
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}
// LDS row layout in (Gz, Gzz) pairs: two pad pairs after every four, so that the consumers' 16-byte reads
// at a 4-pair lane stride become a 48-byte stride = 8 lanes on 32 distinct banks (conflict free)
__device__ __forceinline__ int pad2(int i) { return i + 2 * (i >> 2); }

using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
template <typename T> struct vox;
template <> struct vox<uint8_t> {
    static __device__ __forceinline__ float load(rsrc_t r, unsigned o, unsigned so = 0) { return __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b8(r, o, so, 0)); }
    static __device__ __forceinline__ float act(float raw) { return (float)__float_as_uint(raw); }
};
template <> struct vox<uint16_t> {
    static __device__ __forceinline__ float load(rsrc_t r, unsigned o, unsigned so = 0) { return __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b16(r, o, so, 0)); }
    static __device__ __forceinline__ float act(float raw) { return (float)__float_as_uint(raw); }
};
template <> struct vox<float> {
    static __device__ __forceinline__ float load(rsrc_t r, unsigned o, unsigned so = 0) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, o, so, 0)); }
    static __device__ __forceinline__ float act(float raw) { return raw; }
};

template <int R> struct xgeom2 {
    static constexpr int LEAD = R & 1;                  // odd radius: window starts one pair early (16-B reads)
    static constexpr int S = (R + LEAD + 7) & ~7;       // staged position of x = 0
    static constexpr int WIN = kT + 2 * R + 2 * LEAD;   // pairs read per item (even)
    static constexpr int SPAN = S + kMaxPx + R + LEAD;
    static constexpr int PITCH = ((SPAN + 2 * (SPAN >> 2)) + 3) & ~1;   // compile-time row pitch (pairs)
};

// Producer wave.  TAIL = false: lane = one column x, marching along z (8 outputs per group).
// TAIL = true (block widths just past a multiple of 64, e.g. 261 = 256 + the 5 overlap columns): the
// wave takes the last <= 8 columns with lane = (plane of the group, column): the same register window and
// prefetch ring, shifted along z by (plane - 7) per lane, and ONE output per lane and group (window index 7)
// -- about a quarter of a marching wave's instructions instead of a whole wave for 5 useful lanes.
template <int R, typename InT, bool TAIL>
__device__ __forceinline__ void zx2_producer(const InT* __restrict__ vol, int64_t stride_z, int stride_y, int stride_x,
                                             const mmx_block& bd, int y, int xlane, v2f* tile, const mmx_taps_zx2& T,
                                             float* __restrict__ gp, int64_t sbase)
{
    using io = vox<InT>;
    using xg = xgeom2<R>;
    constexpr int NA = 2 * R + kG;       // register window: inputs z0-R .. z0+R+G-1 (+ zs)
    constexpr int PW = xg::PITCH;
    constexpr int S0 = TAIL ? kG - 1 : 0;            // first output index of the window this wave computes
    constexpr int NS = TAIL ? 1 : kG;                // outputs per lane and group
    constexpr int I0 = TAIL ? kG - 1 : 0;            // first window entry in use
    const int W = bd.nx, px = bd.px, nz = bd.nz;
    const int ngroups = (nz + kG - 1) / kG;
    const int lane = threadIdx.x & 63;
    const int x = TAIL ? (W & ~63) + (lane & 7) : xlane;             // column of this lane
    const int row0 = TAIL ? (lane >> 3) : 0;                         // tile row of output S0
    const int zs = TAIL ? row0 - (kG - 1) : 0;                       // z shift of this lane's window
    const bool lane_on = TAIL ? true : x < px;
    const bool col_real = x < W;
    const int xl = col_real ? x : W - 1;    // pitch lanes re-read the last column
    const InT* in = vol + bd.src_off + (int64_t)y * stride_y;
    const unsigned zstride_b = (unsigned)(stride_z * (int64_t)sizeof(InT));    // < 4 GiB / 8 (checked by the launcher)
    // interior prefetches address plane (zf + zs + j) as descriptor(zf - S0) + voff + j * zstride_b
    const unsigned voff0 = (unsigned)(xl * stride_x) * (unsigned)sizeof(InT);
    const unsigned voff = voff0 + (unsigned)(zs + S0) * zstride_b;
    const int qmain = pad2(col_real ? xg::S + x : xg::S + R + x);     // tile position of this lane's column
    const int qleft = pad2(xg::S - 1 - x);
    const int qright = pad2(xg::S + W + (W - 1 - x));
    // active window: inputs z0-R .. z0+R+G-1 of the current group; pf[u]: the 8 planes that enter the
    // window after group g (g % kPF == u), loaded kPF groups ahead.  With one workgroup per CU nothing
    // else hides HBM latency, and a load must never be moved while in flight -- hence a ring of
    // register groups with static indices (the group loop is unrolled by kPF) instead of a longer
    // shifted tail.
    float w[NA];
    float pf[kPF][kG];
    // a load of one (possibly reflected) plane.  Marching waves: the plane is wave-uniform and goes into the
    // descriptor.  Tail wave: it differs per lane, and a per-lane descriptor would cost a waterfall loop per
    // load -- so the descriptor sits at a uniform plane `pb` at most 64 planes below and the rest is offset.
    auto load_plane = [&](int plane, int pb) __attribute__((always_inline)) {
        if constexpr (TAIL)
            return io::load(make_rsrc(in + (int64_t)pb * stride_z), voff0 + (unsigned)(plane - pb) * zstride_b);
        else
            return io::load(make_rsrc(in + (int64_t)plane * stride_z), voff0);
    };
    const int pb_end = nz > 64 ? nz - 64 : 0;       // base plane for the loads near the far face
    if (lane_on) {
#pragma unroll
        for (int i = I0; i < NA; ++i)
            w[i] = io::act(load_plane(reflect_clamped(i - R + zs, nz), 0));
#pragma unroll
        for (int u = 0; u < kPF; ++u)
#pragma unroll
            for (int j = 0; j < kG; ++j)
                pf[u][j] = load_plane(reflect_clamped(u * kG + R + kG + j + zs, nz), 0);
    }
#pragma unroll 1
    for (int g0 = 0; g0 <= ngroups; g0 += kPF) {
#pragma unroll
        for (int u = 0; u < kPF; ++u) {
            const int g = g0 + u;
            if (g > ngroups) break;
            if (g < ngroups && lane_on) {
                v2f* rows = tile + (g & 1) * (kG * PW) + row0 * PW;
                const int z0 = g * kG;
                v2f av[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const float c = w[R + S0 + s];
                    v2f a = (v2f){c, c} * T.zw[0];
#pragma unroll
                    for (int k = 1; k <= R; ++k) {
                        const float p = w[R + S0 + s - k] + w[R + S0 + s + k];
                        a = __builtin_elementwise_fma((v2f){p, p}, T.zw[k], a);
                    }
                    if (!TAIL || col_real) rows[s * PW + qmain] = a;   // (pitch lanes write past the right halo: no branch)
                    av[s] = a;
                    __builtin_amdgcn_sched_barrier(0);
                }
                // reflect halos of the rows, once per group: x = -1-t <- x = t ; x = W+j <- x = W-1-j
                if (x < R) {
#pragma unroll
                    for (int s = 0; s < NS; ++s) rows[s * PW + qleft] = av[s];
                }
                if (x >= W - R && col_real) {
#pragma unroll
                    for (int s = 0; s < NS; ++s) rows[s * PW + qright] = av[s];
                }
                // shift the active window by 8 (ascending, in place), take the planes loaded kPF groups
                // ago and reload their registers for group g + kPF
#pragma unroll
                for (int i = I0; i < 2 * R; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(w[i]) : "v"(w[i + kG]));
                const int zf = z0 + (kPF + 1) * kG + R;      // first plane of the group being prefetched
                if (zf + kG <= nz) {
                    // interior: one descriptor per group, the plane inside the group is a scalar offset
                    const rsrc_t rs = make_rsrc(in + (int64_t)(zf - S0) * stride_z);
#pragma unroll
                    for (int j = 0; j < kG; ++j) {
                        w[2 * R + j] = io::act(pf[u][j]);
                        pf[u][j] = io::load(rs, voff, (unsigned)j * zstride_b);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < kG; ++j) {
                        w[2 * R + j] = io::act(pf[u][j]);
                        pf[u][j] = load_plane(reflect_clamped(zf + j + zs, nz), pb_end);
                    }
                }
            }
            __syncthreads();
        }
    }
}

// Wave roles.  A workgroup is one block row: ceil(px / 64) producer waves and 9 consumer waves.  The
// hardware deals the waves of a workgroup to the 4 SIMDs of the CU round-robin (wave i -> SIMD i % 4) and
// the kernel is bound by VALU issue on the most loaded SIMD, so in the common geometry (14 waves, tail
// producer: 4 marching P at ~420 instructions per group, 1 tail T at ~100, 9 consumers C at ~300) the roles
// are dealt so that the SIMDs get  P T C C | C C C C | P P C | P C C  (1120 / 1200 / 1140 / 1020)
// instead of  P T C C | P C C C | P C C | P C C  (1120 / 1320 / 1020 / 1020).
constexpr unsigned long long kRole14 = 0xAAA8908ull;            // 2 bits per wave: 0 = P, 1 = T, 2 = C
constexpr unsigned long long kOrd14 = 0x87654323102100ull;     // 4 bits per wave: ordinal within the role

template <int R, typename InT>
__global__ void __launch_bounds__(kMaxPx + kCons)
zx2_kernel(const InT* __restrict__ vol, int64_t stride_z, int stride_y, int stride_x,
           const mmx_block* __restrict__ blocks, int64_t slot_elems,
           float* __restrict__ gp, float* __restrict__ gq, mmx_taps_zx2 T)
{
    using xg = xgeom2<R>;
    constexpr int PW = xg::PITCH;
    extern __shared__ v2f tile[];        // [2][kG][PW]
    const mmx_block bd = blocks[blockIdx.y];
    const int y = blockIdx.x;
    if (y >= bd.ny) return;              // whole workgroup
    const int W = bd.nx, px = bd.px, nz = bd.nz;
    const int npw = (px + 63) >> 6;      // producer waves
    const int t = threadIdx.x;
    const int wv = t >> 6;
    const int nwaves = (int)blockDim.x >> 6;
    const int ngroups = (nz + kG - 1) / kG;
    const int64_t sbase = (int64_t)bd.slot * slot_elems + (int64_t)y * px;
    const int64_t plane = (int64_t)bd.ny * px;
    // tail producer: the last producer wave would hold <= 8 real columns
    const bool tailmode = (W & 63) != 0 && (W & 63) <= 8 && ((W + 63) >> 6) == npw;
    int role, ord;
    if (tailmode && npw == 5 && nwaves == 14) {
        role = (int)((kRole14 >> (2 * wv)) & 3);
        ord = (int)((kOrd14 >> (4 * wv)) & 15);
    } else {
        role = wv < npw ? (tailmode && wv == npw - 1 ? 1 : 0) : 2;
        ord = wv < npw ? wv : wv - npw;
    }

    if (role == 0) {
        zx2_producer<R, InT, false>(vol, stride_z, stride_y, stride_x, bd, y, ord * 64 + (t & 63), tile, T, gp, sbase);
    } else if (role == 1) {
        zx2_producer<R, InT, true>(vol, stride_z, stride_y, stride_x, bd, y, 0, tile, T, gp, sbase);
    } else {
        // ------------------------------------------------------------------ consumers: x pass on the tile
        const int nct = (nwaves - npw) * 64;
        const int CH = px / kT;              // chunks per row
        const int nitems = kG * CH;
        const float inv_ch = 1.0f / (float)CH;
        const int lane = t & 63;
        // this lane's item of the first round is the same in every group: its row, tile offset and output
        // offset are computed once (one round per wave is the common case: 576 items, 9 waves) -- the 64-bit
        // address products are quarter-rate instructions
        const int item0 = ord * 64 + lane;
        const int r0 = (int)(((float)item0 + 0.5f) * inv_ch);        // item < 2^12: exact
        const int c0 = item0 - r0 * CH;
        const int lds0 = r0 * PW + pad2(xg::S - R - xg::LEAD + c0 * kT);
        int64_t o0 = sbase + (int64_t)r0 * plane + c0 * kT;           // + z0 * plane, advanced per group
        constexpr int kB0 = (xg::S - R - xg::LEAD) & 3;              // chunks start at multiples of 4 columns:
#pragma unroll 1                                                     // pad2(base + i) - pad2(base) is a constant
        for (int g = 0; g <= ngroups; ++g) {
            if (g >= 1) {
                const v2f* rows = tile + ((g - 1) & 1) * (kG * PW);
                const int z0 = (g - 1) * kG;
                for (int first = ord * 64; first < nitems; first += nct) {     // one round per wave for px = 288
                    const int item = first + lane;
                    int r = r0, lds_at = lds0;
                    int64_t o = o0;
                    if (first != ord * 64) {                    // further rounds (rows wider than 288 floats)
                        r = (int)(((float)item + 0.5f) * inv_ch);
                        const int c = item - r * CH;
                        lds_at = r * PW + pad2(xg::S - R - xg::LEAD + c * kT);
                        o = sbase + (int64_t)(z0 + r) * plane + c * kT;
                    }
                    if (item < nitems && z0 + r < nz) {       // (no `continue`: all lanes meet again at the loop top)
                    v2f win[xg::WIN];
                    const v2f* pr = rows + lds_at;
#pragma unroll
                    for (int i = 0; i < xg::WIN; i += 2) {
                        const float4 v = *reinterpret_cast<const float4*>(pr + (pad2(kB0 + i) - pad2(kB0)));
                        win[i] = (v2f){v.x, v.y};
                        win[i + 1] = (v2f){v.z, v.w};
                    }
                    float P[kT], Q[kT];
#pragma unroll
                    for (int oo = 0; oo < kT; ++oo) {
                        const v2f cc = win[xg::LEAD + oo + R];
                        v2f ps = cc * T.xw0[0];                 // (G(x) Gz, G(x) Gzz)
                        float q = cc.x * T.xw2[0];              // G''(x) Gz
#pragma unroll
                        for (int k = 1; k <= R; ++k) {
                            const v2f sm = win[xg::LEAD + oo + R - k] + win[xg::LEAD + oo + R + k];
                            ps = __builtin_elementwise_fma(sm, T.xw0[k], ps);
                            q = fmaf(sm.x, T.xw2[k], q);
                        }
                        P[oo] = ps.x;
                        Q[oo] = ps.y + q;
                    }
                    *reinterpret_cast<float4*>(gp + o) = make_float4(P[0], P[1], P[2], P[3]);
                    *reinterpret_cast<float4*>(gq + o) = make_float4(Q[0], Q[1], Q[2], Q[3]);
                    }
                }
                o0 += (int64_t)kG * plane;
            }
            __syncthreads();
        }
    }
}

template <int R>
int launch_zx2(const mmx_volume* vol, const mmx_block* d_blocks, int n_blocks, int max_ny, int max_px,
               int64_t slot_elems, const mmx_taps_f32& tz, const mmx_taps_f32& tx, float* d_p, float* d_q,
               hipStream_t s)
{
    if (max_px > kMaxPx) return MMX_ERR_UNSUPPORTED;
    if (vol->stride_z * 8 * (int64_t)sizeof(double) >= (int64_t(1) << 32)) return MMX_ERR_UNSUPPORTED;   // scalar plane offsets
    mmx_taps_zx2 T;
    for (int k = 0; k <= MMX_MAX_RADIUS_FAST; ++k) {
        T.zw[k] = (v2f){tz.w0[k], tz.w2[k]};
        T.xw0[k] = (v2f){tx.w0[k], tx.w0[k]};
        T.xw2[k] = tx.w2[k];
    }
    T._pad = 0.f;
    const int np = (max_px + 63) / 64 * 64;
    const int threads = np + kCons;
    const size_t lds = (size_t)2 * kG * xgeom2<R>::PITCH * sizeof(v2f);
    dim3 grid(max_ny, n_blocks);
    const int sy = (int)vol->stride_y, sx = (int)vol->stride_x;
#define MMX_ZX2_LAUNCH(TT)                                                                                 \
    do {                                                                                                   \
        auto k = zx2_kernel<R, TT>;                                                                        \
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
            return MMX_ERR_HIP;                                                                            \
        hipLaunchKernelGGL(k, grid, dim3(threads), lds, s, (const TT*)vol->d_data, vol->stride_z, sy, sx,   \
                           d_blocks, slot_elems, d_p, d_q, T);                                             \
    } while (0)
    if (vol->dtype == MMX_U16) MMX_ZX2_LAUNCH(uint16_t);
    else if (vol->dtype == MMX_F32) MMX_ZX2_LAUNCH(float);
    else if (vol->dtype == MMX_U8) MMX_ZX2_LAUNCH(uint8_t);
    else return MMX_ERR_UNSUPPORTED;
#undef MMX_ZX2_LAUNCH
    return hipGetLastError() == hipSuccess ? MMX_OK : MMX_ERR_HIP;
}

}  // namespace

#define MMX_FOR_EACH_RADIUS(X) \
    X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) \
    X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24)

int mmx_launch_zx2(const mmx_volume* vol, const mmx_block* d_blocks, int n_blocks, int max_ny, int max_px,
                   int64_t slot_elems, const mmx_taps_f32& tz, const mmx_taps_f32& tx, int radius,
                   float* d_p, float* d_q, hipStream_t stream)
{
    switch (radius) {
#define X(R) case R: return launch_zx2<R>(vol, d_blocks, n_blocks, max_ny, max_px, slot_elems, tz, tx, d_p, d_q, stream);
        MMX_FOR_EACH_RADIUS(X)
#undef X
        default: return MMX_ERR_UNSUPPORTED;
    }
}
