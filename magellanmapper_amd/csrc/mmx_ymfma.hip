// Y pass of the tiled path on the matrix cores (16-bit tiles, gfx950): y6_kernel's results from MFMAs.
//
//   ym_kernel :  tiles of (P unorm16 | Q snorm16) dwords  ->  LoG = -s^2 ( G''(y) P + G(y) Q )  + NMS entries, sparse store
//
// Why.  y6_kernel (mmx_fused.hip) forms every output with 2R + 1 packed VALU taps per lane: at R = 16 that is 34
// instructions per wave and row, ~190 cycles -- as long as the memory system needs to deliver the row -- and the two
// overlap only partly: removing the taps takes the kernel from 2.1 to 1.5 ms per 64 blocks (R = 20: 2.5 to 1.4;
// profiles/r03_y6_experiments.txt).  A 1-D convolution along y is a product with a banded Toeplitz matrix, and the
// bytes of the 16-bit tiles are exact float16 numbers, so the taps can run on v_mfma_f32_16x16x32_f16 instead.
//
// Layout.  A workgroup owns one (column tile c, z tile U) as in y6_kernel, each wave 4 planes x 16 columns of it = 64
// columns.  The MFMA wants 8 consecutive y per lane (k) and 16 columns per instruction (n), so a lane does not keep
// "its" column: lane (kgroup = l >> 4, n = l & 15) loads 16 bytes = 4 consecutive x of (row k0 + 8 kgroup + i, plane
// n >> 2, x quad n & 3) -- per k-group 256 contiguous bytes, the wave's quarter of the tile row -- and the four dwords
// of a lane are the same (k, n) element of FOUR B operands (x mod 4 picks the MFMA).  Each dword holds P (low half)
// and Q (high half); the four bytes become four float16 pieces with ONE byte permute each: a byte in the low half of a
// float16 is the subnormal b x 2^-24, which the matrix cores take exactly (tools/exp/mfma_denorm.hip: measured); the
// high byte of Q, two's complement, is flipped to offset binary and its 128 leaves through the accumulators' start
// value.  So
//     acc = sum_k  W2[k] (256 Phi + Plo) + W0[k] (256 Qhi + Qlo),      W = weight x tile scale x 2^e  (float16 range)
// is six MFMAs per (output tile of 16 rows, k-block of 32 rows, column set): Phi x 256 wh, Plo x wh, Phi x wl
// (wl = 256 (W - wh); the Plo x (W - wh) term, <= 2^-12 of 255 / 65535 of the sum, is dropped and counted in
// mmx_tiled_q16_error_bound), the same for Q.  The A operands are Toeplitz fragments w[|k - m + delta|] for the NB
// block offsets delta = RB - 16 t, t = 0 .. NB - 1, of a k-block that starts RB = 8 (NB - 2) rows before the first tile
// it feeds: NB = 3 / 4 / 5 offsets reach every tap of radius <= 8 / 16 / 24 (the next offset at either end starts
// 8 (NB - 2) + 1 rows away).  Built once per workgroup in LDS.  SciPy's "reflect" boundary is taken by the loads (rows
// mirrored, rows beyond reach clamped: zero weights).
// Two output tiles are complete after every k-block; their accumulators (four consecutive y per lane) go through LDS
// (4 KiB per wave and tile) to come back as one row of 64 columns per step, lane = column as in y6_kernel, whose
// ballot / sparse store / entry code then runs unchanged -- except that a tile with nothing above the threshold (a
// ballot over the accumulators) skips all of it and writes its sixteen zero entries with one store.

#include <type_traits>

#include <cstdlib>

#include "mmx_common.h"

typedef _Float16 h2_y __attribute__((ext_vector_type(2)));
typedef _Float16 h8_y __attribute__((ext_vector_type(8)));
typedef float f2_y __attribute__((ext_vector_type(2)));
typedef float f4_y __attribute__((ext_vector_type(4)));
typedef unsigned u4_y __attribute__((ext_vector_type(4)));

struct ym_cfg {
    float w2[MMX_MAX_RADIUS_FAST + 1];      // G''(y) x (-norm) x (bound(P) / 65535) x 2^e: what a P count weighs
    float w0[MMX_MAX_RADIUS_FAST + 1];      // G(y) x (-norm) x (bound(Q) / 32767) x 2^e: what a Q count weighs
    float unscale;                          // 2^-e x 2^24 (the pieces are b x 2^-24)
    float start;                            // accumulators start here: minus what the 128 of Q's high byte adds
    float lo, eps;                          // nms_lo, nms_eps in accumulator units (/ unscale: a power of two, exact)
    int radius;
    int reverse;                            // 1: blockIdx.y = n - 1 takes the first block (the launch walks the batch backwards)
};

namespace {

using rsrc_y = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_y make_rsrc_y(const void* p)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ unsigned pack_h2y(float a, float b)
{
    const f2_y v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, h2_y));
}
__device__ __forceinline__ f4_y mfma_y(const u4_y& a, const u4_y& b, const f4_y& c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_y, a), __builtin_bit_cast(h8_y, b), c, 0, 0, 0);
}

// Workgroup = four waves = one tile column.  NB <= 4 (R <= 16): 164 registers and 44 / 48 KiB of LDS, three workgroups
// per CU; NB == 5 (R <= 24): 180 registers and 52 KiB, two (held to 168 registers it spills 20 and is slower: 2.10
// against 1.72 ms per 64 blocks).  Measured with six offsets (RB = 32, what R <= 24 took before the five-offset
// alignment was found: 1.99 ms) and kept as the YM6_* switches: pieces built for two column sets at a time -- 32
// registers less, the fragments read twice -- with TWELVE waves = three tile columns sharing one fragment table, i.e.
// three waves per SIMD too: 2.42 ms -- the LDS then carries as many cycles of fragment reads as the matrix pipe has
// MFMAs.
#ifndef YM5_OCC
#define YM5_OCC 2
#endif
#ifndef YM6_WAVES
#define YM6_WAVES 4
#define YM6_JH 4
#define YM6_NF 4
#endif
template <int NB, bool MASK>
__global__ void __launch_bounds__(NB <= 4 ? 256 : YM6_WAVES * 64, NB <= 4 ? 3 : (YM6_WAVES == 4 ? YM5_OCC : 1))
ym_kernel(const mmx_block* __restrict__ blocks, int64_t slot_elems, int64_t tile_stride,
          const unsigned* __restrict__ gt, float* __restrict__ out, ym_cfg cfg,
          unsigned long long* __restrict__ mask)
{
    constexpr int RB = 8 * (NB - 2);            // rows a k-block starts before the first output tile it feeds
    constexpr int NF = NB <= 4 ? 4 : YM6_NF;    // fragments per tile: [kernel: G'' (P), G (Q)][piece: wh, wl(, 256 wh)]
    __shared__ u4_y frag[NB * NF * 64];         // (NF == 4: 256 wh is made from wh -- 16 KiB instead of 24: the third workgroup)
    constexpr int WAVES = NB <= 4 ? 4 : YM6_WAVES;     // per workgroup: WAVES / 4 tile columns
    constexpr int JH = NB <= 4 ? 4 : YM6_JH;           // column sets whose pieces are held at a time
    __shared__ float tr[WAVES * 2 * 1024];      // per wave: two finished tiles of 16 rows x 64 columns

    // (reverse: the Z+X kernel has just written the batch's P / Q tiles block after block; walking the batch from its
    //  last block back starts with the tiles written last -- what the 256 MiB memory-side cache still holds)
    const mmx_block bd = blocks[cfg.reverse ? gridDim.y - 1 - blockIdx.y : blockIdx.y];
    const int ntx = (bd.nx + 15) >> 4, ntz = (bd.nz + 15) >> 4;
    if ((int)blockIdx.x * (WAVES / 4) >= ntx * ntz) return;     // (the whole workgroup)
    // ---- Toeplitz fragments: A[m][k] = w[|k - m + delta_t|], lane = (m = l & 15, k = 8 (l >> 4) + i)
    for (int e = threadIdx.x; e < NB * NF * 64; e += WAVES * 64) {
        const int ln = e & 63, f = e >> 6;
        const int piece = (f % NF) % (NF / 2), kern = (f % NF) / (NF / 2), t = f / NF;
        const int m = ln & 15, kq = ln >> 4;
        const int delta = RB - 16 * t;
        unsigned pk[4];
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            float v[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                int d = 8 * kq + i + u - m + delta;
                d = d < 0 ? -d : d;
                const float ws = d <= cfg.radius ? (kern ? cfg.w0[d] : cfg.w2[d]) : 0.f;
                const float h = (float)(_Float16)ws;
                v[u] = piece == 0 ? h : (piece == 1 ? (ws - h) * 256.f : h * 256.f);
            }
            pk[i >> 1] = pack_h2y(v[0], v[1]);
        }
        frag[e] = (u4_y){pk[0], pk[1], pk[2], pk[3]};
    }
    __syncthreads();

    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tile = (int)blockIdx.x * (WAVES / 4) + (wv >> 2);
    if (tile >= ntx * ntz) return;                              // (whole waves; no barriers below)
    const int c = tile / ntz, U = tile - c * ntz;
    const int zq = wv & 3;
    if (16 * U + 4 * zq >= bd.nz) return;                       // whole wave past the block
    const int lane = threadIdx.x & 63;
    const int n = bd.ny;
    const float nms_lo = cfg.lo, nms_eps = cfg.eps, us = cfg.unscale;
    // ---- as a loader / MFMA operand holder: k-group kq, column index n16 (plane n16 >> 2, x quad n16 & 3)
    const int kq = lane >> 4, n16 = lane & 15;
    const unsigned trow_b = (unsigned)(ntx * ntz) * 1024u;      // bytes from one y to the next
    const rsrc_y rs = make_rsrc_y(gt + (int64_t)bd.slot * tile_stride + (int64_t)tile * 256);
    const unsigned voff = (unsigned)(zq * 256 + n16 * 16);
    const unsigned voff_s = voff + (unsigned)(8 * kq) * trow_b;  // blocks wholly inside the column: rows by scalar offset
    // ---- as a row worker (y6_kernel's lane): plane lane >> 4 of the quarter, column lane & 15
    const int xi = lane & 15;
    const int z = 16 * U + 4 * zq + (lane >> 4), x = 16 * c + xi;
    const bool real = z < bd.nz && x < bd.nx;
    float* w1 = out + (int64_t)bd.slot * slot_elems;
    const rsrc_y rsw = make_rsrc_y(w1);
    const unsigned ooff = (unsigned)((real ? z : 0) * bd.ny * bd.px + (real ? x : 0)) * 4u;
    const unsigned row_b = (unsigned)bd.px * 4u;
    const int nent = ((bd.nz + 3) >> 2) * ntx;
    ulonglong2* mrow = MASK ? reinterpret_cast<ulonglong2*>(mask) + ((int64_t)bd.slot * slot_elems >> 5) +
                              (4 * U + zq) * ntx + c
                            : nullptr;
    unsigned long long ab_prev = 0;
    const bool has_l = xi > 0, has_r = xi < 15 && x + 1 < bd.nx;
    float prev1 = -INFINITY, prev2 = -INFINITY, nbx_prev = -INFINITY;
    int ydone = 0;
    float* trw = tr + wv * 2048;

    // one output row (y6_kernel's step after its taps; values and thresholds in accumulator units)
    auto row = [&](float acc, int y) __attribute__((always_inline)) {
        const unsigned long long ab = MASK ? __ballot(real & (acc > nms_lo)) : ~0ull;
#ifndef YM_NOSTORE
        if (ab && real) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc * us), rsw, ooff, (unsigned)y * row_b, 0);
#endif
        if constexpr (MASK) {
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
    };
    // the rows of one finished tile, out of LDS slot s; `any`: something of it is above the threshold
    auto rows_of = [&](int T, int s, bool any) __attribute__((always_inline)) {
        if (T < 0 || 16 * T >= n) return;
        const int cnt = n - 16 * T < 16 ? n - 16 * T : 16;
        if (MASK && !any) {
            // nothing above the threshold in 16 rows x 64 columns: a value below it neither is a candidate nor beats
            // one, so the pending row is decided as if its successor were -inf and the rest are zero entries
            if (ydone > 0) {
                unsigned long long m = 0;
                if (ab_prev) {
                    const bool cand = real & (prev1 > nms_lo) & !(fmaxf(prev2, nbx_prev) > prev1 + nms_eps);
                    m = __ballot(cand);
                }
                if (lane == 0) *mrow = make_ulonglong2(m, ab_prev);
                mrow += nent;
            }
            if (lane < cnt - 1) mrow[(int64_t)lane * nent] = make_ulonglong2(0ull, 0ull);
            mrow += (int64_t)(cnt - 1) * nent;
            prev1 = prev2 = nbx_prev = -INFINITY;
            ab_prev = 0;
            ydone += cnt;
            return;
        }
        const float* src = trw + s * 1024 + lane;
        if (!MASK || cnt < 16) {                   // (every row stored, or the column's last, partial tile: row by row)
#pragma unroll 1
            for (int r = 0; r < cnt; ++r) row(src[r * 64], 16 * T + r);
            return;
        }
        // A whole tile at once: all sixteen rows in registers, one ballot each; only rows with something above the
        // threshold cost more than that (store, x neighbours, candidate test), and the sixteen entries decided here
        // -- the pending row and rows 0 .. 14 -- leave in one store, lane i holding the entry of row 16 T - 1 + i.
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = src[r * 64];
        unsigned long long ab[16];
        unsigned long long some = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) { ab[r] = __ballot(real & (v[r] > nms_lo)); some |= ab[r]; }
        unsigned e0 = 0, e1 = 0, e2 = 0, e3 = 0;                   // entry = (cand lo, cand hi, above lo, above hi)
        auto put = [&](int ln, unsigned long long m, unsigned long long a) __attribute__((always_inline)) {
            const bool me = lane == ln;
            e0 = me ? (unsigned)m : e0;
            e1 = me ? (unsigned)(m >> 32) : e1;
            e2 = me ? (unsigned)a : e2;
            e3 = me ? (unsigned)(a >> 32) : e3;
        };
        if (ab_prev) {                                              // the pending row: its successor is v[0]
            const bool cand = real & (prev1 > nms_lo) & !(fmaxf(fmaxf(prev2, v[0]), nbx_prev) > prev1 + nms_eps);
            put(0, __ballot(cand), ab_prev);
        }
        float nbx15 = -INFINITY;
        if (some) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (ab[r]) {
#ifndef YM_NOSTORE
                    if (real) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[r] * us), rsw, ooff, (unsigned)(16 * T + r) * row_b, 0);
#endif
                    const float l = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(
                        (int)__float_as_uint(v[r]), (int)__float_as_uint(v[r]), 0x111 /* row_shr:1 */, 0xf, 0xf, false));
                    const float rr = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(
                        (int)__float_as_uint(v[r]), (int)__float_as_uint(v[r]), 0x101 /* row_shl:1 */, 0xf, 0xf, false));
                    const float nbx = fmaxf(has_l ? l : -INFINITY, has_r ? rr : -INFINITY);
                    if (r < 15) {
                        const float below = r > 0 ? v[r > 0 ? r - 1 : 0] : prev1;
                        const bool cand = real & (v[r] > nms_lo) & !(fmaxf(fmaxf(below, v[r < 15 ? r + 1 : 15]), nbx) > v[r] + nms_eps);
                        put(r + 1, __ballot(cand), ab[r]);
                    } else {
                        nbx15 = nbx;
                    }
                }
            }
        }
        if (lane < 16 && (lane > 0 || ydone > 0)) {
            ulonglong2* dst = mrow + (int64_t)(ydone > 0 ? lane : lane - 1) * nent;
            *dst = make_ulonglong2((unsigned long long)e0 | ((unsigned long long)e1 << 32),
                                   (unsigned long long)e2 | ((unsigned long long)e3 << 32));
        }
        mrow += (int64_t)(ydone > 0 ? 16 : 15) * nent;
        prev2 = v[14]; prev1 = v[15]; nbx_prev = nbx15;
        ab_prev = ab[15];
        ydone += 16;
    };

    // ---- k-blocks of 32 input rows: block b holds rows -RB + 32 b + [0, 32), lane row 8 kq + i
#ifndef YM_PFD
#define YM_PFD 1
#endif
    constexpr int PFD = YM_PFD;                     // k-blocks of loads in flight per wave
    u4_y rawA[8], rawB[PFD == 2 ? 8 : 1];
    auto load_block = [&](int b, u4_y (&raw)[8]) __attribute__((always_inline)) {
        const int r0 = -RB + 32 * b;
        if (r0 >= 0 && r0 + 32 <= n) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                raw[i] = __builtin_bit_cast(u4_y, __builtin_amdgcn_raw_buffer_load_b128(rs, voff_s, (unsigned)(r0 + i) * trow_b, 0));
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int v = r0 + 8 * kq + i;
                v = v < 0 ? -1 - v : v;
                v = v >= n ? 2 * n - 1 - v : v;
                v = v < 0 ? 0 : (v > n - 1 ? n - 1 : v);       // beyond the taps' reach: any row, zero weights
                raw[i] = __builtin_bit_cast(u4_y, __builtin_amdgcn_raw_buffer_load_b128(rs, voff + (unsigned)v * trow_b, 0, 0));
            }
        }
    };
    const int nT = (n + 15) >> 4;
    const int nKB = (nT + NB - 3) / 2 + 1;          // last block b with 2 b - NB + 2 <= nT - 1
    const f4_y start = {cfg.start, cfg.start, cfg.start, cfg.start};
    f4_y acc[NB][4];                                // output tiles t = 0 .. NB - 1 of the current block (the last two start in it)
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = start;
    bool any0 = false, any1 = false;                // of the two tiles waiting in LDS

    const _Float16 k256 = (_Float16)256.f;
    // one k-block: rows of the tiles the previous block finished, then per group of JH column sets: this block's pieces,
    // (last group: the next block's loads,) the MFMAs, the hand-off of the two tiles that are complete
    auto iter = [&](int b, u4_y (&raw)[8]) __attribute__((always_inline)) {
        // -- rows first (before this block's operands take their registers; their stores are older than the loads
        //    issued below, so waiting for those does not wait for these)
#ifndef YM_ROWS_LATE
#define YM_ROWS_LATE 1
#endif
        if (!YM_ROWS_LATE || b == nKB) {
#pragma unroll 1
            for (int h = 0; h < 2; ++h) rows_of(2 * (b - 1) - NB + 2 + h, h, h ? any1 : any0);
        }
        if (b == nKB) return false;
        if constexpr (JH < 4) { any0 = !MASK; any1 = !MASK; }
#pragma unroll
        for (int j0 = 0; j0 < 4; j0 += JH) {
            __builtin_amdgcn_sched_barrier(0);
            // -- the four float16 pieces of every dword, one byte permute each: [j][Phi, Plo, Qhi, Qlo], k pairs (2p, 2p + 1)
            u4_y pc[JH][4];
#pragma unroll
            for (int j = 0; j < JH; ++j)
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const unsigned d0 = raw[2 * p][j0 + j], d1 = raw[2 * p + 1][j0 + j];
                    // v_perm_b32 {S0 = d1, S1 = d0}: selector 0-3 = bytes of d0, 4-7 = bytes of d1, 0x0c = 0x00
                    pc[j][0][p] = __builtin_amdgcn_perm(d1, d0, 0x0c050c01u);
                    pc[j][1][p] = __builtin_amdgcn_perm(d1, d0, 0x0c040c00u);
                    pc[j][2][p] = __builtin_amdgcn_perm(d1, d0, 0x0c070c03u) ^ 0x00800080u;
                    pc[j][3][p] = __builtin_amdgcn_perm(d1, d0, 0x0c060c02u);
                }
            __builtin_amdgcn_sched_barrier(0);
            if (j0 + JH == 4 && b + PFD < nKB) load_block(b + PFD, raw);
            __builtin_amdgcn_sched_barrier(0);
            if (YM_ROWS_LATE && JH == 4) {
#pragma unroll 1
                for (int h = 0; h < 2; ++h) rows_of(2 * (b - 1) - NB + 2 + h, h, h ? any1 : any0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // -- output tile T = 2 b - NB + 2 + t.  Tiles t = 0, 1 get their last block here and leave for LDS at once
            //    (their registers are free for the two tiles that start with this block: t = NB - 2, NB - 1)
            auto tile_mfmas = [&](int t, bool fresh) __attribute__((always_inline)) {
                f4_y* a = &acc[t][j0];
                const u4_y* fr = frag + (t * NF) * 64 + lane;
                const u4_y ah = fr[0], al = fr[64], bh = fr[(NF / 2) * 64], bl = fr[(NF / 2 + 1) * 64];
                // (x 256: an exponent shift, exact)
                const u4_y a256 = NF == 6 ? fr[128] : __builtin_bit_cast(u4_y, __builtin_bit_cast(h8_y, ah) * k256);
                const u4_y b256 = NF == 6 ? fr[320] : __builtin_bit_cast(u4_y, __builtin_bit_cast(h8_y, bh) * k256);
#pragma unroll
                for (int j = 0; j < JH; ++j) a[j] = mfma_y(a256, pc[j][0], fresh ? start : a[j]);
#pragma unroll
                for (int j = 0; j < JH; ++j) a[j] = mfma_y(ah, pc[j][1], a[j]);
#pragma unroll
                for (int j = 0; j < JH; ++j) a[j] = mfma_y(al, pc[j][0], a[j]);
#pragma unroll
                for (int j = 0; j < JH; ++j) a[j] = mfma_y(b256, pc[j][2], a[j]);
#pragma unroll
                for (int j = 0; j < JH; ++j) a[j] = mfma_y(bh, pc[j][3], a[j]);
#pragma unroll
                for (int j = 0; j < JH; ++j) a[j] = mfma_y(bl, pc[j][2], a[j]);
            };
#pragma unroll
            for (int t = 0; t < 2; ++t) tile_mfmas(t, t >= NB - 2);      // (NB == 3: tile 1 starts and ends in this block)
            // (accumulator register r of lane (g, n16) is row 4 g + r, column 4 n16 + j of the wave's 64)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const f4_y* a = &acc[t][j0];
                float mx = -INFINITY;
#pragma unroll
                for (int j = 0; j < JH; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, a[j][r]);
                const bool any = MASK ? __ballot(mx > nms_lo) != 0ull : true;
                if constexpr (JH == 4) {
                    if (t == 0) any0 = any; else any1 = any;
                    if (any) {
                        f4_y* dst = reinterpret_cast<f4_y*>(trw + t * 1024) + (4 * kq) * 16 + n16;
#pragma unroll
                        for (int r = 0; r < 4; ++r) dst[r * 16] = (f4_y){a[0][r], a[1][r], a[2][r], a[3][r]};
                    }
                } else {        // (half a tile: the other half may be what is above the threshold)
                    if (t == 0) any0 |= any; else any1 |= any;
                    f2_y* dst = reinterpret_cast<f2_y*>(trw + t * 1024) + (4 * kq) * 32 + 2 * n16 + (j0 >> 1);
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[r * 32] = (f2_y){a[0][r], a[1][r]};
                }
            }
            __builtin_amdgcn_sched_barrier(0);     // (the finished tiles' registers are free from here on)
#pragma unroll
            for (int t = 2; t < NB; ++t) {
                tile_mfmas(t, t >= NB - 2);
                if constexpr (NB > 4) __builtin_amdgcn_sched_barrier(0);       // one tile's fragments at a time
            }
        }
        // the window moves on by two tiles (a rotation by renaming needs the loop unrolled NB / 2 times with the row code in
        // every copy -- 70 KiB of instructions --, or a switch, after which the compiler keeps every set alive: 251 registers)
#pragma unroll
        for (int t = 0; t + 2 < NB; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[t][j] = acc[t + 2][j];
        __builtin_amdgcn_wave_barrier();
        return true;
    };
    load_block(0, rawA);
    if constexpr (PFD == 2) {
        if (nKB > 1) load_block(1, reinterpret_cast<u4_y (&)[8]>(rawB));
#pragma unroll 1
        for (int b = 0;; b += 2) {
            if (!iter(b, rawA)) break;
            if (!iter(b + 1, reinterpret_cast<u4_y (&)[8]>(rawB))) break;
        }
    } else {
#pragma unroll 1
        for (int b = 0; iter(b, rawA); ++b) {}
    }
    if constexpr (MASK) {     // the last row has no successor
        unsigned long long m = 0;
        if (ab_prev) {
            const bool cand = real & (prev1 > nms_lo) & !(fmaxf(prev2, nbx_prev) > prev1 + nms_eps);
            m = __ballot(cand);
        }
        if (lane == 0) *mrow = make_ulonglong2(m, ab_prev);
    }
}

template <int NB>
int launch_ym(const mmx_block* d_blocks, int n_blocks, const mmx_zx6_plan& plan, int64_t slot_elems,
              const ym_cfg& cfg, const float* d_p, float* d_log, unsigned long long* d_mask, hipStream_t s)
{
    constexpr int TPW = NB <= 4 ? 1 : YM6_WAVES / 4;        // tile columns per workgroup
    dim3 grid((plan.max_tiles + TPW - 1) / TPW, n_blocks);
    if (d_mask)
        hipLaunchKernelGGL((ym_kernel<NB, true>), grid, dim3(256 * TPW), 0, s, d_blocks, slot_elems, plan.tile_stride,
                           reinterpret_cast<const unsigned*>(d_p), d_log, cfg, d_mask);
    else
        hipLaunchKernelGGL((ym_kernel<NB, false>), grid, dim3(256 * TPW), 0, s, d_blocks, slot_elems, plan.tile_stride,
                           reinterpret_cast<const unsigned*>(d_p), d_log, cfg, d_mask);
    return hipGetLastError() == hipSuccess ? MMX_OK : MMX_ERR_HIP;
}

}  // namespace

// cp, cq: what one count of P (unorm16) and of Q (snorm16) is worth (mmx_launch_y6's)
int mmx_launch_ym(const mmx_block* d_blocks, int n_blocks, const mmx_zx6_plan& plan, int64_t slot_elems,
                  const mmx_taps_f32& taps, int radius, const float* d_p, float cp, float cq,
                  float* d_log, unsigned long long* d_mask, float nms_lo, float nms_eps, hipStream_t stream)
{
    if (radius < 1 || radius > 24 || radius > MMX_MAX_RADIUS_FAST || !(cp > 0.f) || !(cq > 0.f)) return MMX_ERR_UNSUPPORTED;
    ym_cfg cfg;
    float mx = 0.f;
    for (int k = 0; k <= radius; ++k) {
        cfg.w2[k] = taps.w2[k] * cp;
        cfg.w0[k] = taps.w0[k] * cq;
        mx = fmaxf(mx, fmaxf(fabsf(cfg.w2[k]), fabsf(cfg.w0[k])));
    }
    if (!(mx > 1e-30f) || !(mx < 1e30f)) return MMX_ERR_UNSUPPORTED;
    int e;
    frexpf(mx, &e);                             // mx = f x 2^e, f in [0.5, 1)
    const float up = ldexpf(1.f, -e - 1);       // largest weight into [0.25, 0.5)
    double q128 = 0.0;                          // sum over all taps of the two pieces Q's high byte meets
    for (int k = 0; k <= MMX_MAX_RADIUS_FAST; ++k) {
        cfg.w2[k] = k <= radius ? cfg.w2[k] * up : 0.f;
        cfg.w0[k] = k <= radius ? cfg.w0[k] * up : 0.f;
        const float h = (float)(_Float16)cfg.w0[k];
        const float l = (float)(_Float16)((cfg.w0[k] - h) * 256.f);
        q128 += (k ? 2.0 : 1.0) * ((double)h * 256.0 + (double)l);
    }
    // the pieces are bytes x 2^-24 (float16 subnormals); Q's high byte arrives as offset binary, 128 too large
    cfg.start = (float)(-128.0 * 0x1p-24 * q128);
    cfg.unscale = ldexpf(1.f, e + 1 + 24);
    cfg.lo = nms_lo / cfg.unscale;
    cfg.eps = nms_eps / cfg.unscale;
    cfg.radius = radius;
    cfg.reverse = 1;         // (a batch is walked from its last block backwards: the tiles Z+X wrote last)
    // NB block offsets cover every tap when the first offset beyond either end, RB + 16 - 15 and 16 NB - RB - 31 rows away,
    // is out of reach: R <= 8 (NB - 2) with RB = 8 (NB - 2)
    if (radius <= 8) return launch_ym<3>(d_blocks, n_blocks, plan, slot_elems, cfg, d_p, d_log, d_mask, stream);
    if (radius <= 16) return launch_ym<4>(d_blocks, n_blocks, plan, slot_elems, cfg, d_p, d_log, d_mask, stream);
    return launch_ym<5>(d_blocks, n_blocks, plan, slot_elems, cfg, d_p, d_log, d_mask, stream);
}
