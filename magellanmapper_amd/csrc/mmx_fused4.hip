// Fused X+Z pass on the matrix cores with split-float16 operands (gfx950), the default Z+X path for integer
// voxels.
//
//   zx4_kernel :  I (u8 / u16, read once, x contiguous)  ->  P = G(z) G(x) I
//                                                            Q = G(z) G''(x) I + G''(z) G(x) I
// in two forms that share the arithmetic below:
//   zx_mode 6                   voxels from the operand-ordered copy zx6_pack_kernel makes once per batch, P / Q out
//                               as 16 x 16 tiles of float32 -- every access one contiguous KiB -- feeding y6_kernel;
//   zx_mode 7  (Q16)            the same with the tiles as one dword per voxel (P unorm16, Q snorm16 of value /
//                               bound): the default, whenever the stated rounding bound is inside MMX_LOG_ABS_TOL and
//                               the caller's NMS band covers it.
// (The row-major form of round 4 -- zx_mode 4: the same arithmetic at 1.8 ms in a kernel of 5.3, because every access
// was "16 planes x 64 bytes" -- is what the tiling replaced; DESIGN.md section 4b has the measurements.)
//
// Why.  zx2_kernel needs 5R packed VALU instructions per voxel and a workgroup-wide LDS hand-off per 8 planes;
// measured, neither its arithmetic nor its skeleton (loads, LDS, barriers: 3.5 of its 5.4 ms per 64 blocks at
// R = 16) gets near the 1.8 ms its 10 B/voxel cost at HBM speed.  A 1-D convolution of
// 16 rows is a product with a banded Toeplitz matrix, and v_mfma_f32_16x16x32_f16 multiplies 16 x the flops
// per clock of the float32 pipes.  Float16 carries 11 significant bits, so every float32 factor is split into
// two float16 pieces, x = xh + xl / 2048 (xh = half(x), xl = half((x - xh) * 2048)), and a product is three
// MFMAs (xh wh -> acc0; xh wl + xl wh -> acc1; result acc0 + acc1 / 2048; the dropped xl wl term is 2^-22 of
// the product).  uint16 voxels split EXACTLY into two float16 pieces (high byte / 256 and low byte / 65536, the
// latter float16 subnormals), so the X pass, taken first and straight from global memory, has no data error at
// all.  Measured against the exact float64 values the result is as close as the float32 FMA chains were
// (bench.py: max_f32_error), far inside the eps / 4 the exactness machinery allows.
//
// Layout: no LDS, no barriers.  A WAVE owns one 16-column output tile of one block row (y) and marches along z
// in tiles of 16 planes:
//   X pass   D1[z][x_out] = sum_k In[z][x_in(k)] Wx[x_in(k)][x_out].  A operand = voxels: lane l loads 8
//            consecutive x of plane z0 + (l & 15) with ONE 16-byte load per k-step (k = 8 (l >> 4) + j) and
//            unpacks them into the two exact float16 pieces with byte permutes; B operand = the Toeplitz
//            fragments of this column (constant registers).  The accumulators come out with x_out on
//            lane & 15 and four consecutive z in the four registers --
//   Z pass   D2[x][z_out] = sum_k D1[z_in(k)][x] Wz[z_in(k)][z_out] -- which is exactly the A-operand layout
//            of the next product once two z tiles are put side by side (k = 8 (l >> 4) + j <-> tile j >> 2,
//            plane 4 (l >> 4) + (j & 3): the MFMA does not care in which order k runs as long as both
//            operands agree).  So the X results never leave the registers: they are split into float16 pieces
//            and kept in a sliding window of 2 LA + 2 z tiles; the Z Toeplitz fragments follow the same k order.
//            The result has z_out on lane & 15 and four consecutive x in the registers: one 16-byte store per
//            lane for P and for Q.
// SciPy's "reflect" boundaries are folded into the Toeplitz fragments (the weight of a mirrored tap is added
// to the tap it mirrors), so no halo is ever materialised and no load leaves the block: chunks and planes that
// would are clamped into it and get zero weights.  The fragments are built on the device by a small setup
// kernel per launch (zx4_setup: one table per distinct block width / depth of the batch, in the part of the
// workspace the fused path does not use) and live in registers in the main kernel, except the Z fragments of
// the first and last z tiles, which are reloaded when the march reaches them.

#include <type_traits>

#include "mmx_common.h"

typedef _Float16 h2_4 __attribute__((ext_vector_type(2)));
typedef _Float16 h8_4 __attribute__((ext_vector_type(8)));
typedef float f2_4 __attribute__((ext_vector_type(2)));
typedef float f4_4 __attribute__((ext_vector_type(4)));
typedef unsigned u4_4 __attribute__((ext_vector_type(4)));
typedef unsigned u2_4 __attribute__((ext_vector_type(2)));

#define MMX_ZX4_MAXCLS 8
// cache policy bits of the streaming accesses (gfx940+: 1 = sc0, 2 = nt, 16 = sc1)
#ifndef ZX4_ST_AUX
#define ZX4_ST_AUX 0
#endif
#ifndef ZX4_LD_AUX
#define ZX4_LD_AUX 0
#endif

struct mmx_zx4_cfg {
    float w0[MMX_MAX_RADIUS_FAST + 1];     // order-0 half kernel (plain)
    float w2[MMX_MAX_RADIUS_FAST + 1];     // order-2 half kernel (plain)
    float xscale;                          // 65536 / 65535 (u16) or 256 / 255 (u8): the pieces carry v / 2^16
    int radius;
    int ncw, ncz;                          // distinct block widths / depths of the batch
    int wcls[MMX_ZX4_MAXCLS];              // widths (nx)
    int zcls[MMX_ZX4_MAXCLS];              // depths (nz)
    int maxcol, maxu;                      // table extents: columns per width class, z tiles per depth class
    float qp, qq;                          // Q16 tiles: P / bound(P) and Q / bound(Q) land in [0, 1] and [-1, 1]
    int staged;                            // 0: chunks clamped into the row (zx4); 1: at their natural position (zx5); 2: same, windows at 16 c - R8 (zx6)
    int ntw;                               // column tiles per wave (tiled form): 2 = tiles (2 p, 2 p + 1) share the window of tile 2 p
};

namespace {

#ifndef ZX4_PF
#define ZX4_PF 3
#endif
constexpr int kPF4 = ZX4_PF;       // z tiles of voxels in flight per wave
#ifndef ZX6_PF
#define ZX6_PF 2
#endif
constexpr float kLoScale = 2048.f;
constexpr float kLoInv = 1.f / 2048.f;

using rsrc4_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc4_t make_rsrc4(const void* p)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}

template <int R> struct geom4 {
    static constexpr int R8 = (R + 7) & ~7;                       // x halo rounded to the 8-voxel chunks
    static constexpr int NKX = (16 + 2 * R8 + 31) / 32;           // k-steps of the X pass
    static constexpr int LA = (R + 15) / 16;                      // z tiles of look-ahead / look-back
    static constexpr int NKZ = LA + 1;                            // k-steps of the Z pass (pairs of z tiles)
    static constexpr int NT = 2 * NKZ;                            // window tiles: U - LA .. U + LA + 1
};
// geometry classes actually compiled: (NKX, LA) = (1, 1) for R <= 8, (2, 1) for R <= 16, (2, 2) for R <= 24
template <int NKX, int LA> struct cls4 {
    static constexpr int R8 = NKX == 1 ? 8 : (LA == 1 ? 16 : 24);
    // first input column of output column tile c: 16 c - R8, moved down to a multiple of 32 voxels where the
    // k-steps still reach the last input (16 c + 15 + R8) -- whole 64-byte pieces per plane for uint16
    static __host__ __device__ constexpr int xstart(int c)
    {
        const int lo = 16 * c - R8;
        const int al = lo & ~31;
        return (al + 32 * NKX >= 16 * c + 16 + R8) ? al : lo;
    }
    static constexpr int NKZ = LA + 1;
    static constexpr int NT = 2 * NKZ;
};

__device__ __forceinline__ unsigned pack_h2(float a, float b)
{
    const f2_4 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, h2_4));
}

// ------------------------------------------------------------------------------------ setup: Toeplitz fragments
// weight of input position p for output o on an axis of n voxels with SciPy "reflect" folded in
__device__ __forceinline__ float folded_tap(const float* w, int R, int p, int o, int n)
{
    float s = 0.f;
    int d = p - o; d = d < 0 ? -d : d;
    if (d <= R) s += w[d];
    d = (-1 - p) - o; d = d < 0 ? -d : d;            // left mirror image of p
    if (d <= R) s += w[d];
    d = (2 * n - 1 - p) - o; d = d < 0 ? -d : d;     // right mirror image of p
    if (d <= R) s += w[d];
    return s;
}

// One thread per (table entry, lane): 8 float16 weights (one MFMA operand register quad) as hi and lo pieces.
//   X table [class][column][m][kernel][piece][lane], Z table [class][U][ks][kernel][piece][lane]
template <int NKX, int LA>
__global__ void __launch_bounds__(256)
zx4_setup(mmx_zx4_cfg cfg, u4_4* __restrict__ xtab, u4_4* __restrict__ ztab)
{
    using cg = cls4<NKX, LA>;
    const int R = cfg.radius;
    const int lane = threadIdx.x & 63;
    const int kq = lane >> 4, col = lane & 15;
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);           // entry index over both tables: (.., kernel)
    const int nx_entries = cfg.ncw * cfg.maxcol * NKX * 2;
    const int nz_entries = cfg.ncz * cfg.maxu * cg::NKZ * 2;
    float wt[8];
    u4_4* dst;
    if (e < nx_entries) {
        const int kern = e & 1;
        const int m = (e >> 1) % NKX;
        const int c = ((e >> 1) / NKX) % cfg.maxcol;
        const int cl = ((e >> 1) / NKX) / cfg.maxcol;
        const int W = cfg.wcls[cl];
        const float* w = kern ? cfg.w2 : cfg.w0;
        const int xo = 16 * c + col;
        // natural start of this lane's chunk (tiled form, staged == 2: windows start at 16 c - R8, any multiple of 8)
        // (two tiles per wave: the odd tile of a pair reads the even one's window, 16 columns further left)
        const int cwin = cfg.ntw == 2 ? (c & ~1) : c;
        const int xc = (cfg.staged == 2 ? 16 * cwin - cg::R8 : cg::xstart(c)) + 32 * m + 8 * kq;
        int xl = xc < 0 ? 0 : xc;
        xl = xl > W - 8 ? W - 8 : xl;                               // where the kernel loads it from
        if (cfg.staged) xl = xc;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int p = xl + j;
            const bool ok = p >= xc && p < xc + 8 && p >= 0 && p < W && xo < W;
            // (16-bit tiles: their scales ride in the weights -- G(x) carries 1 / bound(P), G''(x) 1 / bound(Q), and
            //  the order-2 Z fragments below the ratio -- so that the kernel's results come out ready to convert)
            wt[j] = ok ? folded_tap(w, R, p, xo, W) * cfg.xscale * (cfg.qp > 0.f ? (kern ? cfg.qq : cfg.qp) : 1.f) : 0.f;
        }
        dst = xtab + (size_t)e * 2 * 64;
    } else if (e < nx_entries + nz_entries) {
        const int ez = e - nx_entries;
        const int kern = ez & 1;
        const int ks = (ez >> 1) % cg::NKZ;
        const int U = ((ez >> 1) / cg::NKZ) % cfg.maxu;
        const int cl = ((ez >> 1) / cg::NKZ) / cfg.maxu;
        const int nz = cfg.zcls[cl];
        const float* w = kern ? cfg.w2 : cfg.w0;
        const int zo = 16 * U + col;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = 2 * ks + (j >> 2);                        // window tile of this k
            const int zi = 16 * (U - LA + i) + 4 * kq + (j & 3);
            const bool ok = i < cg::NT - 1 && zi >= 0 && zi < nz && zo < nz;
            wt[j] = ok ? folded_tap(w, R, zi, zo, nz) * (cfg.qp > 0.f && kern ? cfg.qq / cfg.qp : 1.f) : 0.f;
        }
        dst = ztab + (size_t)ez * 2 * 64;
    } else {
        return;
    }
    unsigned hi[4], lo[4];
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const f2_4 v = {wt[j], wt[j + 1]};
        const h2_4 h = __builtin_convertvector(v, h2_4);
        const f2_4 r = {(wt[j] - (float)h.x) * kLoScale, (wt[j + 1] - (float)h.y) * kLoScale};
        hi[j >> 1] = __builtin_bit_cast(unsigned, h);
        lo[j >> 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, h2_4));
    }
    dst[lane] = (u4_4){hi[0], hi[1], hi[2], hi[3]};
    dst[64 + lane] = (u4_4){lo[0], lo[1], lo[2], lo[3]};
}

// ------------------------------------------------------------------------------------------------ main kernel
template <typename InT> struct pieces4;
// 8 consecutive uint16 voxels (4 dwords) -> high bytes / 256 and low bytes / 65536 as packed float16:
//   0x4400 | b  =  4 + b / 256      (float16, exponent 2^2, ulp 2^-8)
//   0x2400 | b  =  2^-6 + b / 65536 (exponent 2^-6, ulp 2^-16)
template <> struct pieces4<uint16_t> {
    static constexpr int NP = 2;
    using raw_t = u4_4;
    static __device__ __forceinline__ raw_t load(rsrc4_t r, unsigned off, unsigned soff = 0)
    {
        return __builtin_bit_cast(u4_4, __builtin_amdgcn_raw_buffer_load_b128(r, off, soff, ZX4_LD_AUX));
    }
    static __device__ __forceinline__ void split(const raw_t& d, u4_4& hi, u4_4& lo)
    {
        const h2_4 four = {(_Float16)4.0f, (_Float16)4.0f};
        const h2_4 sixty4th = {(_Float16)0.015625f, (_Float16)0.015625f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // v_perm_b32: bytes of {S0 = constant, S1 = data}; selector bytes 0-3 pick from S1, 4-7 from S0
            const unsigned th = __builtin_amdgcn_perm(0x44004400u, d[i], 0x07030501u);   // [44 b3 44 b1]
            const unsigned tl = __builtin_amdgcn_perm(0x24002400u, d[i], 0x07020500u);   // [24 b2 24 b0]
            hi[i] = __builtin_bit_cast(unsigned, __builtin_bit_cast(h2_4, th) - four);
            lo[i] = __builtin_bit_cast(unsigned, __builtin_bit_cast(h2_4, tl) - sixty4th);
        }
    }
    // The same pieces with their exponent offsets left in: hi = 4 + b / 256, lo = 2^-6 + b / 65536.  A constant added
    // to every element of an A operand adds (constant x column sum of B) to every row of the product: the kernel
    // starts its accumulators at minus that, computed once per wave -- half the split's instructions.
    static constexpr float kBiasHi = 4.0f, kBiasLo = 0.015625f;
    static __device__ __forceinline__ void split_biased(const raw_t& d, u4_4& hi, u4_4& lo)
    {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            hi[i] = __builtin_amdgcn_perm(0x44004400u, d[i], 0x07030501u);
            lo[i] = __builtin_amdgcn_perm(0x24002400u, d[i], 0x07020500u);
        }
    }
};
// (uint8 voxels: zx6_pack_kernel widens them to uint16, shifted into the high byte -- the uint16 pieces serve both)

// float voxels: zx6_pack_f32_kernel has split them already -- per unit of 8 columns x 16 planes the high
// float16 pieces (256 bytes) then the low ones, v = hi + lo / 2048 -- two 16-byte loads per k-step, no unpacking
struct presplit_t { u4_4 h, l; };
template <> struct pieces4<float> {
    [[maybe_unused]] static constexpr int NP = 2;
    using raw_t = presplit_t;
    static __device__ __forceinline__ raw_t load(rsrc4_t r, unsigned off, unsigned soff = 0)
    {
        raw_t v;
        v.h = __builtin_bit_cast(u4_4, __builtin_amdgcn_raw_buffer_load_b128(r, off, soff, ZX4_LD_AUX));
        v.l = __builtin_bit_cast(u4_4, __builtin_amdgcn_raw_buffer_load_b128(r, off + 256u, soff, ZX4_LD_AUX));
        return v;
    }
    static __device__ __forceinline__ void split(const raw_t& d, u4_4& hi, u4_4& lo) { hi = d.h; lo = d.l; }
    [[maybe_unused]] static constexpr float kBiasHi = 0.f, kBiasLo = 0.f;      // (never biased: read in discarded branches only)
    static __device__ __forceinline__ void split_biased(const raw_t& d, u4_4& hi, u4_4& lo) { hi = d.h; lo = d.l; }
};
template <typename InT> struct is_f32_4 { static constexpr bool value = false; };
template <> struct is_f32_4<float> { static constexpr bool value = true; };
template <typename InT> struct lo_scaled4 { static constexpr bool value = false; };     // low piece carries x 2048?
template <> struct lo_scaled4<float> { static constexpr bool value = true; };

__device__ __forceinline__ f4_4 mfma16(const u4_4& a, const u4_4& b, const f4_4& c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_4, a), __builtin_bit_cast(h8_4, b), c, 0, 0, 0);
}

// The voxels come from the operand-ordered copy zx6_pack_kernel leaves (`vol` = that copy,
// stride_z = its elements per block) and P / Q leave as 16 x 16 tiles of 1 KiB (slot_elems = tile elements per
// block) in (y, c, U) order, which y6_kernel (mmx_fused.hip) reads: every global access of a wave is then one
// contiguous KiB.
// Q16 (zx_mode 7): the tile holds one dword per voxel, P as unorm16 of P / bound(P) in the low half and Q as snorm16
// of Q / bound(Q) in the high half (the bounds follow from the weights alone: mmx_tiled_q16_error_bound) -- half
// the bytes for the Y pass to read and for this kernel to write, at a known error that the caller's band must cover.
// NTW = 2 (Q16, 8 < radius <= 16): a wave owns TWO adjacent column tiles (2 p, 2 p + 1).  Their windows -- 48
// of the 64 columns two k-steps load -- overlap by two thirds, and the 64 columns from 16 c - 16 on hold both: the
// same two loads per z step now feed two output tiles (half the L2 read requests per tile: the counters had shown
// 4.3 x as many bytes requested from L2 as read from HBM, neighbouring waves re-reading each other's windows), and
// the voxels are unpacked into float16 pieces once for both (-16 of ~92 VALU instructions per tile step).  The second
// tile's X fragments and window are another 64 registers: two waves per SIMD instead of three, each with twice the
// independent work per step.  An odd tile count leaves the last wave of a row one tile: it computes the second with
// zero weights and its stores are dropped by a zero-length buffer descriptor (no branch in the march).
template <int NKX, int LA, typename InT, bool Q16 = false, int NTW = 1>
__global__ void __launch_bounds__(256, (NTW == 1 && (LA == 1 || Q16) && !is_f32_4<InT>::value) ? 3 : 2)   // (float32 tiles, LA == 2: 232 registers)
zx4_kernel(const InT* __restrict__ vol, int64_t stride_z, int64_t stride_y,
           const mmx_block* __restrict__ blocks, int64_t slot_elems,
           float* __restrict__ gp, float* __restrict__ gq,
           const u4_4* __restrict__ xtab, const u4_4* __restrict__ ztab, mmx_zx4_cfg cfg)
{
    using cg = cls4<NKX, LA>;
    using pc = pieces4<InT>;
    constexpr int NKZ = cg::NKZ, NT = cg::NT;
    constexpr int kUnit = std::is_same<InT, float>::value ? 512 : 256;      // bytes of a unit of the voxel copy
    constexpr bool LO_SCALED = lo_scaled4<InT>::value;
    static_assert(NTW == 1 || (NTW == 2 && Q16 && NKX == 2 && LA == 1), "two tiles per wave: 16-bit tiles, 8 < radius <= 16");
    // radius <= 16: the Z fragments of the interior z tiles live in LDS, shared by the workgroup's waves, and
    // two z tiles are in flight instead of three: 154 registers, three waves per SIMD.  (Radius > 16 has three
    // k-steps of Z fragments: with float32 tiles, which are bound by their stores, fetching them from LDS every step
    // costs more than the third wave gives -- 4.19 against 3.87 ms --; with 16-bit tiles, 168 registers, it pays:
    // 3.10 against 3.15 ms.)
    constexpr bool ZLDS = LA == 1 || Q16;
    constexpr int PF = ZLDS ? ZX6_PF : kPF4;         // z tiles of voxels in flight per wave
    const mmx_block bd = blocks[blockIdx.y];
    const int W = bd.nx, nz = bd.nz;
    const int ntx = (W + 15) >> 4;
    // (readfirstlane: the wave index is uniform, but only this tells the compiler -- otherwise every buffer
    //  descriptor below is built per lane and each load becomes a waterfall loop)
    // workgroups are dealt to the 8 XCDs round robin; neighbours along x read overlapping windows, so the
    // workgroups one XCD gets (every 8th) are made neighbours: its L2 then serves the overlap (gridDim.x % 8 == 0)
    const int bx = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    const int gw = bx * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave of this block: (y, column)
    const int ntp = (ntx + NTW - 1) / NTW;                      // waves per row: one per tile, or per pair of tiles
    const int y = gw / ntp;
    const int c = (gw - y * ntp) * NTW;                         // (first) column tile of this wave
    const int lane = threadIdx.x & 63;
    const int li = lane & 15, kq = lane >> 4;
    int cw = 0, cz = 0;
    for (int i = 1; i < cfg.ncw; ++i) cw = cfg.wcls[i] == W ? i : cw;
    for (int i = 1; i < cfg.ncz; ++i) cz = cfg.zcls[i] == nz ? i : cz;
    const int ntz = (nz + 15) >> 4;
    const int R = cfg.radius;
    const int u_lo = (R + 15) >> 4;                             // first U with 16 U - R >= 0
    const int u_hi = (nz - 16 - R) >> 4;                        // last U with 16 U + 15 + R <= nz - 1 (may be < u_lo)
    const u4_4* zt = ztab + ((size_t)cz * cfg.maxu * NKZ * 2) * 128 + lane;
    // (the steps at the ends of the block take their fragments straight from the table)
    // ROT: in the steady steps z tile t keeps the window slot t mod NT it was written to -- no shifting, the operand
    // quads (slots 2 q, 2 q + 1) never move -- and the fragments rotate instead: by whole k-steps when the window
    // starts on an even slot (the table as it is), and through a second table, shifted by one tile, when it starts on
    // an odd one: F1[k] = [F0[k - 1] upper half | F0[k] lower half].  The slot outside the window holds the tile that
    // left it (finite values) against zero fragments.  A third fewer register moves per step; built where it measured
    // faster (tools/kbench.py, 16-bit tiles, ms per 22 blocks: radius 8 0.752 -> 0.70, radius 18 / 20 1.02 -> 0.975);
    // two column tiles per wave spill with it (76 bytes: 0.75 -> 1.20), one tile per wave at radius 9..16 loses to
    // the pair as before (0.845 against 0.74), float32 tiles read +6 %: profiles/r05_experiments.txt, section 6.
    constexpr bool ROT = ZLDS && Q16 && NTW == 1 && !is_f32_4<InT>::value && (NKX == 1 || LA == 2);
    constexpr int ZL1 = NKZ * 4 * 64;                          // entries of one table
    __shared__ u4_4 zl[ZLDS ? (ROT ? 2 : 1) * ZL1 : 1];
    if constexpr (ZLDS) {
        const u4_4* zi = ztab + ((size_t)(cz * cfg.maxu + (u_lo < cfg.maxu ? u_lo : 0)) * NKZ * 2) * 128;
        for (int e = threadIdx.x; e < ZL1; e += 256) zl[e] = zi[e];
        __syncthreads();
        if constexpr (ROT) {
            for (int e = threadIdx.x; e < ZL1; e += 256) {
                const u4_4 cur = zl[e];
                const u4_4 prev = e >= 256 ? zl[e - 256] : (u4_4){0u, 0u, 0u, 0u};  // (k-step before: 4 x 64 entries back)
                zl[ZL1 + e] = (u4_4){prev.z, prev.w, cur.x, cur.y};
            }
            __syncthreads();
        }
    }
    if (y >= bd.ny) return;                                   // whole wave (no barriers below)

    // X fragments of this wave's column tile(s): [tile][m][kernel][piece]
    u4_4 xw[NTW][NKX][2][2];
    const bool has2 = NTW == 2 && c + 1 < ntx;                 // (wave-uniform)
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        // (a pair's missing second tile: any column's fragments -- its results are never stored)
        const u4_4* xt = xtab + ((size_t)(cw * cfg.maxcol + (i && !has2 ? c : c + i)) * NKX * 2) * 128 + lane;
#pragma unroll
        for (int m = 0; m < NKX; ++m)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                xw[i][m][k][0] = xt[(m * 2 + k) * 128];
                xw[i][m][k][1] = xt[(m * 2 + k) * 128 + 64];
            }
    }
    // 16-bit tiles: the voxel pieces keep their exponent offsets (pieces4::split_biased) and the X accumulators start
    // at minus what the offsets add -- (offset x column sum of the fragments), the same for the four rows a lane holds.
    // The sums come from the matrix cores themselves: an A operand of ones.  Their float32 rounding (values of ~5
    // instead of <= 1: 5e-7) is inside what mmx_tiled_q16_error_bound states.
    // (radius > 16: the eight start registers would push the kernel past three waves per SIMD)
    constexpr bool BIASED = Q16 && !is_f32_4<InT>::value && LA == 1;
    constexpr bool MIXSPLIT = Q16;
    const f4_4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f4_4 a_start[NTW], b_start[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) { a_start[i] = zero4; b_start[i] = zero4; }
    if constexpr (BIASED) {
        const u4_4 ones = {0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            f4_4 sh[2] = {zero4, zero4}, sl[2] = {zero4, zero4};
#pragma unroll
            for (int m = 0; m < NKX; ++m)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    sh[k] = mfma16(ones, xw[i][m][k][0], sh[k]);
                    sl[k] = mfma16(ones, xw[i][m][k][1], sl[k]);
                }
            // a0 takes (hi + lo pieces) x high fragments, a1 hi pieces x low fragments (scaled by 2048), v = a0 + a1 / 2048
            const float ca = -((pc::kBiasHi + (pc::NP == 2 ? pc::kBiasLo : 0.f)) * sh[0][0] + pc::kBiasHi * sl[0][0] * kLoInv);
            const float cb = -((pc::kBiasHi + (pc::NP == 2 ? pc::kBiasLo : 0.f)) * sh[1][0] + pc::kBiasHi * sl[1][0] * kLoInv);
            a_start[i] = (f4_4){ca, ca, ca, ca};
            b_start[i] = (f4_4){cb, cb, cb, cb};
        }
    }
    // Z fragments: [ks][kernel][piece], of the z tile being produced (reloaded when its class changes);
    // interior z tiles share one set: the first tile whose taps all fall inside the block
    u4_4 zw[ZLDS ? 1 : NKZ][2][2];
    int zset = -1;

    // voxel rows: plane z0 + li, chunk of 8 x at xl[m] (clamped into the block; the fragments know)
    const InT* in = vol + (int64_t)bd.slot * stride_z;
    const int nch8 = (W + 7) >> 3;                               // 256-byte units (8 columns x 16 planes) per row tile
    unsigned xoff[NKX];
#pragma unroll
    for (int m = 0; m < NKX; ++m) {
        int j = 2 * c - cg::R8 / 8 + 4 * m + kq;                 // unit of this lane; outside the row: any unit, zero weights
        j = j < 0 ? 0 : (j > nch8 - 1 ? nch8 - 1 : j);
        xoff[m] = (unsigned)(j * kUnit + li * 16);
    }
    const rsrc4_t rin = make_rsrc4(in);
    auto load_tile = [&](int t, typename pc::raw_t (&raw)[NKX]) __attribute__((always_inline)) {
        const unsigned so = (unsigned)((y * ntz + t) * nch8) * (unsigned)kUnit;          // wave-uniform: the row tile
#pragma unroll
        for (int m = 0; m < NKX; ++m) raw[m] = pc::load(rin, xoff[m], so);
    };

    // window of X results as float16 pieces: [column tile][array: A hi, A lo, B hi, B lo][z tile][2 dwords]
    unsigned win[NTW][4][NT][2];
#pragma unroll
    for (int w = 0; w < NTW; ++w)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int i = 0; i < NT; ++i) { win[w][a][i][0] = 0u; win[w][a][i][1] = 0u; }

    // One descriptor per wave, over the tile column (y, c, all U) it writes -- ntz KiB -- and ENDING where the
    // block's planes end: the rows of the last z tile past the block (11 of 16 at 261 planes) fall outside it and the
    // hardware drops their part of the store, 4 % of the kernel's write requests, without a branch or an exec mask.
    // Nothing reads them: the Y pass works within a plane and discards the planes past the block (ym_kernel: `real`).
    // (Second tile of a pair: the next ntz KiB, 0 records when the row has no such tile -- its stores are dropped.)
    const unsigned col_b = (unsigned)ntz * 1024u;                                        // bytes of one tile column
    const unsigned live_b = col_b - (unsigned)(16 * ntz - nz) * 64u;
    const int64_t wave_e = (int64_t)((y * ntx + c) * ntz) * 256;                          // elements before this wave's tiles
    const rsrc4_t rp = __builtin_amdgcn_make_buffer_rsrc(gp + (int64_t)bd.slot * slot_elems + wave_e, 0, (int)live_b, 0x00020000);
    const rsrc4_t rq = __builtin_amdgcn_make_buffer_rsrc(gq + (int64_t)bd.slot * slot_elems + wave_e, 0, (int)live_b, 0x00020000);
    const rsrc4_t rp2 = __builtin_amdgcn_make_buffer_rsrc(gp + (int64_t)bd.slot * slot_elems + wave_e + (int64_t)ntz * 256, 0,
                                                          has2 ? (int)live_b : 0, 0x00020000);
    // Tile (y, c, U) of 16 z x 16 x floats, row-major, at ((y ntx + c) ntz + U) KiB: what a wave writes
    // during its march is contiguous, the waves of a workgroup and the workgroups of a row follow each other --
    // the kernel's stores are one sequential stream -- and y6_kernel's workgroups, one per (c, U), all read inside
    // the same ntx ntz KiB at any time.  (Measured against the (c, U, y) order: no difference in either kernel; both
    // run at the request rate the memory system sustains)
    const unsigned plane_b = 64u;
    unsigned obase = (unsigned)((4 * li + kq) * 16);                  // (relative to the wave's tile column)

    // voxels of the next ZX4_PF z tiles, in flight.  vmcnt counts loads and stores together and in issue order:
    // a tile loaded only one step ahead could not be used before the stores of the step in between have been
    // acknowledged by L2 -- measured, that wait and not the arithmetic set the step time (5.3 ms against 2.1 ms
    // without the stores).  With PF steps of distance the stores older than the tile being consumed have had
    // PF - 1 whole steps to complete.
    typename pc::raw_t raw[PF][NKX];
#pragma unroll
    for (int u = 0; u < PF; ++u) load_tile(u < ntz ? u : ntz - 1, raw[u]);

    // One march step: X pass of z tile t into the window, Z pass of z tile t - LA out of it.  STEADY = the
    // branch-free form for the tiles in the middle of the block (every tile real, interior Z fragments already
    // in registers, every output plane real): with no control flow between the memory instructions the
    // compiler's s_waitcnt vmcnt(N) are exact counts and only ever wait for the tile being consumed.
    // Results on their way to memory: a store reads its data registers asynchronously, so the compiler makes the
    // next writer of those registers wait (vmcnt) until the store has completed.  Each ring slot therefore has
    // its own result registers, kept allocated (an empty asm "use") until the slot comes round again.
    f4_4 outP[PF][NTW], outQ[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
#pragma unroll
        for (int w = 0; w < NTW; ++w) outP[u][w] = zero4;
        outQ[u] = zero4;
    }
    // (phase_tag: the window slot of z tile t in a steady step of the rotating form, -1 otherwise: the window is
    //  shifted and t takes slot 2 LA -- the arrangement the rotating steps start from and return to every NT steps)
    auto step = [&](int t, auto steady_tag, auto phase_tag, typename pc::raw_t (&rw)[NKX], f4_4 (&P)[NTW], f4_4& Q) __attribute__((always_inline)) {
        constexpr bool STEADY = decltype(steady_tag)::value;
        constexpr int PH = decltype(phase_tag)::value;
        constexpr bool TURN = STEADY && ROT && PH >= 0;
        constexpr int SLOT = TURN ? PH : 2 * LA;                 // where the X results of z tile t go
        if constexpr (!TURN) {
            // shift the window by one z tile
#pragma unroll
            for (int w = 0; w < NTW; ++w)
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int i = 0; i < 2 * LA; ++i) { win[w][a][i][0] = win[w][a][i + 1][0]; win[w][a][i][1] = win[w][a][i + 1][1]; }
        }
        if (STEADY || t < ntz) {
            // ---- X pass of z tile t
            u4_4 dh[NKX], dl[NKX];
#pragma unroll
            for (int m = 0; m < NKX; ++m) {
                if constexpr (BIASED) pc::split_biased(rw[m], dh[m], dl[m]);
                else pc::split(rw[m], dh[m], dl[m]);
            }
            // (past the last tile the ring re-reads it: loads without a branch, never used)
            // the scheduler must not sink these loads to the end of the unrolled ring (it does, given the chance:
            // all of a ring's loads then sit right in front of their first use)
            __builtin_amdgcn_sched_barrier(0);
            if (STEADY) load_tile(t + PF < ntz ? t + PF : ntz - 1, rw);
            else if (t + PF < ntz) load_tile(t + PF, rw);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int w = 0; w < NTW; ++w) {
                f4_4 a0 = a_start[w], a1 = zero4, b0 = b_start[w], b1 = zero4;
#pragma unroll
                for (int m = 0; m < NKX; ++m) {
                    a0 = mfma16(dh[m], xw[w][m][0][0], a0);
                    b0 = mfma16(dh[m], xw[w][m][1][0], b0);
                    a1 = mfma16(dh[m], xw[w][m][0][1], a1);
                    b1 = mfma16(dh[m], xw[w][m][1][1], b1);
                    if constexpr (LO_SCALED) {
                        // (float voxels: the low piece carries x 2048 like the low fragments; low x low is 2^-22 of a product)
                        a1 = mfma16(dl[m], xw[w][m][0][0], a1);
                        b1 = mfma16(dl[m], xw[w][m][1][0], b1);
                    } else if constexpr (pc::NP == 2) {
                        a0 = mfma16(dl[m], xw[w][m][0][0], a0);
                        b0 = mfma16(dl[m], xw[w][m][1][0], b0);
                        // (low voxel byte x low weight piece: <= 2^-19 of a product.  The float32 tiles keep it; the 16-bit
                        //  tiles' error bound has room for it: mmx_tiled_q16_error_bound counts 1.9e-6 per X sum)
                        if constexpr (!Q16) {
                            a1 = mfma16(dl[m], xw[w][m][0][1], a1);
                            b1 = mfma16(dl[m], xw[w][m][1][1], b1);
                        }
                    }
                }
                // combine the two accumulators and split into float16 pieces: v = acc0 + acc1 / 2048
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const float av0 = __builtin_fmaf(a1[r], kLoInv, a0[r]), av1 = __builtin_fmaf(a1[r + 1], kLoInv, a0[r + 1]);
                    const float bv0 = __builtin_fmaf(b1[r], kLoInv, b0[r]), bv1 = __builtin_fmaf(b1[r + 1], kLoInv, b0[r + 1]);
                    const f2_4 av = {av0, av1}, bv = {bv0, bv1};
                    const h2_4 ah = __builtin_convertvector(av, h2_4), bh = __builtin_convertvector(bv, h2_4);
                    if constexpr (MIXSPLIT) {
                        // 16-bit tiles: the low piece is the plain residual v - half(v), not scaled by 2048 -- values are
                        // <= 1 here (the fragments carry 1 / bound), so the residual is below 2^-12 and float16 keeps it to
                        // 2^-24 absolute, a 250th of the 16-bit quantum; the Z pass below adds it at full weight.
                        // (v_fma_mix{lo,hi}_f16 would make it in one instruction per value, but only as inline assembly,
                        //  whose register writes the compiler's hazard recogniser does not see: one landed right behind an
                        //  MFMA that still read the register as its C operand -- wrong results for radius <= 8.)
                        const f2_4 ar = {av0 - (float)ah.x, av1 - (float)ah.y}, br = {bv0 - (float)bh.x, bv1 - (float)bh.y};
                        win[w][0][SLOT][r >> 1] = __builtin_bit_cast(unsigned, ah);
                        win[w][1][SLOT][r >> 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(ar, h2_4));
                        win[w][2][SLOT][r >> 1] = __builtin_bit_cast(unsigned, bh);
                        win[w][3][SLOT][r >> 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(br, h2_4));
                        continue;
                    }
                    const f2_4 ar = {__builtin_fmaf((float)ah.x, -kLoScale, av0 * kLoScale), __builtin_fmaf((float)ah.y, -kLoScale, av1 * kLoScale)};
                    const f2_4 br = {__builtin_fmaf((float)bh.x, -kLoScale, bv0 * kLoScale), __builtin_fmaf((float)bh.y, -kLoScale, bv1 * kLoScale)};
                    win[w][0][SLOT][r >> 1] = __builtin_bit_cast(unsigned, ah);
                    win[w][1][SLOT][r >> 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(ar, h2_4));
                    win[w][2][SLOT][r >> 1] = __builtin_bit_cast(unsigned, bh);
                    win[w][3][SLOT][r >> 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(br, h2_4));
                }
            }
        } else {
#pragma unroll
            for (int w = 0; w < NTW; ++w)
#pragma unroll
                for (int a = 0; a < 4; ++a) { win[w][a][SLOT][0] = 0u; win[w][a][SLOT][1] = 0u; }
        }
        const int U = t - LA;
        if (STEADY || U >= 0) {
            // ---- Z pass of z tile U
            const int want = (U >= u_lo && U <= u_hi) ? u_lo : U;
            if constexpr (!STEADY && !ZLDS) {
                if (want != zset) {
                    zset = want;
#pragma unroll
                    for (int ks = 0; ks < NKZ; ++ks)
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            zw[ks][k][0] = zt[((size_t)(want * NKZ + ks) * 2 + k) * 128];
                            zw[ks][k][1] = zt[((size_t)(want * NKZ + ks) * 2 + k) * 128 + 64];
                        }
                }
            }
            f4_4 p0[NTW], p1[NTW], q0[NTW], q1[NTW];
#pragma unroll
            for (int w = 0; w < NTW; ++w) { p0[w] = zero4; p1[w] = zero4; q0[w] = zero4; q1[w] = zero4; }
#pragma unroll
            for (int ks = 0; ks < NKZ; ++ks) {
                u4_4 z00, z01, z10, z11;          // [kernel][piece]
                if constexpr (!ZLDS) {
                    z00 = zw[ks][0][0]; z01 = zw[ks][0][1]; z10 = zw[ks][1][0]; z11 = zw[ks][1][1];
                } else if constexpr (STEADY) {
                    // (rotating form: the window starts TR slots after slot 0; quad ks holds the window tiles of k-step
                    //  ks - TR / 2, one tile later when TR is odd)
                    constexpr int TR = TURN ? (PH - (NT - 2) + NT) % NT : 0;
                    const int KF = ((TR & 1) ? ZL1 : 0) + ((ks - TR / 2 + NKZ) % NKZ) * 256;     // (a constant once unrolled)
                    z00 = zl[KF + 0 * 64 + lane]; z01 = zl[KF + 1 * 64 + lane];
                    z10 = zl[KF + 2 * 64 + lane]; z11 = zl[KF + 3 * 64 + lane];
                } else {
                    const u4_4* zp = zt + ((size_t)(want * NKZ + ks) * 2) * 128;
                    z00 = zp[0]; z01 = zp[64]; z10 = zp[128]; z11 = zp[192];
                }
                // (The window's last tile, U + LA, stands alone in its k-step -- the fragments' other half is zero.  The
                //  legacy v_mfma_f32_16x16x16_f16 on that half costs what the 16x16x32 form costs on gfx950 and, mixed
                //  with it on one accumulator chain, gave run-dependent values: profiles/r05_experiments.txt, section 5.)
#pragma unroll
                for (int w = 0; w < NTW; ++w) {
                    const u4_4 ah = {win[w][0][2 * ks][0], win[w][0][2 * ks][1], win[w][0][2 * ks + 1][0], win[w][0][2 * ks + 1][1]};
                    const u4_4 al = {win[w][1][2 * ks][0], win[w][1][2 * ks][1], win[w][1][2 * ks + 1][0], win[w][1][2 * ks + 1][1]};
                    const u4_4 bh = {win[w][2][2 * ks][0], win[w][2][2 * ks][1], win[w][2][2 * ks + 1][0], win[w][2][2 * ks + 1][1]};
                    const u4_4 bl = {win[w][3][2 * ks][0], win[w][3][2 * ks][1], win[w][3][2 * ks + 1][0], win[w][3][2 * ks + 1][1]};
                    p0[w] = mfma16(ah, z00, p0[w]);
                    q0[w] = mfma16(bh, z00, q0[w]);
                    p1[w] = mfma16(ah, z01, p1[w]);
                    q1[w] = mfma16(bh, z01, q1[w]);
                    if constexpr (MIXSPLIT) {          // (unscaled low pieces: into the full-weight accumulators)
                        p0[w] = mfma16(al, z00, p0[w]);
                        q0[w] = mfma16(bl, z00, q0[w]);
                    } else {
                        p1[w] = mfma16(al, z00, p1[w]);
                        q1[w] = mfma16(bl, z00, q1[w]);
                    }
                    q0[w] = mfma16(ah, z10, q0[w]);
                    q1[w] = mfma16(ah, z11, q1[w]);
                    if constexpr (MIXSPLIT) q0[w] = mfma16(al, z10, q0[w]);
                    else q1[w] = mfma16(al, z10, q1[w]);
                }
                if constexpr (ZLDS && !STEADY) __builtin_amdgcn_sched_barrier(0);    // one k-step's fragments at a time
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int w = 0; w < NTW; ++w) asm volatile("" ::"v"(P[w]));     // the slot's previous results stayed in these registers until now
            asm volatile("" ::"v"(Q));
            {                                              // (a tile is stored whole: its padding belongs to it)
                // the results reach the slot's registers through opaque moves: the stores then read registers
                // that nothing else may be allocated to before the slot comes round again
                if constexpr (Q16) {
#pragma unroll
                    for (int w = 0; w < NTW; ++w) {
#pragma unroll
                        for (int r = 0; r < 4; r += 2) {
                            const float pa = __builtin_fmaf(p1[w][r], kLoInv, p0[w][r]), pb = __builtin_fmaf(p1[w][r + 1], kLoInv, p0[w][r + 1]);
                            const float qa = __builtin_fmaf(q1[w][r], kLoInv, q0[w][r]), qb = __builtin_fmaf(q1[w][r + 1], kLoInv, q0[w][r + 1]);
                            const unsigned pu = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pknorm_u16(pa, pb));   // [Pa | Pb]
                            const unsigned qs = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pknorm_i16(qa, qb));   // [Qa | Qb]
                            const unsigned da = __builtin_amdgcn_perm(qs, pu, 0x05040100u);     // [Pa | Qa]
                            const unsigned db = __builtin_amdgcn_perm(qs, pu, 0x07060302u);     // [Pb | Qb]
                            float ra, rb;
                            asm volatile("v_mov_b32 %0, %1" : "=v"(ra) : "v"(__uint_as_float(da)));
                            asm volatile("v_mov_b32 %0, %1" : "=v"(rb) : "v"(__uint_as_float(db)));
                            P[w][r] = ra;
                            P[w][r + 1] = rb;
                        }
                        // (tile (y, c + w, U) lies ntz KiB after tile (y, c, U): rp2's base)
                        if (w == 0) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_4, P[w]), rp, obase, 0, ZX4_ST_AUX);
                        else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_4, P[w]), rp2, obase, 0, ZX4_ST_AUX);
                    }
                } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __builtin_fmaf(p1[0][r], kLoInv, p0[0][r]);
                    const float qv = __builtin_fmaf(q1[0][r], kLoInv, q0[0][r]);
                    float pr, qr;
                    asm volatile("v_mov_b32 %0, %1" : "=v"(pr) : "v"(pv));
                    asm volatile("v_mov_b32 %0, %1" : "=v"(qr) : "v"(qv));
                    P[0][r] = pr;
                    Q[r] = qr;
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_4, P[0]), rp, obase, 0, ZX4_ST_AUX);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_4, Q), rq, obase, 0, ZX4_ST_AUX);
                }
            }
            obase += 16u * plane_b;
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // tiles [0, tA): generic steps up to the first interior output tile (U = t - LA >= u_lo), rounded up to a
    // multiple of the ring; [tA, tB): steady steps (U in [u_lo, u_hi], t < ntz), whole rings; the rest generic.
    const int t_end = ntz + LA;
    int tA = ((u_lo + LA + PF - 1) / PF) * PF;
    int nB = (u_hi + LA + 1 < ntz ? u_hi + LA + 1 : ntz) - tA;       // steady steps available
    constexpr int GROUP = ROT ? NT : PF;                             // (rotating form: whole turns of the window)
    static_assert(GROUP % PF == 0, "a turn of the window is a whole number of rings");
    nB = nB > 0 ? (nB / GROUP) * GROUP : 0;
    if (tA > t_end) tA = ((t_end + PF - 1) / PF) * PF;
    const int tB = tA + nB;
#pragma unroll 1
    for (int t0 = 0; t0 < tA; t0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u)
            if (t0 + u < t_end) step(t0 + u, std::false_type{}, std::integral_constant<int, -1>{}, raw[u], outP[u], outQ[u]);
    }
    if (nB > 0) {
        // interior Z fragments (the generic steps may have left an edge set in the registers)
        if (!ZLDS && zset != u_lo) {
            zset = u_lo;
#pragma unroll
            for (int ks = 0; ks < NKZ; ++ks)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    zw[ks][k][0] = zt[((size_t)(u_lo * NKZ + ks) * 2 + k) * 128];
                    zw[ks][k][1] = zt[((size_t)(u_lo * NKZ + ks) * 2 + k) * 128 + 64];
                }
        }
        // nothing may be pending at the loop head: a wait for the fragment loads above, placed inside the loop
        // at their first use, would be a static vmcnt(N) that in steady state waits for the previous step's stores
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
        // (rotating form: step j of a turn writes slot NT - 1 + j (mod NT): the generic steps leave tile t in slot NT - 2)
        auto turn = [&](int t0, auto... J) __attribute__((always_inline)) {
            (step(t0 + decltype(J)::value, std::true_type{},
                  std::integral_constant<int, ROT ? (NT - 1 + decltype(J)::value) % NT : -1>{},
                  raw[decltype(J)::value % PF], outP[decltype(J)::value % PF], outQ[decltype(J)::value % PF]), ...);
        };
        if constexpr (!ROT) {
#pragma unroll 1
            for (int t0 = tA; t0 < tB; t0 += PF) {
#pragma unroll
                for (int u = 0; u < PF; ++u) step(t0 + u, std::true_type{}, std::integral_constant<int, -1>{}, raw[u], outP[u], outQ[u]);
            }
        } else {
#pragma unroll 1
            for (int t0 = tA; t0 < tB; t0 += GROUP) {
                using std::integral_constant;
                if constexpr (GROUP == 4) turn(t0, integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, 2>{},
                                               integral_constant<int, 3>{});
                else turn(t0, integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, 2>{},
                          integral_constant<int, 3>{}, integral_constant<int, 4>{}, integral_constant<int, 5>{});
                static_assert(GROUP == 4 || GROUP == 6, "turn lengths built: 4 (radius <= 16), 6 (radius <= 24)");
            }
        }
    }
#pragma unroll 1
    for (int t0 = tB; t0 < t_end; t0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u)
            if (t0 + u < t_end) step(t0 + u, std::false_type{}, std::integral_constant<int, -1>{}, raw[u], outP[u], outQ[u]);
    }
}



// ------------------------------------------------------------------------------- tiled variant (zx_mode 6)
// Operand-ordered copy of the blocks' voxels, made once per batch: for every block row y and z tile t the row
// tile of 16 planes x nx voxels as units of 8 columns x 16 planes (256 bytes, plane-major inside), widened to
// uint16 (uint8 voxels shifted into the high byte): the four units a lane group of zx4_kernel needs for one
// k-step are one contiguous KiB
// wherever the window starts.  Planes past the block and columns past the row read as zero.  Any strides, any
// alignment: the copy is what lifts zx4's 16-byte alignment rules.
template <typename InT>
__global__ void __launch_bounds__(256)
zx6_pack_kernel(const InT* __restrict__ vol, int64_t stride_z, int64_t stride_y, int64_t stride_x,
                const mmx_block* __restrict__ blocks, uint16_t* __restrict__ pack, int64_t pack_stride)
{
    constexpr int PITCH = 512 + 8;                       // uint16 per LDS row (fused paths take px <= 512)
    __shared__ __attribute__((aligned(16))) uint16_t tile[16][PITCH];
    const mmx_block bd = blocks[blockIdx.y];
    const int ntz = (bd.nz + 15) >> 4, nch8 = (bd.nx + 7) >> 3;
    const int yt = blockIdx.x;
    if (yt >= bd.ny * ntz) return;
    const int y = yt / ntz, t = yt - y * ntz;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const InT* src = vol + bd.src_off + (int64_t)y * stride_y;
    // rows that start on 8-byte boundaries (the usual case: block origins on multiples of 4 voxels): 4 voxels per
    // lane and load; the last, partial group of a row and everything else one voxel at a time
    bool quads = false;
    if constexpr (sizeof(InT) == 2)
        quads = stride_x == 1 && ((bd.src_off | stride_y | stride_z) & 3) == 0 && (reinterpret_cast<uintptr_t>(vol) & 7) == 0;
    if (quads) {
        // All of a wave's loads -- four planes x (up to) two 4-voxel groups per lane -- are issued before the first
        // LDS write: a workgroup moves 8 KiB in and 8 KiB out and nothing else hides its load latency (the loop form
        // had one or two loads in flight per wave: 2.8 TB/s).
        u2_4 v[4][2];
        bool have[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int z = 16 * t + 4 * wave + i;
            const InT* row = src + (int64_t)(z < bd.nz ? z : 0) * stride_z;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int x = lane * 4 + 256 * h;
                have[i][h] = x < 8 * nch8;
                v[i][h] = (u2_4){0u, 0u};
                if (have[i][h] && z < bd.nz) {
                    if (x + 3 < bd.nx) {
                        v[i][h] = *reinterpret_cast<const u2_4*>(row + x);
                    } else {
                        unsigned e[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) e[j] = x + j < bd.nx ? (unsigned)row[x + j] : 0u;
                        v[i][h] = (u2_4){e[0] | (e[1] << 16), e[2] | (e[3] << 16)};
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                if (have[i][h]) *reinterpret_cast<u2_4*>(&tile[4 * wave + i][lane * 4 + 256 * h]) = v[i][h];
    } else
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 4 * wave + i, z = 16 * t + r;
        if (z >= bd.nz) {
            for (int x = lane * 4; x < 8 * nch8; x += 256) *reinterpret_cast<u2_4*>(&tile[r][x]) = (u2_4){0u, 0u};
        } else {
            for (int x = lane; x < 8 * nch8; x += 64)
                tile[r][x] = x < bd.nx ? (uint16_t)((unsigned)src[(int64_t)z * stride_z + (int64_t)x * stride_x] << (sizeof(InT) == 1 ? 8 : 0))
                                       : (uint16_t)0;     // (uint8 voxels go to the HIGH byte: the exact high float16 piece carries them)
        }
    }
    __syncthreads();
    u4_4* dst = reinterpret_cast<u4_4*>(pack + (int64_t)bd.slot * pack_stride) + (int64_t)yt * nch8 * 16;
    for (int u = threadIdx.x; u < nch8 * 16; u += 256) {
        const int j = u >> 4, r = u & 15;
        dst[u] = *reinterpret_cast<const u4_4*>(&tile[r][8 * j]);
    }
}

// The same copy for float32 voxels, split into float16 pieces on the way (v = hi + lo / 2048: 22 significant bits, what
// the fragments carry too): per unit the 16 planes' high pieces (256 bytes), then their low pieces.  Valid for
// |v| < 65504 and loses nothing worth having above ~2^-10: the caller vouches for the range (MMX_ZX_FLOAT_RANGE_OK).
__global__ void __launch_bounds__(256)
zx6_pack_f32_kernel(const float* __restrict__ vol, int64_t stride_z, int64_t stride_y, int64_t stride_x,
                    const mmx_block* __restrict__ blocks, uint16_t* __restrict__ pack, int64_t pack_stride)
{
    constexpr int PITCH = 512 + 4;                       // dwords (hi | lo << 16) per LDS row
    __shared__ __attribute__((aligned(16))) unsigned tile[16][PITCH];
    const mmx_block bd = blocks[blockIdx.y];
    const int ntz = (bd.nz + 15) >> 4, nch8 = (bd.nx + 7) >> 3;
    const int yt = blockIdx.x;
    if (yt >= bd.ny * ntz) return;
    const int y = yt / ntz, t = yt - y * ntz;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float* src = vol + bd.src_off + (int64_t)y * stride_y;
    auto pieces = [](float v) __attribute__((always_inline)) {
        const _Float16 h = (_Float16)v;
        const _Float16 l = (_Float16)((v - (float)h) * kLoScale);
        return (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
    };
    // rows that start on 16-byte boundaries (preprocessed slot buffers, most float images): 4 voxels per lane and load
    const bool quads = stride_x == 1 && ((bd.src_off | stride_y | stride_z) & 3) == 0 && (reinterpret_cast<uintptr_t>(vol) & 15) == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 4 * wave + i, z = 16 * t + r;
        if (quads && z < bd.nz) {
            const float* row = src + (int64_t)z * stride_z;
            for (int x = lane * 4; x < 8 * nch8; x += 256) {
                f4_4 v = {0.f, 0.f, 0.f, 0.f};
                if (x + 3 < bd.nx) v = *reinterpret_cast<const f4_4*>(row + x);
                else
                    for (int j = 0; j < 4; ++j) v[j] = x + j < bd.nx ? row[x + j] : 0.f;
                *reinterpret_cast<u4_4*>(&tile[r][x]) = (u4_4){pieces(v[0]), pieces(v[1]), pieces(v[2]), pieces(v[3])};
            }
        } else {
            for (int x = lane; x < 8 * nch8; x += 64) {
                float v = 0.f;
                if (z < bd.nz && x < bd.nx) v = src[(int64_t)z * stride_z + (int64_t)x * stride_x];
                tile[r][x] = pieces(v);
            }
        }
    }
    __syncthreads();
    u4_4* dst = reinterpret_cast<u4_4*>(pack + (int64_t)bd.slot * pack_stride) + (int64_t)yt * nch8 * 32;
    for (int u = threadIdx.x; u < nch8 * 16; u += 256) {
        const int j = u >> 4, r = u & 15;
        const u4_4 a = *reinterpret_cast<const u4_4*>(&tile[r][8 * j]);
        const u4_4 b = *reinterpret_cast<const u4_4*>(&tile[r][8 * j + 4]);
        // dwords (hi | lo << 16) of 8 columns -> 4 dwords of packed high pieces, 4 of packed low pieces
        const u4_4 hi = {__builtin_amdgcn_perm(a[1], a[0], 0x05040100u), __builtin_amdgcn_perm(a[3], a[2], 0x05040100u),
                         __builtin_amdgcn_perm(b[1], b[0], 0x05040100u), __builtin_amdgcn_perm(b[3], b[2], 0x05040100u)};
        const u4_4 lo = {__builtin_amdgcn_perm(a[1], a[0], 0x07060302u), __builtin_amdgcn_perm(a[3], a[2], 0x07060302u),
                         __builtin_amdgcn_perm(b[1], b[0], 0x07060302u), __builtin_amdgcn_perm(b[3], b[2], 0x07060302u)};
        dst[j * 32 + r] = hi;
        dst[j * 32 + 16 + r] = lo;
    }
}

template <int NKX, int LA>
int launch_zx6(const mmx_volume* vol, const mmx_block* d_blocks, const mmx_block* h_blocks, int n_blocks,
               const mmx_zx6_plan& plan, const mmx_taps_f32& tx, int radius, void* d_work, float qp, float qq, hipStream_t s)
{
    using cg = cls4<NKX, LA>;
    mmx_zx4_cfg cfg{};          // (every field the setup kernel reads has a value: qp = qq = 0 means float32 tiles)
    for (int k = 0; k <= MMX_MAX_RADIUS_FAST; ++k) { cfg.w0[k] = tx.w0[k]; cfg.w2[k] = tx.w2[k]; }
    cfg.radius = radius;
    // the pieces carry v / 2^16 of the widened voxel: skimage's img_as_float scale on top
    cfg.xscale = vol->dtype == MMX_U16 ? (float)(65536.0 / 65535.0) :
                 (vol->dtype == MMX_U8 ? (float)(65536.0 / (255.0 * 256.0)) : 1.f);      // (float voxels: as they are)
    cfg.ncw = cfg.ncz = 0;
    cfg.staged = 2;
    cfg.qp = qp; cfg.qq = qq;
    cfg.maxcol = cfg.maxu = 0;
    // two column tiles per wave (zx4_kernel's NTW): 16-bit tiles of integer voxels, 8 < radius <= 16; tile rows past
    // the block are not stored (the A/B runs of both: profiles/r04_zx_experiments.txt)
    const bool pair = NKX == 2 && LA == 1 && qp > 0.f && vol->dtype != MMX_F32;
    cfg.ntw = pair ? 2 : 1;
    int max_waves = 0;
    for (int i = 0; i < n_blocks; ++i) {
        const mmx_block& b = h_blocks[i];
        if (b.nx < radius || b.nz < radius) return MMX_ERR_UNSUPPORTED;       // single reflection
        int j;
        for (j = 0; j < cfg.ncw && cfg.wcls[j] != b.nx; ++j) {}
        if (j == cfg.ncw) { if (j == MMX_ZX4_MAXCLS) return MMX_ERR_UNSUPPORTED; cfg.wcls[cfg.ncw++] = b.nx; }
        for (j = 0; j < cfg.ncz && cfg.zcls[j] != b.nz; ++j) {}
        if (j == cfg.ncz) { if (j == MMX_ZX4_MAXCLS) return MMX_ERR_UNSUPPORTED; cfg.zcls[cfg.ncz++] = b.nz; }
        const int ntx = (b.nx + 15) / 16, ntz = (b.nz + 15) / 16;
        if (ntx > cfg.maxcol) cfg.maxcol = ntx;
        if (ntz > cfg.maxu) cfg.maxu = ntz;
        const int per_row = pair ? (ntx + 1) / 2 : ntx;
        if (b.ny * per_row > max_waves) max_waves = b.ny * per_row;
    }
    for (int j = cfg.ncw; j < MMX_ZX4_MAXCLS; ++j) cfg.wcls[j] = -1;
    for (int j = cfg.ncz; j < MMX_ZX4_MAXCLS; ++j) cfg.zcls[j] = -1;
    const int nx_entries = cfg.ncw * cfg.maxcol * NKX * 2;
    const int nz_entries = cfg.ncz * cfg.maxu * cg::NKZ * 2;
    const size_t xbytes = (size_t)nx_entries * 2 * 64 * sizeof(u4_4);
    const size_t zbytes = (size_t)nz_entries * 2 * 64 * sizeof(u4_4);
    if ((int64_t)(xbytes + zbytes) > plan.tab_bytes) return MMX_ERR_UNSUPPORTED;
    char* w = reinterpret_cast<char*>(d_work);
    u4_4* xtab = reinterpret_cast<u4_4*>(w + plan.tab_off);
    u4_4* ztab = reinterpret_cast<u4_4*>(w + plan.tab_off + xbytes);
    hipLaunchKernelGGL((zx4_setup<NKX, LA>), dim3((nx_entries + nz_entries + 3) / 4), dim3(256), 0, s, cfg, xtab, ztab);
    dim3 grid((((max_waves + 3) / 4) + 7) & ~7, n_blocks);        // (a multiple of 8: the XCD-aware order in the kernel)
    if (vol->dtype == MMX_F32) {
        if (qp > 0.f)          // (16-bit tiles: the caller's bounds cover the voxels' range, mmx_volume.value_range)
            hipLaunchKernelGGL((zx4_kernel<NKX, LA, float, true>), grid, dim3(256), 0, s,
                               reinterpret_cast<const float*>(w + plan.pack_off), plan.pack_stride / 2, (int64_t)0, d_blocks,
                               plan.tile_stride, reinterpret_cast<float*>(w), reinterpret_cast<float*>(w + plan.q_off),
                               xtab, ztab, cfg);
        else
            hipLaunchKernelGGL((zx4_kernel<NKX, LA, float, false>), grid, dim3(256), 0, s,
                               reinterpret_cast<const float*>(w + plan.pack_off), plan.pack_stride / 2, (int64_t)0, d_blocks,
                               plan.tile_stride, reinterpret_cast<float*>(w), reinterpret_cast<float*>(w + plan.q_off),
                               xtab, ztab, cfg);
    } else if (pair) {
        if constexpr (NKX == 2 && LA == 1)
            hipLaunchKernelGGL((zx4_kernel<NKX, LA, uint16_t, true, 2>), grid, dim3(256), 0, s,
                               reinterpret_cast<const uint16_t*>(w + plan.pack_off), plan.pack_stride, (int64_t)0, d_blocks,
                               plan.tile_stride, reinterpret_cast<float*>(w), reinterpret_cast<float*>(w + plan.q_off),
                               xtab, ztab, cfg);
    } else if (qp > 0.f)
        hipLaunchKernelGGL((zx4_kernel<NKX, LA, uint16_t, true>), grid, dim3(256), 0, s,
                           reinterpret_cast<const uint16_t*>(w + plan.pack_off), plan.pack_stride, (int64_t)0, d_blocks,
                           plan.tile_stride, reinterpret_cast<float*>(w), reinterpret_cast<float*>(w + plan.q_off),
                           xtab, ztab, cfg);
    else
        hipLaunchKernelGGL((zx4_kernel<NKX, LA, uint16_t, false>), grid, dim3(256), 0, s,
                           reinterpret_cast<const uint16_t*>(w + plan.pack_off), plan.pack_stride, (int64_t)0, d_blocks,
                           plan.tile_stride, reinterpret_cast<float*>(w), reinterpret_cast<float*>(w + plan.q_off),
                           xtab, ztab, cfg);
    return hipGetLastError() == hipSuccess ? MMX_OK : MMX_ERR_HIP;
}

}  // namespace

int mmx_zx6_plan_make(const mmx_block* h_blocks, int n_blocks, int64_t slot_elems, int voxel_dtype, mmx_zx6_plan* plan)
{
    const int pieces = voxel_dtype == MMX_F32 ? 2 : 1;        // float voxels: two float16 pieces per voxel in the copy
    int64_t tile = 0, pk = 0;
    int max_tiles = 0, max_rowtiles = 0, maxcol = 0, maxu = 0;
    int wcls[MMX_ZX4_MAXCLS], zcls[MMX_ZX4_MAXCLS], ncw = 0, ncz = 0;
    for (int i = 0; i < n_blocks; ++i) {
        const mmx_block& b = h_blocks[i];
        if (b.px > 512) return MMX_ERR_UNSUPPORTED;
        int j;
        for (j = 0; j < ncw && wcls[j] != b.nx; ++j) {}
        if (j == ncw) { if (j == MMX_ZX4_MAXCLS) return MMX_ERR_UNSUPPORTED; wcls[ncw++] = b.nx; }
        for (j = 0; j < ncz && zcls[j] != b.nz; ++j) {}
        if (j == ncz) { if (j == MMX_ZX4_MAXCLS) return MMX_ERR_UNSUPPORTED; zcls[ncz++] = b.nz; }
        const int ntx = (b.nx + 15) / 16, ntz = (b.nz + 15) / 16, nch8 = (b.nx + 7) / 8;
        const int64_t te = (int64_t)ntx * ntz * b.ny * 256, pe = (int64_t)b.ny * ntz * nch8 * 128 * pieces;
        if (te > tile) tile = te;
        if (pe > pk) pk = pe;
        if (ntx * ntz > max_tiles) max_tiles = ntx * ntz;
        if (b.ny * ntz > max_rowtiles) max_rowtiles = b.ny * ntz;
        if (ntx > maxcol) maxcol = ntx;
        if (ntz > maxu) maxu = ntz;
    }
    if (tile * 4 >= (int64_t(1) << 31) || pk * 2 >= (int64_t(1) << 31)) return MMX_ERR_UNSUPPORTED;   // 32-bit offsets in a block
    plan->tile_stride = tile;
    plan->pack_stride = pk;
    plan->q_off = (int64_t)n_blocks * tile * 4;
    plan->tab_off = 2 * plan->q_off;
    // the largest tables any radius class needs: widths x columns x 2 k-steps, depths x z tiles x 3, two kernels each
    plan->tab_bytes = (int64_t)(ncw * maxcol * 2 * 2 + ncz * maxu * 3 * 2) * 2 * 64 * 16;
    plan->pack_off = (plan->tab_off + plan->tab_bytes + 255) & ~int64_t(255);
    plan->max_tiles = max_tiles;
    plan->max_rowtiles = max_rowtiles;
    const int64_t need = plan->pack_off + (int64_t)n_blocks * pk * 2;
    return need <= 4 * (int64_t)n_blocks * slot_elems * 4 ? MMX_OK : MMX_ERR_UNSUPPORTED;
}

int mmx_launch_zx6_pack(const mmx_volume* vol, const mmx_block* d_blocks, const mmx_block* h_blocks, int n_blocks,
                        const mmx_zx6_plan& plan, void* d_work, hipStream_t stream)
{
    (void)h_blocks;
    if (vol->dtype != MMX_U16 && vol->dtype != MMX_U8 && vol->dtype != MMX_F32) return MMX_ERR_UNSUPPORTED;
    uint16_t* pack = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(d_work) + plan.pack_off);
    dim3 grid(plan.max_rowtiles, n_blocks);
    if (vol->dtype == MMX_F32)
        hipLaunchKernelGGL(zx6_pack_f32_kernel, grid, dim3(256), 0, stream, (const float*)vol->d_data,
                           vol->stride_z, vol->stride_y, vol->stride_x, d_blocks, pack, plan.pack_stride);
    else if (vol->dtype == MMX_U16)
        hipLaunchKernelGGL((zx6_pack_kernel<uint16_t>), grid, dim3(256), 0, stream, (const uint16_t*)vol->d_data,
                           vol->stride_z, vol->stride_y, vol->stride_x, d_blocks, pack, plan.pack_stride);
    else
        hipLaunchKernelGGL((zx6_pack_kernel<uint8_t>), grid, dim3(256), 0, stream, (const uint8_t*)vol->d_data,
                           vol->stride_z, vol->stride_y, vol->stride_x, d_blocks, pack, plan.pack_stride);
    return hipGetLastError() == hipSuccess ? MMX_OK : MMX_ERR_HIP;
}

// qp, qq > 0: Q16 tiles (P qp in [0, 1], Q qq in [-1, 1]); 0: float32 tiles
int mmx_launch_zx6(const mmx_volume* vol, const mmx_block* d_blocks, const mmx_block* h_blocks, int n_blocks,
                   const mmx_zx6_plan& plan, const mmx_taps_f32& tx, int radius, void* d_work, float qp, float qq,
                   hipStream_t stream)
{
    if (vol->dtype != MMX_U16 && vol->dtype != MMX_U8 && vol->dtype != MMX_F32) return MMX_ERR_UNSUPPORTED;
    if (radius < 1 || radius > MMX_MAX_RADIUS_FAST) return MMX_ERR_UNSUPPORTED;
    if (radius <= 8) return launch_zx6<1, 1>(vol, d_blocks, h_blocks, n_blocks, plan, tx, radius, d_work, qp, qq, stream);
    if (radius <= 16) return launch_zx6<2, 1>(vol, d_blocks, h_blocks, n_blocks, plan, tx, radius, d_work, qp, qq, stream);
    return launch_zx6<2, 2>(vol, d_blocks, h_blocks, n_blocks, plan, tx, radius, d_work, qp, qq, stream);
}

