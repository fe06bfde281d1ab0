// Device helpers shared by the preprocessing kernels (mmx_preproc.hip: one workgroup per tile with the whole
// life of the tile in one kernel; mmx_preproc_pipe.hip: statistics kernel + pipelined blur kernel).
#pragma once

#include <algorithm>

#include "mmx_common.h"

#define PP_R 32          // int(4 * 8 + 0.5): sigma 8 is hard-coded in plot_3d.py:151
#define PP_MAXL 32       // longest line of the register-resident pass
#define PP_WG 640        // 10 waves: 625 lines of a 25^3 sub-block in one round
#define PP_WG_GENERIC 1024
#define PP_HIST (5 * 256)
#define PP_NB_LOAD 13     // global loads in flight per lane (25 rows = 13 + 12)
#define PP_NB 9           // voxels in flight per lane in the arithmetic stages (25 rows = 9 + 9 + 7)

struct pp_args {
    double clip_min, clip_max, max_thresh, strength, ero_thr;
    double tv_weight, tv_factor;          // total-variation denoising: weight (0 = off), tau / weight
    int64_t dst_sy, dst_sz;
    int32_t do_unsharp, do_erosion, rgb_guess, _pad;
};
#define PP_TV_SCRATCH 7      // doubles of scratch per voxel with total-variation denoising on (2 without)
#define PP_MAXLEAF 2048      // leaves of NumPy's pairwise summation tree kept in LDS (n <= ~130 000 voxels)

namespace {

__device__ __forceinline__ double pp_clip(double x, double lo, double hi)
{
    // np.clip == minimum(maximum(x, lo), hi); no NaNs on this path, so v_max_f64 / v_min_f64 agree
    return fmin(fmax(x, lo), hi);
}

// numpy _lerp (numpy/lib/_function_base_impl.py): a + (b-a)*t, or b - (b-a)*(1-t) for t >= 0.5
__device__ __forceinline__ double pp_lerp(double a, double b, double t)      // (float64 images: the same NumPy formula)
{
    const double d = b - a;
    double r = a + d * t;
    if (t >= 0.5) r = b - d * (1.0 - t);
    return r;
}
__device__ __forceinline__ double pp_lerp(int a, int b, double t)
{
    const double d = (double)(b - a);
    double r = (double)a + d * t;
    if (t >= 0.5) r = (double)b - d * (1.0 - t);
    return r;
}

// saturate_roi's stretch (clip(x, vmin, vmax) - vmin) / span.  The division is IEEE-exact: it is the
// compiler's own float64 expansion (v_div_scale / v_rcp + 2 Newton steps / q = n*r; e = fma(-d, q, n);
// v_div_fmas; v_div_fixup) with the denominator-only part hoisted out of the voxel loop.  That is valid
// while v_div_scale leaves both operands unscaled, i.e. for ordinary magnitudes: `fast` is set only
// when 2^-200 < span < 2^200, and the numerator is 0 or in [2^-60, span] (a difference of a <= 16-bit
// integer and an interpolated one).  Otherwise the plain `/` is used.
// An *identity* tile (vmin == vmax: the reference leaves the voxels alone) runs the same formula with
// vmin = 0, vmax = +inf, span = 1, which returns x exactly.
struct pp_sat {
    double vmin, vmax, span, rcp;
    int identity, fast;
    __device__ __forceinline__ void finish()
    {
        if (identity) { vmin = 0.; vmax = __builtin_inf(); span = 1.; }
        fast = span > 0x1p-200 && span < 0x1p200;
        double r = __builtin_amdgcn_rcp(span);
        r = __builtin_fma(__builtin_fma(-span, r, 1.0), r, r);
        r = __builtin_fma(__builtin_fma(-span, r, 1.0), r, r);
        rcp = r;
    }
    __device__ __forceinline__ double operator()(double raw) const      // requires `fast`
    {
        const double num = pp_clip(raw, vmin, vmax) - vmin;
        const double q = num * rcp;
        const double e = __builtin_fma(-span, q, num);
        return __builtin_fma(e, rcp, q);
    }
    __device__ __forceinline__ double plain(double raw) const            // any magnitude
    {
        return (pp_clip(raw, vmin, vmax) - vmin) / span;
    }
};

// histogram increment with the lanes that share the first active lane's bin folded into one LDS
// atomic: a tile is mostly background, whose voxels all land in one or two bins
__device__ __forceinline__ void pp_hist_add(uint32_t* hist, int bin, bool active)
{
    const unsigned long long act = __ballot(active);
    if (!act) return;
    const int leader = __ffsll((long long)act) - 1;
    const int lead_bin = __shfl(bin, leader);
    const unsigned long long same = __ballot(active && bin == lead_bin);
    if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[lead_bin], (uint32_t)__popcll(same));
    if (active && bin != lead_bin) atomicAdd(&hist[bin], 1u);
}

// Executed by one full wave: the bin of `h[0..255]` holding 0-based rank `rank`, and the rank
// inside that bin.
__device__ __forceinline__ void pp_select(const uint32_t* h, uint32_t rank, int& bin, uint32_t& res)
{
    const int lane = threadIdx.x & 63;
    const uint32_t c0 = h[4 * lane], c1 = h[4 * lane + 1], c2 = h[4 * lane + 2], c3 = h[4 * lane + 3];
    const uint32_t s = c0 + c1 + c2 + c3;
    uint32_t incl = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    const uint32_t excl = incl - s;
    const bool mine = rank >= excl && rank < incl;
    const unsigned long long m = __ballot(mine);
    const int src = m ? __ffsll((long long)m) - 1 : 63;
    uint32_t r = rank - excl;
    int b = 4 * lane;
    if (r >= c0) { r -= c0; ++b; if (r >= c1) { r -= c1; ++b; if (r >= c2) { r -= c2; ++b; } } }
    bin = __shfl(b, src);
    res = __shfl(r, src);
}

__device__ __forceinline__ double pp_wave_sum(double v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d);
    return v;
}

// NumPy's pairwise summation of val(0..n-1) (DOUBLE_pairwise_sum), by ONE lane.  `stk` is
// 4 x 40 ints/doubles of LDS for the explicit recursion stack.
template <typename F>
__device__ double pp_pairwise_leaf(F val, int lo, int n)
{
    if (n < 8) {
        double res = 0.;
        for (int i = 0; i < n; ++i) res += val(lo + i);
        return res;
    }
    double r0 = val(lo), r1 = val(lo + 1), r2 = val(lo + 2), r3 = val(lo + 3);
    double r4 = val(lo + 4), r5 = val(lo + 5), r6 = val(lo + 6), r7 = val(lo + 7);
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
        r0 += val(lo + i); r1 += val(lo + i + 1); r2 += val(lo + i + 2); r3 += val(lo + i + 3);
        r4 += val(lo + i + 4); r5 += val(lo + i + 5); r6 += val(lo + i + 6); r7 += val(lo + i + 7);
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res += val(lo + i);
    return res;
}

struct pp_stack { int lo[40]; int n[40]; int st[40]; double acc[40]; };

template <typename F>
__device__ double pp_pairwise(F val, int n, pp_stack* S)
{
    int sp = 0;
    double ret = 0.;
    S->lo[0] = 0; S->n[0] = n; S->st[0] = 0; sp = 1;
    while (sp > 0) {
        const int t = sp - 1;
        const int lo = S->lo[t], m = S->n[t];
        if (m <= 128) { ret = pp_pairwise_leaf(val, lo, m); --sp; continue; }
        int n2 = m / 2; n2 -= n2 % 8;
        if (S->st[t] == 0) { S->st[t] = 1; S->lo[sp] = lo; S->n[sp] = n2; S->st[sp] = 0; ++sp; }
        else if (S->st[t] == 1) {
            S->acc[t] = ret; S->st[t] = 2;
            S->lo[sp] = lo + n2; S->n[sp] = m - n2; S->st[sp] = 0; ++sp;
        } else { ret = S->acc[t] + ret; --sp; }
    }
    return 0. + ret;   // the reduction starts from the identity (add.reduce)
}

// NumPy's pairwise summation tree evaluated in parallel: the leaves (runs of <= 128 elements, fixed by n alone)
// are enumerated once per tile, every lane sums whole leaves (pp_pairwise_leaf, NumPy's 8 accumulators), one
// lane folds the leaf sums in the tree's order.  Bit-equal to ndarray.sum() of a contiguous float64 array.
struct pp_leaves { int lo[PP_MAXLEAF]; int n[PP_MAXLEAF]; double sum[2][PP_MAXLEAF]; int count; };

__device__ void pp_enum_leaves(int n, pp_stack* S, pp_leaves* Lv)       // one lane
{
    int sp = 0, cnt = 0;
    S->lo[0] = 0; S->n[0] = n; sp = 1;
    while (sp > 0) {
        --sp;
        const int lo = S->lo[sp], m = S->n[sp];
        if (m <= 128) {
            if (cnt < PP_MAXLEAF) { Lv->lo[cnt] = lo; Lv->n[cnt] = m; }
            ++cnt;
            continue;
        }
        int n2 = m / 2; n2 -= n2 % 8;
        S->lo[sp] = lo + n2; S->n[sp] = m - n2; ++sp;       // right half: popped after the left one
        S->lo[sp] = lo; S->n[sp] = n2; ++sp;
    }
    Lv->count = cnt;
}

__device__ double pp_fold_leaves(int n, pp_stack* S, const pp_leaves* Lv, int which)     // one lane
{
    int sp = 0, next = 0;
    double ret = 0.;
    S->lo[0] = 0; S->n[0] = n; S->st[0] = 0; sp = 1;
    while (sp > 0) {
        const int t = sp - 1;
        const int lo = S->lo[t], m = S->n[t];
        if (m <= 128) { ret = Lv->sum[which][next++]; --sp; continue; }
        int n2 = m / 2; n2 -= n2 % 8;
        if (S->st[t] == 0) { S->st[t] = 1; S->lo[sp] = lo; S->n[sp] = n2; S->st[sp] = 0; ++sp; }
        else if (S->st[t] == 1) {
            S->acc[t] = ret; S->st[t] = 2;
            S->lo[sp] = lo + n2; S->n[sp] = m - n2; S->st[sp] = 0; ++sp;
        } else { ret = S->acc[t] + ret; --sp; }
    }
    return 0. + ret;
}

// One in-place Gaussian pass over the lines of one axis, lines held in registers.
// base/stride address the LDS tile; L <= PP_MAXL.
// The weights come through a `const __restrict__` kernel argument (scalar loads next to their use),
// not by value: 66 SGPRs held from kernel entry made the compiler spill scalars in every other stage.
template <int MAXL = PP_MAXL>
__device__ __forceinline__ void pp_line_pass(double* __restrict__ tile, int64_t base, int64_t stride, int L,
                                             const double (&w)[PP_R + 1])
{
    double r[MAXL];
#pragma unroll
    for (int i = 0; i < MAXL; ++i) r[i] = tile[base + (i < L ? i : L - 1) * stride];
#pragma unroll
    for (int i = 0; i < MAXL; ++i) {
        if (i < L) {
            double acc = r[i] * w[0];
#pragma unroll
            for (int k = PP_R; k >= 1; --k) {
                const int a = i - k < 0 ? 0 : i - k;
                const int b = i + k > MAXL - 1 ? MAXL - 1 : i + k;
                acc += (r[a] + r[b]) * w[k];
            }
            tile[base + i * stride] = acc;
        }
    }
}

// (z, y, x) of the dense index i of an n-voxel tile: exact for i < 2^15 (float has 24 bits and
// (i + 0.5) / nx is at least 0.5 / 32 away from an integer)
struct pp_coord { int t, x, y, z; };
__device__ __forceinline__ pp_coord pp_decode(int i, int nx, int ny, float inv_nx, float inv_ny)
{
    pp_coord c;
    c.t = (int)(((float)i + 0.5f) * inv_nx);
    c.x = i - c.t * nx;
    c.z = (int)(((float)c.t + 0.5f) * inv_ny);
    c.y = c.t - c.z * ny;
    return c;
}

inline pp_args pp_make_args(const mmx_preproc_params* p, int64_t dst_sy, int64_t dst_sz)
{
    pp_args A;
    A.clip_min = p->clip_min; A.clip_max = p->clip_max; A.max_thresh = p->max_thresh;
    A.strength = p->unsharp_strength; A.ero_thr = p->erosion_threshold;
    A.tv_weight = p->tv_weight; A.tv_factor = p->tv_factor;
    A.dst_sy = dst_sy; A.dst_sz = dst_sz;
    A.do_unsharp = p->unsharp_strength != 0.0;          // Python truthiness: `if unsharp_strength:`
    A.do_erosion = p->erosion_threshold != 0.0;         // `if thresh_eros and ...`
    A.rgb_guess = p->rgb_guess;
    A._pad = 0;
    return A;
}

}  // namespace

// mmx_preproc_pipe.hip
int64_t mmx_pp_pipe_work_bytes(const mmx_subblock* h_subs, int n_subs);
int mmx_launch_pp_pipe(const mmx_volume* vol, const mmx_subblock* d_subs, const mmx_subblock* h_subs, int n_subs,
                       const mmx_quantile_class* d_qclasses, const double* d_weights, const pp_args& A,
                       float* d_out32, double* d_out64, mmx_subblock_info* d_info, void* d_work,
                       int tiles_per_wg, hipStream_t s);
