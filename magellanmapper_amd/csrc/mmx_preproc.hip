// Per-block preprocessing ahead of blob detection: contrast stretch + unsharp mask + erosion.
//
// What the reference does to every denoise sub-block of every detection block
// (magmap/cv/stack_detect.py:122-150) before `blob_log` sees it:
//   saturate_roi (magmap/plot/plot_3d.py:55-112)
//       vmin, vmax = np.percentile(roi, (clip_vmin, clip_vmax))          # linear interpolation
//       vmax = max(vmax, near_max * max_thresh_factor);  roi unchanged when vmin == vmax
//       sat  = (np.clip(roi, vmin, vmax) - vmin) / (vmax - vmin)
//   denoise_roi  (plot_3d.py:115-172)
//       mean = np.mean(sat);  den = np.clip(sat, clip_min, clip_max)
//       den  = den + (den - unsharp_strength * gaussian(den, sigma 8, 'nearest', truncate 4))
//       den  = grey_erosion(den, octahedron(1))  if  mean > erosion_threshold
// all in float64.  The result feeds an exact float64 re-score (mmx_rescore.hip), so it is
// reproduced BIT FOR BIT: the same IEEE operations in the same order as NumPy / SciPy
// (NI_Correlate1D symmetric branch: acc = in[c]*w0; for k = R..1: acc += (in[c-k]+in[c+k])*w[k]),
// no FMA contraction (this file is built with -ffp-contract=off), correctly rounded division.
// oracle/preprocess_oracle.py states the same on the CPU and is pinned to the real reference.
//
// Design (gfx950).  A stock sub-block is 25^3 voxels = 122 KiB of float64: it lives in LDS for
// its whole life, ONE workgroup per sub-block (a CU's 160 KiB holds exactly one), HBM is
// touched once for the 2-byte voxels and once for the 12 bytes of output per voxel.
//   1. voxels -> LDS as doubles + 256-bin histogram of the high byte (LDS atomics)
//   2. the 2x2 order statistics np.percentile needs: two-level radix select (wave scan over
//      the bins), then NumPy's _lerp in double
//   3. saturate, block-sum for the mean, clip -> LDS.  The mean only gates the erosion; it is
//      summed in parallel and, if that lands within 1e-9 of the threshold, re-summed by one
//      lane in NumPy's pairwise order (numpy/_core/src/umath/loops_utils.h.src) -- bit exact
//   4. three in-place Gaussian passes: a lane owns a whole line (<= 32 voxels) in registers,
//      'nearest' extension is the last register replicated, so all tap indices are
//      compile-time; the 33 weights sit in SGPRs (kernarg).  ~97 float64 ops per voxel and
//      axis: this stage is float64-ALU bound (about 30 us per sub-block and CU)
//   5. unsharp in place, then the 7-point minimum (reflect == skip outside) if eroding,
//      float64 + float32 copies out (the float32 copy feeds the LoG passes)
// Sub-blocks with a side above 32 or more voxels than LDS holds take pp_generic_kernel:
// same arithmetic, data in a global scratch, one output per lane and pass.


#include "mmx_pp_common.h"

namespace {

template <typename InT, int WG>
__global__ void __launch_bounds__(WG)
pp_fast_kernel(const InT* __restrict__ vol, int64_t sz, int64_t sy, int64_t sx,
               const mmx_subblock* __restrict__ subs, int n_subs,
               const mmx_quantile_class* __restrict__ qcs, const double* __restrict__ wts,
               pp_args A, float* __restrict__ out32, double* __restrict__ out64,
               mmx_subblock_info* __restrict__ info)
{
    extern __shared__ double tile[];
    __shared__ int s_bin[4];
    __shared__ uint32_t s_res[4];
    __shared__ int s_val[4];
    __shared__ double s_red[WG / 64];
    __shared__ double s_mean;
    __shared__ int s_flags;
    __shared__ pp_stack s_stack;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Workgroups are dealt to the 8 XCDs round-robin; give each XCD a CONTIGUOUS run of tiles so
    // that x-neighbouring tiles (which share 128-byte lines of the source rows and of both
    // outputs) meet in one L2.
    const int per = (n_subs + 7) >> 3;
    const int sub_id = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (sub_id >= n_subs) return;
    const mmx_subblock sb = subs[sub_id];
    const int nz = sb.nz, ny = sb.ny, nx = sb.nx, n = nz * ny * nx;
    const int px = nx | 1;                        // odd row pitch: no LDS bank conflicts on x lines
    uint32_t* hist = (uint32_t*)(tile + nz * ny * px);
    uint16_t* raw = (uint16_t*)(hist + PP_HIST);  // the voxels themselves, dense index
    const InT* src = vol + sb.src_off;
    const float inv_nx = 1.0f / (float)nx, inv_ny = 1.0f / (float)ny;
#define PP_STAMP() do {} while (0)
    PP_STAMP();

    for (int i = tid; i < PP_HIST; i += WG) hist[i] = 0;
    __syncthreads();

    // Lane <-> voxel mapping of the element-wise stages: a lane keeps its x and walks rows
    // (z*ny + y) in steps of `rpi`, so no per-voxel index decoding; rpi * nx <= WG lanes work.
    const int nrows = nz * ny;
    const int rpi = WG / nx;
    const int r_first = (int)(((float)tid + 0.5f) * inv_nx);
    const int x = tid - r_first * nx;
    const int r0 = r_first < rpi ? r_first : nrows;          // idle lanes start past the end

    // 1. voxels -> LDS, histogram of the high byte.  Loads are issued PP_NB at a time: with one
    //    workgroup per CU nothing else hides their latency.
    for (int rb0 = 0; rb0 < nrows; rb0 += PP_NB_LOAD * rpi) {     // same trip count in every lane (ballots)
        const int rb = r0 + rb0;
        int v[PP_NB_LOAD];
#pragma unroll
        for (int j = 0; j < PP_NB_LOAD; ++j) {
            const int row = rb + j * rpi < nrows ? rb + j * rpi : 0;
            const int z = (int)(((float)row + 0.5f) * inv_ny);
            const int y = row - z * ny;
            v[j] = (int)src[z * sz + y * sy + (r0 < nrows ? x : 0) * sx];
        }
#pragma unroll
        for (int j = 0; j < PP_NB_LOAD; ++j) {
            const int row = rb + j * rpi;
            if (row < nrows) raw[row * nx + x] = (uint16_t)v[j];
            pp_hist_add(hist, v[j] >> 8, row < nrows);
        }
    }
    __syncthreads();
    PP_STAMP();

    // 2. order statistics: ranks lo_prev, lo_next, hi_prev, hi_next
    const mmx_quantile_class qc = qcs[sb.qclass];
    if (wave < 4) {
        const uint32_t rank = wave == 0 ? qc.lo_prev : wave == 1 ? qc.lo_next : wave == 2 ? qc.hi_prev : qc.hi_next;
        int b; uint32_t r;
        pp_select(hist, rank, b, r);
        if (lane == 0) { s_bin[wave] = b; s_res[wave] = r; }
    }
    __syncthreads();
    {
        // one low-byte histogram per DISTINCT high-byte bin (prev / next ranks usually share theirs)
        const int b0 = s_bin[0], b1 = s_bin[1], b2 = s_bin[2], b3 = s_bin[3];
        const bool u1 = b1 != b0, u2 = b2 != b0 && b2 != b1, u3 = b3 != b0 && b3 != b1 && b3 != b2;
        for (int i = tid; i < n; i += WG) {
            const int v = raw[i];
            const int hi = v >> 8, lo = v & 255;
            if (hi == b0) atomicAdd(&hist[256 + lo], 1u);
            if (u1 && hi == b1) atomicAdd(&hist[512 + lo], 1u);
            if (u2 && hi == b2) atomicAdd(&hist[768 + lo], 1u);
            if (u3 && hi == b3) atomicAdd(&hist[1024 + lo], 1u);
        }
    }
    __syncthreads();
    if (wave < 4) {
        int slot = wave;
        for (int w = wave - 1; w >= 0; --w) if (s_bin[w] == s_bin[wave]) slot = w;
        int b; uint32_t r;
        pp_select(hist + 256 * (1 + slot), s_res[wave], b, r);
        if (lane == 0) s_val[wave] = (s_bin[wave] << 8) | b;
    }
    __syncthreads();
    PP_STAMP();

    pp_sat S;
    double info_vmin, info_vmax;
    {
        const double vmin = pp_lerp(s_val[0], s_val[1], qc.lo_gamma);
        double vmax = pp_lerp(s_val[2], s_val[3], qc.hi_gamma);
        S.identity = vmin == vmax;
        if (vmax < A.max_thresh) vmax = A.max_thresh;
        S.vmin = vmin; S.vmax = vmax; S.span = vmax - vmin;
        info_vmin = vmin; info_vmax = vmax;
        S.finish();
    }

    // 3. saturate, sum, clip -> the float64 tile (PP_NB independent divisions in flight per lane)
    double part = 0.;
    for (int rb = r0; rb < nrows; rb += PP_NB * rpi) {
        double s[PP_NB];
#pragma unroll
        for (int j = 0; j < PP_NB; ++j) {
            const int row = rb + j * rpi < nrows ? rb + j * rpi : rb;
            s[j] = S((double)(int)raw[row * nx + x]);
        }
#pragma unroll
        for (int j = 0; j < PP_NB; ++j) {
            const int row = rb + j * rpi;
            if (row < nrows) {
                part += s[j];
                tile[row * px + x] = pp_clip(s[j], A.clip_min, A.clip_max);
            }
        }
    }
    if (!S.fast) {                 // extreme span (never with integer voxels): plain division, same stores
        part = 0.;
        for (int row = r0; row < nrows; row += rpi) {
            const double sv = S.plain((double)(int)raw[row * nx + x]);
            part += sv;
            tile[row * px + x] = pp_clip(sv, A.clip_min, A.clip_max);
        }
    }
    part = pp_wave_sum(part);
    if (lane == 0) s_red[wave] = part;
    __syncthreads();
    if (tid == 0) {
        double tot = 0.;
        for (int w = 0; w < WG / 64; ++w) tot += s_red[w];
        double mean = tot / (double)n;
        int flags = S.identity ? MMX_PP_IDENTITY : 0;
        if (A.do_erosion) {
            const double tol = 1e-9 * (fabs(A.ero_thr) > 1. ? fabs(A.ero_thr) : 1.);
            if (!S.identity && fabs(mean - A.ero_thr) <= tol) {
                // knife edge: NumPy's own summation order decides
                auto val = [&](int i) { return S.plain((double)(int)raw[i]); };
                mean = pp_pairwise(val, n, &s_stack) / (double)n;
                flags |= MMX_PP_EXACT_MEAN;
            }
            if (mean > A.ero_thr) flags |= MMX_PP_ERODED;
        }
        s_mean = mean;
        s_flags = flags;
    }
    __syncthreads();
    const int flags = s_flags;
    if (info && tid == 0) {
        mmx_subblock_info o;
        o.vmin = info_vmin; o.vmax = info_vmax; o.mean = s_mean; o.flags = flags; o._pad = 0;
        info[sub_id] = o;
    }
    PP_STAMP();

    // 4. Gaussian blur, axis 0, 1, 2 in place (scipy gaussian_filter order)
    if (A.do_unsharp) {
        double wl[PP_R + 1];
#pragma unroll
        for (int k = 0; k <= PP_R; ++k) wl[k] = wts[k];          // uniform: scalar loads, SGPR resident
#pragma unroll 1
        for (int axis = 0; axis < 3; ++axis) {
            const int L = axis == 0 ? nz : axis == 1 ? ny : nx;
            if (axis == 2 && A.rgb_guess && nx == 3) break;
            const int nlines = n / L;
            for (int l = tid; l < nlines; l += WG) {
                int base, stride;
                if (axis == 2) { base = l * px; stride = 1; }
                else {
                    const int t = (int)(((float)l + 0.5f) * inv_nx);
                    const int x = l - t * nx;
                    if (axis == 0) { base = t * px + x; stride = ny * px; }
                    else { base = t * ny * px + x; stride = px; }
                }
                pp_line_pass(tile, base, stride, L, wl);
            }
            __syncthreads();
        }
    }
    PP_STAMP();

    // 5. unsharp mask (+ erosion), write out
    const bool erode = flags & MMX_PP_ERODED;
    for (int rb = r0; rb < nrows; rb += PP_NB * rpi) {
        double o[PP_NB];
#pragma unroll
        for (int j = 0; j < PP_NB; ++j) {
            const int row = rb + j * rpi < nrows ? rb + j * rpi : rb;
            const double blur = tile[row * px + x];
            if (A.do_unsharp) {
                const double den = pp_clip(S((double)(int)raw[row * nx + x]), A.clip_min, A.clip_max);
                const double m = A.strength * blur;
                const double hp = den - m;
                o[j] = den + hp;
            } else {
                o[j] = blur;
            }
        }
        if (A.do_unsharp && !S.fast) {
            for (int j = 0; j < PP_NB; ++j) {
                const int row = rb + j * rpi < nrows ? rb + j * rpi : rb;
                const double den = pp_clip(S.plain((double)(int)raw[row * nx + x]), A.clip_min, A.clip_max);
                const double m = A.strength * tile[row * px + x];
                const double hp = den - m;
                o[j] = den + hp;
            }
        }
#pragma unroll
        for (int j = 0; j < PP_NB; ++j) {
            const int row = rb + j * rpi;
            if (row >= nrows) continue;
            if (erode) tile[row * px + x] = o[j];
            else {
                const int z = (int)(((float)row + 0.5f) * inv_ny);
                const int y = row - z * ny;
                const int64_t d = sb.dst_off + z * A.dst_sz + y * A.dst_sy + x;
                out64[d] = o[j];
                out32[d] = (float)o[j];
            }
        }
    }
    if (!erode) return;
    __syncthreads();
    for (int row = r0; row < nrows; row += rpi) {
        const int z = (int)(((float)row + 0.5f) * inv_ny);
        const int y = row - z * ny;
        const int pos = row * px + x;
        double o = tile[pos];
        if (x > 0) o = fmin(o, tile[pos - 1]);
        if (x < nx - 1) o = fmin(o, tile[pos + 1]);
        if (y > 0) o = fmin(o, tile[pos - px]);
        if (y < ny - 1) o = fmin(o, tile[pos + px]);
        if (z > 0) o = fmin(o, tile[pos - ny * px]);
        if (z < nz - 1) o = fmin(o, tile[pos + ny * px]);
        const int64_t d = sb.dst_off + z * A.dst_sz + y * A.dst_sy + x;
        out64[d] = o;
        out32[d] = (float)o;
    }
}

// ---- tiles with every side <= 64 that do not fit LDS (finer-than-0.8-um voxels make the stock
// denoise_size 25 um a 33..64-voxel tile): the float64 working copy lives in a global scratch (L2),
// one 256-lane workgroup per tile.  Same register-resident line pass as the LDS kernel (64-register
// lines): z and y lines are read straight from the scratch (lanes along x: coalesced), x lines go
// through an LDS transpose in chunks of PP_MID_ROWS rows.
#define PP_MIDL 64
#define PP_WG_MID 256
#define PP_MID_ROWS 128

template <typename InT>
__global__ void __launch_bounds__(PP_WG_MID)
pp_mid_kernel(const InT* __restrict__ vol, int64_t sz, int64_t sy, int64_t sx,
              const mmx_subblock* __restrict__ subs, const mmx_quantile_class* __restrict__ qcs,
              const double* __restrict__ wts,
              pp_args A, float* __restrict__ out32, double* __restrict__ out64,
              mmx_subblock_info* __restrict__ info, double* __restrict__ scratch)
{
    extern __shared__ double xrows[];              // [PP_MID_ROWS][nx | 1] for the x pass
    __shared__ uint32_t hist[PP_HIST];
    __shared__ int s_bin[4];
    __shared__ uint32_t s_res[4];
    __shared__ int s_val[4];
    __shared__ double s_red[PP_WG_MID / 64];
    __shared__ double s_mean;
    __shared__ int s_flags;
    __shared__ pp_stack s_stack;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const mmx_subblock sb = subs[blockIdx.x];
    const int nz = sb.nz, ny = sb.ny, nx = sb.nx;
    const int n = nz * ny * nx;
    const int nrows = nz * ny;
    const InT* src = vol + sb.src_off;
    double* buf = scratch + sb.scratch_off;
    auto raw = [&](int i) {                 // (rare paths only: two integer divisions)
        const int t = i / nx, x = i - t * nx, z = t / ny, y = t - z * ny;
        return (int)src[z * sz + y * sy + x * sx];
    };
    // element-wise stages: a lane keeps its x and walks rows (z*ny + y) in steps of rpi, no per-voxel
    // index arithmetic; rows < 4096, so the float reciprocal gives row / ny exactly
    const float inv_nx = 1.0f / (float)nx, inv_ny = 1.0f / (float)ny;
    const int rpi = PP_WG_MID / nx;
    const int r_first = (int)(((float)tid + 0.5f) * inv_nx);
    const int x = tid - r_first * nx;
    const bool lane_on = r_first < rpi;
    auto raw_at = [&](int row) {
        const int z = (int)(((float)row + 0.5f) * inv_ny);
        const int y = row - z * ny;
        return (int)src[z * sz + y * sy + x * sx];
    };

    for (int i = tid; i < PP_HIST; i += PP_WG_MID) hist[i] = 0;
    __syncthreads();
    // (8 rows per lane in flight: with 8 waves per CU nothing else hides the load latency)
    for (int rb = 0; rb < nrows; rb += 8 * rpi) {          // same trip count in every lane (ballots)
        int v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = rb + j * rpi + r_first;
            v[j] = (lane_on && row < nrows) ? raw_at(row) : 0;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = rb + j * rpi + r_first;
            pp_hist_add(hist, v[j] >> 8, lane_on && row < nrows);
        }
    }
    __syncthreads();
    const mmx_quantile_class qc = qcs[sb.qclass];
    {
        const uint32_t rank = wave == 0 ? qc.lo_prev : wave == 1 ? qc.lo_next : wave == 2 ? qc.hi_prev : qc.hi_next;
        int b; uint32_t r;
        pp_select(hist, rank, b, r);
        if (lane == 0) { s_bin[wave] = b; s_res[wave] = r; }
    }
    __syncthreads();
    {
        const int b0 = s_bin[0], b1 = s_bin[1], b2 = s_bin[2], b3 = s_bin[3];
        const bool u1 = b1 != b0, u2 = b2 != b0 && b2 != b1, u3 = b3 != b0 && b3 != b1 && b3 != b2;
        for (int rb = lane_on ? r_first : nrows; rb < nrows; rb += 8 * rpi) {
            int vv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) vv[j] = rb + j * rpi < nrows ? raw_at(rb + j * rpi) : -1;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (vv[j] < 0) continue;
                const int hi = vv[j] >> 8, lo = vv[j] & 255;
                if (hi == b0) atomicAdd(&hist[256 + lo], 1u);
                if (u1 && hi == b1) atomicAdd(&hist[512 + lo], 1u);
                if (u2 && hi == b2) atomicAdd(&hist[768 + lo], 1u);
                if (u3 && hi == b3) atomicAdd(&hist[1024 + lo], 1u);
            }
        }
    }
    __syncthreads();
    {
        int slot = wave;
        for (int w = wave - 1; w >= 0; --w) if (s_bin[w] == s_bin[wave]) slot = w;
        int b; uint32_t r;
        pp_select(hist + 256 * (1 + slot), s_res[wave], b, r);
        if (lane == 0) s_val[wave] = (s_bin[wave] << 8) | b;
    }
    __syncthreads();

    pp_sat S;
    double info_vmin, info_vmax;
    {
        const double vmin = pp_lerp(s_val[0], s_val[1], qc.lo_gamma);
        double vmax = pp_lerp(s_val[2], s_val[3], qc.hi_gamma);
        S.identity = vmin == vmax;
        if (vmax < A.max_thresh) vmax = A.max_thresh;
        S.vmin = vmin; S.vmax = vmax; S.span = vmax - vmin;
        info_vmin = vmin; info_vmax = vmax;
        S.finish();
    }

    double part = 0.;
    for (int rb = lane_on ? r_first : nrows; rb < nrows; rb += 8 * rpi) {
        int vv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) vv[j] = rb + j * rpi < nrows ? raw_at(rb + j * rpi) : -1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (vv[j] < 0) continue;
            const double rv = (double)vv[j];
            const double s = S.fast ? S(rv) : S.plain(rv);
            part += s;
            buf[(rb + j * rpi) * nx + x] = pp_clip(s, A.clip_min, A.clip_max);
        }
    }
    part = pp_wave_sum(part);
    if (lane == 0) s_red[wave] = part;
    __syncthreads();
    if (tid == 0) {
        double tot = 0.;
        for (int w = 0; w < PP_WG_MID / 64; ++w) tot += s_red[w];
        double mean = tot / (double)n;
        int flags = S.identity ? MMX_PP_IDENTITY : 0;
        if (A.do_erosion) {
            const double tol = 1e-9 * (fabs(A.ero_thr) > 1. ? fabs(A.ero_thr) : 1.);
            if (!S.identity && fabs(mean - A.ero_thr) <= tol) {
                auto val = [&](int i) { return S.plain((double)raw(i)); };
                mean = pp_pairwise(val, n, &s_stack) / (double)n;
                flags |= MMX_PP_EXACT_MEAN;
            }
            if (mean > A.ero_thr) flags |= MMX_PP_ERODED;
        }
        s_mean = mean;
        s_flags = flags;
    }
    __syncthreads();          // also orders the scratch writes above before the passes below
    const int flags = s_flags;
    if (info && tid == 0) {
        mmx_subblock_info o;
        o.vmin = info_vmin; o.vmax = info_vmax; o.mean = s_mean; o.flags = flags; o._pad = 0;
        info[blockIdx.x] = o;
    }

    if (A.do_unsharp) {
        double wl[PP_R + 1];
#pragma unroll
        for (int k = 0; k <= PP_R; ++k) wl[k] = wts[k];
        // z lines, then y lines: lane <-> (row of the other axis, x), x fastest
#pragma unroll 1
        for (int axis = 0; axis < 2; ++axis) {
            const int L = axis == 0 ? nz : ny;
            const int nlines = n / L;
            for (int l = tid; l < nlines; l += PP_WG_MID) {
                int64_t base, stride;
                if (axis == 0) { base = l; stride = (int64_t)ny * nx; }
                else { const int z = l / nx, x = l - z * nx; base = (int64_t)z * ny * nx + x; stride = nx; }
                pp_line_pass<PP_MIDL>(buf, base, stride, L, wl);
            }
            __syncthreads();
        }
        // x lines through LDS: PP_MID_ROWS rows at a time, odd row pitch
        if (!(A.rgb_guess && nx == 3)) {
            const int pxl = nx | 1;
            for (int row0 = 0; row0 < nrows; row0 += PP_MID_ROWS) {
                const int nr = min(PP_MID_ROWS, nrows - row0);
                for (int i = tid; i < nr * nx; i += PP_WG_MID) {
                    const int r = i / nx, x = i - r * nx;
                    xrows[r * pxl + x] = buf[(int64_t)row0 * nx + i];
                }
                __syncthreads();
                if (tid < nr) pp_line_pass<PP_MIDL>(xrows, (int64_t)tid * pxl, 1, nx, wl);
                __syncthreads();
                for (int i = tid; i < nr * nx; i += PP_WG_MID) {
                    const int r = i / nx, x = i - r * nx;
                    buf[(int64_t)row0 * nx + i] = xrows[r * pxl + x];
                }
                __syncthreads();
            }
        }
    }

    const bool erode = flags & MMX_PP_ERODED;
    for (int rb = lane_on ? r_first : nrows; rb < nrows; rb += 8 * rpi) {
      int vv[8];
      double bl8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
          const int row = rb + j * rpi;
          vv[j] = row < nrows ? (A.do_unsharp ? raw_at(row) : 0) : -1;
          bl8[j] = row < nrows ? buf[row * nx + x] : 0.;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (vv[j] < 0) continue;
        const int row = rb + j * rpi;
        const int z = (int)(((float)row + 0.5f) * inv_ny);
        const int y = row - z * ny;
        const int i = row * nx + x;
        double o;
        if (A.do_unsharp) {
            const double rv = (double)vv[j];
            const double sv = S.fast ? S(rv) : S.plain(rv);
            const double den = pp_clip(sv, A.clip_min, A.clip_max);
            const double m = A.strength * bl8[j];
            const double hp = den - m;
            o = den + hp;
        } else {
            o = bl8[j];
        }
        if (erode) buf[i] = o;
        else {
            const int64_t d = sb.dst_off + z * A.dst_sz + y * A.dst_sy + x;
            out64[d] = o;
            out32[d] = (float)o;
        }
      }
    }
    if (!erode) return;
    __syncthreads();
    for (int row = lane_on ? r_first : nrows; row < nrows; row += rpi) {
        const int z = (int)(((float)row + 0.5f) * inv_ny);
        const int y = row - z * ny;
        const int i = row * nx + x;
        double o = buf[i];
        if (x > 0) o = fmin(o, buf[i - 1]);
        if (x < nx - 1) o = fmin(o, buf[i + 1]);
        if (y > 0) o = fmin(o, buf[i - nx]);
        if (y < ny - 1) o = fmin(o, buf[i + nx]);
        if (z > 0) o = fmin(o, buf[i - ny * nx]);
        if (z < nz - 1) o = fmin(o, buf[i + ny * nx]);
        const int64_t d = sb.dst_off + z * A.dst_sz + y * A.dst_sy + x;
        out64[d] = o;
        out32[d] = (float)o;
    }
}

// ---- any extent: data in a global scratch (2 * n doubles per sub-block), one output per lane and pass
template <typename InT>
__global__ void __launch_bounds__(PP_WG_GENERIC)
pp_generic_kernel(const InT* __restrict__ vol, int64_t sz, int64_t sy, int64_t sx,
                  const mmx_subblock* __restrict__ subs, const mmx_quantile_class* __restrict__ qcs,
                  const double* __restrict__ wts,
                  pp_args A, float* __restrict__ out32, double* __restrict__ out64,
                  mmx_subblock_info* __restrict__ info, double* __restrict__ scratch)
{
    __shared__ uint32_t hist[PP_HIST];
    __shared__ int s_bin[4];
    __shared__ uint32_t s_res[4];
    __shared__ int s_val[4];
    __shared__ double s_red[PP_WG_GENERIC / 64];
    __shared__ double s_mean;
    __shared__ int s_flags;
    __shared__ pp_stack s_stack;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const mmx_subblock sb = subs[blockIdx.x];
    const int64_t nz = sb.nz, ny = sb.ny, nx = sb.nx, n = nz * ny * nx;
    const InT* src = vol + sb.src_off;
    double* bufA = scratch + sb.scratch_off;
    double* bufB = bufA + n;
    constexpr bool F64 = std::is_same<InT, double>::value;
    auto raw = [&](int64_t i) {
        const int64_t t = i / nx, x = i - t * nx, z = t / ny, y = t - z * ny;
        if constexpr (F64) return (double)src[z * sz + y * sy + x * sx];
        else return (int)src[z * sz + y * sy + x * sx];
    };
    const mmx_quantile_class qc = qcs[sb.qclass];
    double q_val[4];                        // the four order statistics np.percentile interpolates between
    if constexpr (F64) {
        // float64 voxels: an eight-level radix select on the order-preserving bit pattern of the doubles (sign bit
        // flipped for positives, all bits for negatives), the four ranks side by side -- one 256-bin histogram each
        // per level, counted over the voxels that share the rank's prefix so far
        __shared__ unsigned long long s_pre[4];
        auto key = [](double v) {
            const unsigned long long u = (unsigned long long)__double_as_longlong(v + 0.0);     // (-0.0 sorts as +0.0)
            return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
        };
        for (int64_t i = tid; i < n; i += PP_WG_GENERIC) bufA[i] = raw(i);
        if (tid < 4) {
            s_pre[tid] = 0ull;
            s_res[tid] = tid == 0 ? qc.lo_prev : tid == 1 ? qc.lo_next : tid == 2 ? qc.hi_prev : qc.hi_next;
        }
        __syncthreads();
        for (int level = 7; level >= 0; --level) {
            for (int i = tid; i < PP_HIST; i += PP_WG_GENERIC) hist[i] = 0;
            __syncthreads();
            const unsigned long long p0 = s_pre[0], p1 = s_pre[1], p2 = s_pre[2], p3 = s_pre[3];
            const int sh = 8 * level;
            for (int64_t i = tid; i < n; i += PP_WG_GENERIC) {
                const unsigned long long k = key(bufA[i]);
                const unsigned long long up = level == 7 ? 0ull : k >> (sh + 8);
                const unsigned b8 = (unsigned)(k >> sh) & 255u;
                if (up == p0) atomicAdd(&hist[256 + b8], 1u);
                if (up == p1) atomicAdd(&hist[512 + b8], 1u);
                if (up == p2) atomicAdd(&hist[768 + b8], 1u);
                if (up == p3) atomicAdd(&hist[1024 + b8], 1u);
            }
            __syncthreads();
            if (wave < 4) {
                int b; uint32_t r;
                pp_select(hist + 256 * (1 + wave), s_res[wave], b, r);
                if (lane == 0) { s_pre[wave] = (s_pre[wave] << 8) | (unsigned long long)b; s_res[wave] = r; }
            }
            __syncthreads();
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const unsigned long long k = s_pre[w];
            const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
            q_val[w] = __longlong_as_double((long long)u);
        }
    } else {

        for (int i = tid; i < PP_HIST; i += PP_WG_GENERIC) hist[i] = 0;
        __syncthreads();
        for (int64_t i = tid; i < n; i += PP_WG_GENERIC) {
            const int v = raw(i);
            bufA[i] = (double)v;
            atomicAdd(&hist[v >> 8], 1u);
        }
        __syncthreads();
        if (wave < 4) {
            const uint32_t rank = wave == 0 ? qc.lo_prev : wave == 1 ? qc.lo_next : wave == 2 ? qc.hi_prev : qc.hi_next;
            int b; uint32_t r;
            pp_select(hist, rank, b, r);
            if (lane == 0) { s_bin[wave] = b; s_res[wave] = r; }
        }
        __syncthreads();
        {
            const int b0 = s_bin[0], b1 = s_bin[1], b2 = s_bin[2], b3 = s_bin[3];
            for (int64_t i = tid; i < n; i += PP_WG_GENERIC) {
                const int v = (int)bufA[i];
                const int hi = v >> 8, lo = v & 255;
                if (hi == b0) atomicAdd(&hist[256 + lo], 1u);
                if (hi == b1) atomicAdd(&hist[512 + lo], 1u);
                if (hi == b2) atomicAdd(&hist[768 + lo], 1u);
                if (hi == b3) atomicAdd(&hist[1024 + lo], 1u);
            }
        }
        __syncthreads();
        if (wave < 4) {
            int b; uint32_t r;
            pp_select(hist + 256 * (1 + wave), s_res[wave], b, r);
            if (lane == 0) s_val[wave] = (s_bin[wave] << 8) | b;
        }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 4; ++w) q_val[w] = (double)s_val[w];
    }

    pp_sat S;
    double info_vmin, info_vmax;
    {
        const double vmin = pp_lerp(q_val[0], q_val[1], qc.lo_gamma);
        double vmax = pp_lerp(q_val[2], q_val[3], qc.hi_gamma);
        S.identity = vmin == vmax;
        if (vmax < A.max_thresh) vmax = A.max_thresh;
        S.vmin = vmin; S.vmax = vmax; S.span = vmax - vmin;
        info_vmin = vmin; info_vmax = vmax;
        S.finish();
    }

    // (an identity tile -- vmin == vmax: the reference leaves the voxels alone -- of a float64 image may hold negative
    //  values, which the vmin = 0 stand-in of pp_sat would clip)
    auto sat = [&](double v) { return (F64 && S.identity) ? v : S.plain(v); };
    double part = 0.;
    for (int64_t i = tid; i < n; i += PP_WG_GENERIC) {
        const double s = sat(bufA[i]);
        part += s;
        bufA[i] = pp_clip(s, A.clip_min, A.clip_max);
    }
    part = pp_wave_sum(part);
    if (lane == 0) s_red[wave] = part;
    __syncthreads();
    if (tid == 0) {
        double tot = 0.;
        for (int w = 0; w < PP_WG_GENERIC / 64; ++w) tot += s_red[w];
        double mean = tot / (double)n;
        int flags = S.identity ? MMX_PP_IDENTITY : 0;
        if (A.do_erosion) {
            const double tol = 1e-9 * (fabs(A.ero_thr) > 1. ? fabs(A.ero_thr) : 1.);
            if (!S.identity && fabs(mean - A.ero_thr) <= tol && n < (1ll << 31)) {
                auto val = [&](int i) { return sat((double)raw(i)); };
                mean = pp_pairwise(val, (int)n, &s_stack) / (double)n;
                flags |= MMX_PP_EXACT_MEAN;
            }
            if (mean > A.ero_thr) flags |= MMX_PP_ERODED;
        }
        s_mean = mean;
        s_flags = flags;
    }
    __syncthreads();
    const int flags = s_flags;
    if (info && tid == 0) {
        mmx_subblock_info o;
        o.vmin = info_vmin; o.vmax = info_vmax; o.mean = s_mean; o.flags = flags; o._pad = 0;
        info[blockIdx.x] = o;
    }

    // ---- total-variation denoising of the clipped tile (reference plot_3d.py:147-149 ->
    // skimage.restoration.denoise_tv_chambolle, restoration/_denoise.py:315-393): Chambolle's projection
    // iteration on the dual field p, every voxel operation and both energy sums (NumPy pairwise order) as
    // NumPy does them, so that the iteration stops where the reference's stops and returns the same bits.
    const bool tv = A.tv_weight != 0.0;
    double* p0 = bufB + n;
    if (tv) {
        __shared__ pp_leaves s_leaves;
        __shared__ int s_done;
        __shared__ double s_e[2];                         // E_init, E_previous
        double* img = bufA;
        double* outb = bufB;
        double* p1 = p0 + n;
        double* p2 = p1 + n;
        double* sq = p2 + n;                              // d^2, then the gradient norms
        double* nm = sq + n;
        const int64_t pl = ny * nx;
        const double tau = 1.0 / 6.0;                     // 1 / (2 * ndim)
        for (int64_t i = tid; i < n; i += PP_WG_GENERIC) { p0[i] = 0.0; p1[i] = 0.0; p2[i] = 0.0; }
        if (tid == 0) { pp_enum_leaves((int)n, &s_stack, &s_leaves); s_done = 0; }
        __syncthreads();
        const bool par = s_leaves.count <= PP_MAXLEAF;
#pragma unroll 1
        for (int it = 0; it < 200; ++it) {
            for (int64_t i = tid; i < n; i += PP_WG_GENERIC) {
                const int64_t t = i / nx, x = i - t * nx, z = t / ny, y = t - z * ny;
                double d = 0.0;
                if (it > 0) {
                    d = -((p0[i] + p1[i]) + p2[i]);               // -p.sum(0)
                    if (z > 0) d += p0[i - pl];
                    if (y > 0) d += p1[i - nx];
                    if (x > 0) d += p2[i - 1];
                }
                outb[i] = it > 0 ? img[i] + d : img[i];
                sq[i] = d * d;
            }
            __syncthreads();
            for (int64_t i = tid; i < n; i += PP_WG_GENERIC) {
                const int64_t t = i / nx, x = i - t * nx, z = t / ny, y = t - z * ny;
                const double o = outb[i];
                const double g0 = z < nz - 1 ? outb[i + pl] - o : 0.0;
                const double g1 = y < ny - 1 ? outb[i + nx] - o : 0.0;
                const double g2 = x < nx - 1 ? outb[i + 1] - o : 0.0;
                double nrm = sqrt((g0 * g0 + g1 * g1) + g2 * g2);
                nm[i] = nrm;
                nrm = nrm * A.tv_factor;
                nrm = nrm + 1.0;
                p0[i] = (p0[i] - tau * g0) / nrm;
                p1[i] = (p1[i] - tau * g1) / nrm;
                p2[i] = (p2[i] - tau * g2) / nrm;
            }
            __syncthreads();
            if (par) {
                const int nl = s_leaves.count;
                for (int k = tid; k < 2 * nl; k += PP_WG_GENERIC) {
                    const int w = k >= nl, l = w ? k - nl : k;
                    const double* a = w ? nm : sq;
                    auto val = [&](int i) { return a[i]; };
                    s_leaves.sum[w][l] = pp_pairwise_leaf(val, s_leaves.lo[l], s_leaves.n[l]);
                }
                __syncthreads();
            }
            if (tid == 0) {
                double e1, e2;
                if (par) { e1 = pp_fold_leaves((int)n, &s_stack, &s_leaves, 0); e2 = pp_fold_leaves((int)n, &s_stack, &s_leaves, 1); }
                else {
                    auto v1 = [&](int i) { return sq[i]; };
                    auto v2 = [&](int i) { return nm[i]; };
                    e1 = pp_pairwise(v1, (int)n, &s_stack);
                    e2 = pp_pairwise(v2, (int)n, &s_stack);
                }
                double E = e1;
                E = E + A.tv_weight * e2;
                E = E / (double)n;
                if (it == 0) { s_e[0] = E; s_e[1] = E; }
                else if (fabs(s_e[1] - E) < 2.e-4 * s_e[0]) s_done = 1;
                else s_e[1] = E;
            }
            __syncthreads();
            if (s_done) break;
        }
    }

    // buffers from here on: `cur` = blur input / result, t1 / t2 = the blur's ping-pong pair, `den_buf` = the
    // denoised tile the unsharp mask adds to (recomputed from the voxels without total-variation denoising)
    const double* den_buf = tv ? bufB : nullptr;
    double* cur = tv ? bufB : bufA;
    double* const t1 = tv ? p0 : bufB;
    double* const t2 = tv ? p0 + n : bufA;
    double* oth = tv ? p0 + 2 * n : bufB;
    if (A.do_unsharp) {
        int pass = 0;
#pragma unroll 1
        for (int axis = 0; axis < 3; ++axis) {
            if (axis == 2 && A.rgb_guess && nx == 3) break;
            double* const dstb = (pass & 1) ? t2 : t1;
            const int64_t L = axis == 0 ? nz : axis == 1 ? ny : nx;
            const int64_t stride = axis == 0 ? ny * nx : axis == 1 ? nx : 1;
            for (int64_t i = tid; i < n; i += PP_WG_GENERIC) {
                const int64_t c = (i / stride) % L;
                const double* line = cur + (i - c * stride);
                double acc = line[c * stride] * wts[0];
                for (int k = PP_R; k >= 1; --k) {
                    const int64_t a = c - k < 0 ? 0 : c - k;
                    const int64_t b = c + k > L - 1 ? L - 1 : c + k;
                    acc += (line[a * stride] + line[b * stride]) * wts[k];
                }
                dstb[i] = acc;
            }
            __syncthreads();
            cur = dstb;
            ++pass;
        }
        if (!tv) oth = cur == bufA ? bufB : bufA;
    }

    const bool erode = flags & MMX_PP_ERODED;
    for (int64_t i = tid; i < n; i += PP_WG_GENERIC) {
        const int64_t t = i / nx, x = i - t * nx, z = t / ny, y = t - z * ny;
        double o;
        if (A.do_unsharp) {
            const double den = tv ? den_buf[i]
                                  : pp_clip(sat((double)raw(i)), A.clip_min, A.clip_max);
            const double m = A.strength * cur[i];
            const double hp = den - m;
            o = den + hp;
        } else {
            o = cur[i];
        }
        if (erode) oth[i] = o;
        else {
            const int64_t d = sb.dst_off + z * A.dst_sz + y * A.dst_sy + x;
            out64[d] = o;
            out32[d] = (float)o;
        }
    }
    if (!erode) return;
    __syncthreads();
    for (int64_t i = tid; i < n; i += PP_WG_GENERIC) {
        const int64_t t = i / nx, x = i - t * nx, z = t / ny, y = t - z * ny;
        double o = oth[i];
        if (x > 0) o = fmin(o, oth[i - 1]);
        if (x < nx - 1) o = fmin(o, oth[i + 1]);
        if (y > 0) o = fmin(o, oth[i - nx]);
        if (y < ny - 1) o = fmin(o, oth[i + nx]);
        if (z > 0) o = fmin(o, oth[i - ny * nx]);
        if (z < nz - 1) o = fmin(o, oth[i + ny * nx]);
        const int64_t d = sb.dst_off + z * A.dst_sz + y * A.dst_sy + x;
        out64[d] = o;
        out32[d] = (float)o;
    }
}


}  // namespace

extern "C" {

int64_t mmx_preprocess_fast_lds(int nz, int ny, int nx)
{
    if (nz < 1 || ny < 1 || nx < 1 || nz > PP_MAXL || ny > PP_MAXL || nx > PP_MAXL) return 0;
    const int64_t n = (int64_t)nz * ny * nx;
    const int64_t b = (int64_t)nz * ny * (nx | 1) * (int64_t)sizeof(double) + PP_HIST * (int64_t)sizeof(uint32_t)
                      + ((n * 2 + 7) & ~7ll);      // float64 tile, histograms, the uint16 voxels
    return b <= MMX_PP_MAX_LDS - 1024 ? b : 0;     // < 1 KiB of static LDS (selection, recursion stack)
}

static int pp_check(const mmx_volume* vol, const mmx_subblock* d_subs, const mmx_subblock* h_subs, int n_subs,
                    const mmx_quantile_class* d_qc, int n_qc, const mmx_preproc_params* p, const double* h_w,
                    float* d_out32, double* d_out64)
{
    if (!vol || !vol->d_data || !d_subs || !h_subs || n_subs < 0 || !d_qc || n_qc < 1 || !p || !h_w ||
        !d_out32 || !d_out64)
        return MMX_ERR_ARG;
    if (p->radius != PP_R) return MMX_ERR_UNSUPPORTED;
    // (float64 images: the one-output-per-lane kernel of the generic entry only; float32 images compute in float32 in
    //  the reference -- another arithmetic, not built)
    if (vol->dtype != MMX_U8 && vol->dtype != MMX_U16 && vol->dtype != MMX_F64) return MMX_ERR_UNSUPPORTED;
    for (int i = 0; i < n_subs; ++i) {
        const mmx_subblock& b = h_subs[i];
        if (b.nz < 1 || b.ny < 1 || b.nx < 1 || b.qclass < 0 || b.qclass >= n_qc || b.src_off < 0 || b.dst_off < 0)
            return MMX_ERR_ARG;
    }
    return MMX_OK;
}

int64_t mmx_preprocess_work_bytes(const mmx_subblock* h_subs, int n_subs)
{
    if (!h_subs || n_subs < 0) return 0;
    return mmx_pp_pipe_work_bytes(h_subs, n_subs);
}

int mmx_preprocess_batch_mode(const mmx_volume* vol, const mmx_subblock* d_subs, const mmx_subblock* h_subs,
                              int n_subs, const mmx_quantile_class* d_qclasses, int n_qclasses,
                              const mmx_preproc_params* params, const double* d_weights,
                              int64_t dst_sy, int64_t dst_sz, float* d_out32, double* d_out64,
                              mmx_subblock_info* d_info, int mode, int tiles_per_wg,
                              void* d_work, int64_t work_bytes, void* stream)
{
    const int st = pp_check(vol, d_subs, h_subs, n_subs, d_qclasses, n_qclasses, params, d_weights, d_out32, d_out64);
    if (st != MMX_OK) return st;
    if (mode != MMX_PP_AUTO && mode != MMX_PP_SINGLE && mode != MMX_PP_PIPELINED) return MMX_ERR_ARG;
    if (params->tv_weight != 0.0) return MMX_ERR_UNSUPPORTED;      // total-variation denoising: the generic entry
    if (vol->dtype == MMX_F64) return MMX_ERR_UNSUPPORTED;         // float64 voxels: the generic entry
    if (n_subs == 0) return MMX_OK;
    int64_t lds = 0;
    for (int i = 0; i < n_subs; ++i) {
        const int64_t b = mmx_preprocess_fast_lds(h_subs[i].nz, h_subs[i].ny, h_subs[i].nx);
        if (!b) return MMX_ERR_UNSUPPORTED;
        lds = b > lds ? b : lds;
    }
    const pp_args A = pp_make_args(params, dst_sy, dst_sz);
    hipStream_t s = (hipStream_t)stream;
    const int grid = ((n_subs + 7) / 8) * 8;       // 8 XCD-contiguous runs of tiles
    // small tiles (anisotropic voxels: 25 um is 4 x 23 x 23 voxels at 6.6 x 1.1 x 1.1 um): 256-lane
    // workgroups, so that LDS (not the 32-wave limit) decides how many tiles a CU works on at once
    int64_t max_vox = 0;
    for (int i = 0; i < n_subs; ++i)
        max_vox = std::max<int64_t>(max_vox, (int64_t)h_subs[i].nz * h_subs[i].ny * h_subs[i].nx);
    const bool small = max_vox <= 4096;
    mmx_timed_scope ts(MMX_K_PREPROC, s);
    // tiles that fill a CU's LDS one at a time: statistics kernel + pipelined blur kernel (mmx_preproc_pipe.hip)
    if (mode == MMX_PP_PIPELINED || (mode == MMX_PP_AUTO && !small)) {
        const int64_t need = mmx_pp_pipe_work_bytes(h_subs, n_subs);
        if (d_work && work_bytes < need) return MMX_ERR_WORKSPACE;
        // (callers of the plain entry bring neither: both come from the stream's memory pool for the call)
        mmx_subblock_info* inf = d_info;
        void* work = d_work;
        if (!inf && hipMallocAsync((void**)&inf, (size_t)n_subs * sizeof(mmx_subblock_info), s) != hipSuccess)
            return MMX_ERR_HIP;
        if (!work && hipMallocAsync(&work, (size_t)need, s) != hipSuccess) {
            if (!d_info) (void)hipFreeAsync(inf, s);       // (the table just taken from the pool goes back)
            return MMX_ERR_HIP;
        }
        const int rc = mmx_launch_pp_pipe(vol, d_subs, h_subs, n_subs, d_qclasses, d_weights, A, d_out32, d_out64,
                                          inf, work, tiles_per_wg > 0 ? tiles_per_wg : MMX_PP_TILES_PER_WG, s);
        // (both are returned whatever the first one's status: a failed free must not strand the other block)
        const bool freed_inf = d_info || hipFreeAsync(inf, s) == hipSuccess;
        const bool freed_work = d_work || hipFreeAsync(work, s) == hipSuccess;
        if (!freed_inf || !freed_work) return MMX_ERR_HIP;
        if (rc != MMX_ERR_UNSUPPORTED || mode == MMX_PP_PIPELINED) return rc;
    }
#define PP_FAST_LAUNCH(T, W)                                                                              \
    do {                                                                                                  \
        auto k = pp_fast_kernel<T, W>;                                                                    \
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
            return MMX_ERR_HIP;                                                                           \
        hipLaunchKernelGGL(k, dim3(grid), dim3(W), (size_t)lds, s, (const T*)vol->d_data, vol->stride_z,  \
                           vol->stride_y, vol->stride_x, d_subs, n_subs, d_qclasses, d_weights, A,        \
                           d_out32, d_out64, d_info);                                                     \
    } while (0)
    if (vol->dtype == MMX_U16) { if (small) PP_FAST_LAUNCH(uint16_t, 256); else PP_FAST_LAUNCH(uint16_t, PP_WG); }
    else { if (small) PP_FAST_LAUNCH(uint8_t, 256); else PP_FAST_LAUNCH(uint8_t, PP_WG); }
#undef PP_FAST_LAUNCH
    return hipGetLastError() == hipSuccess ? MMX_OK : MMX_ERR_HIP;
}

int mmx_preprocess_batch(const mmx_volume* vol, const mmx_subblock* d_subs, const mmx_subblock* h_subs,
                         int n_subs, const mmx_quantile_class* d_qclasses, int n_qclasses,
                         const mmx_preproc_params* params, const double* d_weights,
                         int64_t dst_sy, int64_t dst_sz, float* d_out32, double* d_out64,
                         mmx_subblock_info* d_info, void* stream)
{
    return mmx_preprocess_batch_mode(vol, d_subs, h_subs, n_subs, d_qclasses, n_qclasses, params, d_weights,
                                     dst_sy, dst_sz, d_out32, d_out64, d_info, MMX_PP_AUTO, 0, nullptr, 0, stream);
}

int mmx_preprocess_batch_generic(const mmx_volume* vol, const mmx_subblock* d_subs,
                                 const mmx_subblock* h_subs, int n_subs,
                                 const mmx_quantile_class* d_qclasses, int n_qclasses,
                                 const mmx_preproc_params* params, const double* d_weights,
                                 int64_t dst_sy, int64_t dst_sz, float* d_out32, double* d_out64,
                                 mmx_subblock_info* d_info, double* d_scratch, int64_t scratch_doubles,
                                 void* stream)
{
    const int st = pp_check(vol, d_subs, h_subs, n_subs, d_qclasses, n_qclasses, params, d_weights, d_out32, d_out64);
    if (st != MMX_OK) return st;
    if (n_subs == 0) return MMX_OK;
    if (!d_scratch) return MMX_ERR_ARG;
    for (int i = 0; i < n_subs; ++i) {
        const mmx_subblock& b = h_subs[i];
        const int64_t n = (int64_t)b.nz * b.ny * b.nx;
        const int64_t per_voxel = params->tv_weight != 0.0 ? PP_TV_SCRATCH : 2;
        if (b.scratch_off < 0 || b.scratch_off + per_voxel * n > scratch_doubles) return MMX_ERR_WORKSPACE;
        if (params->tv_weight != 0.0 && n >= (1ll << 31)) return MMX_ERR_UNSUPPORTED;
    }
    const pp_args A = pp_make_args(params, dst_sy, dst_sz);
    hipStream_t s = (hipStream_t)stream;
    mmx_timed_scope ts(MMX_K_PREPROC, s);
    // every side <= 64: register-resident lines over the scratch (pp_mid_kernel); else one output per lane
    bool mid = params->tv_weight == 0.0 && vol->dtype != MMX_F64;   // (the iteration, and float64 voxels, live in the one-output-per-lane kernel only)
    int max_nx = 1;
    for (int i = 0; i < n_subs; ++i) {
        const mmx_subblock& b = h_subs[i];
        if (b.nz > PP_MIDL || b.ny > PP_MIDL || b.nx > PP_MIDL || (int64_t)b.nz * b.ny * b.nx >= (1ll << 30)) mid = false;
        max_nx = b.nx > max_nx ? b.nx : max_nx;
    }
    if (mid) {
        const size_t lds = (size_t)PP_MID_ROWS * (max_nx | 1) * sizeof(double);
        if (vol->dtype == MMX_U16) {
            auto k = pp_mid_kernel<uint16_t>;
            if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return MMX_ERR_HIP;
            hipLaunchKernelGGL(k, dim3(n_subs), dim3(PP_WG_MID), lds, s, (const uint16_t*)vol->d_data,
                               vol->stride_z, vol->stride_y, vol->stride_x, d_subs, d_qclasses, d_weights, A,
                               d_out32, d_out64, d_info, d_scratch);
        } else {
            auto k = pp_mid_kernel<uint8_t>;
            if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return MMX_ERR_HIP;
            hipLaunchKernelGGL(k, dim3(n_subs), dim3(PP_WG_MID), lds, s, (const uint8_t*)vol->d_data,
                               vol->stride_z, vol->stride_y, vol->stride_x, d_subs, d_qclasses, d_weights, A,
                               d_out32, d_out64, d_info, d_scratch);
        }
        return hipGetLastError() == hipSuccess ? MMX_OK : MMX_ERR_HIP;
    }
    if (vol->dtype == MMX_F64)
        hipLaunchKernelGGL(pp_generic_kernel<double>, dim3(n_subs), dim3(PP_WG_GENERIC), 0, s,
                           (const double*)vol->d_data, vol->stride_z, vol->stride_y, vol->stride_x, d_subs,
                           d_qclasses, d_weights, A, d_out32, d_out64, d_info, d_scratch);
    else if (vol->dtype == MMX_U16)
        hipLaunchKernelGGL(pp_generic_kernel<uint16_t>, dim3(n_subs), dim3(PP_WG_GENERIC), 0, s,
                           (const uint16_t*)vol->d_data, vol->stride_z, vol->stride_y, vol->stride_x, d_subs,
                           d_qclasses, d_weights, A, d_out32, d_out64, d_info, d_scratch);
    else
        hipLaunchKernelGGL(pp_generic_kernel<uint8_t>, dim3(n_subs), dim3(PP_WG_GENERIC), 0, s,
                           (const uint8_t*)vol->d_data, vol->stride_z, vol->stride_y, vol->stride_x, d_subs,
                           d_qclasses, d_weights, A, d_out32, d_out64, d_info, d_scratch);
    return hipGetLastError() == hipSuccess ? MMX_OK : MMX_ERR_HIP;
}

}  // extern "C"
