// extern "C" entry points of libmmx_hip.so (see include/mmx.h for the contract).

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "mmx_common.h"

int mmx_launch_generic_pass(int pass, const mmx_volume* vol, const mmx_block* d_blocks, int n_blocks,
                            int max_vox, int64_t slot_elems, const float* w0, const float* w2, int radius,
                            const float* in1, const float* in2, float* out1, float* out2, hipStream_t s);
int mmx_launch_peaks(const float* d_log, int n_sigma, int64_t sigma_stride, const mmx_block* d_blocks,
                     int n_blocks, int max_vox, int64_t slot_elems, float thr, float eps,
                     mmx_cand* d_cands, uint32_t cap, uint32_t* d_count, hipStream_t stream);
int mmx_launch_peaks_sparse(const float* d_log, const unsigned long long* d_mask, int n_sigma,
                            int64_t sigma_stride, const mmx_block* d_blocks, int n_blocks, int max_vox,
                            int64_t slot_elems, float thr, float eps, mmx_cand* d_cands, uint32_t cap,
                            uint32_t* d_count, int quads, hipStream_t stream);

#include <mutex>
#include <vector>

namespace {
thread_local char g_hip_err[256] = "";

struct span { hipEvent_t a, b; int kind; };
std::mutex g_tm;
bool g_timing = false;
uint32_t g_kinds = ~0u;              // kernel families that record events while g_timing is on
std::vector<span> g_spans;          // recorded spans of the current window
std::vector<hipEvent_t> g_pool;     // recycled events
hipEvent_t g_open[MMX_K_COUNT];

hipEvent_t take_event()
{
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    hipEventCreate(&e);
    return e;
}

int hip_fail(hipError_t e, const char* what)
{
    snprintf(g_hip_err, sizeof g_hip_err, "%s: %s", what, hipGetErrorString(e));
    return MMX_ERR_HIP;
}

// The register-ring column kernels prefetch kPrefetch (= 4) steps ahead and reflect once.
constexpr int kColPrefetch = 4;
}  // namespace

void mmx_time_begin(int kind, hipStream_t s)
{
    std::lock_guard<std::mutex> lk(g_tm);
    if (!g_timing || !((g_kinds >> kind) & 1u)) return;
    g_open[kind] = take_event();
    hipEventRecord(g_open[kind], s);
}

void mmx_time_end(int kind, hipStream_t s)
{
    std::lock_guard<std::mutex> lk(g_tm);
    if (!g_timing || !g_open[kind]) return;
    hipEvent_t b = take_event();
    hipEventRecord(b, s);
    g_spans.push_back({g_open[kind], b, kind});
    g_open[kind] = nullptr;
}

extern "C" {

int mmx_timing_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_tm);
    for (auto& sp : g_spans) { g_pool.push_back(sp.a); g_pool.push_back(sp.b); }
    g_spans.clear();
    for (int k = 0; k < MMX_K_COUNT; ++k) g_open[k] = nullptr;
    g_timing = on != 0;
    // 1: every family; any other non-zero value: bit (k + 1) selects family MMX_K_k
    g_kinds = on == 1 ? ~0u : ((uint32_t)on >> 1);
    return MMX_OK;
}

int mmx_timing_is_enabled(void)
{
    std::lock_guard<std::mutex> lk(g_tm);
    return g_timing ? 1 : 0;
}

int mmx_timing_read(double* ms, int64_t* launches, int n)
{
    if (!ms || !launches || n < MMX_K_COUNT) return MMX_ERR_ARG;
    std::lock_guard<std::mutex> lk(g_tm);
    for (int k = 0; k < n; ++k) { ms[k] = 0.0; launches[k] = 0; }
    for (auto& sp : g_spans) {
        hipError_t r = hipEventSynchronize(sp.b);
        if (r != hipSuccess) return hip_fail(r, "hipEventSynchronize");
        float t = 0.f;
        r = hipEventElapsedTime(&t, sp.a, sp.b);
        if (r != hipSuccess) return hip_fail(r, "hipEventElapsedTime");
        ms[sp.kind] += t;
        launches[sp.kind] += 1;
        g_pool.push_back(sp.a);
        g_pool.push_back(sp.b);
    }
    g_spans.clear();
    return MMX_OK;
}

int mmx_abi_version(void) { return MMX_ABI_VERSION; }
size_t mmx_workspace_bytes(int n_blocks, int64_t slot_elems, int n_sigma, int with_masks)
{
    if (n_blocks < 1 || slot_elems < 1 || n_sigma < 1) return 0;
    size_t bytes = (size_t)(4 + n_sigma) * (size_t)n_blocks * (size_t)slot_elems * sizeof(float);
    if (with_masks) {
        bytes = (bytes + 15) & ~(size_t)15;
        bytes += (size_t)n_sigma * (((size_t)n_blocks * (size_t)slot_elems) >> 5) * 16;
    }
    return bytes;
}

const char* mmx_strerror(int status)
{
    switch (status) {
        case MMX_OK: return "ok";
        case MMX_ERR_ARG: return "bad argument";
        case MMX_ERR_HIP: return "HIP runtime error";
        case MMX_ERR_NO_DEVICE: return "no gfx950 device";
        case MMX_ERR_WORKSPACE: return "workspace too small";
        case MMX_ERR_UNSUPPORTED: return "unsupported configuration";
        case MMX_DEFERRED: return "left to the caller";
        default: return "unknown status";
    }
}

const char* mmx_last_hip_error(void) { return g_hip_err; }


int mmx_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { hip_fail(e, "hipGetDeviceCount"); return -1; }
    int ok = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}

// Q16 tiles (MMX_ZX_TILED_Q16): P in [0, BP] as unorm16, Q in [-BQ, BQ] as snorm16.  For voxels in [0, 1] (integer
// types after img_as_float) the bounds follow from the weights alone.  |P| <= (sum w0)^2.  Q = sum K I with the 2-D
// kernel K(i, j) = w2(i) w0(j) + w0(i) w2(j) and every voxel I in [0, 1], so Q lies in [-sum of K's negative taps, sum of
// its positive taps] -- about HALF of sum|K| <= 2 sum|w2| sum w0 either way, a second-derivative kernel summing to ~0
// (round 6: BQ is that, the larger of the two one-sided sums; rounds 3-5 quantised Q over the two-sided 2 sum|w2| sum w0
// and carried twice the rounding error for it).  Folding reflected taps at a block face only merges weights, which
// can only shrink both one-sided sums.  The error the rounding leaves in the LoG value follows likewise:
//   norm (sum|w2| BP / 65535 + sum w0 BQ / 32767) / 2,
// i.e. 2.2e-5 whatever sigma (sum|w2| ~ 0.97 / sigma^2), plus the float32 arithmetic's own few 1e-7, the product
// term the 16-bit kernel leaves out (0.55e-5) and the rounding of its X accumulators, which run with the voxel
// pieces' exponent offsets still in them (values up to 8 instead of 1: four roundings of 2^-22 each, 0.2e-5):
// 3.0e-5; the Y pass on the matrix cores (mmx_ymfma.hip) leaves out its own low x low product -- low byte of a count x
// (weight - float16(weight)): 255 x 2^-12 = 0.062 counts per unit of weight against the 0.5 of the rounding -- which adds an
// eighth: 3.3e-5 in all (5.1e-5 with the two-sided BQ).
static void q16_bounds(const double* w0, const double* w2, int radius, double norm, double* bp, double* bq, double* err)
{
    double s0 = w0[0], s2 = fabs(w2[0]);
    for (int k = 1; k <= radius; ++k) { s0 += 2.0 * w0[k]; s2 += 2.0 * fabs(w2[k]); }
    *bp = s0 * s0 * (1.0 + 1e-6);
    // one-sided sums of K over its (2 R + 1)^2 taps (K is symmetric in both indices: a quadrant, weighted)
    double pos = 0.0, neg = 0.0;
    for (int i = 0; i <= radius; ++i)
        for (int j = 0; j <= radius; ++j) {
            const double k = (w2[i] * w0[j] + w0[i] * w2[j]) * ((i ? 2.0 : 1.0) * (j ? 2.0 : 1.0));
            if (k > 0.0) pos += k; else neg -= k;
        }
    // (1e-4 of slack: the kernel's own float32 / split-float16 arithmetic may land a hair beyond the exact extreme, and
    //  a value beyond BQ would clamp)
    *bq = (pos > neg ? pos : neg) * (1.0 + 1e-4);
    // ... plus what the 16-bit kernel drops in the X pass (low voxel byte x low weight piece: 255 / 65536 x 2^-11 per
    // unit of weight): P off by 1.9e-6 s0^2, Q by 1.9e-6 x 2 s2 s0
    const double drop = 255.0 / 65536.0 / 2048.0;
    // ... and the float32 rounding of X accumulators that carry the pieces' offsets (<= 8: ulp 2^-21, half of it per
    // MFMA, four MFMAs into each; relative to the bounds, the fragments carry 1 / bound)
    const double biased = 4.0 * 0x1p-22;
    // ... and the Y pass's dropped product, in counts of P and of Q
    const double ydrop = 255.0 / 4096.0;
    *err = norm * (s2 * (*bp / 65535.0 * (0.5 + ydrop) + drop * s0 * s0 + biased * *bp) +
                   s0 * (*bq / 32767.0 * (0.5 + ydrop) + drop * 2.0 * s2 * s0 + biased * *bq)) + 1e-6;
}

double mmx_tiled_q16_error_bound(const double* h_w0, const double* h_w2, int radius, double norm)
{
    if (!h_w0 || !h_w2 || radius < 0 || radius > MMX_MAX_RADIUS_GENERIC) return -1.0;
    double bp, bq, err;
    q16_bounds(h_w0, h_w2, radius, norm, &bp, &bq, &err);
    return err;
}

int mmx_log_batch_f32(const mmx_volume* vol, const mmx_block* d_blocks, const mmx_block* h_blocks,
                      int n_blocks, int64_t slot_elems,
                      const double* h_w0, const double* h_w2, int radius, double norm,
                      float* d_log, float* d_work, uint64_t* d_nms_mask, float nms_lo, float nms_eps,
                      int* h_mask_written, int zx_mode, int* h_zx_path, void* stream)
{
    if (h_mask_written) *h_mask_written = 0;
    if (h_zx_path) *h_zx_path = MMX_ZX_SEPARATE;
    const bool y_valu = zx_mode >= 0 && (zx_mode & MMX_ZX_Y_VALU);
    if (y_valu) zx_mode &= ~MMX_ZX_Y_VALU;
    const bool prepacked = zx_mode == (MMX_ZX_TILED | MMX_ZX_PREPACKED) || zx_mode == (MMX_ZX_TILED_Q16 | MMX_ZX_PREPACKED);
    if (prepacked) zx_mode &= ~MMX_ZX_PREPACKED;
    if (zx_mode < MMX_ZX_AUTO || zx_mode > MMX_ZX_TILED_Q16 || zx_mode == 1 || (zx_mode >= 3 && zx_mode <= 5))
        return MMX_ERR_ARG;             // (3, 4, 5: retired experiment kernels)
    if (!vol || !vol->d_data || !d_blocks || !h_blocks || !h_w0 || !h_w2 || !d_log || !d_work)
        return MMX_ERR_ARG;
    if (n_blocks < 1 || radius < 0 || slot_elems < 1) return MMX_ERR_ARG;
    if (n_blocks > MMX_MAX_BLOCKS) return MMX_ERR_UNSUPPORTED;
    if (radius > MMX_MAX_RADIUS_GENERIC) return MMX_ERR_UNSUPPORTED;
    if (slot_elems >= (int64_t(1) << 29)) return MMX_ERR_UNSUPPORTED;  // 32-bit byte offsets in a slot
    if (slot_elems % MMX_ROW_ALIGN) return MMX_ERR_ARG;
    double in_scale = 1.0;
    if (vol->dtype == MMX_U8) in_scale = 1.0 / 255.0;        // skimage img_as_float: x * (1/imax)
    else if (vol->dtype == MMX_U16) in_scale = 1.0 / 65535.0;
    else if (vol->dtype != MMX_F32) return MMX_ERR_UNSUPPORTED;  // float64 volumes: pass a float32 copy

    int min_nz = 1 << 30, min_ny = 1 << 30, min_nx = 1 << 30;
    int max_zcols = 0, max_ycols = 0, max_rows = 0, max_nx = 0, max_vox = 0;
    int64_t max_lane_in = 0;
    for (int i = 0; i < n_blocks; ++i) {
        const mmx_block& b = h_blocks[i];
        if (b.nz < 1 || b.ny < 1 || b.nx < 1 || b.slot != i) return MMX_ERR_ARG;
        if (b.px < b.nx || b.px % MMX_ROW_ALIGN) return MMX_ERR_ARG;
        if ((int64_t)b.nz * b.ny * b.px > slot_elems) return MMX_ERR_WORKSPACE;
        if (b.nz < min_nz) min_nz = b.nz;
        if (b.ny < min_ny) min_ny = b.ny;
        if (b.nx < min_nx) min_nx = b.nx;
        if (b.ny * b.px > max_zcols) max_zcols = b.ny * b.px;
        if (b.nz * b.px > max_ycols) max_ycols = b.nz * b.px;
        if (b.nz * b.ny > max_rows) max_rows = b.nz * b.ny;
        if (b.nx > max_nx) max_nx = b.nx;
        if (b.nz * b.ny * b.px > max_vox) max_vox = b.nz * b.ny * b.px;
        const int64_t lane = (int64_t)(b.ny - 1) * vol->stride_y + (int64_t)(b.nx - 1) * vol->stride_x;
        if (lane > max_lane_in) max_lane_in = lane;
    }
    const int64_t n_slots = n_blocks;
    float* t0 = d_work;                          // Gz
    float* t1 = d_work + n_slots * slot_elems;   // Gzz
    float* t2 = t1 + n_slots * slot_elems;       // A
    float* t3 = t2 + n_slots * slot_elems;       // BC
    hipStream_t s = (hipStream_t)stream;

    // weights per pass: input scale into the z pass, -norm into the x pass
    float wz0[MMX_MAX_RADIUS_GENERIC + 1], wz2[MMX_MAX_RADIUS_GENERIC + 1];
    float wy0[MMX_MAX_RADIUS_GENERIC + 1], wy2[MMX_MAX_RADIUS_GENERIC + 1];
    float wx0[MMX_MAX_RADIUS_GENERIC + 1], wx2[MMX_MAX_RADIUS_GENERIC + 1];
    for (int k = 0; k <= radius; ++k) {
        wz0[k] = (float)(h_w0[k] * in_scale);
        wz2[k] = (float)(h_w2[k] * in_scale);
        wy0[k] = (float)h_w0[k];
        wy2[k] = (float)h_w2[k];
        wx0[k] = (float)(-norm * h_w0[k]);
        wx2[k] = (float)(-norm * h_w2[k]);
    }
    const bool fast_r = radius >= 1 && radius <= MMX_MAX_RADIUS_FAST;
    const bool lane_ok = max_lane_in * 8 < (int64_t(1) << 31);
    const bool fast_z = fast_r && lane_ok && min_nz >= radius + kColPrefetch && vol->stride_y < (1 << 30);
    const bool fast_y = fast_r && min_ny >= radius + kColPrefetch;
    const bool fast_x = fast_r && min_nx >= radius;
    auto taps = [&](const float* a, const float* b) {
        mmx_taps_f32 t;
        for (int k = 0; k <= MMX_MAX_RADIUS_FAST; ++k) {
            t.w0[k] = k <= radius ? a[k] : 0.f;
            t.w2[k] = k <= radius ? b[k] : 0.f;
        }
        return t;
    };
    int rc;
    // Fused path: Z and X in one kernel (Gz / Gzz never touch HBM), then Y.
    int max_ny = 0, max_px = 0;
    for (int i = 0; i < n_blocks; ++i) {
        if (h_blocks[i].ny > max_ny) max_ny = h_blocks[i].ny;
        if (h_blocks[i].px > max_px) max_px = h_blocks[i].px;
    }
    const bool fused = zx_mode != MMX_ZX_SEPARATE && fast_r && lane_ok && fast_y && min_nz >= radius + 1 &&
                       min_nx >= radius && max_px <= 512 && vol->stride_y < (1 << 30);
    if (fused) {
        mmx_taps_f32 tzz = taps(wz0, wz2), txx = taps(wy0, wy2), tyy = taps(wx0, wx2);
        int path = MMX_ZX_PACKED;
        // AUTO = the tiled matrix-core path for integer voxels, else the packed-VALU kernel.  (The register-only
        // and LDS-staged matrix-core kernels are correct and selectable, but no faster than the packed one:
        // DESIGN.md section 4b -- their 16 planes x 64 bytes accesses were the limit, which the tiled form removes.)
        mmx_zx6_plan plan;
        // Q16 tiles when asked for, or under AUTO when the caller's NMS band covers their rounding error fourfold and
        // that error, in value units, is inside the LoG contract (MMX_LOG_ABS_TOL)
        // (the condition under which every true maximum is still nominated)
        double q_bp = 0, q_bq = 0, q_err = 0;
        q16_bounds(h_w0, h_w2, radius, norm, &q_bp, &q_bq, &q_err);
        // float voxels: the tiled path when the volume states its value range (or when asked for by name: the float16
        // pieces of its copy cover |v| < 65504), 16-bit tiles when that range is [0, m]: their bounds scale with m
        const bool integer = vol->dtype == MMX_U8 || vol->dtype == MMX_U16;
        const bool ranged = vol->dtype == MMX_F32 && vol->value_range != 0.f && fabsf(vol->value_range) < 60000.f;
        const bool nonneg = integer || (ranged && vol->value_range > 0.f);
        if (!integer && nonneg) { q_bp *= vol->value_range; q_bq *= vol->value_range; q_err *= vol->value_range; }
        const bool q16 = nonneg && (zx_mode == MMX_ZX_TILED_Q16 ||
                         (zx_mode == MMX_ZX_AUTO && d_nms_mask && h_mask_written && (double)nms_eps >= 4.0 * q_err &&
                          q_err <= MMX_LOG_ABS_TOL));
        bool tiled = (zx_mode == MMX_ZX_TILED || (zx_mode == MMX_ZX_TILED_Q16 && nonneg) ||
                      (zx_mode == MMX_ZX_AUTO && (integer || ranged))) &&
                     mmx_zx6_plan_make(h_blocks, n_blocks, slot_elems, vol->dtype, &plan) == MMX_OK;
        if (tiled && !prepacked) {
            mmx_timed_scope ts(MMX_K_ZXPACK, s);
            rc = mmx_launch_zx6_pack(vol, d_blocks, h_blocks, n_blocks, plan, d_work, s);
            if (rc == MMX_ERR_HIP) return hip_fail(hipGetLastError(), "voxel copy of the tiled path");
            tiled = rc == MMX_OK;
        }
        // the mask rows of a block (ny rows of ceil(nz * px / 64) words) must fit its slot / 32 words
        auto mask_fits = [&](bool tiles) {
            bool ok = d_nms_mask != nullptr && h_mask_written != nullptr;
            for (int b = 0; ok && b < n_blocks; ++b) {
                const mmx_block& hb = h_blocks[b];
                const int64_t need = tiles ? (int64_t)hb.ny * ((hb.nz + 3) >> 2) * ((hb.nx + 15) >> 4)
                                           : (int64_t)hb.ny * (((int64_t)hb.nz * hb.px + 63) >> 6);
                if (need > (slot_elems >> 5) - 1) ok = false;
            }
            return ok;
        };
        { mmx_timed_scope ts(MMX_K_ZX, s);
          rc = MMX_ERR_UNSUPPORTED;
          if (tiled) {
              path = q16 ? MMX_ZX_TILED_Q16 : MMX_ZX_TILED;
              rc = mmx_launch_zx6(vol, d_blocks, h_blocks, n_blocks, plan, txx, radius, d_work,
                                  q16 ? (float)(1.0 / q_bp) : 0.f, q16 ? (float)(1.0 / q_bq) : 0.f, s);
              tiled = rc == MMX_OK;
          }
          if (rc == MMX_ERR_UNSUPPORTED) {
              path = MMX_ZX_PACKED;
              rc = mmx_launch_zx2(vol, d_blocks, n_blocks, max_ny, max_px, slot_elems, tzz, txx, radius, t0, t1, s);
          } }
        if (rc == MMX_OK && h_zx_path) *h_zx_path = path;
        if (rc == MMX_OK) {
            mmx_timed_scope ts(MMX_K_Y2, s);
            const bool want_mask = mask_fits(tiled);      // (after the Z+X launch: `tiled` says which kernel ran)
            rc = MMX_ERR_UNSUPPORTED;
            if (tiled && q16 && !y_valu)
                rc = mmx_launch_ym(d_blocks, n_blocks, plan, slot_elems, tyy, radius, d_work,
                                   (float)(q_bp / 65535.0), (float)(q_bq / 32767.0), d_log,
                                   want_mask ? (unsigned long long*)d_nms_mask : nullptr, nms_lo, nms_eps, s);
            if (rc != MMX_ERR_UNSUPPORTED) ;
            else if (tiled)
                rc = mmx_launch_y6(d_blocks, n_blocks, plan, slot_elems, tyy, radius, d_work,
                                   reinterpret_cast<const float*>(reinterpret_cast<const char*>(d_work) + plan.q_off),
                                   q16 ? (float)(q_bp / 65535.0) : 0.f, q16 ? (float)(q_bq / 32767.0) : 0.f, d_log,
                                   want_mask ? (unsigned long long*)d_nms_mask : nullptr, nms_lo, nms_eps, s);
            else
                rc = mmx_launch_y2(d_blocks, n_blocks, max_ycols, slot_elems, tyy, radius, t0, t1, d_log,
                                   want_mask ? (unsigned long long*)d_nms_mask : nullptr, nms_lo, nms_eps, s);
            if (rc == MMX_OK && want_mask) *h_mask_written = tiled ? MMX_MASK_QUADS : MMX_MASK_ROWS;
        }
        if (rc == MMX_ERR_HIP) return hip_fail(hipGetLastError(), "fused passes");
        if (rc == MMX_OK) return MMX_OK;
        if (rc != MMX_ERR_UNSUPPORTED) return rc;   // unsupported geometry: separate passes below
    }
    { mmx_timed_scope ts(fast_z ? MMX_K_ZPASS : MMX_K_GENERIC, s);
    if (fast_z) rc = mmx_launch_zpass(vol, d_blocks, n_blocks, max_zcols, slot_elems, taps(wz0, wz2), radius, t0, t1, s);
    else rc = mmx_launch_generic_pass(0, vol, d_blocks, n_blocks, max_vox, slot_elems, wz0, wz2, radius, nullptr, nullptr, t0, t1, s); }
    if (rc != MMX_OK) return rc == MMX_ERR_HIP ? hip_fail(hipGetLastError(), "z pass") : rc;
    { mmx_timed_scope ts(fast_y ? MMX_K_YPASS : MMX_K_GENERIC, s);
    if (fast_y) rc = mmx_launch_ypass(d_blocks, n_blocks, max_ycols, slot_elems, taps(wy0, wy2), radius, t0, t1, t2, t3, s);
    else rc = mmx_launch_generic_pass(1, vol, d_blocks, n_blocks, max_vox, slot_elems, wy0, wy2, radius, t0, t1, t2, t3, s); }
    if (rc != MMX_OK) return rc == MMX_ERR_HIP ? hip_fail(hipGetLastError(), "y pass") : rc;
    { mmx_timed_scope ts(fast_x ? MMX_K_XPASS : MMX_K_GENERIC, s);
    if (fast_x) rc = mmx_launch_xpass(d_blocks, n_blocks, max_rows, max_nx, slot_elems, taps(wx0, wx2), radius, t2, t3, d_log, s);
    else rc = mmx_launch_generic_pass(2, vol, d_blocks, n_blocks, max_vox, slot_elems, wx0, wx2, radius, t2, t3, d_log, nullptr, s); }
    if (rc != MMX_OK) return rc == MMX_ERR_HIP ? hip_fail(hipGetLastError(), "x pass") : rc;
    return MMX_OK;
}

// Same as mmx_log_batch_f32 but always through the generic kernels (tests cross-check the
// register-ring kernels against it; also what very large sigmas take).
int mmx_log_batch_f32_generic(const mmx_volume* vol, const mmx_block* d_blocks, const mmx_block* h_blocks,
                              int n_blocks, int64_t slot_elems,
                              const double* h_w0, const double* h_w2, int radius, double norm,
                              float* d_log, float* d_work, void* stream)
{
    if (!vol || !vol->d_data || !d_blocks || !h_blocks || !h_w0 || !h_w2 || !d_log || !d_work)
        return MMX_ERR_ARG;
    if (n_blocks < 1 || radius < 0 || radius > MMX_MAX_RADIUS_GENERIC || slot_elems < 1) return MMX_ERR_ARG;
    if (n_blocks > MMX_MAX_BLOCKS) return MMX_ERR_UNSUPPORTED;
    double in_scale = 1.0;
    if (vol->dtype == MMX_U8) in_scale = 1.0 / 255.0;
    else if (vol->dtype == MMX_U16) in_scale = 1.0 / 65535.0;
    int max_vox = 0;
    for (int i = 0; i < n_blocks; ++i) {
        const mmx_block& b = h_blocks[i];
        if (b.nz < 1 || b.ny < 1 || b.nx < 1 || b.slot != i) return MMX_ERR_ARG;
        if (b.px < b.nx || b.px % MMX_ROW_ALIGN) return MMX_ERR_ARG;
        if ((int64_t)b.nz * b.ny * b.px > slot_elems) return MMX_ERR_WORKSPACE;
        if (b.nz * b.ny * b.px > max_vox) max_vox = b.nz * b.ny * b.px;
    }
    const int64_t n_slots = n_blocks;
    float* t0 = d_work;
    float* t1 = d_work + n_slots * slot_elems;
    float* t2 = t1 + n_slots * slot_elems;
    float* t3 = t2 + n_slots * slot_elems;
    hipStream_t s = (hipStream_t)stream;
    float a[MMX_MAX_RADIUS_GENERIC + 1], b[MMX_MAX_RADIUS_GENERIC + 1];
    int rc;
    for (int k = 0; k <= radius; ++k) { a[k] = (float)(h_w0[k] * in_scale); b[k] = (float)(h_w2[k] * in_scale); }
    rc = mmx_launch_generic_pass(0, vol, d_blocks, n_blocks, max_vox, slot_elems, a, b, radius, nullptr, nullptr, t0, t1, s);
    if (rc != MMX_OK) return rc;
    for (int k = 0; k <= radius; ++k) { a[k] = (float)h_w0[k]; b[k] = (float)h_w2[k]; }
    rc = mmx_launch_generic_pass(1, vol, d_blocks, n_blocks, max_vox, slot_elems, a, b, radius, t0, t1, t2, t3, s);
    if (rc != MMX_OK) return rc;
    for (int k = 0; k <= radius; ++k) { a[k] = (float)(-norm * h_w0[k]); b[k] = (float)(-norm * h_w2[k]); }
    return mmx_launch_generic_pass(2, vol, d_blocks, n_blocks, max_vox, slot_elems, a, b, radius, t2, t3, d_log, nullptr, s);
}

int mmx_zx_pack(const mmx_volume* vol, const mmx_block* d_blocks, const mmx_block* h_blocks, int n_blocks,
                int64_t slot_elems, float* d_work, void* stream)
{
    if (!vol || !vol->d_data || !d_blocks || !h_blocks || !d_work || n_blocks < 1 || slot_elems < 1) return MMX_ERR_ARG;
    if (n_blocks > MMX_MAX_BLOCKS) return MMX_ERR_UNSUPPORTED;
    if (vol->dtype != MMX_U8 && vol->dtype != MMX_U16 && vol->dtype != MMX_F32) return MMX_ERR_UNSUPPORTED;
    for (int i = 0; i < n_blocks; ++i) {
        const mmx_block& b = h_blocks[i];
        if (b.nz < 1 || b.ny < 1 || b.nx < 1 || b.slot != i) return MMX_ERR_ARG;
        if (b.px < b.nx || b.px % MMX_ROW_ALIGN) return MMX_ERR_ARG;
    }
    mmx_zx6_plan plan;
    int rc = mmx_zx6_plan_make(h_blocks, n_blocks, slot_elems, vol->dtype, &plan);
    if (rc != MMX_OK) return rc;
    mmx_timed_scope ts(MMX_K_ZXPACK, (hipStream_t)stream);
    rc = mmx_launch_zx6_pack(vol, d_blocks, h_blocks, n_blocks, plan, d_work, (hipStream_t)stream);
    return rc == MMX_ERR_HIP ? hip_fail(hipGetLastError(), "voxel copy of the tiled path") : rc;
}

int mmx_peaks_batch(const float* d_log, const uint64_t* d_nms_mask, int mask_layout, int n_sigma, const mmx_block* d_blocks,
                    const mmx_block* h_blocks, int n_blocks, int64_t slot_elems,
                    float thr, float eps, mmx_cand* d_cands, uint32_t cap,
                    uint32_t* d_count, void* stream)
{
    if (!d_log || !d_blocks || !h_blocks || !d_cands || !d_count) return MMX_ERR_ARG;
    if (n_sigma < 1 || n_blocks < 1 || slot_elems < 1 || !(eps >= 0.f)) return MMX_ERR_ARG;
    if (n_blocks > MMX_MAX_BLOCKS) return MMX_ERR_UNSUPPORTED;
    if (slot_elems % MMX_ROW_ALIGN) return MMX_ERR_ARG;
    if (d_nms_mask && mask_layout != MMX_MASK_ROWS && mask_layout != MMX_MASK_QUADS) return MMX_ERR_ARG;
    int max_vox = 0;
    for (int i = 0; i < n_blocks; ++i) {
        const mmx_block& b = h_blocks[i];
        if (b.nz < 1 || b.ny < 1 || b.nx < 1 || b.slot != i) return MMX_ERR_ARG;
        if (b.px < b.nx || b.px % MMX_ROW_ALIGN) return MMX_ERR_ARG;
        if ((int64_t)b.nz * b.ny * b.px > slot_elems) return MMX_ERR_WORKSPACE;
        if (b.nz * b.ny * b.px > max_vox) max_vox = b.nz * b.ny * b.px;
    }
    mmx_timed_scope ts(MMX_K_PEAKS, (hipStream_t)stream);
    int rc;
    if (d_nms_mask)
        rc = mmx_launch_peaks_sparse(d_log, (const unsigned long long*)d_nms_mask, n_sigma,
                                     (int64_t)n_blocks * slot_elems, d_blocks, n_blocks, max_vox, slot_elems,
                                     thr, eps, d_cands, cap, d_count, mask_layout == MMX_MASK_QUADS, (hipStream_t)stream);
    else
        rc = mmx_launch_peaks(d_log, n_sigma, (int64_t)n_blocks * slot_elems, d_blocks, n_blocks, max_vox,
                              slot_elems, thr, eps, d_cands, cap, d_count, (hipStream_t)stream);
    return rc == MMX_ERR_HIP ? hip_fail(hipGetLastError(), "peaks") : rc;
}

// A rectangle of a host image into its place in the device copy: `height` rows of `width` bytes, the rows spitch /
// dpitch bytes apart (hipMemcpy2DAsync, host -> device, on `stream`; the host side pinned for the copy to be
// asynchronous).  What lets a host volume go up block row by block row -- the y-band of a z-range is `planes` rows of
// (band rows x row bytes) bytes, one plane pitch apart -- instead of whole z-slabs (volume._SlabUpload).
int mmx_copy_rect_h2d(void* d_dst, size_t dpitch, const void* h_src, size_t spitch, size_t width, size_t height,
                      void* stream)
{
    if (!d_dst || !h_src || width > dpitch || width > spitch) return MMX_ERR_ARG;
    if (!width || !height) return MMX_OK;
    hipError_t r = hipMemcpy2DAsync(d_dst, dpitch, h_src, spitch, width, height, hipMemcpyHostToDevice, (hipStream_t)stream);
    return r == hipSuccess ? MMX_OK : hip_fail(r, "hipMemcpy2DAsync");
}

int mmx_event_create(void** ev)
{
    if (!ev) return MMX_ERR_ARG;
    hipEvent_t e;
    hipError_t r = hipEventCreate(&e);
    if (r != hipSuccess) return hip_fail(r, "hipEventCreate");
    *ev = (void*)e;
    return MMX_OK;
}

int mmx_event_destroy(void* ev)
{
    hipError_t r = hipEventDestroy((hipEvent_t)ev);
    return r == hipSuccess ? MMX_OK : hip_fail(r, "hipEventDestroy");
}

int mmx_event_record(void* ev, void* stream)
{
    hipError_t r = hipEventRecord((hipEvent_t)ev, (hipStream_t)stream);
    return r == hipSuccess ? MMX_OK : hip_fail(r, "hipEventRecord");
}

int mmx_event_elapsed_ms(void* start, void* stop, float* ms)
{
    if (!ms) return MMX_ERR_ARG;
    hipError_t r = hipEventSynchronize((hipEvent_t)stop);
    if (r != hipSuccess) return hip_fail(r, "hipEventSynchronize");
    r = hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
    return r == hipSuccess ? MMX_OK : hip_fail(r, "hipEventElapsedTime");
}

}  // extern "C"
