// Shared declarations for the gfx950 kernels of libmmx_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mmx.h"

#define MMX_WG 256  // workgroup size used by the streaming kernels (4 waves of 64)
#define MMX_MAX_GRID_X 2147483647  // workgroups along x of one launch

// Half kernels (index k = distance from the centre tap) passed BY VALUE in the
// kernarg segment, so that with a fully unrolled tap loop every weight is a scalar
// (SGPR) operand of its v_fma.
struct mmx_taps_f32 {
    float w0[MMX_MAX_RADIUS_FAST + 1];  // order-0 Gaussian
    float w2[MMX_MAX_RADIUS_FAST + 1];  // order-2 (second derivative) Gaussian
};

// scipy "reflect" (half-sample symmetric): d c b a | a b c d | d c b a.
// Valid for any i (also |i| >> n), as scipy's NI_ExtendLine is.
__host__ __device__ __forceinline__ int mmx_reflect(int i, int n)
{
    if (n == 1) return 0;
    const int period = 2 * n;
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - 1 - i;
}

// internal launchers (defined one per translation unit, called by mmx_api.hip)
int mmx_launch_zpass(const mmx_volume* vol, const mmx_block* d_blocks, int n_blocks, int max_cols,
                     int64_t slot_elems, const mmx_taps_f32& taps, int radius,
                     float* d_gz, float* d_gzz, hipStream_t stream);
int mmx_launch_ypass(const mmx_block* d_blocks, int n_blocks, int max_cols, int64_t slot_elems,
                     const mmx_taps_f32& taps, int radius, const float* d_gz, const float* d_gzz,
                     float* d_a, float* d_bc, hipStream_t stream);
int mmx_launch_xpass(const mmx_block* d_blocks, int n_blocks, int max_rows, int max_nx,
                     int64_t slot_elems, const mmx_taps_f32& taps, int radius,
                     const float* d_a, const float* d_bc, float* d_log, hipStream_t stream);

int mmx_launch_zx2(const mmx_volume* vol, const mmx_block* d_blocks, int n_blocks, int max_ny, int max_px,
                   int64_t slot_elems, const mmx_taps_f32& tz, const mmx_taps_f32& tx, int radius,
                   float* d_p, float* d_q, hipStream_t s);
// Tiled fused path (zx_mode 6): where its pieces live inside the four intermediate arrays of d_work
// (4 n_blocks slot_elems floats).  P and Q as 16 x 16 tiles (mmx_fused4.hip: zx4_kernel), the Toeplitz
// fragment tables of the current sigma, and the operand-ordered copy of the blocks' voxels (zx6_pack_kernel),
// which survives from one sigma of a batch to the next.
struct mmx_zx6_plan {
    int64_t tile_stride;    // floats per block of tiled P (and of Q): max over blocks of ntx ntz ny 256
    int64_t pack_stride;    // uint16 elements per block of packed voxels: max over blocks of ny ntz nch8 128
    int64_t q_off, tab_off, pack_off;      // byte offsets into d_work (P at 0)
    int64_t tab_bytes;
    int max_tiles;          // max ntx ntz
    int max_rowtiles;       // max ny ntz
};
int mmx_zx6_plan_make(const mmx_block* h_blocks, int n_blocks, int64_t slot_elems, int voxel_dtype, mmx_zx6_plan* plan);
int mmx_launch_zx6_pack(const mmx_volume* vol, const mmx_block* d_blocks, const mmx_block* h_blocks, int n_blocks,
                        const mmx_zx6_plan& plan, void* d_work, hipStream_t stream);
int mmx_launch_zx6(const mmx_volume* vol, const mmx_block* d_blocks, const mmx_block* h_blocks, int n_blocks,
                   const mmx_zx6_plan& plan, const mmx_taps_f32& tx, int radius, void* d_work, float qp, float qq,
                   hipStream_t stream);
// cp, cq > 0: d_p holds Q16 tiles, P = unorm16 * cp, Q = snorm16 * cq (d_q unused); 0: float32 tiles
int mmx_launch_y6(const mmx_block* d_blocks, int n_blocks, const mmx_zx6_plan& plan, int64_t slot_elems,
                  const mmx_taps_f32& taps, int radius, const float* d_p, const float* d_q, float cp, float cq,
                  float* d_log, unsigned long long* d_mask, float nms_lo, float nms_eps, hipStream_t stream);
// the same on the matrix cores (mmx_ymfma.hip): 16-bit tiles only, radius <= 24; MMX_ERR_UNSUPPORTED otherwise
int mmx_launch_ym(const mmx_block* d_blocks, int n_blocks, const mmx_zx6_plan& plan, int64_t slot_elems,
                  const mmx_taps_f32& taps, int radius, const float* d_p, float cp, float cq,
                  float* d_log, unsigned long long* d_mask, float nms_lo, float nms_eps, hipStream_t stream);
int mmx_launch_y2(const mmx_block* d_blocks, int n_blocks, int max_cols, int64_t slot_elems,
                  const mmx_taps_f32& taps, int radius, const float* d_p, const float* d_q,
                  float* d_log, unsigned long long* d_mask, float nms_lo, float nms_eps, hipStream_t stream);

// ---- optional per-kernel-family timing with HIP events on the launch stream (bench.py) ----
enum mmx_kernel_kind {
    MMX_K_ZPASS = 0, MMX_K_YPASS, MMX_K_XPASS, MMX_K_GENERIC, MMX_K_PEAKS, MMX_K_RESCORE,
    MMX_K_PAIRS, MMX_K_CLOSE, MMX_K_ZX, MMX_K_Y2, MMX_K_PREPROC, MMX_K_COLOC, MMX_K_ZXPACK, MMX_K_END
};
void mmx_time_begin(int kind, hipStream_t s);
void mmx_time_end(int kind, hipStream_t s);
static_assert(MMX_K_END == MMX_K_COUNT, "kernel kinds out of sync with mmx.h");
struct mmx_timed_scope {
    int kind; hipStream_t s;
    mmx_timed_scope(int k, hipStream_t st) : kind(k), s(st) { mmx_time_begin(k, st); }
    ~mmx_timed_scope() { mmx_time_end(kind, s); }
};
