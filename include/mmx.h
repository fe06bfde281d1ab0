/* mmx.h -- C ABI of the MI355X-native blob-detection hot path (libmmx_hip.so).
 *
 * Drop-in boundary for ONE path of MagellanMapper: whole-volume nuclei detection,
 * i.e. what `magmap.cv.detector.detect_blobs` (reference magmap/cv/detector.py:874-957)
 * gets from its single third-party call `skimage.feature.blob_log`
 * (detector.py:931-933) plus the cross-block duplicate search of
 * `detector.remove_close_blobs` (detector.py:1000-1085).  The reference has no
 * FFI of its own (it is pure Python over SciPy's `_nd_image` C extension); the
 * entry points below are what a ctypes binding on the reference side binds --
 * see INTEGRATION.md for the stub.
 *
 * Conventions
 *   - plain C, caller-owned buffers, no exceptions: every call returns an
 *     `mmx_status` (0 = OK); `mmx_strerror` names it.
 *   - pointers prefixed d_ are DEVICE pointers (HBM), h_ are HOST pointers.
 *   - `stream` is a `hipStream_t` passed as `void*`; all device work is enqueued
 *     on it and NOT synchronised -- the caller owns ordering.
 *   - arrays are C-ordered (z, y, x), x contiguous ("z-major").
 *   - a *block* is one sub-ROI of `chunking.stack_splitter`
 *     (reference magmap/cv/chunking.py:214-256).  Every block is filtered on its
 *     own extent with SciPy "reflect" boundaries -- exactly what each reference
 *     worker sees (magmap/cv/stack_detect.py:79, 242).  A *batch* is a set of
 *     blocks processed by one launch sequence; block i of a batch owns slot i of
 *     every workspace array (`slot_elems` floats, [nz][ny][px], px = row pitch).
 */
#ifndef MMX_H
#define MMX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMX_ABI_VERSION 16

typedef enum {
    MMX_OK = 0,
    MMX_ERR_ARG = 1,        /* bad argument (null pointer, size, dtype, radius)        */
    MMX_ERR_HIP = 2,        /* a HIP runtime call failed; see mmx_last_hip_error()      */
    MMX_ERR_NO_DEVICE = 3,  /* no gfx950 device visible                                 */
    MMX_ERR_WORKSPACE = 4,  /* workspace too small                                      */
    MMX_ERR_UNSUPPORTED = 5,/* e.g. kernel radius above MMX_MAX_RADIUS_GENERIC          */
    MMX_DEFERRED = 6        /* not an error: mmx_host_finish_stack leaves a decision to the caller */
} mmx_status;

/* input voxel types (reference accepts any dtype through skimage.img_as_float,
 * skimage/util/dtype.py:310-328) */
typedef enum { MMX_U8 = 0, MMX_U16 = 1, MMX_F32 = 2, MMX_F64 = 3 } mmx_dtype;

/* largest kernel radius with a register-resident (fully unrolled) column pass;
 * larger radii take the generic path.  radius = int(4*sigma + 0.5)
 * (scipy/ndimage/_filters.py:313-315). */
#define MMX_MAX_RADIUS_FAST 24
#define MMX_MAX_RADIUS_GENERIC 255
/* most blocks one call takes (a launch puts the block on grid.y): MMX_ERR_UNSUPPORTED beyond; callers split batches */
#define MMX_MAX_BLOCKS 65535
/* voxels of ball(2), the neighbourhood a blob can own in the intensity co-localisation */
#define MMX_COLOC_BALL 33

/* One block of a batch (device array of these is passed to the kernels). */
typedef struct {
    int64_t src_off;   /* element offset of the block origin inside the source volume */
    int32_t nz, ny, nx;/* block extent in voxels                                       */
    int32_t slot;      /* workspace slot (0 .. n_blocks-1)                             */
    int32_t px;        /* row pitch of the block's workspace arrays, in floats:
                          a multiple of MMX_ROW_ALIGN >= nx (128-byte aligned rows and
                          planes: every wave-level store is whole cache lines)         */
    int32_t _pad;
} mmx_block;
#define MMX_ROW_ALIGN 32

/* Source volume view (one channel): element strides, x stride must be 1 for
 * integer inputs laid out (z,y,x); a (z,y,x,c) image passes stride_x = n_channels. */
typedef struct {
    const void* d_data;
    int32_t dtype;     /* mmx_dtype */
    float value_range; /* float voxels only, what the caller knows about them (integer voxels: ignored, their range is
                          the type's): > 0: every voxel lies in [0, value_range]; < 0: |voxel| <= -value_range;
                          0: unknown.  A known range lets MMX_ZX_AUTO take the tiled matrix-core path for float
                          voxels (their float16 pieces need |v| < 65504), a non-negative one also its 16-bit tiles. */
    int64_t stride_z, stride_y, stride_x; /* in elements */
} mmx_volume;

/* Scale-space candidate (A4).  48 bytes.  `flags` bit 0: contested -- within eps of
 * a neighbour or of the threshold, the float64 value must decide.  Bit 1 (MMX_CAND_BAND, set by the sparse NMS):
 * `band` bit j / `flags` bit 16 + (j - 64) tells whether neighbour j of the 80 (C order of (ds, dz, dy, dx) over
 * {-1, 0, 1}^4, the centre left out) has a float32 value within eps below the candidate's or above it -- the
 * only neighbours whose exact values can decide a contested candidate. */
typedef struct {
    int32_t slot;      /* block slot in the batch          */
    int32_t s;         /* sigma index                      */
    int32_t z, y, x;   /* block-relative voxel             */
    uint32_t flags;
    float v;           /* float32 scale-normalised -LoG    */
    float nbr_max;     /* float32 max over the 80 neighbours, 0-padded outside the cube */
    double v64;        /* exact float64 value (filled by mmx_rescore_f64), NaN before   */
    uint64_t band;     /* with MMX_CAND_BAND: neighbours 0..63 in the band (64..79: flags bits 16..31) */
} mmx_cand;

#define MMX_CAND_CONTESTED 1u
#define MMX_CAND_BAND 2u
/* an entry appended by mmx_expand_probes: a neighbour of the contested candidate `band` (its table index) */
#define MMX_CAND_PROBE 4u

/* ---- library / device ---------------------------------------------------- */
int mmx_abi_version(void);
const char* mmx_strerror(int status);
const char* mmx_last_hip_error(void);
/* number of visible devices whose arch is gfx950; <0 on HIP error */
int mmx_device_count(void);

/* ---- A0 + A2 + A3: scale-normalised -LoG of every block of a batch, one sigma
 * replaces: skimage.feature.blob_log's
 *     -gaussian_laplace(img_as_float(image), sigma) * sigma**2
 * (skimage/feature/blob.py:470, 501-502 -> scipy/ndimage/_filters.py:644-707),
 * computed in float32 (3 separable passes, shared order-0 passes).
 *   h_w0, h_w2 : float64 half kernels, index k = 0..radius (k = distance from centre),
 *                computed by the caller exactly as scipy's _gaussian_kernel1d does
 *   norm       : mean(sigma)**2
 *   d_log      : out, [n_blocks][slot_elems] float32
 *   d_work     : scratch, 4 * n_blocks * slot_elems float32
 *   d_nms_mask : optional out, (n_blocks * slot_elems) >> 5 entries of two uint64 (16-byte aligned): the block
 *                in slot b starts at entry (b * slot_elems) >> 5; entry (c >> 6) + y * ceil(nz*px/64), bit
 *                c & 63, c = z*px + x.  Word 0: the response exceeds nms_lo and no y / x neighbour exceeds it by
 *                more than nms_eps -- a superset of the local maxima, the only voxels mmx_peaks_batch visits.
 *                Word 1: the response exceeds nms_lo.  WITH A MASK, 64-VOXEL SEGMENTS WHOSE WORD 1 IS ZERO ARE
 *                NOT WRITTEN TO d_log (a response below the threshold can neither be a peak nor out-vote
 *                one): d_log is then only meaningful together with the entries.  Only the fused path produces
 *                the entries, and only when every block's rows fit its share (tiny blocks do not):
 *                *h_mask_written (host) says whether this call did (0: no, d_log is complete) and in which
 *                layout: MMX_MASK_ROWS (1) as above; MMX_MASK_QUADS (2, left by MMX_ZX_TILED): per row y one entry
 *                per 4 planes x 16 columns, entry y * ceil(nz/4) * ceil(nx/16) + (z >> 2) * ceil(nx/16) + (x >> 4),
 *                bit ((z & 3) << 4) | (x & 15), the unwritten segments of d_log being those 64 voxels.  Pass the
 *                value on to mmx_peaks_batch (every sigma of a batch must have produced the same layout).
 *   zx_mode    : how the Z and X passes run (a per-call argument: the library keeps no mode).
 *                MMX_ZX_AUTO (default): the fastest kernel that takes the geometry (MMX_ZX_TILED for integer
 *                voxels, else MMX_ZX_PACKED, else the separate passes); the others exist for cross-checks and
 *                measurements.  Float voxels take the tiled path when the volume states its value range
 *                (mmx_volume.value_range) or when MMX_ZX_TILED is asked for by name: its copy holds every voxel as
 *                two float16 pieces (22 significant bits, like the weights), which covers |v| < 65504.  All agree within float32 rounding, and the peak decisions are taken on exact
 *                float64 values either way (mmx_rescore_f64).  With entries requested and nms_eps at least four times
 *                mmx_tiled_q16_error_bound(), AUTO hands the intermediates over as 16-bit fixed point
 *                (MMX_ZX_TILED_Q16).  MMX_ZX_TILED works from an operand-ordered copy of
 *                the blocks' voxels inside d_work, which does not depend on sigma: mmx_zx_pack makes it once per
 *                batch, and MMX_ZX_TILED | MMX_ZX_PREPACKED then skips making it again for every sigma.
 *   h_zx_path  : optional out (host): the MMX_ZX_* kernel this call actually ran (MMX_ZX_SEPARATE when the
 *                geometry fell back to the three separate passes)                                          */
typedef enum {
    MMX_ZX_AUTO = -1,
    MMX_ZX_SEPARATE = 0,  /* three separate passes (register-ring column kernels + LDS row kernel)          */
    MMX_ZX_PACKED = 2,    /* zx2_kernel: fused Z+X, wave-specialised, packed float32 VALU math              */
    /* 1, 3, 4, 5: retired (the matrix-core experiments that led to the tiled form; measurements in
       profiles/HISTORY.md): MMX_ERR_ARG */
    MMX_ZX_TILED = 6,     /* zx4's arithmetic on an operand-ordered copy of the voxels (zx6_pack_kernel), P / Q
                             handed to the Y pass (y6_kernel) as 16 x 16 tiles: every access one contiguous KiB */
    MMX_ZX_TILED_Q16 = 7  /* the same with the tiles as 16-bit fixed point (half the intermediate bytes): the LoG
                             values carry a rounding error of at most mmx_tiled_q16_error_bound(); AUTO picks it
                             only when entries are asked for with nms_eps >= 4 x that bound                     */
} mmx_zx_mode;
#define MMX_ZX_PREPACKED 0x100   /* or-ed into MMX_ZX_TILED / MMX_ZX_TILED_Q16: mmx_zx_pack ran on this d_work for these blocks */
#define MMX_ZX_Y_VALU    0x200   /* or-ed into MMX_ZX_TILED_Q16 (any zx_mode >= 0 accepts it): the Y pass of the 16-bit tiles on the
                                    VALU (y6_kernel) instead of the matrix cores (ym_kernel, the default); for cross-checks */
#define MMX_MASK_ROWS 1
#define MMX_MASK_QUADS 2
int mmx_log_batch_f32(const mmx_volume* vol, const mmx_block* d_blocks, const mmx_block* h_blocks,
                      int n_blocks, int64_t slot_elems,
                      const double* h_w0, const double* h_w2, int radius, double norm,
                      float* d_log, float* d_work, uint64_t* d_nms_mask, float nms_lo, float nms_eps,
                      int* h_mask_written, int zx_mode, int* h_zx_path, void* stream);

/* The LoG contract: nominated (float32 / 16-bit) cube values stay within this ABSOLUTE distance of the reference's
 * float64 values.  Under MMX_ZX_AUTO 16-bit intermediates are chosen only when their error bound in value units
 * (mmx_tiled_q16_error_bound x the stated value range) is below it AND the caller's NMS band covers it fourfold;
 * MMX_ZX_TILED_Q16 by name takes them regardless (tests, experiments). */
#define MMX_LOG_ABS_TOL 1e-4

/* Largest deviation of an MMX_ZX_TILED_Q16 LoG value from the float32 paths' (which are within a few 1e-7 of the
 * exact value), for voxels in [0, 1] (uint8 / uint16 after img_as_float; float voxels in [0, m]: times m): a
 * function of the weights alone (5.1e-5 for any sigma >= 1).  A true maximum is nominated as long as the NMS band is four
 * times this (mmx_rescore_f64 then decides on exact values as always).  < 0 on bad arguments. */
double mmx_tiled_q16_error_bound(const double* h_w0, const double* h_w2, int radius, double norm);

/* The sigma-independent part of MMX_ZX_TILED: the operand-ordered copy of the blocks' voxels, written into the
 * part of d_work (same pointer, blocks and slot_elems as the mmx_log_batch_f32 calls that follow) that the tiled
 * path leaves alone.  MMX_ERR_UNSUPPORTED (nothing written) for float64 voxels or when the pieces do not fit
 * d_work: call mmx_log_batch_f32 without MMX_ZX_PREPACKED then.  float32 voxels are split into float16 pieces here
 * (see zx_mode above for the value range this suits). */
int mmx_zx_pack(const mmx_volume* vol, const mmx_block* d_blocks, const mmx_block* h_blocks, int n_blocks,
                int64_t slot_elems, float* d_work, void* stream);

/* Bytes of device workspace one batch needs for `n_sigma` scales: the 4 intermediate arrays of
 * mmx_log_batch_f32 (d_work), one d_log array per scale and, with `with_masks`, the NMS entries of every scale
 * (d_nms_mask, 16-byte aligned).  slot_elems = the largest nz * ny * px of the batch (px = nx rounded up to
 * MMX_ROW_ALIGN).  Layout used by the Python host code: [d_work | d_log x n_sigma | pad to 16 | masks]. */
size_t mmx_workspace_bytes(int n_blocks, int64_t slot_elems, int n_sigma, int with_masks);

/* Same contract, always through the generic (any radius <= MMX_MAX_RADIUS_GENERIC, any block
 * extent) kernels.  mmx_log_batch_f32 picks per pass between the register-ring kernels and
 * these; this entry exists so that tests can cross-check the two paths. */
int mmx_log_batch_f32_generic(const mmx_volume* vol, const mmx_block* d_blocks,
                              const mmx_block* h_blocks, int n_blocks, int64_t slot_elems,
                              const double* h_w0, const double* h_w2, int radius, double norm,
                              float* d_log, float* d_work, void* stream);

/* ---- A4: 3x3x3x3 local maxima over (z,y,x,sigma) strictly above the threshold
 * replaces: skimage.feature.peak_local_max(footprint=ones(3,3,3,3), mode='constant')
 * (skimage/feature/peak.py:28-50, 114-319).
 *   d_log        : [n_sigma][n_blocks][slot_elems] float32 (sigma-major)
 *   d_nms_mask   : optional [n_sigma][(n_blocks * slot_elems) >> 5] 16-byte entries written by mmx_log_batch_f32 with
 *                  nms_lo = thr - eps and nms_eps = eps for EVERY sigma (NULL = read every voxel)
 *   mask_layout  : MMX_MASK_ROWS / MMX_MASK_QUADS as reported by those calls (ignored without d_nms_mask)
 *   eps          : candidates are emitted when v >= nbr_max - eps and v > thr - eps
 *   d_cands/cap  : output table; *d_count keeps counting past cap (caller retries)  */
int mmx_peaks_batch(const float* d_log, const uint64_t* d_nms_mask, int mask_layout, int n_sigma, const mmx_block* d_blocks,
                    const mmx_block* h_blocks, int n_blocks, int64_t slot_elems,
                    float thr, float eps, mmx_cand* d_cands, uint32_t cap,
                    uint32_t* d_count, void* stream);

/* ---- exact float64 value of the cube at given points (bit-for-bit the reference's
 * arithmetic: scipy NI_Correlate1D operation order, no FMA contraction)
 * replaces: the float64 cube values that peak_local_max compares
 *   d_pts[i].{slot,s,z,y,x} in, d_pts[i].v64 out; n points are taken from
 *   *d_count (clamped to cap) when d_count != NULL, else n = cap.
 *   d_w0/d_w2 : [n_sigma][MMX_MAX_RADIUS_GENERIC+1] float64 half kernels (device)
 *   h_radius  : [n_sigma] kernel radii, h_norm: [n_sigma] mean(sigma)**2
 *   store_f32 : 1 = the input is float32 and SciPy keeps float32 intermediates      */
int mmx_rescore_f64(const mmx_volume* vol, const mmx_block* d_blocks, int n_blocks,
                    mmx_cand* d_pts, uint32_t cap, const uint32_t* d_count,
                    const double* d_w0, const double* d_w2, const int32_t* h_radius,
                    const double* h_norm, int n_sigma, int store_f32, void* stream);

/* ---- the neighbours that can out-vote a contested candidate, appended to the candidate table (A4)
 * replaces: the exact `image == maximum_filter(image)` comparison of peak_local_max (skimage/feature/peak.py:35-49)
 * for the candidates float32 cannot settle: after this call and one mmx_rescore_f64 over the whole table the host
 * has every float64 value those decisions need.
 *   d_cands/cap/d_count : the table mmx_peaks_batch filled; *d_count keeps counting past cap
 *   d_n_cands           : out, *d_count as it was before this call (entries below it are candidates, the rest
 *                         probes: flags = MMX_CAND_PROBE, slot/s/z/y/x of the neighbour, band = index of its candidate)
 * For a candidate flagged MMX_CAND_BAND only the neighbours in its band are appended, otherwise all (<= 80)
 * neighbours inside the cube. */
int mmx_expand_probes(mmx_cand* d_cands, uint32_t cap, uint32_t* d_count, uint32_t* d_n_cands,
                      const mmx_block* d_blocks, int n_blocks, int n_sigma, void* stream);

/* ---- A0-A4 of one batch of blocks in ONE call (the fused `mmx_detect_block`)
 * replaces: everything between the reference's call `blob_log(roi, min_sigma, max_sigma, num_sigma, threshold, overlap)`
 * (magmap/cv/detector.py:931-933 -> skimage/feature/blob.py:470-504, peak.py:28-50) and the comparison of float64 cube
 * values: the voxel copy of the tiled path, (Z+X, Y) per sigma -- with the rules that keep all scales on one kernel path
 * and one NMS entry layout --, the counter reset, mmx_peaks_batch, mmx_expand_probes, mmx_rescore_f64 and the copies of
 * the counters and of the head of the candidate table to pinned host memory, all enqueued by native code.  Nothing
 * waits for the GPU; the caller synchronises on `ev_done` and then owns h_count / h_cands.
 *   vol32 / vol_exact : what the float32 passes read (uint8 / uint16 / float32; float voxels state their range in
 *                 value_range) and what the exact re-score reads (the same volume, or its float64 original)
 *   h_w0, h_w2 / d_w0, d_w2 : [n_sigma][MMX_MAX_RADIUS_GENERIC + 1] float64 half kernels on the host and on the device
 *   d_work / work_bytes : >= mmx_workspace_bytes(n_blocks, slot_elems, n_sigma, 1)
 *   thr, eps    : threshold and nomination band (candidates: v >= nbr_max - eps and v > thr - eps)
 *   d_cands / cap / d_count : candidate table; d_count[0] = entries (keeps counting past cap: the caller retries with
 *                 a larger table), d_count[1] = candidates among them (the rest are probes)
 *   h_count, h_cands / h_prefix : pinned host copies of d_count[0..1] and of the first h_prefix entries (NULL: no copy)
 *   zx_mode / zx_flags : as for mmx_log_batch_f32 (flags: MMX_ZX_Y_VALU)
 *   exact       : re-score every candidate (and probe) in float64;  expand : append the probes (mmx_expand_probes)
 *   stream      : the LoG passes;  tail_stream (NULL = stream): NMS, probes, re-score, copies -- beside the next batch's
 *                 passes when it is another stream;  pack_stream (NULL = stream): the voxel copy
 *   ev_work_free: NULL or an event to wait for before d_work is written (its previous reader);
 *   ev_work_read: NULL or an event recorded once the NMS has read d_work;  ev_done: NULL or recorded at the very end. */
typedef struct {
    const mmx_volume* vol32;
    const mmx_volume* vol_exact;
    const mmx_block* d_blocks;
    const mmx_block* h_blocks;
    int32_t n_blocks, n_sigma;
    int64_t slot_elems;
    const double* h_w0; const double* h_w2;
    const double* d_w0; const double* d_w2;
    const int32_t* h_radius;
    const double* h_norm;
    float* d_work;
    size_t work_bytes;
    float thr, eps;
    mmx_cand* d_cands;
    uint32_t cap, h_prefix;
    uint32_t* d_count;
    uint32_t* h_count;
    mmx_cand* h_cands;
    int32_t zx_mode, zx_flags, store_f32, exact, expand, _pad;
    void* stream; void* tail_stream; void* pack_stream;
    void* ev_work_free; void* ev_work_read; void* ev_done;
} mmx_detect_args;
typedef struct {
    int32_t zx_path;        /* mmx_zx_mode the last scale ran */
    int32_t mask_layout;    /* 0 = the NMS read the full cube, MMX_MASK_ROWS / MMX_MASK_QUADS */
    int32_t n_pass_rounds;  /* 1, or more when the scales had to be computed again on another path */
    int32_t _pad;
    double q16_bound;       /* error bound of the 16-bit intermediates in value units (0: not used) */
} mmx_detect_info;
int mmx_detect_batch(const mmx_detect_args* args, mmx_detect_info* info);
const char* mmx_detect_last_error(void);
/* the same launches captured as a hipGraph (every argument frozen; refused with MMX_ERR_UNSUPPORTED while per-kernel
 * timing is on: its events cannot live inside a capture) and replayed with one launch on `stream` */
int mmx_detect_batch_capture(const mmx_detect_args* args, mmx_detect_info* info, void** graph);
int mmx_graph_launch(void* graph, void* stream, void* ev_done, mmx_detect_info* info);
int mmx_graph_destroy(void* graph);
int mmx_event_synchronize(void* ev);
int mmx_stream_wait_event(void* stream, void* ev);
int mmx_timing_is_enabled(void);

/* ---- A5 support: all blob pairs of one block whose sphere-overlap fraction exceeds
 * `overlap` (skimage/feature/blob.py:84-187: _blob_overlap / _prune_blobs)
 *   d_blobs : [n][4] float64 (z, y, x, sigma), blocks delimited by d_offsets[n_blocks+1]
 *   max_sigma: largest sigma in the table (pairs farther apart than 2*sqrt(3)*max_sigma are skipped)
 *   d_pairs : out [cap][2] int32 global row indices (i < j), *d_count total found
 *   d_frac  : out [cap] float64 overlap fraction                                   */
int mmx_overlap_pairs(const double* d_blobs, const int32_t* d_offsets, int n_blocks,
                      double overlap, double band, double max_sigma, int32_t* d_pairs, double* d_frac,
                      uint32_t cap, uint32_t* d_count, void* stream);

/* ---- A13 support: for every master row the LAST check row within `tol` on all
 * three axes, and for every check row whether any master row matched
 * (magmap/cv/detector.py:1000-1085: _find_close_blobs / remove_close_blobs)
 *   d_master : [n_master][3] int32, d_check : [n_check][3] int32
 *   d_last   : out [n_master] int32, -1 = no match
 *   d_hit    : out [n_check] uint8                                               */
int mmx_close_pairs(const int32_t* d_master, int n_master, const int32_t* d_check, int n_check,
                    const int32_t tol[3], int32_t* d_last, uint8_t* d_hit, void* stream);

/* ---- P1-P3: per-block preprocessing ahead of detection
 * replaces: the sub-sub-block loop of StackDetector.detect_sub_roi (magmap/cv/stack_detect.py:122-150):
 *     plot_3d.saturate_roi (magmap/plot/plot_3d.py:55-112) -- np.percentile contrast stretch
 *     plot_3d.denoise_roi  (plot_3d.py:115-172) -- clip, unsharp mask with
 *         skimage.filters.gaussian(sigma 8, 'nearest', truncate 4), erosion(octahedron(1)) when
 *         the sub-block mean exceeds erosion_threshold
 * for uint8 / uint16 voxels, in float64, bit for bit (same IEEE operations in NumPy's / SciPy's
 * order).  One *sub-block* is one tile of chunking.stack_splitter(block.shape, denoise_max_shape)
 * of one block and channel; it is processed as an independent image. */
typedef struct {
    int64_t src_off;      /* element offset of the sub-block origin inside the source volume        */
    int64_t dst_off;      /* element offset of the sub-block origin inside d_out32 / d_out64        */
    int64_t scratch_off;  /* generic entry only: offset (in doubles) of 2*nz*ny*nx doubles of scratch */
    int32_t nz, ny, nx;   /* extent in voxels                                                       */
    int32_t qclass;       /* row of the quantile-class table                                        */
} mmx_subblock;           /* 40 bytes */

/* What np.percentile(a, (clip_vmin, clip_vmax)) needs for an `a` of n values, computed by the caller
 * exactly as NumPy does (numpy/lib/_function_base_impl.py: _quantile, method "linear"):
 * virtual index (n-1)*q, its floor / floor+1 as 0-based ranks into sorted(a) (both n-1 when the
 * index is >= n-1), gamma = virtual - floor. */
typedef struct {
    int32_t lo_prev, lo_next, hi_prev, hi_next;
    double lo_gamma, hi_gamma;
} mmx_quantile_class;     /* 32 bytes */

typedef struct {
    double clip_min, clip_max;   /* profile "clip_min"/"clip_max" (np.clip bounds after stretching)  */
    double max_thresh;           /* config.near_max[channel] * profile "max_thresh_factor"           */
    double unsharp_strength;     /* 0 = no unsharp mask (Python falsy)                               */
    double erosion_threshold;    /* 0 = never erode (Python falsy)                                   */
    int32_t radius;              /* Gaussian radius; must be 32 = int(4*8+0.5): sigma 8 is hard-coded
                                    in the reference (plot_3d.py:151)                               */
    int32_t rgb_guess;           /* 1 = scikit-image < 0.19: an array whose LAST axis has length 3 is
                                    taken for RGB and not blurred along it (filters/_gaussian.py)   */
    double tv_weight;            /* profile tot_var_denoise as a float (True = 1.0); 0 = off.  Total-variation
                                    denoising of the clipped tile (plot_3d.py:147-149 ->
                                    skimage.restoration.denoise_tv_chambolle): only through
                                    mmx_preprocess_batch_generic, 7 doubles of scratch per voxel            */
    double tv_factor;            /* (1 / 6) / weight, as the host's float division gives it               */
} mmx_preproc_params;     /* 64 bytes */

#define MMX_PP_IDENTITY 1    /* vmin == vmax: voxels pass through unstretched                        */
#define MMX_PP_ERODED 2      /* mean > erosion_threshold                                             */
#define MMX_PP_EXACT_MEAN 4  /* the mean was re-summed in NumPy's pairwise order (knife edge)         */
typedef struct {
    double vmin, vmax, mean;     /* percentiles (vmax after the near_max floor), np.mean(saturated)  */
    int32_t flags;
    int32_t _pad;
} mmx_subblock_info;      /* 32 bytes */

#define MMX_PP_RADIUS 32
#define MMX_PP_MAX_SIDE 32          /* register-resident lines of the fast kernel                   */
#define MMX_PP_MAX_LDS 163840       /* one workgroup may hold a whole sub-block in LDS              */

/* LDS bytes the fast kernel needs for a sub-block, 0 if it does not qualify (a side above
 * MMX_PP_MAX_SIDE or more than MMX_PP_MAX_LDS bytes). */
int64_t mmx_preprocess_fast_lds(int nz, int ny, int nx);

/* Fast entry: every sub-block must qualify (mmx_preprocess_fast_lds != 0), else MMX_ERR_UNSUPPORTED.
 *   h_subs      : host copy of d_subs (validation, launch geometry)
 *   d_weights   : DEVICE float64 half kernel, index k = 0..radius, as scipy's _gaussian_kernel1d(8, 0, 32)
 *   dst_sy/sz   : row / plane strides (elements) of the two outputs; x stride is 1
 *   d_out32/64  : float32 copy (feeds mmx_log_batch_f32) and exact float64 result (feeds
 *                 mmx_rescore_f64); only the voxels of the given sub-blocks are written
 *   d_info      : optional [n_subs] diagnostics                                                    */
int mmx_preprocess_batch(const mmx_volume* vol, const mmx_subblock* d_subs, const mmx_subblock* h_subs,
                         int n_subs, const mmx_quantile_class* d_qclasses, int n_qclasses,
                         const mmx_preproc_params* params, const double* d_weights,
                         int64_t dst_sy, int64_t dst_sz, float* d_out32, double* d_out64,
                         mmx_subblock_info* d_info, void* stream);

/* The same with the kernel choice and the workspace in the caller's hands (tests cross-check the forms; tools time them):
 *   MMX_PP_AUTO      : what mmx_preprocess_batch does -- tiles of > 4 096 voxels take the pipelined form
 *                      (mmx_preproc_pipe.hip: tile-major voxel copy, statistics kernel, blur kernel that walks runs
 *                      of tiles), smaller ones the one-kernel-per-tile form
 *   MMX_PP_SINGLE    : one kernel per tile whatever its size (pp_fast_kernel)
 *   MMX_PP_PIPELINED : the pipelined form whatever the size (MMX_ERR_UNSUPPORTED if a tile does not fit)
 *   tiles_per_wg     : tiles one workgroup of the blur kernel walks (0 = MMX_PP_TILES_PER_WG)
 *   d_work           : mmx_preprocess_work_bytes(h_subs, n_subs) bytes of device scratch for the pipelined form
 *                      (2 bytes per voxel + 12 per tile), or NULL: taken from the stream's memory pool for the call;
 *                      likewise d_info may be NULL in every mode.  MMX_ERR_WORKSPACE when work_bytes is too small.
 * Results are bit-identical across modes.                                                                     */
#define MMX_PP_AUTO 0
#define MMX_PP_SINGLE 1
#define MMX_PP_PIPELINED 2
#define MMX_PP_TILES_PER_WG 8
int64_t mmx_preprocess_work_bytes(const mmx_subblock* h_subs, int n_subs);
int mmx_preprocess_batch_mode(const mmx_volume* vol, const mmx_subblock* d_subs, const mmx_subblock* h_subs,
                              int n_subs, const mmx_quantile_class* d_qclasses, int n_qclasses,
                              const mmx_preproc_params* params, const double* d_weights,
                              int64_t dst_sy, int64_t dst_sz, float* d_out32, double* d_out64,
                              mmx_subblock_info* d_info, int mode, int tiles_per_wg,
                              void* d_work, int64_t work_bytes, void* stream);

/* Same contract for any extent: data in d_scratch (each sub-block owns 2*nz*ny*nx doubles at its
 * scratch_off -- 7*nz*ny*nx with params->tv_weight set), one output per lane and pass. */
int mmx_preprocess_batch_generic(const mmx_volume* vol, const mmx_subblock* d_subs,
                                 const mmx_subblock* h_subs, int n_subs,
                                 const mmx_quantile_class* d_qclasses, int n_qclasses,
                                 const mmx_preproc_params* params, const double* d_weights,
                                 int64_t dst_sy, int64_t dst_sz, float* d_out32, double* d_out64,
                                 mmx_subblock_info* d_info, double* d_scratch, int64_t scratch_doubles,
                                 void* stream);

/* ---- C1: intensity co-localisation: for every blob the mean of ONE image
 * channel over the voxels the blob owns
 * replaces: the label-volume dilation + per-blob np.mean of colocalizer.colocalize_blobs
 * (magmap/cv/colocalizer.py:372-421): ball(2) around every blob centre, contested voxels go to the
 * higher row index among blobs of the same channel of the same block; float64 mean in NumPy's
 * summation order (bit-equal), NaN when the blob owns no voxel.
 *   vol      : the image channel to average (any mmx_dtype; raw voxels or a preprocessed slot buffer)
 *   d_blocks : block geometry (src_off into vol, extent); blob coordinates are block-relative
 *   d_blobs  : [n_blobs][5] int32 (block slot, z, y, x, channel of the blob), grouped by block in table
 *              order; d_offsets[n_blocks+1] delimits the blocks
 *   d_mean   : out [n_blobs] float64; d_count: out [n_blobs] owned voxels (0..33)                   */
int mmx_coloc_means(const mmx_volume* vol, const mmx_block* d_blocks, int n_blocks,
                    const int32_t* d_blobs, const int32_t* d_offsets, int n_blobs,
                    double* d_mean, int32_t* d_count, void* stream);
/* The same, and the owned voxels themselves: d_voxels[n_blobs][MMX_COLOC_BALL] float64, the first d_count[b]
 * entries of row b in the C order of the reference's boolean-mask selection -- what its percentile thresholds
 * are taken over (`np.percentile(roi[mask >= 0, chl], thresh)`, magmap/cv/colocalizer.py:403-409).
 * d_voxels may be NULL (then this is mmx_coloc_means). */
int mmx_coloc_voxels(const mmx_volume* vol, const mmx_block* d_blocks, int n_blocks,
                     const int32_t* d_blobs, const int32_t* d_offsets, int n_blobs,
                     double* d_mean, int32_t* d_count, double* d_voxels, void* stream);

/* ---- U1: spectral unmixing ahead of detection
 * replaces: detector.detect_blobs' `roi_detect = np.subtract(roi_detect, fac * roi[..., k]);
 * roi_detect[roi_detect < 0] = 0` for every (k, fac) of the profile's spectral_unmixing entry
 * (magmap/cv/detector.py:910-921), float64, bit-equal.
 *   vol, h_subs[k] : the detected channel and the channels to subtract (same dtype, same block table)
 *   d_blocks       : block geometry (src_off into every source, extent, slot)
 *   d_out32/64     : [n_blocks][dst_slot] with strides (dst_sz, dst_sy, 1): float32 for the LoG passes,
 *                    float64 for the exact re-score                                               */
int mmx_unmix_batch(const mmx_volume* vol, const mmx_volume* h_subs, const double* h_facs, int n_subs,
                    const mmx_block* d_blocks, const mmx_block* h_blocks, int n_blocks,
                    int64_t dst_slot, int64_t dst_sy, int64_t dst_sz,
                    float* d_out32, double* d_out64, void* stream);

/* ---- R1: isotropic rescale ahead of detection
 * replaces: cv_nd.make_isotropic (magmap/cv/cv_nd.py:1070-1164) as detect_blobs calls it
 * (magmap/cv/detector.py:893-897) = skimage.transform.resize(mode="reflect", preserve_range=True) =
 * scipy.ndimage.zoom(order=1, mode='mirror', grid_mode=True), clipped to the input range, cast back to the
 * input dtype -- per block and channel, bit-equal to SciPy's float64 arithmetic.
 * mmx_minmax_batch folds min / max of one channel's blocks into d_minmax[n_blocks][2] (the caller presets
 * +inf / -inf and calls it for every channel of the ROI: scikit-image clips to the range of the whole
 * multichannel block).  mmx_resize_batch resamples one channel.
 *   d_index/d_weight : per axis and output index the two border-mapped source indices and the two linear
 *                      weights, as NI_ZoomShift precomputes them (tz/ty/tx: the block's table offsets)
 *   d_out            : uint8 / uint16 / float64 like the input, [n_blocks][dst_slot], strides (dst_sz,
 *                      dst_sy, 1); d_out32: float32 copy (float64 inputs only)                       */
typedef struct {
    int64_t src_off;                  /* element offset of the block origin in the source               */
    int32_t in_nz, in_ny, in_nx;      /* source extent                                                   */
    int32_t out_nz, out_ny, out_nx;   /* resized extent                                                  */
    int32_t slot;                     /* output slot and row of d_minmax                                 */
    int32_t tz, ty, tx;               /* offsets (in output indices) of the axis tables                  */
} mmx_resize_block;                   /* 48 bytes */
int mmx_minmax_batch(const mmx_volume* vol, const mmx_block* d_blocks, const mmx_block* h_blocks,
                     int n_blocks, double* d_minmax, void* stream);
int mmx_resize_batch(const mmx_volume* vol, const mmx_resize_block* d_blocks,
                     const mmx_resize_block* h_blocks, int n_blocks,
                     const int32_t* d_index, const double* d_weight, const double* d_minmax,
                     int64_t dst_slot, int64_t dst_sy, int64_t dst_sz,
                     void* d_out, float* d_out32, void* stream);

/* The same with the result's voxel type given explicitly (MMX_U8 / MMX_U16 from a float64 source: the
 * anti-aliased path below; otherwise out_dtype must equal vol->dtype). */
int mmx_resize_batch_as(const mmx_volume* vol, const mmx_resize_block* d_blocks,
                        const mmx_resize_block* h_blocks, int n_blocks,
                        const int32_t* d_index, const double* d_weight, const double* d_minmax,
                        int64_t dst_slot, int64_t dst_sy, int64_t dst_sz,
                        int out_dtype, void* d_out, float* d_out32, void* stream);

/* Anti-aliasing ahead of a down-sampling resize (skimage.transform.resize, anti_aliasing=True ->
 * scipy.ndimage.gaussian_filter(image.astype(float), (factor - 1) / 2, mode='mirror' | 'nearest')): ONE exact
 * float64 correlate1d pass along `axis` for every block; block b uses d_weights[b * w_pitch + 0..d_radius[b]]
 * (half kernel, weight at distance k) -- a truncated block has its own zoom factor.  vol: uint8 / uint16 /
 * float64 at any strides (d_blocks[b].src_off) -> d_out float64 [n_blocks][dst_slot], strides (dst_sz, dst_sy,
 * 1); a float32 volume -> float32 d_out (SciPy rounds every pass of a float32 image to float32).
 * nearest: 0 = every block extends its lines by mirroring, 1 = by repeating the edge sample ('nearest': the
 * reference's mode for blocks with an axis of length 1, cv_nd.py:1095-1101), 2 = per block: those whose
 * mmx_block._pad has bit 0 set repeat the edge sample, the others mirror. */
int mmx_gauss_axis_batch(const mmx_volume* vol, const mmx_block* d_blocks, const mmx_block* h_blocks,
                         int n_blocks, int axis, const double* d_weights, const int32_t* d_radius,
                         int w_pitch, int nearest, int64_t dst_slot, int64_t dst_sy, int64_t dst_sz,
                         void* d_out, void* stream);

/* ---- measurement helpers (bench.py): HIP-event timing on the caller's stream.
 * mmx_timing_enable(1) makes every kernel launch of this library record a HIP event
 * before and after itself on its launch stream; mmx_timing_read() synchronises those
 * events, returns summed milliseconds and launch counts per kernel family (index =
 * MMX_K_*: 0 z pass, 1 y pass, 2 x pass, 3 generic passes, 4 peaks, 5 rescore,
 * 6 overlap pairs, 7 close pairs, 8 fused z+x pass, 9 y pass of the fused path, 10 preprocessing,
 * 11 co-localisation means, 12 operand-ordered voxel copy of the tiled path) and starts a new window.
 * mmx_timing_enable(m), m > 1: only the families whose bit (k + 1) is set in m record events (an event between two
 * kernels keeps the second from starting under the first one's tail: timing one family perturbs a step less). */
#define MMX_K_COUNT 13
int mmx_timing_enable(int on);
int mmx_timing_read(double* ms, int64_t* launches, int n);

/* A rectangle of a host image into its place in the device copy (ABI v16): `height` rows of `width` bytes, rows
 * spitch / dpitch bytes apart, host -> device on `stream` (asynchronous for pinned host memory).  The upload of a host
 * volume block row by block row: a y-band of a z-range is one such rectangle (rows = planes). */
int mmx_copy_rect_h2d(void* d_dst, size_t dpitch, const void* h_src, size_t spitch, size_t width, size_t height,
                      void* stream);
/* The staged upload of a pageable / memory-mapped host image as ONE call (ABI v16; no interpreter lock needed while
 * it runs): for each of the n_regions (z0, z1, y0, y1) of the (nz, ny, row_bytes) image in turn -- wait for
 * events[k - depth] (the DMA that last read staging buffer k % depth), copy the region into that buffer with n_threads
 * threads (planes packed: (y1 - y0) * row_bytes each), queue mmx_copy_rect_h2d's rectangle on `stream`, record
 * events[k], store k + 1 to *n_queued.  *cancel != 0 (written by another thread) ends the loop at the next region.
 * h_staging[depth]: pinned buffers of staging_bytes each; events[n_regions]: created by the caller. */
int mmx_host_stage_upload(const void* h_src, void* d_dst, const int64_t* regions, int32_t n_regions, int64_t nz,
                          int64_t ny, int64_t row_bytes, void* const* h_staging, int64_t staging_bytes, int32_t depth,
                          void* const* events, void* stream, int32_t device, int64_t* n_queued, const int32_t* cancel,
                          int32_t n_threads);
int mmx_event_query(void* ev);      /* 0: completed, 1: not yet, else an mmx_status */
int mmx_event_create(void** ev);
int mmx_event_destroy(void* ev);
int mmx_event_record(void* ev, void* stream);
int mmx_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on stop */

/* ---- A12/A13 on the host (no device work): one axis of the cross-block duplicate pruning of
 * magmap/cv/stack_detect.py:679-861 (StackPruner.prune_blobs_mp) with the match/average/delete
 * rules of magmap/cv/detector.py:1000-1085 (remove_close_blobs).
 *   zyx, tag : int32 [n_table][3] detection coordinates / block grid coordinates
 *   abs_zyx  : float64 [n_table][3], updated in place (round-half-even means)
 *   cur      : current row ids in table order; bounds: the 2*n_sections-1 region starts along `axis`
 *              (pass 0, slab 0, pass 1, ...); last_end: end of the last block
 *   nxt_lo/hi: per slab the "adjacent region" range for the pruning-ratio statistic (NaN = none)
 *   out_cur  : surviving row ids in the reference's new order, *out_n of them
 *   n_slab, n_after, n_next : per slab row counts (before, after, adjacent region)          */
int mmx_host_prune_axis(const int32_t* zyx, const int32_t* tag, double* abs_zyx,
                        const int64_t* cur, int64_t n_cur, int axis, int n_sections,
                        const double* bounds, double last_end, const int32_t tol[3],
                        const double* nxt_lo, const double* nxt_hi,
                        int64_t* out_cur, int64_t* out_n,
                        int64_t* n_slab, int64_t* n_after, int64_t* n_next);

/* Output assembly of the same step: out[i] = table[rows[i]][:n_cols] with the three absolute-coordinate
 * columns taken from the compact (n_table, 3) array the axis steps updated (the reference's
 * `blobs_all[:, :-3]` after pruning, stack_detect.py:858-861).  table: float64, row pitch ld. */
int mmx_host_take_rows(const double* table, int64_t ld, const int64_t* rows, int64_t n,
                       int64_t n_cols, const double* abs_zyx, const int32_t abs_cols[3], double* out);
/* ... and with the reference's last two steps on the pruned table folded in (magmap/cv/stack_detect.py:455-470:
 * `replace_rel_with_abs_blob_coords`, `remove_abs_blob_coords(True)` -- two more passes over the whole table when made
 * afterwards): out[i][j] = table[rows[i]][src_cols[j]], j < n_out (3..64), then out[i][abs_dst0 .. abs_dst0 + 3] =
 * abs_zyx[rows[i]].  mmx_host_gather_parts_by_key_final: the same for mmx_host_gather_by_key, on the concatenation of
 * n_parts survivor lists (ids[p], keys[p], abs_rows[p]: n_rows[p] entries each) that nobody has to concatenate;
 * out_rows must equal their total. */
int mmx_host_take_rows_final(const double* table, int64_t ld, const int64_t* rows, int64_t n,
                             const int32_t* src_cols, int32_t n_out, const double* abs_zyx, int32_t abs_dst0,
                             double* out);
int mmx_host_gather_parts_by_key_final(const double* table, int64_t ld, int32_t n_parts, const int64_t* const* ids,
                                       const int64_t* const* keys, const double* const* abs_rows,
                                       const int64_t* n_rows, int64_t n_keys, const int32_t* src_cols, int32_t n_out,
                                       int32_t abs_dst0, double* out, int64_t out_rows);
/* ... with the output in TWO tables (ABI v16): columns [0, n_main) of the layout into `out` (row pitch n_main,
 * abs_dst0 + 3 <= n_main), the other n_out - n_main into `out_rest` (row pitch n_out - n_main).  A stack detected with
 * co-localisation ends as eight final columns plus the columns its flags are read from -- `segments_all[:, 10:10 + C]`,
 * magmap/cv/stack_detect.py:463-464 -- and both leave the one pass that gathers the surviving rows.  out_rest NULL:
 * the plain form above. */
int mmx_host_take_rows_split(const double* table, int64_t ld, const int64_t* rows, int64_t n,
                             const int32_t* src_cols, int32_t n_out, const double* abs_zyx, int32_t abs_dst0,
                             double* out, int32_t n_main, double* out_rest);
int mmx_host_gather_parts_by_key_split(const double* table, int64_t ld, int32_t n_parts, const int64_t* const* ids,
                                       const int64_t* const* keys, const double* const* abs_rows,
                                       const int64_t* n_rows, int64_t n_keys, const int32_t* src_cols, int32_t n_out,
                                       int32_t abs_dst0, double* out, int64_t out_rows, int32_t n_main,
                                       double* out_rest);

/* All three axis passes of the pruning for one REGION of the stack (the whole stack, one rank's blocks, or a group
 * of blocks pruned while the GPU still works on later ones): a table holding the region's own rows (ids
 * [own_lo, own_hi)) between those rows of its neighbours that lie within 3 x tol of its extent (a pass looks tol far
 * and sees the outcome of the passes before it), in the merged table's order.  Own rows get the verdicts and
 * averaged coordinates the reference's whole-table passes (magmap/cv/stack_detect.py:680-861) give them; the
 * passes being stable sorts by group, a survivor's place in the final table is the place of its key
 * (group on axis 2, axis 1, axis 0) among all survivors, regions in order on equal keys.
 *   cur[n_cur]: rows of one channel (own and halo), table order; n_sections[a] <= 1: no pass on axis a
 *   bounds / nxt_lo / nxt_hi / last_end / tol: as for mmx_host_prune_axis, per axis
 *   out_rows / out_keys / *out_n: own survivors in final order; abs_zyx updated in place
 *   n_slab / n_after / n_next: [3][stat_ld] statistics over OWN rows
 * mmx_host_merge_by_key: out = the stable sort by key of rows[n][ld] (first n_cols columns), keys < n_keys; keys == NULL:
 *   a row's key is the value in its column n_cols.
 * mmx_host_gather_by_key: the same for survivors still in the merged table: row ids[i], its three abs columns
 *   replaced by abs_rows[i][3]. */
int mmx_host_prune_region(const int32_t* zyx, const int32_t* tag, double* abs_zyx, const int64_t* cur, int64_t n_cur,
                          int64_t own_lo, int64_t own_hi, const int32_t n_sections[3], const double* const bounds[3],
                          const double last_end[3], const int32_t tol[3], const double* const nxt_lo[3],
                          const double* const nxt_hi[3], int64_t* out_rows, int64_t* out_keys, int64_t* out_n,
                          int64_t* n_slab, int64_t* n_after, int64_t* n_next, int64_t stat_ld);
/* mmx_host_prune_region for a region whose rows are still in the merged table: `parts` [n_parts][2] are ascending row
 * ranges of it, part `own_part` the region itself, the others its neighbours, of which only rows inside
 * [box_lo, box_hi) take part; every channel of `channels` in turn (chan: the table's channel column, pitch chan_ld
 * doubles, or NULL = all rows are channels[0]); out_ids: the region's survivors as rows of the merged table,
 * out_keys: channel position x n_keys + key, out_abs: their averaged coordinates [..][3]; the merged table itself is
 * not written to.  Statistics [n_channels][3][stat_ld].  (out_keys of mmx_host_prune_region may be NULL.) */
int mmx_host_prune_parts(const int32_t* zyx, const int32_t* tag, const double* abs_zyx, const double* chan,
                         int64_t chan_ld, const int64_t* parts, int n_parts, int own_part, const int32_t box_lo[3],
                         const int32_t box_hi[3], const double* channels, int n_channels,
                         const int32_t n_sections[3], const double* const bounds[3], const double last_end[3],
                         const int32_t tol[3], const double* const nxt_lo[3], const double* const nxt_hi[3],
                         int64_t n_keys, int64_t* out_ids, int64_t* out_keys, double* out_abs, int64_t* out_n,
                         int64_t* n_slab, int64_t* n_after, int64_t* n_next, int64_t stat_ld);
/* the distributed pruning's table plumbing (stack_detect.StackPruner._prune_distributed), native:
 * mmx_host_rows_in_boxes: the rows of a rank's table inside any of the other ranks' (widened) boxes, ten float64 values a
 *   row (zyx, block tag, abs zyx, channel) -- the payload of the first exchange; *out_n keeps counting past cap;
 * mmx_host_append_rows: the rows of a received payload inside this rank's box, appended to its compact columns from row
 *   `at` on (MMX_ERR_WORKSPACE when they do not fit `cap` rows; *out_n says how many there are);
 * mmx_host_emit_survivors: rows ids[i] of the merged table with their averaged abs columns and their sort key as an
 *   extra last column -- the payload of the second exchange.
 * (mmx_host_prune_parts takes its parts in LOCAL order: a rank lists the halo rows of earlier ranks, its own rows, the
 *  halo rows of later ranks, wherever they sit in its arrays.) */
int mmx_host_rows_in_boxes(const int32_t* zyx, const int32_t* tag, const double* abs_zyx, const double* chan,
                           int64_t chan_ld, int64_t n, const int32_t* box_lo, const int32_t* box_hi, int n_boxes,
                           double* out, int64_t cap, int64_t* out_n);
int mmx_host_append_rows(const double* payload, int64_t n, const int32_t lo[3], const int32_t hi[3], int32_t* zyx,
                         int32_t* tag, double* abs_zyx, double* chan, int64_t chan_ld, int64_t at, int64_t cap,
                         int64_t* out_n);
int mmx_host_emit_survivors(const double* table, int64_t ld, const int64_t* ids, const int64_t* keys, int64_t n,
                            int64_t n_cols, const double* abs_rows, const int32_t abs_cols[3], double* out);
/* ... in the final columns (src_cols / abs_dst0 as for mmx_host_take_rows_final): out[i][n_out + 1], the key last. */
int mmx_host_emit_survivors_final(const double* table, int64_t ld, const int64_t* ids, const int64_t* keys, int64_t n,
                                  const int32_t* src_cols, int32_t n_out, const double* abs_rows, int32_t abs_dst0,
                                  double* out);
/* ... for n_parts survivor lists at once, in order (a rank's regions): out[out_rows][n_out + 1]. */
int mmx_host_emit_parts_final(const double* table, int64_t ld, int32_t n_parts, const int64_t* const* ids,
                              const int64_t* const* keys, const double* const* abs_rows, const int64_t* n_rows,
                              const int32_t* src_cols, int32_t n_out, int32_t abs_dst0, double* out, int64_t out_rows);
int mmx_host_merge_by_key(const double* rows, int64_t ld, const int64_t* keys, int64_t n, int64_t n_keys,
                          int64_t n_cols, double* out);
/* ... on the concatenation of n_parts row blocks (parts[p]: n_rows[p] rows of pitch ld, the key in column n_cols) that
 * nobody has to concatenate -- what an all_gather of the ranks' survivors, padded to the longest block, leaves. */
int mmx_host_merge_parts_by_key(const double* const* parts, const int64_t* n_rows, int32_t n_parts, int64_t ld,
                                int64_t n_keys, int64_t n_cols, double* out, int64_t out_rows);
int mmx_host_gather_by_key(const double* table, int64_t ld, const int64_t* ids, const int64_t* keys, int64_t n,
                           int64_t n_keys, int64_t n_cols, const double* abs_rows, const int32_t abs_cols[3],
                           double* out);

/* out[i][dst_col0 + j] = table[i][src_cols[j]], i < n, j < n_map (<= 64), threaded.  The column shuffles that
 * end a stack detection (reference magmap/cv/stack_detect.py:461-467 -> detector.py
 * replace_rel_with_abs_blob_coords / remove_abs_blob_coords).  `out` may alias `table`. */
int mmx_host_map_columns(const double* table, int64_t ld, int64_t n, const int32_t* src_cols,
                         int32_t n_map, double* out, int64_t out_ld, int32_t dst_col0);

/* ---- per-batch host work of the detection, native and threaded over blocks (no device work; host pointers).
 * mmx_host_resolve_peaks: peak membership on the exact float64 values and the reference's two orders
 *   replaces: `image == maximum_filter(image)` & `image > threshold`, np.nonzero, argsort(-values) of
 *   skimage.feature.peak_local_max (skimage/feature/peak.py:9-50) for the candidates of one batch.
 *   cands[0, n_cands) candidates, cands[n_cands, n_total) probes (mmx_expand_probes), all re-scored.
 *   out_coords/out_vals: per block the peaks ([z, y, x, sigma index] int32, float64) by descending value, equal
 *   values in np.nonzero order; offsets[n_blocks + 1];
 *   ties[b] = 1 when block b holds two equal values (np.argsort's order of equal keys is its own: ask NumPy);
 *   out_nz_coords/out_nz_vals: the same rows in np.nonzero order -- written for the blocks with ties[b] == 1 only;
 *   stats[4]: contested candidates, peaks, max |float32 - float64| (inf when a value is not finite: nothing else
 *   is then written), blocks dropped as constant cubes (peak.py:41-43).
 * mmx_host_overlap_prune: skimage.feature.blob._prune_blobs (blob.py:84-187) on those peaks: `alive` per row, final
 *   for blocks with open_blocks[b] == 0 when *n_knife == 0 and *n_pairs <= cap; all pairs above overlap - band in
 *   pairs/frac for the caller's exact re-evaluation (knife-edge fractions) and reference pair order (open blocks).
 * mmx_host_emit_tables: the 11-column block tables (magmap/cv/detector.py:88-113, 934-943) shifted to ROI
 *   coordinates (stack_detect.py:164-170) and tagged with the block's grid coordinate (chunking.py:410-445),
 *   written into the caller's merged table from row0 on (row pitch ld >= 14: 11 + extra columns + 3 tags), plus
 *   the compact int32 / float64 columns mmx_host_prune_axis reads.  MMX_ERR_WORKSPACE when `capacity` rows do not
 *   suffice (nothing written). */
int mmx_host_resolve_peaks(const mmx_cand* cands, uint32_t n_cands, uint32_t n_total, const mmx_block* blocks,
                           int n_blocks, int n_sigma, double thr, int32_t* out_nz_coords, double* out_nz_vals,
                           int32_t* out_coords, double* out_vals, int32_t* offsets, uint8_t* ties, double* stats);
int mmx_host_overlap_prune(const int32_t* coords, const int32_t* offsets, int n_blocks, const double* sigmas,
                           int n_sigma, double overlap, double band, uint8_t* alive, uint8_t* open_blocks,
                           int32_t* pairs, double* frac, int64_t cap, int64_t* n_pairs, int64_t* n_knife);
int mmx_host_emit_tables(const int32_t* coords, const uint8_t* alive, const int32_t* offsets, int n_blocks,
                         const double* sigmas, int n_sigma, double channel, const double* block_offsets,
                         const int32_t* block_tags, const int32_t* interior, double* store, int64_t ld,
                         int32_t* zyx, int32_t* tag, double* abs_zyx, int64_t row0, int64_t capacity,
                         int64_t* rows_per_block);
/* ... for blocks detected in SEVERAL channels (ABI v16; the reference's per-channel loop and `np.vstack` in
 * detect_blobs, magmap/cv/detector.py:899-943): block b's table = channel 0's rows, then channel 1's ..., each channel
 * from its own peak arrays over the same n_blocks blocks.  n_extra >= 0 columns behind the 11 named ones are zeroed
 * (-1: untouched); any_before[b] (optional): a blob of block b existed before the border exclusion (None vs an EMPTY
 * table, :941-942); coloc_rows (optional) [rows][5] int32: block, z, y, x (block-relative), channel per written row --
 * mmx_coloc_means' d_blobs.
 * mmx_host_coloc_flags: colocalizer.colocalize_blobs' thresholds and flags (magmap/cv/colocalizer.py:372-441, thresh
 * "min") for a batch of such tables from the per-blob channel means -- per block, per channel present among its in-ROI
 * blobs: threshold = smallest mean of that channel over the channel's own in-ROI blobs (NaN-poisoned like np.amin),
 * flag 1 for every in-ROI blob whose mean reaches it -- written to flags[r * ld + channel] (the tables' extra columns). */
int mmx_host_emit_tables_multi(int32_t n_channels, const int32_t* const* coords, const uint8_t* const* alive,
                               const int32_t* const* offsets, int n_blocks, const double* const* sigmas,
                               const int32_t* n_sigma, const double* channel_ids, const double* block_offsets,
                               const int32_t* block_tags, const int32_t* interior, double* store, int64_t ld,
                               int32_t n_extra, int32_t* zyx, int32_t* tag, double* abs_zyx, int64_t row0,
                               int64_t capacity, int64_t* rows_per_block, uint8_t* any_before, int32_t* coloc_rows);
int mmx_host_coloc_flags(const double* means, const int32_t* mean_channels, int32_t n_mean_channels, int64_t n,
                         const int32_t* rows, const int64_t* row_offsets, int n_blocks, const int32_t* shapes,
                         int32_t n_channels, double* flags, int64_t ld);

/* A small stack -- all blocks in ONE batch: the GUI's ROI, a grid-search step (magmap/cv/detector.py:931-933 once per
 * ROI; gui/visualizer.py:2758, io/cli.py:1111-1151) -- from the re-scored candidate table to the final table in one call
 * (ABI v16): mmx_host_resolve_peaks -> mmx_host_overlap_prune -> mmx_host_emit_tables (rows from 0 on) ->
 * mmx_host_prune_region over the whole table -> mmx_host_take_rows_final, each with the meaning of its own entry point.
 *   cands .. n_sigma, thr : as for mmx_host_resolve_peaks; eps: the nomination band (max |f32 - f64| must stay < eps / 4)
 *   sigmas, overlap, overlap_band : as for mmx_host_overlap_prune
 *   channel .. any_before : as for mmx_host_emit_tables[_multi] (one channel; capacity rows in store / zyx / tag / abs)
 *   n_sections .. stat_ld : as for mmx_host_prune_region (own rows = all rows)
 *   src_cols, n_out, abs_dst0, out[out_capacity][n_out], *out_rows : as for mmx_host_take_rows_final
 *   stats[8] : contested candidates, peaks, max |f32 - f64|, constant cubes, overlap pairs, blobs after the per-block
 *              prune, and [6] = why the call returned MMX_DEFERRED: 1 equal peak values in a block (NumPy's argsort
 *              order decides), 2 the band is too narrow (or a non-finite value), 3 a knife-edge overlap fraction or a
 *              blob that wins one pair and loses another (the reference's libm calls / pair order), 4 more rows than
 *              `capacity` / `out_capacity`.  MMX_DEFERRED leaves the merged table untouched: the caller takes the
 *              call-by-call path on the same candidates. */
typedef struct {
    const mmx_cand* cands; uint32_t n_cands, n_total;
    const mmx_block* blocks; int32_t n_blocks, n_sigma;
    double thr, eps;
    const double* sigmas; double overlap, overlap_band;
    double channel;
    const double* block_offsets; const int32_t* block_tags; const int32_t* interior;
    double* store; int64_t ld; int32_t* zyx; int32_t* tag; double* abs_zyx; int64_t capacity;
    int64_t* rows_per_block; uint8_t* any_before;
    const int32_t* n_sections; const double* const* bounds; const double* last_end; const int32_t* tol;
    const double* const* nxt_lo; const double* const* nxt_hi;
    int64_t* n_slab; int64_t* n_after; int64_t* n_next; int64_t stat_ld;
    const int32_t* src_cols; int32_t n_out, abs_dst0;
    double* out; int64_t out_capacity; int64_t* out_rows;
    double* stats;
} mmx_finish_stack_args;
int mmx_host_finish_stack(const mmx_finish_stack_args* a);

/* ---- match-based co-localisation: the two third-party calls of the reference's
 * verifier.find_closest_blobs_cdist (magmap/cv/verifier.py:47-119).
 * mmx_cdist_f64: d_out[i * m + j] = || a_i - b_j ||_2 for float64 points of `dim` (<= 64) coordinates -- replaces
 *   scipy.spatial.distance.cdist(a, b) (:85), same operation order, no FMA: bit-equal.  n <= 65535.
 * mmx_host_lsap: optimal assignment of a dense nr x nc float64 cost matrix (host memory) -- replaces
 *   scipy.optimize.linear_sum_assignment(dists) (:86): min(nr, nc) pairs in ascending row order, and where the
 *   optimum is not unique (blob coordinates are integers: tied distances are common) the SAME optimum SciPy's
 *   shortest-augmenting-path solver returns. */
int mmx_cdist_f64(const double* d_a, int64_t n, const double* d_b, int64_t m, int dim, double* d_out, void* stream);
int mmx_host_lsap(const double* cost, int64_t nr, int64_t nc, int64_t* out_rows, int64_t* out_cols);

/* PMC calibration (tools/pmc_calib.py): one streaming launch over n_elems elements with a known
 * byte count.  kind 0: float copy, 4 B per lane; 1: float copy, 16 B per lane; 2: uint16 read. */
int mmx_calib_stream(int kind, const void* d_in, void* d_out, int64_t n_elems, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MMX_H */
