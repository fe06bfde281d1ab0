// mmx_detect_batch: one batch of blocks from voxels to the re-scored candidate table in ONE call (SURVEY.md section
// 8b: "mmx_detect_block -- fused A0-A4").  Replaces, for the blocks of a batch, everything between the reference's
// call `blob_log(roi, ...)` (magmap/cv/detector.py:931-933) and the point where its float64 cube values are compared:
// img_as_float + gaussian_laplace per sigma + scale normalisation (A0-A3) and the nomination half of peak_local_max
// (A4).  What the Python host (blob_log._enqueue_detect) used to enqueue call by call -- the voxel copy of the tiled
// path, (Z+X, Y) per sigma with the fall-back rules between kernel paths, the counter reset, the sparse NMS, the probe
// expansion, the exact float64 re-score, the copies of the counters and of the table's head to pinned host memory --
// is enqueued here by native code on the caller's streams.  Nothing in it waits for the GPU.
//
// A batch the caller runs again and again with the same arguments (a small volume detected once per step) can be
// captured as a hipGraph by the caller (mmx_graph_*): every node's arguments are then fixed at capture time.

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "mmx_common.h"

namespace {

thread_local char g_detect_err[256] = "";

// fork / join events between the caller's streams: a small ring, reused (a stream's wait captures the state of the
// event at the time of the call, so recording it again later does not disturb a wait already enqueued)
std::mutex g_ev_mutex;
std::vector<hipEvent_t> g_ev_ring;
size_t g_ev_next = 0;
hipEvent_t ring_event()
{
    std::lock_guard<std::mutex> lk(g_ev_mutex);
    if (g_ev_ring.size() < 64) {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        g_ev_ring.push_back(e);
        return e;
    }
    hipEvent_t e = g_ev_ring[g_ev_next];
    g_ev_next = (g_ev_next + 1) % g_ev_ring.size();
    return e;
}

int fail(hipError_t e, const char* what)
{
    snprintf(g_detect_err, sizeof g_detect_err, "%s: %s", what, hipGetErrorString(e));
    return MMX_ERR_HIP;
}

// stream b continues after everything enqueued on stream a so far
int after(hipStream_t a, hipStream_t b, const char* what)
{
    if (a == b) return MMX_OK;
    hipEvent_t e = ring_event();
    if (!e) return fail(hipErrorOutOfMemory, "hipEventCreate");
    hipError_t r = hipEventRecord(e, a);
    if (r != hipSuccess) return fail(r, what);
    r = hipStreamWaitEvent(b, e, 0);
    return r == hipSuccess ? MMX_OK : fail(r, what);
}

// Every scale of the batch through mmx_log_batch_f32; `layouts`: bit (1 << layout) for every NMS entry layout a call
// reported (bit 0: none).  The rules are blob_log.py's (round 3), moved here unchanged:
//   * the tiled path works from an operand-ordered copy of the voxels that does not depend on sigma: made once, trusted
//     by the calls below for as long as every call so far ran the tiled path;
//   * 16-bit intermediates when the nomination band covers their rounding error fourfold and the error in value units
//     is inside the LoG contract (MMX_LOG_ABS_TOL) -- or when asked for by name.
int passes(const mmx_detect_args* a, bool with_mask, int mode, float* d_log, uint64_t* d_mask, size_t mask_words,
           unsigned* layouts, int* zx_path, double* q16_bound, bool* pack_side)
{
    *layouts = 0;
    const mmx_volume* vol = a->vol32;
    const bool is_float = vol->dtype == MMX_F32;
    const bool float_ok = is_float && vol->value_range != 0.f;
    const bool nonneg = !is_float || (float_ok && vol->value_range > 0.f);
    int tiled_mode = MMX_ZX_TILED;
    const int nb = a->n_blocks, ns = a->n_sigma;
    const int64_t slot = a->slot_elems;
    const size_t tab = MMX_MAX_RADIUS_GENERIC + 1;
    if ((mode == MMX_ZX_AUTO || mode == MMX_ZX_TILED_Q16) && nonneg) {
        double bound = 0.0;
        for (int s = 0; s < ns; ++s) {
            const double b = mmx_tiled_q16_error_bound(a->h_w0 + s * tab, a->h_w2 + s * tab, a->h_radius[s], a->h_norm[s]);
            if (b < 0) { bound = -1.0; break; }
            if (b > bound) bound = b;
        }
        if (is_float) bound *= (double)vol->value_range;
        const bool covers = bound >= 0.0 && 4.0 * bound <= (double)a->eps && bound <= MMX_LOG_ABS_TOL;
        // (by name: taken whatever the band; the caller's run-time check of |float32 - float64| against eps / 4 on the
        //  re-scored candidates is what then widens it)
        if (mode == MMX_ZX_TILED_Q16 || covers) { tiled_mode = MMX_ZX_TILED_Q16; if (q16_bound) *q16_bound = bound; }
    }
    bool packed = false;
    hipStream_t main = (hipStream_t)a->stream;
    if (((mode == MMX_ZX_AUTO || mode == MMX_ZX_TILED || mode == MMX_ZX_TILED_Q16) && !is_float) ||
        ((mode == MMX_ZX_AUTO || mode == MMX_ZX_TILED) && float_ok)) {
        hipStream_t ps = (*pack_side && a->pack_stream) ? (hipStream_t)a->pack_stream : main;
        int rc = mmx_zx_pack(vol, a->d_blocks, a->h_blocks, nb, slot, a->d_work, ps);
        if (rc != MMX_OK && rc != MMX_ERR_UNSUPPORTED) return rc;
        if (ps != main) {
            const int st = after(ps, main, "voxel copy -> main stream");
            if (st != MMX_OK) return st;
            *pack_side = false;           // (a second round of passes, should one be needed: in stream order)
        }
        packed = rc == MMX_OK;
    }
    for (int s = 0; s < ns; ++s) {
        int written = 0, path = 0;
        const int rc = mmx_log_batch_f32(vol, a->d_blocks, a->h_blocks, nb, slot, a->h_w0 + s * tab, a->h_w2 + s * tab,
                                         a->h_radius[s], a->h_norm[s], d_log + (size_t)s * nb * slot, a->d_work,
                                         with_mask ? d_mask + (size_t)s * mask_words * 2 : nullptr, a->thr - a->eps, a->eps,
                                         &written, packed ? (tiled_mode | MMX_ZX_PREPACKED | a->zx_flags) : mode, &path,
                                         (void*)main);
        if (rc != MMX_OK) return rc;
        *zx_path = path;
        packed = packed && path == tiled_mode;
        *layouts |= 1u << (with_mask ? written : 0);
    }
    return MMX_OK;
}

}  // namespace

extern "C" {

const char* mmx_detect_last_error(void) { return g_detect_err; }

int mmx_detect_batch(const mmx_detect_args* a, mmx_detect_info* info)
{
    if (!a || !info) return MMX_ERR_ARG;
    memset(info, 0, sizeof *info);
    if (!a->vol32 || !a->d_blocks || !a->h_blocks || !a->h_w0 || !a->h_w2 || !a->h_radius || !a->h_norm || !a->d_work ||
        !a->d_cands || !a->d_count || a->n_blocks < 1 || a->n_sigma < 1 || a->slot_elems < 1 || a->cap < 1 || !(a->eps >= 0.f))
        return MMX_ERR_ARG;
    if (a->exact && (!a->vol_exact || !a->d_w0 || !a->d_w2)) return MMX_ERR_ARG;
    const int nb = a->n_blocks, ns = a->n_sigma;
    const int64_t slot = a->slot_elems;
    if (mmx_workspace_bytes(nb, slot, ns, 1) > a->work_bytes) return MMX_ERR_WORKSPACE;
    hipStream_t main = (hipStream_t)a->stream;
    hipStream_t tail = a->tail_stream ? (hipStream_t)a->tail_stream : main;
    hipError_t r;
    // the workspace may still be read by the tail of the batch that used it before
    bool pack_side = a->pack_stream != nullptr && a->pack_stream != a->stream;
    if (a->ev_work_free) {
        // (both: the voxel copy is the first writer when there is one, the passes when there is none)
        r = hipStreamWaitEvent(main, (hipEvent_t)a->ev_work_free, 0);
        if (r == hipSuccess && pack_side) r = hipStreamWaitEvent((hipStream_t)a->pack_stream, (hipEvent_t)a->ev_work_free, 0);
        if (r != hipSuccess) return fail(r, "wait for the workspace");
    }
    float* d_log = a->d_work + (size_t)4 * nb * slot;
    // NMS entries, [ns][(nb * slot) >> 5] 16-byte entries behind the LoG arrays (mmx_workspace_bytes' layout)
    const size_t mask_words = ((size_t)nb * slot) >> 5;
    uint64_t* d_mask = reinterpret_cast<uint64_t*>(
        (reinterpret_cast<uintptr_t>(d_log + (size_t)ns * nb * slot) + 15) & ~(uintptr_t)15);

    // With the entries the Y pass leaves whole segments of the cube unwritten, so it is all scales, in one layout, or
    // none: if one scale cannot produce them (a radius outside the fused kernels, tiny blocks) or the scales ran
    // different kernels, every scale is computed again -- with the packed kernel's entries if a scale produced those,
    // else in full.
    unsigned layouts = 0;
    int zx_path = 0;
    double q16_bound = 0.0;
    int rc = passes(a, true, a->zx_mode, d_log, d_mask, mask_words, &layouts, &zx_path, &q16_bound, &pack_side);
    if (rc != MMX_OK) return rc;
    if (layouts == ((1u << MMX_MASK_ROWS) | (1u << MMX_MASK_QUADS))) {
        rc = passes(a, true, MMX_ZX_PACKED, d_log, d_mask, mask_words, &layouts, &zx_path, &q16_bound, &pack_side);
        if (rc != MMX_OK) return rc;
        info->n_pass_rounds++;
    }
    if (layouts & (layouts - 1)) {       // more than one kind
        rc = passes(a, false, a->zx_mode, d_log, d_mask, mask_words, &layouts, &zx_path, &q16_bound, &pack_side);
        if (rc != MMX_OK) return rc;
        info->n_pass_rounds++;
    }
    int mask_layout = 0;
    while (!(layouts & (1u << mask_layout))) ++mask_layout;
    info->zx_path = zx_path;
    info->mask_layout = mask_layout;
    info->q16_bound = zx_path == MMX_ZX_TILED_Q16 ? q16_bound : 0.0;
    info->n_pass_rounds++;

    // ---- the tail: NMS, probes, exact values, copies (on its own stream when the caller gives one)
    rc = after(main, tail, "LoG passes -> tail stream");
    if (rc != MMX_OK) return rc;
    r = hipMemsetAsync(a->d_count, 0, 2 * sizeof(uint32_t), tail);
    if (r != hipSuccess) return fail(r, "reset of the candidate counters");
    rc = mmx_peaks_batch(d_log, mask_layout ? d_mask : nullptr, mask_layout, ns, a->d_blocks, a->h_blocks, nb, slot,
                         a->thr, a->eps, a->d_cands, a->cap, a->d_count, (void*)tail);
    if (rc != MMX_OK) return rc;
    if (a->ev_work_read) {               // the workspace may be written again once the NMS has read it
        r = hipEventRecord((hipEvent_t)a->ev_work_read, tail);
        if (r != hipSuccess) return fail(r, "record: workspace read");
    }
    if (a->expand) {
        rc = mmx_expand_probes(a->d_cands, a->cap, a->d_count, a->d_count + 1, a->d_blocks, nb, ns, (void*)tail);
        if (rc != MMX_OK) return rc;
    }
    if (a->exact) {
        rc = mmx_rescore_f64(a->vol_exact, a->d_blocks, nb, a->d_cands, a->cap, a->d_count, a->d_w0, a->d_w2,
                             a->h_radius, a->h_norm, ns, a->store_f32, (void*)tail);
        if (rc != MMX_OK) return rc;
    }
    if (a->h_count) {
        r = hipMemcpyAsync(a->h_count, a->d_count, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, tail);
        if (r != hipSuccess) return fail(r, "copy of the counters");
    }
    if (a->h_cands && a->h_prefix) {
        const size_t n = a->h_prefix < a->cap ? a->h_prefix : a->cap;
        r = hipMemcpyAsync(a->h_cands, a->d_cands, n * sizeof(mmx_cand), hipMemcpyDeviceToHost, tail);
        if (r != hipSuccess) return fail(r, "copy of the table's head");
    }
    if (a->ev_done) {
        r = hipEventRecord((hipEvent_t)a->ev_done, tail);
        if (r != hipSuccess) return fail(r, "record: batch done");
    }
    return MMX_OK;
}

int mmx_event_synchronize(void* ev)
{
    hipError_t r = hipEventSynchronize((hipEvent_t)ev);
    return r == hipSuccess ? MMX_OK : fail(r, "hipEventSynchronize");
}

int mmx_event_query(void* ev)
{
    // 0: the event has completed, 1: not yet (not an error), else a status
    hipError_t r = hipEventQuery((hipEvent_t)ev);
    if (r == hipSuccess) return 0;
    if (r == hipErrorNotReady) { (void)hipGetLastError(); return 1; }
    return fail(r, "hipEventQuery");
}

int mmx_stream_wait_event(void* stream, void* ev)
{
    hipError_t r = hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0);
    return r == hipSuccess ? MMX_OK : fail(r, "hipStreamWaitEvent");
}

// ---- a captured batch: the launches of one mmx_detect_batch call as a hipGraph, replayed with one launch.
// Every argument of every node is frozen at capture: the caller replays a graph only for the very same buffers,
// geometry, scales and band (its key), and only while per-kernel timing is off.
struct mmx_graph_s { hipGraph_t graph; hipGraphExec_t exec; mmx_detect_info info; };

int mmx_detect_batch_capture(const mmx_detect_args* a, mmx_detect_info* info, void** graph_out)
{
    if (!a || !info || !graph_out || !a->stream) return MMX_ERR_ARG;      // (the legacy stream cannot be captured)
    *graph_out = nullptr;
    hipStream_t main = (hipStream_t)a->stream;
    // (the per-kernel timing scopes record events of their own, which a capture would swallow)
    if (mmx_timing_is_enabled()) return MMX_ERR_UNSUPPORTED;
    hipError_t r = hipStreamBeginCapture(main, hipStreamCaptureModeThreadLocal);
    if (r != hipSuccess) return fail(r, "hipStreamBeginCapture");
    mmx_detect_args b = *a;
    // inside a capture the caller's cross-call events have no meaning: the graph is ordered as a whole by the stream it
    // is launched on; the side streams join the capture through the fork / join events
    b.ev_work_free = b.ev_work_read = b.ev_done = nullptr;
    // (the voxel copy's own stream joins a batch through ev_work_free, which a capture does not have: inside the graph
    //  the copy runs on the main stream)
    b.pack_stream = nullptr;
    const int rc = mmx_detect_batch(&b, info);
    hipGraph_t g = nullptr;
    if (rc == MMX_OK && b.tail_stream && b.tail_stream != b.stream) {
        // join the tail back into the origin stream before the capture ends
        const int st = after((hipStream_t)b.tail_stream, main, "tail stream -> main stream");
        if (st != MMX_OK) { hipStreamEndCapture(main, &g); if (g) hipGraphDestroy(g); return st; }
    }
    r = hipStreamEndCapture(main, &g);
    if (rc != MMX_OK) { if (g) hipGraphDestroy(g); return rc; }
    if (r != hipSuccess || !g) return fail(r, "hipStreamEndCapture");
    hipGraphExec_t ex = nullptr;
    r = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    if (r != hipSuccess) { hipGraphDestroy(g); return fail(r, "hipGraphInstantiate"); }
    mmx_graph_s* out = new mmx_graph_s{g, ex, *info};
    *graph_out = out;
    return MMX_OK;
}

int mmx_graph_launch(void* graph, void* stream, void* ev_done, mmx_detect_info* info)
{
    if (!graph) return MMX_ERR_ARG;
    mmx_graph_s* g = static_cast<mmx_graph_s*>(graph);
    hipError_t r = hipGraphLaunch(g->exec, (hipStream_t)stream);
    if (r != hipSuccess) return fail(r, "hipGraphLaunch");
    if (ev_done) {
        r = hipEventRecord((hipEvent_t)ev_done, (hipStream_t)stream);
        if (r != hipSuccess) return fail(r, "record: graph done");
    }
    if (info) *info = g->info;
    return MMX_OK;
}

int mmx_graph_destroy(void* graph)
{
    if (!graph) return MMX_OK;
    mmx_graph_s* g = static_cast<mmx_graph_s*>(graph);
    hipGraphExecDestroy(g->exec);
    hipGraphDestroy(g->graph);
    delete g;
    return MMX_OK;
}

}  // extern "C"


// ---------------------------------------------------------------------------------------------------------------------
// The staged upload of a pageable / memory-mapped host image (volume._SlabUpload), the whole loop in one call so that it
// needs nothing from the host language while it runs: per region (z0, z1, y0, y1) of the (nz, ny, row_bytes) image --
// wait until the DMA has read the staging buffer's previous contents, copy the region into it with `n_threads` threads
// (its rows packed: (y1 - y0) * row_bytes per plane), queue the rectangle copy into its place in the device image on
// `stream`, record events[k], publish k + 1 in *n_queued.  Round 5 / 6 ran this loop in a Python thread: between two
// regions it needed the interpreter lock, which a busy detection (two channels, co-localisation) holds most of the
// time -- 413 against 289 ms for a C5 tile from a memory map.  *cancel != 0 ends the loop at the next region.  The
// caller owns every buffer and event and keeps them alive until the call has returned AND the last recorded event has
// completed.  Returns MMX_OK also when cancelled (n_queued says how far it came).
namespace {
struct spin_barrier {
    std::atomic<int> count{0};
    std::atomic<int> phase{0};
    int n = 1;
    void wait()
    {
        const int p = phase.load(std::memory_order_acquire);
        if (count.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
            count.store(0, std::memory_order_relaxed);
            phase.store(p + 1, std::memory_order_release);
        } else {
            // (a few microseconds of polling for the common case -- the threads arrive together -- then sleep: the leader
            //  may be waiting two milliseconds for a DMA, and polling helpers take cores from the detection's host side)
            int spins = 0;
            while (phase.load(std::memory_order_acquire) == p)
                if (++spins > 4000) std::this_thread::sleep_for(std::chrono::microseconds(40));
        }
    }
};
}  // namespace

int mmx_host_stage_upload(const void* h_src, void* d_dst, const int64_t* regions, int32_t n_regions, int64_t nz,
                          int64_t ny, int64_t row_bytes, void* const* h_staging, int64_t staging_bytes, int32_t depth,
                          void* const* events, void* stream, int32_t device, int64_t* n_queued, const int32_t* cancel,
                          int32_t n_threads)
{
    if (!h_src || !d_dst || !regions || n_regions < 0 || nz < 0 || ny < 1 || row_bytes < 1 || !h_staging || depth < 1 ||
        !events || !n_queued || !cancel)
        return MMX_ERR_ARG;
    for (int k = 0; k < n_regions; ++k) {
        const int64_t* r = regions + 4 * (int64_t)k;
        if (r[0] < 0 || r[1] > nz || r[0] >= r[1] || r[2] < 0 || r[3] > ny || r[2] >= r[3] ||
            (r[1] - r[0]) * (r[3] - r[2]) * row_bytes > staging_bytes)
            return MMX_ERR_ARG;
    }
    if (hipSetDevice(device) != hipSuccess) return fail(hipGetLastError(), "hipSetDevice");
    const int T = std::max(1, std::min<int>(n_threads, 64));
    const int64_t plane = ny * row_bytes;
    std::atomic<int> status{MMX_OK};
    std::atomic<int> stop{0};
    spin_barrier go, filled;
    go.n = filled.n = T;
    const bool prof = getenv("MMX_STAGE_PROF") != nullptr;
    double t_wait = 0, t_fill = 0, t_queue = 0;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    auto work = [&](int t) {
        for (int k = 0; k < n_regions; ++k) {
            const int64_t* r = regions + 4 * (int64_t)k;
            const int which = k % depth;
            double t0 = 0;
            if (t == 0) {
                if (prof) t0 = now();
                // the leader: the DMA that last read this staging buffer must be through with it
                if (*(volatile const int32_t*)cancel) stop.store(1);
                else if (k >= depth && hipEventSynchronize((hipEvent_t)events[k - depth]) != hipSuccess) {
                    status.store(fail(hipGetLastError(), "hipEventSynchronize")); stop.store(1);
                }
                if (prof) { const double t1 = now(); t_wait += t1 - t0; t0 = t1; }
            }
            go.wait();
            if (stop.load()) return;
            const int64_t planes = r[1] - r[0], width = (r[3] - r[2]) * row_bytes;
            const int64_t a = planes * t / T, b = planes * (t + 1) / T;
            const char* src = (const char*)h_src + r[0] * plane + r[2] * row_bytes;
            char* dst = (char*)h_staging[which];
            for (int64_t z = a; z < b; ++z) std::memcpy(dst + z * width, src + z * plane, (size_t)width);
            filled.wait();
            if (t == 0) {
                if (prof) { const double t1 = now(); t_fill += t1 - t0; t0 = t1; }
                hipError_t e = hipMemcpy2DAsync((char*)d_dst + r[0] * plane + r[2] * row_bytes, (size_t)plane,
                                                h_staging[which], (size_t)width, (size_t)width, (size_t)planes,
                                                hipMemcpyHostToDevice, (hipStream_t)stream);
                if (e == hipSuccess) e = hipEventRecord((hipEvent_t)events[k], (hipStream_t)stream);
                if (e != hipSuccess) { status.store(fail(e, "staged upload")); stop.store(1); }
                else __atomic_store_n(n_queued, (int64_t)k + 1, __ATOMIC_RELEASE);
                if (prof) t_queue += now() - t0;
            }
        }
    };
    std::vector<std::thread> helpers;
    for (int t = 1; t < T; ++t) helpers.emplace_back(work, t);
    work(0);
    // (every thread, the leader included, passes go.wait() before it looks at `stop`: nobody is left in a barrier)
    for (auto& h : helpers) h.join();
    if (prof)
        fprintf(stderr, "mmx_host_stage_upload: %d regions, %d threads, %.1f ms: waiting for staging buffers %.1f, filling %.1f, "
                "queueing copies %.1f\n", n_regions, T, now() - t_begin, t_wait, t_fill, t_queue);
    return status.load();
}
