// Host-side table code of libmmx_hip.so (plain C++, no device work).
//
// mmx_host_prune_axis -- one axis of the cross-block duplicate pruning that the reference does
// with NumPy on the host (magmap/cv/stack_detect.py:679-861, StackPruner.prune_blobs_mp, with
// detector.remove_close_blobs :1000-1085 inside).  north_star keeps this de-duplication on the
// host; it is native here because at 3e5 blobs the NumPy formulation (a dozen full-table passes
// per axis) was the longest serial piece after the last kernel.
//
// Semantics per axis (see the Python docstring of prune_blobs_mp for the derivation):
//   the axis is tiled by regions  [pass 0][slab 0][pass 1][slab 1] ... [pass last];
//   in slab j rows tagged block j are masters, rows tagged j+1 are checked against them
//   (|d| <= tol on all three axes, integer compare), other generations are dropped;
//   every matched check row is removed; a matched master's abs coordinates become
//   rint((abs_master + abs_check) / 2) (round half to even) for its LAST matching check row
//   in table order (NumPy duplicate fancy-index assignment: last write wins);
//   new row order = pass sections in turn, then per slab its masters followed by the surviving
//   check rows, each group in the current table order.

#include <algorithm>
#include <atomic>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <limits>
#include <vector>

#include <unistd.h>

#include "../../include/mmx.h"

namespace {
// A small persistent pool: the tables have ~3e5 rows, so one parallel section is a fraction of a
// millisecond of work and creating threads per section would cost as much as the work itself.
// fork(): the child inherits the pool object but none of its threads (the reference's default start method is
// "fork"), so a parallel section there would wait for workers that do not exist.  The pool remembers the pid it
// was built in; in any other process every section runs on the calling thread.
// Waking: a small stack's sections are 20-100 us of work each and come in chains (resolve, overlap prune, tables,
// pruning: half a dozen within a few hundred microseconds), and a condition-variable wake-up of sixteen sleepers cost
// as much as the section (measured on a 4-block stack: the resolve step 0.20 ms with 16 threads, 0.13 with 4).  The
// workers therefore keep polling the section counter for `spin_us_` (150 us where there are cores to spare) after their last section before they go to sleep,
// and the caller polls the completion counter for as long before it does.
class pool {
public:
    static pool& get() { static pool p; return p; }
    int size() const { return owner_ == getpid() ? (int)workers_.size() + 1 : 1; }
    // run fn(t, n) for t = 0..n-1, n <= size(); the caller is thread 0
    void run(int n, const std::function<void(int, int)>& fn)
    {
        if (n <= 1) { fn(0, 1); return; }
        std::unique_lock<std::mutex> serial(serial_);          // one section at a time
        fn_.store(&fn);
        pending_.store(n - 1);
        // (section number and thread count change together: a worker always pairs a section with ITS thread count)
        state_.store(((state_.load() >> 8) + 1) << 8 | (uint64_t)n);
        if (sleepers_.load() > 0) {
            // (a worker that has counted itself a sleeper holds m_ until it waits: taking the lock orders this
            //  notification after its wait has begun)
            { std::lock_guard<std::mutex> lk(m_); }
            cv_.notify_all();
        }
        fn(0, n);
        if (pending_.load() != 0) {
            const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(spin_us_);
            for (int k = 0; spin_us_ && pending_.load() != 0; ++k) {
                relax();
                if ((k & 63) == 63 && std::chrono::steady_clock::now() > until) break;
            }
            if (pending_.load() != 0) {
                std::unique_lock<std::mutex> lk(m_);
                caller_sleeps_.store(true);
                done_.wait(lk, [&] { return pending_.load() == 0; });
                caller_sleeps_.store(false);
            }
        }
        fn_.store(nullptr);
    }
private:
    static void relax()
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
    }
    pool()
    {
        const unsigned hw = std::thread::hardware_concurrency();
        // (MMX_HOST_THREADS: 1 .. 64; the default of 16 is where the tables of the benchmark volume stop gaining)
        const char* env = getenv("MMX_HOST_THREADS");
        const unsigned cap = env ? (unsigned)std::max(1, std::min(64, atoi(env))) : 16u;
        const int n = (int)std::max(1u, std::min(cap, hw ? hw : 1u));
        owner_ = getpid();
        // (polling needs cores to spare: with as many pool threads as cores -- an 8-core container -- any other runnable
        //  thread preempts a poller or a worker and the section waits a scheduler slice for it: measured 3.2 against
        //  0.5 ms per back-to-back section there; MMX_HOST_SPIN_US overrides, 0 = always sleep)
        const char* spin = getenv("MMX_HOST_SPIN_US");
        spin_us_ = spin ? std::max(0, std::min(10000, atoi(spin))) : (hw >= 2u * (unsigned)n ? 150 : 0);
        for (int t = 1; t < n; ++t) workers_.emplace_back([this, t] { loop(t); });
    }
    ~pool()
    {
        if (owner_ != getpid()) {          // a forked child: the threads were never here
            for (auto& w : workers_) w.detach();
            return;
        }
        stop_.store(true);
        state_.store(((state_.load() >> 8) + 1) << 8);          // (a section for nobody)
        { std::lock_guard<std::mutex> lk(m_); }
        cv_.notify_all();
        for (auto& w : workers_) w.join();
    }
    void loop(int t)
    {
        uint64_t seen = 0;              // the last section number this worker has looked at
        for (;;) {
            // ---- the next section: poll for a while, then sleep
            const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(spin_us_);
            for (int k = 0; spin_us_ && (state_.load() >> 8) == seen; ++k) {
                relax();
                if ((k & 63) == 63 && std::chrono::steady_clock::now() > until) break;
            }
            if ((state_.load() >> 8) == seen) {
                std::unique_lock<std::mutex> lk(m_);
                sleepers_.fetch_add(1);
                cv_.wait(lk, [&] { return (state_.load() >> 8) != seen; });
                sleepers_.fetch_sub(1);
            }
            const uint64_t st = state_.load();
            seen = st >> 8;
            if (stop_.load()) return;
            const int n = (int)(st & 0xff);
            // (a section this worker has no part in may be over already -- it then finds the next one at once; a
            //  section that counts on it cannot end, nor fn_ change, before it has reported)
            if (t >= n) continue;
            const std::function<void(int, int)>* fn = fn_.load();
            (*fn)(t, n);
            if (pending_.fetch_sub(1) == 1 && caller_sleeps_.load()) {
                { std::lock_guard<std::mutex> lk(m_); }
                done_.notify_one();
            }
        }
    }
    std::vector<std::thread> workers_;
    pid_t owner_ = 0;
    int spin_us_ = 0;
    std::mutex m_, serial_;
    std::condition_variable cv_, done_;
    std::atomic<const std::function<void(int, int)>*> fn_{nullptr};
    std::atomic<int> pending_{0}, sleepers_{0};
    std::atomic<uint64_t> state_{0};                // section number << 8 | threads of the section
    std::atomic<bool> stop_{false}, caller_sleeps_{false};
};

template <typename F>
void parallel(int n_threads, F fn)
{
    pool& p = pool::get();
    const std::function<void(int, int)> f = fn;
    p.run(std::min(n_threads, p.size()), f);
}

int host_threads(int64_t n_rows)
{
    if (n_rows < 20000) return 1;
    return pool::get().size();
}
}  // namespace

namespace {
// One axis of the pruning.  `own_lo` / `own_hi`: only rows whose id lies in [own_lo, own_hi) count towards the
// statistics (a table that holds a region's own rows plus a halo of its neighbours' rows: every row is counted by
// exactly one region); `group_out` (optional, indexed like `cur`): the group id each row was given (-1: dropped).
int prune_axis_impl(const int32_t* zyx, const int32_t* tag, double* abs_zyx,
                    const int64_t* cur, int64_t n_cur, int axis, int n_sections,
                    const double* bounds, double last_end, const int32_t tol[3],
                    const double* nxt_lo, const double* nxt_hi,
                    int64_t* out_cur, int64_t* out_n,
                    int64_t* n_slab, int64_t* n_after, int64_t* n_next,
                    int64_t own_lo, int64_t own_hi, int32_t* group_out)
{
    if ((n_cur && (!zyx || !tag || !abs_zyx || !cur || !out_cur)) || !bounds || !tol || !out_n ||
        !n_slab || !n_after || !n_next || axis < 0 || axis > 2 || n_sections < 2 || n_cur < 0)
        return MMX_ERR_ARG;
    constexpr bool prof = false;          // (the split of every axis step on stderr: a debugging aid, compiled out)
    auto tnow = [] { return std::chrono::steady_clock::now(); };
    auto t0 = tnow();
    const int n_regions = 2 * n_sections - 1;
    const int n_slabs = n_sections - 1;
    // group ids: pass j -> j ; slab j master -> n_sections + 2j ; slab j kept check -> n_sections + 2j + 1
    const int n_groups = n_sections + 2 * n_slabs;
    std::vector<int32_t> group((size_t)n_cur, -1);
    for (int j = 0; j < n_slabs; ++j) { n_slab[j] = 0; n_after[j] = 0; n_next[j] = 0; }

    // ---- classification, in contiguous chunks of the current order (one per thread)
    const int T = host_threads(n_cur);
    struct part { std::vector<std::vector<int64_t>> masters, checks; std::vector<int64_t> n_slab, n_next; };
    std::vector<part> parts((size_t)T);
    parallel(T, [&](int t, int nt) {
        part& P = parts[(size_t)t];
        P.masters.assign((size_t)n_slabs, {});
        P.checks.assign((size_t)n_slabs, {});
        P.n_slab.assign((size_t)n_slabs, 0);
        P.n_next.assign((size_t)n_slabs, 0);
        const int64_t lo = n_cur * t / nt, hi = n_cur * (t + 1) / nt;
        for (int64_t i = lo; i < hi; ++i) {
            const int64_t row = cur[i];
            const double pos = (double)zyx[3 * row + axis];
            const bool own = row >= own_lo && row < own_hi;
            if (own)
                for (int j = 0; j < n_slabs; ++j)
                    if (nxt_lo[j] == nxt_lo[j] && pos >= nxt_lo[j] && pos < nxt_hi[j]) ++P.n_next[(size_t)j];
            // region = (number of bounds <= pos) - 1   (np.searchsorted(bounds, pos, side="right") - 1)
            const int region = (int)(std::upper_bound(bounds, bounds + n_regions, pos) - bounds) - 1;
            if (region < 0 || !(pos < last_end)) continue;
            const int sec = region >> 1;
            if ((region & 1) == 0) { group[(size_t)i] = sec; continue; }
            if (own) ++P.n_slab[(size_t)sec];
            const int32_t tg = tag[3 * row + axis];
            if (tg == sec) { group[(size_t)i] = n_sections + 2 * sec; P.masters[(size_t)sec].push_back(i); }
            else if (tg == sec + 1) { group[(size_t)i] = n_sections + 2 * sec + 1; P.checks[(size_t)sec].push_back(i); }
        }
    });
    auto t1 = tnow();
    std::vector<std::vector<int64_t>> masters((size_t)n_slabs), checks((size_t)n_slabs);
    for (int j = 0; j < n_slabs; ++j)
        for (int t = 0; t < T; ++t) {           // chunk order == table order
            const part& P = parts[(size_t)t];
            masters[(size_t)j].insert(masters[(size_t)j].end(), P.masters[(size_t)j].begin(), P.masters[(size_t)j].end());
            checks[(size_t)j].insert(checks[(size_t)j].end(), P.checks[(size_t)j].begin(), P.checks[(size_t)j].end());
            n_slab[j] += P.n_slab[(size_t)j];
            n_next[j] += P.n_next[(size_t)j];
        }

    auto t2 = tnow();
    // ---- matching: the slabs are independent (disjoint rows), one at a time per thread
    // the axis to sort the check rows on: any of the two other axes works
    const int sa = axis == 0 ? 1 : 0;
    // (a) per slab: check rows sorted on `sa` (slabs in parallel); (b) every master of every slab looks its
    // window up (masters in parallel: each writes only its own `last`, `hit` bytes are set, never cleared);
    // (c) per slab: averages from the values before any update of this stage, then the removals.
    std::vector<std::vector<std::pair<int32_t, int64_t>>> orders((size_t)n_slabs);   // (coordinate on sa, position in checks[j])
    std::vector<std::vector<char>> hits((size_t)n_slabs);
    std::vector<std::vector<int64_t>> lasts((size_t)n_slabs);
    std::vector<int64_t> m_start((size_t)n_slabs + 1, 0);
    for (int j = 0; j < n_slabs; ++j) {
        const bool live = !masters[(size_t)j].empty() && !checks[(size_t)j].empty();
        m_start[(size_t)j + 1] = m_start[(size_t)j] + (live ? (int64_t)masters[(size_t)j].size() : 0);
    }
    parallel(std::min(T, n_slabs), [&](int t, int nt) {
        for (int j = t; j < n_slabs; j += nt) {
            const auto& M = masters[(size_t)j];
            const auto& C = checks[(size_t)j];
            if (M.empty() || C.empty()) continue;
            auto& order = orders[(size_t)j];
            order.reserve(C.size());
            for (size_t k = 0; k < C.size(); ++k) order.emplace_back(zyx[3 * cur[C[k]] + sa], (int64_t)k);
            std::sort(order.begin(), order.end());
            hits[(size_t)j].assign(C.size(), 0);
            lasts[(size_t)j].assign(M.size(), -1);
        }
    });
    const int64_t n_masters = m_start[(size_t)n_slabs];
    parallel(n_masters > 2000 ? T : 1, [&](int t, int nt) {
        const int64_t lo_m = n_masters * t / nt, hi_m = n_masters * (t + 1) / nt;
        int j = 0;
        for (int64_t g = lo_m; g < hi_m; ++g) {
            while (g >= m_start[(size_t)j + 1]) ++j;
            const size_t m = (size_t)(g - m_start[(size_t)j]);
            const auto& M = masters[(size_t)j];
            const auto& C = checks[(size_t)j];
            const auto& order = orders[(size_t)j];
            char* hit = hits[(size_t)j].data();
            const int32_t* mz = zyx + 3 * cur[M[m]];
            const int32_t lo = mz[sa] - tol[sa], hi = mz[sa] + tol[sa];
            int64_t last = -1;
            auto it = std::lower_bound(order.begin(), order.end(), std::make_pair(lo, (int64_t)-1));
            for (; it != order.end() && it->first <= hi; ++it) {
                const int32_t* cz = zyx + 3 * cur[C[(size_t)it->second]];
                if (std::abs(mz[0] - cz[0]) <= tol[0] && std::abs(mz[1] - cz[1]) <= tol[1] &&
                    std::abs(mz[2] - cz[2]) <= tol[2]) {
                    hit[(size_t)it->second] = 1;
                    if (it->second > last) last = it->second;
                }
            }
            lasts[(size_t)j][m] = last;
        }
    });
    parallel(std::min(T, n_slabs), [&](int t, int nt) {
        std::vector<double> new_abs;
        for (int j = t; j < n_slabs; j += nt) {
            const auto& M = masters[(size_t)j];
            const auto& C = checks[(size_t)j];
            auto is_own = [&](int64_t i) { const int64_t row = cur[i]; return row >= own_lo && row < own_hi; };
            int64_t kept = 0, own_masters = 0;
            for (size_t k = 0; k < C.size(); ++k) kept += is_own(C[k]);
            for (size_t m = 0; m < M.size(); ++m) own_masters += is_own(M[m]);
            if (!M.empty() && !C.empty()) {
                const auto& last = lasts[(size_t)j];
                const auto& hit = hits[(size_t)j];
                new_abs.assign(M.size() * 3, 0.0);
                for (size_t m = 0; m < M.size(); ++m) {
                    if (last[m] < 0) continue;
                    const double* am = abs_zyx + 3 * cur[M[m]];
                    const double* ac = abs_zyx + 3 * cur[C[(size_t)last[m]]];
                    for (int a = 0; a < 3; ++a) new_abs[3 * m + a] = std::nearbyint((am[a] + ac[a]) / 2);
                }
                for (size_t m = 0; m < M.size(); ++m) {
                    if (last[m] < 0) continue;
                    double* am = abs_zyx + 3 * cur[M[m]];
                    for (int a = 0; a < 3; ++a) am[a] = new_abs[3 * m + a];
                }
                for (size_t k = 0; k < C.size(); ++k)
                    if (hit[k]) { group[(size_t)C[k]] = -1; kept -= is_own(C[k]); }
            }
            n_after[j] = own_masters + kept;
        }
    });

    auto t3 = tnow();
    // stable counting sort of the surviving rows by group: per-chunk histograms, one prefix over
    // (group, chunk), then every chunk scatters its own rows
    std::vector<std::vector<int64_t>> hist((size_t)T, std::vector<int64_t>((size_t)n_groups, 0));
    parallel(T, [&](int t, int nt) {
        auto& h = hist[(size_t)t];
        const int64_t lo = n_cur * t / nt, hi = n_cur * (t + 1) / nt;
        for (int64_t i = lo; i < hi; ++i)
            if (group[(size_t)i] >= 0) ++h[(size_t)group[(size_t)i]];
    });
    int64_t total = 0;
    for (int g = 0; g < n_groups; ++g)
        for (int t = 0; t < T; ++t) {
            const int64_t c = hist[(size_t)t][(size_t)g];
            hist[(size_t)t][(size_t)g] = total;
            total += c;
        }
    *out_n = total;
    parallel(T, [&](int t, int nt) {
        auto& h = hist[(size_t)t];
        const int64_t lo = n_cur * t / nt, hi = n_cur * (t + 1) / nt;
        for (int64_t i = lo; i < hi; ++i) {
            const int g = group[(size_t)i];
            if (g >= 0) out_cur[h[(size_t)g]++] = cur[i];
            if (group_out) group_out[i] = g;
        }
    });
    if (prof) {
        auto us = [](auto a, auto b) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
        auto t4 = tnow();
        fprintf(stderr, "prune axis %d: n %ld classify %ld us, merge %ld, match %ld, sort %ld\n", axis, (long)n_cur,
                us(t0, t1), us(t1, t2), us(t2, t3), us(t3, t4));
    }
    return MMX_OK;
}
}  // namespace

extern "C" int mmx_host_prune_axis(const int32_t* zyx, const int32_t* tag, double* abs_zyx,
                                   const int64_t* cur, int64_t n_cur, int axis, int n_sections,
                                   const double* bounds, double last_end, const int32_t tol[3],
                                   const double* nxt_lo, const double* nxt_hi,
                                   int64_t* out_cur, int64_t* out_n,
                                   int64_t* n_slab, int64_t* n_after, int64_t* n_next)
{
    return prune_axis_impl(zyx, tag, abs_zyx, cur, n_cur, axis, n_sections, bounds, last_end, tol, nxt_lo, nxt_hi,
                           out_cur, out_n, n_slab, n_after, n_next, INT64_MIN, INT64_MAX, nullptr);
}

// All three axes of the pruning for one REGION of the stack: a table that holds the region's own rows (ids
// [own_lo, own_hi)) between the rows of its neighbours that lie within reach of them (the halo), all in the merged
// table's order.  Every decision of the reference's three passes about a row depends only on rows within
// 3 x tol of it (a pass looks tol far, and what it finds depends on the passes before), so the region's own rows
// get the verdicts and averaged coordinates the whole-table passes give them as long as the halo is that wide;
// halo rows may come out wrong and are dropped from the output.  A surviving row's place in the whole table's final
// order is that of its key among all survivors -- the passes are stable sorts by group, so the final order is the
// stable sort by (group on axis 2, group on axis 1, group on axis 0) -- ties between regions in region order.
//   cur / n_cur  : the rows of one channel, table order (own and halo)
//   n_sections[a] <= 1 : axis a has one block, no pass (group 0)
//   bounds[a], nxt_lo[a], nxt_hi[a], last_end[a] : as for mmx_host_prune_axis, per axis
//   out_rows / out_keys / out_n : the own survivors in final order and their keys; abs_zyx holds their averaged
//                  coordinates afterwards (and garbage-free but possibly unfinished values for halo rows)
//   n_slab / n_after / n_next : [3][max_slabs] statistics over OWN rows (row pitch `stat_ld`)
namespace {
int prune_region_impl(const int32_t* zyx, const int32_t* tag, double* abs_zyx,
                      const int64_t* cur, int64_t n_cur, int64_t own_lo, int64_t own_hi,
                      const int32_t n_sections[3], const double* const bounds[3],
                      const double last_end[3], const int32_t tol[3],
                      const double* const nxt_lo[3], const double* const nxt_hi[3],
                      int64_t* out_rows, int64_t* out_keys, int64_t* out_n,
                      int64_t* n_slab, int64_t* n_after, int64_t* n_next, int64_t stat_ld)
{
    for (int a = 0; a < 3; ++a)
        if (n_sections[a] > 1 && (stat_ld < n_sections[a] - 1 || !bounds[a] || !nxt_lo[a] || !nxt_hi[a])) return MMX_ERR_ARG;
    int64_t max_row = -1;
    for (int64_t i = 0; i < n_cur; ++i) { if (cur[i] < 0) return MMX_ERR_ARG; max_row = std::max(max_row, cur[i]); }
    std::vector<int64_t> key(out_keys ? (size_t)(max_row + 1) : 0, 0);
    std::vector<int64_t> a_rows(cur, cur + n_cur), b_rows((size_t)n_cur);
    std::vector<int32_t> group(out_keys ? (size_t)n_cur : 0);
    int64_t n = n_cur, stride = 1;
    for (int a = 0; a < 3; ++a) {
        if (n_sections[a] <= 1) continue;
        int64_t n_out = 0;
        const int st = prune_axis_impl(zyx, tag, abs_zyx, a_rows.data(), n, a, n_sections[a], bounds[a], last_end[a], tol,
                                       nxt_lo[a], nxt_hi[a], b_rows.data(), &n_out, n_slab + a * stat_ld,
                                       n_after + a * stat_ld, n_next + a * stat_ld, own_lo, own_hi,
                                       out_keys ? group.data() : nullptr);
        if (st != MMX_OK) return st;
        if (out_keys) {
            const int64_t* rows = a_rows.data();
            parallel(host_threads(n), [&](int t, int nt) {
                for (int64_t i = n * t / nt; i < n * (t + 1) / nt; ++i)       // (a row appears once in `rows`)
                    if (group[(size_t)i] >= 0) key[(size_t)rows[i]] += stride * group[(size_t)i];
            });
        }
        stride *= (int64_t)n_sections[a] + 2 * ((int64_t)n_sections[a] - 1);
        a_rows.swap(b_rows);
        n = n_out;
    }
    int64_t k = 0;
    if (own_lo == INT64_MIN && own_hi == INT64_MAX && !out_keys) {       // the whole table: every survivor, no keys
        if (n) std::memcpy(out_rows, a_rows.data(), (size_t)n * sizeof(int64_t));      // (an empty vector's data() may be null)
        k = n;
    } else {
        for (int64_t i = 0; i < n; ++i) {
            const int64_t row = a_rows[(size_t)i];
            if (row >= own_lo && row < own_hi) { out_rows[k] = row; if (out_keys) out_keys[k] = key[(size_t)row]; ++k; }
        }
    }
    *out_n = k;
    return MMX_OK;
}
}  // namespace

extern "C" int mmx_host_prune_region(const int32_t* zyx, const int32_t* tag, double* abs_zyx,
                                     const int64_t* cur, int64_t n_cur, int64_t own_lo, int64_t own_hi,
                                     const int32_t n_sections[3], const double* const bounds[3],
                                     const double last_end[3], const int32_t tol[3],
                                     const double* const nxt_lo[3], const double* const nxt_hi[3],
                                     int64_t* out_rows, int64_t* out_keys, int64_t* out_n,
                                     int64_t* n_slab, int64_t* n_after, int64_t* n_next, int64_t stat_ld)
{
    if ((n_cur && (!zyx || !tag || !abs_zyx || !cur || !out_rows)) || n_cur < 0 || !n_sections || !bounds ||
        !last_end || !tol || !nxt_lo || !nxt_hi || !out_n || !n_slab || !n_after || !n_next)
        return MMX_ERR_ARG;
    return prune_region_impl(zyx, tag, abs_zyx, cur, n_cur, own_lo, own_hi, n_sections, bounds, last_end, tol, nxt_lo,
                             nxt_hi, out_rows, out_keys, out_n, n_slab, n_after, n_next, stat_ld);
}

// The same for a region whose rows are still in the merged table: `parts` are row ranges of that table in ascending
// order, part `own_part` the region itself, the others its neighbours, of which only the rows inside the box
// [box_lo, box_hi) (the region's extent plus the reach of the pruning) take part.  The local table is assembled
// here (the merged table and its abs column are not written to), every channel of `channels` is pruned in turn
// (`chan`: the channel column of the merged table, row pitch chan_ld doubles; NULL: every row belongs to
// channels[0]) and the region's survivors come back as rows of the merged table with their keys (channel position
// x n_keys + key) and their averaged coordinates.  Statistics: [n_channels][3][stat_ld].
extern "C" int mmx_host_prune_parts(const int32_t* zyx, const int32_t* tag, const double* abs_zyx,
                                    const double* chan, int64_t chan_ld,
                                    const int64_t* parts, int n_parts, int own_part,
                                    const int32_t box_lo[3], const int32_t box_hi[3],
                                    const double* channels, int n_channels,
                                    const int32_t n_sections[3], const double* const bounds[3],
                                    const double last_end[3], const int32_t tol[3],
                                    const double* const nxt_lo[3], const double* const nxt_hi[3], int64_t n_keys,
                                    int64_t* out_ids, int64_t* out_keys, double* out_abs, int64_t* out_n,
                                    int64_t* n_slab, int64_t* n_after, int64_t* n_next, int64_t stat_ld)
{
    if (!parts || n_parts < 1 || own_part < 0 || own_part >= n_parts || !box_lo || !box_hi || !channels || n_channels < 1 ||
        !n_sections || !bounds || !last_end || !tol || !nxt_lo || !nxt_hi || !out_n || !n_slab || !n_after || !n_next)
        return MMX_ERR_ARG;
    // ---- the local table: ids of the merged table's rows, own rows between the neighbours' rows within reach
    // (the parts come in the order of the LOCAL table -- for regions of one arena that is ascending row order; a rank's
    //  table keeps the halo rows it received from other ranks behind its own rows and lists them before / after)
    std::vector<int64_t> ids;
    int64_t own_lo = 0, own_hi = 0;
    for (int p = 0; p < n_parts; ++p) {
        const int64_t a = parts[2 * p], b = parts[2 * p + 1];
        if (a < 0 || b < a) return MMX_ERR_ARG;
        for (int q = 0; q < p; ++q)          // no row twice
            if (a < parts[2 * q + 1] && parts[2 * q] < b && b > a && parts[2 * q + 1] > parts[2 * q]) return MMX_ERR_ARG;
        if (b > a && (!zyx || !tag || !abs_zyx)) return MMX_ERR_ARG;
        if (p == own_part) {
            own_lo = (int64_t)ids.size();
            for (int64_t r = a; r < b; ++r) ids.push_back(r);
            own_hi = (int64_t)ids.size();
        } else {
            for (int64_t r = a; r < b; ++r) {
                const int32_t* c = zyx + 3 * r;
                if (c[0] >= box_lo[0] && c[0] < box_hi[0] && c[1] >= box_lo[1] && c[1] < box_hi[1] &&
                    c[2] >= box_lo[2] && c[2] < box_hi[2])
                    ids.push_back(r);
            }
        }
    }
    const int64_t n = (int64_t)ids.size();
    if (own_hi > own_lo && (!out_ids || !out_keys || !out_abs)) return MMX_ERR_ARG;
    std::vector<int32_t> lz((size_t)n * 3), lt((size_t)n * 3);
    std::vector<double> la((size_t)n * 3);
    for (int64_t i = 0; i < n; ++i) {
        std::memcpy(lz.data() + 3 * i, zyx + 3 * ids[(size_t)i], 3 * sizeof(int32_t));
        std::memcpy(lt.data() + 3 * i, tag + 3 * ids[(size_t)i], 3 * sizeof(int32_t));
        std::memcpy(la.data() + 3 * i, abs_zyx + 3 * ids[(size_t)i], 3 * sizeof(double));
    }
    std::vector<int64_t> cur, rows((size_t)n), keys((size_t)n);
    int64_t k_out = 0;
    for (int ci = 0; ci < n_channels; ++ci) {
        cur.clear();
        for (int64_t i = 0; i < n; ++i)
            if (!chan || chan[ids[(size_t)i] * chan_ld] == channels[ci]) cur.push_back(i);
        int64_t k = 0;
        const int st = prune_region_impl(lz.data(), lt.data(), la.data(), cur.data(), (int64_t)cur.size(), own_lo, own_hi,
                                         n_sections, bounds, last_end, tol, nxt_lo, nxt_hi, rows.data(), keys.data(), &k,
                                         n_slab + (int64_t)ci * 3 * stat_ld, n_after + (int64_t)ci * 3 * stat_ld,
                                         n_next + (int64_t)ci * 3 * stat_ld, stat_ld);
        if (st != MMX_OK) return st;
        for (int64_t i = 0; i < k; ++i) {
            out_ids[k_out] = ids[(size_t)rows[(size_t)i]];
            out_keys[k_out] = keys[(size_t)i] + (int64_t)ci * n_keys;
            std::memcpy(out_abs + 3 * k_out, la.data() + 3 * rows[(size_t)i], 3 * sizeof(double));
            ++k_out;
        }
    }
    *out_n = k_out;
    return MMX_OK;
}

// Final table of a stack from the survivors of its regions: the stable sort of the concatenated rows (regions in
// order) by key.  `keys` are small (the product of the three group counts): a counting sort.
//   rows : [n][ld] float64, the first n_cols columns are copied; out : [n][n_cols]
extern "C" int mmx_host_merge_by_key(const double* rows, int64_t ld, const int64_t* keys, int64_t n, int64_t n_keys,
                                     int64_t n_cols, double* out)
{
    // keys == nullptr: a row's key is the value in its column n_cols (how the ranks' survivors arrive: the key rides
    // behind the columns) -- no separate key array has to be pulled out of the table first
    if (n < 0 || n_keys < 1 || n_cols < 1 || n_cols > ld || (!keys && n_cols >= ld) || (n && (!rows || !out)))
        return MMX_ERR_ARG;
    if (n_keys > (int64_t(1) << 26)) return MMX_ERR_UNSUPPORTED;
    if (n == 0) return MMX_OK;
    // threaded counting sort: every thread counts the keys of one contiguous run of rows, the runs' counts are laid end
    // to end per key, every thread places its own rows (rows of one key keep their order)
    int T = host_threads(n);
    if ((int64_t)T * n_keys > (int64_t(1) << 24)) T = 1;
    std::vector<int64_t> at((size_t)T * (size_t)n_keys, 0);
    std::vector<int> bad((size_t)T, 0);
    auto key_of = [&](int64_t i) -> int64_t {
        if (keys) return keys[i];
        const double v = rows[i * ld + n_cols];
        return (v >= 0.0 && v < 9.0e15) ? (int64_t)v : -1;
    };
    parallel(T, [&](int t, int) {
        int64_t* h = at.data() + (size_t)t * (size_t)n_keys;
        for (int64_t i = n * t / T; i < n * (t + 1) / T; ++i) {
            const int64_t k = key_of(i);
            if (k < 0 || k >= n_keys) { bad[(size_t)t] = 1; break; }
            ++h[k];
        }
    });
    for (int t = 0; t < T; ++t)
        if (bad[(size_t)t]) return MMX_ERR_ARG;
    {
        int64_t run = 0;
        for (int64_t k = 0; k < n_keys; ++k)
            for (int t = 0; t < T; ++t) {
                int64_t& c = at[(size_t)t * (size_t)n_keys + (size_t)k];
                const int64_t here = c;
                c = run;
                run += here;
            }
    }
    parallel(T, [&](int t, int) {
        int64_t* pos = at.data() + (size_t)t * (size_t)n_keys;
        for (int64_t i = n * t / T; i < n * (t + 1) / T; ++i)
            std::memcpy(out + pos[key_of(i)]++ * n_cols, rows + i * ld, (size_t)n_cols * sizeof(double));
    });
    return MMX_OK;
}

// mmx_host_merge_by_key on the CONCATENATION of n_parts row blocks (every rank's survivors as an all_gather leaves
// them: padded to the longest block, so not contiguous), keys in column n_cols of each row -- without the 20 MB copy
// that would make them one array.  Blocks are cut into segments so that every thread has work whatever their number.
extern "C" int mmx_host_merge_parts_by_key(const double* const* parts, const int64_t* n_rows, int32_t n_parts, int64_t ld,
                                           int64_t n_keys, int64_t n_cols, double* out, int64_t out_rows)
{
    if (n_parts < 0 || n_keys < 1 || n_cols < 1 || n_cols >= ld || (n_parts && (!parts || !n_rows))) return MMX_ERR_ARG;
    if (n_keys > (int64_t(1) << 26)) return MMX_ERR_UNSUPPORTED;
    int64_t n = 0;
    for (int p = 0; p < n_parts; ++p) {
        if (n_rows[p] < 0 || (n_rows[p] && !parts[p])) return MMX_ERR_ARG;
        n += n_rows[p];
    }
    if (n != out_rows || (n && !out)) return MMX_ERR_ARG;
    if (n == 0) return MMX_OK;
    int T = host_threads(n);
    if ((int64_t)T * n_keys > (int64_t(1) << 24)) T = 1;
    struct seg { const double* rows; int64_t n; };
    std::vector<seg> segs;
    const int64_t piece = std::max<int64_t>(1, (n + 4 * T - 1) / (4 * T));
    for (int p = 0; p < n_parts; ++p)
        for (int64_t a = 0; a < n_rows[p]; a += piece)
            segs.push_back(seg{parts[p] + a * ld, std::min(piece, n_rows[p] - a)});
    const int S = (int)segs.size();
    std::vector<int> first((size_t)T + 1, S);
    {
        int64_t seen = 0;
        int t = 0;
        first[0] = 0;
        for (int q = 0; q < S; ++q) {
            while (t + 1 < T && seen >= n * (t + 1) / T) first[(size_t)++t] = q;
            seen += segs[(size_t)q].n;
        }
    }
    auto key_of = [&](const double* row) -> int64_t {
        const double v = row[n_cols];
        return (v >= 0.0 && v < 9.0e15) ? (int64_t)v : -1;
    };
    std::vector<int64_t> at((size_t)T * (size_t)n_keys, 0);
    std::vector<int> bad((size_t)T, 0);
    parallel(T, [&](int t, int) {
        int64_t* h = at.data() + (size_t)t * (size_t)n_keys;
        for (int q = first[(size_t)t]; q < first[(size_t)t + 1] && !bad[(size_t)t]; ++q)
            for (int64_t i = 0; i < segs[(size_t)q].n; ++i) {
                const int64_t k = key_of(segs[(size_t)q].rows + i * ld);
                if (k < 0 || k >= n_keys) { bad[(size_t)t] = 1; break; }
                ++h[k];
            }
    });
    for (int t = 0; t < T; ++t)
        if (bad[(size_t)t]) return MMX_ERR_ARG;
    {
        int64_t run = 0;
        for (int64_t k = 0; k < n_keys; ++k)
            for (int t = 0; t < T; ++t) {
                int64_t& c = at[(size_t)t * (size_t)n_keys + (size_t)k];
                const int64_t here = c;
                c = run;
                run += here;
            }
    }
    parallel(T, [&](int t, int) {
        int64_t* pos = at.data() + (size_t)t * (size_t)n_keys;
        for (int q = first[(size_t)t]; q < first[(size_t)t + 1]; ++q)
            for (int64_t i = 0; i < segs[(size_t)q].n; ++i) {
                const double* row = segs[(size_t)q].rows + i * ld;
                std::memcpy(out + pos[key_of(row)]++ * n_cols, row, (size_t)n_cols * sizeof(double));
            }
    });
    return MMX_OK;
}

// The same for survivors that still live in the merged table: row ids[i] of `table` (its first n_cols columns), the
// three abs columns replaced by abs_rows[i], written to its place by key.
extern "C" int mmx_host_gather_by_key(const double* table, int64_t ld, const int64_t* ids, const int64_t* keys,
                                      int64_t n, int64_t n_keys, int64_t n_cols, const double* abs_rows,
                                      const int32_t abs_cols[3], double* out)
{
    if (n < 0 || n_keys < 1 || n_cols < 1 || n_cols > ld || !abs_cols || (n && (!table || !ids || !keys || !out || !abs_rows)))
        return MMX_ERR_ARG;
    if (n_keys > (int64_t(1) << 26)) return MMX_ERR_UNSUPPORTED;
    for (int a = 0; a < 3; ++a)
        if (abs_cols[a] < 0 || abs_cols[a] >= n_cols) return MMX_ERR_ARG;
    std::vector<int64_t> at((size_t)n_keys + 1, 0);
    for (int64_t i = 0; i < n; ++i) {
        if (keys[i] < 0 || keys[i] >= n_keys || ids[i] < 0) return MMX_ERR_ARG;
        ++at[(size_t)keys[i] + 1];
    }
    for (int64_t k = 0; k < n_keys; ++k) at[(size_t)k + 1] += at[(size_t)k];
    std::vector<int64_t> dst((size_t)n);
    for (int64_t i = 0; i < n; ++i) dst[(size_t)i] = at[(size_t)keys[i]]++;
    parallel(host_threads(n), [&](int t, int nt) {
        const int64_t lo = n * t / nt, hi = n * (t + 1) / nt;
        for (int64_t i = lo; i < hi; ++i) {
            double* o = out + dst[(size_t)i] * n_cols;
            std::memcpy(o, table + ids[i] * ld, (size_t)n_cols * sizeof(double));
            for (int a = 0; a < 3; ++a) o[abs_cols[a]] = abs_rows[3 * i + a];
        }
    });
    return MMX_OK;
}

// The merge of the regions' survivor lists with the table leaving in the caller's FINAL column layout (the reference's
// last two steps on the pruned table, magmap/cv/stack_detect.py:455-470: rel <- abs, abs and unnamed columns dropped --
// two more passes over a 3e5-row table when done afterwards): out[k][j] = table[ids][src_cols[j]], then
// out[k][abs_dst0 .. +3] = abs_rows; rows in key order, equal keys in the order of the lists.  The `n_parts` lists (the
// regions of a stack pruned one by one) are taken as they are
// -- concatenating them is 12 MB of copies for 3e5 rows, a millisecond of the step's tail -- and
// the counting sort itself threaded: every thread counts the keys of a contiguous run of parts, the runs' counts are
// laid end to end per key, and every thread places its own rows (the order within a key stays the lists' order).
extern "C" int mmx_host_gather_parts_by_key_final(const double* table, int64_t ld, int32_t n_parts,
                                                  const int64_t* const* ids, const int64_t* const* keys,
                                                  const double* const* abs_rows, const int64_t* n_rows, int64_t n_keys,
                                                  const int32_t* src_cols, int32_t n_out, int32_t abs_dst0,
                                                  double* out, int64_t out_rows)
{
    return mmx_host_gather_parts_by_key_split(table, ld, n_parts, ids, keys, abs_rows, n_rows, n_keys, src_cols, n_out,
                                              abs_dst0, out, out_rows, n_out, nullptr);
}

// ... with the output in TWO tables: columns [0, n_main) of the layout into `out` (row pitch n_main), the remaining
// n_out - n_main into `out_rest` (row pitch n_out - n_main) -- a stack detected with co-localisation: the eight final
// columns as one contiguous table, the columns its flags are read from (magmap/cv/stack_detect.py:463-464: columns 10
// .. 10 + C of the pruned table) beside it, in the one pass that gathers the rows.  out_rest NULL: n_main = n_out.
extern "C" int mmx_host_gather_parts_by_key_split(const double* table, int64_t ld, int32_t n_parts,
                                                  const int64_t* const* ids, const int64_t* const* keys,
                                                  const double* const* abs_rows, const int64_t* n_rows, int64_t n_keys,
                                                  const int32_t* src_cols, int32_t n_out, int32_t abs_dst0,
                                                  double* out, int64_t out_rows, int32_t n_main, double* out_rest)
{
    if (n_parts < 0 || n_keys < 1 || n_out < 3 || n_out > 64 || !src_cols || abs_dst0 < 0 || abs_dst0 + 3 > n_out ||
        (n_parts && (!ids || !keys || !abs_rows || !n_rows)))
        return MMX_ERR_ARG;
    if (!out_rest) n_main = n_out;
    if (n_main < abs_dst0 + 3 || n_main > n_out || (out_rest && n_main == n_out)) return MMX_ERR_ARG;
    const int n_rest = n_out - n_main;
    if (n_keys > (int64_t(1) << 26)) return MMX_ERR_UNSUPPORTED;
    for (int j = 0; j < n_out; ++j)
        if (src_cols[j] < 0 || src_cols[j] >= ld) return MMX_ERR_ARG;
    int64_t n = 0;
    for (int p = 0; p < n_parts; ++p) {
        if (n_rows[p] < 0 || (n_rows[p] && (!ids[p] || !keys[p] || !abs_rows[p]))) return MMX_ERR_ARG;
        n += n_rows[p];
    }
    if (n != out_rows || (n && (!table || !out))) return MMX_ERR_ARG;
    if (n == 0) return MMX_OK;
    // runs of parts with about the same number of rows each (a thread's run may be empty)
    int T = host_threads(n);
    if ((int64_t)T * n_keys > (int64_t(1) << 24)) T = 1;
    std::vector<int> first((size_t)T + 1, n_parts);
    {
        int64_t seen = 0;
        int t = 0;
        first[0] = 0;
        for (int p = 0; p < n_parts; ++p) {
            while (t + 1 < T && seen >= n * (t + 1) / T) first[(size_t)++t] = p;
            seen += n_rows[p];
        }
        for (++t; t <= T; ++t) first[(size_t)t] = n_parts;
    }
    std::vector<int64_t> at((size_t)T * (size_t)n_keys, 0);
    std::vector<int> bad((size_t)T, 0);
    parallel(T, [&](int t, int) {
        int64_t* h = at.data() + (size_t)t * (size_t)n_keys;
        for (int p = first[(size_t)t]; p < first[(size_t)t + 1]; ++p)
            for (int64_t i = 0; i < n_rows[p]; ++i) {
                const int64_t k = keys[p][i];
                if (k < 0 || k >= n_keys || ids[p][i] < 0) { bad[(size_t)t] = 1; break; }
                ++h[k];
            }
    });
    for (int t = 0; t < T; ++t)
        if (bad[(size_t)t]) return MMX_ERR_ARG;
    {
        int64_t run = 0;
        for (int64_t k = 0; k < n_keys; ++k)
            for (int t = 0; t < T; ++t) {
                int64_t& c = at[(size_t)t * (size_t)n_keys + (size_t)k];
                const int64_t here = c;
                c = run;
                run += here;
            }
    }
    parallel(T, [&](int t, int) {
        int64_t* pos = at.data() + (size_t)t * (size_t)n_keys;
        for (int p = first[(size_t)t]; p < first[(size_t)t + 1]; ++p)
            for (int64_t i = 0; i < n_rows[p]; ++i) {
                const int64_t at_row = pos[keys[p][i]]++;
                double* o = out + at_row * n_main;
                const double* src = table + ids[p][i] * ld;
                for (int j = 0; j < n_main; ++j) o[j] = src[src_cols[j]];
                for (int a = 0; a < 3; ++a) o[abs_dst0 + a] = abs_rows[p][3 * i + a];
                if (n_rest) {
                    double* o2 = out_rest + at_row * n_rest;
                    for (int j = 0; j < n_rest; ++j) o2[j] = src[src_cols[n_main + j]];
                }
            }
    });
    return MMX_OK;
}

// Rows of a table that lie inside ANY of `n_boxes` boxes [lo, hi): what a rank sends to the ranks whose blocks its
// rows can influence (the distributed pruning's first exchange), ten values a row -- detection coordinates, block
// tags, absolute coordinates, channel.  out: [cap][10] float64; *out_n keeps counting past cap.
extern "C" int mmx_host_rows_in_boxes(const int32_t* zyx, const int32_t* tag, const double* abs_zyx, const double* chan,
                                      int64_t chan_ld, int64_t n, const int32_t* box_lo, const int32_t* box_hi,
                                      int n_boxes, double* out, int64_t cap, int64_t* out_n)
{
    if (n < 0 || n_boxes < 0 || !out_n || (n && (!zyx || !tag || !abs_zyx || !chan)) || (n_boxes && (!box_lo || !box_hi)) ||
        (cap && !out) || cap < 0)
        return MMX_ERR_ARG;
    *out_n = 0;
    if (n == 0 || n_boxes == 0) return MMX_OK;
    auto inside = [&](int64_t i) {
        const int32_t* c = zyx + 3 * i;
        for (int b = 0; b < n_boxes; ++b) {
            const int32_t* lo = box_lo + 3 * b; const int32_t* hi = box_hi + 3 * b;
            if (c[0] >= lo[0] && c[0] < hi[0] && c[1] >= lo[1] && c[1] < hi[1] && c[2] >= lo[2] && c[2] < hi[2]) return true;
        }
        return false;
    };
    // two passes over contiguous runs of rows: count, then write from each run's offset (the rows keep their order)
    const int T = host_threads(n);
    std::vector<int64_t> first((size_t)T + 1, 0);
    parallel(T, [&](int t, int) {
        int64_t k = 0;
        for (int64_t i = n * t / T; i < n * (t + 1) / T; ++i) k += inside(i);
        first[(size_t)t + 1] = k;
    });
    for (int t = 0; t < T; ++t) first[(size_t)t + 1] += first[(size_t)t];
    *out_n = first[(size_t)T];
    parallel(T, [&](int t, int) {
        int64_t k = first[(size_t)t];
        for (int64_t i = n * t / T; i < n * (t + 1) / T && k < cap; ++i) {
            if (!inside(i)) continue;
            const int32_t* c = zyx + 3 * i;
            double* o = out + 10 * k;
            o[0] = c[0]; o[1] = c[1]; o[2] = c[2];
            o[3] = tag[3 * i]; o[4] = tag[3 * i + 1]; o[5] = tag[3 * i + 2];
            o[6] = abs_zyx[3 * i]; o[7] = abs_zyx[3 * i + 1]; o[8] = abs_zyx[3 * i + 2];
            o[9] = chan[i * chan_ld];
            ++k;
        }
    });
    return MMX_OK;
}

// The reverse on the receiving rank: rows of such a payload inside the box [lo, hi) appended to the compact columns
// of a table from row `at` on (capacity `cap` rows): zyx / tag as int32, abs as float64, the channel into `chan`.
extern "C" int mmx_host_append_rows(const double* payload, int64_t n, const int32_t lo[3], const int32_t hi[3],
                                    int32_t* zyx, int32_t* tag, double* abs_zyx, double* chan, int64_t chan_ld,
                                    int64_t at, int64_t cap, int64_t* out_n)
{
    if (n < 0 || at < 0 || cap < at || !lo || !hi || !out_n || (n && (!payload || !zyx || !tag || !abs_zyx || !chan)))
        return MMX_ERR_ARG;
    int64_t k = 0;
    for (int64_t i = 0; i < n; ++i) {
        const double* r = payload + 10 * i;
        const int32_t z = (int32_t)r[0], y = (int32_t)r[1], x = (int32_t)r[2];
        if (z < lo[0] || z >= hi[0] || y < lo[1] || y >= hi[1] || x < lo[2] || x >= hi[2]) continue;
        const int64_t row = at + k;
        if (row < cap) {
            zyx[3 * row] = z; zyx[3 * row + 1] = y; zyx[3 * row + 2] = x;
            tag[3 * row] = (int32_t)r[3]; tag[3 * row + 1] = (int32_t)r[4]; tag[3 * row + 2] = (int32_t)r[5];
            abs_zyx[3 * row] = r[6]; abs_zyx[3 * row + 1] = r[7]; abs_zyx[3 * row + 2] = r[8];
            chan[row * chan_ld] = r[9];
        }
        ++k;
    }
    *out_n = k;
    return at + k <= cap ? MMX_OK : MMX_ERR_WORKSPACE;
}

// Survivors of a rank in the form the second exchange carries: row ids[i] of `table` (its first n_cols columns), the
// three abs columns replaced by abs_rows[i], the sort key appended as column n_cols -- in the order given.
//   out : [n][n_cols + 1]
extern "C" int mmx_host_emit_survivors(const double* table, int64_t ld, const int64_t* ids, const int64_t* keys,
                                       int64_t n, int64_t n_cols, const double* abs_rows, const int32_t abs_cols[3],
                                       double* out)
{
    if (n < 0 || n_cols < 1 || n_cols > ld || !abs_cols || (n && (!table || !ids || !keys || !out || !abs_rows)))
        return MMX_ERR_ARG;
    for (int a = 0; a < 3; ++a)
        if (abs_cols[a] < 0 || abs_cols[a] >= n_cols) return MMX_ERR_ARG;
    for (int64_t i = 0; i < n; ++i)
        if (ids[i] < 0) return MMX_ERR_ARG;
    parallel(host_threads(n), [&](int t, int nt) {
        const int64_t lo = n * t / nt, hi = n * (t + 1) / nt;
        for (int64_t i = lo; i < hi; ++i) {
            double* o = out + i * (n_cols + 1);
            std::memcpy(o, table + ids[i] * ld, (size_t)n_cols * sizeof(double));
            for (int a = 0; a < 3; ++a) o[abs_cols[a]] = abs_rows[3 * i + a];
            o[n_cols] = (double)keys[i];
        }
    });
    return MMX_OK;
}

// mmx_host_emit_survivors with the rows leaving in the caller's final column layout (see
// mmx_host_gather_parts_by_key_final): out[i] = table[ids[i]][src_cols[0 .. n_out)], abs_rows[i] at abs_dst0, then the
// key -- what travels in the distributed pruning's second exchange is a quarter smaller, and so is the merge after it.
extern "C" int mmx_host_emit_survivors_final(const double* table, int64_t ld, const int64_t* ids, const int64_t* keys,
                                             int64_t n, const int32_t* src_cols, int32_t n_out, const double* abs_rows,
                                             int32_t abs_dst0, double* out)
{
    if (n < 0 || n_out < 3 || n_out > 64 || !src_cols || abs_dst0 < 0 || abs_dst0 + 3 > n_out ||
        (n && (!table || !ids || !keys || !out || !abs_rows)))
        return MMX_ERR_ARG;
    for (int j = 0; j < n_out; ++j)
        if (src_cols[j] < 0 || src_cols[j] >= ld) return MMX_ERR_ARG;
    for (int64_t i = 0; i < n; ++i)
        if (ids[i] < 0) return MMX_ERR_ARG;
    parallel(host_threads(n), [&](int t, int nt) {
        const int64_t lo = n * t / nt, hi = n * (t + 1) / nt;
        for (int64_t i = lo; i < hi; ++i) {
            double* o = out + i * (n_out + 1);
            const double* src = table + ids[i] * ld;
            for (int j = 0; j < n_out; ++j) o[j] = src[src_cols[j]];
            for (int a = 0; a < 3; ++a) o[abs_dst0 + a] = abs_rows[3 * i + a];
            o[n_out] = (double)keys[i];
        }
    });
    return MMX_OK;
}

// mmx_host_emit_survivors_final for n_parts survivor lists at once (a rank's regions, in order): one threaded pass over
// all of them -- a region's few thousand rows are too few for a call of their own to thread itself, and sixteen calls
// from Python threads cost more in hand-offs than in copying.
extern "C" int mmx_host_emit_parts_final(const double* table, int64_t ld, int32_t n_parts, const int64_t* const* ids,
                                         const int64_t* const* keys, const double* const* abs_rows,
                                         const int64_t* n_rows, const int32_t* src_cols, int32_t n_out, int32_t abs_dst0,
                                         double* out, int64_t out_rows)
{
    if (n_parts < 0 || n_out < 3 || n_out > 64 || !src_cols || abs_dst0 < 0 || abs_dst0 + 3 > n_out ||
        (n_parts && (!ids || !keys || !abs_rows || !n_rows)))
        return MMX_ERR_ARG;
    for (int j = 0; j < n_out; ++j)
        if (src_cols[j] < 0 || src_cols[j] >= ld) return MMX_ERR_ARG;
    std::vector<int64_t> first((size_t)n_parts + 1, 0);
    for (int p = 0; p < n_parts; ++p) {
        if (n_rows[p] < 0 || (n_rows[p] && (!ids[p] || !keys[p] || !abs_rows[p]))) return MMX_ERR_ARG;
        first[(size_t)p + 1] = first[(size_t)p] + n_rows[p];
    }
    const int64_t n = first[(size_t)n_parts];
    if (n != out_rows || (n && (!table || !out))) return MMX_ERR_ARG;
    if (n == 0) return MMX_OK;
    const int T = host_threads(n);
    std::vector<int> bad((size_t)T, 0);
    parallel(T, [&](int t, int) {
        // rows [lo, hi) of the concatenation: walk the parts they fall into
        const int64_t lo = n * t / T, hi = n * (t + 1) / T;
        int p = (int)(std::upper_bound(first.begin(), first.end(), lo) - first.begin()) - 1;
        for (int64_t g = lo; g < hi; ++p) {
            const int64_t end = std::min(hi, first[(size_t)p + 1]);
            for (; g < end; ++g) {
                const int64_t i = g - first[(size_t)p];
                if (ids[p][i] < 0) { bad[(size_t)t] = 1; return; }
                double* o = out + g * (n_out + 1);
                const double* src = table + ids[p][i] * ld;
                for (int j = 0; j < n_out; ++j) o[j] = src[src_cols[j]];
                for (int a = 0; a < 3; ++a) o[abs_dst0 + a] = abs_rows[p][3 * i + a];
                o[n_out] = (double)keys[p][i];
            }
        }
    });
    for (int t = 0; t < T; ++t)
        if (bad[(size_t)t]) return MMX_ERR_ARG;
    return MMX_OK;
}

// out[i][dst_col0 + j] = table[i][src_cols[j]] for every row: the column shuffles that end a stack
// detection (Blobs.replace_rel_with_abs_blob_coords: out = table, columns 7..9 -> 0..2;
// Blobs.remove_abs_blob_coords: the kept columns into a new table).  `out` may be `table` itself (a row is
// read completely before it is written).
extern "C" int mmx_host_map_columns(const double* table, int64_t ld, int64_t n, const int32_t* src_cols,
                                    int32_t n_map, double* out, int64_t out_ld, int32_t dst_col0)
{
    if (!table || !out || !src_cols || n < 0 || n_map < 1 || n_map > 64 || dst_col0 < 0 || dst_col0 + n_map > out_ld)
        return MMX_ERR_ARG;
    for (int j = 0; j < n_map; ++j)
        if (src_cols[j] < 0 || src_cols[j] >= ld) return MMX_ERR_ARG;
    parallel(host_threads(n), [&](int t, int nt) {
        const int64_t lo = n * t / nt, hi = n * (t + 1) / nt;
        double tmp[64];
        for (int64_t i = lo; i < hi; ++i) {
            const double* r = table + i * ld;
            for (int j = 0; j < n_map; ++j) tmp[j] = r[src_cols[j]];
            std::memcpy(out + i * out_ld + dst_col0, tmp, (size_t)n_map * sizeof(double));
        }
    });
    return MMX_OK;
}

// Rows `rows[0..n)` of a float64 table (row pitch `ld`), first `n_cols` columns, with three of the
// columns replaced from a compact (n_table, 3) array -- the output of prune_blobs_mp
// (`merged[rows][:, :-3]` with the updated absolute coordinates put back).
extern "C" int mmx_host_take_rows(const double* table, int64_t ld, const int64_t* rows, int64_t n,
                                  int64_t n_cols, const double* abs_zyx, const int32_t abs_cols[3],
                                  double* out)
{
    if (!table || (!rows && n) || !out || n < 0 || n_cols < 1 || n_cols > ld || !abs_zyx || !abs_cols)
        return MMX_ERR_ARG;
    for (int a = 0; a < 3; ++a)
        if (abs_cols[a] < 0 || abs_cols[a] >= n_cols) return MMX_ERR_ARG;
    parallel(host_threads(n), [&](int t, int nt) {
        const int64_t lo = n * t / nt, hi = n * (t + 1) / nt;
        for (int64_t i = lo; i < hi; ++i) {
            const int64_t r = rows[i];
            double* o = out + i * n_cols;
            std::memcpy(o, table + r * ld, (size_t)n_cols * sizeof(double));
            for (int a = 0; a < 3; ++a) o[abs_cols[a]] = abs_zyx[3 * r + a];
        }
    });
    return MMX_OK;
}

// mmx_host_take_rows with the table leaving in the caller's final column layout (see mmx_host_gather_parts_by_key_final):
// out[i][j] = table[rows[i]][src_cols[j]], then out[i][abs_dst0 .. +3] = abs_zyx[rows[i]].
extern "C" int mmx_host_take_rows_final(const double* table, int64_t ld, const int64_t* rows, int64_t n,
                                        const int32_t* src_cols, int32_t n_out, const double* abs_zyx,
                                        int32_t abs_dst0, double* out)
{
    return mmx_host_take_rows_split(table, ld, rows, n, src_cols, n_out, abs_zyx, abs_dst0, out, n_out, nullptr);
}

// ... with the output in two tables (see mmx_host_gather_parts_by_key_split).
extern "C" int mmx_host_take_rows_split(const double* table, int64_t ld, const int64_t* rows, int64_t n,
                                        const int32_t* src_cols, int32_t n_out, const double* abs_zyx,
                                        int32_t abs_dst0, double* out, int32_t n_main, double* out_rest)
{
    if (!table || (!rows && n) || !out || n < 0 || n_out < 3 || n_out > 64 || !src_cols || !abs_zyx || abs_dst0 < 0 ||
        abs_dst0 + 3 > n_out)
        return MMX_ERR_ARG;
    if (!out_rest) n_main = n_out;
    if (n_main < abs_dst0 + 3 || n_main > n_out || (out_rest && n_main == n_out)) return MMX_ERR_ARG;
    const int n_rest = n_out - n_main;
    for (int j = 0; j < n_out; ++j)
        if (src_cols[j] < 0 || src_cols[j] >= ld) return MMX_ERR_ARG;
    parallel(host_threads(n), [&](int t, int nt) {
        const int64_t lo = n * t / nt, hi = n * (t + 1) / nt;
        for (int64_t i = lo; i < hi; ++i) {
            const int64_t r = rows[i];
            double* o = out + i * n_main;
            const double* src = table + r * ld;
            for (int j = 0; j < n_main; ++j) o[j] = src[src_cols[j]];
            for (int a = 0; a < 3; ++a) o[abs_dst0 + a] = abs_zyx[3 * r + a];
            if (n_rest) {
                double* o2 = out_rest + i * n_rest;
                for (int j = 0; j < n_rest; ++j) o2[j] = src[src_cols[n_main + j]];
            }
        }
    });
    return MMX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// mmx_host_lsap -- rectangular linear sum assignment (minimum total cost, every row of the shorter side assigned),
// what the reference's match-based co-localisation gets from scipy.optimize.linear_sum_assignment
// (magmap/cv/verifier.py:86: `rowis, colis = optimize.linear_sum_assignment(dists)`).
//
// Algorithm: shortest augmenting paths with dual variables (D. F. Crouse, "On implementing 2D rectangular
// assignment algorithms", IEEE Trans. Aerospace and Electronic Systems 52(4), 2016), the algorithm SciPy
// implements.  Blob coordinates are integers, so many distances are EQUAL and the optimum is often not unique:
// to return the assignment the reference gets, the free choices follow SciPy's: rows are inserted in order; the
// columns not yet scanned are kept in an array initialised in DESCENDING order and scanned front to back, a
// scanned column is replaced by the array's last entry; among equal tentative distances the LAST candidate in
// scan order that is still unassigned wins, otherwise the first one found; a matrix with more rows than columns
// is solved transposed and its pairs returned sorted by row.  Outputs are `min(nr, nc)` (row, column) pairs in
// ascending row order.  Returns MMX_ERR_ARG for NaN / -inf entries or when no finite assignment exists.
extern "C" int mmx_host_lsap(const double* cost, int64_t nr, int64_t nc, int64_t* out_rows, int64_t* out_cols)
{
    if (nr < 0 || nc < 0 || (nr && nc && (!cost || !out_rows || !out_cols))) return MMX_ERR_ARG;
    if (nr == 0 || nc == 0) return MMX_OK;
    const bool transposed = nc < nr;
    std::vector<double> tmp;
    const double* c = cost;
    int64_t n_small = nr, n_big = nc;        // the problem solved has n_small rows <= n_big columns
    if (transposed) {
        tmp.resize((size_t)(nr * nc));
        for (int64_t i = 0; i < nr; ++i)
            for (int64_t j = 0; j < nc; ++j) tmp[(size_t)(j * nr + i)] = cost[i * nc + j];
        c = tmp.data();
        n_small = nc;
        n_big = nr;
    }
    for (int64_t k = 0; k < nr * nc; ++k)
        if (c[k] != c[k] || c[k] == -INFINITY) return MMX_ERR_ARG;

    std::vector<double> dual_row((size_t)n_small, 0.0), dual_col((size_t)n_big, 0.0), dist((size_t)n_big);
    std::vector<int64_t> pred((size_t)n_big, -1), col_of_row((size_t)n_small, -1), row_of_col((size_t)n_big, -1);
    std::vector<int64_t> todo((size_t)n_big);
    std::vector<char> row_seen((size_t)n_small), col_seen((size_t)n_big);

    for (int64_t start = 0; start < n_small; ++start) {
        // ---- grow a shortest-path tree from `start` until it reaches an unassigned column
        int64_t n_todo = n_big;
        for (int64_t k = 0; k < n_big; ++k) todo[(size_t)k] = n_big - 1 - k;
        std::fill(row_seen.begin(), row_seen.end(), 0);
        std::fill(col_seen.begin(), col_seen.end(), 0);
        std::fill(dist.begin(), dist.end(), INFINITY);
        double reach = 0.0;
        int64_t row = start, sink = -1;
        while (sink < 0) {
            row_seen[(size_t)row] = 1;
            int64_t best_at = -1;
            double best = INFINITY;
            for (int64_t k = 0; k < n_todo; ++k) {
                const int64_t col = todo[(size_t)k];
                const double through = reach + c[row * n_big + col] - dual_row[(size_t)row] - dual_col[(size_t)col];
                if (through < dist[(size_t)col]) {
                    dist[(size_t)col] = through;
                    pred[(size_t)col] = row;
                }
                if (dist[(size_t)col] < best || (dist[(size_t)col] == best && row_of_col[(size_t)col] < 0)) {
                    best = dist[(size_t)col];
                    best_at = k;
                }
            }
            reach = best;
            if (reach == INFINITY) return MMX_ERR_ARG;       // no finite assignment
            const int64_t col = todo[(size_t)best_at];
            if (row_of_col[(size_t)col] < 0) sink = col;
            else row = row_of_col[(size_t)col];
            col_seen[(size_t)col] = 1;
            todo[(size_t)best_at] = todo[(size_t)--n_todo];
        }
        // ---- dual update
        dual_row[(size_t)start] += reach;
        for (int64_t i = 0; i < n_small; ++i)
            if (row_seen[(size_t)i] && i != start) dual_row[(size_t)i] += reach - dist[(size_t)col_of_row[(size_t)i]];
        for (int64_t j = 0; j < n_big; ++j)
            if (col_seen[(size_t)j]) dual_col[(size_t)j] -= reach - dist[(size_t)j];
        // ---- flip the path from the sink back to `start`
        for (int64_t col = sink;;) {
            const int64_t i = pred[(size_t)col];
            row_of_col[(size_t)col] = i;
            std::swap(col_of_row[(size_t)i], col);
            if (i == start) break;
        }
    }
    if (!transposed) {
        for (int64_t i = 0; i < n_small; ++i) { out_rows[i] = i; out_cols[i] = col_of_row[(size_t)i]; }
    } else {
        // solved on the transpose: col_of_row[j] is the ORIGINAL row assigned to original column j; pairs in
        // ascending original-row order
        std::vector<int64_t> order((size_t)n_small);
        for (int64_t j = 0; j < n_small; ++j) order[(size_t)j] = j;
        std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return col_of_row[(size_t)a] < col_of_row[(size_t)b]; });
        for (int64_t k = 0; k < n_small; ++k) { out_rows[k] = col_of_row[(size_t)order[(size_t)k]]; out_cols[k] = order[(size_t)k]; }
    }
    return MMX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Per-batch host work of the detection (A4 decisions, A5, block tables), native and threaded over blocks.
//
// What the reference does here, per block, on float64 NumPy arrays (skimage/feature/peak.py:9-50, 114-319 and
// blob.py:146-187, then magmap/cv/detector.py:934-943, stack_detect.py:164-170):
//   coordinates = np.nonzero(mask)            -> C order over (z, y, x, sigma)
//   order = np.argsort(-values)               -> descending response
//   pairs within 2 sigma_max sqrt(3) whose sphere overlap exceeds `overlap`: the smaller sigma is zeroed (the
//   first of the pair on equal sigmas), a zeroed blob zeroes nothing;  radius = sigma sqrt(3); 11 columns;
//   block offset added to the rel and abs coordinates.
// The device nominates candidates in float32 and re-scores them (and the neighbours that can out-vote the contested
// ones, mmx_expand_probes) in float64; these functions take the decisions on those float64 values.

namespace {
struct cand_rec {           // mmx_cand, 48 bytes
    int32_t slot, s, z, y, x;
    uint32_t flags;
    float v, nbr_max;
    double v64;
    uint64_t band;
};
static_assert(sizeof(cand_rec) == 48 && sizeof(cand_rec) == sizeof(mmx_cand), "mmx_cand layout");
}  // namespace

// Exact peak membership and the reference's two orders.
//   cands[0, n_cands): candidates; cands[n_cands, n_total): probes (band = index of the candidate they may out-vote)
//   out_nz_* : the peaks in np.nonzero order (C order of the (z, y, x, sigma) cube) -- filled for the blocks whose
//              `ties` flag is set only (nobody else needs that order)
//   out_*    : the same rows per block by descending value (stable: equal values keep the nonzero order)
//   offsets  : [n_blocks + 1] row ranges of the blocks in both
//   ties     : [n_blocks] 1 when two peaks of the block have EQUAL values -- np.argsort's order of equal keys is
//              its own (an unstable introsort): the caller then takes that block's order from NumPy itself
//   stats    : [4] candidates flagged contested, peaks, max |float32 - float64| over the candidates (NaN / inf when
//              a value is not finite), blocks dropped as constant cubes
extern "C" int mmx_host_resolve_peaks(const mmx_cand* cands_, uint32_t n_cands, uint32_t n_total,
                                      const mmx_block* blocks, int n_blocks, int n_sigma, double thr,
                                      int32_t* out_nz_coords, double* out_nz_vals, int32_t* out_coords, double* out_vals,
                                      int32_t* offsets, uint8_t* ties, double* stats)
{
    if ((!cands_ && n_total) || n_cands > n_total || !blocks || n_blocks < 1 || n_sigma < 1 || !offsets || !ties || !stats ||
        (n_cands && (!out_nz_coords || !out_nz_vals || !out_coords || !out_vals)))
        return MMX_ERR_ARG;
    const cand_rec* cands = reinterpret_cast<const cand_rec*>(cands_);
    // ---- the best exact value among the neighbours that can out-vote each contested candidate
    std::vector<double> rival;
    if (n_total > n_cands) {
        rival.assign(n_cands, -INFINITY);
        for (uint32_t p = n_cands; p < n_total; ++p) {
            const cand_rec& q = cands[p];
            if (!(q.flags & MMX_CAND_PROBE) || q.band >= n_cands) return MMX_ERR_ARG;
            if (!(q.v64 == q.v64)) {                                       // (never re-scored / NaN voxels)
                stats[0] = stats[1] = stats[3] = 0.0; stats[2] = INFINITY;
                for (int b = 0; b <= n_blocks; ++b) offsets[b] = 0;
                return MMX_OK;
            }
            if (q.v64 > rival[q.band]) rival[q.band] = q.v64;
        }
    }
    // ---- keep / drop per candidate, and its place in the cube
    const int T = n_cands < 4096 ? 1 : std::min(pool::get().size(), n_blocks > 1 ? 16 : 4);
    std::vector<int64_t> key(n_cands);               // position in the block's (z, y, x, sigma) cube, -1: dropped
    std::vector<std::vector<int64_t>> per_block((size_t)T, std::vector<int64_t>((size_t)n_blocks + 1, 0));
    std::vector<double> errs((size_t)T, 0.0);
    std::vector<int64_t> n_cont((size_t)T, 0);
    std::vector<int> bad((size_t)T, 0);
    parallel(T, [&](int t, int nt) {
        const uint32_t lo = (uint32_t)((uint64_t)n_cands * t / nt), hi = (uint32_t)((uint64_t)n_cands * (t + 1) / nt);
        double err = 0.0;
        for (uint32_t i = lo; i < hi; ++i) {
            const cand_rec& c = cands[i];
            key[i] = -1;
            if (c.slot < 0 || c.slot >= n_blocks || c.s < 0 || c.s >= n_sigma) { bad[(size_t)t] = 1; continue; }
            const mmx_block& bd = blocks[c.slot];
            if (c.z < 0 || c.z >= bd.nz || c.y < 0 || c.y >= bd.ny || c.x < 0 || c.x >= bd.nx) { bad[(size_t)t] = 1; continue; }
            const double e = std::fabs((double)c.v - c.v64);
            if (e != e) err = INFINITY;               // (NaN voxels: reported as a non-finite deviation)
            else if (e > err) err = e;
            bool keep = c.v64 > thr;
            if (c.flags & MMX_CAND_CONTESTED) {
                ++n_cont[(size_t)t];
                double m = rival.empty() ? -INFINITY : rival[i];
                const bool border = c.s == 0 || c.s == n_sigma - 1 || c.z == 0 || c.z == bd.nz - 1 || c.y == 0 ||
                                    c.y == bd.ny - 1 || c.x == 0 || c.x == bd.nx - 1;
                if (border && !(m > 0.0)) m = 0.0;     // the cube is zero-padded (mode='constant', cval 0)
                keep = keep && c.v64 >= m;
            }
            if (keep) {
                key[i] = (((int64_t)c.z * bd.ny + c.y) * bd.nx + c.x) * n_sigma + c.s;
                ++per_block[(size_t)t][(size_t)c.slot];
            }
        }
        errs[(size_t)t] = err;
    });
    double err = 0.0;
    int64_t contested = 0;
    for (int t = 0; t < T; ++t) {
        if (bad[(size_t)t]) return MMX_ERR_ARG;
        if (errs[(size_t)t] > err) err = errs[(size_t)t];
        contested += n_cont[(size_t)t];
    }
    stats[0] = (double)contested; stats[1] = 0.0; stats[2] = err; stats[3] = 0.0;
    // ---- rows per block (a block whose every voxel is a "peak" is a constant cube: no peaks, peak.py:41-43)
    std::vector<int64_t> count((size_t)n_blocks, 0);
    for (int b = 0; b < n_blocks; ++b)
        for (int t = 0; t < T; ++t) count[(size_t)b] += per_block[(size_t)t][(size_t)b];
    std::vector<char> trivial((size_t)n_blocks, 0);
    offsets[0] = 0;
    for (int b = 0; b < n_blocks; ++b) {
        const int64_t cube = (int64_t)blocks[b].nz * blocks[b].ny * blocks[b].nx * n_sigma;
        if (count[(size_t)b] == cube && cube > 1) { trivial[(size_t)b] = 1; stats[3] += 1.0; }
        const int64_t next = (int64_t)offsets[b] + (trivial[(size_t)b] ? 0 : count[(size_t)b]);
        if (next > INT32_MAX) return MMX_ERR_UNSUPPORTED;
        offsets[b + 1] = (int32_t)next;
        ties[b] = 0;
    }
    stats[1] = (double)offsets[n_blocks];
    if (!std::isfinite(err) || offsets[n_blocks] == 0) return MMX_OK;
    // ---- scatter the kept candidates to their blocks (chunk order: any; sorted next), then per block the two orders
    std::vector<std::pair<int64_t, uint32_t>> rows((size_t)offsets[n_blocks]);       // (cube position, candidate)
    {
        std::vector<int64_t> at((size_t)n_blocks);
        for (int b = 0; b < n_blocks; ++b) at[(size_t)b] = offsets[b];
        for (uint32_t i = 0; i < n_cands; ++i) {
            if (key[i] < 0) continue;
            const int b = cands[i].slot;
            if (trivial[(size_t)b]) continue;
            rows[(size_t)at[(size_t)b]++] = std::make_pair(key[i], i);
        }
    }
    // per block: ONE sort, by descending value and -- for equal values -- ascending cube position, which is what a stable
    // sort of the np.nonzero rows by value gives; the np.nonzero order itself is only needed where values tie (the
    // caller then asks NumPy for its order of them): that block is sorted a second time
    parallel(std::min(T, n_blocks), [&](int t, int nt) {
        struct row { double v; int64_t key; uint32_t cand; };
        std::vector<row> rs;
        for (int b = t; b < n_blocks; b += nt) {
            const int64_t lo = offsets[b], hi = offsets[b + 1];
            if (lo == hi) continue;
            rs.resize((size_t)(hi - lo));
            for (int64_t r = lo; r < hi; ++r) {
                const uint32_t ci = rows[(size_t)r].second;
                rs[(size_t)(r - lo)] = row{cands[ci].v64, rows[(size_t)r].first, ci};
            }
            std::sort(rs.begin(), rs.end(), [](const row& a, const row& b2) {
                return a.v > b2.v || (a.v == b2.v && a.key < b2.key);          // a voxel appears once: keys are unique
            });
            uint8_t tie = 0;
            for (int64_t r = lo; r < hi; ++r) {
                const row& q = rs[(size_t)(r - lo)];
                const cand_rec& c = cands[q.cand];
                int32_t* o = out_coords + 4 * r;
                o[0] = c.z; o[1] = c.y; o[2] = c.x; o[3] = c.s;
                out_vals[r] = q.v;
                if (r > lo && q.v == rs[(size_t)(r - lo - 1)].v) tie = 1;
            }
            ties[b] = tie;
            if (tie) {
                std::sort(rs.begin(), rs.end(), [](const row& a, const row& b2) { return a.key < b2.key; });
                for (int64_t r = lo; r < hi; ++r) {
                    const row& q = rs[(size_t)(r - lo)];
                    const cand_rec& c = cands[q.cand];
                    int32_t* o = out_nz_coords + 4 * r;
                    o[0] = c.z; o[1] = c.y; o[2] = c.x; o[3] = c.s;
                    out_nz_vals[r] = q.v;
                }
            }
        }
    });
    return MMX_OK;
}

// Sphere-overlap pruning of every block (skimage/feature/blob.py:84-187), on the peaks in descending-response order.
//   coords  : [n][4] int32 (z, y, x, sigma index), blocks delimited by offsets[n_blocks + 1]
//   alive   : out [n] 1 = the blob survives.  Final for every block whose flag in `open_blocks` is 0.
//   open_blocks : out [n_blocks] 1 = some blob of the block loses one over-limit pair and wins another: the outcome
//             depends on the ORDER scikit-image visits the pairs in (cKDTree.query_pairs, implementation defined),
//             which the caller takes from the same call; such a block's `alive` flags are all 1
//   pairs / frac / cap / n_pairs : every pair (global rows i < j) whose overlap fraction exceeds overlap - band, in
//             no particular order; *n_pairs counts past cap (the caller retries with a larger table)
//   n_knife : out, pairs within `band` of the limit: their fraction must be re-evaluated with the reference's exact
//             libm calls (caller); when non-zero NOTHING in `alive` is final
extern "C" int mmx_host_overlap_prune(const int32_t* coords, const int32_t* offsets, int n_blocks,
                                      const double* sigmas, int n_sigma, double overlap, double band,
                                      uint8_t* alive, uint8_t* open_blocks, int32_t* pairs, double* frac,
                                      int64_t cap, int64_t* n_pairs, int64_t* n_knife)
{
    if (!offsets || n_blocks < 1 || !sigmas || n_sigma < 1 || !open_blocks || !n_pairs || !n_knife || cap < 0 ||
        (cap && (!pairs || !frac)))
        return MMX_ERR_ARG;
    const int64_t n = offsets[n_blocks];
    if (n && (!coords || !alive)) return MMX_ERR_ARG;
    double smax = 0.0;
    for (int s = 0; s < n_sigma; ++s) smax = std::max(smax, sigmas[s]);
    if (!(smax > 0.0)) return MMX_ERR_ARG;
    for (int64_t i = 0; i < n; ++i) {
        if (coords[4 * i + 3] < 0 || coords[4 * i + 3] >= n_sigma) return MMX_ERR_ARG;
        alive[i] = 1;
    }
    for (int b = 0; b < n_blocks; ++b) open_blocks[b] = 0;
    const double root3 = std::sqrt(3.0);
    const double kPi = 3.141592653589793;           // math.pi
    // no overlap beyond sqrt(3) (s_i + s_j) <= 2 sqrt(3) s_max: cells of that size, 27 of them around a blob
    const double cell = 2.0 * root3 * smax + 1.0;
    const int T = n < 2000 ? 1 : std::min(pool::get().size(), 16);
    struct found { int32_t i, j; double f; };
    std::vector<std::vector<found>> per_thread((size_t)T);
    // (few blocks -- a small stack -- : each block's blobs are dealt to `parts` threads, every one of which builds the
    //  block's cell lists for itself: ~5 % of the pair search it shares)
    const int parts = T > n_blocks ? std::min(8, T / n_blocks) : 1;
    const int n_items = n_blocks * parts;
    parallel(std::min(T, n_items), [&](int t, int nt) {
        std::vector<int32_t> cell_of, start, sorted;
        auto& out = per_thread[(size_t)t];
        for (int item = t; item < n_items; item += nt) {
            const int b = item / parts, part = item % parts;
            const int32_t lo = offsets[b], hi = offsets[b + 1];
            const int m = hi - lo;
            if (m < 2) continue;
            int mx[3] = {0, 0, 0};
            for (int32_t r = lo; r < hi; ++r)
                for (int a = 0; a < 3; ++a) {
                    if (coords[4 * (int64_t)r + a] < 0) { mx[0] = -1; break; }
                    mx[a] = std::max(mx[a], coords[4 * (int64_t)r + a]);
                }
            if (mx[0] < 0) continue;                    // (negative coordinates never come from the detector)
            const int gz = (int)(mx[0] / cell) + 1, gy = (int)(mx[1] / cell) + 1, gx = (int)(mx[2] / cell) + 1;
            const int64_t n_cells = (int64_t)gz * gy * gx;
            cell_of.resize((size_t)m);
            start.assign((size_t)n_cells + 1, 0);
            for (int k = 0; k < m; ++k) {
                const int32_t* c = coords + 4 * (int64_t)(lo + k);
                const int ci = ((int)(c[0] / cell) * gy + (int)(c[1] / cell)) * gx + (int)(c[2] / cell);
                cell_of[(size_t)k] = ci;
                ++start[(size_t)ci + 1];
            }
            for (int64_t c = 0; c < n_cells; ++c) start[(size_t)c + 1] += start[(size_t)c];
            sorted.resize((size_t)m);
            {
                std::vector<int32_t> at(start.begin(), start.end() - 1);
                for (int k = 0; k < m; ++k) sorted[(size_t)at[(size_t)cell_of[(size_t)k]]++] = k;
            }
            for (int i = part; i < m; i += parts) {          // (interleaved: blob i meets only the blobs after it)
                const int32_t* ci = coords + 4 * (int64_t)(lo + i);
                const double zi = ci[0], yi = ci[1], xi = ci[2], si = sigmas[ci[3]];
                const int cz = (int)(ci[0] / cell), cy = (int)(ci[1] / cell), cx = (int)(ci[2] / cell);
                for (int az = std::max(0, cz - 1); az <= std::min(gz - 1, cz + 1); ++az)
                    for (int ay = std::max(0, cy - 1); ay <= std::min(gy - 1, cy + 1); ++ay)
                        for (int ax = std::max(0, cx - 1); ax <= std::min(gx - 1, cx + 1); ++ax) {
                            const int64_t cc = ((int64_t)az * gy + ay) * gx + ax;
                            for (int32_t q = start[(size_t)cc]; q < start[(size_t)cc + 1]; ++q) {
                                const int j = sorted[(size_t)q];
                                if (j <= i) continue;                       // every pair once, i < j
                                const int32_t* cj = coords + 4 * (int64_t)(lo + j);
                                const double sj = sigmas[cj[3]];
                                if (si == 0.0 && sj == 0.0) continue;
                                double r1, r2, ms;
                                if (si > sj) { ms = si; r1 = 1.0; r2 = sj / si; }
                                else         { ms = sj; r2 = 1.0; r1 = si / sj; }
                                const double den = ms * root3;
                                const double d0 = cj[0] / den - zi / den;
                                const double d1 = cj[1] / den - yi / den;
                                const double d2 = cj[2] / den - xi / den;
                                const double d = std::sqrt((d0 * d0 + d1 * d1) + d2 * d2);
                                if (d > r1 + r2) continue;
                                double f;
                                if (d <= std::fabs(r1 - r2)) {
                                    f = 1.0;
                                } else {
                                    const double rs = r1 + r2;
                                    const double tt = rs - d;
                                    const double vol = kPi / (12 * d) * (tt * tt) *
                                                       (d * d + 2 * d * rs - 3 * (r1 * r1 + r2 * r2) + 6 * r1 * r2);
                                    const double rm = r1 < r2 ? r1 : r2;
                                    f = vol / (4. / 3 * kPi * (rm * rm * rm));
                                }
                                if (f > overlap - band) out.push_back(found{lo + i, lo + j, f});
                            }
                        }
            }
        }
    });
    int64_t total = 0, knife = 0;
    for (auto& v : per_thread) total += (int64_t)v.size();
    *n_pairs = total;
    {
        int64_t at = 0;
        for (auto& v : per_thread)
            for (const found& p : v) {
                if (std::fabs(p.f - overlap) <= band) ++knife;
                if (at < cap) { pairs[2 * at] = p.i; pairs[2 * at + 1] = p.j; frac[at] = p.f; }
                ++at;
            }
    }
    *n_knife = knife;
    if (knife || total > cap) return MMX_OK;
    // ---- the sequential rule where its outcome does not depend on the order: loser = the smaller sigma, the first of
    // the pair on equal sigmas; a block is "open" when one of its blobs both loses and wins
    std::vector<uint8_t> role((size_t)n, 0);        // bit 0: loses some pair, bit 1: wins some pair
    for (int64_t k = 0; k < total; ++k) {
        if (!(frac[k] > overlap)) continue;
        const int32_t i = pairs[2 * k], j = pairs[2 * k + 1];
        const bool first_bigger = sigmas[coords[4 * (int64_t)i + 3]] > sigmas[coords[4 * (int64_t)j + 3]];
        role[(size_t)(first_bigger ? j : i)] |= 1;
        role[(size_t)(first_bigger ? i : j)] |= 2;
    }
    for (int b = 0; b < n_blocks; ++b) {
        bool open = false;
        for (int32_t r = offsets[b]; r < offsets[b + 1] && !open; ++r) open = role[(size_t)r] == 3;
        open_blocks[b] = open ? 1 : 0;
        if (!open)
            for (int32_t r = offsets[b]; r < offsets[b + 1]; ++r)
                if (role[(size_t)r] & 1) alive[r] = 0;
    }
    return MMX_OK;
}

// Block tables of the surviving blobs, written straight into the caller's merged table (the arena the pruning
// step works on): per row the reference's 11 columns (magmap/cv/detector.py:88-113, 325-364: z, y, x, radius =
// sigma sqrt(3), confirmed -1, truth -1, channel, abs z, y, x, region -1), coordinates shifted by the block's offset
// in the ROI (stack_detect.py:164-170), `n_extra` further columns left untouched, then the block's grid coordinate in
// the 3 tag columns (chunking.merge_blobs :410-445); and the compact copies the native pruning reads.
//   interior : optional [n_blocks][6] block-relative bounds lo z, y, x, hi z, y, x: rows outside are dropped
//              (detector.get_blobs_interior, applied before the shift as detect_blobs does, :952-955)
//   store    : [.. ][ld] float64, rows from row0 on are written; zyx/tag: [..][3] int32; abs_zyx: [..][3] float64
//   rows_per_block : out [n_blocks]
extern "C" int mmx_host_emit_tables(const int32_t* coords, const uint8_t* alive, const int32_t* offsets, int n_blocks,
                                    const double* sigmas, int n_sigma, double channel, const double* block_offsets,
                                    const int32_t* block_tags, const int32_t* interior,
                                    double* store, int64_t ld, int32_t* zyx, int32_t* tag, double* abs_zyx,
                                    int64_t row0, int64_t capacity, int64_t* rows_per_block)
{
    const int32_t ns = n_sigma;
    return mmx_host_emit_tables_multi(1, &coords, &alive, &offsets, n_blocks, &sigmas, &ns, &channel, block_offsets,
                                      block_tags, interior, store, ld, -1, zyx, tag, abs_zyx, row0, capacity,
                                      rows_per_block, nullptr, nullptr);
}

// The same for blocks detected in SEVERAL channels (the reference's per-channel loop in detect_blobs,
// magmap/cv/detector.py:899-943: `blobs_all.append(...)`, `np.vstack(blobs_all)`): block b's table holds channel 0's rows,
// then channel 1's ..., every channel from its own peak arrays (coords[c] / alive[c] / offsets[c] over the SAME n_blocks
// blocks) and sigma table.  n_extra >= 0: that many columns behind the 11 named ones are zeroed (the co-localisation
// flags land there: mmx_host_coloc_flags); -1: left untouched.  any_before[b] (optional): some channel held a blob of
// block b before the border exclusion (detect_blobs returns None, not an EMPTY table, when none did, :941-942).
// coloc_rows (optional): [rows][5] int32 per written row -- block, z, y, x (block-relative), channel: what
// mmx_coloc_means takes as d_blobs.
extern "C" int mmx_host_emit_tables_multi(int32_t n_channels, const int32_t* const* coords, const uint8_t* const* alive,
                                          const int32_t* const* offsets, int n_blocks, const double* const* sigmas,
                                          const int32_t* n_sigma, const double* channel_ids,
                                          const double* block_offsets, const int32_t* block_tags,
                                          const int32_t* interior, double* store, int64_t ld, int32_t n_extra,
                                          int32_t* zyx, int32_t* tag, double* abs_zyx, int64_t row0, int64_t capacity,
                                          int64_t* rows_per_block, uint8_t* any_before, int32_t* coloc_rows)
{
    if (n_channels < 1 || n_channels > 64 || !coords || !alive || !offsets || n_blocks < 1 || !sigmas || !n_sigma ||
        !channel_ids || !block_offsets || !block_tags || !store || !zyx || !tag || !abs_zyx || row0 < 0 ||
        !rows_per_block || n_extra < -1 || ld < 14 + (n_extra > 0 ? n_extra : 0))
        return MMX_ERR_ARG;
    int64_t n = 0;
    for (int c = 0; c < n_channels; ++c) {
        if (!offsets[c] || !sigmas[c]) return MMX_ERR_ARG;
        const int64_t nc = offsets[c][n_blocks];
        if (nc && (!coords[c] || !alive[c])) return MMX_ERR_ARG;
        n += nc;
    }
    auto inside = [&](int b, const int32_t* c) {
        if (!interior) return true;
        const int32_t* q = interior + 6 * (int64_t)b;
        return c[0] >= q[0] && c[1] >= q[1] && c[2] >= q[2] && c[0] < q[3] && c[1] < q[4] && c[2] < q[5];
    };
    std::vector<int64_t> first((size_t)n_blocks + 1, row0);
    for (int b = 0; b < n_blocks; ++b) {
        int64_t k = 0;
        bool any = false;
        for (int c = 0; c < n_channels; ++c)
            for (int32_t r = offsets[c][b]; r < offsets[c][b + 1]; ++r) {
                const int32_t* q = coords[c] + 4 * (int64_t)r;
                if (q[3] < 0 || q[3] >= n_sigma[c]) return MMX_ERR_ARG;
                any = any || alive[c][r];
                k += alive[c][r] && inside(b, q);
            }
        rows_per_block[b] = k;
        if (any_before) any_before[b] = any ? 1 : 0;
        first[(size_t)b + 1] = first[(size_t)b] + k;
    }
    if (first[(size_t)n_blocks] > capacity) return MMX_ERR_WORKSPACE;
    const double root3 = std::sqrt(3.0);
    // (block b on thread b mod T, as in the resolve step that wrote its rows: they are in that core's cache)
    const int T = n < 2000 ? 1 : std::min(pool::get().size(), 16);
    parallel(std::min(T, n_blocks), [&](int t, int nt) {
        for (int b = t; b < n_blocks; b += nt) {
            int64_t at = first[(size_t)b];
            const double* off = block_offsets + 3 * (int64_t)b;
            const int32_t* tg = block_tags + 3 * (int64_t)b;
            for (int ch = 0; ch < n_channels; ++ch)
                for (int32_t r = offsets[ch][b]; r < offsets[ch][b + 1]; ++r) {
                    const int32_t* c = coords[ch] + 4 * (int64_t)r;
                    if (!alive[ch][r] || !inside(b, c)) continue;
                    double* o = store + at * ld;
                    for (int a = 0; a < 3; ++a) {
                        const double p = (double)c[a] + off[a];
                        o[a] = p; o[7 + a] = p;
                        abs_zyx[3 * at + a] = p;
                        zyx[3 * at + a] = (int32_t)p;
                        tag[3 * at + a] = tg[a];
                        o[ld - 3 + a] = (double)tg[a];
                    }
                    o[3] = sigmas[ch][c[3]] * root3;
                    o[4] = -1.0; o[5] = -1.0; o[6] = channel_ids[ch]; o[10] = -1.0;
                    for (int e = 0; e < n_extra; ++e) o[11 + e] = 0.0;
                    if (coloc_rows) {
                        int32_t* q = coloc_rows + 5 * (at - row0);
                        q[0] = b; q[1] = c[0]; q[2] = c[1]; q[3] = c[2]; q[4] = (int32_t)channel_ids[ch];
                    }
                    ++at;
                }
        }
    });
    return MMX_OK;
}

// The co-localisation flags of a batch of block tables from the per-blob channel means (colocalizer.colocalize_blobs,
// magmap/cv/colocalizer.py:372-441 with the default thresh "min"), written into the tables' extra columns:
//   means[k][r]  : mean of image channel mean_channels[k] over the voxels blob r owns (mmx_coloc_means), NaN when it owns
//                  none; channels without a means row count as NaN everywhere
//   rows[r]      : block, z, y, x (block-relative), channel of blob r (mmx_host_emit_tables_multi's coloc_rows);
//                  row_offsets[n_blocks + 1] delimits the blocks; shapes[b][3] the block extents
//   per block: the channels PRESENT among its in-ROI blobs; for each, the threshold is the smallest mean of that channel
//   over the channel's own in-ROI blobs (np.amin: a NaN among them poisons it -- then nothing reaches it) and every
//   in-ROI blob whose mean of that channel reaches the threshold gets flag 1; blobs outside the ROI keep zeros.
//   flags        : &store[row0][11]: row r's flags at flags[r * ld + c], c < n_channels (zeroed by the caller)
// MMX_ERR_ARG when a blob's channel is not an image channel (the reference raises IndexError there).
extern "C" int mmx_host_coloc_flags(const double* means, const int32_t* mean_channels, int32_t n_mean_channels, int64_t n,
                                    const int32_t* rows, const int64_t* row_offsets, int n_blocks, const int32_t* shapes,
                                    int32_t n_channels, double* flags, int64_t ld)
{
    if (n < 0 || n_blocks < 0 || n_channels < 1 || n_channels > 64 || n_mean_channels < 0 || ld < n_channels ||
        (n && (!rows || !flags)) || (n_blocks && (!row_offsets || !shapes)) || (n_mean_channels && (!means || !mean_channels)))
        return MMX_ERR_ARG;
    int slot_of[64];
    for (int c = 0; c < 64; ++c) slot_of[c] = -1;
    for (int k = 0; k < n_mean_channels; ++k) {
        if (mean_channels[k] < 0 || mean_channels[k] >= n_channels) return MMX_ERR_ARG;
        slot_of[mean_channels[k]] = k;
    }
    const double nan = std::numeric_limits<double>::quiet_NaN();
    auto mean_of = [&](int64_t r, int c) { return slot_of[c] < 0 ? nan : means[(int64_t)slot_of[c] * n + r]; };
    std::vector<int> bad((size_t)std::max(1, n_blocks), 0);
    const int T = n < 4000 ? 1 : std::min(pool::get().size(), 16);
    parallel(std::max(1, std::min(T, n_blocks)), [&](int t, int nt) {
        std::vector<uint8_t> in_roi;
        for (int b = t; b < n_blocks; b += nt) {
            const int64_t lo = row_offsets[b], hi = row_offsets[b + 1];
            if (hi <= lo) continue;
            const int32_t* shp = shapes + 3 * (int64_t)b;
            in_roi.assign((size_t)(hi - lo), 0);
            uint64_t present = 0;
            for (int64_t r = lo; r < hi; ++r) {
                const int32_t* q = rows + 5 * r;
                const bool in = q[1] >= 0 && q[1] < shp[0] && q[2] >= 0 && q[2] < shp[1] && q[3] >= 0 && q[3] < shp[2];
                in_roi[(size_t)(r - lo)] = in ? 1 : 0;
                if (!in) continue;
                if (q[4] < 0 || q[4] >= n_channels) { bad[(size_t)b] = 1; break; }
                present |= uint64_t(1) << q[4];
            }
            if (bad[(size_t)b]) continue;
            for (int other = 0; other < n_channels; ++other) {
                if (!((present >> other) & 1)) continue;
                double thr = std::numeric_limits<double>::infinity();
                bool poisoned = false;
                for (int64_t r = lo; r < hi && !poisoned; ++r) {
                    if (!in_roi[(size_t)(r - lo)] || rows[5 * r + 4] != other) continue;
                    const double m = mean_of(r, other);
                    if (m != m) poisoned = true;
                    else if (m < thr) thr = m;
                }
                if (poisoned) continue;         // (`means >= nan` is False everywhere)
                for (int64_t r = lo; r < hi; ++r)
                    if (in_roi[(size_t)(r - lo)] && mean_of(r, other) >= thr) flags[r * ld + other] = 1.0;
            }
        }
    });
    for (int b = 0; b < n_blocks; ++b)
        if (bad[(size_t)b]) return MMX_ERR_ARG;
    return MMX_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// A SMALL stack -- all its blocks in one batch (the GUI's ROI, a grid-search step: magmap/cv/detector.py:931-933 called
// once per ROI) -- from the re-scored candidate table to the final table in ONE call: peak decisions
// (mmx_host_resolve_peaks), per-block overlap prune (mmx_host_overlap_prune), block tables into the merged table
// (mmx_host_emit_tables), the three pruning passes over the whole table (mmx_host_prune_region) and the gather in the
// final columns (mmx_host_take_rows_final).  Each piece is the entry point of its own name; what this saves is the
// host language between them -- five calls' worth of array set-up, as long as the kernels of such a stack.
// MMX_DEFERRED (not an error; stats[6] says why): a decision needs the caller -- two equal peak values in a block
// (NumPy's argsort order: 1), float32 values further than eps / 4 from the exact ones (a wider band: 2), an overlap
// fraction on the knife edge or a blob that both wins and loses a pair (the reference's libm calls and pair order: 3),
// more rows than the tables take (4).  Nothing has been written to the merged table then: the caller takes the
// call-by-call path on the same candidates.
extern "C" int mmx_host_finish_stack(const mmx_finish_stack_args* a)
{
    if (!a || !a->blocks || a->n_blocks < 1 || a->n_sigma < 1 || !a->sigmas || !a->block_offsets || !a->block_tags ||
        !a->store || !a->zyx || !a->tag || !a->abs_zyx || !a->rows_per_block || !a->n_sections || !a->bounds ||
        !a->last_end || !a->tol || !a->nxt_lo || !a->nxt_hi || !a->n_slab || !a->n_after || !a->n_next || !a->src_cols ||
        !a->out || !a->out_rows || !a->stats || (a->n_total && !a->cands) || a->n_cands > a->n_total)
        return MMX_ERR_ARG;
    double* st = a->stats;
    for (int i = 0; i < 8; ++i) st[i] = 0.0;
    *a->out_rows = 0;
    const int nb = a->n_blocks;
    const size_t nc = std::max<size_t>(1, a->n_cands);
    std::vector<int32_t> nz_coords(4 * nc), coords(4 * nc), offsets((size_t)nb + 1, 0);
    std::vector<double> nz_vals(nc), vals(nc);
    std::vector<uint8_t> ties((size_t)nb, 0);
    double rst[4] = {0, 0, 0, 0};
    int rc = mmx_host_resolve_peaks(a->cands, a->n_cands, a->n_total, a->blocks, nb, a->n_sigma, a->thr, nz_coords.data(),
                                    nz_vals.data(), coords.data(), vals.data(), offsets.data(), ties.data(), rst);
    if (rc != MMX_OK) return rc;
    st[0] = rst[0]; st[1] = rst[1]; st[2] = rst[2]; st[3] = rst[3];
    if (a->n_cands && !(rst[2] < 0.25 * a->eps)) { st[6] = 2; return MMX_DEFERRED; }     // (also a non-finite value)
    for (int b = 0; b < nb; ++b)
        if (ties[(size_t)b]) { st[6] = 1; return MMX_DEFERRED; }
    const int64_t n = offsets[(size_t)nb];
    std::vector<uint8_t> alive((size_t)std::max<int64_t>(1, n), 1), open_blocks((size_t)nb, 0);
    int64_t n_pairs = 0, n_knife = 0;
    if (n) {
        int64_t cap = std::max<int64_t>(1024, 4 * n);
        for (;;) {
            std::vector<int32_t> pairs((size_t)(2 * cap));
            std::vector<double> frac((size_t)cap);
            std::fill(alive.begin(), alive.end(), (uint8_t)1);
            rc = mmx_host_overlap_prune(coords.data(), offsets.data(), nb, a->sigmas, a->n_sigma, a->overlap,
                                        a->overlap_band, alive.data(), open_blocks.data(), pairs.data(), frac.data(), cap,
                                        &n_pairs, &n_knife);
            if (rc != MMX_OK) return rc;
            if (n_pairs <= cap) break;
            cap = n_pairs + 64;
        }
        st[4] = (double)n_pairs;
        if (n_knife) { st[6] = 3; return MMX_DEFERRED; }
        for (int b = 0; b < nb; ++b)
            if (open_blocks[(size_t)b]) { st[6] = 3; return MMX_DEFERRED; }
    }
    int64_t n_alive = 0;
    for (int64_t r = 0; r < n; ++r) n_alive += alive[(size_t)r];
    st[5] = (double)n_alive;
    if (n_alive > a->capacity || n_alive > a->out_capacity) { st[6] = 4; return MMX_DEFERRED; }
    const int32_t* cp = coords.data();
    const uint8_t* ap = alive.data();
    const int32_t* op = offsets.data();
    const int32_t ns = a->n_sigma;
    rc = mmx_host_emit_tables_multi(1, &cp, &ap, &op, nb, &a->sigmas, &ns, &a->channel, a->block_offsets, a->block_tags,
                                    a->interior, a->store, a->ld, -1, a->zyx, a->tag, a->abs_zyx, 0, a->capacity,
                                    a->rows_per_block, a->any_before, nullptr);
    if (rc != MMX_OK) return rc;
    int64_t rows = 0;
    for (int b = 0; b < nb; ++b) rows += a->rows_per_block[b];
    if (rows == 0) return MMX_OK;
    // the three passes work on a private copy of the absolute coordinates: the per-block tables stay as detected
    std::vector<double> abs_cur(a->abs_zyx, a->abs_zyx + 3 * rows);
    std::vector<int64_t> cur((size_t)rows), keep((size_t)rows);
    for (int64_t r = 0; r < rows; ++r) cur[(size_t)r] = r;
    int64_t kept = 0;
    rc = mmx_host_prune_region(a->zyx, a->tag, abs_cur.data(), cur.data(), rows, INT64_MIN, INT64_MAX, a->n_sections,
                               a->bounds, a->last_end, a->tol, a->nxt_lo, a->nxt_hi, keep.data(), nullptr, &kept,
                               a->n_slab, a->n_after, a->n_next, a->stat_ld);
    if (rc != MMX_OK) return rc;
    rc = mmx_host_take_rows_final(a->store, a->ld, keep.data(), kept, a->src_cols, a->n_out, abs_cur.data(), a->abs_dst0,
                                  a->out);
    if (rc != MMX_OK) return rc;
    *a->out_rows = kept;
    return MMX_OK;
}
