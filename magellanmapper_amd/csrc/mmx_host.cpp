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
#include <vector>

#include <unistd.h>

#include "../../include/mmx.h"

namespace {
// A small persistent pool: the tables have ~3e5 rows, so one parallel section is a fraction of a
// millisecond of work and creating threads per section would cost as much as the work itself.
// fork(): the child inherits the pool object but none of its threads (the reference's default start method is
// "fork"), so a parallel section there would wait for workers that do not exist.  The pool remembers the pid it
// was built in; in any other process every section runs on the calling thread.
class pool {
public:
    static pool& get() { static pool p; return p; }
    int size() const { return owner_ == getpid() ? (int)workers_.size() + 1 : 1; }
    // run fn(t, n) for t = 0..n-1, n <= size(); the caller is thread 0
    void run(int n, const std::function<void(int, int)>& fn)
    {
        if (n <= 1) { fn(0, 1); return; }
        std::unique_lock<std::mutex> serial(serial_);          // one section at a time
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &fn; n_ = n; pending_ = n - 1; ++epoch_;
        }
        cv_.notify_all();
        fn(0, n);
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&] { return pending_ == 0; });
        fn_ = nullptr;
    }
private:
    pool()
    {
        const unsigned hw = std::thread::hardware_concurrency();
        const int n = (int)std::max(1u, std::min(16u, hw ? hw : 1u));
        owner_ = getpid();
        for (int t = 1; t < n; ++t) workers_.emplace_back([this, t] { loop(t); });
    }
    ~pool()
    {
        if (owner_ != getpid()) {          // a forked child: the threads were never here
            for (auto& w : workers_) w.detach();
            return;
        }
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; ++epoch_; }
        cv_.notify_all();
        for (auto& w : workers_) w.join();
    }
    void loop(int t)
    {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int, int)>* fn = nullptr;
            int n = 0;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return epoch_ != seen; });
                seen = epoch_;
                if (stop_) return;
                if (t < n_) { fn = fn_; n = n_; }
            }
            if (fn) {
                (*fn)(t, n);
                std::lock_guard<std::mutex> lk(m_);
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }
    std::vector<std::thread> workers_;
    pid_t owner_ = 0;
    std::mutex m_, serial_;
    std::condition_variable cv_, done_;
    const std::function<void(int, int)>* fn_ = nullptr;
    int n_ = 0, pending_ = 0;
    uint64_t epoch_ = 0;
    bool stop_ = false;
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

extern "C" int mmx_host_prune_axis(const int32_t* zyx, const int32_t* tag, double* abs_zyx,
                                   const int64_t* cur, int64_t n_cur, int axis, int n_sections,
                                   const double* bounds, double last_end, const int32_t tol[3],
                                   const double* nxt_lo, const double* nxt_hi,
                                   int64_t* out_cur, int64_t* out_n,
                                   int64_t* n_slab, int64_t* n_after, int64_t* n_next)
{
    if (!zyx || !tag || !abs_zyx || (!cur && n_cur) || !bounds || !tol || !out_cur || !out_n ||
        !n_slab || !n_after || !n_next || axis < 0 || axis > 2 || n_sections < 2 || n_cur < 0)
        return MMX_ERR_ARG;
    static const bool prof = getenv("MMX_PRUNE_PROF") != nullptr;
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
            for (int j = 0; j < n_slabs; ++j)
                if (nxt_lo[j] == nxt_lo[j] && pos >= nxt_lo[j] && pos < nxt_hi[j]) ++P.n_next[(size_t)j];
            // region = (number of bounds <= pos) - 1   (np.searchsorted(bounds, pos, side="right") - 1)
            const int region = (int)(std::upper_bound(bounds, bounds + n_regions, pos) - bounds) - 1;
            if (region < 0 || !(pos < last_end)) continue;
            const int sec = region >> 1;
            if ((region & 1) == 0) { group[(size_t)i] = sec; continue; }
            ++P.n_slab[(size_t)sec];
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
            int64_t kept = (int64_t)C.size();
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
                    if (hit[k]) { group[(size_t)C[k]] = -1; --kept; }
            }
            n_after[j] = (int64_t)M.size() + kept;
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
