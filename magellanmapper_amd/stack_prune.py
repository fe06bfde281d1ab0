"""Overlap de-duplication of the per-block blob tables (the reference's ``StackPruner``, magmap/cv/stack_detect.py:618-861)
and its region-wise form that runs while the GPU is still detecting (``_RegionPruner``).  Split out of ``stack_detect.py``
(round 6); ``stack_detect`` re-exports every name, the reference's import paths keep working."""
from __future__ import annotations

import ctypes
import os
import sys
from time import time
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _native as nat
from . import chunking, config, detector, roi_prof

_logger = config.logger.getChild(__name__)

#: print the phases of the pruning step to stderr (tools/prune_prof.py)
PRUNE_PROF = False

_COORDS_CACHE: dict = {}


def grid_coords(grid):
    """``list(np.ndindex(*grid))``, remembered per grid shape (a tuple of tuples: nobody writes to it)."""
    grid = tuple(int(v) for v in grid)
    hit = _COORDS_CACHE.get(grid)
    if hit is None:
        if len(_COORDS_CACHE) >= 16:
            _COORDS_CACHE.clear()
        hit = _COORDS_CACHE[grid] = tuple(np.ndindex(*grid))
    return hit


def _region_reach(tol3) -> np.ndarray:
    """How far beyond a region's extent rows can influence the pruning of the region's own rows: a pass matches
    rows up to ``tol`` apart and depends on the outcome of the passes before it, three passes in all; one ``tol``
    of margin on top."""
    return 4 * np.asarray(tol3, dtype=np.int64)


def _rows_within(zyx: np.ndarray, lo: np.ndarray, hi: np.ndarray) -> np.ndarray:
    """Row numbers of ``zyx`` inside the box ``[lo, hi)``."""
    return np.flatnonzero(np.all((zyx >= lo) & (zyx < hi), axis=1))


_REGION_POOL = [None, 0]        # the executor the regions run on, and the process it was made in


def _region_workers():
    """A few threads for pruning regions side by side (the native call releases the GIL), kept for the life of the
    process: making eight threads costs as much as pruning a region.  A forked child makes its own (an executor does
    not survive a fork: its threads are gone, and it would wait for them)."""
    if _REGION_POOL[0] is None or _REGION_POOL[1] != os.getpid():
        from concurrent.futures import ThreadPoolExecutor
        _REGION_POOL[0] = ThreadPoolExecutor(max_workers=min(8, max(1, (os.cpu_count() or 2) // 2)),
                                             thread_name_prefix="mmx-region")
        _REGION_POOL[1] = os.getpid()
    return _REGION_POOL[0]


class _RegionPruner:
    """The overlap pruning of one process' table done region by region while later blocks are still being detected.

    A region is a run of consecutive blocks (one row of the block grid along x); it is pruned -- all three passes,
    ``StackPruner._prune_table`` on its own rows plus the rows of neighbouring regions within reach -- as soon as
    it and its neighbours have landed, which leaves the last few regions and the merge for the end of the step.
    Results equal the whole-table passes (``mmx_host_prune_region`` says why); ``StackPruner.prune_blobs_mp`` uses
    them when it is called with the very parameters they were made for, and prunes the whole table otherwise."""

    def __init__(self, arena: _TableArena, plan, channels, sub_roi_slices, shape3, share, halo=None, min_regions=1):
        self.arena, self.plan, self.channels = arena, plan, list(channels)
        # several ranks: the row ranges (behind the arena's own rows) of the seam rows received from the ranks before
        # and after this one -- every region sees them as the first and the last part of its local table
        self.halo = halo
        grid = sub_roi_slices.shape
        coords = grid_coords(grid)
        run = max(1, int(grid[2]))
        # (a rank's share pruned in one go after the exchange: 32 blocks are four x-rows -- half rows give every
        #  region thread something to do)
        while run > 1 and -(-len(share) // run) < min_regions:
            run = -(-run // 2)
        reach = _region_reach(plan["tol"])
        self.regions = []
        for k0 in range(0, len(share), run):
            ks = range(k0, min(k0 + run, len(share)))
            ext = np.array([[s.indices(n)[:2] for s, n in zip(sub_roi_slices[coords[share[k]]], shape3)] for k in ks])
            lo, hi = ext[:, :, 0].min(axis=0), ext[:, :, 1].max(axis=0)
            self.regions.append(dict(k_lo=ks[0], k_hi=ks[-1] + 1, lo=lo - reach, hi=hi + reach, box=(lo, hi)))
        # neighbours: regions whose extent reaches into this one's box (all pairs at once)
        box_lo = np.array([r["box"][0] for r in self.regions])
        box_hi = np.array([r["box"][1] for r in self.regions])
        lo = np.array([r["lo"] for r in self.regions])
        hi = np.array([r["hi"] for r in self.regions])
        touch = np.all(box_lo[None, :, :] < hi[:, None, :], axis=2) & np.all(box_hi[None, :, :] > lo[:, None, :], axis=2)
        np.fill_diagonal(touch, False)
        k_hi = np.array([r["k_hi"] for r in self.regions])
        for i, r in enumerate(self.regions):
            near = np.flatnonzero(touch[i])
            r["near"] = [int(j) for j in near]
            r["ready_at"] = int(max(r["k_hi"], k_hi[near].max(initial=0)))
        self.done = [None] * len(self.regions)
        self.pending = list(range(len(self.regions)))
        self._futures = []
        self._channels = np.ascontiguousarray(self.channels, dtype=np.float64)

    def matches(self, arena, plan, channels) -> bool:
        same = arena is self.arena and list(channels) == self.channels and plan["n_keys"] == self.plan["n_keys"]
        same = same and np.array_equal(plan["tol"], self.plan["tol"])
        for a, b in zip(plan["axes"], self.plan["axes"]):
            same = same and ((a is None) == (b is None))
            if same and a is not None:
                same = all(np.array_equal(a[k], b[k], equal_nan=True) for k in ("bounds", "nxt_lo", "nxt_hi")) and \
                       a["last_end"] == b["last_end"]
        return bool(same)

    def advance(self) -> None:
        """Prune every region whose blocks and neighbours have all landed (``arena.row_end`` tells), in whatever
        order they become ready; several at once on a few threads (the native call releases the GIL)."""
        landed = len(self.arena.row_end) - 1
        ready = [i for i in self.pending if self.regions[i]["ready_at"] <= landed]
        if not ready:
            return
        self.pending = [i for i in self.pending if self.regions[i]["ready_at"] > landed]
        # (not waited for: towards the end of a stack the batches are small and the host thread is what the step waits
        #  for -- 1.5 ms per batch when the regions ran inside this call; finish() collects them)
        self._submit(ready)

    def _submit(self, regions) -> None:
        """Queue ``regions`` on the region threads: one job per thread at most (a hand-off costs 30-50 us, a third of a
        small region's pruning), each job its share of the regions in turn."""
        pool = _region_workers()
        n_jobs = max(1, min(len(regions), getattr(pool, "_max_workers", 8)))
        for j in range(n_jobs):
            self._futures.append(pool.submit(self._run_many, regions[j::n_jobs]))

    def _run_many(self, regions) -> None:
        for i in regions:
            self._run(i)

    def _run(self, i: int) -> None:
        """One region: its rows and its neighbours' rows within reach, straight from the arena
        (``mmx_host_prune_parts``: the local table is put together natively)."""
        ar, r = self.arena, self.regions[i]
        # (this may run beside the arena growing: the rows it reads have landed and never change, and these references
        #  keep the arrays it reads them from alive should the arena move to larger ones meanwhile)
        a_zyx, a_tag, a_abs, a_store = ar.zyx, ar.tag, ar.abs, ar.store
        ends = ar.row_end
        members = sorted(r["near"] + [i])
        ranges = [[ends[self.regions[j]["k_lo"]], ends[self.regions[j]["k_hi"]]] for j in members]
        own_at = members.index(i)
        if self.halo is not None:       # (local order: earlier ranks' seam rows, own regions, later ranks' seam rows)
            ranges = [list(self.halo[0])] + ranges + [list(self.halo[1])]
            own_at += 1
        parts = np.array(ranges, dtype=np.int64)
        n_own = int(ends[r["k_hi"]] - ends[r["k_lo"]])
        ids = np.empty(max(1, n_own), dtype=np.int64)
        keys = np.empty(max(1, n_own), dtype=np.int64)
        abs_rows = np.empty((max(1, n_own), 3))
        out_n = ctypes.c_int64(0)
        ld = self.plan["max_slabs"]
        stat = np.zeros((3, len(self.channels), 3, ld), dtype=np.int64)        # [kind][channel][axis][slab]
        one_channel = len(self.channels) == 1 and (ar.chan_lo == ar.chan_hi == self.channels[0] or ar.n == 0)
        n_sec, bounds, last_end, tol3, nxt_lo, nxt_hi = self.plan["c_args"]
        lo = np.ascontiguousarray(r["lo"], dtype=np.int32)
        hi = np.ascontiguousarray(r["hi"], dtype=np.int32)
        nat.check(nat.lib().mmx_host_prune_parts(
            a_zyx.ctypes.data, a_tag.ctypes.data, a_abs.ctypes.data,
            None if one_channel else a_store.ctypes.data + 6 * 8, a_store.strides[0] // 8,
            parts.ctypes.data, len(parts), own_at,
            lo.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), hi.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
            self._channels.ctypes.data, len(self.channels), n_sec, bounds, last_end, tol3, nxt_lo, nxt_hi,
            self.plan["n_keys"], ids.ctypes.data, keys.ctypes.data, abs_rows.ctypes.data, ctypes.byref(out_n),
            stat[0].ctypes.data, stat[1].ctypes.data, stat[2].ctypes.data, ld), "mmx_host_prune_parts")
        k = out_n.value
        self.done[i] = (ids[:k], keys[:k], abs_rows[:k], np.moveaxis(stat, 0, -1))

    def cancel(self) -> None:
        """Give up on pruning ahead (``prune_blobs_mp`` was called with other parameters, the arena is no longer
        intact, the detection failed): regions not started are dropped, running ones are waited for, and an
        exception a region raised surfaces here instead of vanishing with its future."""
        self.pending = []
        futures, self._futures = self._futures, []
        for f in futures:
            f.cancel()
        for f in futures:
            if not f.cancelled():
                f.result()

    def run_all(self) -> None:
        """Every region at once (everything has landed), waited for: an exception of a region surfaces here."""
        if self.pending:
            todo, self.pending = self.pending, []
            self._submit(todo)
        futures, self._futures = self._futures, []
        failure = None
        for f in futures:
            try:
                f.result()
            except Exception as exc:        # (the others are still waited for: they read arrays the caller owns)
                failure = failure or exc
        if failure is not None:
            raise failure

    def finish(self, abs_inds, final=None, _lap=lambda what: None):
        """Whatever is left, then the merge: ``(final table, counts)``.  ``final = (source columns, place of the abs
        coordinates)``: the table in those columns (``StackPruner._final_columns``)."""
        self.run_all()              # (everything has landed by now)
        _lap("  regions: the last ones done")
        ar = self.arena
        counts = sum(d[3] for d in self.done)
        ncol = ar.store.shape[1] - 3
        if final is not None:
            # the regions' survivor lists go to the merge as they are (no concatenation: 12 MB of copies for 3e5 rows)
            src, dst0, n_main = final
            parts = [d for d in self.done if len(d[0])]
            n_rows = np.array([len(d[0]) for d in parts], dtype=np.int64)
            ptrs = [(ctypes.c_void_p * max(1, len(parts)))(*[d[c].ctypes.data for d in parts]) for c in range(3)]
            total = int(n_rows.sum())
            out = np.empty((total, n_main))
            rest = np.empty((total, len(src) - n_main)) if n_main < len(src) else None
            nat.check(nat.lib().mmx_host_gather_parts_by_key_split(
                ar.store.ctypes.data, ar.store.strides[0] // 8, len(parts), ptrs[0], ptrs[1], ptrs[2],
                n_rows.ctypes.data, self.plan["n_keys"] * len(self.channels), (ctypes.c_int32 * len(src))(*src),
                len(src), dst0, out.ctypes.data, total, n_main, None if rest is None else rest.ctypes.data),
                "mmx_host_gather_parts_by_key_split")
            _lap("  regions: merge by key, final columns")
            if rest is not None:
                out = out.view(_FinalTable)
                out.coloc_cols = rest
            return out, counts
        ids = np.ascontiguousarray(np.concatenate([d[0] for d in self.done]), dtype=np.int64)
        keys = np.ascontiguousarray(np.concatenate([d[1] for d in self.done]), dtype=np.int64)
        abs_rows = np.ascontiguousarray(np.concatenate([d[2] for d in self.done]), dtype=np.float64)
        _lap("  regions: survivors concatenated")
        out = np.empty((len(ids), ncol))
        cols3 = (ctypes.c_int32 * 3)(*[int(v) for v in abs_inds])
        nat.check(nat.lib().mmx_host_gather_by_key(
            ar.store.ctypes.data, ar.store.strides[0] // 8, ids.ctypes.data, keys.ctypes.data, len(ids),
            self.plan["n_keys"] * len(self.channels), ncol, abs_rows.ctypes.data, cols3, out.ctypes.data),
            "mmx_host_gather_by_key")
        return out, counts


class _FinalTable(np.ndarray):
    """A pruned table that left ``StackPruner.prune_blobs_mp(..., final_form=True)`` already in the reference's final
    columns (rel <- abs, abs and unnamed columns dropped): ``col_names`` are the columns it holds.  ``coloc_cols``: for a
    table with co-localisation columns, the columns the reference reads the flags from (``[:, 10:10 + C]`` of the pruned
    table, stack_detect.py:463-464), row for row, as float64 -- ``None`` otherwise."""
    col_names = None
    coloc_cols = None


class StackPruner:
    """Removes duplicates of blobs that were detected in two overlapping blocks."""
    blobs_to_prune = None

    @classmethod
    def prune_overlap_by_index(cls, i):
        return cls.prune_overlap(i, cls.blobs_to_prune[i])

    @classmethod
    def prune_overlap(cls, i, pruner):
        """One overlap slab: rows tagged block ``i`` along ``axis`` are the master set,
        rows tagged ``i + 1`` are checked against it (:643-677)."""
        blobs, axis, tol, blobs_next = pruner
        if blobs is None:
            return None, None
        tag_col = blobs.shape[1] - 3 + axis
        n_orig = len(blobs)
        master = blobs[blobs[:, tag_col] == i]
        check = blobs[blobs[:, tag_col] == i + 1]
        pruned, master = detector.remove_close_blobs(check, master, tol)
        after = np.concatenate((master, pruned))
        ratios = None
        if blobs_next is not None:
            ratios = detector.meas_pruning_ratio(n_orig, len(after), len(blobs_next))
        return after, ratios

    @staticmethod
    def _axis_geometry(axis, shape3, overlap, overlap_padding, sub_roi_slices, sub_rois_offsets):
        """``(start_j, end_j)`` of the blocks along ``axis`` and whether the reference's regions tile it:
        pass 0 | slab 0 | pass 1 | ... with slab j = [end_j - shift, end_j + pad) ending exactly where pass
        j + 1 = [start_{j+1} + shift, ...) begins, and no region of negative length."""
        n_sections = sub_rois_offsets.shape[axis]
        shift = overlap[axis] + overlap_padding[axis]
        spans = []
        for j in range(n_sections):
            coord = [0, 0, 0]
            coord[axis] = j
            start = int(sub_rois_offsets[tuple(coord)][axis])
            spans.append((start, start + len(range(*sub_roi_slices[tuple(coord)][axis].indices(shape3[axis])))))
        regular = True
        for j, (start, end) in enumerate(spans):
            pass_lo = start + (shift if j > 0 else 0)
            if j < n_sections - 1:
                regular &= pass_lo <= end - shift                                   # pass j, then slab j
                regular &= end + overlap_padding[axis] == spans[j + 1][0] + shift   # slab j meets pass j + 1
            else:
                regular &= pass_lo <= end
        return spans, bool(regular)

    @classmethod
    def _prune_blobs_general(cls, merged, shape3, overlap, tol, sub_roi_slices, sub_rois_offsets, channels,
                             overlap_padding):
        """The reference's region arithmetic as it stands (stack_detect.py:679-861), on materialised tables:
        used when the regions do not tile an axis -- blocks not much larger than their overlap, where a
        truncated block at the far face or an overlap beyond the block stride makes slabs overlap each
        other and passes empty.  The reference then lists a blob once per region it falls into and drops
        those tagged for neither block of a slab; the index-based fast path cannot express that."""
        coord_last = tuple(np.subtract(sub_roi_slices.shape, 1))
        ratio_cols = ("blobs", "ratio_pruning", "ratio_adjacent")
        ratios_all, blobs_all = {}, []
        for chl in channels:
            blobs = detector.Blobs.blobs_in_channel(merged, chl)
            for axis in range(3):
                n_sections = sub_rois_offsets.shape[axis]
                if n_sections <= 1:
                    continue
                spans, _ = cls._axis_geometry(axis, shape3, overlap, overlap_padding, sub_roi_slices,
                                              sub_rois_offsets)
                shift = overlap[axis] + overlap_padding[axis]
                pos = blobs[:, axis]
                passes, pruners = [], []
                for j, (start, end) in enumerate(spans):
                    lo = start + (shift if j > 0 else 0)
                    if j < n_sections - 1:
                        slab = blobs[(pos >= end - shift) & (pos < end + overlap_padding[axis])]
                        nxt_lo = end + tol[axis]
                        nxt_hi = nxt_lo + overlap[axis] + 2 * overlap_padding[axis]
                        roi_end = sub_rois_offsets[coord_last][axis] + (end - start)
                        nxt = None
                        if nxt_lo < roi_end and nxt_hi < roi_end:
                            nxt = blobs[(pos >= nxt_lo) & (pos < nxt_hi)]
                        passes.append(blobs[(pos < end - shift) & (pos >= lo)])
                        pruners.append((slab, axis, tol, nxt))
                    else:
                        passes.append(blobs[(pos < end) & (pos >= lo)])
                        pruners.append((None, axis, tol, None))
                kept = []
                for j, pruner in enumerate(pruners):
                    after, ratios = cls.prune_overlap(j, pruner)
                    if after is not None:
                        kept.append(after)
                    if ratios:
                        for col, val in zip(ratio_cols, ratios):
                            ratios_all.setdefault(col, []).append(val)
                blobs = np.concatenate(passes + kept)
            blobs_all.append(blobs)
        return np.vstack(blobs_all)[:, :-3], ratios_all

    #: the last few block geometries: (ids of the slice / offset arrays, shape, overlap, tol, padding) -> (plan, regular).
    #: A stack detected again and again (a step loop, channel groups) hands over the very same ``Blocks`` arrays; the
    #: entry keeps them alive, so an id cannot come back as another array.  (Editing a ``Blocks`` array in place between
    #: calls is not supported -- the reference builds them once per call and never writes to them.)
    _geometry_cache: dict = {}

    @classmethod
    def _geometry(cls, shape3, overlap, tol, overlap_padding, sub_roi_slices, sub_rois_offsets):
        """``(plan, regular)``: :meth:`_axis_plan` and whether every axis with more than one section is tiled by the
        reference's regions (:meth:`_axis_geometry`), remembered per block geometry."""
        key = (id(sub_roi_slices), id(sub_rois_offsets), tuple(int(v) for v in shape3),
               np.asarray(overlap).tobytes(), np.asarray(tol).tobytes(), np.asarray(overlap_padding).tobytes())
        hit = cls._geometry_cache.get(key)
        if hit is None:
            regular = all(cls._axis_geometry(a, shape3, overlap, overlap_padding, sub_roi_slices, sub_rois_offsets)[1]
                          for a in range(3) if sub_rois_offsets.shape[a] > 1)
            plan = cls._axis_plan(shape3, overlap, tol, overlap_padding, sub_roi_slices, sub_rois_offsets) if regular else None
            if len(cls._geometry_cache) >= 8:
                cls._geometry_cache.clear()
            hit = cls._geometry_cache[key] = (plan, regular, sub_roi_slices, sub_rois_offsets)
        return hit[0], hit[1]

    @classmethod
    def _axis_plan(cls, shape3, overlap, tol, overlap_padding, sub_roi_slices, sub_rois_offsets):
        """The constants of the three passes (regular geometry): per axis ``None`` (one section: no pass) or the
        region boundaries ``[pass 0 | slab 0 | pass 1 | ...]``, the far end, and the "adjacent region" of every
        slab's pruning-ratio statistic (reference :757-785); plus the tolerances."""
        grid = sub_roi_slices.shape
        coord_last = tuple(np.subtract(grid, 1))
        axes = []
        for axis in range(3):
            n_sections = sub_rois_offsets.shape[axis]
            if n_sections <= 1:
                axes.append(None)
                continue
            # The axis is tiled by [pass 0][slab 0][pass 1][slab 1] ... [pass last]; slab j
            # = [end_j - (overlap + pad), end_j + pad) belongs to the boundary j | j + 1.
            shift = overlap[axis] + overlap_padding[axis]
            bounds, nxt_lo, nxt_hi = [], [], []
            last_end = 0
            for j in range(n_sections):
                coord = [0, 0, 0]
                coord[axis] = j
                coord = tuple(coord)
                start = sub_rois_offsets[coord][axis]
                extent = len(range(*sub_roi_slices[coord][axis].indices(shape3[axis])))
                end = start + extent
                last_end = end
                bounds.append(start + (shift if j > 0 else 0))          # pass j begins
                if j < n_sections - 1:
                    bounds.append(end - shift)                          # slab j begins
                    lo = end + tol[axis]
                    hi = lo + overlap[axis] + 2 * overlap_padding[axis]
                    roi_end = sub_rois_offsets[coord_last][axis] + extent
                    ok = lo < roi_end and hi < roi_end
                    nxt_lo.append(lo if ok else np.nan)
                    nxt_hi.append(hi if ok else np.nan)
            axes.append(dict(n_sections=int(n_sections), bounds=np.asarray(bounds, dtype=np.float64),
                             last_end=float(last_end), nxt_lo=np.asarray(nxt_lo, dtype=np.float64),
                             nxt_hi=np.asarray(nxt_hi, dtype=np.float64)))
        tol3 = np.array([int(v) for v in np.broadcast_to(np.asarray(tol), (3,))], dtype=np.int32)
        n_keys = 1
        for ax in axes:
            if ax is not None:
                n_keys *= 3 * ax["n_sections"] - 2
        plan = dict(axes=axes, tol=tol3, n_keys=int(n_keys),
                    max_slabs=max([1] + [ax["n_sections"] - 1 for ax in axes if ax]))

        def ptrs(name):
            return (ctypes.c_void_p * 3)(*[None if ax is None else ax[name].ctypes.data for ax in axes])

        # the same constants as the native calls take them (the arrays above stay alive in `axes`)
        plan["c_args"] = ((ctypes.c_int32 * 3)(*[0 if ax is None else ax["n_sections"] for ax in axes]), ptrs("bounds"),
                          (ctypes.c_double * 3)(*[0.0 if ax is None else ax["last_end"] for ax in axes]),
                          tol3.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), ptrs("nxt_lo"), ptrs("nxt_hi"))
        return plan

    @classmethod
    def _prune_table(cls, zyx, tags, abs_cur, chan, own_lo, own_hi, channels, plan, need_keys: bool = False):
        """The three passes over one table (``mmx_host_prune_region``), channel by channel: ``(rows, keys, counts)``
        -- the ids of the surviving rows among ``[own_lo, own_hi)`` in their final order, the key of each (the
        channel's position in ``channels`` is the most significant part), and the statistics
        ``counts[channel][axis][slab] = (rows in the slab, rows left, rows in the adjacent region)`` over own rows.
        ``abs_cur`` is updated in place.  ``chan``: channel of every row, ``None`` when all belong to ``channels[0]``.
        ``need_keys``: the survivors will be merged with other tables' (several ranks), so the keys are computed even
        when the own rows happen to be the whole table -- a rank that received no halo rows (the other ranks have no
        blobs, or none near the seam) still needs them; without it the keyless shortcut is taken for one region."""
        lib = nat.lib()
        n_sec, bounds, last_end, tol3, nxt_lo, nxt_hi = plan["c_args"]
        ld = plan["max_slabs"]
        # one region whose output order is final: no keys needed
        whole = own_lo == 0 and own_hi == len(zyx) and not need_keys
        counts = np.zeros((len(channels), 3, ld, 3), dtype=np.int64)
        rows_all, keys_all = [], []
        for ci, chl in enumerate(channels):
            if chan is None:
                cur = np.arange(len(zyx), dtype=np.int64)
            else:
                cur = np.flatnonzero(chan == chl).astype(np.int64, copy=False)      # row ids, table order (np.isin of a
                #                                                              scalar: 4 ms per 4e5 rows)
            out_rows = np.empty(len(cur), dtype=np.int64)
            out_keys = None if whole else np.empty(len(cur), dtype=np.int64)
            out_n = ctypes.c_int64(0)
            stat = np.zeros((3, 3, ld), dtype=np.int64)       # [kind][axis][slab]
            nat.check(lib.mmx_host_prune_region(
                zyx.ctypes.data, tags.ctypes.data, abs_cur.ctypes.data, cur.ctypes.data, len(cur),
                -(1 << 63) if whole else int(own_lo), (1 << 63) - 1 if whole else int(own_hi), n_sec, bounds, last_end,
                tol3, nxt_lo, nxt_hi, out_rows.ctypes.data, None if whole else out_keys.ctypes.data,
                ctypes.byref(out_n), stat[0].ctypes.data, stat[1].ctypes.data, stat[2].ctypes.data, ld),
                "mmx_host_prune_region")
            counts[ci] = np.moveaxis(stat, 0, -1)
            rows_all.append(out_rows[:out_n.value])
            if not whole:
                keys_all.append(out_keys[:out_n.value] + ci * plan["n_keys"])
        rows = rows_all[0] if len(rows_all) == 1 else np.concatenate(rows_all)
        keys = None if whole else (keys_all[0] if len(keys_all) == 1 else np.concatenate(keys_all))
        return (np.ascontiguousarray(rows, dtype=np.int64),
                None if keys is None else np.ascontiguousarray(keys, dtype=np.int64), counts)

    @staticmethod
    def _ratios_from_counts(counts, plan):
        """Pruning-ratio columns (reference :673-676, 836-838) from the slab statistics, in the reference's order:
        channels, then axes, then slabs."""
        ratios_all = {}
        for per_channel in counts:
            for axis, ax in enumerate(plan["axes"]):
                if ax is None:
                    continue
                for j in range(ax["n_sections"] - 1):
                    if np.isnan(ax["nxt_lo"][j]):
                        continue
                    n_slab, n_after, n_next = (int(v) for v in per_channel[axis][j])
                    ratios = detector.meas_pruning_ratio(n_slab, n_after, n_next)
                    if ratios:
                        for col, val in zip(("blobs", "ratio_pruning", "ratio_adjacent"), ratios):
                            ratios_all.setdefault(col, []).append(val)
        return ratios_all

    @staticmethod
    def _final_columns(merged, abs_inds, n_flag_cols: int = 0):
        """What the reference's last steps on the pruned table (``replace_rel_with_abs_blob_coords``, [the flags read
        from ``[:, 10:10 + C]``,] ``remove_abs_blob_coords(True)``, :455-470) leave of the merged table's columns, for
        the gather to write directly: ``(source columns, place of the abs coordinates among them, names of the final
        columns, how many of the source columns they are)`` -- with ``n_flag_cols`` = C co-localisation columns behind the
        named ones the source columns end with the C columns the flags are read from -- or ``None`` where the steps do not
        reduce to that (columns beyond the named ones that were not announced, an unusual registry, a table the native
        gather does not take)."""
        if not (merged.dtype == np.float64 and merged.strides[1] == 8 and merged.strides[0] % 8 == 0):
            return None
        registry = detector.Blobs._col_inds
        named = [(c, i) for c, i in registry.items() if i is not None]
        n_flag_cols = int(n_flag_cols)
        if merged.shape[1] - 3 != len(named) + n_flag_cols or sorted(i for _, i in named) != list(range(len(named))):
            return None                 # (columns beyond the named ones that nobody announced)
        rel = detector.Blobs._get_rel_inds()
        drop = set(abs_inds)
        keep = [(c, i) for c, i in named if i not in drop]
        src = [i for _, i in keep]
        if any(r is None for r in rel) or rel[0] not in src:
            return None
        dst0 = src.index(rel[0])
        if src[dst0:dst0 + 3] != list(rel):
            return None
        if n_flag_cols and len(named) != 11:
            return None                 # (the reference's literal `10:10 + C` is only what it means with the 11 standard columns)
        flag_src = list(range(10, 10 + n_flag_cols))        # (the literal columns of stack_detect.py:464, region first)
        return src + flag_src, dst0, [c.value for c, _ in keep], len(src)

    @staticmethod
    def _take_rows(merged, rows, abs_cur, abs_inds, final=None):
        """``merged[rows][:, :-3]`` with the three abs columns taken from ``abs_cur[rows]``; with ``final = (source
        columns, place of the abs coordinates)`` the table in those columns instead (:meth:`_final_columns`)."""
        ncol = merged.shape[1]
        if final is not None:
            src, dst0, n_main = final
            out = np.empty((len(rows), n_main))
            rest = np.empty((len(rows), len(src) - n_main)) if n_main < len(src) else None
            nat.check(nat.lib().mmx_host_take_rows_split(
                merged.ctypes.data, merged.strides[0] // 8, rows.ctypes.data, len(rows),
                (ctypes.c_int32 * len(src))(*src), len(src), abs_cur.ctypes.data, dst0, out.ctypes.data, n_main,
                None if rest is None else rest.ctypes.data), "mmx_host_take_rows_split")
            if rest is not None:
                out = out.view(_FinalTable)
                out.coloc_cols = rest
            return out
        if merged.dtype == np.float64 and merged.strides[1] == 8 and merged.strides[0] % 8 == 0:
            out = np.empty((len(rows), ncol - 3))
            cols3 = (ctypes.c_int32 * 3)(*[int(v) for v in abs_inds])
            nat.check(nat.lib().mmx_host_take_rows(
                merged.ctypes.data, merged.strides[0] // 8, rows.ctypes.data, len(rows), ncol - 3,
                abs_cur.ctypes.data, cols3, out.ctypes.data), "mmx_host_take_rows")
            return out
        out = np.take(merged, rows, axis=0)[:, :-3]
        out[:, abs_inds] = np.take(abs_cur, rows, axis=0)
        return out

    @classmethod
    def _prune_distributed(cls, seg_rois, shape3, plan, sub_roi_slices, channels, final=None):
        """Several ranks, each holding the tables of its own blocks (``seg_rois.local_only``): every rank prunes
        its own rows -- the three passes on its rows plus the other ranks' rows within reach of its blocks
        (``mmx_host_prune_region``) -- and the survivors are merged by key on every rank.  Collective: all ranks
        call it, all get the same ``(table, counts)``; ``(None, None)`` when no rank holds a table.

        Two exchanges (RCCL all_gather over xGMI on GPUs): the rows near another rank's blocks -- a few per cent of
        the table: 10 values a row --, then the surviving rows in their final form with their keys."""
        from . import dist
        ar = seg_rois.arena
        world, me = dist.world_size(), dist.rank()
        from time import perf_counter
        _prof = PRUNE_PROF and (me == 0 or dist._loopback is not None)
        _t = [perf_counter()]

        def _lap(what):
            if _prof:
                now = perf_counter()
                print(f"distributed prune, rank {me}: {what}: {(now - _t[0]) * 1e3:.2f} ms", file=sys.stderr)
                _t[0] = now
        # Every rank-local stage runs under try / except and its failure travels with the NEXT collective (a status
        # word in the all_reduce, in the row counts of the two exchanges): a rank that fails -- a native error, tables
        # of the wrong width -- makes every rank raise at that collective instead of leaving the others waiting in it.
        grid = sub_roi_slices.shape
        coords = grid_coords(grid)
        n = ar.n
        ncol = ar.store.shape[1]
        failure, payload, boxes, reach, abs_inds = None, None, None, None, None
        has_table = False
        try:
            has_table = any(seg_rois[c] is not None and not isinstance(seg_rois[c], (int, np.integer)) for c in coords)
            abs_inds = detector.Blobs._get_abs_inds()
            reach = _region_reach(plan["tol"])
            boxes = cls._rank_boxes(len(coords), world, coords, sub_roi_slices, shape3, reach)
            payload = cls._seam_rows(ar, boxes, me, reach)
        except Exception as exc:
            failure = exc
        flags = dist.all_reduce_sum(np.array([1 if has_table else 0, n, 0 if failure is None else 1], dtype=np.int64))
        if failure is not None:
            raise failure
        if flags[2]:
            raise RuntimeError("distributed pruning failed on another rank before the first exchange; see its log")
        if flags[0] == 0:
            return None, None
        _lap("rows near the other ranks' blocks")
        parts = dist.all_gather_rows(payload, 10)
        _lap("exchange 1 (seam rows)")
        mine, counts = None, None
        try:
            mine, counts = cls._prune_own_rows(ar, parts, boxes[me], me, channels, plan, abs_inds, _lap, final,
                                               (sub_roi_slices, shape3, dist.my_share(len(coords))))
        except Exception as exc:
            failure = exc
        width = (ncol - 3) if final is None else len(final[0])          # columns of a survivor's row; its key follows
        blocks_, n_per_rank, _ = dist.all_gather_rows_padded(mine, width + 1, failure, "distributed pruning (own rows)")
        _lap("exchange 2 (survivors)")
        out = None
        try:
            total = int(sum(n_per_rank)) if blocks_ is not None else 0
            out = np.empty((total, width))
            if total:       # (every rank's block as the exchange left it; keys: the column behind the table's own)
                n_rows = np.ascontiguousarray(n_per_rank, dtype=np.int64)
                step = blocks_.strides[0]
                ptrs = (ctypes.c_void_p * len(n_per_rank))(*[blocks_.ctypes.data + r * step for r in range(len(n_per_rank))])
                nat.check(nat.lib().mmx_host_merge_parts_by_key(
                    ptrs, n_rows.ctypes.data, len(n_per_rank), blocks_.strides[1] // 8, plan["n_keys"] * len(channels),
                    width, out.ctypes.data, total), "mmx_host_merge_parts_by_key")
        except Exception as exc:
            failure = exc
        _lap("merge by key")
        summed = dist.all_reduce_sum(np.append(counts.reshape(-1), 0 if failure is None else 1))
        if failure is not None:
            raise failure
        if summed[-1]:
            raise RuntimeError("distributed pruning: the merge failed on another rank; see its log")
        counts = summed[:-1].reshape(counts.shape)
        _lap("counts all_reduce")
        return out, counts

    _rank_box_cache: dict = {}
    #: own rows from which a rank prunes its blocks region by region (below: one region, no thread hand-offs)
    REGION_MIN_ROWS = 8000

    @classmethod
    def _rank_boxes(cls, n_blocks, world, coords, sub_roi_slices, shape3, reach):
        """The extent of every rank's blocks, widened by the reach of the pruning (``None`` for a rank without
        blocks); remembered per block geometry like :meth:`_geometry` (256 blocks: a millisecond of slice arithmetic
        per call otherwise)."""
        from . import dist
        key = (id(sub_roi_slices), int(n_blocks), int(world), tuple(int(v) for v in shape3), np.asarray(reach).tobytes())
        hit = cls._rank_box_cache.get(key)
        if hit is not None:
            return hit[0]
        boxes = []
        for q in range(world):
            lo_b, hi_b = dist.share_bounds(n_blocks, q, world)
            if hi_b <= lo_b:
                boxes.append(None)
                continue
            ext = np.array([[s.indices(m)[:2] for s, m in zip(sub_roi_slices[coords[i]], shape3)]
                            for i in range(lo_b, hi_b)])
            boxes.append((ext[:, :, 0].min(axis=0) - reach, ext[:, :, 1].max(axis=0) + reach))
        if len(cls._rank_box_cache) >= 8:
            cls._rank_box_cache.clear()
        cls._rank_box_cache[key] = (boxes, sub_roi_slices)
        return boxes

    @staticmethod
    def _seam_rows(ar, boxes, me, reach):
        """The rows of this rank's arena that lie within reach of another rank's blocks, ten values a row:
        detection coordinates, block tags, absolute coordinates, channel (``mmx_host_rows_in_boxes``)."""
        n = ar.n
        own = boxes[me]
        near = []
        for q, box in enumerate(boxes):
            if q == me or box is None or own is None or not n:
                continue
            # (both boxes carry the reach: a rank whose blocks are further away than twice that cannot hold a row in it)
            if np.any(own[0] + reach >= box[1]) or np.any(own[1] - reach <= box[0]):
                continue
            near.append(box)
        if not near:
            return np.empty((0, 10))
        lo = np.ascontiguousarray([b[0] for b in near], dtype=np.int32)
        hi = np.ascontiguousarray([b[1] for b in near], dtype=np.int32)
        payload = np.empty((n, 10))
        k = ctypes.c_int64(0)
        nat.check(nat.lib().mmx_host_rows_in_boxes(
            ar.zyx.ctypes.data, ar.tag.ctypes.data, ar.abs.ctypes.data, ar.store.ctypes.data + 6 * 8,
            ar.store.strides[0] // 8, n, lo.ctypes.data, hi.ctypes.data, len(near), payload.ctypes.data, n,
            ctypes.byref(k)), "mmx_host_rows_in_boxes")
        return payload[:k.value]

    @classmethod
    def _prune_own_rows(cls, ar, parts, mine_box, me, channels, plan, abs_inds, _lap=lambda what: None, final=None,
                        geometry=None):
        """The three passes on this rank's rows between the seam rows received from the ranks before and after it:
        ``(own survivors in their final form + one column with the key that places them, statistics)``.

        The received rows are appended to the arena's compact columns behind the rank's own rows
        (``mmx_host_append_rows``) and ``mmx_host_prune_parts`` is told the order of the local table -- earlier
        ranks' halo, own rows, later ranks' halo: what the whole-table passes would see of them -- so that no table is
        put together in Python; the survivors leave through ``mmx_host_emit_survivors``.

        ``geometry = (sub_roi_slices, shape3, this rank's block indices)``: with enough rows the rank's blocks are
        pruned region by region on a few threads, as one process does while it detects (:class:`_RegionPruner`), every
        region seeing the seam rows as the first and last part of its table -- one region for the whole rank is a
        single thread's 4-6 ms at two to four ranks."""
        lib = nat.lib()
        n = ar.n
        ncol = ar.store.shape[1]
        halo = [(q, p) for q, p in enumerate(parts) if q != me and mine_box is not None and len(p)]
        room = n + sum(len(p) for _, p in halo)
        if room > ar.cap:
            ar._grow(room)
        lo = np.ascontiguousarray(mine_box[0] if mine_box is not None else (0, 0, 0), dtype=np.int32)
        hi = np.ascontiguousarray(mine_box[1] if mine_box is not None else (0, 0, 0), dtype=np.int32)
        lo_p, hi_p = (v.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)) for v in (lo, hi))
        at = n
        edges = [n]
        for side in (lambda q: q < me, lambda q: q > me):
            for q, p in halo:
                if not side(q):
                    continue
                p = np.ascontiguousarray(p, dtype=np.float64)
                k = ctypes.c_int64(0)
                nat.check(lib.mmx_host_append_rows(
                    p.ctypes.data, len(p), lo_p, hi_p, ar.zyx.ctypes.data, ar.tag.ctypes.data, ar.abs.ctypes.data,
                    ar.store.ctypes.data + 6 * 8, ar.store.strides[0] // 8, at, ar.cap, ctypes.byref(k)),
                    "mmx_host_append_rows")
                at += k.value
            edges.append(at)
        # local order: halo of the ranks before, own rows, halo of the ranks after
        local = np.array([[edges[0], edges[1]], [0, n], [edges[1], edges[2]]], dtype=np.int64)
        _lap("own + halo tables")
        if geometry is not None and mine_box is not None and n >= cls.REGION_MIN_ROWS and \
                len(ar.row_end) == len(geometry[2]) + 1 and ar.row_end[-1] == n:
            rp = _RegionPruner(ar, plan, channels, geometry[0], geometry[1], list(geometry[2]),
                               halo=((edges[0], edges[1]), (edges[1], edges[2])),
                               min_regions=getattr(_region_workers(), "_max_workers", 8))
            if len(rp.regions) > 1:
                rp.run_all()
                _lap(f"three passes on own + halo rows ({len(rp.regions)} regions)")
                counts = np.ascontiguousarray(sum(d[3] for d in rp.done))
                width = (ncol - 3) if final is None else len(final[0])
                mine = np.empty((sum(len(d[0]) for d in rp.done), width + 1))
                row = 0
                csrc = None if final is None else (ctypes.c_int32 * width)(*final[0])
                cols3 = (ctypes.c_int32 * 3)(*[int(v) for v in abs_inds])
                # (regions in order: the merge by key is stable)
                live = [d for d in rp.done if len(d[0])]
                if final is not None:       # one threaded pass over all the regions' lists
                    if live:
                        n_rows = np.array([len(d[0]) for d in live], dtype=np.int64)
                        ptrs = [(ctypes.c_void_p * len(live))(*[d[c].ctypes.data for d in live]) for c in range(3)]
                        nat.check(lib.mmx_host_emit_parts_final(
                            ar.store.ctypes.data, ar.store.strides[0] // 8, len(live), ptrs[0], ptrs[1], ptrs[2],
                            n_rows.ctypes.data, csrc, width, final[1], mine.ctypes.data, len(mine)),
                            "mmx_host_emit_parts_final")
                else:                       # (tables with co-localisation columns: region by region)
                    for r_ids, r_keys, r_abs, _ in live:
                        k = len(r_ids)
                        nat.check(lib.mmx_host_emit_survivors(
                            ar.store.ctypes.data, ar.store.strides[0] // 8, r_ids.ctypes.data, r_keys.ctypes.data, k,
                            width, r_abs.ctypes.data, cols3, mine[row:row + k].ctypes.data), "mmx_host_emit_survivors")
                        row += k
                _lap("own survivors in final form")
                return mine, counts
        ids = np.empty(max(1, n), dtype=np.int64)
        keys = np.empty(max(1, n), dtype=np.int64)
        abs_rows = np.empty((max(1, n), 3))
        out_n = ctypes.c_int64(0)
        ld = plan["max_slabs"]
        stat = np.zeros((3, len(channels), 3, ld), dtype=np.int64)        # [kind][channel][axis][slab]
        n_sec, bounds, last_end, tol3, nxt_lo, nxt_hi = plan["c_args"]
        chans = np.ascontiguousarray(channels, dtype=np.float64)
        # (every row takes part that lies inside the box: append_rows has filtered the halo already, the own rows are
        #  the own part, which is never filtered)
        nat.check(lib.mmx_host_prune_parts(
            ar.zyx.ctypes.data, ar.tag.ctypes.data, ar.abs.ctypes.data, ar.store.ctypes.data + 6 * 8,
            ar.store.strides[0] // 8, local.ctypes.data, 3, 1, lo_p, hi_p, chans.ctypes.data, len(channels),
            n_sec, bounds, last_end, tol3, nxt_lo, nxt_hi, plan["n_keys"], ids.ctypes.data, keys.ctypes.data,
            abs_rows.ctypes.data, ctypes.byref(out_n), stat[0].ctypes.data, stat[1].ctypes.data, stat[2].ctypes.data,
            ld), "mmx_host_prune_parts")
        k = out_n.value
        counts = np.ascontiguousarray(np.moveaxis(stat, 0, -1))
        _lap("three passes on own + halo rows")
        if final is not None:       # (the survivors leave in the table's final columns: fewer values to exchange and merge)
            src, dst0 = final
            mine = np.empty((k, len(src) + 1))
            if k:
                nat.check(lib.mmx_host_emit_survivors_final(
                    ar.store.ctypes.data, ar.store.strides[0] // 8, ids.ctypes.data, keys.ctypes.data, k,
                    (ctypes.c_int32 * len(src))(*src), len(src), abs_rows.ctypes.data, dst0, mine.ctypes.data),
                    "mmx_host_emit_survivors_final")
            _lap("own survivors in final form")
            return mine, counts
        mine = np.empty((k, ncol - 2))
        if k:
            cols3 = (ctypes.c_int32 * 3)(*[int(v) for v in abs_inds])
            nat.check(lib.mmx_host_emit_survivors(
                ar.store.ctypes.data, ar.store.strides[0] // 8, ids.ctypes.data, keys.ctypes.data, k, ncol - 3,
                abs_rows.ctypes.data, cols3, mine.ctypes.data), "mmx_host_emit_survivors")
        _lap("own survivors in final form")
        return mine, counts

    @classmethod
    def prune_blobs_mp(cls, img, seg_rois, overlap, tol, sub_roi_slices, sub_rois_offsets,
                       channels, overlap_padding=None, final_form: bool = False, untouched: bool = False,
                       n_flag_cols: int = 0):
        """Prune duplicates in the overlap slabs, per channel, axis by axis (:679-861).

        For every axis with more than one block, every block boundary ``j | j+1`` defines a
        slab ``[end_j - (overlap + pad), end_j + pad)`` spanning the whole plane; blobs in it
        are de-duplicated between the two block generations (:meth:`prune_overlap`),
        everything else passes through, and the recombined table goes on to the next axis.
        Returns ``(table, DataFrame)`` or ``(None, None)``.  ``final_form`` (not in the reference; ``_StackRun`` asks
        for it): where possible the table comes back as a :class:`_FinalTable`, already in the columns the reference's
        next two steps would leave (rel <- abs, abs dropped) -- two passes over the whole table less; ``n_flag_cols``
        = C says the tables carry C co-localisation columns behind the 11 named ones, and the columns the reference reads
        the flags from come back beside the table (``_FinalTable.coloc_cols``).  ``untouched``:
        the caller vouches that nobody has had the tables since ``detect_blobs_sub_rois`` returned them (``_StackRun``
        calls one right after the other), which spares the sampled comparison that looks for in-place edits -- 5000
        cache misses on a 3e5-row table, 0.3 ms.

        Same results and row order as the reference, but rows are tracked as indices into the
        merged table (only the 3 abs columns ever change), so the big table is gathered once, and
        the per-axis classify / match / reorder step is native host code
        (``mmx_host_prune_axis``; the de-duplication stays on the host as in the reference).
        """
        import pandas as pd
        _prof = PRUNE_PROF
        _t = [time()]

        def _lap(what):
            if _prof:
                _t.append(time())
                print(f"prune_blobs_mp {what}: {(_t[-1] - _t[-2]) * 1e3:.2f} ms", file=sys.stderr)

        if overlap_padding is None:
            overlap_padding = tol
        shape3 = img.shape[:3]
        if getattr(seg_rois, "local_only", False):
            # several ranks, each with the tables of its own blocks: a collective (every rank calls this)
            detector.Blobs(np.ones((1, 4))).format_blobs()      # bind the class-level column registry
            plan = cls._geometry(shape3, overlap, tol, overlap_padding, sub_roi_slices, sub_rois_offsets)[0]
            if plan is None:            # (cannot be: the tables stay on their ranks only for a regular geometry)
                plan = cls._axis_plan(shape3, overlap, tol, overlap_padding, sub_roi_slices, sub_rois_offsets)
            # (the same decision on every rank: it follows from the arena's width and the registry alone)
            final = (cls._final_columns(seg_rois.arena.store, detector.Blobs._get_abs_inds(), n_flag_cols)
                     if final_form else None)
            out, counts = cls._prune_distributed(seg_rois, shape3, plan, sub_roi_slices, channels,
                                                 None if final is None else final[:2])
            if out is None:
                return None, None
            if final is not None:
                rest = None
                if final[3] < len(final[0]):        # (the merge leaves one table: final columns | the flags' columns)
                    rest = np.ascontiguousarray(out[:, final[3]:])
                    out = np.ascontiguousarray(out[:, :final[3]])
                out = out.view(_FinalTable)
                out.col_names, out.coloc_cols = final[2], rest
            return out, cls._ratio_frame(cls._ratios_from_counts(counts, plan))
        arena = getattr(seg_rois, "arena", None)
        if arena is not None and not arena.intact(seg_rois, sample_columns=not untouched):
            arena = None
        _lap("arena check")
        early = getattr(seg_rois, "pruner", None)
        if early is not None:
            seg_rois.pruner = None        # one shot: used below or cancelled
        try:
            merged = arena.store[:arena.n] if arena is not None and arena.n else chunking.merge_blobs(seg_rois)
        except Exception:
            if early is not None:
                early.cancel()
            raise
        if merged is None:
            if early is not None:
                early.cancel()
            return None, None
        grid = sub_roi_slices.shape
        coord_last = tuple(np.subtract(grid, 1))
        ratio_cols = ("blobs", "ratio_pruning", "ratio_adjacent")
        ratios_all = {}
        plan, regular = cls._geometry(shape3, overlap, tol, overlap_padding, sub_roi_slices, sub_rois_offsets)
        if not regular:
            if early is not None:
                early.cancel()
            out, ratios_all = cls._prune_blobs_general(merged, shape3, overlap, tol, sub_roi_slices,
                                                       sub_rois_offsets, channels, overlap_padding)
            return out, pd.DataFrame(ratios_all)
        ncol = merged.shape[1]
        detector.Blobs(merged)      # bind the class-level column registry to the 11 standard columns
        abs_inds = detector.Blobs._get_abs_inds()
        final = cls._final_columns(merged, abs_inds, n_flag_cols) if final_form else None
        gather_as = None if final is None else (final[0], final[1], final[3])
        # regions of this very call finished while the GPU was still detecting (StackDetector.plan_pruning)
        if early is not None and arena is not None and early.matches(arena, plan, channels) and \
                getattr(early, "serves", lambda g: True)(gather_as):
            _lap("set-up (arena check, geometry, registry)")
            out, counts = early.finish(abs_inds, gather_as, _lap)
            _lap("regions pruned during detection: the rest + merge")
        else:
            if early is not None:       # other parameters than planned for, or tables edited since: not usable
                early.cancel()
            chan = detector.Blobs.get_blobs_channel(merged)
            # compact columns for the native step (libmmx_hip.so: mmx_host_prune_region)
            if arena is not None:                 # filled while the GPU was busy
                zyx, tags = arena.zyx[:arena.n], arena.tag[:arena.n]
            else:
                zyx = np.ascontiguousarray(merged[:, :3], dtype=np.int32)  # detection coordinates never change
                tags = np.ascontiguousarray(merged[:, ncol - 3:], dtype=np.int32)
            # the only values pruning changes (a private copy: the per-block tables stay as detected)
            if arena is not None and list(abs_inds) == [7, 8, 9]:
                abs_cur = arena.abs[:arena.n].copy()
            else:
                abs_cur = np.ascontiguousarray(merged[:, abs_inds], dtype=np.float64)
            one_channel = arena is not None and len(channels) == 1 and arena.chan_lo == arena.chan_hi == channels[0]
            _lap("set-up (arena check, geometry, column copies)")
            rows, _, counts = cls._prune_table(zyx, tags, abs_cur, None if one_channel else chan, 0, len(zyx),
                                               channels, plan)
            _lap("three axis passes")
            out = cls._take_rows(merged, rows, abs_cur, abs_inds, gather_as)
            _lap("gather of the output table")
        if final is not None:
            rest = getattr(out, "coloc_cols", None)
            out = out.view(_FinalTable)
            out.col_names, out.coloc_cols = final[2], rest
        df = cls._ratio_frame(cls._ratios_from_counts(counts, plan))
        _lap("ratio frame")
        return out, df

    _frame_names: dict = {}

    @staticmethod
    def _ratio_frame(ratios):
        """The pruning-ratio data frame (reference :836-838, 859) from the column lists, without the per-element type
        inference of the dict-of-lists constructor: half the time of a small stack's whole pruning step."""
        import pandas as pd
        cols = {k: np.asarray(v, dtype=np.int64 if k == "blobs" else np.float64) for k, v in ratios.items()}
        # (the frame from ready-made columns: a third of the dict constructor's time, which in turn is what a small
        #  stack's pruning step spends most on; a pandas without that constructor takes the public one)
        n_rows = {len(v) for v in cols.values()}
        if len(n_rows) == 1:
            names = StackPruner._frame_names.get(tuple(cols))
            if names is None:           # (the column index is immutable: made once per set of names)
                names = StackPruner._frame_names[tuple(cols)] = pd.Index(list(cols))
            try:
                return pd.DataFrame._from_arrays(list(cols.values()), names, pd.RangeIndex(n_rows.pop()),
                                                 verify_integrity=False)
            except (AttributeError, TypeError):
                pass
        return pd.DataFrame(cols, copy=False)
