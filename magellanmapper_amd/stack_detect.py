"""Whole-volume blob detection in blocks (mirror of ``magmap.cv.stack_detect``).

Same public surface as the reference (magmap/cv/stack_detect.py): :class:`StackTimes`
(:27-31), :class:`StackDetector` (:34-257), :class:`Blocks` / :func:`setup_blocks`
(:260-335), :func:`detect_blobs_blocks` (:338-517), :func:`detect_blobs_stack` (:520-615),
:class:`StackPruner` (:618-861).

What changes is *where blocks run*: the reference fans blocks out to a
``multiprocessing.Pool`` (:222-257); here the volume is uploaded once and every block of
this rank's share is filtered by the HIP kernels in a few batched launches
(:mod:`magellanmapper_amd.blob_log`).  With ``torch.distributed`` initialised, blocks are
sharded over ranks (one GPU each) and the per-block tables are all-gathered
(:mod:`magellanmapper_amd.dist`); the overlap de-duplication then runs once.  Block geometry,
table layout, pruning rules and all quirks (first-channel profile for block settings, rel <-
abs replacement, dropped columns) are the reference's.
"""
from __future__ import annotations

import ctypes
import os
import sys
from enum import Enum
from time import time
from typing import NamedTuple, Optional, Sequence, Tuple

import numpy as np

from . import _native as nat
from . import chunking, config, detector, roi_prof

_logger = config.logger.getChild(__name__)


class StackTimes(Enum):
    DETECTION = "Detection"
    PRUNING = "Pruning"
    TOTAL = "Total_stack"


#: several ranks with a regular block geometry: every rank prunes its own rows (False: gather, rank 0 prunes, broadcast)
DIST_PRUNE = True
#: prune finished regions of a raw stack while the GPU is still detecting: "" = stacks of 64 blocks and more, "1" /
#: "0" = always / never (what tests and tools set; DESIGN.md section 4b has the measurements)
PRUNE_AHEAD = ""
#: print the phases of the pruning step to stderr (tools/prune_prof.py)
PRUNE_PROF = False
#: small one-batch stacks: the host chain behind the kernels as ONE native call (``_StackFinisher``); False keeps the
#: call-by-call form (what tests compare it with)
STACK_FINISHER = True


class Image5d:
    """Minimal stand-in for ``magmap.io.np_io.Image5d`` (reference np_io.py:33-70): the
    ``(t, z, y, x[, c])`` array plus the attributes this path reads."""

    def __init__(self, img=None, path_img=None, path_meta=None, img_io=None):
        self.img = img
        self.path_img = path_img
        self.path_meta = path_meta
        self.img_io = img_io
        self.subimg_offset = None
        self.subimg_size = None
        self.meta = None
        self.rgb = False
        self.is_roi = False
        #: the first time point already on its way to the device (`prefetch`): what a whole-image detection then reads
        self.device_volume = None

    def prefetch(self):
        """Start the upload of the first time point now (``blob_log.DeviceVolume``: block row by block row on a copy
        stream); a later whole-image ``detect_blobs_blocks`` / ``detect_blobs_stack`` of this image detects on it while
        the rest is still in flight.  What `detect_blobs_tiles` calls for tile k + 1 before it detects tile k.  The image
        must stay as it is until the detection has returned or :meth:`release` has been called (the upload reads it in
        the background)."""
        from . import blob_log as bl
        if self.device_volume is None and self.img is not None:
            cells = None
            try:        # (where the first channel's profile puts the block rows: only the upload ORDER depends on it)
                blocks = setup_blocks(config.get_roi_profile(0), self.img.shape[1:4])
                cells = _upload_cells(blocks.sub_roi_slices, self.img.shape[1:4])
            except Exception:
                pass
            self.device_volume = bl.DeviceVolume(self.img[0], streamed=True, cells=cells)
        return self

    def release(self):
        """Drop the prefetched device copy, cancelling whatever of its upload has not been queued yet."""
        dv, self.device_volume = self.device_volume, None
        if dv is not None:
            dv.close()


def _upload_cells(sub_roi_slices, shape3):
    """``(z ends, y ends)`` of the block grid's layers and rows: where a host image on its way to the device is cut so
    that a block can start once the cells it touches have landed (``volume._SlabUpload``)."""
    gz, gy = sub_roi_slices.shape[:2]
    z_ends = [int(sub_roi_slices[(l, 0, 0)][0].indices(int(shape3[0]))[1]) for l in range(gz)]
    y_ends = [int(sub_roi_slices[(0, j, 0)][1].indices(int(shape3[1]))[1]) for j in range(gy)]
    return z_ends, y_ends


class _TableArena:
    """Per-block tables stored back to back, in block order, while the GPU is still busy:
    the merged table ``chunking.merge_blobs`` would build (``store``: 11 columns + 3 block-tag
    columns) plus the compact columns the pruning step works on.  The per-block tables handed
    out are views of ``store``."""

    def __init__(self, n_cols: int = 11, n_expected: int = 0):
        self.n_cols = n_cols
        self.n_expected = int(n_expected)      # blocks that will be added (0: unknown), for the growth estimate
        #: room kept beyond the rows asked for (a rank's arena: the seam rows of the other ranks are appended behind its
        #: own rows for the pruning -- growing for them would copy all four arrays inside the step's tail)
        self.headroom = 0.0
        self.cap = 4096
        self.store = np.empty((self.cap, n_cols + 3))
        self.zyx = np.empty((self.cap, 3), dtype=np.int32)
        self.tag = np.empty((self.cap, 3), dtype=np.int32)
        self.abs = np.empty((self.cap, 3))
        self.n = 0
        self.spans = {}
        self._views = {}                       # coord -> the view of `store` last handed out for it (view())
        self.chan_lo, self.chan_hi = np.inf, -np.inf       # range of the channel column over all rows
        # rows before the k-th block that was added (blocks arrive in grid order; blocks without rows count too):
        # what the region-wise pruning addresses blocks by
        self.row_end = [0]

    def _grow(self, need: int):
        cap = max(2 * self.cap, need)
        if self.n_expected > len(self.spans) > 0:
            # blocks hold similar numbers of blobs: size for all of them at once (the last doublings would
            # otherwise copy a few hundred thousand rows while the GPU has nothing left to hide them)
            cap = max(cap, int(need * 1.15 * self.n_expected / (len(self.spans) + 1)) + 1024)
        cap = int(cap * (1.0 + self.headroom))
        for name in ("store", "zyx", "tag", "abs"):
            old = getattr(self, name)
            new = np.empty((cap,) + old.shape[1:], dtype=old.dtype)
            new[:self.n] = old[:self.n]
            setattr(self, name, new)
        self.cap = cap

    def add(self, coord, table: np.ndarray) -> None:
        rows = table.shape[0]
        if self.n + rows > self.cap:
            self._grow(self.n + rows)
        a = self.n
        self.store[a:a + rows, :self.n_cols] = table
        self.store[a:a + rows, self.n_cols:] = coord
        self.zyx[a:a + rows] = table[:, :3]
        self.tag[a:a + rows] = coord
        self.abs[a:a + rows] = table[:, 7:10]
        if rows:
            self.chan_lo = min(self.chan_lo, table[:, 6].min())
            self.chan_hi = max(self.chan_hi, table[:, 6].max())
        self.n += rows
        self.spans[tuple(coord)] = (a, a + rows)

    def landed(self, n_blocks: int = 1) -> None:
        """``n_blocks`` more blocks of the share are complete (their rows, if any, are in the arena)."""
        self.row_end.extend([self.n] * n_blocks)

    def view(self, coord):
        """The block's table as a view of the store -- the same object for as long as the store stays where it is
        (``intact`` recognises the tables it handed out by identity)."""
        coord = tuple(coord)
        v = self._views.get(coord)
        if v is None or v.base is not self.store:
            a, b = self.spans[coord]
            v = self._views[coord] = self.store[a:b, :self.n_cols]
        return v

    @classmethod
    def from_rows(cls, idx: np.ndarray, rows: np.ndarray, coords: np.ndarray):
        """The arena of tables that arrive as ONE array in block order (``idx``: block index per row,
        ascending; ``coords``: grid coordinate of every block index): whole-array copies, no per-block loop."""
        self = cls(rows.shape[1], 0)
        n = len(rows)
        self.cap = max(n, 1)
        self.store = np.empty((self.cap, self.n_cols + 3))
        self.store[:n, :self.n_cols] = rows
        tags = coords[idx]
        self.store[:n, self.n_cols:] = tags
        self.zyx = np.ascontiguousarray(rows[:, :3], dtype=np.int32) if n else np.empty((1, 3), dtype=np.int32)
        self.tag = np.ascontiguousarray(tags, dtype=np.int32) if n else np.empty((1, 3), dtype=np.int32)
        self.abs = np.ascontiguousarray(rows[:, 7:10]) if n else np.empty((1, 3))
        self.n = n
        if n:
            blocks, first = np.unique(idx, return_index=True)
            ends = np.append(first[1:], n)
            self.spans = {tuple(int(v) for v in coords[b]): (int(a), int(e)) for b, a, e in zip(blocks, first, ends)}
            self.chan_lo, self.chan_hi = rows[:, 6].min(), rows[:, 6].max()
        return self

    def intact(self, blob_rois, sample_columns: bool = True) -> bool:
        """True when ``blob_rois`` still holds exactly the arena's tables, in grid order (and, with ``sample_columns``,
        a sample of their rows still says what the compact columns say: ``_columns_unedited``)."""
        at = 0
        views = self._views
        # (the very view objects the arena handed out -- assemble_seg_rois' -- are recognised by identity: 256 blocks in
        #  ~30 us; any other array has to share the store's memory: ~4 us each)
        for coord, tbl in zip(StackDetector._grid_coords(blob_rois.shape), blob_rois.ravel().tolist()):
            if tbl is None or isinstance(tbl, (int, np.integer)) or len(tbl) == 0:
                continue
            span = self.spans.get(coord)
            if span is None or span[0] != at:
                return False
            known = views.get(coord)
            if not (tbl is known and known.base is self.store) and not np.shares_memory(tbl, self.store):
                return False
            at = span[1]
        return at == self.n and (not sample_columns or self._columns_unedited())

    def _columns_unedited(self) -> bool:
        """The compact columns the pruning reads (``zyx``, ``abs``, ``tag``: copies made when the rows landed) still
        say what the tables say -- every row (an in-place edit of one small block's table must not slip through; a few
        ms for 3 x 10^5 rows, skipped when ``_StackRun`` vouches for tables nobody else has seen).  Tables handed out by
        ``detect_blobs_sub_rois`` are views of the arena and the reference's API lets a caller edit them in place
        before ``prune_blobs_mp`` (shift them, say); such an edit is meant to be seen, and the arena's shortcuts would
        not see it -- ``prune_blobs_mp`` then works from the tables themselves."""
        n = self.n
        if n == 0:
            return True
        st, nc = self.store[:n], self.n_cols
        return bool(np.array_equal(st[:, :3], self.zyx[:n]) and np.array_equal(st[:, 7:10], self.abs[:n]) and
                    np.array_equal(st[:, nc:], self.tag[:n]))


class _ArenaSink:
    """Finished block tables straight from the native host path into the arena (``mmx_host_emit_tables``): what
    ``detect_blobs`` (11 columns, border exclusion), ``detect_sub_roi`` (shift to ROI coordinates) and
    ``merge_blobs`` (grid-coordinate tags) do per block in the reference, for a whole batch in one native call."""

    def __init__(self, arena: _TableArena, grid_coords, block_offsets, shapes, exclude_of):
        self.arena = arena
        self.grid_coords = np.asarray(grid_coords, dtype=np.int32).reshape(-1, 3)     # per block of this rank's share
        self.block_offsets = np.ascontiguousarray(block_offsets, dtype=np.float64).reshape(-1, 3)
        self.shapes = shapes
        self.exclude_of = exclude_of
        self.pruner = None
        self.pruner_factory = None      # () -> _RegionPruner, called when the first batch lands

    def ensure_pruner(self):
        """The regions' pruner, made when the first rows are about to land (by then every batch is queued)."""
        if self.pruner is None and self.pruner_factory is not None:
            self.pruner, self.pruner_factory = self.pruner_factory(), None
        return self.pruner

    def __call__(self, indices, pb, chl):
        return self.emit(indices, [pb], [chl])

    def emit(self, indices, pbs, chls, flags_fn=None):
        """The tables of one batch of blocks from the peak arrays of every channel they were detected in (``pbs[c]``: a
        ``PeakBatch`` over the same blocks, channel ``chls[c]``): a block's table holds channel 0's rows, then channel
        1's ... (the reference's ``np.vstack`` in ``detect_blobs``, detector.py:943).  With extra columns in the arena
        (co-localisation) ``flags_fn(indices, rows5, row_offsets, flags_ptr, ld)`` fills them for the rows just written
        -- ``rows5``: block, z, y, x (block-relative), channel per row; ``flags_ptr``: address of the first row's first
        extra column -- before the regions are told that the blocks have landed."""
        self.ensure_pruner()
        ar = self.arena
        idx = np.asarray(indices, dtype=np.int64)
        nb = len(idx)
        nch = len(pbs)
        n_extra = ar.n_cols - 11
        need = ar.n + int(sum(int(pb.alive.sum()) for pb in pbs))
        if need > ar.cap:
            ar._grow(need)
        interior = self.interior_of(indices)
        offs = np.ascontiguousarray(self.block_offsets[idx])
        tags = np.ascontiguousarray(self.grid_coords[idx])
        rows = np.zeros(nb, dtype=np.int64)
        any_before = np.zeros(nb, dtype=np.uint8)
        rows5 = np.empty((max(1, need - ar.n), 5), dtype=np.int32) if flags_fn is not None else None
        ptrs = lambda arrs: (ctypes.c_void_p * nch)(*[a.ctypes.data for a in arrs])
        sig = [np.ascontiguousarray(pb.sigmas, dtype=np.float64) for pb in pbs]
        nat.check(nat.lib().mmx_host_emit_tables_multi(
            nch, ptrs([pb.coords for pb in pbs]), ptrs([pb.alive for pb in pbs]), ptrs([pb.offsets for pb in pbs]), nb,
            ptrs(sig), (ctypes.c_int32 * nch)(*[len(v) for v in sig]), (ctypes.c_double * nch)(*[float(c) for c in chls]),
            offs.ctypes.data, tags.ctypes.data, None if interior is None else interior.ctypes.data,
            ar.store.ctypes.data, ar.store.shape[1], n_extra if n_extra > 0 else -1,
            ar.zyx.ctypes.data, ar.tag.ctypes.data, ar.abs.ctypes.data, ar.n, ar.cap, rows.ctypes.data,
            any_before.ctypes.data, None if rows5 is None else rows5.ctypes.data), "mmx_host_emit_tables_multi")
        total = int(rows.sum())
        if flags_fn is not None and total:
            row_offsets = np.concatenate(([0], np.cumsum(rows))).astype(np.int64)
            flags_fn(indices, rows5[:total], row_offsets,
                     ar.store.ctypes.data + (ar.n * ar.store.shape[1] + 11) * 8, ar.store.shape[1])
        out = self.landed(tags, rows, any_before, chls)
        if self.pruner is not None:
            self.pruner.advance()
        return out

    def interior_of(self, indices):
        """``[lo z, y, x, hi z, y, x]`` per block of a batch: the block-relative bounds rows must lie in (border
        exclusion, ``detector.get_blobs_interior``), ``None`` without exclusion."""
        if self.exclude_of is None:
            return None
        interior = np.empty((len(indices), 6), dtype=np.int32)
        for k, i in enumerate(indices):
            ex = self.exclude_of(i)
            lo = np.zeros(3) if ex is None else np.asarray(ex[0], dtype=float)
            hi = np.asarray(self.shapes[i], dtype=float) - (0 if ex is None else np.asarray(ex[1], dtype=float))
            interior[k, :3] = np.ceil(lo)            # integer coordinates: z >= lo  <=>  z >= ceil(lo)
            interior[k, 3:] = np.ceil(hi)            #                      z < hi   <=>  z < ceil(hi)
        return interior

    def landed(self, tags, rows, any_before, chls):
        """Book-keeping for rows a native call has just written behind the arena's last row: the per-block tables
        (views of the store; ``None`` for a block without blobs, an EMPTY table where all were excluded)."""
        ar = self.arena
        out = []
        at = ar.n
        for k in range(len(rows)):
            if not any_before[k]:
                out.append(None)                         # no blobs at all: detect_blobs returns None (:941-942)
            elif rows[k] == 0:
                out.append(np.zeros((0, ar.n_cols)))     # all excluded: an EMPTY table
            else:
                coord = tuple(int(v) for v in tags[k])
                ar.spans[coord] = (at, at + int(rows[k]))
                out.append(ar.store[at:at + int(rows[k]), :ar.n_cols])
                at += int(rows[k])
        if at > ar.n:
            ar.chan_lo, ar.chan_hi = min(ar.chan_lo, *chls), max(ar.chan_hi, *chls)
        ar.n = at
        # (row_end per block of the batch: rows of the blocks before it)
        ends = ar.row_end[-1] + np.cumsum(rows)
        ar.row_end.extend(int(v) for v in ends)
        return out


class _StackFinisher:
    """A SMALL stack -- all its blocks in one batch (the GUI's ROI, a grid-search step) -- from the re-scored candidates
    to the final table in ONE native call (``mmx_host_finish_stack``: peak decisions, per-block overlap prune, block
    tables into the arena, the three pruning passes, the gather in the final columns) instead of five calls with array
    set-up in Python between them: those five are as long as the kernels of such a stack.

    Plays the part of a :class:`_RegionPruner` towards ``StackPruner.prune_blobs_mp``: made by
    ``detect_blobs_sub_rois`` from the planned pruning parameters, it hands its table over when ``prune_blobs_mp`` is
    called with those very parameters and ``final_form`` -- otherwise the arena it filled is pruned as always.  Where a
    decision needs the reference's own calls (equal peak values, a knife-edge overlap, a pruning chain, a band that
    proved too narrow) the native call changes nothing and the batch takes the call-by-call path."""

    def __init__(self, sink: "_ArenaSink", plan, channels):
        self.sink, self.arena, self.plan, self.channels = sink, sink.arena, plan, list(channels)
        self.layout = None          # (source columns, place of the abs coordinates, names, n_main) of the table made
        self.result = None          # (final table, counts)
        self.deferred = 0           # why the last run was left to the caller (mmx_host_finish_stack's stats[6])

    def run(self, indices, cands, n_cands: int, blocks, space, thr: float, eps: float, overlap: float, stats, chl):
        """The tables of the batch (as ``_ArenaSink.emit`` returns them), or ``None`` when the call was deferred."""
        from .host_resolve import OVERLAP_BAND
        ar = self.arena
        if ar.n or self.result is not None or len(self.channels) != 1 or chl != self.channels[0]:
            return None
        layout = StackPruner._final_columns(ar.store, detector.Blobs._get_abs_inds())
        if layout is None or layout[3] != len(layout[0]):
            return None
        nb = len(indices)
        if max(n_cands, 1) > ar.cap:
            ar._grow(n_cands)
        idx = np.asarray(indices, dtype=np.int64)
        offs = np.ascontiguousarray(self.sink.block_offsets[idx])
        tags = np.ascontiguousarray(self.sink.grid_coords[idx])
        interior = self.sink.interior_of(indices)
        sig = np.ascontiguousarray(space.sigmas, dtype=np.float64)
        rows = np.zeros(nb, dtype=np.int64)
        any_before = np.zeros(nb, dtype=np.uint8)
        ld = self.plan["max_slabs"]
        stat = np.zeros((3, 3, ld), dtype=np.int64)           # [kind][axis][slab]
        src = layout[0]
        out = np.empty((max(n_cands, 1), len(src)))
        out_rows = ctypes.c_int64(0)
        st = np.zeros(8)
        n_sec, bounds, last_end, tol3, nxt_lo, nxt_hi = self.plan["c_args"]
        src_c = (ctypes.c_int32 * len(src))(*src)
        a = nat.FinishStackArgs()
        a.cands, a.n_cands, a.n_total = (cands.ctypes.data if len(cands) else None), int(n_cands), len(cands)
        a.blocks, a.n_blocks, a.n_sigma = blocks.ctypes.data, nb, len(sig)
        a.thr, a.eps = float(thr), float(eps)
        a.sigmas, a.overlap, a.overlap_band = sig.ctypes.data, float(overlap), float(OVERLAP_BAND)
        a.channel = float(chl)
        a.block_offsets, a.block_tags = offs.ctypes.data, tags.ctypes.data
        a.interior = None if interior is None else interior.ctypes.data
        a.store, a.ld = ar.store.ctypes.data, ar.store.shape[1]
        a.zyx, a.tag, a.abs_zyx, a.capacity = ar.zyx.ctypes.data, ar.tag.ctypes.data, ar.abs.ctypes.data, ar.cap
        a.rows_per_block, a.any_before = rows.ctypes.data, any_before.ctypes.data
        a.n_sections = ctypes.cast(n_sec, ctypes.c_void_p)
        a.bounds, a.last_end = ctypes.cast(bounds, ctypes.c_void_p), ctypes.cast(last_end, ctypes.c_void_p)
        a.tol = ctypes.cast(tol3, ctypes.c_void_p)
        a.nxt_lo, a.nxt_hi = ctypes.cast(nxt_lo, ctypes.c_void_p), ctypes.cast(nxt_hi, ctypes.c_void_p)
        a.n_slab, a.n_after, a.n_next, a.stat_ld = stat[0].ctypes.data, stat[1].ctypes.data, stat[2].ctypes.data, ld
        a.src_cols, a.n_out, a.abs_dst0 = ctypes.cast(src_c, ctypes.c_void_p), len(src), layout[1]
        a.out, a.out_capacity = out.ctypes.data, len(out)
        a.out_rows = ctypes.cast(ctypes.pointer(out_rows), ctypes.c_void_p)
        a.stats = st.ctypes.data
        rc = nat.lib().mmx_host_finish_stack(ctypes.byref(a))
        if rc == nat.MMX_DEFERRED:
            self.deferred = int(st[6])
            return None
        nat.check(rc, "mmx_host_finish_stack")
        err = float(st[2])
        stats.max_f32_error = max(stats.max_f32_error, err) if n_cands else stats.max_f32_error
        stats.n_contested += int(st[0])
        stats.n_probes += len(cands) - int(n_cands)
        stats.n_peaks += int(st[1])
        stats.n_overlap_pairs += int(st[4])
        stats.n_blobs += int(st[5])
        tables = self.sink.landed(tags, rows, any_before, [chl])
        counts = np.zeros((1, 3, ld, 3), dtype=np.int64)
        counts[0] = np.moveaxis(stat, 0, -1)
        self.layout = layout
        self.result = (out[:out_rows.value], counts)
        return tables

    # ---- towards prune_blobs_mp: the part of a _RegionPruner
    def matches(self, arena, plan, channels) -> bool:
        return self.result is not None and _RegionPruner.matches(self, arena, plan, channels)

    def serves(self, gather_as) -> bool:
        lay = self.layout
        return (gather_as is not None and lay is not None and list(gather_as[0]) == list(lay[0])
                and gather_as[1] == lay[1] and gather_as[2] == lay[3])

    def finish(self, abs_inds, final=None, _lap=lambda what: None):
        return self.result

    def advance(self) -> None:
        pass

    def cancel(self) -> None:
        pass


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
        coords = StackDetector._grid_coords(grid)
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


class _SegRois(np.ndarray):
    """Object array of per-block tables that remembers the arena its tables live in, the regions already pruned
    while the detection ran, and whether it holds this rank's blocks only."""
    arena = None
    pruner = None
    local_only = False


class StackDetector:
    """Detects blobs block by block.  Class attributes mirror the reference's fork-shared
    state (:51-57) but are only informational here."""
    img5d = None
    img = None
    last_coord = None
    denoise_max_shape = None
    exclude_border = None
    coloc = False
    channel = None
    #: counters of the last :meth:`detect_blobs_sub_rois` call (``blob_log.BatchStats``)
    last_stats = None
    #: pruning parameters the NEXT :meth:`detect_blobs_sub_rois` call may prune ahead with (:meth:`plan_pruning`)
    prune_hint = None

    @classmethod
    def plan_pruning(cls, overlap, tol, overlap_padding, channels) -> None:
        """Tell the next :meth:`detect_blobs_sub_rois` call what ``StackPruner.prune_blobs_mp`` will be called
        with, so that finished regions of the stack are pruned while the GPU is still busy with later blocks
        (the host idles through most of a detection).  One shot; purely an optimisation: ``prune_blobs_mp`` checks
        that its own arguments are the ones planned for and that the block tables are still the ones that landed
        (they are views of one arena: editing them IN PLACE between the two calls is allowed, as in the reference, and
        is detected on a sample of rows -- a caller that does so should not plan ahead) and otherwise cancels the
        regions pruned ahead and prunes the whole table as always."""
        cls.prune_hint = (np.asarray(overlap), np.asarray(tol),
                          None if overlap_padding is None else np.asarray(overlap_padding), list(channels))

    @staticmethod
    def _exclude_matrix(coord, last_coord, exclude_border):
        """Border exclusion per block: none on faces that are outer faces of the ROI (:152-157)."""
        if exclude_border is None:
            return None
        exclude = np.array([exclude_border, exclude_border])
        exclude[0, np.equal(coord, 0)] = 0
        exclude[1, np.equal(coord, last_coord)] = 0
        return exclude

    @classmethod
    def _finish_block(cls, segments, shape, exclude, offset):
        if segments is not None and exclude is not None:
            segments = detector.get_blobs_interior(segments, shape, *exclude)
        if segments is not None:
            detector.Blobs.shift_blob_rel_coords(segments, offset)
            detector.Blobs.shift_blob_abs_coords(segments, offset)
        return segments

    @classmethod
    def detect_sub_roi_from_data(cls, coord, sub_roi_slices, offset):
        return cls.detect_sub_roi(coord, offset, cls.last_coord, cls.denoise_max_shape,
                                  cls.exclude_border, cls.img5d, cls.img[sub_roi_slices],
                                  cls.channel, coloc=cls.coloc)

    @classmethod
    def detect_sub_roi(cls, coord, offset, last_coord, denoise_max_shape, exclude_border, img5d,
                       sub_roi, channel, img_path=None, coloc=False):
        """One block given as an array -> ``(coord, table | None)`` with coordinates shifted
        to the full ROI (both the rel and the abs set, :164-170)."""
        exclude = cls._exclude_matrix(coord, last_coord, exclude_border)
        if denoise_max_shape is None and not coloc:
            segments = detector.detect_blobs(sub_roi, channel, exclude)
        else:
            # saturate + denoise tile by tile (:122-150), detect on the float64 result,
            # co-localise on the same image (:159-162)
            from . import blob_log as bl
            dvol = sub_roi if isinstance(sub_roi, bl.DeviceVolume) else bl.DeviceVolume(sub_roi)
            segments = detector.detect_blobs_blocks_device(
                dvol, channel, [(0, 0, 0)], [dvol.shape[:3]], denoise_max_shape=denoise_max_shape,
                exclude=lambda i: exclude, coloc=coloc)[0]
        if segments is not None:
            detector.Blobs.shift_blob_rel_coords(segments, offset)
            detector.Blobs.shift_blob_abs_coords(segments, offset)
        return coord, segments

    _extent_cache: dict = {}
    _coords_cache: dict = {}

    @classmethod
    def _grid_coords(cls, grid):
        """``list(np.ndindex(*grid))``, remembered per grid shape (a tuple of tuples: nobody writes to it)."""
        grid = tuple(int(v) for v in grid)
        hit = cls._coords_cache.get(grid)
        if hit is None:
            if len(cls._coords_cache) >= 16:
                cls._coords_cache.clear()
            hit = cls._coords_cache[grid] = tuple(np.ndindex(*grid))
        return hit

    @classmethod
    def _block_extents(cls, sub_roi_slices, shape3, mine):
        """``(origins, shapes)`` of the blocks ``mine`` (indices in C order of the grid): every slice resolved against the
        ROI with Python's rules.  Remembered per slice array and share (the entry keeps the array alive): a stack
        detected step after step pays the 256-block loop once, and the device pipeline recognises the SAME lists and
        reuses the block tables it uploaded for them."""
        key = (id(sub_roi_slices), tuple(int(v) for v in shape3), len(mine), mine[0] if mine else -1,
               mine[-1] if mine else -1)
        hit = cls._extent_cache.get(key)
        if hit is not None:
            return hit[0], hit[1]
        origins, shapes = [], []
        n0, n1, n2 = (int(v) for v in shape3)
        flat = sub_roi_slices.reshape(-1)            # (C order: the order of np.ndindex)
        for i in mine:
            z, y, x = flat[i]
            a0, a1, a2, b0, b1, b2 = z.start, y.start, x.start, z.stop, y.stop, x.stop
            if (a0 is None or a1 is None or a2 is None or b0 is None or b1 is None or b2 is None
                    or a0 < 0 or a1 < 0 or a2 < 0 or b0 < 0 or b1 < 0 or b2 < 0 or b0 > n0 or b1 > n1 or b2 > n2
                    or z.step not in (None, 1) or y.step not in (None, 1) or x.step not in (None, 1)):
                a0, b0, _ = z.indices(n0)             # (open-ended or negative bounds: Python's rules)
                a1, b1, _ = y.indices(n1)
                a2, b2, _ = x.indices(n2)
            origins.append((int(a0), int(a1), int(a2)))
            shapes.append((int(b0 - a0), int(b1 - a1), int(b2 - a2)))
        origins, shapes = tuple(origins), tuple(shapes)       # (immutable: the device pipeline may remember them by identity)
        if len(cls._extent_cache) >= 8:
            cls._extent_cache.clear()
        cls._extent_cache[key] = (origins, shapes, sub_roi_slices)
        return origins, shapes

    @classmethod
    def detect_blobs_sub_rois(cls, img5d, img, sub_roi_slices, sub_rois_offsets,
                              denoise_max_shape, exclude_border, coloc, channel):
        """All blocks -> object array (grid shaped) of per-block tables / ``None``.

        ``img`` is the ``(z, y, x[, c])`` ROI: a host array (uploaded once) or an
        already resident ``DeviceVolume``.
        """
        from . import blob_log as bl
        from . import dist
        cls.img5d, cls.img, cls.channel, cls.coloc = img5d, img, channel, coloc
        cls.denoise_max_shape, cls.exclude_border = denoise_max_shape, exclude_border
        grid = sub_roi_slices.shape
        last_coord = np.subtract(grid, 1)
        cls.last_coord = last_coord
        coords = cls._grid_coords(grid)
        mine = dist.my_share(len(coords))            # all of them without torch.distributed
        shape3 = img.shape[:3]
        origins, shapes = cls._block_extents(sub_roi_slices, shape3, mine)
        stats = bl.BatchStats()
        tables = []
        n_extra = (img.shape[3] if len(img.shape) > 3 else 0) if coloc else 0
        hint, cls.prune_hint = cls.prune_hint, None
        regular = hint is not None and StackPruner._geometry(
            shape3, hint[0], hint[1], hint[1] if hint[2] is None else hint[2], sub_roi_slices, sub_rois_offsets)[1]
        # several ranks: with the pruning planned (plan_pruning) and a regular block geometry every rank keeps its
        # own tables and the pruning itself is distributed; otherwise the tables are gathered on rank 0
        local_only = dist.world_size() > 1 and regular and DIST_PRUNE
        arena = _TableArena(11 + n_extra, len(mine)) if (dist.world_size() == 1 or local_only) else None
        if local_only:
            arena.headroom = 0.35       # (seam rows of the neighbouring ranks: ~10 % of a rank's rows per neighbour)
        pos = {i: k for k, i in enumerate(mine)}

        def exclude_of(k):
            return cls._exclude_matrix(coords[mine[k]], last_coord, exclude_border)

        pruner = None
        # (PRUNE_AHEAD "1" / "0" / "": always / never / for stacks of 64 blocks and more.  On the benchmark volume it
        #  moves ~4 ms of pruning under the GPU's last batches and adds most of that in the merge: 0.8-1.0 ms per volume
        #  in four alternating pairs of bench.py runs; it costs small stacks 0.6 ms: DESIGN.md.  With per-block
        #  preprocessing on: tail after the last kernel 11.4 -> 6.3 ms, tools/steptrace.py --denoise 25 -- once the
        #  host no longer waited for the tile tables' staging buffer, round 5; 218.9 against 216.6 ms before that)
        ahead = PRUNE_AHEAD
        make_pruner = None
        if regular and dist.world_size() == 1 and mine and (
                ahead == "1" or (ahead != "0" and len(mine) >= 64)):
            ov, tl, pad, prune_channels = hint

            def make_pruner():
                return _RegionPruner(arena, StackPruner._axis_plan(shape3, ov, tl, tl if pad is None else pad,
                                                                   sub_roi_slices, sub_rois_offsets),
                                     prune_channels, sub_roi_slices, shape3, mine)

        sink = None
        finisher = None

        def finish(k, tbl):
            # shift to ROI coordinates as soon as the block's batch is done (border exclusion and
            # co-localisation have happened on the block-relative table, in the reference's order)
            coord = coords[mine[k]]
            tbl = cls._finish_block(tbl, shapes[k], None, sub_rois_offsets[coord])
            if arena is not None:
                ahead_of_time = pruner if sink is None else sink.ensure_pruner()
                if tbl is not None and len(tbl):
                    arena.add(coord, tbl)
                arena.landed()
                if ahead_of_time is not None:
                    ahead_of_time.advance()
            return tbl

        own_dvol = None
        if mine:
            if isinstance(img, bl.DeviceVolume):
                dvol = img
            else:
                # a host image handed over for the length of this call: it goes up beside the detection of the blocks
                # that have landed, and whatever of it this rank's blocks never touched is cancelled before returning
                dvol = own_dvol = bl.DeviceVolume(img, streamed=True, cells=_upload_cells(sub_roi_slices, shape3))
            if arena is not None:
                # finished tables go straight from the native host path into the arena where the detection can hand
                # over peak arrays (one channel; several channels with co-localisation: the tables then land during the
                # LAST channel's pass, flags included); tables it has to build itself come through finish() -- both ways
                # the regions of the stack are pruned as their blocks land.  (Round 4 measured pruning ahead at +35-55 ms
                # per C5 volume with tables landing block by block -- the regions were pruned by Python then; with the
                # native region step and the final columns written by the merge it is 246.2 -> 240.5 ms, pruning + final
                # columns 15.9 -> 5.8 ms: profiles/r06_experiments.txt)
                flat_offsets = np.asarray(sub_rois_offsets, dtype=np.float64).reshape(-1, 3)    # (C order: coords' order)
                sink = _ArenaSink(arena, np.asarray(coords, dtype=np.int32)[mine[0]:mine[-1] + 1],
                                  flat_offsets[mine[0]:mine[-1] + 1],
                                  shapes, exclude_of if exclude_border is not None else None)
                # (the pruner's set-up -- 0.7 ms for 256 blocks -- waits until the first batch has landed: by then every
                #  batch is queued and the GPU busy)
                sink.pruner_factory = make_pruner
                # a small stack of one channel (all blocks in one batch: the GUI's ROI, a grid-search step): the whole
                # host chain behind its kernels as one native call
                if (regular and dist.world_size() == 1 and make_pruner is None and n_extra == 0 and STACK_FINISHER
                        and len(list(channel or [0])) == 1 and len(mine) <= bl.GRAPH_BLOCKS
                        and denoise_max_shape is None and list(hint[3]) == list(channel or [0])):
                    ov, tl, pad, _ = hint
                    plan_ = StackPruner._geometry(shape3, ov, tl, tl if pad is None else pad, sub_roi_slices,
                                                  sub_rois_offsets)[0]
                    if plan_ is not None:
                        finisher = _StackFinisher(sink, plan_, hint[3])
            try:
                tables = detector.detect_blobs_blocks_device(dvol, channel, origins, shapes, stats, finish,
                                                             denoise_max_shape=denoise_max_shape,
                                                             exclude=exclude_of, coloc=coloc, sink=sink,
                                                             stack_finisher=finisher)
            except BaseException:
                # the detection failed: the regions pruned ahead have nobody to collect them
                for p_ in (pruner, None if sink is None else sink.pruner):
                    if p_ is not None:
                        try:
                            p_.cancel()
                        except Exception:       # (the detection's own exception is the one to report)
                            pass
                raise
            finally:
                if own_dvol is not None:
                    own_dvol.close()
            if sink is not None and sink.pruner is not None:
                pruner = sink.pruner
            if finisher is not None and finisher.result is not None:
                pruner = finisher           # (its table is what prune_blobs_mp hands out, asked the planned way)
        cls.last_stats = stats
        local = [(i, tbl) for i, tbl in zip(mine, tables)]
        seg_rois = cls.assemble_seg_rois(local, grid, n_extra, arena, local_only=local_only)
        if pruner is not None and seg_rois.arena is arena:
            seg_rois.pruner = pruner
        return seg_rois

    @staticmethod
    def assemble_seg_rois(local, grid, n_extra: int = 0, arena=None, local_only: bool = False):
        """``(block index, table | None)`` pairs of this rank -> the grid-shaped object array of ALL blocks.
        With torch.distributed initialised the tables of every rank are gathered first; only rank 0 (the rank
        that prunes) unpacks them, the other ranks get ``None`` placeholders -- unless ``local_only``: then every
        rank keeps the tables of its own blocks (``None`` for the others) and ``StackPruner.prune_blobs_mp`` prunes
        them as a collective."""
        from . import dist
        coords = StackDetector._grid_coords(grid)
        seg_rois = np.zeros(grid, dtype=object).view(_SegRois)
        if dist.world_size() > 1 and local_only:
            for coord in coords:
                seg_rois[coord] = None
            for i, tbl in local:
                if arena is not None and tbl is not None and len(tbl):
                    tbl = arena.view(coords[i])
                seg_rois[coords[i]] = tbl
            seg_rois.arena, seg_rois.local_only = arena, True
            return seg_rois
        if dist.world_size() > 1:
            # several ranks: the pruning rank receives all rows as one array in block order and lays them out as
            # its arena with whole-array copies (merge_blobs and the native prune step then take their fast path
            # as on one GPU); the other ranks keep None placeholders
            got = dist.gather_tables(local, len(coords), decode_on=0, raw=True)
            if got is not None:
                idx, rows, empties = got
                arena = _TableArena.from_rows(idx, rows, np.asarray(coords, dtype=np.int64))
                for coord in arena.spans:
                    seg_rois[coord] = arena.view(coord)
                for i in empties:
                    seg_rois[coords[i]] = np.zeros((0, rows.shape[1]))
                for coord in np.ndindex(*grid):
                    if isinstance(seg_rois[coord], (int, np.integer)):
                        seg_rois[coord] = None
                seg_rois.arena = arena
            else:
                for coord in np.ndindex(*grid):
                    seg_rois[coord] = None
            return seg_rois
        for i, tbl in sorted(local, key=lambda e: e[0]):
            if arena is not None and tbl is not None and len(tbl):
                tbl = arena.view(coords[i])         # the copy that lives in the arena
            seg_rois[coords[i]] = tbl
        if arena is not None:
            seg_rois.arena = arena
        return seg_rois


class Blocks(NamedTuple):
    """Block processing parameters (same 9 fields as the reference, :260-279)."""
    sub_roi_slices: np.ndarray
    sub_rois_offsets: np.ndarray
    denoise_max_shape: Optional[np.ndarray]
    exclude_border: Optional[Sequence[int]]
    tol: np.ndarray
    overlap_base: np.ndarray
    overlap: np.ndarray
    overlap_padding: np.ndarray
    max_pixels: np.ndarray


def setup_blocks(settings, shape: Sequence[int]) -> Blocks:
    """Block grid and pruning distances from a profile and ``config.resolutions``."""
    scale = detector.calc_scaling_factor()
    denoise_size = settings["denoise_size"]
    denoise_max_shape = (np.ceil(np.multiply(scale, denoise_size)).astype(int)
                         if denoise_size else None)
    overlap_base = detector.calc_overlap()
    tol = np.multiply(overlap_base, settings["prune_tol_factor"]).astype(int)
    overlap_padding = tol.copy()
    overlap = overlap_base.copy()
    exclude_border = settings["exclude_border"]
    if exclude_border is not None:
        # overlap must exceed twice the excluded border so no plane is excluded from both
        # neighbours; one more plane where a border is excluded, and no padding there
        twice = np.multiply(2, exclude_border)
        overlap = np.where(overlap < twice, twice, overlap)
        has_border = np.greater(exclude_border, 0)
        overlap[has_border] += 1
        overlap_padding[has_border] = 0
    max_pixels = np.ceil(np.multiply(scale, settings["segment_size"])).astype(int)
    slices, offsets = chunking.stack_splitter(shape, max_pixels, overlap)
    return Blocks(slices, offsets, denoise_max_shape, exclude_border, tol, overlap_base, overlap,
                  overlap_padding, max_pixels)


def _combine_paths(base: Optional[str], suffix: str) -> str:
    """``libmag.combine_paths`` for the default arguments (reference libmag.py:331-380)."""
    if not base:
        return suffix
    if not os.path.basename(base):
        return os.path.join(base, suffix)
    return os.path.splitext(base)[0] + "_" + suffix


def _subimage_name(base: str, offset, shape) -> str:
    """``naming.make_subimage_name`` (reference naming.py:9-38): x,y,z order in the name."""
    site = "{}x{}".format(tuple(offset[::-1]), tuple(shape[::-1])).replace(" ", "")
    stem, ext = os.path.splitext(base)
    return f"{stem}_{site}{ext}"


def _prepare_subimg(image5d, offset, size):
    """``plot_3d.prepare_subimg`` (reference plot_3d.py:340-375): ``[t=0, z, y, x]`` slab."""
    sl = tuple(slice(int(o), int(o) + int(s)) for o, s in zip(offset, size))
    return image5d[0][sl]


class _StackRun:
    """One whole-image detection from ROI to ``Blobs`` (what ``detect_blobs_blocks`` does, reference :338-517, as four
    steps over shared state): :meth:`resolve_roi` (which voxels, which channels, where results are written),
    :meth:`detect` (blocks -> per-block tables), :meth:`prune` (the overlap de-duplication, on one rank or as a
    collective) and :meth:`finish` (final columns, metadata, the side files the reference writes)."""

    def __init__(self, filename_base, img5d, save_dfs: bool):
        self.t0 = time()
        self.base = filename_base
        self.volume = img5d.img
        self.img5d = img5d
        self.save_dfs = save_dfs
        self.seconds = {}

    def resolve_roi(self, offset, size, channels, full_roi: bool, coloc: bool):
        """ROI voxels (the whole first time point or a sub-image), output paths, channels, co-localisation only
        with two channels and more (:374-397)."""
        vol = self.volume
        self.path_base = self.base
        whole = full_roi
        if offset is None or size is None:
            offset, size = (0, 0, 0), vol.shape[1:4]
        else:
            self.path_base = _subimage_name(self.base, offset, size)
        self.offset, self.size = offset, size
        pre = getattr(self.img5d, "device_volume", None)
        if pre is not None and whole and tuple(pre.shape[:3]) == tuple(vol.shape[1:4]):
            self.roi = pre                      # (already uploading: Image5d.prefetch)
        else:
            self.roi = vol[0] if full_roi else _prepare_subimg(vol, offset, size)
        self.n_roi_channels = self.roi.shape[3] if len(self.roi.shape) > 3 else 1
        self.coloc = bool(coloc) and self.n_roi_channels > 1
        self.channels = (detector._channels_of(len(self.roi.shape), self.n_roi_channels, None)[1]
                         if channels is None else channels)
        return self

    def _timed(self, key, fn):
        start = time()
        out = fn()
        self.seconds[key] = time() - start
        return out

    def detect(self):
        """Block geometry from the FIRST channel's profile (:399-404), then every block of this rank's share; the
        pruning parameters are announced first so that finished regions can be pruned while the GPU is busy."""
        def run():
            self.blocks = bk = setup_blocks(config.get_roi_profile(self.channels[0]), self.roi.shape)
            StackDetector.plan_pruning(bk.overlap, bk.tol, bk.overlap_padding, self.channels)
            return StackDetector.detect_blobs_sub_rois(
                self.img5d, self.roi, bk.sub_roi_slices, bk.sub_rois_offsets, bk.denoise_max_shape,
                bk.exclude_border, self.coloc, self.channels)
        self.seg_rois = self._timed(StackTimes.DETECTION, run)
        return self

    def _prune_here(self):
        bk = self.blocks
        from . import dist
        # (no co-localisation columns, and either one process or every rank pruning its own rows: the table may come
        #  back in its final columns, see finish(); a table pruned on rank 0 and broadcast keeps the merged columns)
        own = dist.world_size() == 1 or getattr(self.seg_rois, "local_only", False)
        return StackPruner.prune_blobs_mp(self.roi, self.seg_rois, bk.overlap, bk.tol, bk.sub_roi_slices,
                                          bk.sub_rois_offsets, self.channels, bk.overlap_padding,
                                          final_form=own, untouched=True,
                                          n_flag_cols=self.n_roi_channels if self.coloc else 0)

    def prune(self):
        """The merged, pruned table on every rank.  One rank: a plain call.  Several ranks: either the tables stayed
        on their ranks and the pruning is a collective, or they were gathered on rank 0, which prunes and broadcasts
        (telling the others first if it failed: they are about to wait for the table)."""
        from . import dist

        def run():
            if getattr(self.seg_rois, "local_only", False):
                return self._prune_here()
            table, frame, failure = None, None, None
            if dist.rank() == 0:
                try:
                    table, frame = self._prune_here()
                except Exception as exc:
                    failure = exc
            dist.raise_together(failure, "pruning on rank 0")
            return dist.broadcast_table(table), frame
        self.table, self.ratios = self._timed(StackTimes.PRUNING, run)
        return self

    def finish(self):
        """``Blobs`` in the reference's final form (:455-498): rel <- abs, co-localisation flags read from column 10
        on (the reference's own off-by-one, kept: DESIGN.md section 2c), abs columns dropped; metadata; the CSVs and
        the optional sub-image file, on rank 0 only."""
        from . import dist
        root = dist.rank() == 0
        if root and self.save_dfs and self.ratios is not None and len(self.ratios):
            _save_pruning_ratios(self.ratios)
        final, flags = self.table, None
        path = _combine_paths(self.path_base, config.SUFFIX_BLOBS)
        if isinstance(final, _FinalTable):
            # the pruning step wrote the final columns itself; the column registry ends as the two steps below leave it
            blobs = detector.Blobs(None, path=path)
            blobs.cols = list(final.col_names)
            if final.coloc_cols is not None:        # (`segments_all[:, 10:10 + C].astype(np.uint8)`, :463-464)
                flags = final.coloc_cols.astype(np.uint8)
            final = final.view(np.ndarray)
        else:
            blobs = detector.Blobs(final, path=path)
            if final is not None:
                blobs.replace_rel_with_abs_blob_coords(final)
                blobs.blobs = final
                if self.coloc:
                    flags = final[:, 10:10 + self.n_roi_channels].astype(np.uint8)
                final = blobs.remove_abs_blob_coords(True)
        if config.save_subimg and root:
            roi = self.roi if isinstance(self.roi, np.ndarray) else self.volume[0]     # (a prefetched device volume)
            _save_subimage(_combine_paths(self.path_base, config.SUFFIX_SUBIMG), self.volume, roi)
        blobs.blobs, blobs.colocalizations = final, flags
        blobs.resolutions = config.resolutions
        blobs.basename = os.path.basename(config.filename) if config.filename else None
        blobs.roi_offset, blobs.roi_size = self.offset, self.size
        blobs.times = {StackTimes.DETECTION: [self.seconds[StackTimes.DETECTION]],
                       StackTimes.PRUNING: [self.seconds[StackTimes.PRUNING]], StackTimes.TOTAL: time() - self.t0}
        if self.save_dfs and root:
            import pandas as pd
            pd.DataFrame({k.value: v for k, v in blobs.times.items()}).to_csv("stack_detection_times.csv", index=False)
        _logger.info("No blobs detected" if final is None else f"Total blobs found: {len(final)}")
        return blobs


def _save_subimage(path: str, volume, roi) -> None:
    """``config.save_subimg``: the ROI as a ``.npy`` file next to the blobs archive (reference :477-489); skipped with
    a warning when the image itself is a memory map of that very file (saving would truncate what is being read)."""
    if isinstance(volume, np.memmap) and getattr(volume, "filename", None) == os.path.abspath(path):
        _logger.warning("%s is currently open, cannot save sub-image", path)
        return
    if not isinstance(roi, np.ndarray):
        # an image already resident in HBM (a DeviceVolume / a tensor): its voxels as they are there -- the
        # caller's own for the voxel types the kernels read in place (uint8 / uint16 / float32 / float64)
        tensor = getattr(roi, "tensor", roi)
        if not hasattr(tensor, "cpu"):
            raise TypeError(f"config.save_subimg: cannot save a ROI of type {type(roi).__name__}")
        roi = tensor.cpu().numpy()
    with open(path, "wb") as f:
        np.save(f, roi)


def detect_blobs_blocks(filename_base: str, img5d, offset=None, size=None, channels=None,
                        verify: bool = False, save_dfs: bool = True, full_roi: bool = False,
                        coloc: bool = False):
    """Detect blobs in a large image block by block -> ``(stats, fdbk, Blobs)`` (reference :338-517).

    Several ranks (``torch.distributed``): every rank detects its share of the blocks and all return the same
    table; only rank 0 writes ``blob_ratios*.csv``, ``stack_detection_times.csv`` and the sub-image."""
    if img5d.img is None:
        raise ValueError("Image data is None")
    if verify:
        raise NotImplementedError("truth-set verification is outside this path's scope")
    run = _StackRun(filename_base, img5d, save_dfs)
    blobs = run.resolve_roi(offset, size, channels, full_roi, coloc).detect().prune().finish()
    return None, None, blobs


def detect_blobs_tiles(filename_bases, tiles, channels=None, coloc: bool = False, save_dfs: bool = False,
                       shard: Optional[str] = None):
    """Whole-image detection of consecutive tiles of a tiled stack (BASELINE.json configs[4]: a light-sheet stack as
    tiles), one ``detect_blobs_blocks`` each: the upload of tile k + 1 is queued before tile k is detected and runs
    beside it, the device buffers of the batched passes and of the per-block preprocessing are reused from tile to
    tile.  ``tiles``: an iterable of ``Image5d`` (host images: memory-mapped ``image5d.npy`` files, arrays, pinned
    tensors); yields ``(index, Blobs)`` in order.  Each tile is an image of its own, exactly as the reference treats a
    file (stack_detect.py:338-517); placing the tables in a common frame is the caller's (the importer's) business.

    Several ranks (``torch.distributed``), ``shard``:

    * ``None`` / ``"blocks"``: every rank walks ALL tiles and the blocks of each tile are cut over the ranks, as
      ``detect_blobs_blocks`` does for one image (two small exchanges per tile, every rank gets every table);
    * ``"tiles"``: rank r detects tiles r, r + N, ... (``dist.tile_share``) as one process would -- no exchange at all,
      the pruning local, each rank uploading over its own PCIe link -- and yields ``(index, Blobs)`` for ITS tiles only
      (``index``: the tile's place in ``tiles``).  Tiles of other ranks are never touched (an iterator is advanced past
      them).  :func:`gather_tiles` afterwards puts every rank's tables on every rank, for a caller that wants them."""
    import itertools
    from . import dist
    if shard not in (None, "blocks", "tiles"):
        raise ValueError(f"shard must be None, 'blocks' or 'tiles', not {shard!r}")
    by_tile = shard == "tiles" and dist.world_size() > 1
    step, first = (dist.world_size(), dist.rank()) if by_tile else (1, 0)
    it = itertools.islice(enumerate(tiles), first, None, step)
    named = None if isinstance(filename_bases, str) else iter(filename_bases)
    names_at = [0]

    def base_of(k):
        if named is None:
            return f"{filename_bases}_{k}"
        name = None
        while names_at[0] <= k:                     # (an iterator of names is advanced past the other ranks' tiles too)
            name = next(named)
            names_at[0] += 1
        return name

    def detect(base, tile):
        if by_tile:
            with dist.solo():                       # this tile is this rank's alone: all its blocks, no collective
                return detect_blobs_blocks(base, tile, None, None, channels, False, save_dfs, True, coloc)
        return detect_blobs_blocks(base, tile, None, None, channels, False, save_dfs, True, coloc)

    cur = nxt = None
    try:
        k, cur = next(it, (0, None))
        if cur is not None:
            cur.prefetch()
        while cur is not None:
            k_nxt, nxt = next(it, (0, None))
            if nxt is not None:
                nxt.prefetch()                      # (its copies are queued on its own stream before tile k's kernels)
            _, _, blobs = detect(base_of(k), cur)
            cur.release()                           # (the tile's voxels leave the device with it)
            yield k, blobs
            cur, nxt, k = nxt, None, k_nxt
    finally:
        # a failed detection, or a consumer that stops early (GeneratorExit): the uploads still in flight are cancelled
        # and joined before their device blocks go back to the allocator
        for tile in (cur, nxt):
            if tile is not None:
                tile.release()


def gather_tiles(results, failure: Optional[BaseException] = None):
    """After ``detect_blobs_tiles(..., shard="tiles")``: every rank's ``(index, Blobs)`` pairs on every rank, sorted by
    tile index (collective; a rank without tiles passes ``[]``).  What travels is each tile's final table and its
    co-localisation flags (``dist.gather_tile_tables``: two small all-gathers, RCCL on GPUs); the ``Blobs`` made for
    another rank's tile carry those plus this process' resolutions -- paths and ROI metadata stay with the rank that
    detected the tile.  Without a process group: ``results``, sorted.  ``failure``: what this rank's detection raised,
    if anything (every rank then raises)."""
    from . import dist
    results = sorted(results, key=lambda e: e[0])
    if dist.world_size() == 1:
        if failure is not None:
            raise failure
        return results
    own = {int(k): b for k, b in results}
    local = []
    for k, b in results:
        tbl, n_cols = b.blobs, 0
        if tbl is not None:
            n_cols = tbl.shape[1]
            if b.colocalizations is not None:
                tbl = np.hstack((tbl, np.asarray(b.colocalizations, dtype=np.float64)))
        local.append((int(k), tbl, n_cols))
    out = []
    # the columns of a final table (stack_detect.py:455-470: rel <- abs, abs dropped): this rank's own tiles say, a rank
    # without tiles takes the registry's standard order
    col_names = next((list(b.cols) for b in own.values() if b.blobs is not None and b.cols), None) or [
        c.value for c in detector.Blobs.Cols if not c.name.startswith("ABS_")]
    for k, tbl, n_cols in dist.gather_tile_tables(local, failure):
        if k in own:
            out.append((k, own[k]))
            continue
        blobs = detector.Blobs(None)
        if tbl is not None:
            blobs.cols = col_names[:n_cols] if n_cols <= len(col_names) else None
            blobs.blobs = np.ascontiguousarray(tbl[:, :n_cols])
            if tbl.shape[1] > n_cols:
                blobs.colocalizations = tbl[:, n_cols:].astype(np.uint8)
        blobs.resolutions = config.resolutions
        out.append((k, blobs))
    return out


def _save_pruning_ratios(df):
    """``blob_ratios.csv`` and the blob-count weighted means (reference :424-442)."""
    df.to_csv("blob_ratios.csv", index=False)
    cols = df.columns.tolist()
    if "blobs" in cols:
        weights = df["blobs"]
        total = np.sum(weights)
        means = {f"mean_{c}": [np.sum(np.multiply(df[c], weights)) / total] for c in cols[1:]}
        import pandas as pd
        pd.DataFrame(means).to_csv("blob_ratios_means.csv", index=False)


def _combine_arrs(arrs):
    arrs = [a for a in arrs if a is not None]
    if not arrs:
        return None
    return arrs[0] if len(arrs) == 1 else np.concatenate(arrs)


def detect_blobs_stack(filename_base: str, img5d, subimg_offset=None, subimg_size=None,
                       coloc: bool = False):
    """Detect blobs in a whole image; channels whose profiles agree on
    ``ROIProfile.BLOCK_SIZES`` share one set of blocks, others get their own (:554-561).
    Saves ``<base>_blobs.npz``."""
    if img5d is None or img5d.img is None:
        raise IOError("No image data available for blob detection")
    n_chl = img5d.img.shape[4] if img5d.img.ndim > 4 else 1
    channels = detector._channels_of(img5d.img.ndim, n_chl, config.channel, 4)[1]
    channels = list(channels)
    if roi_prof.ROIProfile.is_identical_settings(
            [config.get_roi_profile(c) for c in channels], roi_prof.ROIProfile.BLOCK_SIZES):
        channels = [channels]
    outs = []
    for chl in channels:
        chl = list(chl) if isinstance(chl, (list, tuple, range)) else [chl]
        outs.append(detect_blobs_blocks(
            filename_base, img5d, subimg_offset, subimg_size, chl, False,
            not config.grid_search_profile, getattr(img5d, "is_roi", False), coloc))
    blobs_all = None
    if outs:
        blobs_all = outs[0][2]
        blobs_all.blobs = _combine_arrs([o[2].blobs for o in outs])
        blobs_all.colocalizations = _combine_arrs([o[2].colocalizations for o in outs])
        if blobs_all.blobs is not None:
            detector.Blobs.show_blobs_per_channel(blobs_all.blobs)
        from . import dist
        if dist.rank() == 0:
            blobs_all.save_archive()
    return None, "", blobs_all


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
        coords = StackDetector._grid_coords(grid)
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
