"""Where the per-block blob tables of a stack live while it is detected: one arena per stack (``_TableArena``), the sink
the detection writes them through (``_ArenaSink``) and, for small one-batch stacks, the whole host chain behind the kernels
as one native call (``_StackFinisher``).  Split out of ``stack_detect.py`` (round 6); ``stack_detect`` re-exports the names."""
from __future__ import annotations

import ctypes
import os
import sys
from time import time
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _native as nat
from . import config, detector
from .stack_prune import StackPruner, _RegionPruner, grid_coords

_logger = config.logger.getChild(__name__)


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
        for coord, tbl in zip(grid_coords(blob_rois.shape), blob_rois.ravel().tolist()):
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


class _ArenaSinkPart(_ArenaSink):
    """Blocks ``[k_lo, k_hi)`` of a sink's share as a sink of their own (block indices start at 0 again): the same arena,
    and ONE pruner whoever makes it first -- a chunk's native tables, or the parent through ``finish()`` for tables built
    in Python (what a stack detected z-chunk by z-chunk hands each chunk's detection, ``stack_detect._detect_chunks``)."""

    def __init__(self, parent: _ArenaSink, k_lo: int, k_hi: int):
        self.parent = parent
        self.arena = parent.arena
        self.grid_coords = parent.grid_coords[k_lo:k_hi]
        self.block_offsets = parent.block_offsets[k_lo:k_hi]
        self.shapes = parent.shapes[k_lo:k_hi]
        self.exclude_of = None if parent.exclude_of is None else (lambda j: parent.exclude_of(k_lo + j))

    pruner = property(lambda self: self.parent.pruner, lambda self, v: setattr(self.parent, "pruner", v))
    pruner_factory = property(lambda self: self.parent.pruner_factory,
                              lambda self, v: setattr(self.parent, "pruner_factory", v))


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
