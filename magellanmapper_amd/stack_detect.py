"""Whole-volume blob detection in blocks (mirror of ``magmap.cv.stack_detect``).

Same public surface as the reference (magmap/cv/stack_detect.py): :class:`StackTimes`
(:27-31), :class:`StackDetector` (:34-257), :class:`Blocks` / :func:`setup_blocks`
(:260-335), :func:`detect_blobs_blocks` (:338-517), :func:`detect_blobs_stack` (:520-615),
:class:`StackPruner` (:618-861; the code lives in :mod:`magellanmapper_amd.stack_prune`, the tables' arena and sink in
:mod:`magellanmapper_amd.stack_tables`, both re-exported here).

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
from . import stack_prune
# (the reference keeps the pruner in this module, magmap/cv/stack_detect.py:618-861: same import path here)
from .stack_prune import StackPruner, _FinalTable, _RegionPruner, _region_reach, _region_workers, _rows_within, grid_coords
from .stack_tables import _ArenaSink, _ArenaSinkPart, _StackFinisher, _TableArena

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
#: (the phases of the pruning step on stderr: ``stack_prune.PRUNE_PROF``)
#: regions of the pruning ahead per x-row of blocks: 0 = one per channel of the table (a two-channel stack has twice the
#: rows per block: its last regions, pruned after the last kernel, are halved -- bench.py --region-split has the A/B)
REGION_SPLIT = 0
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

    def prefetch(self, own_planes: bool = False):
        """Start the upload of the first time point now (``blob_log.DeviceVolume``: block row by block row on a copy
        stream); a later whole-image ``detect_blobs_blocks`` / ``detect_blobs_stack`` of this image detects on it while
        the rest is still in flight.  What `detect_blobs_tiles` calls for tile k + 1 before it detects tile k.  The image
        must stay as it is until the detection has returned or :meth:`release` has been called (the upload reads it in
        the background).  ``own_planes``: with a process group, only the planes the blocks of THIS rank's share touch
        (the block split of a whole-image detection; not for a tile a rank detects alone)."""
        from . import blob_log as bl
        from . import dist
        if self.device_volume is None and self.img is not None:
            if _image_bytes(self.img[0]) > _resident_limit():
                return self             # (too large to be resident: detected z-chunk by z-chunk from the host)
            shape3 = tuple(int(v) for v in self.img.shape[1:4])
            cells = blocks = None
            try:        # (where the first channel's profile puts the block rows: only the upload ORDER depends on it)
                blocks = setup_blocks(config.get_roi_profile(0), shape3)
                cells = _upload_cells(blocks.sub_roi_slices, shape3)
            except Exception:
                pass
            if own_planes and blocks is not None and dist.world_size() > 1:
                grid = blocks.sub_roi_slices.shape
                coords = grid_coords(grid)
                mine = dist.my_share(len(coords))
                if not mine:
                    return self
                ext = [blocks.sub_roi_slices[coords[i]][0].indices(shape3[0])[:2] for i in mine]
                z_lo, z_hi = min(e[0] for e in ext), max(e[1] for e in ext)
                cells = ([z - z_lo for z in cells[0] if z_lo < z < z_hi] + [z_hi - z_lo], cells[1])
                self.device_volume = bl.DeviceVolume(self.img[0][z_lo:z_hi], streamed=True, cells=cells, z_off=z_lo,
                                                     full_shape=shape3)
            else:
                self.device_volume = bl.DeviceVolume(self.img[0], streamed=True, cells=cells)
        return self

    def release(self):
        """Drop the prefetched device copy, cancelling whatever of its upload has not been queued yet."""
        dv, self.device_volume = self.device_volume, None
        if dv is not None:
            dv.close()


#: a HOST image larger than this many bytes is detected z-chunk by z-chunk -- whole layers of blocks, each chunk a device
#: volume of its own with the next one on its way up meanwhile -- instead of going to the device whole (the reference
#: reads any size through its memory map, importer.py:794); the tables land in the one arena and are pruned once, as
#: always.  ``None``: a third of the device memory that is free when the call starts.
MAX_RESIDENT_BYTES = None


def _resident_limit() -> int:
    if MAX_RESIDENT_BYTES is not None:
        return int(MAX_RESIDENT_BYTES)
    import torch
    return int(torch.cuda.mem_get_info()[0] // 3)


def _image_bytes(img) -> int:
    n = 1
    for v in img.shape:
        n *= int(v)
    item = img.element_size() if hasattr(img, "element_size") else np.dtype(img.dtype).itemsize
    return n * int(item)


def _z_chunks(coords, mine, origins, shapes, plane_bytes: int, limit: int):
    """This rank's blocks (z-major, ``mine[k]`` -> ``coords``) cut into runs of whole block LAYERS whose planes take at
    most ``limit / 2`` bytes (two chunks are on the device while one is detected and the next goes up; a single layer
    may exceed it): ``[(k_lo, k_hi, z_lo, z_hi)]``."""
    layers = []                       # (k_lo, k_hi, z_lo, z_hi) per layer of the block grid
    for k, i in enumerate(mine):
        z0, z1 = int(origins[k][0]), int(origins[k][0]) + int(shapes[k][0])
        if layers and coords[i][0] == layers[-1][4]:
            a = layers[-1]
            layers[-1] = (a[0], k + 1, min(a[2], z0), max(a[3], z1), a[4])
        else:
            layers.append((k, k + 1, z0, z1, coords[i][0]))
    chunks = []
    for k_lo, k_hi, z_lo, z_hi, _ in layers:
        if chunks and (max(chunks[-1][3], z_hi) - chunks[-1][2]) * plane_bytes <= limit // 2:
            c = chunks[-1]
            chunks[-1] = (c[0], k_hi, c[2], max(c[3], z_hi))
        else:
            chunks.append((k_lo, k_hi, z_lo, z_hi))
    return chunks


def _upload_cells(sub_roi_slices, shape3):
    """``(z ends, y ends)`` of the block grid's layers and rows: where a host image on its way to the device is cut so
    that a block can start once the cells it touches have landed (``volume._SlabUpload``)."""
    gz, gy = sub_roi_slices.shape[:2]
    z_ends = [int(sub_roi_slices[(l, 0, 0)][0].indices(int(shape3[0]))[1]) for l in range(gz)]
    y_ends = [int(sub_roi_slices[(0, j, 0)][1].indices(int(shape3[1]))[1]) for j in range(gy)]
    return z_ends, y_ends


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
    _grid_coords = staticmethod(grid_coords)

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
                                     prune_channels, sub_roi_slices, shape3, mine,
                                     min_regions=-(-len(mine) // max(1, int(grid[2]))) * int(
                                         REGION_SPLIT or max(1, len(prune_channels))))

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
            chunks = None
            if isinstance(img, bl.DeviceVolume):
                dvol = img
                held = (int(dvol.z_off), int(dvol.z_off) + int(dvol.tensor.shape[0]))
                if held != (0, int(shape3[0])):
                    # a volume that holds some planes only (Image5d.prefetch(own_planes=True), a rank's slab): they
                    # must be the ones this rank's blocks touch
                    z_lo = min(int(o[0]) for o in origins)
                    z_hi = max(int(o[0]) + int(s_[0]) for o, s_ in zip(origins, shapes))
                    if z_lo < held[0] or z_hi > held[1]:
                        raise nat.MmxError(f"the device volume holds planes [{held[0]}, {held[1]}) but this rank's blocks "
                                           f"touch [{z_lo}, {z_hi})")
            else:
                # the planes this rank's blocks touch (all of them without torch.distributed)
                z_lo = min(int(o[0]) for o in origins)
                z_hi = max(int(o[0]) + int(s_[0]) for o, s_ in zip(origins, shapes))
                plane = _image_bytes(img) // max(1, int(img.shape[0]))
                dvol = None
                on_host = getattr(getattr(img, "device", None), "type", "cpu") == "cpu"     # (not a tensor on a device)
                if not on_host:
                    pass
                elif (z_hi - z_lo) * plane > _resident_limit() and len({coords[i][0] for i in mine}) > 1:
                    # too large to be resident: whole layers of blocks at a time, each from a device volume of its own
                    chunks = _z_chunks(coords, mine, origins, shapes, plane, _resident_limit())
                elif z_lo > 0 or z_hi < int(shape3[0]):
                    # a rank's share of a host image: only its planes go up (over this rank's own link)
                    chunks = [(0, len(mine), z_lo, z_hi)]
            if chunks is None and dvol is None:
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
                        and chunks is None and len(list(channel or [0])) == 1 and len(mine) <= bl.GRAPH_BLOCKS
                        and denoise_max_shape is None and list(hint[3]) == list(channel or [0])):
                    ov, tl, pad, _ = hint
                    plan_ = StackPruner._geometry(shape3, ov, tl, tl if pad is None else pad, sub_roi_slices,
                                                  sub_rois_offsets)[0]
                    if plan_ is not None:
                        finisher = _StackFinisher(sink, plan_, hint[3])
            try:
                if chunks is None:
                    tables = detector.detect_blobs_blocks_device(dvol, channel, origins, shapes, stats, finish,
                                                                 denoise_max_shape=denoise_max_shape,
                                                                 exclude=exclude_of, coloc=coloc, sink=sink,
                                                                 stack_finisher=finisher)
                else:
                    tables = cls._detect_chunks(img, chunks, sub_roi_slices, shape3, channel, origins, shapes, stats,
                                                finish, denoise_max_shape, exclude_of, coloc, sink)
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

    @classmethod
    def _detect_chunks(cls, img, chunks, sub_roi_slices, shape3, channel, origins, shapes, stats, finish,
                       denoise_max_shape, exclude_of, coloc, sink):
        """The blocks of this rank's share chunk by chunk (``_z_chunks``): planes ``[z_lo, z_hi)`` of the host image as
        a device volume that answers for the whole image (``DeviceVolume(z_off=...)``), the next chunk's upload started
        before this one is detected, the tables through per-chunk sinks into the ONE arena (same pruner, same order of
        landing as the resident path).  Returns the tables of all blocks, in order."""
        from . import blob_log as bl
        z_ends_all, y_ends = _upload_cells(sub_roi_slices, shape3)

        def volume(c):
            k_lo, k_hi, z_lo, z_hi = c
            cells = ([z - z_lo for z in z_ends_all if z_lo < z < z_hi] + [z_hi - z_lo], y_ends)
            return bl.DeviceVolume(img[z_lo:z_hi], streamed=True, cells=cells, z_off=z_lo, full_shape=shape3)

        tables = []
        nxt = volume(chunks[0])
        cur = None
        try:
            for ci, (k_lo, k_hi, z_lo, z_hi) in enumerate(chunks):
                cur, nxt = nxt, None
                if ci + 1 < len(chunks):
                    nxt = volume(chunks[ci + 1])     # (its staging starts once this chunk has queued its last region)
                part = None if sink is None else _ArenaSinkPart(sink, k_lo, k_hi)
                tables.extend(detector.detect_blobs_blocks_device(
                    cur, channel, origins[k_lo:k_hi], shapes[k_lo:k_hi], stats,
                    lambda j, tbl, k0=k_lo: finish(k0 + j, tbl), denoise_max_shape=denoise_max_shape,
                    exclude=None if exclude_of is None else (lambda j, k0=k_lo: exclude_of(k0 + j)),
                    coloc=coloc, sink=part, stack_finisher=None))
                cur.close()
                cur = None
        finally:
            for v in (cur, nxt):
                if v is not None:
                    v.close()
        return tables

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
        own = not by_tile and dist.world_size() > 1      # (every tile's BLOCKS over all ranks: a rank's planes only)
        k, cur = next(it, (0, None))
        prefetch = (lambda t: t.prefetch(own_planes=True)) if own else (lambda t: t.prefetch())
        if cur is not None:
            prefetch(cur)
        while cur is not None:
            k_nxt, nxt = next(it, (0, None))
            if nxt is not None:
                prefetch(nxt)                       # (its copies are queued on its own stream before tile k's kernels)
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
