"""Per-ROI blob detection and the blob table (mirror of ``magmap.cv.detector``).

Public surface kept from the reference (magmap/cv/detector.py):

* :class:`Blobs` -- table + archive class: column registry shared at class level
  (:116, :154-162), ``format_blobs`` to the 11 standard columns (:325-364), coordinate
  shifting helpers (:603-709), ``blobs_in_channel`` (:747-772), npz archive version 5
  (:68-86, :185-323).
* :func:`calc_scaling_factor`, :func:`calc_overlap` (:810-841), ``OVERLAP_FACTOR`` (:41).
* :func:`detect_blobs` (:874-957) -- the scikit-image ``blob_log`` call (:931-933) is
  replaced by the device pipeline of :mod:`magellanmapper_amd.blob_log`.
* :func:`remove_close_blobs` (:1000-1085) -- the all-pairs search runs on the device
  (``mmx_close_pairs``); deletion, averaging (round-half-even) and last-write-wins are
  applied on the host exactly as NumPy does in the reference.
* :func:`get_blobs_in_roi`, :func:`get_blobs_interior` (:1210-1268),
  :func:`meas_pruning_ratio` (:1126-1147), :func:`sort_blobs` (:985-997).
"""
from __future__ import annotations

import ctypes
import math
import os
from enum import Enum
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import config

#: blob confirmation flags
CONFIRMATION: Dict[int, str] = {-1: "unverified", 0: "no", 1: "yes", 2: "maybe"}
#: overlap between neighbouring blocks, in multiples of the pixel scaling
OVERLAP_FACTOR: int = 5

_logger = config.logger.getChild(__name__)


def _map_columns(table: np.ndarray, src_cols, out: Optional[np.ndarray] = None, dst0: int = 0) -> np.ndarray:
    """``out[:, dst0:dst0 + len(src_cols)] = table[:, src_cols]`` (``out`` = a new table by default).

    Large float64 tables (the whole-stack table: 10^5..10^6 rows) go through the threaded native copy
    (``mmx_host_map_columns``); anything else through NumPy.
    """
    src = [int(c) for c in src_cols]

    def plain(a):
        return (isinstance(a, np.ndarray) and a.ndim == 2 and a.dtype == np.float64 and a.strides[1] == 8
                and a.strides[0] % 8 == 0 and a.strides[0] >= 8 * a.shape[1])
    if out is None:
        out = np.empty((len(table), len(src)), dtype=table.dtype)
        dst0 = 0
    if len(table) >= 4096 and plain(table) and plain(out) and 1 <= len(src) <= 64:
        from . import _native as nat
        lib = nat.lib()
        cols = (ctypes.c_int32 * len(src))(*src)
        nat.check(lib.mmx_host_map_columns(table.ctypes.data, table.strides[0] // 8, len(table), cols, len(src),
                                           out.ctypes.data, out.strides[0] // 8, int(dst0)), "mmx_host_map_columns")
    else:
        out[:, dst0:dst0 + len(src)] = table[:, src]
    return out


class Blobs:
    """Blob table ``[[z, y, x, radius, ...], ...]`` with named columns and an npz archive."""

    #: archive version (5: column names no longer list the removed abs coords)
    BLOBS_NP_VER: int = 5

    class Keys(Enum):
        VER = "ver"
        BLOBS = "segments"
        COLOCS = "colocs"
        RESOLUTIONS = "resolutions"
        BASENAME = "basename"
        ROI_OFFSET = "offset"
        ROI_SIZE = "roi_size"
        COLS = "columns"

    class Cols(Enum):
        Z = "z"
        Y = "y"
        X = "x"
        RADIUS = "radius"
        CONFIRMED = "confirmed"
        TRUTH = "truth"
        CHANNEL = "channel"
        ABS_Z = "abs_z"
        ABS_Y = "abs_y"
        ABS_X = "abs_x"
        REGION = "region"

    #: column -> index in the table; CLASS level and mutable, as in the reference: setting
    #: ``cols`` on any instance re-targets every accessor (global state)
    _col_inds: Dict["Blobs.Cols", Optional[int]] = {c: i for i, c in enumerate(Cols)}

    def __init__(self, blobs=None, blob_matches=None, colocalizations=None, path=None, cols=None):
        self._cols = None
        self._blobs = None
        self.cols = cols
        self.blobs = blobs
        self.blob_matches = blob_matches
        self.colocalizations = colocalizations
        self.path = path
        self.ver = self.BLOBS_NP_VER
        self.roi_offset = None
        self.roi_size = None
        self.resolutions = None
        self.basename = None
        self.scaling = np.ones(3)

    # -- column registry ---------------------------------------------------------------
    @property
    def cols(self) -> Optional[Sequence[str]]:
        return self._cols

    @cols.setter
    def cols(self, names: Optional[Sequence[str]]):
        self._cols = names
        if names is None:
            return
        index: Dict[Blobs.Cols, Optional[int]] = {c: None for c in self.Cols}
        for i, name in enumerate(names):
            try:
                index[self.Cols(name)] = i
            except ValueError:
                _logger.warning("%s is not a valid Blobs column, skipping", name)
        Blobs._col_inds = index

    @property
    def blobs(self) -> Optional[np.ndarray]:
        return self._blobs

    @blobs.setter
    def blobs(self, table):
        self._blobs = table
        if table is not None and self.cols is None:
            self.cols = [c.value for c in self.Cols][:table.shape[1]]

    @classmethod
    def _ind(cls, col):
        if isinstance(col, cls.Cols):
            return cls._col_inds[col]
        if isinstance(col, (list, tuple)):
            return [cls._ind(c) for c in col]
        return col

    @classmethod
    def _get_rel_inds(cls) -> List[int]:
        return [cls._col_inds[c] for c in (cls.Cols.Z, cls.Cols.Y, cls.Cols.X)]

    @classmethod
    def _get_abs_inds(cls) -> List[int]:
        return [cls._col_inds[c] for c in (cls.Cols.ABS_Z, cls.Cols.ABS_Y, cls.Cols.ABS_X)]

    # -- generic accessors -------------------------------------------------------------
    @classmethod
    def get_blob_col(cls, blob: np.ndarray, col):
        many = blob.ndim > 1
        if col is None:
            return np.array([]) if many else None
        col = cls._ind(col)
        return blob[..., col] if many else blob[col]

    @classmethod
    def set_blob_col(cls, blob: np.ndarray, col, val, mask=np.s_[:], **kwargs) -> np.ndarray:
        col = cls._ind(col)
        if blob.ndim > 1:
            blob[mask, ..., col] = val
        else:
            blob[col] = val
        return blob

    @classmethod
    def get_blob_confirmed(cls, blob):
        return cls.get_blob_col(blob, cls._col_inds[cls.Cols.CONFIRMED])

    @classmethod
    def set_blob_confirmed(cls, blob, *args, **kwargs):
        return cls.set_blob_col(blob, cls._col_inds[cls.Cols.CONFIRMED], *args, **kwargs)

    @classmethod
    def get_blob_truth(cls, blob):
        return cls.get_blob_col(blob, cls._col_inds[cls.Cols.TRUTH])

    @classmethod
    def set_blob_truth(cls, blob, *args, **kwargs):
        return cls.set_blob_col(blob, cls._col_inds[cls.Cols.TRUTH], *args, **kwargs)

    @classmethod
    def get_blobs_channel(cls, blob):
        return cls.get_blob_col(blob, cls._col_inds[cls.Cols.CHANNEL])

    @classmethod
    def set_blob_channel(cls, blob, *args, **kwargs):
        return cls.set_blob_col(blob, cls._col_inds[cls.Cols.CHANNEL], *args, **kwargs)

    @classmethod
    def get_blob_abs_coords(cls, blobs):
        return cls.get_blob_col(blobs, cls._get_abs_inds())

    @classmethod
    def set_blob_abs_coords(cls, blobs, coords, *args, **kwargs):
        cls.set_blob_col(blobs, cls._get_abs_inds(), coords, *args, **kwargs)
        return blobs

    # -- table construction ------------------------------------------------------------
    def format_blobs(self, channel=None) -> np.ndarray:
        """Pad to all 11 columns with -1, mirror rel -> abs coordinates, set the channel."""
        have = self.blobs.shape[1]
        pad = -np.ones((self.blobs.shape[0], len(self.Cols) - have))
        self.blobs = np.concatenate((self.blobs, pad), axis=1)
        self.cols = [c.value for c in self.Cols]
        self.blobs[:, self._get_abs_inds()] = self.blobs[:, self._get_rel_inds()]
        if channel is not None:
            self.set_blob_channel(self.blobs, channel)
        return self.blobs

    # -- coordinate helpers ------------------------------------------------------------
    @classmethod
    def shift_blobs(cls, blob, cols, fn, vals, to_int: bool = False):
        if blob is None:
            return blob
        sub = fn(blob[cols] if blob.ndim == 1 else blob[..., cols], vals)
        if to_int:
            sub = sub.astype(int)
        if blob.ndim == 1:
            blob[cols] = sub
        else:
            blob[..., cols] = sub
        return blob

    @classmethod
    def shift_blob_rel_coords(cls, blob, offset):
        return cls.shift_blobs(blob, cls._get_rel_inds(), np.add, offset)

    @classmethod
    def shift_blob_abs_coords(cls, blob, offset):
        return cls.shift_blobs(blob, cls._get_abs_inds(), np.add, offset)

    @classmethod
    def multiply_blob_rel_coords(cls, blob, factor):
        return cls.shift_blobs(blob, cls._get_rel_inds(), np.multiply, factor, True)

    @classmethod
    def multiply_blob_abs_coords(cls, blob, factor):
        return cls.shift_blobs(blob, cls._get_abs_inds(), np.multiply, factor, True)

    def remove_abs_blob_coords(self, remove_extra: bool = False) -> np.ndarray:
        """Drop the abs columns (and, with ``remove_extra``, any unnamed trailing columns)."""
        candidates = Blobs._col_inds.values() if remove_extra else range(self.blobs.shape[1])
        drop = set(Blobs._get_abs_inds())
        keep = [i for i in candidates if i not in drop]
        self.cols = [self.cols[i] for i in keep]
        self.blobs = _map_columns(self.blobs, keep)
        return self.blobs

    @classmethod
    def replace_rel_with_abs_blob_coords(cls, blobs: np.ndarray) -> np.ndarray:
        rel, ab = list(cls._get_rel_inds()), list(cls._get_abs_inds())
        if rel == list(range(rel[0], rel[0] + len(rel))):
            _map_columns(blobs, ab, out=blobs, dst0=rel[0])
        else:
            blobs[:, rel] = blobs[:, ab]
        return blobs

    @classmethod
    def blobs_in_channel(cls, blobs, channel, return_mask: bool = False):
        mask = None
        sel = blobs
        if channel is not None:
            mask = np.isin(cls.get_blobs_channel(blobs), channel)
            sel = blobs[mask]
        return (sel, mask) if return_mask else sel

    @classmethod
    def show_blobs_per_channel(cls, blobs):
        for chl in np.unique(cls.get_blobs_channel(blobs)):
            _logger.info("- blobs in channel %s: %s", int(chl), len(cls.blobs_in_channel(blobs, chl)))

    @classmethod
    def blob_for_db(cls, blob: np.ndarray) -> np.ndarray:
        rest = [cls._col_inds[c] for c in (cls.Cols.RADIUS, cls.Cols.CONFIRMED, cls.Cols.TRUTH,
                                            cls.Cols.CHANNEL)]
        return np.array([*blob[cls._get_abs_inds()], *blob[rest]])

    # -- archive -----------------------------------------------------------------------
    def save_archive(self, to_add=None, update: bool = False):
        """Write the uncompressed ``.npz`` archive (keys :class:`Keys`), backing up an
        existing file first."""
        if to_add is None:
            present = sorted(((c, i) for c, i in self._col_inds.items() if i is not None),
                             key=lambda e: e[1])
            arc = {
                self.Keys.VER.value: self.ver,
                self.Keys.BLOBS.value: self.blobs,
                self.Keys.RESOLUTIONS.value: self.resolutions,
                self.Keys.BASENAME.value: self.basename,
                self.Keys.ROI_OFFSET.value: self.roi_offset,
                self.Keys.ROI_SIZE.value: self.roi_size,
                self.Keys.COLOCS.value: self.colocalizations,
                self.Keys.COLS.value: [c.value for c, _ in present],
            }
        else:
            arc = to_add
        if update:
            with np.load(self.path, allow_pickle=True) as old:
                arc = {k: old[k] for k in old.files}
                arc.update(to_add)
        _backup_file(self.path)
        with open(self.path, "wb") as f:
            np.savez(f, **arc)
        _logger.info("Saved blobs archive to: %s", self.path)
        return arc

    def load_blobs(self, path: Optional[str] = None) -> "Blobs":
        if path is not None:
            self.path = path
        with np.load(self.path, allow_pickle=True) as arc:
            info = {}
            for k in arc.files:
                v = arc[k]
                info[k] = v.item() if v.ndim == 0 else v   # 0-d arrays hold Python scalars / None
        K = self.Keys
        if K.VER.value in info:
            self.ver = info[K.VER.value]
        if K.COLS.value in info:
            self.cols = list(info[K.COLS.value])
        if K.BLOBS.value in info:
            self.blobs = info[K.BLOBS.value]
        self.colocalizations = info.get(K.COLOCS.value, self.colocalizations)
        self.resolutions = info.get(K.RESOLUTIONS.value, self.resolutions)
        self.basename = info.get(K.BASENAME.value, self.basename)
        self.roi_offset = info.get(K.ROI_OFFSET.value, self.roi_offset)
        self.roi_size = info.get(K.ROI_SIZE.value, self.roi_size)
        if self.ver <= 4 and self.cols is not None:
            self.cols = self.cols[:len(self.cols) - 3]
        self.ver = self.BLOBS_NP_VER
        return self


def _backup_file(path: str) -> None:
    """Move an existing file aside as ``name(1).ext``, ``name(2).ext``, ..."""
    if not path or not os.path.exists(path):
        return
    stem, ext = os.path.splitext(path)
    i = 1
    while os.path.exists(f"{stem}({i}){ext}"):
        i += 1
    os.replace(path, f"{stem}({i}){ext}")


# ------------------------------------------------------------------------------------
def calc_scaling_factor() -> np.ndarray:
    """Pixels per physical unit, ``1 / resolutions[0]``."""
    if config.resolutions is None or len(config.resolutions) < 1:
        raise AttributeError("Must load resolutions from file or set a resolution")
    return np.divide(1.0, config.resolutions[0])


def calc_overlap(factor: Optional[int] = None) -> np.ndarray:
    """Block overlap in pixels per axis: ``ceil(scaling * factor)`` as ints."""
    if factor is None:
        factor = OVERLAP_FACTOR
    return np.ceil(np.multiply(calc_scaling_factor(), factor)).astype(int)


def _channels_of(roi_ndim: int, n_channels: int, channel, dim_channel: int = 3):
    """``plot_3d.setup_channels`` (reference magmap/plot/plot_3d.py:24-52)."""
    multichannel = roi_ndim > dim_channel
    if not multichannel:
        return False, [0]
    return True, (range(n_channels) if channel is None else channel)


def detect_blobs(roi, channel: Optional[Sequence[int]],
                 exclude_border: Optional[Sequence[int]] = None) -> Optional[np.ndarray]:
    """Detect blobs in one ROI -> ``(m, 11)`` float64 table, or ``None`` when nothing is found.

    ``roi`` is a ``(z, y, x[, c])`` array (host) or a
    :class:`magellanmapper_amd.blob_log.DeviceVolume`.  Reads ``config.resolutions`` and the
    per-channel ROI profile at every call.  ``exclude_border`` is ``[pad_start, pad_end]`` in
    z, y, x.
    """
    from . import blob_log as bl
    dvol = roi if isinstance(roi, bl.DeviceVolume) else bl.DeviceVolume(roi)
    tables = detect_blobs_blocks_device(dvol, channel, [(0, 0, 0)], [dvol.shape[:3]])
    blobs_all = tables[0]
    if blobs_all is None:
        return None
    if exclude_border is not None:
        blobs_all = get_blobs_interior(blobs_all, dvol.shape[:3], *exclude_border)
    return blobs_all


def detect_blobs_blocks_device(dvol, channel, origins, shapes, stats=None,
                               on_block=None, denoise_max_shape=None, exclude=None,
                               coloc: bool = False, sink=None, stack_finisher=None) -> List[Optional[np.ndarray]]:
    """:func:`detect_blobs` for many blocks of one resident volume in one device pass.

    Returns one 11-column table (block-relative coordinates) or ``None`` per block, rows
    ordered as the reference orders them: channels in turn, within a channel the pruned
    ``blob_log`` order.  ``on_block(i, table)`` (optional) post-processes each finished block
    table while the GPU is still busy with later batches; its return value replaces the table.
    With ``denoise_max_shape`` every block is saturated + denoised tile by tile on the device
    first (reference stack_detect.py:122-150; :mod:`preprocess`).  ``exclude(i)`` (optional) gives
    block ``i``'s border-exclusion matrix, applied as ``detect_blobs`` applies it (:952-955); with
    ``coloc`` the intensity co-localisation flags are then appended as extra columns
    (stack_detect.py:159-162; :mod:`colocalizer`) -- both before ``on_block``.
    ``sink(indices, peak_batch, channel) -> tables`` (optional) builds the finished tables of a batch itself from
    the native host path's arrays (``blob_log.PeakBatch``; ``stack_detect._ArenaSink`` writes them straight into the
    merged table): used for one channel without rescale / co-localisation, instead of ``exclude`` + ``on_block``.
    ``stack_finisher`` (optional; ``stack_detect._StackFinisher``): for a stack whose blocks are ONE batch, everything
    behind the kernels -- peaks, per-block prune, tables, the stack's pruning, final columns -- in one native call.
    """
    from . import blob_log as bl
    multichannel, channels = _channels_of(dvol.tensor.ndim, dvol.n_channels, channel)
    channels = list(channels)
    first = config.get_roi_profile(channels[0])
    isotropic = first["isotropic"]
    iso_factor = None
    pre = None
    keeper = None
    if denoise_max_shape is not None:
        from . import preprocess
        pre = preprocess.Preprocessor(denoise_max_shape)
        if coloc and isotropic is None and dvol.multichannel and \
                not any(getattr(config.get_roi_profile(c), "spectral_unmixing", None) for c in channels):
            # the co-localisation reads every channel's preprocessed blocks: keep them instead of making them twice
            if pre.retain(dvol, origins, shapes, channels):
                keeper = pre
    log_shapes = shapes
    if isotropic is not None:
        # interpolate every block to (near) isotropy for the detection, first channel's profile
        # (:893-897); blob coordinates go back to the original grid afterwards (:944-951)
        from . import preprocess
        iso_factor = preprocess.calc_isotropic_factor(isotropic)
        log_shapes = [preprocess.isotropic_shape(s, iso_factor) for s in shapes]
        pre = preprocess.Rescaler(iso_factor, channels, denoise_max_shape)
        pre.set_blocks(origins, shapes, log_shapes)
    per_block: List[List[np.ndarray]] = [[] for _ in shapes]
    done: List[Optional[np.ndarray]] = [None] * len(shapes)
    # Block tables of several channels straight from the native host path (``sink.emit``: stack_detect._ArenaSink): the
    # plain per-channel detection -- no rescale, no unmixing -- and, with co-localisation, every channel's image at hand
    # for the means (raw voxels, or the preprocessed blocks kept by ``Preprocessor.retain``)
    multi_sink = bool(
        sink is not None and hasattr(sink, "emit") and len(channels) > 1 and iso_factor is None and dvol.multichannel
        and bl.HOST_PATH == "native"
        and not any(getattr(config.get_roi_profile(c), "spectral_unmixing", None) for c in channels)
        and (not coloc or denoise_max_shape is None or keeper is not None))
    held: dict = {}
    batch_major = bool(len(channels) > 1 and (bl.BATCH_MAJOR is True or (
        bl.BATCH_MAJOR == "uploading" and getattr(dvol, "_upload", None) is not None)))
    Blobs(np.ones((1, 4))).format_blobs()      # bind the class-level column registry to the 11 columns
    lanes = []
    for chl in channels:
        settings = config.get_roi_profile(chl)
        source = pre
        spectral_unmixing = getattr(settings, "spectral_unmixing", None)
        if spectral_unmixing is not None:
            # x -= fac * roi[..., k]; x[x < 0] = 0 for every entry of this channel (:910-921)
            subtract = [(k, f) for spec_chl, spec in spectral_unmixing.items() if spec_chl == chl
                        for k, f in spec.items()]
            if subtract:
                from . import preprocess
                if isotropic is not None:       # resize every channel first, then unmix the resized ones
                    source = preprocess.Unmixer(subtract, denoise_max_shape, rescale=(iso_factor, channels))
                    source.set_blocks(origins, shapes, log_shapes)
                else:
                    source = preprocess.Unmixer(subtract, denoise_max_shape)
                source._raw_scale = (float(np.iinfo(dvol.np_dtype).max) if dvol.np_dtype.kind in "ui"
                                     else dvol.value_scale())
        scaling_factor = calc_scaling_factor()[2]          # x scaling alone, as the reference
        root3 = math.sqrt(3)

        def to_tables(indices, results, chl=chl):
            # radius = sigma * sqrt(3), then the 11 standard columns (reference :937-938:
            # Blobs(blobs_log).format_blobs(chl)) -- for all blocks of the batch in one table, the
            # per-block tables are row ranges of it
            lens = [r.shape[0] if r.size >= 1 else 0 for r in results]
            total = sum(lens)
            if total:
                big = np.empty((total, 11))
                at = 0
                for r, n in zip(results, lens):
                    if n:
                        big[at:at + n, :4] = r
                        at += n
                big[:, 3] = big[:, 3] * root3
                big[:, 4:6] = -1                       # confirmed, truth
                big[:, 6] = chl
                big[:, 7:10] = big[:, 0:3]             # abs <- rel
                big[:, 10] = -1                        # region
                at = 0
                for i, n in zip(indices, lens):
                    if n:
                        per_block[i].append(big[at:at + n])
                        at += n
            if chl == channels[-1]:            # the block tables of this batch are complete
                tbls = []
                for i in indices:
                    parts = per_block[i]
                    tbl = (parts[0] if len(parts) == 1 else np.vstack(parts)) if parts else None
                    if tbl is not None and iso_factor is not None:
                        Blobs.multiply_blob_rel_coords(tbl, 1 / iso_factor)
                        Blobs.multiply_blob_abs_coords(tbl, 1 / iso_factor)
                    ex = exclude(i) if exclude is not None else None
                    if tbl is not None and ex is not None:
                        tbl = get_blobs_interior(tbl, shapes[i], *ex)
                    tbls.append(tbl)
                if coloc:
                    tbls = _append_colocs(dvol, channels, [origins[i] for i in indices],
                                          [shapes[i] for i in indices], tbls, denoise_max_shape, keeper)
                for i, tbl in zip(indices, tbls):
                    done[i] = on_block(i, tbl) if on_block is not None else tbl

        to_sink = None
        if sink is not None and len(channels) == 1 and iso_factor is None and not coloc:
            def to_sink(indices, pb, chl=chl):
                for i, tbl in zip(indices, sink(indices, pb, chl)):
                    done[i] = tbl
        elif multi_sink:
            # several channels: a batch's peak arrays wait (a few hundred KB each) until its LAST channel has been
            # detected, then all of them go to the arena in one native call, the co-localisation flags with them
            def to_sink(indices, pb, chl=chl):
                key = tuple(indices)
                held.setdefault(key, {})[chl] = pb
                if chl != channels[-1]:
                    return
                pbs = held.pop(key)
                if sorted(pbs) != sorted(channels):
                    from . import _native as nat
                    raise nat.MmxError("the channels of a stack were batched differently: cannot assemble block tables")
                fn = None
                if coloc:
                    fn = lambda idx, rows5, row_offsets, flags_ptr, ld: _colocs_into(
                        dvol, channels, [origins[i] for i in idx], [shapes[i] for i in idx], rows5, row_offsets,
                        flags_ptr, ld, denoise_max_shape, keeper)
                for i, tbl in zip(indices, sink.emit(indices, [pbs[c] for c in channels], channels, fn)):
                    done[i] = tbl

        fin = None
        if stack_finisher is not None and to_sink is not None and not multi_sink and len(channels) == 1:
            def fin(indices, cands, n_cands, blocks, space, thr, eps, overlap, stats_, chl=chl):
                tables = stack_finisher.run(indices, cands, n_cands, blocks, space, thr, eps, overlap, stats_, chl)
                if tables is None:
                    return False
                for i, tbl in zip(indices, tables):
                    done[i] = tbl
                return True

        kwargs = dict(min_sigma=settings["min_sigma_factor"] * scaling_factor,
                      max_sigma=settings["max_sigma_factor"] * scaling_factor,
                      num_sigma=settings["num_sigma"], threshold=settings["detection_threshold"],
                      overlap=settings["overlap"], stats=stats, on_batch=to_tables, pre=source, sink=to_sink, finisher=fin)
        if batch_major:
            # every channel a lane of ONE pipeline: both channels of a batch of blocks before the next batch
            # (blob_log.blob_log_lanes; the closures above take the batches as they finish, the last channel's completes
            # the block tables)
            lanes.append(bl.Lane(chl if multichannel else 0, **kwargs))
        else:
            # (one channel after the other: every channel's call cuts the blocks into the same batches -- the
            #  multi-channel sink assembles a batch's tables from all channels' peak arrays)
            bl.blob_log_blocks(dvol, chl if multichannel else 0, origins, log_shapes, **kwargs,
                               plan_num_sigma=max(int(config.get_roi_profile(c)["num_sigma"]) for c in channels))
    if lanes:
        if len({ln.pre is None for ln in lanes}) > 1:       # (some channels preprocess, some do not: one after the other)
            for ln in lanes:
                bl.blob_log_lanes(dvol, [ln], origins, log_shapes)
        else:
            bl.blob_log_lanes(dvol, lanes, origins, log_shapes)
    return done


def _append_colocs(dvol, channels, origins, shapes, tables, denoise_max_shape, keeper=None):
    """``np.hstack((table, colocalize_blobs(block, table)))`` for the blocks of one batch
    (reference stack_detect.py:159-162).  The image the reference hands to ``colocalize_blobs`` is
    the block as detection saw it: preprocessed when ``denoise_max_shape`` is set."""
    from . import blob_log as bl
    from . import colocalizer
    if not dvol.multichannel:
        if any(t is not None for t in tables):       # np.hstack((segments, None)) in the reference
            raise ValueError("all the input arrays must have same number of dimensions: a "
                             "single-channel ROI cannot be co-localised")
        return tables
    dev = dvol.tensor.device
    if not any(t is not None and len(t) for t in tables):
        return [None if t is None else np.hstack((t, np.zeros((len(t), dvol.n_channels)))) for t in tables]
    volumes = {}
    if denoise_max_shape is None:
        blocks, _ = bl._make_blocks(dvol, 0, origins, shapes)
        d_blocks = bl._to_device_bytes(blocks, dev)
        for c in channels:
            volumes[c] = dvol.view(c, False)
        flags = colocalizer.colocalize_blocks_device(volumes, blocks, d_blocks, shapes, tables,
                                                     dvol.n_channels, dev)
    else:
        # one preprocessed channel at a time through a buffer set of its own
        from . import preprocess
        global _coloc_pre
        if _coloc_pre is None or _coloc_pre.dms != [int(v) for v in denoise_max_shape]:
            _coloc_pre = preprocess.Preprocessor(denoise_max_shape)
        kept = {c: (None if keeper is None else keeper.retained_view(c, origins, shapes)) for c in channels}
        if all(v is not None for v in kept.values()):
            # what detection preprocessed is still there for every channel (Preprocessor.retain): all kernels queued,
            # one wait, flags per block
            blocks_of = {c: kept[c][0] for c in channels}
            volumes = {c: kept[c][1] for c in channels}
            d_blocks_of = {c: bl._to_device_bytes(blocks_of[c], dev) for c in channels}
            flags = colocalizer.colocalize_blocks_device(volumes, blocks_of, d_blocks_of, shapes, tables,
                                                         dvol.n_channels, dev)
        else:
            flags = None
            for c in channels:          # (one buffer set: a channel is read before the next one overwrites it)
                if kept[c] is not None:
                    blocks, vol64 = kept[c]
                else:
                    blocks, _, _, vol64 = _coloc_pre.run(dvol, c, origins, shapes, 0)
                d_blocks = bl._to_device_bytes(blocks, dev)
                part = colocalizer.colocalize_blocks_device({c: vol64}, blocks, d_blocks, shapes, tables,
                                                            dvol.n_channels, dev, means_only=True)
                flags = part if flags is None else [None if a is None else np.where(np.isnan(a), b, a)
                                                    for a, b in zip(flags, part)]
            flags = [None if (t is None or m is None) else
                     colocalizer._flags_from_means(t, m, shp, dvol.n_channels)
                     for t, m, shp in zip(tables, flags, shapes)]
    return [None if t is None else np.hstack((t, f)) for t, f in zip(tables, flags)]


def _colocs_into(dvol, channels, origins, shapes, rows5, row_offsets, flags_ptr, ld, denoise_max_shape, keeper):
    """The co-localisation flags of one batch of block tables, written into the tables' extra columns where they lie
    (``mmx_host_coloc_flags``): ``rows5`` -- block, z, y, x (block-relative), channel of every row, ``row_offsets`` the
    blocks' row ranges -- go to the device once, every image channel's per-blob means come back behind ONE wait, thresholds
    and flags are taken natively (``colocalizer._flags_from_means`` is the same rule block by block in NumPy)."""
    import torch
    from . import _native as nat
    from . import blob_log as bl
    L = nat.lib()
    dev = dvol.tensor.device
    n = len(rows5)
    nb = len(shapes)
    order = sorted(int(c) for c in channels)
    if denoise_max_shape is None:
        blocks, _ = bl._make_blocks(dvol, 0, origins, shapes)
        d_blocks = bl._to_device_bytes(blocks, dev)
        views = {c: (blocks, d_blocks, dvol.view(c, False)) for c in order}
    else:
        views = {}
        for c in order:
            blocks, vol64 = keeper.retained_view(c, origins, shapes)
            views[c] = (blocks, bl._to_device_bytes(blocks, dev), vol64)
    d_rows = bl.to_device(rows5.reshape(-1), dev)
    d_off = bl.to_device(row_offsets.astype(np.int32), dev)
    d_mean = torch.empty((len(order), n), dtype=torch.float64, device=dev)
    d_cnt = torch.empty(n, dtype=torch.int32, device=dev)
    stream = bl._stream_ptr()
    for k, c in enumerate(order):
        blocks, d_blocks, vol = views[c]
        nat.check(L.mmx_coloc_means(ctypes.byref(vol), d_blocks.data_ptr(), len(blocks), d_rows.data_ptr(),
                                    d_off.data_ptr(), n, d_mean[k].data_ptr(), d_cnt.data_ptr(), stream),
                  "mmx_coloc_means")
    h_mean = np.ascontiguousarray(d_mean.cpu().numpy())         # (one wait for every channel's kernel)
    shp = np.ascontiguousarray(shapes, dtype=np.int32).reshape(nb, 3)
    chans = np.ascontiguousarray(order, dtype=np.int32)
    offs = np.ascontiguousarray(row_offsets, dtype=np.int64)
    rc = L.mmx_host_coloc_flags(h_mean.ctypes.data, chans.ctypes.data, len(order), n, rows5.ctypes.data, offs.ctypes.data,
                                nb, shp.ctypes.data, dvol.n_channels, flags_ptr, ld)
    if rc == 1:
        raise IndexError(f"a blob's channel is out of bounds for axis 0 with size {dvol.n_channels}")
    nat.check(rc, "mmx_host_coloc_flags")


_coloc_pre = None


# ------------------------------------------------------------------------------------
def sort_blobs(blobs: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    order = np.lexsort(tuple(blobs[:, i] for i in range(2, -1, -1)))
    return blobs[order], order


def _smallest_signed_int(max_val) -> type:
    """``libmag.dtype_within_range(0, max, True, True)`` (reference magmap/io/libmag.py:1116-1152)."""
    for dt in (np.int8, np.int16, np.int32, np.int64):
        if np.iinfo(dt).max >= max_val:
            return dt
    raise TypeError("unable to find an integer type for the coordinate range")


#: positions (in the ``blobs`` argument) of the rows kept by the last :func:`remove_close_blobs`
last_survivors = None


def find_close_pairs(check_zyx: np.ndarray, master_zyx: np.ndarray, tol) -> Tuple[np.ndarray, np.ndarray]:
    """Device all-pairs search: ``(last_check_per_master, check_hit)``.

    For each master row the index of the LAST check row with ``|d| <= tol`` on all three
    axes (-1 if none), and for each check row whether any master row matched.  This is
    what the reference's chunked ``_find_close_blobs`` loop boils down to once NumPy's
    duplicate fancy-index assignment (last write wins) and ``np.delete`` are applied
    (reference detector.py:1049-1083).
    """
    import ctypes

    import torch

    from . import _native as nat
    from .blob_log import _require_gpu, _stream_ptr
    dev = _require_gpu()
    L = nat.lib()
    m = np.ascontiguousarray(master_zyx, dtype=np.int32)
    c = np.ascontiguousarray(check_zyx, dtype=np.int32)
    d_m = torch.from_numpy(m).to(dev)
    d_c = torch.from_numpy(c).to(dev)
    d_last = torch.empty(len(m), dtype=torch.int32, device=dev)
    d_hit = torch.zeros(max(1, len(c)), dtype=torch.uint8, device=dev)
    t = (ctypes.c_int32 * 3)(*[int(v) for v in np.broadcast_to(np.asarray(tol), (3,))])
    nat.check(L.mmx_close_pairs(d_m.data_ptr(), len(m), d_c.data_ptr(), len(c), t,
                                d_last.data_ptr(), d_hit.data_ptr(), _stream_ptr()), "mmx_close_pairs")
    return d_last.cpu().numpy(), d_hit.cpu().numpy()[:len(c)].astype(bool)


def remove_close_blobs(blobs: np.ndarray, blobs_master: np.ndarray, tol,
                       chunk_size: int = 1000) -> Tuple[np.ndarray, np.ndarray]:
    """Remove rows of ``blobs`` that lie within ``tol`` of a row of ``blobs_master``.

    Returns ``(pruned, blobs_master)``; each matched master row's *abs* coordinates become
    ``np.around((abs_master + abs_check) / 2)`` (round half to even) for its last matching
    check row.  ``chunk_size`` is accepted for signature compatibility (the device search
    needs no chunking).
    """
    if len(blobs) < 1 or len(blobs_master) < 1:
        return blobs, blobs_master
    # the reference compares coordinates cast to the smallest signed int type that holds
    # them; integer-valued float coordinates survive the cast unchanged
    dtype = _smallest_signed_int(np.amax((np.amax(blobs[:, :3]), np.amax(blobs_master[:, :3]))))
    last, hit = find_close_pairs(blobs[:, :3].astype(dtype), blobs_master[:, :3].astype(dtype),
                                 np.asarray(tol))
    global last_survivors
    last_survivors = np.nonzero(~hit)[0]     # positions in ``blobs`` that were kept
    pruned = blobs[last_survivors]
    matched = np.nonzero(last >= 0)[0]
    Blobs(blobs)  # as the reference does: (re)binds the class-level column indices to this table
    if len(matched):
        abs_inds = Blobs._get_abs_inds()
        between = np.around(np.divide(
            np.add(blobs_master[matched][:, abs_inds], blobs[last[matched]][:, abs_inds]), 2))
        blobs_master[np.ix_(matched, abs_inds)] = between
    return pruned, blobs_master


def meas_pruning_ratio(num_blobs_orig, num_blobs_after_pruning, num_blobs_next):
    if num_blobs_next > 0 and num_blobs_orig > 0:
        return (num_blobs_orig, num_blobs_after_pruning / num_blobs_orig,
                num_blobs_after_pruning / num_blobs_next)
    return None


def get_blobs_in_roi(blobs: np.ndarray, offset, size, margin=(0, 0, 0), reverse: bool = True):
    """Blobs inside ``[offset - margin, offset + size + margin)``; ``reverse`` takes the
    arguments in x, y, z order (the reference's ROI convention)."""
    if reverse:
        offset, size, margin = offset[::-1], size[::-1], margin[::-1]
    mask = np.ones(len(blobs), dtype=bool)
    for ax in range(3):
        mask &= blobs[:, ax] >= offset[ax] - margin[ax]
        mask &= blobs[:, ax] < offset[ax] + size[ax] + margin[ax]
    return blobs[mask], mask


def get_blobs_interior(blobs: np.ndarray, shape, pad_start, pad_end) -> np.ndarray:
    mask = np.ones(len(blobs), dtype=bool)
    for ax in range(3):
        mask &= blobs[:, ax] >= pad_start[ax]
        mask &= blobs[:, ax] < shape[ax] - pad_end[ax]
    return blobs[mask]
