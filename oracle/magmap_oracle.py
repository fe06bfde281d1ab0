"""CPU ORACLE (test infrastructure, NOT product code) -- magmap block/prune logic restated.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product package ``magellanmapper_amd`` never does.

A compact, purely functional NumPy restatement of what the reference wraps
around ``blob_log`` for whole-volume detection.  State that the reference keeps in
module globals (``config.resolutions``, the per-channel ROI profile) is passed
explicitly.  Every function cites the reference lines it follows
(paths relative to ``/root/reference``).

Pinned by golden vectors captured from the real reference
(``tests/golden/make_golden.py`` -> ``tests/test_oracle_golden.py``).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import blob_log_oracle as blo

#: ``Blobs.Cols`` order, magmap/cv/detector.py:88-113.
COLS = ("z", "y", "x", "radius", "confirmed", "truth", "channel",
        "abs_z", "abs_y", "abs_x", "region")
#: magmap/cv/detector.py:41.
OVERLAP_FACTOR = 5


# ----------------------------------------------------------------- A6 / A7
def scaling_factor(resolutions) -> np.ndarray:
    """``1 / resolutions[0]``; magmap/cv/detector.py:810-825."""
    if resolutions is None or len(resolutions) < 1:
        raise AttributeError("Must load resolutions from file or set a resolution")
    return np.divide(1.0, resolutions[0])


def calc_overlap(resolutions, factor: Optional[int] = None) -> np.ndarray:
    """``ceil(scale * 5)`` as int; magmap/cv/detector.py:828-841."""
    if factor is None:
        factor = OVERLAP_FACTOR
    return np.ceil(np.multiply(scaling_factor(resolutions), factor)).astype(int)


def format_blobs(blobs4: np.ndarray, channel: Optional[int]) -> np.ndarray:
    """Widen ``(z, y, x, radius)`` to the 11 reference columns.

    magmap/cv/detector.py:325-364: pad with ``-1``, copy rel -> abs coords,
    set the channel.
    """
    n = blobs4.shape[0]
    out = np.concatenate((blobs4, np.ones((n, len(COLS) - blobs4.shape[1])) * -1), axis=1)
    out[:, 7:10] = out[:, 0:3]
    if channel is not None:
        out[:, 6] = channel
    return out


def blobs_interior(blobs: np.ndarray, shape, pad_start, pad_end) -> np.ndarray:
    """magmap/cv/detector.py:1248-1268."""
    keep = np.ones(len(blobs), dtype=bool)
    for ax in range(3):
        keep &= blobs[:, ax] >= pad_start[ax]
        keep &= blobs[:, ax] < shape[ax] - pad_end[ax]
    return blobs[keep]


def detect_blobs(roi: np.ndarray, channel: Optional[Sequence[int]], profiles: Sequence[dict],
                 resolutions, exclude_border=None) -> Optional[np.ndarray]:
    """Per-ROI detection -> ``(m, 11)`` float64 table or ``None``.

    magmap/cv/detector.py:874-957.  The isotropic rescale (:893-897, :944-951; first channel's
    profile) goes through ``isotropic_oracle``.  Spectral unmixing (:910-921) is
    taken from the profile dict's ``"spectral_unmixing"`` entry (an attribute of the
    reference's ``ROIProfile``): ``{channel: {channel_to_subtract: factor}}``.  ``profiles[i]`` is what
    ``config.get_roi_profile(i)`` returns (magmap/settings/config.py:887-901: the
    last/only profile serves every channel beyond the list).
    """
    shape = roi.shape
    multichannel = roi.ndim > 3                       # plot_3d.setup_channels, plot_3d.py:24-52
    if multichannel:
        channels = range(roi.shape[3]) if channel is None else channel
    else:
        channels = [0]
    scale_x = scaling_factor(resolutions)[2]          # detector.py:907-908
    first = profiles[channels[0]] if len(profiles) > channels[0] else profiles[0]
    isotropic = first.get("isotropic")
    if isotropic is not None:
        from . import isotropic_oracle
        roi = isotropic_oracle.make_isotropic(roi, isotropic, np.asarray(resolutions)[0])
    tables = []
    for chl in channels:
        roi_detect = roi[..., chl] if multichannel else roi
        prof = profiles[chl] if len(profiles) > chl else profiles[0]
        unmix = prof.get("spectral_unmixing")
        if unmix is not None:
            for spec_chl, spec_subtr in unmix.items():
                if spec_chl != chl:
                    continue
                for subt_chl, subt_fac in spec_subtr.items():
                    roi_subt = roi[..., subt_chl]
                    roi_detect = np.subtract(roi_detect, subt_fac * roi_subt)
                    roi_detect[roi_detect < 0] = 0
        res = blo.blob_log(
            roi_detect,
            min_sigma=prof["min_sigma_factor"] * scale_x,
            max_sigma=prof["max_sigma_factor"] * scale_x,
            num_sigma=prof["num_sigma"], threshold=prof["detection_threshold"],
            overlap=prof["overlap"])
        if res.size < 1:
            continue
        res[:, 3] = res[:, 3] * math.sqrt(3)          # detector.py:937
        tables.append(format_blobs(res, chl))
    if not tables:
        return None
    out = np.vstack(tables)
    if isotropic is not None:                         # back to the original grid (:944-951)
        from . import isotropic_oracle
        factor = isotropic_oracle.calc_isotropic_factor(isotropic, np.asarray(resolutions)[0])
        out[:, 0:3] = np.multiply(out[:, 0:3], 1 / factor).astype(int)
        out[:, 7:10] = np.multiply(out[:, 7:10], 1 / factor).astype(int)
    if exclude_border is not None:
        out = blobs_interior(out, shape, *exclude_border)   # detector.py:953-955
    return out


# ------------------------------------------------------------------ A8 / A9
def stack_splitter(shape, max_pixels, overlap=None):
    """Block grid: slices (object array) and float offsets.

    magmap/cv/chunking.py:170-256: grid = ceil(shape / max_pixels); block k spans
    ``[k*mp, min(k*mp + mp + overlap, size))``.
    """
    shape3 = np.asarray(shape[:3])
    num = np.floor_divide(shape3, max_pixels)
    num[np.remainder(shape3, max_pixels) > 0] += 1
    num = num.astype(int)
    slices = np.zeros(num, dtype=object)
    offsets = np.zeros(np.append(num, 3))
    for z in range(num[0]):
        for y in range(num[1]):
            for x in range(num[2]):
                coord = (z, y, x)
                bounds = []
                for ax in range(3):
                    start = coord[ax] * max_pixels[ax]
                    end = start + max_pixels[ax]
                    if overlap is not None:
                        end += overlap[ax]
                    end = min(end, shape[ax])
                    bounds.append((int(start), int(end)))
                slices[coord] = tuple(slice(*b) for b in bounds)
                offsets[coord] = [b[0] for b in bounds]
    return slices, offsets


def setup_blocks(profile: dict, shape, resolutions) -> Dict[str, object]:
    """Block parameters; magmap/cv/stack_detect.py:282-335 (fields of ``Blocks``, :260-279)."""
    scale = scaling_factor(resolutions)
    denoise_size = profile["denoise_size"]
    denoise_max_shape = None
    if denoise_size:
        denoise_max_shape = np.ceil(np.multiply(scale, denoise_size)).astype(int)
    overlap_base = calc_overlap(resolutions)
    tol = np.multiply(overlap_base, profile["prune_tol_factor"]).astype(int)
    overlap_padding = np.copy(tol)
    overlap = np.copy(overlap_base)
    exclude_border = profile["exclude_border"]
    if exclude_border is not None:
        thresh = np.multiply(2, exclude_border)
        less = np.less(overlap, thresh)
        overlap[less] = thresh[less]
        excluded = np.greater(exclude_border, 0)
        overlap[excluded] += 1
        overlap_padding[excluded] = 0
    max_pixels = np.ceil(np.multiply(scale, profile["segment_size"])).astype(int)
    slices, offsets = stack_splitter(shape, max_pixels, overlap)
    return dict(sub_roi_slices=slices, sub_rois_offsets=offsets,
                denoise_max_shape=denoise_max_shape, exclude_border=exclude_border,
                tol=tol, overlap_base=overlap_base, overlap=overlap,
                overlap_padding=overlap_padding, max_pixels=max_pixels)


# ----------------------------------------------------------------------- A10
def detect_sub_roi(coord, offset, last_coord, exclude_border, sub_roi, channel,
                   profiles, resolutions, denoise_max_shape=None, near_max=(-1.0,),
                   coloc=False) -> Optional[np.ndarray]:
    """One block; magmap/cv/stack_detect.py:81-172.  With ``denoise_max_shape`` the block is
    preprocessed first (:122-150, ``preprocess_oracle``); with ``coloc`` the intensity
    co-localisation flags are appended as extra columns (:159-162, ``coloc_oracle``)."""
    if denoise_max_shape is not None:
        from . import preprocess_oracle as ppo
        sub_roi = ppo.preprocess_block(sub_roi, denoise_max_shape, profiles, near_max)
    if exclude_border is None:
        exclude = None
    else:
        exclude = np.array([exclude_border, exclude_border])
        exclude[0, np.equal(coord, 0)] = 0
        exclude[1, np.equal(coord, last_coord)] = 0
    segments = detect_blobs(sub_roi, channel, profiles, resolutions, exclude)
    if coloc and segments is not None:
        from . import coloc_oracle
        colocs = coloc_oracle.colocalize_blobs(sub_roi, segments)
        if colocs is None:
            raise ValueError("all the input arrays must have same number of dimensions")  # np.hstack
        segments = np.hstack((segments, colocs))
    if segments is not None:
        segments[:, 0:3] = np.add(segments[:, 0:3], offset)
        segments[:, 7:10] = np.add(segments[:, 7:10], offset)
    return segments


def detect_blobs_sub_rois(img, slices, offsets, exclude_border, channel, profiles, resolutions,
                          denoise_max_shape=None, near_max=(-1.0,), coloc=False):
    """Serial version of the Pool fan-out, magmap/cv/stack_detect.py:174-257."""
    last_coord = np.subtract(slices.shape, 1)
    seg_rois = np.zeros(slices.shape, dtype=object)
    for z in range(slices.shape[0]):
        for y in range(slices.shape[1]):
            for x in range(slices.shape[2]):
                coord = (z, y, x)
                seg_rois[coord] = detect_sub_roi(
                    coord, offsets[coord], last_coord, exclude_border,
                    img[slices[coord]], channel, profiles, resolutions, denoise_max_shape, near_max,
                    coloc)
    return seg_rois


# ----------------------------------------------------------------------- A11
def merge_blobs(blob_rois) -> Optional[np.ndarray]:
    """Concatenate block tables, tagging rows with the block (z, y, x); chunking.py:410-445."""
    rows = []
    for z in range(blob_rois.shape[0]):
        for y in range(blob_rois.shape[1]):
            for x in range(blob_rois.shape[2]):
                blobs = blob_rois[z, y, x]
                if blobs is None or (isinstance(blobs, int) and blobs == 0):
                    continue
                extras = np.zeros((blobs.shape[0], 3), dtype=int)
                extras[:] = (z, y, x)
                rows.append(np.concatenate((blobs, extras), axis=1))
    return np.vstack(rows) if rows else None


# ----------------------------------------------------------------------- A13
def _smallest_signed_dtype(max_val):
    """magmap/io/libmag.py:1116-1152 with ``integer=True, signed=True``."""
    for dt in (np.int8, np.int16, np.int32, np.int64):
        if np.iinfo(dt).min <= 0 and np.iinfo(dt).max >= max_val:
            return dt
    raise TypeError("no integer type holds the coordinate range")


def remove_close_blobs(blobs: np.ndarray, blobs_master: np.ndarray, tol, chunk_size: int = 1000):
    """Drop rows of ``blobs`` within ``tol`` (all 3 axes) of any master row.

    magmap/cv/detector.py:1000-1085: integer all-pairs compare in
    ``chunk_size`` x ``chunk_size`` tiles (master-major), every matched check row
    is deleted, every matched master row's *abs* coords become
    ``np.around((abs_master + abs_check) / 2)`` -- round-half-even, computed from
    the pre-update master values, duplicates resolved by NumPy's last-write-wins.
    """
    n_check, n_master = len(blobs), len(blobs_master)
    if n_check < 1 or n_master < 1:
        return blobs, blobs_master
    dtype = _smallest_signed_dtype(
        np.amax((np.amax(blobs[:, :3]), np.amax(blobs_master[:, :3]))))
    match_check = None
    match_master = None
    i = 0
    while i * chunk_size < n_master:
        ref = blobs_master[i * chunk_size:(i + 1) * chunk_size, :3].astype(dtype)
        j = 0
        while j * chunk_size < n_check:
            chk = blobs[j * chunk_size:(j + 1) * chunk_size].astype(dtype)
            diffs = np.abs(ref[:, None, :3] - chk[:, :3])
            close_master, close = np.nonzero((diffs <= tol).all(2))
            close = close + j * chunk_size
            close_master = close_master + i * chunk_size
            match_check = close if match_check is None else np.concatenate((match_check, close))
            match_master = (close_master if match_master is None
                            else np.concatenate((match_master, close_master)))
            j += 1
        i += 1
    pruned = np.delete(blobs, match_check, axis=0)
    abs_between = np.around(np.divide(
        np.add(blobs_master[match_master][:, 7:10], blobs[match_check][:, 7:10]), 2))
    updated = blobs_master[match_master]
    updated[:, 7:10] = abs_between
    blobs_master[match_master] = updated
    return pruned, blobs_master


# ----------------------------------------------------------------------- A12
def _meas_pruning_ratio(n_orig, n_after, n_next):
    """magmap/cv/detector.py:1126-1147."""
    if n_next > 0 and n_orig > 0:
        return (n_orig, n_after / n_orig, n_after / n_next)
    return None


def prune_overlap(i, pruner):
    """magmap/cv/stack_detect.py:643-677."""
    blobs, axis, tol, blobs_next = pruner
    if blobs is None:
        return None, None
    axis_col = blobs.shape[1] - 3 + axis
    n_orig = len(blobs)
    master = blobs[blobs[:, axis_col] == i]
    check = blobs[blobs[:, axis_col] == i + 1]
    pruned, master = remove_close_blobs(check, master, tol)
    after = np.concatenate((master, pruned))
    ratios = None
    if blobs_next is not None:
        ratios = _meas_pruning_ratio(n_orig, len(after), len(blobs_next))
    return after, ratios


def prune_blobs_mp(img_shape, seg_rois, overlap, tol, slices, offsets, channels,
                   overlap_padding=None):
    """Cross-block duplicate removal, axis by axis; stack_detect.py:679-861 (serial).

    ``img_shape`` replaces the ``img`` argument: the reference only uses
    ``img[sub_roi_slices[coord]].shape`` (:743-744).
    Returns ``(blobs_all (M, 11), ratios dict)`` or ``(None, None)``.
    """
    merged = merge_blobs(seg_rois)
    if merged is None:
        return None, None
    blobs_all = []
    ratios_all: Dict[str, list] = {}
    cols = ("blobs", "ratio_pruning", "ratio_adjacent")
    if overlap_padding is None:
        overlap_padding = tol
    for chl in channels:
        blobs = merged[np.isin(merged[:, 6], chl)]       # Blobs.blobs_in_channel, detector.py:747-772
        for axis in range(3):
            num_sections = offsets.shape[axis]
            if num_sections <= 1:
                continue
            non_ol_all = None
            to_prune = []
            coord_last = tuple(np.subtract(slices.shape, 1))
            for j in range(num_sections):
                coord = np.zeros(3, dtype=int)
                coord[axis] = j
                offset = offsets[tuple(coord)]
                size = _slice_shape(slices[tuple(coord)], img_shape)
                blobs_ol = None
                blobs_ol_next = None
                in_non_ol = []
                shift = overlap[axis] + overlap_padding[axis]
                off_ax = offset[axis]
                if j < num_sections - 1:
                    bounds = [off_ax + size[axis] - shift,
                              off_ax + size[axis] + overlap_padding[axis]]
                    blobs_ol = blobs[np.all([blobs[:, axis] >= bounds[0],
                                             blobs[:, axis] < bounds[1]], axis=0)]
                    start = off_ax + size[axis] + tol[axis]
                    bounds_next = [start, start + overlap[axis] + 2 * overlap_padding[axis]]
                    full = np.add(offsets[coord_last], size[:3])
                    if np.all(np.less(bounds_next, full[axis])):
                        blobs_ol_next = blobs[np.all([blobs[:, axis] >= bounds_next[0],
                                                      blobs[:, axis] < bounds_next[1]], axis=0)]
                    in_non_ol.append(blobs[:, axis] < bounds[0])
                else:
                    in_non_ol.append(blobs[:, axis] < off_ax + size[axis])
                start = off_ax
                if j > 0:
                    start += shift
                in_non_ol.append(blobs[:, axis] >= start)
                non_ol = blobs[np.all(in_non_ol, axis=0)]
                if non_ol_all is None:
                    non_ol_all = non_ol
                elif non_ol is not None:
                    non_ol_all = np.concatenate((non_ol_all, non_ol))
                to_prune.append((blobs_ol, axis, tol, blobs_ol_next))
            ol_all = None
            for j, pruner in enumerate(to_prune):
                pruned, ratios = prune_overlap(j, pruner)
                if ol_all is None:
                    ol_all = pruned
                elif pruned is not None:
                    ol_all = np.concatenate((ol_all, pruned))
                if ratios:
                    for col, val in zip(cols, ratios):
                        ratios_all.setdefault(col, []).append(val)
            if ol_all is None:
                blobs = non_ol_all
            elif non_ol_all is None:
                blobs = ol_all
            else:
                blobs = np.concatenate((non_ol_all, ol_all))
        blobs_all.append(blobs)
    out = np.vstack(blobs_all)[:, :-3]
    return out, ratios_all


def _slice_shape(slc, img_shape):
    return tuple(len(range(*s.indices(n))) for s, n in zip(slc, img_shape))


# ----------------------------------------------------------------------- A14
def detect_blobs_blocks(roi: np.ndarray, channels: Optional[Sequence[int]],
                        profiles: Sequence[dict], resolutions, near_max=(-1.0,), coloc=False):
    """Whole-ROI detection + pruning -> final ``(M, 8)`` table (or None) and stages.

    magmap/cv/stack_detect.py:338-517 for ``full_roi=True, coloc=False,
    verify=False``: block settings come from the first channel's profile (:402),
    rel <- abs (:461), abs columns dropped (:467) leaving
    ``z, y, x, radius, confirmed, truth, channel, region``.
    """
    if channels is None:
        channels = range(roi.shape[3]) if roi.ndim > 3 else [0]
    prof0 = profiles[channels[0]] if len(profiles) > channels[0] else profiles[0]
    blocks = setup_blocks(prof0, roi.shape, resolutions)
    seg_rois = detect_blobs_sub_rois(
        roi, blocks["sub_roi_slices"], blocks["sub_rois_offsets"],
        blocks["exclude_border"], channels, profiles, resolutions, blocks["denoise_max_shape"], near_max,
        coloc)
    merged_before = merge_blobs(seg_rois)
    segments_all, ratios = prune_blobs_mp(
        roi.shape, seg_rois, blocks["overlap"], blocks["tol"], blocks["sub_roi_slices"],
        blocks["sub_rois_offsets"], channels, blocks["overlap_padding"])
    final = None
    colocs = None
    if segments_all is not None:
        segments_all[:, 0:3] = segments_all[:, 7:10]
        if coloc:
            # the reference reads the flags starting at column 10 -- the ``region`` column that was
            # added after this line was written -- so row = [uint8(region), flags of all channels but
            # the last] (stack_detect.py:463-464)
            num_chls_roi = 1 if roi.ndim < 4 else roi.shape[3]
            colocs = segments_all[:, 10:10 + num_chls_roi].astype(np.uint8)
        final = segments_all[:, [0, 1, 2, 3, 4, 5, 6, 10]]
    return final, dict(blocks=blocks, seg_rois=seg_rois, merged=merged_before,
                       pruned11=segments_all, ratios=ratios, colocs=colocs)


# ----------------------------------------------------------------------- A15
#: keys that must agree for channels to share one set of blocks (magmap/settings/roi_prof.py:35-41)
BLOCK_SIZES = ("segment_size", "denoise_size", "prune_tol_factor", "sub_stack_max_pixels", "isotropic")


def detect_blobs_stack(roi: np.ndarray, profiles: Sequence[dict], resolutions, near_max=(-1.0,), coloc=False):
    """Whole-image detection over all channels -> ``(final table | None, grouped)``.

    magmap/cv/stack_detect.py:520-615: the channels share one set of blocks when their profiles agree on
    every ``BLOCK_SIZES`` key (:554-561, ``ROIProfile.is_identical_settings``, roi_prof.py:272-297: ``==`` on
    the values of the first profile against each other one); otherwise every channel is detected and pruned
    on its own block grid (``detect_blobs_blocks`` with ``channels=[c]``) and the final tables are concatenated
    in channel order (``libmag.combine_arrs``)."""
    channels = list(range(roi.shape[3])) if roi.ndim > 3 else [0]
    profs = [profiles[c] if len(profiles) > c else profiles[0] for c in channels]
    grouped = all(profs[0].get(k) == p.get(k) for p in profs[1:] for k in BLOCK_SIZES)
    groups = [channels] if grouped else [[c] for c in channels]
    finals = []
    for chl in groups:
        final, _ = detect_blobs_blocks(roi, chl, profiles, resolutions, near_max=near_max, coloc=coloc)
        if final is not None:
            finals.append(final)
    if not finals:
        return None, grouped
    return (finals[0] if len(finals) == 1 else np.concatenate(finals)), grouped
