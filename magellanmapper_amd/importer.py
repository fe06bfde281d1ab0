"""On-disk image format on the input side of the detection path.

Mirror of the parts of ``magmap.io.importer`` / ``magmap.io.np_io`` that ``mm <img> --proc detect``
goes through before ``stack_detect.detect_blobs_stack`` (reference magmap/io/importer.py):

* ``<base>_image5d.npy`` -- the ``(t, z, y, x[, c])`` array, opened memory-mapped (:794);
* ``<base>_meta.yml``    -- YAML metadata written by :func:`save_image_info` (:482-522): ``ver``
  (``IMAGE5D_NP_VER`` = 15, :69), ``names``, ``sizes``, ``resolutions``, ``magnification``, ``zoom``,
  ``near_min``, ``near_max``, ``scaling``, ``plane``; pre-v1.4 archives kept the same keys in
  ``<base>_meta.npz`` (:636-650);
* :func:`load_metadata` / :func:`assign_metadata` (:602-745) push ``resolutions``, ``magnification``,
  ``zoom``, ``near_min`` and ``near_max`` into :mod:`config`, which is where the detection path and
  the preprocessing read them.

Importing from TIFF / Bio-Formats, metadata version upgrades and ROI loading stay in the reference.
The device side: :class:`blob_log.DeviceVolume` uploads a memory-mapped image plane block by plane
block through a pinned staging buffer, so the host never holds a second copy of the stack.
"""
from __future__ import annotations

import logging
import os
from typing import Any, Dict, Optional, Sequence, Tuple

import numpy as np
import yaml

from . import config
from .stack_detect import Image5d

_logger = logging.getLogger("magellanmapper_amd")

IMAGE5D_NP_VER = 15
SUFFIX_IMAGE5D = "image5d.npy"
SUFFIX_META = "meta.yml"
_EXTENSIONS_MULTIPLE = (".tar", ".nii")


def splitext(path: str) -> Tuple[str, str]:
    """``libmag.splitext`` (reference libmag.py:272-293): multi-dot extensions stay whole."""
    i = -1
    for ext in _EXTENSIONS_MULTIPLE:
        i = path.rfind(ext)
        if i != -1:
            break
    if i == -1:
        return os.path.splitext(path)
    return path[:i], path[i:]


def combine_paths(base_path: Optional[str], suffix: str, sep: str = "_", keep_ext: bool = False) -> str:
    """``libmag.combine_paths`` (libmag.py:331-369) without the ``ext`` / ``check_dir`` options."""
    if not base_path:
        return suffix
    if not os.path.basename(base_path):
        return os.path.join(base_path, suffix)
    return (base_path if keep_ext else splitext(base_path)[0]) + sep + suffix


def filename_to_base(filename: str, series: Optional[int] = None, modifier: str = "",
                     keep_ext: bool = False) -> str:
    path = filename if keep_ext else splitext(filename)[0]
    if modifier:
        path = combine_paths(path, modifier, keep_ext=True)
    return path


def make_filenames(filename: str, series: Optional[int] = None, modifier: str = "",
                   keep_ext: bool = False) -> Tuple[str, str]:
    """``(path of the image5d array, path of its metadata)`` (importer.py:272-301)."""
    base = filename_to_base(filename, series, modifier, keep_ext)
    return (combine_paths(base, SUFFIX_IMAGE5D, keep_ext=True),
            combine_paths(base, SUFFIX_META, keep_ext=True))


def _primitives(val):
    if isinstance(val, dict):
        return {k: _primitives(v) for k, v in val.items()}
    if isinstance(val, (list, tuple, np.ndarray)):
        return [_primitives(v) for v in val]
    try:
        return val.item()
    except AttributeError:
        return val


def save_image_info(filename_info, names, sizes, resolutions, magnification, zoom, near_min, near_max,
                    scaling=None, plane=None) -> Dict[str, Any]:
    """Write the metadata YAML exactly as the reference does (importer.py:482-522,
    yaml_io.py:94-143 with ``use_primitives=True``)."""
    data = _primitives({
        "ver": IMAGE5D_NP_VER, "names": names, "sizes": sizes, "resolutions": resolutions,
        "magnification": magnification, "zoom": zoom, "near_min": near_min, "near_max": near_max,
        "scaling": scaling, "plane": plane})
    with open(filename_info, "w") as f:
        yaml.dump(data, f)
    return data


def load_metadata(path: str, check_ver: bool = False, img5d: Optional[Image5d] = None):
    """``(metadata dict | None, version)``; YAML first, the pre-v1.4 ``.npz`` as a fallback
    (importer.py:602-664)."""
    ver = -1
    try:
        with open(path) as f:
            docs = [d for d in yaml.load_all(f, Loader=yaml.FullLoader) if d]
        output = docs[0] if docs else None
    except FileNotFoundError:
        path_npz = f"{os.path.splitext(path)[0]}.npz"
        try:
            with np.load(path_npz, allow_pickle=True) as arc:
                output = {k: (v.item() if v.ndim == 0 else v) for k, v in arc.items()}
        except FileNotFoundError:
            _logger.warning("Could not load metadata file '%s', skipping", path_npz)
            return None, ver
    if output is None:
        return None, ver
    try:
        ver = output["ver"]
    except KeyError:
        pass
    if img5d is not None and (not check_ver or ver >= IMAGE5D_NP_VER):
        assign_metadata(img5d, output)
    return output, ver


def assign_metadata(img5d: Image5d, md: Dict[str, Any]) -> None:
    """Push the metadata into the module globals the path reads (importer.py:667-745)."""
    if "sizes" in md:
        img5d.shapes = md["sizes"]
    if "resolutions" in md:
        config.resolutions = np.array(md["resolutions"])
    if "magnification" in md:
        config.magnification = md["magnification"]
    if "zoom" in md:
        config.zoom = md["zoom"]
    if "near_min" in md:
        config.near_min = md["near_min"]
    if "near_max" in md:
        config.near_max = md["near_max"]


def read_file(filename: str, series: Optional[int] = None, offset=None, size=None,
              update_info: bool = True) -> Image5d:
    """Open ``<base>_image5d.npy`` memory-mapped with its metadata (importer.py:748-829).
    A missing image leaves ``img5d.img`` as ``None`` like the reference (the caller raises)."""
    if offset is not None or size is not None:
        raise NotImplementedError("loading only an ROI of the image stays in the reference")
    if series is None:
        series = 0
    path_img, path_meta = make_filenames(filename, series)
    img5d = Image5d(None, path_img, path_meta, "np")
    try:
        md, ver = load_metadata(path_meta, update_info, img5d)
        img5d.meta = md
        if md is not None and update_info and -1 < ver < IMAGE5D_NP_VER:
            raise NotImplementedError(
                f"image5d metadata version {ver} < {IMAGE5D_NP_VER}: upgrade it with the reference")
        img5d.img = np.load(path_img, mmap_mode="r")
    except OSError as err:
        _logger.warning("Could not load image files for %s: %s", filename, err)
    return img5d
