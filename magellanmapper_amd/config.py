"""Process-wide settings the detection path reads (mirror of ``magmap.settings.config``).

Only the handful of module-level globals that the reference's hot path consults are kept
(reference magmap/settings/config.py): ``resolutions`` (:246), ``cpus`` (:79), ``channel``
(:144), ``filename`` (:132), ``roi_profile`` / ``roi_profiles`` / :func:`get_roi_profile`
(:882-901), ``SUFFIX_BLOBS`` (:126), ``save_subimg`` (:508), ``verbose`` (:108),
``grid_search_profile`` (:905), ``truth_db_mode`` (:539), ``near_max`` (:211).  They are read at every call --
never cached -- because the reference's grid search mutates profiles between calls
(magmap/stats/mlearn.py:31-).
"""
from __future__ import annotations

import logging
from typing import List, Optional, Sequence

logger = logging.getLogger("magellanmapper_amd")

#: number of worker processes in the reference; here only reported, the GPU does the work
cpus: Optional[int] = None
verbose: bool = False
SUFFIX_SUBIMG = "subimg.npy"
SUFFIX_BLOBS = "blobs.npz"
#: image path, used for the archive's ``basename``
filename: Optional[str] = None
#: channels to process; ``None`` = all
channel: Optional[Sequence[int]] = None
#: ``[[z, y, x], ...]`` physical voxel sizes; the first row is the one used
resolutions = None
#: per-channel near-maximum intensities from the image metadata; floors the stretch ceiling of
#: ``saturate_roi`` (reference config.py:211, plot_3d.py:95-99)
near_max = [-1.0]
near_min = [0.0]
magnification = 1.0
zoom = 1.0
save_subimg: bool = False
truth_db_mode = None
grid_search_profile = None

#: default (channel 0) profile and the per-channel list
roi_profile = None
roi_profiles: List = []


def get_roi_profile(i: int):
    """Profile of channel ``i``; channels beyond the list share the default profile."""
    if len(roi_profiles) > i:
        return roi_profiles[i]
    return roi_profile


def setup_roi_profiles(names: Optional[Sequence[str]] = None):
    """Rebuild the profile list: one profile per channel entry, each a comma-layered stack
    of named modifiers and/or YAML files (reference magmap/io/cli.py:1004-1038)."""
    from . import roi_prof
    global roi_profile, roi_profiles
    roi_profile = roi_prof.ROIProfile()
    roi_profiles = [roi_profile]
    for i, name in enumerate(names or []):
        prof = roi_profile if i == 0 else roi_prof.ROIProfile()
        if i > 0:
            roi_profiles.append(prof)
        prof.add_profiles(name)
    return roi_profiles
