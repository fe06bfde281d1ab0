"""ROI (blob detection) profile: the keys and named modifiers the detection path reads.

Mirror of ``magmap.settings.roi_prof.ROIProfile`` restricted to detection: defaults for
the blob detector, block processing and preprocessing keys (reference
magmap/settings/roi_prof.py:69-142) and the named modifiers that change them (:147-334).
Visualisation-only keys of the reference are not carried.  ``profiles/roi_blobs.yaml`` of
the reference loads unchanged through :meth:`SettingsDict.add_profiles`.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

from . import profiles


class ROIProfile(profiles.SettingsDict):
    PATH_PREFIX = "roi"
    #: preprocessing keys (reference roi_prof.py:24-34)
    BLOB_PREPROCESSING = ("clip_vmin", "clip_vmax", "clip_min", "clip_max", "max_thresh_factor",
                          "tot_var_denoise", "unsharp_strength", "erosion_threshold",
                          "adapt_hist_lim")
    #: keys that must agree for channels to share one set of blocks (reference :35-41)
    BLOCK_SIZES = ("segment_size", "denoise_size", "prune_tol_factor", "sub_stack_max_pixels",
                   "isotropic")

    def __init__(self, *args, **kwargs):
        super().__init__()
        self.update({
            # preprocessing before detection (used once SURVEY.md section 8f row 1 lands)
            "clip_vmin": 5, "clip_vmax": 99.5, "clip_min": 0.2, "clip_max": 1.0,
            "max_thresh_factor": 0.5, "tot_var_denoise": None, "unsharp_strength": 0.3,
            "erosion_threshold": 0.2, "adapt_hist_lim": 0.1,
            # 3-D blob detection
            "min_sigma_factor": 3, "max_sigma_factor": 5, "num_sigma": 10,
            "detection_threshold": 0.1, "overlap": 0.5,
            "thresholding": None, "thresholding_size": -1,
            "exclude_border": None,
            # block processing
            "mp_start": "fork", "mp_max_tasks": None,
            "segment_size": 500, "denoise_size": 25,
            "prune_tol_factor": (1, 1, 1), "verify_tol_factor": (1, 1, 1),
            "sub_stack_max_pixels": (1000, 1000, 1000),
            "isotropic": None, "isotropic_vis": (1, 1, 1), "resize_blobs": None,
        })
        #: ``{channel: {channel_to_subtract: factor}}`` (attribute, not key, as in the reference)
        self.spectral_unmixing: Optional[Dict[int, Tuple[int, float]]] = None
        self.update(*args, **kwargs)
        self.profiles = {
            "lightsheet": {
                "clip_vmax": 98.5, "clip_min": 0, "clip_max": 0.5, "unsharp_strength": 0.3,
                "erosion_threshold": 0.3, "min_sigma_factor": 2.6, "max_sigma_factor": 2.8,
                "num_sigma": 10, "overlap": 0.55, "segment_size": 150,
                "prune_tol_factor": (1, 0.9, 0.9), "verify_tol_factor": (3, 1.2, 1.2),
                "isotropic": (0.96, 1, 1), "isotropic_vis": (0.5, 1, 1),
                "sub_stack_max_pixels": (1200, 800, 800), "exclude_border": (1, 0, 0),
            },
            "minpreproc": {
                "clip_vmin": 0, "clip_vmax": 99.99, "clip_max": 1, "tot_var_denoise": 0.01,
                "unsharp_strength": 0, "erosion_threshold": 0,
            },
            "lowres": {
                "min_sigma_factor": 10, "max_sigma_factor": 14, "isotropic": None,
                "denoise_size": 2000, "segment_size": 1000, "max_thresh_factor": 1.5,
                "exclude_border": (8, 1, 1), "verify_tol_factor": (3, 2, 2),
            },
            "2p20x": {
                "clip_vmax": 97, "clip_min": 0, "clip_max": 0.7, "tot_var_denoise": True,
                "unsharp_strength": 2.5, "min_sigma_factor": 2.6, "max_sigma_factor": 4,
                "num_sigma": 20, "overlap": 0.1, "thresholding": None, "thresholding_size": 64,
                "denoise_size": 25, "segment_size": 100, "prune_tol_factor": (1.5, 1.3, 1.3),
            },
            "zebrafish": {"min_sigma_factor": 2.5, "max_sigma_factor": 3},
            "cytoplasm": {
                "clip_min": 0.3, "clip_max": 0.8, "min_sigma_factor": 4, "max_sigma_factor": 10,
                "num_sigma": 10, "overlap": 0.2,
            },
            "isotropic": {"isotropic_vis": (1, 1, 1)},
            "binary": {"denoise_size": None, "detection_threshold": 0.001},
            "4xnuc": {"min_sigma_factor": 3, "max_sigma_factor": 4},
            "20x": {"segment_size": 50},
            "exportdl": {"isotropic": (0.93, 1, 1)},
            "downiso": {"isotropic": None, "resize_blobs": (.2, 1, 1)},
            "register": {"unsharp_strength": 1.5},
            "atlas": {"clip_vmax": 97},
            "spawn": {"mp_start": "spawn"},
        }
