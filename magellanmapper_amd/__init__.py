"""MI355X-native whole-volume nuclei (blob) detection with MagellanMapper's interface.

Modules mirror the reference's names for this one path:

* :mod:`.detector`     <-> ``magmap.cv.detector`` (``detect_blobs``, ``Blobs``, ...)
* :mod:`.stack_detect` <-> ``magmap.cv.stack_detect`` (``StackDetector``, ``setup_blocks``, ...)
* :mod:`.chunking`     <-> ``magmap.cv.chunking`` (``stack_splitter``, ``merge_blobs``)
* :mod:`.config`, :mod:`.profiles`, :mod:`.roi_prof` <-> ``magmap.settings.*``
* :mod:`.blob_log`     replaces the ``skimage.feature.blob_log`` call with HIP kernels
* :mod:`.dist`         block sharding over GPUs + RCCL table gather (new)

The compute path is ``libmmx_hip.so`` (C ABI in ``include/mmx.h``); there is no CPU fallback.
"""
import os as _os

__version__ = "0.1.0"

# The batched detection keeps up to seven HIP streams busy (LoG passes, preprocessing, voxel copy, NMS / re-score tail,
# host-side pruning helpers, and the z-slab upload of a host volume or of the NEXT tile).  ROCm maps streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4) round-robin: with the default the upload stream shares a queue with a
# kernel stream, and the kernels queue behind 150 ms of copies (measured: 302 vs 254 ms per tile of a streamed stack).
# Read when the HIP runtime starts, so it is set here, before anything touches the GPU; a caller's own value wins.
# This only works while the runtime has NOT started: a host application that touched the GPU before importing this
# package keeps the queues it started with (INTEGRATION.md section 4 says what to export instead), and the first
# streamed upload warns once when that -- or a smaller value of the caller's -- is the case (volume._check_hw_queues).
import sys as _sys

_torch = _sys.modules.get("torch")
#: True when torch had already initialised the GPU when this package was imported: the setting below came too late
GPU_WAS_INITIALISED_AT_IMPORT = bool(_torch is not None and getattr(_torch, "cuda", None) is not None
                                     and _torch.cuda.is_initialized())
del _torch
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
