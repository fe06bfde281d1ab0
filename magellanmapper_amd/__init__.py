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
__version__ = "0.1.0"
