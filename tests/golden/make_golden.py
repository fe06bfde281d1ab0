#!/opt/conda/bin/python3.9
"""Generate golden vectors from the REAL reference + REAL scikit-image.

Run in the build container only (it needs ``/root/reference`` and the conda
interpreter that has scikit-image 0.18.3)::

    /opt/conda/bin/python3.9 tests/golden/make_golden.py

It makes a scratch, importable copy of ``/root/reference/magmap`` under ``/tmp``
(two files need ``from __future__ import annotations`` to import on Python 3.9),
imports the real ``magmap.cv.{detector,stack_detect,chunking}`` and the real
``skimage.feature.blob_log``, runs them on small seeded inputs and stores
**inputs and outputs only** (``.npz`` data) next to this script.  No reference
source is copied into the repository.

Fixtures (all ``np.savez_compressed``):

* ``bloblog_<case>.npz``   -- volume, parameters, sigma ladder, ordered raw peaks
  with their float64 LoG values, pruned ``(z,y,x,sigma)``, a LoG cube crop.
* ``detect_<case>.npz``    -- ``magmap.cv.detector.detect_blobs`` 11-column table.
* ``blocks.npz``           -- ``setup_blocks`` / ``stack_splitter`` parameter sweep.
* ``stack_<case>.npz``     -- ``detect_blobs_blocks``: per-block tables, final table.
* ``prune.npz``            -- ``remove_close_blobs`` / ``StackPruner.prune_blobs_mp``
  on hand-made tables (multi-matches, half-even averages, >1000-row chunking).
* ``preproc_f64.npz``      -- the same two functions on FLOAT64 sub-blocks (values in [0, 1], negative and
  fractional values, ties, a constant tile) + ``stack_denoise_f64.npz`` (a float64 stack end to end)
* ``preproc.npz``          -- ``plot_3d.saturate_roi`` / ``plot_3d.denoise_roi`` on denoise
  sub-blocks (sparse, dense/eroded, constant, ragged, x-size-3, uint8, near_max, 2 channels).
* ``stack_denoise*.npz``   -- ``detect_blobs_blocks`` with the profile's ``denoise_size`` on.
"""
import io
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SCRATCH = "/tmp/mmx_refcopy"


def _bootstrap():
    if os.path.isdir(SCRATCH):
        shutil.rmtree(SCRATCH)
    os.makedirs(SCRATCH)
    shutil.copytree("/root/reference/magmap", os.path.join(SCRATCH, "magmap"))
    for rel in ("magmap/io/np_io.py", "magmap/io/importer.py"):
        path = os.path.join(SCRATCH, rel)
        with open(path) as f:
            src = f.read()
        with open(path, "w") as f:
            f.write("from __future__ import annotations\n" + src)
    sys.path.insert(0, SCRATCH)


_bootstrap()

import contextlib  # noqa: E402
import warnings  # noqa: E402

import numpy as np  # noqa: E402

warnings.filterwarnings("ignore")

import skimage  # noqa: E402
import scipy  # noqa: E402
from skimage.feature import blob as ski_blob  # noqa: E402
from skimage.feature import blob_log, peak_local_max  # noqa: E402
from skimage.util import img_as_float  # noqa: E402
from scipy.ndimage import gaussian_laplace  # noqa: E402

from magmap.settings import config  # noqa: E402
from magmap.io import cli, np_io  # noqa: E402
from magmap.cv import chunking, detector, stack_detect  # noqa: E402

VERSIONS = dict(skimage=skimage.__version__, scipy=scipy.__version__, numpy=np.__version__,
                python=sys.version.split()[0])


def make_volume(seed, shape, n_blobs, dtype=np.uint16, amp=40000.0, blob_sigma=3.0,
                bg_mean=500.0, bg_sd=50.0, margin=6, centres=None):
    """Seeded Gaussian-blob volume (SURVEY.md section 8d generator)."""
    rng = np.random.default_rng(seed)
    vol = rng.normal(bg_mean, bg_sd, shape)
    if centres is None:
        lo = np.full(3, margin, dtype=float)
        hi = np.asarray(shape, dtype=float) - margin
        centres = rng.uniform(lo, hi, (n_blobs, 3))
    zz, yy, xx = np.meshgrid(*[np.arange(s, dtype=float) for s in shape], indexing="ij")
    for c in centres:
        d2 = (zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2
        np.maximum(vol, amp * np.exp(-d2 / (2 * blob_sigma ** 2)), out=vol)
    vol = np.clip(vol, 0, 65535)
    if dtype == np.uint16:
        return vol.astype(np.uint16)
    if dtype == np.uint8:
        return (vol / 257.0).astype(np.uint8)
    if dtype in (np.float32, np.float64):
        return (vol / 65535.0).astype(dtype)
    raise ValueError(dtype)


def make_twoscale_volume(seed, shape, n_pairs, small=(30000.0, 1.0), big=(20000.0, 5.0), off=2.5):
    """Small bright blobs riding on large dim ones: two scale-space maxima per site, so that
    the sphere-overlap prune (skimage blob.py:146-187) actually removes rows."""
    rng = np.random.default_rng(seed)
    vol = rng.normal(500, 50, shape)
    zz, yy, xx = np.meshgrid(*[np.arange(s, dtype=float) for s in shape], indexing="ij")
    for _ in range(n_pairs):
        c = rng.uniform(9, np.asarray(shape, dtype=float) - 9)
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        for centre, (amp, bs) in ((c, big), (c + d * off, small)):
            d2 = (zz - centre[0]) ** 2 + (yy - centre[1]) ** 2 + (xx - centre[2]) ** 2
            vol += amp * np.exp(-d2 / (2 * bs ** 2))
    return np.clip(vol, 0, 65535).astype(np.uint16)


def quiet(fn, *args, **kwargs):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        return fn(*args, **kwargs)


def setup_profile(names=None, **over):
    """Real profile set-up through the reference's own CLI helper."""
    names = names or ["/root/reference/profiles/roi_blobs.yaml"]
    quiet(cli.setup_roi_profiles, names)
    for i, prof in enumerate(config.roi_profiles):
        prof["denoise_size"] = None
        for k, v in over.items():
            if isinstance(v, dict) and "per_channel" in v:
                prof[k] = v["per_channel"][i]
            else:
                prof[k] = v
    return config.roi_profiles


def skimage_stages(image, min_sigma, max_sigma, num_sigma, threshold):
    """Raw (ordered) peaks and cube through scikit-image's own functions (blob.py:470-512)."""
    image_f = img_as_float(image)
    max_s = np.full(image_f.ndim, max_sigma, dtype=float)
    min_s = np.full(image_f.ndim, min_sigma, dtype=float)
    scale = np.linspace(0, 1, num_sigma)[:, np.newaxis]
    sig = scale * (max_s - min_s) + min_s
    gl = [-gaussian_laplace(image_f, s) * np.mean(s) ** 2 for s in sig]
    cube = np.stack(gl, axis=-1)
    lm = peak_local_max(cube, threshold_abs=threshold, footprint=np.ones((3,) * 4),
                        threshold_rel=0.0, exclude_border=(0,) * 4)
    return sig, cube, lm


def bloblog_case(name, vol, min_sigma, max_sigma, num_sigma, threshold=0.1, overlap=0.5):
    res = blob_log(vol, min_sigma=min_sigma, max_sigma=max_sigma, num_sigma=num_sigma,
                   threshold=threshold, overlap=overlap)
    sig, cube, lm = skimage_stages(vol, min_sigma, max_sigma, num_sigma, threshold)
    vals = cube[tuple(lm.T)] if lm.size else np.empty(0)
    crop = tuple(slice(s // 4, s // 4 + min(16, s)) for s in vol.shape)
    out = dict(volume=vol, min_sigma=min_sigma, max_sigma=max_sigma, num_sigma=num_sigma,
               threshold=threshold, overlap=overlap, sigmas=sig, peaks=lm, peak_values=vals,
               pruned=res, cube_crop=cube[crop], cube_crop_origin=np.array([c.start for c in crop]),
               cube_dtype=str(cube.dtype), versions=repr(VERSIONS))
    np.savez_compressed(os.path.join(HERE, "bloblog_%s.npz" % name), **out)
    print("bloblog_%s: %d raw peaks -> %d blobs, cube %s" % (name, len(lm), len(res), cube.dtype))


def detect_case(name, roi, channel, exclude_border=None, resolutions=((1., 1., 1.),),
                names=None, unmix=None, **over):
    config.resolutions = np.array(resolutions)
    setup_profile(names, **over)
    if unmix is not None:            # ROIProfile attribute: {channel: {channel_to_subtract: factor}}
        for prof in config.roi_profiles:
            prof.spectral_unmixing = unmix
    table = quiet(detector.detect_blobs, roi, channel, exclude_border)
    for prof in config.roi_profiles:
        prof.spectral_unmixing = None
    profs = [{k: config.get_roi_profile(i)[k] for k in (
        "min_sigma_factor", "max_sigma_factor", "num_sigma", "detection_threshold", "overlap", "isotropic")}
        for i in range(2)]
    np.savez_compressed(
        os.path.join(HERE, "detect_%s.npz" % name), roi=roi,
        channel=np.array(-1 if channel is None else channel),
        exclude_border=np.array(-1 if exclude_border is None else exclude_border),
        resolutions=np.array(resolutions), profiles=repr(profs), unmix=repr(unmix),
        table=np.empty((0, 11)) if table is None else table, is_none=table is None,
        versions=repr(VERSIONS))
    print("detect_%s: %s" % (name, None if table is None else table.shape))


def blocks_cases():
    sweep = []
    cases = [
        # shape, resolutions, segment_size, exclude_border, prune_tol_factor, denoise_size
        ((64, 96, 96), (1., 1., 1.), 40, None, (1, 1, 1), None),
        ((51, 200, 200), (6.6, 1.1, 1.1), 150, (1, 0, 0), (1, 0.9, 0.9), 25),
        ((1024, 2048, 2048), (1., 1., 1.), 256, None, (1, 1, 1), None),
        ((1024, 2048, 2048), (1., 1., 1.), 500, None, (1, 1, 1), 25),
        ((100, 100, 100), (2.0, 0.5, 0.5), 30, (8, 1, 1), (1.5, 1.3, 1.3), 2000),
        ((7, 13, 300), (1., 1., 1.), 50, (0, 3, 4), (1, 1, 1), None),
        ((33, 47, 59), (0.7, 0.33, 0.33), 12.5, None, (1, 0.9, 0.9), None),
    ]
    out = {}
    for i, (shape, res, seg, excl, ptf, dn) in enumerate(cases):
        config.resolutions = np.array([res])
        settings = setup_profile()[0]
        settings["segment_size"] = seg
        settings["exclude_border"] = excl
        settings["prune_tol_factor"] = ptf
        settings["denoise_size"] = dn
        bl = quiet(stack_detect.setup_blocks, settings, shape)
        grid = bl.sub_roi_slices.shape
        sl = np.array([[[s.start, s.stop] for s in bl.sub_roi_slices[c]]
                       for c in np.ndindex(*grid)]).reshape(grid + (3, 2))
        pre = "c%d_" % i
        out[pre + "shape"] = np.array(shape)
        out[pre + "resolutions"] = np.array(res)
        out[pre + "segment_size"] = np.array(seg)
        out[pre + "exclude_border"] = np.array(-1 if excl is None else excl)
        out[pre + "prune_tol_factor"] = np.array(ptf)
        out[pre + "denoise_size"] = np.array(-1 if dn is None else dn)
        out[pre + "slices"] = sl
        out[pre + "offsets"] = bl.sub_rois_offsets
        out[pre + "denoise_max_shape"] = np.array(
            -1 if bl.denoise_max_shape is None else bl.denoise_max_shape)
        out[pre + "tol"] = bl.tol
        out[pre + "overlap_base"] = bl.overlap_base
        out[pre + "overlap"] = bl.overlap
        out[pre + "overlap_padding"] = bl.overlap_padding
        out[pre + "max_pixels"] = bl.max_pixels
        sweep.append(i)
    out["n_cases"] = np.array(len(sweep))
    # stack_splitter alone, the reference's own unit-test geometry (test_chunking.py:47-66)
    config.resolutions = [[6.6, 1.1, 1.1]]
    ov = detector.calc_overlap(2)
    out["calc_overlap_2"] = ov
    for j, overlap in enumerate([np.array((0, 1, 1)), np.array((0, 1, 2)), np.array((1, 1, 2)), ov]):
        sl, off = chunking.stack_splitter((5, 4, 4), [1, 3, 3], overlap)
        grid = sl.shape
        out["ss%d_overlap" % j] = overlap
        out["ss%d_slices" % j] = np.array(
            [[[s.start, s.stop] for s in sl[c]] for c in np.ndindex(*grid)]).reshape(grid + (3, 2))
        out["ss%d_offsets" % j] = off
    out["versions"] = repr(VERSIONS)
    np.savez_compressed(os.path.join(HERE, "blocks.npz"), **out)
    print("blocks: %d cases" % len(sweep))


def stack_case(name, roi, channels=None, resolutions=((1., 1., 1.),), cpus=4, near_max=(-1.0,),
               coloc=False, **over):
    config.resolutions = np.array(resolutions)
    config.near_max = list(near_max)
    config.filename = "golden"
    config.cpus = cpus
    setup_profile(**over)
    quiet(chunking.set_mp_start_method)
    img5d = np_io.Image5d(roi[None])
    settings = config.get_roi_profile(0 if channels is None else channels[0])
    # per-block tables before pruning, through the real StackDetector
    chls = channels
    if chls is None:
        chls = list(range(roi.shape[3])) if roi.ndim > 3 else [0]
    bl = quiet(stack_detect.setup_blocks, settings, roi.shape)
    seg_rois = quiet(stack_detect.StackDetector.detect_blobs_sub_rois,
                     img5d, roi, bl.sub_roi_slices, bl.sub_rois_offsets,
                     bl.denoise_max_shape, bl.exclude_border, coloc, chls)
    merged = chunking.merge_blobs(seg_rois)
    pruned, df = quiet(stack_detect.StackPruner.prune_blobs_mp,
                       roi, seg_rois, bl.overlap, bl.tol, bl.sub_roi_slices,
                       bl.sub_rois_offsets, chls, bl.overlap_padding)
    # and the public entry point end to end
    _, _, blobs = quiet(stack_detect.detect_blobs_blocks, "golden", img5d, None, None,
                        channels, False, False, True, coloc)
    final = blobs.blobs
    out = dict(roi=roi, channels=np.array(-1 if channels is None else channels),
               resolutions=np.array(resolutions),
               grid=np.array(seg_rois.shape),
               merged=np.empty((0, 14)) if merged is None else merged,
               pruned11=np.empty((0, 11)) if pruned is None else pruned,
               final=np.empty((0, 8)) if final is None else final,
               final_cols=np.array(blobs.cols if blobs.cols is not None else []),
               ratios=np.empty((0, 3)) if df is None or df.empty else df.to_numpy(),
               near_max=np.array(near_max, dtype=float), coloc=np.array(bool(coloc)),
               colocs=(np.empty((0, 0), np.uint8) if blobs.colocalizations is None
                       else blobs.colocalizations),
               overrides=repr(over), versions=repr(VERSIONS))
    config.near_max = [-1.0]
    for c in np.ndindex(*seg_rois.shape):
        t = seg_rois[c]
        out["block_%d_%d_%d" % c] = np.empty((0, 11)) if t is None else t
    np.savez_compressed(os.path.join(HERE, "stack_%s.npz" % name), **out)
    print("stack_%s: grid %s, %s merged -> %s final" % (
        name, seg_rois.shape, None if merged is None else len(merged),
        None if final is None else final.shape))


def preproc_cases():
    """saturate_roi / denoise_roi of the real reference on denoise-sized sub-blocks."""
    from magmap.plot import plot_3d
    out = {}
    names = []

    def case(name, roi, near_max=(-1.0,), names_=None, **over):
        setup_profile(names_, **over)
        config.near_max = list(near_max)
        sat = quiet(plot_3d.saturate_roi, roi, channel=None)
        den = quiet(plot_3d.denoise_roi, sat, channel=None)
        out[name + "_roi"] = roi
        out[name + "_near_max"] = np.array(near_max, dtype=float)
        out[name + "_over"] = repr(over)
        out[name + "_sat"] = sat
        out[name + "_den"] = den
        names.append(name)
        print("preproc %-10s %s %s -> sat %s mean %.4f, den %s [%.4f, %.4f]" % (
            name, roi.shape, roi.dtype, sat.dtype, float(np.mean(sat)), den.dtype,
            float(den.min()), float(den.max())))

    sparse = make_volume(41, (25, 25, 25), 3, margin=4)
    case("sparse", sparse)
    densev = make_volume(42, (25, 25, 25), 40, amp=6000.0, blob_sigma=3.5, bg_mean=3000.0,
                         bg_sd=800.0, margin=0)
    case("dense", densev)
    case("const", np.full((25, 25, 25), 1234, dtype=np.uint16))
    case("ragged", make_volume(43, (14, 25, 9), 2, margin=3))
    case("x3", make_volume(44, (25, 25, 3), 2, margin=1))
    case("x3dense", make_volume(45, (12, 20, 3), 6, amp=5000.0, bg_mean=3000.0, bg_sd=900.0,
                                margin=0))
    case("tiny", make_volume(46, (2, 1, 5), 0, margin=0))
    case("u8", make_volume(47, (25, 25, 25), 4, dtype=np.uint8, margin=4))
    case("nearmax", sparse, near_max=(90000.0,))
    case("nearmax_lo", sparse, near_max=(3000.0,))
    case("noero", densev, erosion_threshold=0.0)
    case("nounsharp", densev, unsharp_strength=0.0)
    case("clipvals", densev, clip_vmin=2, clip_vmax=90.5, clip_min=0.1, clip_max=0.8,
         unsharp_strength=0.45, erosion_threshold=0.1)
    # discrete data with many ties around the percentile ranks
    rng = np.random.default_rng(48)
    case("ties", (rng.integers(0, 6, (20, 25, 25)) * 100).astype(np.uint16))
    big = make_volume(49, (32, 40, 36), 10, margin=4)   # larger than a stock sub-block, R=32 > every side
    case("big", big)
    two = np.stack((sparse, densev), axis=-1)
    yaml = "/root/reference/profiles/roi_blobs.yaml"
    case("2ch", two, near_max=(-1.0, 20000.0))
    case("2ch_perchl", two, near_max=(-1.0, -1.0), names_=[yaml, yaml + ",4xnuc"],
         unsharp_strength={"per_channel": [0.3, 0.6]})
    config.near_max = [-1.0]
    out["names"] = np.array(names)
    out["versions"] = repr(VERSIONS)
    # the sigma-8 kernel as THIS environment's SciPy/NumPy builds it: np.exp differs by an ulp between
    # NumPy releases, so parity tests feed these weights to the restatement (data, not code)
    from scipy.ndimage import filters as ndi_filters
    out["gauss8_weights"] = ndi_filters._gaussian_kernel1d(8.0, 0, 32)
    np.savez_compressed(os.path.join(HERE, "preproc.npz"), **out)


def make_coloc_volume(seed, shape, n_blobs, n_chl=2, shared=0.5):
    """Channels whose blobs partly coincide: a fraction ``shared`` of channel 0's centres is reused
    (with another amplitude) in every other channel."""
    rng = np.random.default_rng(seed)
    lo, hi = np.full(3, 6.0), np.asarray(shape, dtype=float) - 6
    base = rng.uniform(lo, hi, (n_blobs, 3))
    chls = []
    for c in range(n_chl):
        if c == 0:
            centres = base
        else:
            keep = base[rng.random(n_blobs) < shared]
            own = rng.uniform(lo, hi, (n_blobs - len(keep), 3))
            centres = np.concatenate((keep, own))
        chls.append(make_volume(seed * 10 + c, shape, 0, centres=centres, amp=30000.0 + 5000 * c))
    return np.stack(chls, axis=-1)


def coloc_cases():
    """colocalize_blobs of the real reference on block-sized inputs."""
    from magmap.cv import colocalizer
    out, names = {}, []

    def case(name, roi, blobs, thresh=None, roi_key=None):
        got = quiet(colocalizer.colocalize_blobs, roi, blobs, thresh)
        if roi_key is None:                  # volumes are stored once and shared between cases
            roi_key = name + "_roi"
            out[roi_key] = roi
        out[name + "_roikey"] = np.array(roi_key)
        out[name + "_blobs"] = blobs
        out[name + "_thresh"] = np.array(-1.0 if thresh is None else thresh)
        out[name + "_colocs"] = np.empty((0, 0), np.uint8) if got is None else got
        names.append(name)
        print("coloc %-10s roi %s blobs %s -> %s" % (
            name, roi.shape, None if blobs is None else blobs.shape,
            None if got is None else got.sum(axis=0)))

    config.resolutions = np.array([[1., 1., 1.]])
    setup_profile(num_sigma=3)
    vol = make_coloc_volume(51, (36, 52, 56), 24)
    blobs = quiet(detector.detect_blobs, vol, None)
    case("two", vol, blobs)
    vol3 = make_coloc_volume(52, (32, 48, 48), 20, n_chl=3, shared=0.4)
    blobs3 = quiet(detector.detect_blobs, vol3, None)
    case("three", vol3, blobs3)
    case("sel1", vol3, quiet(detector.detect_blobs, vol3, [1]), roi_key="three_roi")   # one channel's blobs
    # crowded: balls overlap, higher row index owns the shared voxels; a duplicate centre owns nothing
    crowd = blobs.copy()
    crowd[1, :3] = crowd[0, :3] + (0, 1, 1)
    crowd[3, :3] = crowd[2, :3]
    crowd[5, :3] = (0, 0, 0)
    crowd[6, :3] = np.array(vol.shape[:3]) - 1
    case("crowd", vol, crowd, roi_key="two_roi")
    outside = blobs.copy()
    outside[0, 0] = -1
    outside[1, 2] = vol.shape[2]
    case("outside", vol, outside, roi_key="two_roi")
    case("f64", vol.astype(np.float64) / 65535.0 * 1.7 + 0.2, blobs, roi_key="two_roi:f64")
    case("single", vol[..., 0], blobs, roi_key="two_roi:ch0")         # 3-D ROI -> None
    # percentile thresholds (colocalizer.py:403-409): a channel's threshold is a percentile of its intensities over
    # every voxel its own blobs own
    case("two_p5", vol, blobs, 5, roi_key="two_roi")
    case("two_p50", vol, blobs, 50, roi_key="two_roi")
    case("three_p99", vol3, blobs3, 99.5, roi_key="three_roi")
    case("sel1_p20", vol3, quiet(detector.detect_blobs, vol3, [1]), 20, roi_key="three_roi")
    case("crowd_p30", vol, crowd, 30, roi_key="two_roi")
    case("outside_p5", vol, outside, 5, roi_key="two_roi")
    case("f64_p42", vol.astype(np.float64) / 65535.0 * 1.7 + 0.2, blobs, 42.5, roi_key="two_roi:f64")
    out["names"] = np.array(names)
    out["versions"] = repr(VERSIONS)
    np.savez_compressed(os.path.join(HERE, "coloc.npz"), **out)


def prune_cases():
    rng = np.random.default_rng(77)
    out = {}
    quiet(chunking.set_mp_start_method)
    # (1) remove_close_blobs directly: multi-matches, .5 averages, >1000 rows (chunking)
    for k, (n_m, n_c, span, tol) in enumerate([
            (12, 15, 12, (1, 2, 2)), (1500, 1200, 60, (2, 2, 2)), (300, 5, 200, (5, 5, 5)),
            (40, 40, 300, (3, 1, 2))]):
        def table(n):
            t = np.ones((n, 14)) * -1
            t[:, :3] = rng.integers(0, span, (n, 3))
            t[:, 3] = 5.196
            t[:, 6] = 0
            t[:, 7:10] = t[:, :3] + rng.integers(0, 2, (n, 3))
            t[:, 11:] = rng.integers(0, 3, (n, 3))
            return t
        master, check = table(n_m), table(n_c)
        # force a few exact duplicates and odd-sum pairs
        check[: min(4, n_c), :3] = master[: min(4, n_c), :3]
        check[: min(4, n_c), 7:10] = master[: min(4, n_c), 7:10] + np.array([1, 0, 3])
        detector.Blobs(np.ones((1, 11)) * -1).format_blobs()  # make sure 11-col indices are set
        pruned, master_out = detector.remove_close_blobs(check.copy(), master.copy(), np.array(tol))
        out["rc%d_master" % k] = master
        out["rc%d_check" % k] = check
        out["rc%d_tol" % k] = np.array(tol)
        out["rc%d_pruned" % k] = pruned
        out["rc%d_master_out" % k] = master_out
    out["n_rc"] = np.array(4)

    # (2) StackPruner.prune_blobs_mp on hand-made per-block tables
    config.resolutions = np.array([[1., 1., 1.]])
    config.cpus = 2
    settings = setup_profile()[0]
    settings["segment_size"] = 20
    shape = (44, 50, 65)
    bl = quiet(stack_detect.setup_blocks, settings, shape)
    grid = bl.sub_roi_slices.shape
    seg_rois = np.zeros(grid, dtype=object)
    for c in np.ndindex(*grid):
        sl = bl.sub_roi_slices[c]
        lo = np.array([s.start for s in sl])
        hi = np.array([s.stop for s in sl])
        n = int(rng.integers(0, 14))
        if n == 0:
            seg_rois[c] = None
            continue
        t = np.ones((n, 11)) * -1
        t[:, :3] = rng.integers(lo, hi, (n, 3))
        t[:, 3] = 5.196
        t[:, 6] = rng.integers(0, 2, n)
        t[:, 7:10] = t[:, :3]
        seg_rois[c] = t
    # plant cross-block duplicates in the overlaps (same and +-1 shifted positions)
    for c in np.ndindex(*grid):
        for ax in range(3):
            nb = list(c)
            nb[ax] += 1
            if nb[ax] >= grid[ax] or seg_rois[c] is None:
                continue
            nb = tuple(nb)
            sl, sl_nb = bl.sub_roi_slices[c], bl.sub_roi_slices[nb]
            pos = np.array([(max(a.start, b.start) + min(a.stop, b.stop)) // 2
                            for a, b in zip(sl, sl_nb)], dtype=float)
            for shift, chl in (((0, 0, 0), 0), ((1, 0, 1), 1), ((0, 1, 0), 0)):
                row = np.ones(11) * -1
                row[:3] = pos
                row[3] = 5.196
                row[6] = chl
                row[7:10] = pos
                row2 = row.copy()
                row2[:3] += shift
                row2[7:10] += shift
                seg_rois[c] = np.vstack((seg_rois[c], row))
                seg_rois[nb] = row2[None] if seg_rois[nb] is None else np.vstack((seg_rois[nb], row2))
    dummy_img = np.zeros(shape, dtype=np.uint8)
    pruned, df = quiet(stack_detect.StackPruner.prune_blobs_mp,
                       dummy_img, seg_rois, bl.overlap, bl.tol, bl.sub_roi_slices,
                       bl.sub_rois_offsets, [0, 1], bl.overlap_padding)
    out["sp_shape"] = np.array(shape)
    out["sp_grid"] = np.array(grid)
    out["sp_segment_size"] = np.array(20)
    for c in np.ndindex(*grid):
        out["sp_block_%d_%d_%d" % c] = np.empty((0, 11)) if seg_rois[c] is None else seg_rois[c]
    out["sp_pruned"] = pruned
    out["sp_ratios"] = df.to_numpy()
    out["sp_ratio_cols"] = np.array(list(df.columns))
    out["versions"] = repr(VERSIONS)
    np.savez_compressed(os.path.join(HERE, "prune.npz"), **out)
    print("prune: remove_close x4; prune_blobs_mp %s blocks -> %d rows" % (grid, len(pruned)))


def main():
    print("versions:", VERSIONS)
    if sys.argv[1:] == ["preproc"]:       # only the preprocessing fixtures (added later)
        return main_preproc()
    if sys.argv[1:] == ["preproc_f64"]:   # only the float64-image preprocessing fixtures (added later)
        return main_preproc_f64()
    if sys.argv[1:] == ["coloc_cases"]:   # only coloc.npz (percentile cases added later)
        return coloc_cases()
    if sys.argv[1:] == ["coloc"]:         # only the co-localisation fixtures (added later)
        return main_coloc()
    if sys.argv[1:] == ["image5d"]:       # only the on-disk image fixtures (added later)
        return main_image5d()
    if sys.argv[1:] == ["unmix"]:         # only the spectral-unmixing fixtures (added later)
        return main_unmix()
    if sys.argv[1:] == ["isotropic"]:     # only the isotropic-rescale fixtures (added later)
        return main_isotropic()
    if sys.argv[1:] == ["tv"]:            # only the total-variation denoising fixtures (added later)
        return main_tv()
    if sys.argv[1:] == ["overlap"]:       # only the _prune_blobs fixtures (pair-order pinning; added later)
        return main_overlap()
    if sys.argv[1:] == ["match"]:         # only the match-based co-localisation fixtures (added later)
        return main_match()
    if sys.argv[1:] == ["grouping"]:      # only the channel-grouping fixtures of detect_blobs_stack (added later)
        return main_grouping()
    # ---- blob_log arithmetic
    bloblog_case("u16_1sigma", make_volume(11, (40, 56, 60), 22), 3, 3, 1)
    bloblog_case("u16_5sigma", make_volume(12, (48, 64, 72), 30), 3, 5, 5)
    dense = make_volume(13, (32, 48, 48), 0, centres=np.array(
        [[10, 12, 12], [10, 12, 15.5], [12.5, 14, 13], [20, 30, 30], [20, 33, 30], [22, 31.5, 33],
         [20, 30, 36.2], [9, 36, 10], [11.6, 37, 11], [25, 10, 38], [25.4, 13.2, 38]]))
    bloblog_case("u16_dense10", dense, 3, 5, 10)
    bloblog_case("u16_twoscale10", make_twoscale_volume(19, (36, 52, 52), 7), 1.0, 5.5, 10,
                 threshold=0.05)
    bloblog_case("u16_twoscale_b", make_twoscale_volume(20, (40, 48, 56), 9, small=(40000.0, 1.2),
                                                        big=(15000.0, 5.5), off=1.5), 1.0, 5.5, 10,
                 threshold=0.05, overlap=0.3)
    bloblog_case("u8_3sigma", make_volume(14, (36, 40, 44), 14, dtype=np.uint8), 2.5, 3.5, 3)
    bloblog_case("f32_2sigma", make_volume(15, (30, 44, 40), 12, dtype=np.float32), 3, 4, 2)
    bloblog_case("f64_2sigma", make_volume(16, (30, 40, 44), 12, dtype=np.float64), 3, 4, 2)
    bloblog_case("u16_empty", np.full((20, 24, 28), 700, dtype=np.uint16), 3, 5, 5)
    bloblog_case("u16_thin", make_volume(17, (5, 40, 44), 6, margin=2), 3, 4, 3)
    bloblog_case("u16_smallsig", make_volume(18, (24, 32, 32), 20, blob_sigma=1.2), 1.0, 2.0, 3,
                 threshold=0.05)

    # ---- magmap.cv.detector.detect_blobs
    vol1 = make_volume(21, (40, 60, 64), 24)
    detect_case("1ch", vol1, None, num_sigma=5)
    detect_case("1ch_border", vol1, None, exclude_border=np.array([[2, 5, 5], [1, 6, 6]]), num_sigma=5)
    vol2 = np.stack((make_volume(22, (36, 52, 56), 18), make_volume(23, (36, 52, 56), 14)), axis=-1)
    detect_case("2ch", vol2, None, num_sigma=3)
    detect_case("2ch_sel1", vol2, [1], num_sigma=3)
    yaml = "/root/reference/profiles/roi_blobs.yaml"
    detect_case("2ch_perchl", vol2, None, names=[yaml, yaml + ",4xnuc"], num_sigma=3,
                detection_threshold={"per_channel": [0.1, 0.2]})
    # the second profile is only used if one exists per channel
    config.resolutions = np.array([[1., 1., 1.]])
    detect_case("res_x2", vol1, None, resolutions=((2.0, 0.8, 0.8),), num_sigma=3)
    detect_case("empty", np.full((24, 30, 30), 512, dtype=np.uint16), None, num_sigma=3)

    # ---- block geometry
    blocks_cases()

    # ---- whole-stack detection, multi-block
    stack_case("u16_2x3x3", make_volume(31, (64, 96, 96), 60), None, segment_size=40, num_sigma=5)
    stack_case("u16_border", make_volume(32, (60, 80, 84), 45), None, segment_size=36, num_sigma=3,
               exclude_border=(1, 0, 0), prune_tol_factor=(1, 0.9, 0.9))
    vol2s = np.stack((make_volume(33, (50, 70, 70), 30), make_volume(34, (50, 70, 70), 26)), axis=-1)
    stack_case("2ch", vol2s, None, segment_size=32, num_sigma=3)
    stack_case("empty", np.full((40, 50, 50), 300, dtype=np.uint16), None, segment_size=30,
               num_sigma=2)

    # ---- table pruning
    prune_cases()

    main_preproc()
    main_coloc()
    main_image5d()
    main_unmix()
    main_isotropic()
    main_grouping()
    main_match()
    main_tv()


def main_isotropic():
    """cv_nd.make_isotropic itself and detect_blobs / detect_blobs_blocks with the profile's isotropic
    set.  scikit-image 0.18.3 resizes a 3-D array whose LAST axis keeps its length with its 2-D warp
    (x as channels), unlike the release the reference pins; every case here is multichannel or changes
    all three axes, so that 0.18.3 and >= 0.19 run the same SciPy interpolation."""
    from magmap.cv import cv_nd
    out, names = {}, []

    def iso_case(name, roi, scale, res):
        got = quiet(cv_nd.make_isotropic, roi, scale, np.array(res))
        out[name + "_roi"], out[name + "_scale"], out[name + "_res"] = roi, np.array(scale, float), np.array(res, float)
        out[name + "_out"] = got
        names.append(name)
        print("isotropic %-8s %s %s -> %s %s" % (name, roi.shape, roi.dtype, got.shape, got.dtype))

    v1 = make_volume(81, (14, 30, 33), 6)
    iso_case("allaxes", v1, (0.96, 1.07, 1), (2.5, 1.0, 1.2))
    iso_case("down", v1, (1, 1, 1), (1.0, 1.15, 1.21))                 # z kept, y/x up
    v2 = np.stack((make_volume(82, (12, 26, 28), 5), make_volume(83, (12, 26, 28), 4)), axis=-1)
    iso_case("2ch_z", v2, (0.96, 1, 1), (3.0, 1.0, 1.0))                # the stock lightsheet shape: z only
    iso_case("2ch_f64", v2.astype(np.float64) / 40000.0 - 0.1, (1, 1, 1), (2.2, 1.0, 1.0))
    iso_case("2ch_slight", v2, (0.96, 1, 1), (1.0, 1.0, 1.0))           # 4 % down-sampling: anti-aliasing is a no-op
    out["names"] = np.array(names)
    out["versions"] = repr(VERSIONS)
    np.savez_compressed(os.path.join(HERE, "isotropic.npz"), **out)

    vol2 = np.stack((make_volume(84, (16, 44, 48), 9), make_volume(85, (16, 44, 48), 7)), axis=-1)
    detect_case("iso_2ch", vol2, None, resolutions=((3.0, 1.0, 1.0),), num_sigma=3, isotropic=(0.96, 1, 1))
    detect_case("iso_allaxes", make_volume(86, (16, 40, 44), 8), None, resolutions=((2.2, 1.0, 1.1),),
                num_sigma=3, isotropic=(1, 1.06, 1), exclude_border=np.array([[1, 3, 3], [1, 2, 2]]))
    stack_case("iso_2ch", np.stack((make_volume(87, (20, 64, 66), 14), make_volume(88, (20, 64, 66), 11)), axis=-1),
               None, resolutions=((3.0, 1.0, 1.0),), segment_size=28, num_sigma=3, isotropic=(0.96, 1, 1))
    stack_case("iso_denoise", np.stack((make_volume(89, (18, 56, 60), 12), make_volume(90, (18, 56, 60), 9)),
                                       axis=-1),
               None, resolutions=((2.5, 1.0, 1.0),), segment_size=30, num_sigma=3, isotropic=(1, 1, 1),
               denoise_size=20, near_max=(-1.0, -1.0))


def main_unmix():
    """detect_blobs with the profile's spectral_unmixing set: the detected channel becomes an UNSCALED
    float64 image (so thresholds are in raw intensity units)."""
    vol = make_coloc_volume(71, (32, 44, 48), 16, n_chl=3, shared=0.6)
    detect_case("unmix_1sub", vol, [1], unmix={1: {0: 0.4}}, num_sigma=3, detection_threshold=900.0)
    detect_case("unmix_2sub", vol, None, unmix={1: {0: 0.5, 2: 0.25}, 2: {0: 1.5}}, num_sigma=3,
                detection_threshold=900.0)


def main_image5d():
    """A small image written the way the reference's importer writes it: ``sample_image5d.npy`` +
    ``sample_meta.yml`` (importer.save_image_info), and what the real ``importer.read_file`` then
    reports (shape, dtype, the config globals it sets)."""
    from magmap.io import importer
    vol = np.stack((make_volume(61, (20, 36, 40), 8), make_volume(62, (20, 36, 40), 6)), axis=-1)
    image5d = vol[None]
    base = os.path.join(HERE, "sample")
    path_img, path_meta = quiet(importer.make_filenames, base + ".czi")
    with open(path_img, "wb") as f:
        np.save(f, image5d)
    lows, highs = np.percentile(vol.reshape(-1, 2), (0.5, 99.5), axis=0)
    quiet(importer.save_image_info, path_meta, ["sample.czi"], [image5d.shape], [[5.0, 1.2, 1.2]], 5.0, 1.0,
          lows.tolist(), highs.tolist())
    config.resolutions = None
    config.near_max = [-1.0]
    img5d = quiet(importer.read_file, base + ".czi", 0)
    np.savez_compressed(os.path.join(HERE, "image5d_expect.npz"),
                        path_img=np.array(os.path.basename(img5d.path_img)),
                        path_meta=np.array(os.path.basename(img5d.path_meta)),
                        shape=np.array(img5d.img.shape), dtype=np.array(str(img5d.img.dtype)),
                        is_memmap=np.array(isinstance(img5d.img, np.memmap)),
                        resolutions=np.array(config.resolutions), near_max=np.array(config.near_max),
                        near_min=np.array(config.near_min), magnification=np.array(config.magnification),
                        zoom=np.array(config.zoom), meta_keys=np.array(sorted(str(k) for k in img5d.meta
                                                                              if isinstance(k, str))),
                        versions=repr(VERSIONS))
    config.near_max = [-1.0]
    print("image5d: %s %s, resolutions %s, near_max %s" % (img5d.img.shape, img5d.img.dtype,
                                                          config.resolutions, highs))


def grouping_case(name, roi, **over):
    """``stack_detect.detect_blobs_stack`` end to end (channel grouping by ROIProfile.BLOCK_SIZES, :554-561, one
    detect_blobs_blocks per group, combine_arrs): the final table, the archive it saved, the grouping decision."""
    import tempfile
    from magmap.settings import roi_prof
    config.resolutions = np.array([[1., 1., 1.]])
    config.near_max = [-1.0] * roi.shape[3]
    config.cpus = 4
    config.channel = None
    yaml = "/root/reference/profiles/roi_blobs.yaml"
    setup_profile(names=[yaml, yaml], **over)          # one profile per channel
    quiet(chunking.set_mp_start_method)
    img5d = np_io.Image5d(roi[None])
    img5d.is_roi = True
    chls = list(range(roi.shape[3]))
    identical = roi_prof.ROIProfile.is_identical_settings(
        [config.get_roi_profile(c) for c in chls], roi_prof.ROIProfile.BLOCK_SIZES)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        config.filename = os.path.join(tmp, "grp")
        try:
            _, _, blobs = quiet(stack_detect.detect_blobs_stack, os.path.join(tmp, "grp"), img5d)
            arch = np.load(os.path.join(tmp, "grp_blobs.npz"))
            arch_segments = arch["segments"]
        finally:
            os.chdir(cwd)
    grids = np.array([quiet(stack_detect.setup_blocks, config.get_roi_profile(c), roi.shape[:3]).sub_roi_slices.shape
                      for c in chls])
    np.savez_compressed(os.path.join(HERE, "grouping_%s.npz" % name), roi=roi, identical=np.array(bool(identical)),
                        final=blobs.blobs, archive_segments=arch_segments, grids=grids,
                        overrides=repr(over), versions=repr(VERSIONS))
    print("grouping_%s: identical block sizes %s, grids %s, %d blobs" % (name, identical, grids.tolist(), len(blobs.blobs)))


def main_grouping():
    vol = np.stack((make_volume(61, (48, 72, 76), 30), make_volume(62, (48, 72, 76), 26)), axis=-1)
    grouping_case("equal", vol, segment_size=40, num_sigma=3)
    grouping_case("unequal", vol, segment_size={"per_channel": [40, 30]}, num_sigma=3)
    grouping_case("unequal_tol", vol, segment_size=36, prune_tol_factor={"per_channel": [(1, 1, 1), (1, 0.9, 0.9)]},
                  num_sigma=3)


def make_blob_table(seed, shape, n, n_chl=2, shared=0.6, jitter=2):
    """An 8-column final blob table (z, y, x, radius, confirmed, truth, channel, region) as detection leaves it:
    ``n`` integer centres per channel, a ``shared`` fraction of every later channel's blobs sitting within
    ``jitter`` voxels of a channel-0 blob (so that distances repeat: sqrt of small integers -- ties for the
    assignment solver), the rest on their own."""
    rng = np.random.default_rng(seed)
    base = rng.integers(0, shape, (n, 3))
    rows = []
    for c in range(n_chl):
        if c == 0:
            pts = base
        else:
            k = int(shared * n)
            near = base[rng.permutation(n)[:k]] + rng.integers(-jitter, jitter + 1, (k, 3))
            pts = np.concatenate((near, rng.integers(0, shape, (n - k, 3))))
            pts = np.clip(pts, 0, np.subtract(shape, 1))
        t = np.full((len(pts), 8), -1.0)
        t[:, :3] = pts
        t[:, 3] = 5.196152422706632
        t[:, 6] = c
        rows.append(t)
    return np.concatenate(rows)


def match_cases():
    """verifier.find_closest_blobs_cdist (cdist + scipy.optimize.linear_sum_assignment + threshold),
    verifier.match_blobs_roi, colocalizer.colocalize_blobs_match and StackColocalizer.colocalize_stack."""
    from magmap.cv import colocalizer, verifier
    out = {}
    rng = np.random.default_rng(71)
    # ---- the assignment itself: rectangular both ways, integer coordinates (tied distances), scaling, threshold
    specs = [(5, 5, 6), (12, 7, 8), (7, 12, 8), (40, 40, 12), (60, 35, 10), (1, 9, 5), (9, 1, 5), (30, 30, 3),
             (80, 100, 20)]
    for k, (n, m, span) in enumerate(specs):
        a = rng.integers(0, span, (n, 4)).astype(float)
        b = rng.integers(0, span, (m, 4)).astype(float)
        scaling = np.array([1.0, 1.0, 1.0]) if k % 2 == 0 else np.array([5.0 / 3.0, 1.0, 1.0])
        thresh = None if k == 3 else 3.0 + (k % 3)
        rowis, colis, dists = verifier.find_closest_blobs_cdist(a, b, thresh, scaling)
        out.update({"lsap%d_a" % k: a, "lsap%d_b" % k: b, "lsap%d_scaling" % k: scaling,
                    "lsap%d_thresh" % k: np.array(np.nan if thresh is None else thresh),
                    "lsap%d_rows" % k: rowis, "lsap%d_cols" % k: colis, "lsap%d_dists" % k: dists})
    out["n_lsap"] = np.array(len(specs))
    # ---- one ROI: inner / outer matching of two channels' blobs
    config.resolutions = np.array([[1., 1., 1.]])
    setup_profile(segment_size=40, num_sigma=3)
    table = make_blob_table(72, (40, 60, 64), 90, n_chl=3)
    blobs = detector.Blobs(table.copy())
    tol = np.array([5.0, 5.0, 5.0])
    for k, (offset, size) in enumerate((((0, 0, 0), (64, 60, 40)), ((10, 8, 4), (40, 44, 30)), ((60, 50, 30), (4, 10, 10)))):
        matches = quiet(colocalizer.colocalize_blobs_match, blobs, offset, size, tol)
        out["roi%d_offset" % k] = np.array(offset)
        out["roi%d_size" % k] = np.array(size)
        out["roi%d_keys" % k] = np.array(sorted(matches.keys())).reshape(-1, 2)
        for key, bm in matches.items():
            df = bm.df
            nrow = 0 if df is None else len(df)
            out["roi%d_%d_%d_blob1" % (k, *key)] = np.vstack(df["Blob1"]) if nrow else np.empty((0, 8))
            out["roi%d_%d_%d_blob2" % (k, *key)] = np.vstack(df["Blob2"]) if nrow else np.empty((0, 8))
            out["roi%d_%d_%d_dist" % (k, *key)] = np.array(df["Distance"], dtype=float) if nrow else np.empty(0)
    out["roi_table"] = table
    out["roi_tol"] = tol
    # ---- verifier.match_blobs_roi itself (all five outputs, flags as it leaves them): anisotropic tolerances, a
    # core smaller than the default padding allows, base blobs whose truth flag is already 0 / 1 outside the core
    flagged = table.copy()
    rng2 = np.random.default_rng(77)
    flagged[:, 5] = rng2.integers(-1, 2, len(flagged))
    flagged[:, 4] = rng2.integers(-1, 2, len(flagged))
    mspecs = (((0, 0, 0), (64, 60, 40), (5.0, 5.0, 5.0), table, 1, 0),
              ((6, 4, 2), (50, 40, 30), (7.0, 5.0, 3.0), flagged, 2, 0),
              ((20, 20, 10), (9, 7, 5), (5.0, 5.0, 5.0), flagged, 2, 1),
              ((0, 0, 0), (64, 60, 40), (2.0, 2.0, 2.0), flagged, 1, 2))
    for k, (offset, size, mtol, tbl, chl_det, chl_base) in enumerate(mspecs):
        thresh, scaling, pad, _, _ = quiet(verifier.setup_match_blobs_roi, np.array(mtol))
        det = tbl[tbl[:, 6] == chl_det]
        base = tbl[tbl[:, 6] == chl_base]
        got = quiet(verifier.match_blobs_roi, det.copy(), base.copy(), offset, size, thresh, scaling, pad)
        df = got[4].df
        nrow = 0 if df is None else len(df)
        out.update({"mroi%d_det" % k: det, "mroi%d_base" % k: base, "mroi%d_offset" % k: np.array(offset),
                    "mroi%d_size" % k: np.array(size), "mroi%d_tol" % k: np.array(mtol),
                    "mroi%d_inner_plus" % k: got[0], "mroi%d_truth_inner_plus" % k: got[1],
                    "mroi%d_offset_inner" % k: np.asarray(got[2], dtype=float),
                    "mroi%d_size_inner" % k: np.asarray(got[3], dtype=float),
                    "mroi%d_blob1" % k: np.vstack(df["Blob1"]) if nrow else np.empty((0, 8)),
                    "mroi%d_blob2" % k: np.vstack(df["Blob2"]) if nrow else np.empty((0, 8)),
                    "mroi%d_dist" % k: np.array(df["Distance"], dtype=float) if nrow else np.empty(0)})
        print("match_blobs_roi %d: %d matches, inner+ %s, truth+ %s" % (k, nrow, got[0].shape, got[1].shape))
    out["n_mroi"] = np.array(len(mspecs))
    # ---- whole stack: larger-overlap block split, per-block matching, shortest-distance de-duplication
    for name, shape, n, n_chl, seg, res in (("stackA", (48, 120, 128), 700, 2, 40, (1., 1., 1.)),
                                            ("stackB", (40, 100, 96), 400, 3, 36, (2.0, 0.8, 0.8))):
        config.resolutions = np.array([res])
        config.cpus = 4
        setup_profile(segment_size=seg, num_sigma=3)
        quiet(chunking.set_mp_start_method)
        tbl = make_blob_table(73 + n_chl, shape, n, n_chl=n_chl, jitter=3)
        blobs = detector.Blobs(tbl.copy())
        matches = quiet(colocalizer.StackColocalizer.colocalize_stack, shape, blobs)
        out[name + "_table"] = tbl
        out[name + "_shape"] = np.array(shape)
        out[name + "_res"] = np.array(res)
        out[name + "_segment_size"] = np.array(seg)
        out[name + "_keys"] = np.array(sorted(matches.keys())).reshape(-1, 2)
        for key, bm in matches.items():
            df = bm.df
            out["%s_%d_%d_blob1" % (name, *key)] = np.vstack(df["Blob1"]) if len(df) else np.empty((0, 8))
            out["%s_%d_%d_blob2" % (name, *key)] = np.vstack(df["Blob2"]) if len(df) else np.empty((0, 8))
            out["%s_%d_%d_dist" % (name, *key)] = np.array(df["Distance"], dtype=float)
            print("match %s %s: %d matches" % (name, key, len(df)))
    out["versions"] = np.array(repr(VERSIONS))
    np.savez_compressed(os.path.join(HERE, "match.npz"), **out)
    print("match.npz: %d arrays" % len(out))


def main_match():
    match_cases()


def main_overlap():
    """skimage.feature.blob._prune_blobs itself on crowded random tables (several scales, many blobs that both win and
    lose an over-limit pair): its outcome then depends on the order cKDTree.query_pairs' set is iterated in."""
    from skimage.feature import blob as sk_blob
    rng = np.random.default_rng(91)
    sigmas = np.array([3.0, 3.5, 4.0, 4.5, 5.0])
    out = {"sigmas": sigmas, "versions": np.array(repr(VERSIONS))}
    specs = [(60, 30, 0.5), (300, 60, 0.5), (1200, 110, 0.5), (500, 60, 0.3), (800, 90, 0.7), (40, 12, 0.5)]
    for k, (n, box, overlap) in enumerate(specs):
        coords = rng.integers(0, box, (n, 3))
        sidx = rng.integers(0, len(sigmas), n)
        table = np.hstack((coords.astype(float), sigmas[sidx][:, None]))
        kept = sk_blob._prune_blobs(table.copy(), overlap)
        out["case%d_coords" % k] = np.hstack((coords, sidx[:, None])).astype(np.int32)
        out["case%d_overlap" % k] = np.array(overlap)
        out["case%d_kept" % k] = kept
        print("overlap prune case %d: %d blobs -> %d" % (k, n, len(kept)))
    out["n_cases"] = np.array(len(specs))
    np.savez_compressed(os.path.join(HERE, "overlap_prune.npz"), **out)


def main_tv():
    """plot_3d.denoise_roi with the profile's ``tot_var_denoise`` on (skimage.restoration.denoise_tv_chambolle):
    sub-block cases (profiles 'minpreproc' and '2p20x' settings, sparse / dense / ragged / uint8 tiles) and one
    whole-stack detection."""
    from magmap.plot import plot_3d
    from skimage import restoration
    out = {}
    names = []

    def case(name, roi, near_max=(-1.0,), **over):
        setup_profile(None, **over)
        config.near_max = list(near_max)
        sat = quiet(plot_3d.saturate_roi, roi, channel=None)
        den = quiet(plot_3d.denoise_roi, sat, channel=None)
        out[name + "_roi"] = roi
        out[name + "_near_max"] = np.array(near_max, dtype=float)
        out[name + "_over"] = repr(over)
        out[name + "_sat"] = sat
        out[name + "_den"] = den
        names.append(name)
        print("tv %-10s %s -> den [%.5f, %.5f]" % (name, roi.shape, float(den.min()), float(den.max())))

    minpre = dict(clip_vmin=0, clip_vmax=99.99, clip_max=1, tot_var_denoise=0.01, unsharp_strength=0,
                  erosion_threshold=0)
    p2p20 = dict(clip_vmax=97, clip_min=0, clip_max=0.7, tot_var_denoise=True, unsharp_strength=2.5)
    sparse = make_volume(81, (25, 25, 25), 3, margin=4)
    densev = make_volume(82, (25, 25, 25), 40, amp=6000.0, blob_sigma=3.5, bg_mean=3000.0, bg_sd=800.0, margin=0)
    case("min_sparse", sparse, **minpre)
    case("min_dense", densev, **minpre)
    case("2p_sparse", sparse, **p2p20)
    case("2p_dense", densev, **p2p20)
    case("w01_ragged", make_volume(83, (14, 25, 9), 2, margin=3), tot_var_denoise=0.1)
    case("w01_u8", make_volume(84, (20, 22, 24), 4, dtype=np.uint8, margin=4), tot_var_denoise=0.1)
    case("w05_big", make_volume(85, (30, 36, 34), 8, margin=4), tot_var_denoise=0.05, unsharp_strength=0.3)
    case("const", np.full((10, 12, 11), 900, dtype=np.uint16), tot_var_denoise=0.1)
    case("tiny", make_volume(86, (2, 1, 5), 0, margin=0), tot_var_denoise=0.1)
    # the bare algorithm on a float image (iteration count and values)
    rng = np.random.default_rng(87)
    img = rng.random((9, 11, 10))
    out["bare_img"] = img
    out["bare_w02"] = restoration.denoise_tv_chambolle(img, weight=0.2)
    config.near_max = [-1.0]
    out["names"] = np.array(names)
    out["versions"] = repr(VERSIONS)
    np.savez_compressed(os.path.join(HERE, "tv.npz"), **out)
    stack_case("tv", make_volume(88, (40, 60, 64), 30), None, segment_size=36, num_sigma=3, denoise_size=20,
               tot_var_denoise=0.05)


def main_coloc():
    # ---- intensity co-localisation (colocalizer.colocalize_blobs) alone and inside whole-stack detection
    coloc_cases()
    stack_case("coloc_2ch", make_coloc_volume(53, (50, 70, 70), 30), None, segment_size=32, num_sigma=3,
               coloc=True)
    stack_case("coloc_3ch", make_coloc_volume(54, (44, 64, 60), 24, n_chl=3, shared=0.4), None,
               segment_size=36, num_sigma=3, coloc=True)
    stack_case("coloc_denoise", make_coloc_volume(55, (48, 66, 64), 26), None, segment_size=34,
               num_sigma=3, denoise_size=25, near_max=(-1.0, -1.0), coloc=True)


def main_preproc_f64():
    """saturate_roi / denoise_roi of the real reference on FLOAT64 sub-blocks, and one float64 stack end to end."""
    from magmap.plot import plot_3d
    out, names = {}, []

    def case(name, roi, near_max=(-1.0,), **over):
        assert roi.dtype == np.float64
        setup_profile(None, **over)
        config.near_max = list(near_max)
        sat = quiet(plot_3d.saturate_roi, roi, channel=None)
        den = quiet(plot_3d.denoise_roi, sat, channel=None)
        out[name + "_roi"] = roi
        out[name + "_near_max"] = np.array(near_max, dtype=float)
        out[name + "_over"] = repr(over)
        out[name + "_sat"] = sat
        out[name + "_den"] = den
        names.append(name)
        print("preproc_f64 %-10s %s -> sat %s mean %.4f, den %s [%.4f, %.4f]" % (
            name, roi.shape, sat.dtype, float(np.mean(sat)), den.dtype, float(den.min()), float(den.max())))

    sparse = make_volume(41, (25, 25, 25), 3, margin=4)
    densev = make_volume(42, (25, 25, 25), 40, amp=6000.0, blob_sigma=3.5, bg_mean=3000.0, bg_sd=800.0, margin=0)
    case("unit", sparse / 65535.0)                           # what img_as_float would have made of the uint16 tile
    case("neg", densev * 0.37 - 1500.25)                     # negative and fractional values
    case("const", np.full((25, 25, 25), -2.5))               # identity tile (vmin == vmax) of negative values
    case("ragged", make_volume(43, (14, 25, 9), 2, margin=3) * 1e-3)
    case("tiny", make_volume(46, (2, 1, 5), 0, margin=0) / 7.0)
    rng = np.random.default_rng(48)
    case("ties", rng.integers(-3, 4, (20, 25, 25)) * 0.1)    # many ties around the percentile ranks, both signs
    case("nearmax", sparse / 65535.0, near_max=(0.9,))
    case("clipvals", densev / 65535.0, clip_vmin=2, clip_vmax=90.5, clip_min=0.1, clip_max=0.8,
         unsharp_strength=0.45, erosion_threshold=0.1)
    case("big", make_volume(49, (32, 40, 36), 10, margin=4) / 65535.0)
    config.near_max = [-1.0]
    out["names"] = np.array(names)
    out["versions"] = repr(VERSIONS)
    np.savez_compressed(os.path.join(HERE, "preproc_f64.npz"), **out)
    stack_case("denoise_f64", make_volume(35, (64, 96, 96), 60) / 65535.0, None, segment_size=40, num_sigma=5,
               denoise_size=25)


def main_preproc():
    # ---- per-block preprocessing (saturate + denoise) and whole-stack detection with it on
    preproc_cases()
    stack_case("denoise", make_volume(35, (64, 96, 96), 60), None, segment_size=40, num_sigma=5,
               denoise_size=25)
    bgvol = make_volume(36, (56, 84, 80), 50, amp=9000.0, bg_mean=2500.0, bg_sd=700.0)
    stack_case("denoise_dense", bgvol, None, segment_size=36, num_sigma=3, denoise_size=20,
               near_max=(30000.0,))
    stack_case("denoise_aniso", make_volume(37, (40, 90, 93), 40), None,
               resolutions=((2.0, 0.8, 0.8),), segment_size=60, num_sigma=3, denoise_size=30)
    vol2d = np.stack((make_volume(38, (44, 60, 62), 24), make_volume(39, (44, 60, 62), 20)), axis=-1)
    stack_case("denoise_2ch", vol2d, None, segment_size=32, num_sigma=3, denoise_size=25,
               near_max=(-1.0, 25000.0))   # one near_max per channel or the reference raises IndexError


if __name__ == "__main__":
    main()
