"""CPU tests of the host-side mirror of the reference interface (no GPU, no compute calls
through the C ABI).  Modelled on the reference's own unit tests
(magmap/tests/test_detector.py, test_chunking.py) plus the golden vectors."""
import ctypes
import os
import re

import math
import numpy as np
import pytest

from conftest import ROOT, lexsorted, load_golden
from magellanmapper_amd import _native, chunking, config, detector, kernels1d, roi_prof, stack_detect


# ---------------------------------------------------------------- C ABI (load + symbols)
def test_library_loads_and_exports_every_declared_symbol():
    assert os.path.exists(_native.LIB_PATH), "build libmmx_hip.so first (__graft_entry__.build())"
    lib = _native.lib()
    header = open(os.path.join(ROOT, "include", "mmx.h")).read()
    declared = set(re.findall(r"\b(mmx_[a-z0-9_]+)\s*\(", header))
    declared -= {"mmx_status", "mmx_dtype"}
    assert declared == set(_native.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.mmx_abi_version() == _native.MMX_ABI_VERSION
    assert lib.mmx_strerror(0) == b"ok" and lib.mmx_strerror(5) == b"unsupported configuration"


def test_struct_layouts_match_header():
    assert _native.BLOCK_DTYPE.itemsize == 32
    assert _native.CAND_DTYPE.itemsize == 48
    assert _native.CAND_DTYPE.fields["v64"][1] == 32
    assert ctypes.sizeof(_native.Volume) == 40


def test_argument_validation_without_gpu():
    """Bad arguments are rejected before any device work."""
    lib = _native.lib()
    assert lib.mmx_peaks_batch(None, None, 0, 1, None, None, 1, 1, 0.1, 1e-5, None, 1, None, None) == 1
    vol = _native.Volume(0, 1, 0, 1, 1, 1)
    w = np.ones(3)
    assert lib.mmx_log_batch_f32(ctypes.byref(vol), None, None, 1, 8, _native.as_double_ptr(w),
                                 _native.as_double_ptr(w), 2, 1.0, None, None, None, 0.0, 0.0, None, -1, None, None) == 1


def test_product_path_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "magellanmapper_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("oracle/ndfilters.c", ""), fn


def test_no_gpu_means_loud_failure():
    torch = pytest.importorskip("torch")
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from magellanmapper_amd import blob_log as bl
    with pytest.raises(_native.MmxError):
        bl.blob_log(np.zeros((8, 8, 8), np.uint16), 1, 2, 2, 0.1, 0.5)


# ---------------------------------------------------------------- filter parameters
@pytest.mark.parametrize("sigma", [0.7, 1.0, 2.6, 2.8, 3.0, 3.5, 5.0, 14.0])
def test_half_kernels_equal_scipy(sigma):
    from scipy.ndimage import _filters as sf
    R = kernels1d.kernel_radius(sigma)
    for order in (0, 2):
        full = sf._gaussian_kernel1d(sigma, order, R)[::-1]
        np.testing.assert_array_equal(kernels1d.gaussian_half_kernel(sigma, order, R), full[R:])


def test_sigma_ladder_equals_oracle():
    from oracle import blob_log_oracle as blo
    for lo, hi, n in ((3, 5, 10), (3, 3, 1), (2.6 / 1.1, 2.8 / 1.1, 10), (1.0, 5.5, 10)):
        sig, norm = kernels1d.sigma_ladder(lo, hi, n)
        want, scalar = blo.sigma_list(lo, hi, n)
        assert scalar
        np.testing.assert_array_equal(sig, want[:, 0])
        np.testing.assert_array_equal(norm, [np.mean(r) ** 2 for r in want])


def test_plan_batches():
    from magellanmapper_amd.blob_log import plan_batches
    shapes = [(10, 10, 10)] * 5 + [(20, 10, 10)] + [(10, 10, 10)] * 3
    b = plan_batches(shapes, 5, budget_bytes=4 * 3200 * 36)
    assert sum(b, []) == list(range(9))
    for batch in b:
        slot = max(shapes[i][0] * shapes[i][1] * 32 for i in batch)     # rows are pitched to 32 floats
        assert len(batch) == 1 or len(batch) * slot * 36 <= 4 * 3200 * 36
    assert plan_batches(shapes, 5, 1 << 40) == [list(range(9))]


# ---------------------------------------------------------------- Blobs (reference test_detector.py)
def test_blob_columns_and_accessors():
    rng = np.random.default_rng(0)
    blobs = rng.random(20).reshape((5, 4))
    blobs[:, :3] = np.multiply(blobs[:, :3], 100).astype(int)
    blobs[:, 3] = blobs[:, 3] * 10
    bl = detector.Blobs(blobs)
    assert bl.cols == [c.value for c in bl.Cols][:4]
    assert bl._col_inds[bl.Cols.RADIUS] == 3
    assert bl._col_inds[bl.Cols.ABS_X] is None
    bl.format_blobs()
    assert bl.cols == [c.value for c in bl.Cols]
    assert bl._col_inds[bl.Cols.RADIUS] == 3 and bl._col_inds[bl.Cols.ABS_X] == 9
    assert bl.blobs.shape == (5, 11)
    np.testing.assert_array_equal(bl.blobs[:, 7:10], bl.blobs[:, :3])
    np.testing.assert_array_equal(bl.blobs[:, [4, 5, 6, 10]], -np.ones((5, 4)))

    np.testing.assert_array_equal(bl.get_blob_confirmed(bl.blobs), bl.blobs[:, 4])
    bl.set_blob_confirmed(bl.blobs, 1)
    assert np.all(bl.get_blob_confirmed(bl.blobs) == 1)
    np.testing.assert_array_equal(bl.get_blob_truth(bl.blobs), bl.blobs[:, 5])
    bl.set_blob_truth(bl.blobs, 2)
    assert np.all(bl.get_blob_truth(bl.blobs) == 2)
    np.testing.assert_array_equal(bl.get_blobs_channel(bl.blobs), bl.blobs[:, 6])
    bl.set_blob_channel(bl.blobs, 3)
    assert np.all(bl.get_blobs_channel(bl.blobs) == 3)
    np.testing.assert_array_equal(bl.get_blob_abs_coords(bl.blobs), bl.blobs[:, 7:10])
    bl.set_blob_abs_coords(bl.blobs, (1, 2, 3))
    assert all(np.all(bl.get_blob_abs_coords(bl.blobs) == (1, 2, 3), axis=1))

    t = bl.blobs.copy()
    detector.Blobs.shift_blob_rel_coords(t, (10, 20, 30))
    np.testing.assert_array_equal(t[:, :3], bl.blobs[:, :3] + (10, 20, 30))
    detector.Blobs.replace_rel_with_abs_blob_coords(t)
    np.testing.assert_array_equal(t[:, :3], t[:, 7:10])
    b2 = detector.Blobs(t)
    out = b2.remove_abs_blob_coords(True)
    assert out.shape[1] == 8
    assert b2.cols == ["z", "y", "x", "radius", "confirmed", "truth", "channel", "region"]
    detector.Blobs(np.ones((1, 4))).format_blobs()   # restore the 11-column registry


def test_archive_roundtrip(tmp_path):
    tbl = np.arange(24, dtype=float).reshape(3, 8)
    b = detector.Blobs(tbl, path=str(tmp_path / "x_blobs.npz"),
                       cols=["z", "y", "x", "radius", "confirmed", "truth", "channel", "region"])
    b.resolutions = np.array([[1., 1., 1.]])
    b.basename, b.roi_offset, b.roi_size = "x", (0, 0, 0), (3, 4, 5)
    arc = b.save_archive()
    assert set(arc) == {"ver", "segments", "resolutions", "basename", "offset", "roi_size", "colocs",
                        "columns"}
    b.save_archive()                                    # second save backs the first one up
    assert os.path.exists(tmp_path / "x_blobs(1).npz")
    back = detector.Blobs().load_blobs(b.path)
    np.testing.assert_array_equal(back.blobs, tbl)
    assert list(back.cols) == b.cols and back.basename == "x"
    detector.Blobs(np.ones((1, 4))).format_blobs()


def test_resolution_helpers():
    config.resolutions = None
    with pytest.raises(AttributeError):
        detector.calc_scaling_factor()
    config.resolutions = [[6.6, 1.1, 1.1]]
    g = load_golden("blocks.npz")
    np.testing.assert_array_equal(detector.calc_overlap(2), g["calc_overlap_2"])
    np.testing.assert_allclose(detector.calc_scaling_factor(), 1 / np.array([6.6, 1.1, 1.1]))


def test_interior_and_roi_filters():
    rng = np.random.default_rng(1)
    t = np.hstack((rng.integers(0, 30, (50, 3)).astype(float), np.ones((50, 8))))
    inner = detector.get_blobs_interior(t, (30, 30, 30), (2, 3, 4), (1, 0, 5))
    keep = ((t[:, 0] >= 2) & (t[:, 0] < 29) & (t[:, 1] >= 3) & (t[:, 1] < 30) & (t[:, 2] >= 4) &
            (t[:, 2] < 25))
    np.testing.assert_array_equal(inner, t[keep])
    sub, mask = detector.get_blobs_in_roi(t, (5, 6, 7), (10, 10, 10))     # x, y, z order
    want = ((t[:, 0] >= 7) & (t[:, 0] < 17) & (t[:, 1] >= 6) & (t[:, 1] < 16) & (t[:, 2] >= 5) &
            (t[:, 2] < 15))
    np.testing.assert_array_equal(mask, want)


# ---------------------------------------------------------------- chunking (reference test_chunking.py)
def _split_remerge(roi, max_pixels, overlap):
    slices, _ = chunking.stack_splitter(roi.shape, max_pixels, overlap)
    out = np.zeros_like(roi)
    grid = slices.shape
    for c in np.ndindex(*grid):
        sub = roi[slices[c]]
        keep = [sub.shape[a] if c[a] == grid[a] - 1 else min(sub.shape[a], max_pixels[a])
                for a in range(3)]
        o = [slices[c][a].start for a in range(3)]
        out[o[0]:o[0] + keep[0], o[1]:o[1] + keep[1], o[2]:o[2] + keep[2]] = \
            sub[:keep[0], :keep[1], :keep[2]]
    return out


def test_stack_splitter_reference_geometry():
    roi = np.arange(5 * 4 * 4).reshape((5, 4, 4))
    g = load_golden("blocks.npz")
    config.resolutions = [[6.6, 1.1, 1.1]]
    for j, ov in enumerate([np.array((0, 1, 1)), np.array((0, 1, 2)), np.array((1, 1, 2)),
                            detector.calc_overlap(2)]):
        sl, off = chunking.stack_splitter(roi.shape, [1, 3, 3], ov)
        got = np.array([[[s.start, s.stop] for s in sl[c]] for c in np.ndindex(*sl.shape)]
                       ).reshape(sl.shape + (3, 2))
        np.testing.assert_array_equal(got, g["ss%d_slices" % j])
        np.testing.assert_array_equal(off, g["ss%d_offsets" % j])
        np.testing.assert_array_equal(_split_remerge(roi, [1, 3, 3], ov), roi)


def test_setup_blocks_sweep_matches_reference():
    g = load_golden("blocks.npz")
    for i in range(int(g["n_cases"])):
        pre = "c%d_" % i
        excl = None if g[pre + "exclude_border"].ndim == 0 else tuple(g[pre + "exclude_border"])
        dn = None if float(g[pre + "denoise_size"]) < 0 else g[pre + "denoise_size"].item()
        config.resolutions = [g[pre + "resolutions"]]
        prof = roi_prof.ROIProfile(segment_size=g[pre + "segment_size"].item(), exclude_border=excl,
                                   prune_tol_factor=tuple(g[pre + "prune_tol_factor"]),
                                   denoise_size=dn)
        bl = stack_detect.setup_blocks(prof, tuple(g[pre + "shape"]))
        grid = bl.sub_roi_slices.shape
        sl = np.array([[[s.start, s.stop] for s in bl.sub_roi_slices[c]]
                       for c in np.ndindex(*grid)]).reshape(grid + (3, 2))
        np.testing.assert_array_equal(sl, g[pre + "slices"])
        np.testing.assert_array_equal(bl.sub_rois_offsets, g[pre + "offsets"])
        for key in ("tol", "overlap_base", "overlap", "overlap_padding", "max_pixels"):
            np.testing.assert_array_equal(getattr(bl, key), g[pre + key])
        if g[pre + "denoise_max_shape"].ndim == 0:
            assert bl.denoise_max_shape is None
        else:
            np.testing.assert_array_equal(bl.denoise_max_shape, g[pre + "denoise_max_shape"])
    # the benchmark grid: 4 x 8 x 8 blocks of at most 261^3
    config.resolutions = [[1., 1., 1.]]
    bl = stack_detect.setup_blocks(roi_prof.ROIProfile(segment_size=256, denoise_size=None),
                                   (1024, 2048, 2048))
    assert bl.sub_roi_slices.shape == (4, 8, 8)
    assert bl.sub_roi_slices[0, 0, 0] == (slice(0, 261), slice(0, 261), slice(0, 261))
    assert bl.sub_roi_slices[3, 7, 7] == (slice(768, 1024), slice(1792, 2048), slice(1792, 2048))


def test_merge_blobs_tags_rows():
    seg = np.zeros((1, 2, 2), dtype=object)
    seg[0, 0, 0] = np.ones((2, 11))
    seg[0, 0, 1] = None
    seg[0, 1, 1] = np.full((3, 11), 2.0)
    m = chunking.merge_blobs(seg)
    assert m.shape == (5, 14)
    np.testing.assert_array_equal(m[:2, 11:], [[0, 0, 0]] * 2)
    np.testing.assert_array_equal(m[2:, 11:], [[0, 1, 1]] * 3)
    assert chunking.merge_blobs(np.zeros((2, 1, 1), dtype=object)) is None


# ---------------------------------------------------------------- profiles
def test_profile_layering_and_yaml(tmp_path):
    prof = roi_prof.ROIProfile()
    assert prof["min_sigma_factor"] == 3 and prof["num_sigma"] == 10 and prof["denoise_size"] == 25
    prof.add_profiles("lightsheet,4xnuc")
    assert prof["settings_name"] == "default,lightsheet,4xnuc"
    assert prof["max_sigma_factor"] == 4 and prof["overlap"] == 0.55        # later wins
    assert prof["exclude_border"] == (1, 0, 0) and prof["isotropic"] == (0.96, 1, 1)
    y = tmp_path / "roi_test.yaml"
    y.write_text("---\nnum_sigma: 5\ndetection_threshold: 0.2\nexclude_border: null\n...\n")
    prof.add_profiles(str(y))
    assert prof["num_sigma"] == 5 and prof["exclude_border"] is None
    assert not prof.check_file_changed()
    os.utime(y, (os.path.getmtime(y) + 10,) * 2)
    assert prof.check_file_changed()
    prof.add_profiles("nonexistent")                                       # skipped, not an error
    a, b = roi_prof.ROIProfile(), roi_prof.ROIProfile()
    assert roi_prof.ROIProfile.is_identical_settings([a, b], roi_prof.ROIProfile.BLOCK_SIZES)
    b.add_profiles("20x")
    assert not roi_prof.ROIProfile.is_identical_settings([a, b], roi_prof.ROIProfile.BLOCK_SIZES)


def test_per_channel_profiles():
    config.setup_roi_profiles(["lightsheet", "4xnuc"])
    assert config.get_roi_profile(0)["segment_size"] == 150
    assert config.get_roi_profile(1)["max_sigma_factor"] == 4
    assert config.get_roi_profile(7) is config.roi_profile      # beyond the list: the default one
    config.setup_roi_profiles(None)


# ---------------------------------------------------------------- overlap pruning (host rules)
def _brute_close(check, master, tol):
    d = np.abs(master[:, None, :].astype(np.int64) - check[None, :, :].astype(np.int64))
    close = (d <= np.asarray(tol)).all(2)
    last = np.where(close.any(1), close.shape[1] - 1 - np.argmax(close[:, ::-1], axis=1), -1)
    return last.astype(np.int32), close.any(0)


def test_remove_close_blobs_apply_rules(monkeypatch):
    """Deletion, round-half-even averaging and last-write-wins vs the reference's outputs;
    the device search is replaced by a brute-force matcher (the search itself is GPU-tested)."""
    monkeypatch.setattr(detector, "find_close_pairs", _brute_close)
    g = load_golden("prune.npz")
    for k in range(int(g["n_rc"])):
        detector.Blobs(np.ones((1, 4))).format_blobs()
        pruned, master = detector.remove_close_blobs(
            g["rc%d_check" % k].copy(), g["rc%d_master" % k].copy(), g["rc%d_tol" % k])
        np.testing.assert_array_equal(pruned, g["rc%d_pruned" % k])
        np.testing.assert_array_equal(master, g["rc%d_master_out" % k])
    # explicit half-even case: (3 + 4) / 2 = 3.5 -> 4, (4 + 5) / 2 = 4.5 -> 4
    m = -np.ones((1, 11)); m[0, :3] = (3, 4, 7); m[0, 7:10] = (3, 4, 7)
    c = -np.ones((1, 11)); c[0, :3] = (4, 5, 7); c[0, 7:10] = (4, 5, 7)
    _, m2 = detector.remove_close_blobs(c, m, (1, 1, 1))
    np.testing.assert_array_equal(m2[0, 7:10], (4, 4, 7))


def test_prune_blobs_mp_matches_reference(monkeypatch):
    monkeypatch.setattr(detector, "find_close_pairs", _brute_close)
    g = load_golden("prune.npz")
    config.resolutions = [[1., 1., 1.]]
    shape = tuple(g["sp_shape"])
    bl = stack_detect.setup_blocks(roi_prof.ROIProfile(segment_size=20, denoise_size=None), shape)
    grid = tuple(g["sp_grid"])
    seg = np.zeros(grid, dtype=object)
    for c in np.ndindex(*grid):
        t = g["sp_block_%d_%d_%d" % c]
        seg[c] = None if t.shape[0] == 0 else t.copy()
    pruned, df = stack_detect.StackPruner.prune_blobs_mp(
        np.zeros(shape, np.uint8), seg, bl.overlap, bl.tol, bl.sub_roi_slices, bl.sub_rois_offsets,
        [0, 1], bl.overlap_padding)
    np.testing.assert_array_equal(pruned, g["sp_pruned"])
    np.testing.assert_allclose(df.to_numpy(), g["sp_ratios"])
    assert list(df.columns) == list(g["sp_ratio_cols"])
    empty = np.zeros(grid, dtype=object)
    empty[:] = None
    assert stack_detect.StackPruner.prune_blobs_mp(
        np.zeros(shape, np.uint8), empty, bl.overlap, bl.tol, bl.sub_roi_slices,
        bl.sub_rois_offsets, [0]) == (None, None)


def test_stack_level_prune_from_golden_block_tables(monkeypatch):
    """Feed the real reference's per-block tables through our pruner + epilogue."""
    monkeypatch.setattr(detector, "find_close_pairs", _brute_close)
    import ast
    for case in ("u16_2x3x3", "u16_border", "2ch"):
        g = load_golden("stack_%s.npz" % case)
        over = ast.literal_eval(str(g["overrides"]))
        config.resolutions = g["resolutions"]
        prof = roi_prof.ROIProfile(denoise_size=None)
        prof.update(over)
        roi = g["roi"]
        bl = stack_detect.setup_blocks(prof, roi.shape)
        grid = tuple(g["grid"])
        seg = np.zeros(grid, dtype=object)
        for c in np.ndindex(*grid):
            t = g["block_%d_%d_%d" % c]
            seg[c] = None if t.shape[0] == 0 else t.copy()
        chls = list(range(roi.shape[3])) if roi.ndim > 3 else [0]
        pruned, _ = stack_detect.StackPruner.prune_blobs_mp(
            roi, seg, bl.overlap, bl.tol, bl.sub_roi_slices, bl.sub_rois_offsets, chls,
            bl.overlap_padding)
        np.testing.assert_array_equal(pruned[:, 3:], g["pruned11"][:, 3:])
        b = detector.Blobs(pruned)
        b.replace_rel_with_abs_blob_coords(pruned)
        b.blobs = pruned
        np.testing.assert_array_equal(b.remove_abs_blob_coords(True), g["final"])


def test_unbuilt_rows_fail_loudly():
    from magellanmapper_amd import colocalizer
    with pytest.raises(_native.MmxError):          # percentile thresholds are built: no GPU -> loud failure, no fallback
        colocalizer.colocalize_blobs(np.zeros((4, 4, 4, 2), np.uint16), np.zeros((1, 11)), thresh=50)
    assert colocalizer.colocalize_blobs(np.zeros((4, 4, 4), np.uint16), np.zeros((1, 11))) is None
    from magellanmapper_amd import preprocess
    config.setup_roi_profiles(["minpreproc"])              # a profile with tot_var_denoise: built since round 2
    try:
        p = preprocess.channel_params(0)[0]
        assert p.tv_weight == 0.01 and p.tv_factor == (1. / 6) / 0.01 and p.unsharp_strength == 0
        config.setup_roi_profiles(["2p20x"])               # tot_var_denoise True: weight 1
        p = preprocess.channel_params(0)[0]
        assert p.tv_weight == 1.0 and p.tv_factor == 1. / 6 and p.unsharp_strength == 2.5
    finally:
        config.setup_roi_profiles()
    with pytest.raises(ValueError):
        stack_detect.detect_blobs_blocks("x", stack_detect.Image5d(None))
    with pytest.raises(IOError):
        stack_detect.detect_blobs_stack("x", None)


# ---------------------------------------------------------------- co-localisation host logic
def test_coloc_flags_from_means_match_reference():
    """Thresholds + flags (host NumPy) from a means matrix built the reference's way on the CPU:
    equals the real reference's colocalize_blobs on every golden case, incl. the NaN-poisoned one."""
    from scipy import ndimage as ndi
    from magellanmapper_amd import colocalizer
    from oracle import coloc_oracle
    from test_oracle_golden import COLOC, coloc_roi, coloc_thresh
    for case in [str(n) for n in COLOC["names"]]:
        roi, blobs, want = coloc_roi(COLOC, case), COLOC[case + "_blobs"], COLOC[case + "_colocs"]
        if roi.ndim < 4:
            continue
        n_chl = roi.shape[3]
        shape3 = roi.shape[:3]
        inside = np.all((blobs[:, :3] >= 0) & (blobs[:, :3] < shape3), axis=1)
        means = np.full((len(blobs), n_chl), np.nan)
        pct = coloc_thresh(COLOC, case)
        thresholds = None if pct is None else {}
        for bc in range(n_chl):
            sel = np.where(inside & (blobs[:, 6] == bc))[0]
            mask = -np.ones(shape3, dtype=int)
            c = blobs[sel, :3].astype(int)
            mask[c[:, 0], c[:, 1], c[:, 2]] = sel
            mask = ndi.grey_dilation(mask, footprint=coloc_oracle.ball(2))
            if pct is not None and len(sel):      # percentile of the channel over every voxel its blobs own
                thresholds[bc] = np.percentile(roi[mask >= 0, bc], pct)
            for b in sel:
                vox = mask == b
                for oc in range(n_chl):
                    means[b, oc] = np.mean(roi[vox, oc]) if vox.any() else np.nan
        got = colocalizer._flags_from_means(blobs, means, shape3, n_chl, thresholds)
        np.testing.assert_array_equal(got, want, err_msg=case)


# ---------------------------------------------------------------- on-disk image (image5d.npy + meta.yml)
def test_read_file_matches_the_reference_importer(tmp_path):
    """sample_image5d.npy / sample_meta.yml were written by the real reference's importer; what its
    read_file reported is in image5d_expect.npz."""
    import shutil
    from conftest import GOLDEN
    from magellanmapper_amd import importer
    exp = load_golden("image5d_expect.npz")
    for fn in ("sample_image5d.npy", "sample_meta.yml"):
        shutil.copy(os.path.join(GOLDEN, fn), tmp_path / fn)
    config.resolutions, config.near_max = None, [-1.0]
    try:
        img5d = importer.read_file(str(tmp_path / "sample.czi"), 0)
        assert os.path.basename(img5d.path_img) == str(exp["path_img"])
        assert os.path.basename(img5d.path_meta) == str(exp["path_meta"])
        assert isinstance(img5d.img, np.memmap) == bool(exp["is_memmap"])
        assert tuple(img5d.img.shape) == tuple(exp["shape"]) and str(img5d.img.dtype) == str(exp["dtype"])
        np.testing.assert_array_equal(config.resolutions, exp["resolutions"])
        np.testing.assert_array_equal(config.near_max, exp["near_max"])
        np.testing.assert_array_equal(config.near_min, exp["near_min"])
        assert config.magnification == float(exp["magnification"]) and config.zoom == float(exp["zoom"])
        assert set(exp["meta_keys"]) <= set(img5d.meta)
        # and the writer: same file content as the reference's save_image_info
        md = img5d.meta
        importer.save_image_info(str(tmp_path / "again_meta.yml"), md["names"], md["sizes"], md["resolutions"],
                                 md["magnification"], md["zoom"], md["near_min"], md["near_max"])
        assert open(tmp_path / "again_meta.yml").read() == open(tmp_path / "sample_meta.yml").read()
        # pre-v1.4 archives kept the metadata in an .npz next to the image
        os.remove(tmp_path / "sample_meta.yml")
        np.savez(tmp_path / "sample_meta.npz", **{k: np.array(v, dtype=object) if v is None else v
                                                  for k, v in md.items()})
        config.resolutions = None
        importer.read_file(str(tmp_path / "sample.czi"))
        np.testing.assert_array_equal(config.resolutions, exp["resolutions"])
        # a missing image leaves img None, which detect_blobs_stack turns into the reference's IOError
        missing = importer.read_file(str(tmp_path / "nothing.czi"))
        assert missing.img is None
        with pytest.raises(NotImplementedError):
            importer.read_file(str(tmp_path / "sample.czi"), offset=(0, 0, 0), size=(4, 4, 4))
    finally:
        config.resolutions, config.near_max = None, [-1.0]


def test_filename_helpers():
    from magellanmapper_amd import importer
    assert importer.make_filenames("/d/brain.czi") == ("/d/brain_image5d.npy", "/d/brain_meta.yml")
    assert importer.make_filenames("/d/brain.nii.gz")[0] == "/d/brain_image5d.npy"
    assert importer.make_filenames("/d/brain.v2.tif", keep_ext=True)[0] == "/d/brain.v2.tif_image5d.npy"
    assert importer.combine_paths("/d/", "x.npy") == "/d/x.npy" and importer.combine_paths(None, "x") == "x"


# ---------------------------------------------------------------- isotropic rescale: host tables
def test_zoom_axis_tables_reproduce_scipy_zoom():
    """index / weight tables (host) + NI_ZoomShift's accumulation order == scipy.ndimage.zoom(order=1,
    mode='mirror', grid_mode=True), bit for bit, on random shapes (up- and down-sampling)."""
    from scipy import ndimage as ndi
    from magellanmapper_amd import preprocess
    rng = np.random.default_rng(4)
    for trial in range(80):
        # 'nearest' is what the reference's mode='edge' becomes for blocks with an axis of length 1
        mode = "mirror" if trial < 40 else "nearest"
        shp = tuple(int(v) for v in rng.integers(2, 11, 3))
        if mode == "nearest" and trial % 2:
            shp = tuple(1 if a == trial % 3 else v for a, v in enumerate(shp))
        out = tuple(max(1, int(s * f)) for s, f in zip(shp, rng.uniform(0.6, 3.3, 3)))
        img = rng.random(shp) * 3 - 1 if trial % 2 else rng.integers(0, 60000, shp).astype(np.float64)
        want = ndi.zoom(img, [o / i for o, i in zip(out, shp)], order=1, mode=mode, grid_mode=True)
        if want.shape != out:
            continue
        (iz, wz), (iy, wy), (ix, wx) = (preprocess.zoom_axis_table(i, o, mode) for i, o in zip(shp, out))
        acc = np.zeros(out)
        for a in range(2):
            for b in range(2):
                for c in range(2):
                    v = img[iz[:, a]][:, iy[:, b]][:, :, ix[:, c]]
                    acc = acc + ((v * wz[:, a, None, None]) * wy[None, :, b, None]) * wx[None, None, :, c]
        np.testing.assert_array_equal(acc, want, err_msg=str((mode, shp, out)))
    with pytest.raises(ValueError):
        preprocess.zoom_axis_table(1, 3)            # mirror needs two samples
    config.resolutions = [[3.0, 1.0, 1.0]]
    np.testing.assert_array_equal(preprocess.calc_isotropic_factor((0.96, 1, 1)), [2.88, 1.0, 1.0])
    assert preprocess.isotropic_shape((12, 26, 28), preprocess.calc_isotropic_factor((0.96, 1, 1))) == (34, 26, 28)
    # scikit-image's anti-aliasing Gaussian: sigma = (factor - 1) / 2, radius = int(4 sigma + 0.5)
    assert preprocess.Rescaler.aa_radius(20, 14) == ((20 / 14 - 1) / 2, 1)
    assert preprocess.Rescaler.aa_radius(20, 10) == (0.5, 2)
    assert preprocess.Rescaler.aa_radius(20, 19)[1] == 0 and preprocess.Rescaler.aa_radius(10, 30) == (0.0, 0)
    config.resolutions = None


def test_map_columns_native_matches_numpy():
    """Blobs.replace_rel_with_abs_blob_coords / remove_abs_blob_coords on a whole-stack-sized table
    (threaded native column copy, include/mmx.h: mmx_host_map_columns) against the plain NumPy form."""
    from magellanmapper_amd import detector
    rng = np.random.default_rng(5)
    for rows, width in ((20000, 11), (9000, 13), (100, 11)):
        a = rng.random((rows, width + 2))[:, :width] if width == 13 else rng.random((rows, width))
        b = a.copy()
        bb = detector.Blobs(a)
        bb.replace_rel_with_abs_blob_coords(a)
        out = bb.remove_abs_blob_coords(True)
        b[:, 0:3] = b[:, 7:10]
        assert np.array_equal(a, b)
        assert np.array_equal(out, b[:, [0, 1, 2, 3, 4, 5, 6, 10]])
        assert bb.cols == ["z", "y", "x", "radius", "confirmed", "truth", "channel", "region"]
    lib = _native.lib()
    cols = (ctypes.c_int32 * 2)(0, 99)
    t = np.zeros((4, 4))
    assert lib.mmx_host_map_columns(t.ctypes.data, 4, 4, cols, 2, t.ctypes.data, 4, 0) == 1     # bad column


def test_native_gathers_in_final_columns():
    """mmx_host_take_rows_final / mmx_host_gather_parts_by_key_final: out[i][j] = table[row][src_cols[j]], the abs
    coordinates written at abs_dst0 -- against NumPy, on a table large enough for the threaded path; bad arguments
    are refused."""
    from magellanmapper_amd import _native as nat
    L = nat.lib()
    rng = np.random.default_rng(21)
    n_table, n, ld = 50000, 30000, 14
    table = rng.random((n_table, ld))
    rows = rng.permutation(n_table)[:n].astype(np.int64)
    absz = rng.random((n_table, 3))
    src = [0, 1, 2, 3, 4, 5, 6, 10]
    csrc = (ctypes.c_int32 * len(src))(*src)
    out = np.empty((n, len(src)))
    nat.check(L.mmx_host_take_rows_final(table.ctypes.data, ld, rows.ctypes.data, n, csrc, len(src), absz.ctypes.data, 0,
                                         out.ctypes.data), "take_rows_final")
    want = table[rows][:, src]
    want[:, 0:3] = absz[rows]
    np.testing.assert_array_equal(out, want)
    # by key, from several lists: rows land in key order, equal keys in the order of the concatenated lists
    keys = rng.integers(0, 97, n).astype(np.int64)
    abs_rows = rng.random((n, 3))

    def gather(cuts, n_keys=97):
        edges = [0] + list(cuts) + [n]
        parts = [(rows[a:b], keys[a:b], abs_rows[a:b]) for a, b in zip(edges[:-1], edges[1:])]
        n_rows = np.array([len(p_[0]) for p_ in parts], dtype=np.int64)
        ptrs = [(ctypes.c_void_p * len(parts))(*[p_[c].ctypes.data for p_ in parts]) for c in range(3)]
        out2 = np.full((n, len(src)), np.nan)
        rc = L.mmx_host_gather_parts_by_key_final(table.ctypes.data, ld, len(parts), ptrs[0], ptrs[1], ptrs[2],
                                                  n_rows.ctypes.data, n_keys, csrc, len(src), 0, out2.ctypes.data, n)
        return rc, out2
    order = np.argsort(keys, kind="stable")
    want2 = table[rows[order]][:, src]
    want2[:, 0:3] = abs_rows[order]
    for cuts in ([], [1], [0, 0, 7000, 7001, 29999], sorted(rng.integers(0, n, 40))):
        rc, out2 = gather(cuts)
        assert rc == 0
        np.testing.assert_array_equal(out2, want2)
    # ... and equal to the two-step form: gather in the table's columns, then the column shuffles
    cols3 = (ctypes.c_int32 * 3)(7, 8, 9)
    full = np.empty((n, ld - 3))
    nat.check(L.mmx_host_gather_by_key(table.ctypes.data, ld, rows.ctypes.data, keys.ctypes.data, n, 97, ld - 3,
                                       abs_rows.ctypes.data, cols3, full.ctypes.data), "gather_by_key")
    full[:, 0:3] = full[:, 7:10]
    np.testing.assert_array_equal(out2, full[:, src])
    bad = (ctypes.c_int32 * 3)(0, 1, 99)
    assert L.mmx_host_take_rows_final(table.ctypes.data, ld, rows.ctypes.data, n, bad, 3, absz.ctypes.data, 0,
                                      out.ctypes.data) == 1
    assert L.mmx_host_take_rows_final(table.ctypes.data, ld, rows.ctypes.data, n, csrc, len(src), absz.ctypes.data, 6,
                                      out.ctypes.data) == 1                    # abs columns past the row
    assert gather([100], n_keys=5)[0] == 1                                       # key >= n_keys


def test_native_merge_by_key_reads_keys_in_place():
    """mmx_host_merge_by_key: the stable sort of rows by key (threaded counting sort), keys from an array or from the
    column behind the rows' own (how the ranks' survivors arrive); keys out of range are refused."""
    from magellanmapper_amd import _native as nat
    L = nat.lib()
    rng = np.random.default_rng(8)
    for n in (0, 1, 777, 60000):
        n_cols, n_keys = 8, 500
        rows = rng.random((n, n_cols + 1))
        keys = rng.integers(0, n_keys, n).astype(np.int64)
        rows[:, n_cols] = keys
        want = rows[np.argsort(keys, kind="stable"), :n_cols]
        for kp in (keys.ctypes.data, None):
            out = np.full((n, n_cols), np.nan)
            assert L.mmx_host_merge_by_key(rows.ctypes.data, n_cols + 1, kp, n, n_keys, n_cols, out.ctypes.data) == 0
            np.testing.assert_array_equal(out, want)
    out = np.empty((n, n_cols))
    assert L.mmx_host_merge_by_key(rows.ctypes.data, n_cols + 1, None, n, 100, n_cols, out.ctypes.data) == 1     # key >= n_keys
    rows[5, n_cols] = -1.0
    assert L.mmx_host_merge_by_key(rows.ctypes.data, n_cols + 1, None, n, n_keys, n_cols, out.ctypes.data) == 1
    rows[5, n_cols] = np.nan
    assert L.mmx_host_merge_by_key(rows.ctypes.data, n_cols + 1, None, n, n_keys, n_cols, out.ctypes.data) == 1
    # (no room for a key column behind the rows' own: a key array is required)
    assert L.mmx_host_merge_by_key(rows.ctypes.data, n_cols + 1, None, n, n_keys, n_cols + 1, out.ctypes.data) == 1


def test_native_merge_of_padded_blocks_by_key():
    """mmx_host_merge_parts_by_key: the stable sort by key of the CONCATENATION of row blocks that are not contiguous
    (every rank's survivors in an all_gather's padded receive buffer), from two to many blocks, empty ones included."""
    from magellanmapper_amd import _native as nat
    L = nat.lib()
    rng = np.random.default_rng(9)
    n_cols, n_keys = 8, 700
    for counts in ([5], [0, 3, 0], [30000, 41000], [9000, 0, 12000, 7000, 1, 15000, 8000, 11000]):
        most = max(counts)
        blocks = rng.random((len(counts), most, n_cols + 1))
        blocks[:, :, n_cols] = rng.integers(0, n_keys, (len(counts), most))
        live = np.concatenate([blocks[r, :c] for r, c in enumerate(counts)])
        want = live[np.argsort(live[:, n_cols].astype(np.int64), kind="stable"), :n_cols]
        n_rows = np.array(counts, dtype=np.int64)
        ptrs = (ctypes.c_void_p * len(counts))(*[blocks.ctypes.data + r * blocks.strides[0] for r in range(len(counts))])
        out = np.full((len(live), n_cols), np.nan)
        assert L.mmx_host_merge_parts_by_key(ptrs, n_rows.ctypes.data, len(counts), n_cols + 1, n_keys, n_cols,
                                             out.ctypes.data, len(live)) == 0
        np.testing.assert_array_equal(out, want)
    assert L.mmx_host_merge_parts_by_key(ptrs, n_rows.ctypes.data, len(counts), n_cols + 1, n_keys, n_cols,
                                         out.ctypes.data, len(live) - 1) == 1            # wrong output size
    assert L.mmx_host_merge_parts_by_key(ptrs, n_rows.ctypes.data, len(counts), n_cols + 1, 100, n_cols,
                                         out.ctypes.data, len(live)) == 1                # key >= n_keys


def test_native_emit_of_several_survivor_lists_at_once():
    """mmx_host_emit_parts_final == mmx_host_emit_survivors_final list by list, back to back (threaded over the
    concatenation: lists of any length, empty ones included)."""
    from magellanmapper_amd import _native as nat
    L = nat.lib()
    rng = np.random.default_rng(10)
    n_table, ld = 90000, 14
    table = rng.random((n_table, ld))
    src = [0, 1, 2, 3, 4, 5, 6, 10]
    csrc = (ctypes.c_int32 * len(src))(*src)
    for counts in ([7], [0, 0], [4000, 0, 1, 25000, 9000, 12000]):
        lists = [(rng.integers(0, n_table, c).astype(np.int64), rng.integers(0, 500, c).astype(np.int64), rng.random((c, 3)))
                 for c in counts]
        want = []
        for ids, keys, ab in lists:
            o = np.empty((len(ids), len(src) + 1))
            if len(ids):
                nat.check(L.mmx_host_emit_survivors_final(table.ctypes.data, ld, ids.ctypes.data, keys.ctypes.data, len(ids),
                                                          csrc, len(src), ab.ctypes.data, 0, o.ctypes.data), "emit")
            want.append(o)
        want = np.concatenate(want)
        n_rows = np.array(counts, dtype=np.int64)
        ptrs = [(ctypes.c_void_p * len(lists))(*[(d[c].ctypes.data if len(d[0]) else None) for d in lists]) for c in range(3)]
        out = np.full((len(want), len(src) + 1), np.nan)
        assert L.mmx_host_emit_parts_final(table.ctypes.data, ld, len(lists), ptrs[0], ptrs[1], ptrs[2], n_rows.ctypes.data,
                                           csrc, len(src), 0, out.ctypes.data, len(want)) == 0
        np.testing.assert_array_equal(out, want)
    assert L.mmx_host_emit_parts_final(table.ctypes.data, ld, len(lists), ptrs[0], ptrs[1], ptrs[2], n_rows.ctypes.data,
                                       csrc, len(src), 0, out.ctypes.data, len(want) + 1) == 1


def test_native_prune_works_in_a_forked_child():
    """The host thread pool lives in the dlopen'd library; after fork() its threads are gone (the reference's
    default start method is 'fork').  A table large enough for the threaded path must still prune in the child."""
    import multiprocessing as mp
    from magellanmapper_amd import _native as nat
    rng = np.random.default_rng(0)
    n = 60000
    table = rng.random((n, 6))
    rows = np.arange(n, dtype=np.int64)[::-1].copy()
    absz = rng.random((n, 3))
    cols = (ctypes.c_int32 * 3)(0, 1, 2)

    def take():
        out = np.empty((n, 5))
        nat.check(nat.lib().mmx_host_take_rows(table.ctypes.data, 6, rows.ctypes.data, n, 5, absz.ctypes.data, cols,
                                               out.ctypes.data), "take")
        return out

    want = take()                      # builds the pool (threads) in this process

    def child(q):
        q.put(bool(np.array_equal(take(), want)))

    ctx = mp.get_context("fork")
    q = ctx.Queue()
    p = ctx.Process(target=child, args=(q,))
    p.start()
    p.join(60)
    assert p.exitcode == 0, "the forked child hung or died in the native table code"
    assert q.get(timeout=5) is True


def test_region_threads_work_in_a_forked_child():
    """The executor the pruning regions run on is kept for the life of the process; a forked child (the reference's
    default start method) must not wait for the parent's threads: it makes its own."""
    import multiprocessing as mp
    from magellanmapper_amd import stack_detect as sd
    assert sd._region_workers().submit(lambda: 41 + 1).result(timeout=30) == 42        # threads exist in the parent

    def child(q):
        q.put(sd._region_workers().submit(lambda: os.getpid()).result(timeout=30))

    ctx = mp.get_context("fork")
    q = ctx.Queue()
    p = ctx.Process(target=child, args=(q,))
    p.start()
    p.join(60)
    assert p.exitcode == 0, "the forked child hung waiting for region threads it does not have"
    assert q.get(timeout=5) == p.pid


def test_get_mp_pool_follows_config():
    from magellanmapper_amd import chunking, config
    config.setup_roi_profiles(None)
    old = config.cpus
    config.cpus = 2
    try:
        with chunking.get_mp_pool() as pool:
            assert pool._processes == 2
            assert pool.map(abs, [-1, 2, -3]) == [1, 2, 3]
    finally:
        config.cpus = old


def test_native_assignment_solver_returns_scipys_optimum():
    """mmx_host_lsap (host code of the C ABI) on the reference's own cases (tests/golden/match.npz: integer
    coordinates, so tied distances; rectangular both ways) and on random tie-heavy matrices: the same pairs as
    scipy.optimize.linear_sum_assignment, not merely the same cost."""
    from scipy import optimize
    from scipy.spatial import distance
    from conftest import load_golden
    from magellanmapper_amd import verifier
    g = load_golden("match.npz")
    for k in range(int(g["n_lsap"])):
        sc = g["lsap%d_scaling" % k]
        cost = distance.cdist(g["lsap%d_a" % k][:, :3] * sc, g["lsap%d_b" % k][:, :3] * sc)
        rows, cols = verifier.linear_sum_assignment(cost)
        thresh = g["lsap%d_thresh" % k]
        keep = np.ones(len(rows), bool) if np.isnan(thresh) else cost[rows, cols] < thresh
        np.testing.assert_array_equal(rows[keep], g["lsap%d_rows" % k])
        np.testing.assert_array_equal(cols[keep], g["lsap%d_cols" % k])
    rng = np.random.default_rng(5)
    for trial in range(400):
        n, m = rng.integers(1, 30, 2)
        cost = rng.integers(0, 4, (n, m)).astype(float) if trial % 2 else distance.cdist(
            rng.integers(0, 6, (n, 3)).astype(float), rng.integers(0, 6, (m, 3)).astype(float))
        rows, cols = verifier.linear_sum_assignment(cost)
        want = optimize.linear_sum_assignment(cost)
        np.testing.assert_array_equal(rows, want[0])
        np.testing.assert_array_equal(cols, want[1])
    assert _native.lib().mmx_host_lsap(np.array([[np.nan]]).ctypes.data, 1, 1, None, None) == 1


def _check_match_blobs_roi(g, verifier):
    """``verifier.match_blobs_roi`` against the real function's five outputs (tests/golden/match.npz, ``mroi*``)."""
    for k in range(int(g["n_mroi"])):
        thresh, scaling, pad, _, _ = verifier.setup_match_blobs_roi(g["mroi%d_tol" % k])
        det, base = g["mroi%d_det" % k].copy(), g["mroi%d_base" % k].copy()
        got = verifier.match_blobs_roi(det, base, tuple(g["mroi%d_offset" % k]), tuple(g["mroi%d_size" % k]),
                                       thresh, scaling, pad)
        np.testing.assert_array_equal(got[0], g["mroi%d_inner_plus" % k])
        np.testing.assert_array_equal(got[1], g["mroi%d_truth_inner_plus" % k])
        np.testing.assert_array_equal(np.asarray(got[2], dtype=float), g["mroi%d_offset_inner" % k])
        np.testing.assert_array_equal(np.asarray(got[3], dtype=float), g["mroi%d_size_inner" % k])
        pairs = got[4].pairs
        np.testing.assert_array_equal(pairs.first.reshape(-1, 8), g["mroi%d_blob1" % k])
        np.testing.assert_array_equal(pairs.second.reshape(-1, 8), g["mroi%d_blob2" % k])
        np.testing.assert_array_equal(pairs.dist, g["mroi%d_dist" % k])
        # the caller's tables are not written to (the reference works on copies too)
        np.testing.assert_array_equal(det, g["mroi%d_det" % k])
        np.testing.assert_array_equal(base, g["mroi%d_base" % k])


def test_match_logic_on_the_host(monkeypatch):
    """The row-number bookkeeping of ``verifier.match_blobs_roi``, ``colocalizer.colocalize_blobs_match`` and the
    shortest-match de-duplication of ``StackColocalizer`` against fixtures from the real reference, with SciPy's
    ``cdist`` standing in for the device distance matrix (``mmx_cdist_f64`` is bit-equal to it, checked on the GPU);
    the native assignment solver is the product's."""
    from scipy.spatial import distance
    from conftest import load_golden
    from magellanmapper_amd import colocalizer, detector, verifier
    monkeypatch.setattr(verifier, "_cdist", lambda a, b: distance.cdist(a, b) if len(a) and len(b)
                        else np.zeros((len(a), len(b))))
    g = load_golden("match.npz")
    config.setup_roi_profiles(None)
    config.cpus = 2
    try:
        _check_match_blobs_roi(g, verifier)
        for name in ("stackA", "stackB"):
            config.roi_profile.update(segment_size=int(g[name + "_segment_size"]), num_sigma=3, denoise_size=None)
            config.resolutions = np.array([tuple(g[name + "_res"])])
            blobs = detector.Blobs(g[name + "_table"].copy())
            got = colocalizer.StackColocalizer.colocalize_stack(tuple(int(v) for v in g[name + "_shape"]), blobs)
            assert sorted(got) == [tuple(int(v) for v in key) for key in g[name + "_keys"]]
            for key, bm in got.items():
                df = bm.df
                assert list(df.columns) == [c.value for c in colocalizer.BlobMatch.Cols]
                np.testing.assert_array_equal(np.vstack(df["Blob1"]), g["%s_%d_%d_blob1" % (name, *key)])
                np.testing.assert_array_equal(np.vstack(df["Blob2"]), g["%s_%d_%d_blob2" % (name, *key)])
                np.testing.assert_array_equal(df["Distance"].to_numpy(), g["%s_%d_%d_dist" % (name, *key)])
                # a frame handed back in is read into the same table
                again = colocalizer.BlobMatch(df=df)
                np.testing.assert_array_equal(again.get_blobs(2), bm.get_blobs(2))
                np.testing.assert_array_equal(again.get_mean_coords(), bm.get_mean_coords())
        empty = colocalizer.BlobMatch([])
        assert empty.get_blobs(1) is None and empty.get_blobs_all() is None and len(empty.df) == 0
        assert colocalizer.BlobMatch().df is None and repr(colocalizer.BlobMatch()) == "Empty blob matches"
        bm = colocalizer.BlobMatch([(np.arange(8.), np.arange(8.) + 1, 1.5)], match_id=[7])
        assert bm.df["MatchID"].tolist() == [7] and bm.df["RoiID"].tolist() == [None]
    finally:
        config.resolutions = np.array([[1.0, 1.0, 1.0]])
        detector.Blobs(np.ones((1, 4))).format_blobs()


def test_plan_batches_ramp_and_taper():
    """Batches of a full volume: the first budget-sized batch is split into a ramp (the host has nothing to do until
    the first candidates arrive), the last into a taper (nothing hides the host work of the last batch); every block
    once, in order."""
    from magellanmapper_amd import blob_log as bl
    shapes = [(261, 261, 261)] * 256
    b = bl.plan_batches(shapes, 5, 64 << 30)
    n = [len(x) for x in b]
    assert n[:2] == [16, 32] and n[-4:] == [16, 10, 6, 4] and max(n) <= 89        # ramp ... taper
    assert sum(b, []) == list(range(256))
    b = bl.plan_batches(shapes, 5, 16 << 30)
    n = [len(x) for x in b]
    assert sum(b, []) == list(range(256)) and max(n) == 22 and n[-4:] == [16, 10, 6, 4]
    assert all(n[i] >= n[i + 1] for i in range(len(n) - 5, len(n) - 1))          # shrinking towards the end
    # blocks of different sizes: no batch may exceed the budget, whatever the taper merged
    mixed = [(261, 261, 261)] * 30 + [(261, 261, 64)] * 40 + [(261, 261, 261)] * 10
    per_vox = (4 + 5) * 4 + 3
    for batch in bl.plan_batches(mixed, 5, 4 << 30):
        slot = max(m[0] * m[1] * (-(-m[2] // 32) * 32) for m in (mixed[i] for i in batch))
        assert len(batch) == 1 or len(batch) * slot * per_vox <= (4 << 30)
    assert [len(x) for x in bl.plan_batches([(40, 40, 40)] * 9, 3, 1 << 30)] == [9]      # tiny blocks: one batch


def test_q16_error_bound_is_a_function_of_the_weights():
    """``mmx_tiled_q16_error_bound`` (host code, no GPU): the bound the 16-bit intermediates of the default path carry
    -- norm * (sum|w2| bound_P / 65535 + sum w0 bound_Q / 32767) / 2 plus the dropped low x low products of the X and of
    the Y pass, bound_Q the larger one-sided sum of the 2-D kernel w2 (x) w0 + w0 (x) w2 (what Q can reach for voxels in
    [0, 1]) -- restated in NumPy; 2.9e-5 for the benchmark's scales, an eighth of the band the host nominates raw integer
    volumes with."""
    from magellanmapper_amd import _native as nat, blob_log as bl, kernels1d as k1
    lib = nat.lib()
    for sigma in (1.0, 2.0, 3.0, 3.5, 4.0, 4.5, 5.0, 6.0):
        R = k1.kernel_radius(sigma)
        w0, w2 = k1.gaussian_half_kernel(sigma, 0, R), k1.gaussian_half_kernel(sigma, 2, R)
        got = lib.mmx_tiled_q16_error_bound(nat.as_double_ptr(w0), nat.as_double_ptr(w2), R, sigma * sigma)
        s0 = w0[0] + 2 * w0[1:].sum()
        s2 = abs(w2[0]) + 2 * np.abs(w2[1:]).sum()
        full0, full2 = np.concatenate((w0[:0:-1], w0)), np.concatenate((w2[:0:-1], w2))
        kern = np.outer(full2, full0) + np.outer(full0, full2)
        bp, bq = s0 * s0 * (1 + 1e-6), max(kern[kern > 0].sum(), -kern[kern < 0].sum()) * (1 + 1e-4)
        assert 0.36 * 2 * s2 * s0 < bq < 0.62 * 2 * s2 * s0          # about half of the two-sided sum rounds 3-5 used
        drop = 255.0 / 65536.0 / 2048.0
        biased = 4.0 * 2.0 ** -22          # X accumulators that carry the voxel pieces' exponent offsets
        ydrop = 255.0 / 4096.0             # Y pass on the matrix cores: low byte of a count x (weight - float16(weight))
        want = sigma * sigma * (s2 * (bp / 65535 * (0.5 + ydrop) + drop * s0 * s0 + biased * bp) +
                                s0 * (bq / 32767 * (0.5 + ydrop) + drop * 2 * s2 * s0 + biased * bq)) + 1e-6
        assert abs(got - want) < 1e-11, sigma
        assert 2.5e-5 < got < 5.5e-5 and 4 * got <= bl.EPS_REL_Q16, (sigma, got)
    assert lib.mmx_tiled_q16_error_bound(None, None, 3, 1.0) < 0
    # every radius the fast kernels take, sigma swept across each: the constant the host quotes holds from radius 4 on;
    # radii 1..3 carry up to 5.4e-5, four times which still fits the band
    worst = {}
    for R in range(1, nat.MMX_MAX_RADIUS_FAST + 1):
        for sigma in np.linspace(max(0.06, (R - 0.5) / 4.0), (R + 0.5) / 4.0, 40):
            if k1.kernel_radius(sigma) != R:
                continue
            w0, w2 = k1.gaussian_half_kernel(sigma, 0, R), k1.gaussian_half_kernel(sigma, 2, R)
            b = lib.mmx_tiled_q16_error_bound(nat.as_double_ptr(w0), nat.as_double_ptr(w2), R, sigma * sigma)
            worst[R] = max(worst.get(R, 0.0), b)
    assert len(worst) == nat.MMX_MAX_RADIUS_FAST
    assert all(b <= bl.Q16_BOUND_ANY_SIGMA for R, b in worst.items() if R >= 4), worst
    assert all(4 * b <= bl.EPS_REL_Q16 for R, b in worst.items())
    assert max(worst.values()) < 5.4e-5


# ---------------------------------------------------------------- native per-batch host work (mmx_host.cpp)
def _random_candidates(rng, shapes, ns, n_per_block, tie_every=0):
    """A candidate table as the device leaves it: random voxels (each once), float64 values, float32 stand-ins,
    some contested with probes appended (band = index of their candidate)."""
    from magellanmapper_amd import _native as nat
    recs, probes = [], []
    for slot, shp in enumerate(shapes):
        cube = int(np.prod(shp)) * ns
        lin = rng.choice(cube, size=min(n_per_block, cube), replace=False)
        s = lin % ns
        x = (lin // ns) % shp[2]
        y = (lin // (ns * shp[2])) % shp[1]
        z = lin // (ns * shp[2] * shp[1])
        v = rng.random(len(lin)) * 0.5
        if tie_every:
            v[::tie_every] = 0.3125               # exact ties inside the block
        for k in range(len(lin)):
            recs.append((slot, s[k], z[k], y[k], x[k], 0, np.float32(v[k]), 0.0, v[k], 0))
    cands = np.array(recs, dtype=nat.CAND_DTYPE)
    rng.shuffle(cands)
    contested = rng.random(len(cands)) < 0.2
    cands["flags"][contested] = nat.MMX_CAND_CONTESTED
    for i in np.nonzero(contested)[0]:
        for _ in range(int(rng.integers(0, 4))):
            # a rival somewhere next to it: sometimes above, sometimes below the candidate
            probes.append((cands["slot"][i], 0, 0, 0, 0, nat.MMX_CAND_PROBE, 0.0, 0.0,
                           cands["v64"][i] + rng.choice([-0.01, 0.0, 0.01]), i))
    table = np.concatenate([cands, np.array(probes, dtype=nat.CAND_DTYPE)]) if probes else cands
    return table, len(cands)


def _resolve_with_numpy(table, n_cands, shapes, ns, thr):
    """The rules of blob_log._resolve_peaks on a fully re-scored table, as plain NumPy per block."""
    c, p = table[:n_cands], table[n_cands:]
    rival = np.full(n_cands, -np.inf)
    np.maximum.at(rival, p["band"].astype(np.int64), p["v64"])
    dims = np.array([shapes[k] for k in c["slot"]]).reshape(-1, 3)
    border = ((c["s"] == 0) | (c["s"] == ns - 1) | (c["z"] == 0) | (c["z"] == dims[:, 0] - 1) | (c["y"] == 0) |
              (c["y"] == dims[:, 1] - 1) | (c["x"] == 0) | (c["x"] == dims[:, 2] - 1))
    rival[border] = np.maximum(rival[border], 0.0)
    keep = c["v64"] > thr
    cont = (c["flags"] & 1) != 0
    keep[cont] &= c["v64"][cont] >= rival[cont]
    out = []
    for b, shp in enumerate(shapes):
        rows = c[keep & (c["slot"] == b)]
        if len(rows) == int(np.prod(shp)) * ns and len(rows) > 1:
            rows = rows[:0]
        lin = ((rows["z"].astype(np.int64) * shp[1] + rows["y"]) * shp[2] + rows["x"]) * ns + rows["s"]
        rows = rows[np.argsort(lin)]
        rank = np.argsort(-rows["v64"])
        out.append((np.stack([rows[f][rank] for f in ("z", "y", "x", "s")], axis=1).astype(np.int64).reshape(-1, 4),
                    rows["v64"][rank]))
    return out


def test_native_peak_resolution_matches_the_numpy_rules():
    """mmx_host_resolve_peaks (through blob_log._resolve_peaks_native): contested candidates against their probes,
    zero padding at the cube border, the threshold, np.nonzero order, descending order, exact ties handed to
    np.argsort, a constant cube, the float32 deviation check."""
    from magellanmapper_amd import _native as nat, blob_log as bl
    rng = np.random.default_rng(11)
    for trial in range(12):
        ns = int(rng.integers(1, 6))
        shapes = [tuple(int(v) for v in rng.integers(2, 9, 3)) for _ in range(int(rng.integers(1, 7)))]
        if trial == 3:
            shapes.append((2, 1, 2))            # every voxel of this one a candidate: constant cube -> no peaks
        table, n_c = _random_candidates(rng, shapes, ns, 10 ** 6 if trial == 3 else 40, tie_every=7 if trial % 3 == 0 else 0)
        if trial == 3:
            keep = (table["slot"] != len(shapes) - 1) | (table["flags"] & nat.MMX_CAND_PROBE != 0)
            keep[:n_c] |= True                  # keep all of the constant cube's voxels, uncontested
            table["flags"][:n_c][table["slot"][:n_c] == len(shapes) - 1] = 0
            table["v64"][:n_c][table["slot"][:n_c] == len(shapes) - 1] = 0.4
            table["v"][:n_c][table["slot"][:n_c] == len(shapes) - 1] = np.float32(0.4)
        blocks = np.zeros(len(shapes), dtype=nat.BLOCK_DTYPE)
        for k, shp in enumerate(shapes):
            blocks[k] = (0, shp[0], shp[1], shp[2], k, 32, 0)
        stats = bl.BatchStats()
        pb = bl._resolve_peaks_native(table, n_c, blocks, ns, 0.1, stats, 1e-3)
        want = _resolve_with_numpy(table, n_c, shapes, ns, 0.1)
        assert len(pb) == len(shapes)
        for b in range(len(shapes)):
            coords, vals = pb.block(b)
            np.testing.assert_array_equal(coords, want[b][0])
            np.testing.assert_array_equal(vals, want[b][1])
        assert stats.n_peaks == sum(len(w[1]) for w in want) and stats.n_probes == len(table) - n_c
        assert stats.n_contested == int(np.count_nonzero(table["flags"][:n_c] & 1))
    # float32 too far from float64: the caller widens the band; NaN: an error
    table, n_c = _random_candidates(rng, [(4, 4, 4)], 2, 10)
    blocks = np.zeros(1, dtype=nat.BLOCK_DTYPE)
    blocks[0] = (0, 4, 4, 4, 0, 32, 0)
    table["v"][0] += np.float32(0.01)
    with pytest.raises(bl._BandTooNarrow):
        bl._resolve_peaks_native(table, n_c, blocks, 2, 0.1, bl.BatchStats(), 1e-3)
    table["v64"][1] = np.nan
    with pytest.raises(nat.MmxError):
        bl._resolve_peaks_native(table, n_c, blocks, 2, 0.1, bl.BatchStats(), 1e-3)
    empty = bl._resolve_peaks_native(table[:0], 0, blocks, 2, 0.1, bl.BatchStats(), 1e-3)
    assert len(empty.coords) == 0 and list(empty.offsets) == [0, 0]


def _prune_with_numpy(allb, offsets, overlap):
    """skimage's _prune_blobs rule per block by brute force (every pair, in any order: only used where no blob both
    wins and loses, where the order cannot matter)."""
    from magellanmapper_amd import blob_log as bl
    sig = allb[:, 3].copy()
    chain = []
    for b in range(len(offsets) - 1):
        lo, hi = offsets[b], offsets[b + 1]
        losers, winners = set(), set()
        for i in range(lo, hi):
            for j in range(i + 1, hi):
                if bl._exact_overlap(allb[i], allb[j]) > overlap:
                    l, w = (j, i) if allb[i, 3] > allb[j, 3] else (i, j)
                    losers.add(l)
                    winners.add(w)
        if losers & winners:
            chain.append(b)
        else:
            sig[list(losers)] = 0
    return sig > 0, chain


def test_native_overlap_prune_matches_brute_force():
    """mmx_host_overlap_prune: crowded random peaks of several scales; closed blocks equal the brute-force rule, open
    blocks (a blob both wins and loses) are reported, and the full path (blob_log._prune_batch_native, which takes
    the reference's pair order for those) equals the array formulation fed with the same pairs."""
    from magellanmapper_amd import blob_log as bl
    rng = np.random.default_rng(12)
    space = bl.ScaleSpace.make(2.0, 5.0, 4)
    n_open = 0
    for trial in range(6):
        sizes = rng.integers(0, 60, int(rng.integers(1, 6)))
        offsets = np.concatenate(([0], np.cumsum(sizes))).astype(np.int32)
        n = int(offsets[-1])
        coords = np.empty((n, 4), dtype=np.int32)
        coords[:, :3] = rng.integers(0, 40, (n, 3))
        coords[:, 3] = rng.integers(0, 4, n)
        pb = bl.PeakBatch(coords, rng.random(n), offsets)
        stats = bl.BatchStats()
        pb = bl._prune_batch_native(pb, space, 0.5, stats)
        allb = coords.astype(np.float64)
        allb[:, 3] = space.sigmas[coords[:, 3]]
        want_alive, chain = _prune_with_numpy(allb, offsets, 0.5)
        n_open += len(chain)
        assert stats.n_order_fallbacks == len(chain)
        for b in range(len(sizes)):
            lo, hi = offsets[b], offsets[b + 1]
            if b not in chain:
                np.testing.assert_array_equal(pb.alive[lo:hi].astype(bool), want_alive[lo:hi], err_msg=f"{trial}/{b}")
            got = pb.blobs(b)
            assert got.shape == ((int(pb.alive[lo:hi].sum()), 4) if hi > lo else (0, 3))
        # open blocks: the same pairs through the array formulation alone
        sig = allb[:, 3].copy()
        pairs = np.array([(i, j) for b in range(len(sizes)) for i in range(offsets[b], offsets[b + 1])
                          for j in range(i + 1, offsets[b + 1])], dtype=np.int64).reshape(-1, 2)
        frac = np.array([bl._exact_overlap(allb[i], allb[j]) for i, j in pairs])
        sel = frac > 0.5 - bl.OVERLAP_BAND
        bl._apply_pairs(allb, sig, offsets, pairs[sel], frac[sel], 0.5, bl.BatchStats())
        np.testing.assert_array_equal(pb.alive.astype(bool), sig > 0)
    assert n_open > 0          # (the trials do exercise the order-dependent case)


def _big_peak_batch(rng, sizes, extent=110):
    from magellanmapper_amd import blob_log as bl
    offsets = np.concatenate(([0], np.cumsum(sizes))).astype(np.int32)
    n = int(offsets[-1])
    coords = np.empty((n, 4), dtype=np.int32)
    coords[:, :3] = rng.integers(0, extent, (n, 3))
    coords[:, 3] = rng.integers(0, 4, n)
    return bl.PeakBatch(coords, rng.random(n), offsets)


def test_native_overlap_prune_of_few_big_blocks_splits_each_block_over_threads():
    """A small stack's batch (2 blocks, thousands of peaks each): every block's pair search is dealt to several
    threads (mmx_host.cpp: `parts`).  The pairs found equal those of SciPy's cKDTree + the reference's overlap
    formula, and the flags are the sequential rule's on them."""
    from scipy.spatial import cKDTree
    from magellanmapper_amd import blob_log as bl
    rng = np.random.default_rng(77)
    space = bl.ScaleSpace.make(2.0, 5.0, 4)
    pb = _big_peak_batch(rng, [2600, 0, 3100])
    stats = bl.BatchStats()
    pb = bl._prune_batch_native(pb, space, 0.5, stats)
    allb = pb.coords.astype(np.float64)
    allb[:, 3] = space.sigmas[pb.coords[:, 3]]
    sig = allb[:, 3].copy()
    n_over = 0
    found = []
    for b in range(len(pb)):
        lo, hi = pb.offsets[b], pb.offsets[b + 1]
        if hi - lo < 2:
            continue
        for i, j in cKDTree(allb[lo:hi, :3]).query_pairs(2 * 5.0 * math.sqrt(3)):
            f = bl._exact_overlap(allb[lo + i], allb[lo + j])
            if f > 0.5 - bl.OVERLAP_BAND:
                found.append((lo + min(i, j), lo + max(i, j), f))
                n_over += 1
    assert n_over > 200 and stats.n_overlap_pairs == n_over
    pairs = np.array([(i, j) for i, j, _ in found], dtype=np.int64)
    frac = np.array([f for _, _, f in found])
    bl._apply_pairs(allb, sig, pb.offsets, pairs, frac, 0.5, bl.BatchStats())
    np.testing.assert_array_equal(pb.alive.astype(bool), sig > 0)


def test_host_thread_pool_gives_the_same_answer_whether_its_workers_poll_or_sleep():
    """The pool's workers poll for ~150 us after a section and sleep afterwards; the caller does the same while it
    waits.  Sections spaced from back-to-back to a millisecond apart, from two Python threads at once: every call
    returns what the first one did."""
    import threading
    import time as _time
    from magellanmapper_amd import blob_log as bl
    space = bl.ScaleSpace.make(2.0, 5.0, 4)
    base = _big_peak_batch(np.random.default_rng(5), [1300, 1200], extent=200)      # (sparse: no order-dependent blocks)
    want = bl._prune_batch_native(bl.PeakBatch(base.coords, base.vals, base.offsets), space, 0.5, bl.BatchStats())
    want_alive = want.alive.copy()
    assert 0 < want_alive.sum() < len(want_alive)
    errors = []

    def work(seed):
        rng = np.random.default_rng(seed)
        try:
            for k in range(400):
                st = bl.BatchStats()
                got = bl._prune_batch_native(bl.PeakBatch(base.coords, base.vals, base.offsets), space, 0.5, st)
                if not np.array_equal(got.alive, want_alive):
                    errors.append(f"call {k} of thread {seed}")
                    return
                pause = (0.0, 0.0, 5e-5, 2e-4, 1e-3)[int(rng.integers(0, 5))]
                if pause:
                    _time.sleep(pause)
        except Exception as exc:            # noqa: BLE001 -- reported below
            errors.append(repr(exc))

    threads = [threading.Thread(target=work, args=(s,)) for s in (1, 2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not any(t.is_alive() for t in threads), "a parallel section never finished"
    assert not errors, errors


def test_native_table_emit_matches_the_python_tables():
    """mmx_host_emit_tables: 11 columns + tags + compact copies, block offsets, border exclusion, capacity check --
    against the table building of detector.detect_blobs_blocks_device + StackDetector._finish_block."""
    import ctypes
    from magellanmapper_amd import _native as nat, detector
    rng = np.random.default_rng(13)
    sig = np.array([3.0, 3.5, 4.0])
    offsets = np.array([0, 5, 5, 12, 20], dtype=np.int32)
    n = 20
    coords = np.empty((n, 4), dtype=np.int32)
    coords[:, :3] = rng.integers(0, 30, (n, 3))
    coords[:, 3] = rng.integers(0, 3, n)
    alive = (rng.random(n) < 0.7).astype(np.uint8)
    boffs = rng.integers(0, 500, (4, 3)).astype(np.float64)
    tags = rng.integers(0, 8, (4, 3)).astype(np.int32)
    interior = np.array([[2, 2, 2, 28, 28, 28]] * 4, dtype=np.int32)
    detector.Blobs(np.ones((1, 4))).format_blobs()      # bind the class-level column registry to the 11 columns
    for use_interior in (False, True):
        ld, row0, cap = 16, 3, 40
        store = np.full((cap, ld), 7.5)
        zyx = np.zeros((cap, 3), np.int32)
        tag = np.zeros((cap, 3), np.int32)
        absz = np.zeros((cap, 3))
        per = np.zeros(4, dtype=np.int64)
        rc = nat.lib().mmx_host_emit_tables(coords.ctypes.data, alive.ctypes.data, offsets.ctypes.data, 4,
                                            sig.ctypes.data, 3, 1.0, boffs.ctypes.data, tags.ctypes.data,
                                            interior.ctypes.data if use_interior else None, store.ctypes.data, ld,
                                            zyx.ctypes.data, tag.ctypes.data, absz.ctypes.data, row0, cap,
                                            per.ctypes.data)
        assert rc == 0
        at = row0
        for b in range(4):
            rows = coords[offsets[b]:offsets[b + 1]][alive[offsets[b]:offsets[b + 1]].astype(bool)]
            tbl = rows.astype(np.float64)
            tbl[:, 3] = sig[rows[:, 3]]
            big = np.empty((len(rows), 11))
            big[:, :4] = tbl
            big[:, 3] = big[:, 3] * np.sqrt(3)
            big[:, 4:6] = -1
            big[:, 6] = 1.0
            big[:, 7:10] = big[:, :3]
            big[:, 10] = -1
            if use_interior:
                big = detector.get_blobs_interior(big, (30, 30, 30), (2, 2, 2), (2, 2, 2))
            detector.Blobs.shift_blob_rel_coords(big, boffs[b])
            detector.Blobs.shift_blob_abs_coords(big, boffs[b])
            assert per[b] == len(big)
            np.testing.assert_array_equal(store[at:at + len(big), :11], big)
            np.testing.assert_array_equal(store[at:at + len(big), 11:13], 7.5)          # extra columns untouched
            np.testing.assert_array_equal(store[at:at + len(big), 13:], np.broadcast_to(tags[b], (len(big), 3)))
            np.testing.assert_array_equal(zyx[at:at + len(big)], big[:, :3].astype(np.int32))
            np.testing.assert_array_equal(absz[at:at + len(big)], big[:, 7:10])
            np.testing.assert_array_equal(tag[at:at + len(big)], np.broadcast_to(tags[b], (len(big), 3)))
            at += len(big)
        np.testing.assert_array_equal(store[:row0], 7.5)
        np.testing.assert_array_equal(store[at:], 7.5)
        assert nat.lib().mmx_host_emit_tables(
            coords.ctypes.data, alive.ctypes.data, offsets.ctypes.data, 4, sig.ctypes.data, 3, 1.0, boffs.ctypes.data,
            tags.ctypes.data, None, store.ctypes.data, ld, zyx.ctypes.data, tag.ctypes.data, absz.ctypes.data, row0,
            row0 + 1, per.ctypes.data) == 4          # MMX_ERR_WORKSPACE


# ---------------------------------------------------------------- region-wise pruning (stack_detect._RegionPruner)
def _synthetic_block_tables(rng, shape, blocks, n_blobs, channels=(0,), jitter=3, n_extra=0):
    """Per-block tables as a detection leaves them: every block reports the blobs of a common field that fall into
    its extent, each with its own jitter (so the overlaps hold near-duplicates, some within the pruning tolerance,
    some just outside it), coordinates in the ROI frame."""
    field = {c: rng.integers(0, shape, (n_blobs, 3)) for c in channels}
    tables = {}
    for coord in np.ndindex(*blocks.sub_roi_slices.shape):
        ext = [s.indices(n)[:2] for s, n in zip(blocks.sub_roi_slices[coord], shape)]
        parts = []
        for c in channels:
            pts = field[c] + rng.integers(-jitter, jitter + 1, field[c].shape)
            inside = np.all([(pts[:, a] >= ext[a][0]) & (pts[:, a] < ext[a][1]) for a in range(3)], axis=0)
            pts = pts[inside]
            t = np.full((len(pts), 11 + n_extra), -1.0)
            t[:, :3] = pts
            t[:, 3] = 5.2
            t[:, 6] = c
            t[:, 7:10] = pts
            if n_extra:
                t[:, 11:] = rng.integers(0, 2, (len(pts), n_extra))
            parts.append(t)
        tbl = np.concatenate(parts)
        tables[coord] = tbl if len(tbl) else None
    return tables


@pytest.mark.parametrize("case", ["one_channel", "two_channels_extra_columns", "thin_last_blocks", "one_axis"])
def test_region_wise_pruning_equals_the_whole_table_passes(case, monkeypatch):
    """The three passes run region by region (own rows + a halo of neighbouring rows) as blocks land, then merged
    by key, against the same passes over the whole table: identical rows in identical order, identical averaged
    coordinates, identical pruning-ratio statistics."""
    from magellanmapper_amd import stack_detect as sd
    rng = np.random.default_rng({"one_channel": 21, "two_channels_extra_columns": 22, "thin_last_blocks": 23,
                                 "one_axis": 24}[case])
    shape = {"thin_last_blocks": (80, 140, 110), "one_axis": (40, 40, 300)}.get(case, (96, 150, 170))
    channels = [0, 1] if case == "two_channels_extra_columns" else [0]
    n_extra = 2 if case == "two_channels_extra_columns" else 0
    config.resolutions = np.array([[1.0, 1.0, 1.0]])
    config.setup_roi_profiles(None)
    config.roi_profile.update(segment_size=32 if case != "one_axis" else 50, denoise_size=None)
    blocks = sd.setup_blocks(config.roi_profile, shape)
    tables = _synthetic_block_tables(rng, shape, blocks, 9000 if case != "one_axis" else 1500, channels, n_extra=n_extra)
    grid = blocks.sub_roi_slices.shape
    share = list(range(int(np.prod(grid))))
    coords = list(np.ndindex(*grid))

    before_last = []

    def build(with_pruner):
        arena = sd._TableArena(11 + n_extra, len(share))
        pruner = None
        if with_pruner:
            plan = sd.StackPruner._axis_plan(shape, blocks.overlap, blocks.tol, blocks.overlap_padding,
                                             blocks.sub_roi_slices, blocks.sub_rois_offsets)
            pruner = sd._RegionPruner(arena, plan, channels, blocks.sub_roi_slices, shape, share)
        seg = np.zeros(grid, dtype=object).view(sd._SegRois)
        for k in share:
            if pruner is not None and k == share[-1]:
                before_last.append(len(pruner.regions) - len(pruner.pending))
            tbl = tables[coords[k]]
            if tbl is not None:
                arena.add(coords[k], tbl)
                tbl = arena.view(coords[k])
            arena.landed()
            seg[coords[k]] = tbl
            if pruner is not None:
                pruner.advance()
        # (the arena moves to larger arrays as it grows: the views are taken once everything has landed, as
        #  StackDetector.assemble_seg_rois does -- otherwise prune_blobs_mp finds the arena no longer intact and prunes
        #  the whole table, which is not what this test is about)
        for k in share:
            if seg[coords[k]] is not None:
                seg[coords[k]] = arena.view(coords[k])
        assert arena.intact(seg)
        seg.arena, seg.pruner = arena, pruner
        return seg, pruner

    class Img:
        pass
    Img.shape = shape
    seg_a, _ = build(False)
    want, df_want = sd.StackPruner.prune_blobs_mp(Img, seg_a, blocks.overlap, blocks.tol, blocks.sub_roi_slices,
                                                   blocks.sub_rois_offsets, channels, blocks.overlap_padding)
    seg_b, pruner = build(True)
    merges = []
    merge = sd._RegionPruner.finish
    monkeypatch.setattr(sd._RegionPruner, "finish", lambda self, cols, final=None, lap=lambda what: None: (merges.append(self), merge(self, cols, final, lap))[1])
    got, df_got = sd.StackPruner.prune_blobs_mp(Img, seg_b, blocks.overlap, blocks.tol, blocks.sub_roi_slices,
                                                 blocks.sub_rois_offsets, channels, blocks.overlap_padding)
    assert merges == [pruner] and not pruner.pending and all(d is not None for d in pruner.done)      # (it was used)
    if len(pruner.regions) > 4:
        assert 0 < before_last[0] < len(pruner.regions)      # some regions early, the last ones once everything landed
    assert len(want) < sum(len(t) for t in tables.values() if t is not None)                    # duplicates were removed
    np.testing.assert_array_equal(got, want)
    assert list(df_got.columns) == list(df_want.columns)
    np.testing.assert_array_equal(df_got.to_numpy(), df_want.to_numpy())
    # final_form: the table comes back in the columns the reference's last two steps leave (rel <- abs, abs dropped),
    # from the whole-table gather and from the regions' merge alike -- except with co-localisation columns behind the
    # named ones, which those steps still have to read
    def final_of(table):
        bb = detector.Blobs(table.copy())
        bb.replace_rel_with_abs_blob_coords(bb.blobs)
        return bb.remove_abs_blob_coords(True), list(bb.cols)
    want_final, want_cols = final_of(want)
    for with_pruner in (False, True):
        seg_f, _ = build(with_pruner)
        got_f, df_f = sd.StackPruner.prune_blobs_mp(Img, seg_f, blocks.overlap, blocks.tol, blocks.sub_roi_slices,
                                                     blocks.sub_rois_offsets, channels, blocks.overlap_padding,
                                                     final_form=True, untouched=with_pruner)
        np.testing.assert_array_equal(df_f.to_numpy(), df_want.to_numpy())
        if n_extra:
            assert not isinstance(got_f, sd._FinalTable)
            np.testing.assert_array_equal(got_f, want)
            # ... unless they are announced: the eight final columns as one contiguous table and, beside it, the columns
            # the reference reads its flags from (`segments_all[:, 10:10 + C]`, stack_detect.py:463-464)
            seg_g, _ = build(with_pruner)
            got_g, df_g = sd.StackPruner.prune_blobs_mp(Img, seg_g, blocks.overlap, blocks.tol, blocks.sub_roi_slices,
                                                         blocks.sub_rois_offsets, channels, blocks.overlap_padding,
                                                         final_form=True, untouched=with_pruner, n_flag_cols=n_extra)
            assert isinstance(got_g, sd._FinalTable) and got_g.col_names == want_cols
            assert got_g.view(np.ndarray).flags["C_CONTIGUOUS"] and got_g.shape[1] == 8
            np.testing.assert_array_equal(got_g.view(np.ndarray), want_final)
            np.testing.assert_array_equal(got_g.coloc_cols, want[:, 10:10 + n_extra])
            np.testing.assert_array_equal(df_g.to_numpy(), df_want.to_numpy())
        else:
            assert isinstance(got_f, sd._FinalTable) and got_f.col_names == want_cols and got_f.coloc_cols is None
            np.testing.assert_array_equal(got_f.view(np.ndarray), want_final)
    # other parameters than planned for: the regions are ignored, the whole table is pruned
    seg_c, pruner_c = build(True)
    tol2 = np.asarray(blocks.tol) - 1
    got2, _ = sd.StackPruner.prune_blobs_mp(Img, seg_c, blocks.overlap, tol2, blocks.sub_roi_slices,
                                            blocks.sub_rois_offsets, channels, blocks.overlap_padding)
    want2, _ = sd.StackPruner.prune_blobs_mp(Img, build(False)[0], blocks.overlap, tol2, blocks.sub_roi_slices,
                                             blocks.sub_rois_offsets, channels, blocks.overlap_padding)
    np.testing.assert_array_equal(got2, want2)
    assert not pruner_c._futures and not pruner_c.pending            # cancelled: nothing of it left queued or running
    # tables edited IN PLACE between detection and pruning (the reference's API allows it: they are plain arrays):
    # neither the regions pruned ahead nor the arena's compact columns may be used -- the result is that of pruning
    # the edited tables
    seg_d, pruner_d = build(True)
    for k in share:
        if seg_d[coords[k]] is not None:
            seg_d[coords[k]][:, [0, 7]] += 2.0           # every blob two planes deeper, rel and abs
    got3, _ = sd.StackPruner.prune_blobs_mp(Img, seg_d, blocks.overlap, blocks.tol, blocks.sub_roi_slices,
                                            blocks.sub_rois_offsets, channels, blocks.overlap_padding)
    seg_e = np.zeros(grid, dtype=object)
    for k in share:
        t = tables[coords[k]]
        if t is not None:
            t = t.copy()
            t[:, [0, 7]] += 2.0
        seg_e[coords[k]] = t
    want3, _ = sd.StackPruner.prune_blobs_mp(Img, seg_e, blocks.overlap, blocks.tol, blocks.sub_roi_slices,
                                             blocks.sub_rois_offsets, channels, blocks.overlap_padding)
    np.testing.assert_array_equal(got3, want3)
    assert not pruner_d._futures and not np.array_equal(got3, want)


def test_cancelled_region_pruner_surfaces_a_region_failure(monkeypatch):
    """A region that raised while pruning ahead is not lost with its future: ``cancel`` (what ``prune_blobs_mp`` calls
    when it cannot use the regions) re-raises it, and leaves nothing of the pruner queued or running either way."""
    from magellanmapper_amd import _native as nat, stack_detect as sd
    config.resolutions = np.array([[1.0, 1.0, 1.0]])
    config.setup_roi_profiles(None)
    config.roi_profile.update(segment_size=32, denoise_size=None)
    shape = (70, 70, 70)
    blocks = sd.setup_blocks(config.roi_profile, shape)
    tables = _synthetic_block_tables(np.random.default_rng(5), shape, blocks, 800)
    grid = blocks.sub_roi_slices.shape
    coords = list(np.ndindex(*grid))
    plan = sd.StackPruner._axis_plan(shape, blocks.overlap, blocks.tol, blocks.overlap_padding, blocks.sub_roi_slices,
                                     blocks.sub_rois_offsets)
    arena = sd._TableArena(11, len(coords))
    pruner = sd._RegionPruner(arena, plan, [0], blocks.sub_roi_slices, shape, list(range(len(coords))))

    def boom(self, i):
        raise nat.MmxError("region %d failed" % i)
    monkeypatch.setattr(sd._RegionPruner, "_run", boom)
    for c in coords:
        if tables[c] is not None:
            arena.add(c, tables[c])
        arena.landed()
        pruner.advance()
    assert pruner._futures
    import concurrent.futures
    concurrent.futures.wait(pruner._futures[:1], timeout=30)      # (a region HAS run and raised; others may not have started)
    with pytest.raises(nat.MmxError, match="region"):
        pruner.cancel()
    assert not pruner._futures and not pruner.pending
    pruner.cancel()         # idempotent


def test_overlap_prune_reproduces_scikit_image_on_every_fixture():
    """A5 on the host alone: the ordered raw peaks the real ``blob_log`` found (fixtures) through the native pair
    search and the sequential rule -- for blocks where a blob both wins and loses, in the order
    ``cKDTree.query_pairs`` + set iteration give under the INSTALLED SciPy / Python (``_reference_pair_order``) --
    must leave exactly the blobs the real scikit-image left, in its order.  Pins the one place where the product
    path leans on an implementation-defined order (fixtures: SciPy 1.7.1 / Python 3.9)."""
    import glob
    from conftest import GOLDEN
    from magellanmapper_amd import blob_log as bl
    n_chain = n_cases = 0
    for path in sorted(glob.glob(os.path.join(GOLDEN, "bloblog_*.npz"))):
        g = np.load(path, allow_pickle=True)
        peaks = np.asarray(g["peaks"]).reshape(-1, 4).astype(np.int32)
        space = bl.ScaleSpace.make(float(g["min_sigma"]), float(g["max_sigma"]), int(g["num_sigma"]))
        pb = bl.PeakBatch(np.ascontiguousarray(peaks), np.zeros(len(peaks)), np.array([0, len(peaks)], dtype=np.int32))
        stats = bl.BatchStats()
        pb = bl._prune_batch_native(pb, space, float(g["overlap"]), stats)
        got = pb.blobs(0)
        want = np.asarray(g["pruned"])
        assert got.shape == want.shape, os.path.basename(path)
        np.testing.assert_array_equal(got, want, err_msg=os.path.basename(path))
        n_chain += stats.n_order_fallbacks
        n_cases += 1
    assert n_cases >= 10
    # crowded random tables through the real skimage.feature.blob._prune_blobs (tests/golden/overlap_prune.npz): here
    # the outcome does depend on the visiting order
    g = load_golden("overlap_prune.npz")
    space = bl.ScaleSpace.make(3.0, 5.0, 5)
    np.testing.assert_array_equal(space.sigmas, g["sigmas"])
    import scipy
    for k in range(int(g["n_cases"])):
        coords = np.ascontiguousarray(g["case%d_coords" % k], dtype=np.int32)
        pb = bl.PeakBatch(coords, np.zeros(len(coords)), np.array([0, len(coords)], dtype=np.int32))
        stats = bl.BatchStats()
        pb = bl._prune_batch_native(pb, space, float(g["case%d_overlap" % k]), stats)
        same = pb.blobs(0).shape == g["case%d_kept" % k].shape and np.array_equal(pb.blobs(0), g["case%d_kept" % k])
        if not same and scipy.__version__ not in bl.VERIFIED_SCIPY:
            # the order is SciPy's to define: a release nobody has checked may legitimately differ from the fixture's
            pytest.skip(f"SciPy {scipy.__version__} orders the pairs of a pruning chain differently from the releases "
                        f"the fixture was checked with {bl.VERIFIED_SCIPY}: the product then follows THIS SciPy, as the "
                        "reference would on this machine; the fixture cannot say which is right")
        assert same, "case %d" % k
        n_chain += stats.n_order_fallbacks
    assert n_chain >= 3


def test_save_subimage_writes_the_roi_unless_it_is_the_open_memmap(tmp_path, caplog):
    """``config.save_subimg`` (reference stack_detect.py:477-489): the ROI as ``<base>_subimg.npy``; when the image is
    a memory map of that very file it is left alone with a warning."""
    roi = np.arange(2 * 3 * 4, dtype=np.uint16).reshape(2, 3, 4)
    path = str(tmp_path / "img_subimg.npy")
    stack_detect._save_subimage(path, roi[None], roi)
    np.testing.assert_array_equal(np.load(path), roi)
    mapped = np.load(path, mmap_mode="r+")
    assert isinstance(mapped, np.memmap)
    before = os.path.getmtime(path)
    with caplog.at_level("WARNING", logger="magellanmapper_amd"):
        stack_detect._save_subimage(path, mapped, np.zeros((1, 1, 1), dtype=np.uint16))
    assert "currently open" in caplog.text and os.path.getmtime(path) == before
    np.testing.assert_array_equal(np.load(path), roi)
    with pytest.raises(TypeError):
        stack_detect._save_subimage(str(tmp_path / "x.npy"), roi[None], object())


def test_rank_table_plumbing_natives():
    """``mmx_host_rows_in_boxes`` / ``mmx_host_append_rows`` / ``mmx_host_emit_survivors`` (the table plumbing of the
    distributed pruning, host code only) against their NumPy statements, incl. the overflow counts and bad arguments."""
    from magellanmapper_amd import _native as nat
    lib = nat.lib()
    rng = np.random.default_rng(3)
    n, ncol = 50000, 14                     # (enough rows for the threaded forms)
    store = rng.normal(size=(n, ncol))
    zyx = rng.integers(0, 200, (n, 3)).astype(np.int32)
    tag = rng.integers(0, 5, (n, 3)).astype(np.int32)
    ab = zyx + rng.integers(-2, 3, (n, 3)).astype(np.float64)
    store[:, 6] = rng.integers(0, 2, n)
    lo = np.array([[0, 0, 0], [150, 20, 30]], dtype=np.int32)
    hi = np.array([[40, 200, 200], [200, 90, 160]], dtype=np.int32)
    inside = np.zeros(n, dtype=bool)
    for b in range(2):
        inside |= np.all((zyx >= lo[b]) & (zyx < hi[b]), axis=1)
    want = np.column_stack((zyx[inside], tag[inside], ab[inside], store[inside, 6])).astype(np.float64)
    out = np.empty((n, 10))
    k = ctypes.c_int64(0)
    nat.check(lib.mmx_host_rows_in_boxes(zyx.ctypes.data, tag.ctypes.data, ab.ctypes.data, store.ctypes.data + 48,
                                         ncol, n, lo.ctypes.data, hi.ctypes.data, 2, out.ctypes.data, n,
                                         ctypes.byref(k)), "rows_in_boxes")
    assert k.value == len(want) > 100
    np.testing.assert_array_equal(out[:k.value], want)
    small = np.empty((7, 10))                   # too small: filled to its capacity, the count keeps counting
    nat.check(lib.mmx_host_rows_in_boxes(zyx.ctypes.data, tag.ctypes.data, ab.ctypes.data, store.ctypes.data + 48,
                                         ncol, n, lo.ctypes.data, hi.ctypes.data, 2, small.ctypes.data, 7,
                                         ctypes.byref(k)), "rows_in_boxes")
    assert k.value == len(want)
    np.testing.assert_array_equal(small, want[:7])
    assert lib.mmx_host_rows_in_boxes(None, None, None, None, ncol, n, lo.ctypes.data, hi.ctypes.data, 2,
                                      out.ctypes.data, n, ctypes.byref(k)) == 1
    # ... and back into compact columns behind `at` rows, filtered by the receiver's box
    box_lo = np.array([10, 0, 0], dtype=np.int32)
    box_hi = np.array([180, 150, 150], dtype=np.int32)
    keep = np.all((want[:, :3] >= box_lo) & (want[:, :3] < box_hi), axis=1)
    at, cap = 11, 11 + int(keep.sum())
    z2 = np.zeros((cap, 3), dtype=np.int32)
    t2 = np.zeros((cap, 3), dtype=np.int32)
    a2 = np.zeros((cap, 3))
    st2 = np.zeros((cap, ncol))
    p32 = ctypes.POINTER(ctypes.c_int32)
    nat.check(lib.mmx_host_append_rows(want.ctypes.data, len(want), box_lo.ctypes.data_as(p32), box_hi.ctypes.data_as(p32),
                                       z2.ctypes.data, t2.ctypes.data, a2.ctypes.data, st2.ctypes.data + 48, ncol, at, cap,
                                       ctypes.byref(k)), "append_rows")
    assert k.value == keep.sum() > 50
    np.testing.assert_array_equal(z2[at:], want[keep, :3])
    np.testing.assert_array_equal(t2[at:], want[keep, 3:6])
    np.testing.assert_array_equal(a2[at:], want[keep, 6:9])
    np.testing.assert_array_equal(st2[at:, 6], want[keep, 9])
    assert not z2[:at].any()
    assert lib.mmx_host_append_rows(want.ctypes.data, len(want), box_lo.ctypes.data_as(p32), box_hi.ctypes.data_as(p32),
                                    z2.ctypes.data, t2.ctypes.data, a2.ctypes.data, st2.ctypes.data + 48, ncol, at,
                                    cap - 1, ctypes.byref(k)) == 4            # MMX_ERR_WORKSPACE, the count still right
    assert k.value == keep.sum()
    # survivors: rows by id, abs columns replaced, key appended
    ids = np.ascontiguousarray(rng.permutation(n)[:30000 % n + 700], dtype=np.int64)
    keys = rng.integers(0, 99, len(ids)).astype(np.int64)
    abs_rows = rng.normal(size=(len(ids), 3))
    got = np.empty((len(ids), 12))
    cols3 = (ctypes.c_int32 * 3)(7, 8, 9)
    nat.check(lib.mmx_host_emit_survivors(store.ctypes.data, ncol, ids.ctypes.data, keys.ctypes.data, len(ids), 11,
                                          abs_rows.ctypes.data, cols3, got.ctypes.data), "emit_survivors")
    ref = store[ids, :11].copy()
    ref[:, 7:10] = abs_rows
    np.testing.assert_array_equal(got[:, :11], ref)
    np.testing.assert_array_equal(got[:, 11], keys)
    assert lib.mmx_host_emit_survivors(store.ctypes.data, ncol, ids.ctypes.data, keys.ctypes.data, len(ids), 15,
                                       abs_rows.ctypes.data, cols3, got.ctypes.data) == 1


def test_geometry_memo_and_ratio_frame():
    """``StackPruner._geometry`` remembers a block geometry by the identity of its arrays (and recomputes for other
    arrays, tolerances or shapes); ``_ratio_frame`` builds the data frame the dict-of-lists constructor would."""
    import pandas as pd
    from magellanmapper_amd import stack_detect as sd
    config.setup_roi_profiles(None)
    config.resolutions = np.array([[1.0, 1.0, 1.0]])
    config.roi_profile.update(segment_size=40, denoise_size=None)
    shape = (96, 150, 170)
    b1, b2 = sd.setup_blocks(config.roi_profile, shape), sd.setup_blocks(config.roi_profile, shape)
    args = lambda b, tol=None: (shape, b.overlap, b.tol if tol is None else tol, b.overlap_padding, b.sub_roi_slices,
                                b.sub_rois_offsets)
    p1, r1 = sd.StackPruner._geometry(*args(b1))
    assert r1 and sd.StackPruner._geometry(*args(b1))[0] is p1                 # remembered
    p2, _ = sd.StackPruner._geometry(*args(b2))
    assert p2 is not p1 and p2["n_keys"] == p1["n_keys"]                      # other arrays: computed again, same answer
    p3, _ = sd.StackPruner._geometry(*args(b1, np.asarray(b1.tol) - 1))
    assert p3 is not p1 and not np.array_equal(p3["tol"], p1["tol"])
    fresh = sd.StackPruner._axis_plan(*args(b1))
    for a, b in zip(p1["axes"], fresh["axes"]):
        assert (a is None) == (b is None)
        if a is not None:
            np.testing.assert_array_equal(a["bounds"], b["bounds"])
    ratios = {"blobs": [12, 7, 30], "ratio_pruning": [0.5, 1.0, 0.25], "ratio_adjacent": [1.5, 0.75, 2.0]}
    pd.testing.assert_frame_equal(sd.StackPruner._ratio_frame(ratios), pd.DataFrame(ratios))
    assert sd.StackPruner._ratio_frame({}).shape == pd.DataFrame({}).shape


def test_multi_channel_tables_and_coloc_flags_native():
    """``mmx_host_emit_tables_multi`` + ``mmx_host_coloc_flags`` (the co-localisation path's tables straight into the
    arena): block tables of two channels -- channel 0's rows, then channel 1's, shifted, tagged, border rows dropped --
    equal what ``detect_blobs`` / ``detect_sub_roi`` / ``merge_blobs`` build step by step, ``None`` vs EMPTY blocks are
    told apart, and the flags written into the extra columns equal ``colocalizer._flags_from_means`` block by block
    (NaN means poison a channel's threshold; blobs outside the ROI keep zeros)."""
    from magellanmapper_amd import _native as nat, colocalizer, stack_detect as sd
    from magellanmapper_amd.host_resolve import PeakBatch
    rng = np.random.default_rng(77)
    nb, shapes = 5, [(30, 40, 50)] * 5
    sig = [np.array([3.0, 4.0, 5.0]), np.array([2.0, 6.0])]

    def peaks(c, counts):
        offsets = np.concatenate(([0], np.cumsum(counts))).astype(np.int32)
        n = int(offsets[-1])
        coords = np.column_stack([rng.integers(0, 30, n), rng.integers(0, 40, n), rng.integers(0, 50, n),
                                  rng.integers(0, len(sig[c]), n)]).astype(np.int32)
        pb = PeakBatch(coords, rng.random(n), offsets)
        pb.alive = (rng.random(n) < 0.8).astype(np.uint8)
        pb.sigmas = sig[c]
        return pb
    pbs = [peaks(0, [12, 0, 7, 3, 9]), peaks(1, [5, 0, 0, 4, 11])]
    pbs[0].alive[pbs[0].offsets[3]:pbs[0].offsets[4]] = 0            # block 3: channel 0's peaks all pruned away
    grid_coords = np.array([[0, 0, k] for k in range(nb)], dtype=np.int32)
    offsets3 = np.array([[0.0, 0.0, 45.0 * k] for k in range(nb)])
    exclude = np.array([[2, 3, 4], [1, 2, 3]])
    arena = sd._TableArena(13, nb)
    sink = sd._ArenaSink(arena, grid_coords, offsets3, shapes, lambda i: exclude)
    seen = {}

    def flags_fn(idx, rows5, row_offsets, flags_ptr, ld):
        seen["rows5"], seen["off"] = rows5.copy(), row_offsets.copy()
        n = len(rows5)
        means = rng.random((2, n)) * 3
        means[0, rng.integers(0, n, 3)] = np.nan
        seen["means"] = means
        chans = np.array([0, 1], dtype=np.int32)
        shp = np.ascontiguousarray(shapes, dtype=np.int32)
        nat.check(nat.lib().mmx_host_coloc_flags(means.ctypes.data, chans.ctypes.data, 2, n, rows5.ctypes.data,
                                                 row_offsets.ctypes.data, nb, shp.ctypes.data, 2, flags_ptr, ld), "flags")
    out = sink.emit(list(range(nb)), pbs, [0, 1], flags_fn)
    at = 0
    for b in range(nb):
        parts = []
        for c in (0, 1):
            lo, hi = pbs[c].offsets[b], pbs[c].offsets[b + 1]
            rows = pbs[c].coords[lo:hi][pbs[c].alive[lo:hi].view(bool)]
            t = np.zeros((len(rows), 11))
            t[:, :3] = rows[:, :3]
            t[:, 3] = sig[c][rows[:, 3]] * np.sqrt(3)
            t[:, 4:6], t[:, 6], t[:, 7:10], t[:, 10] = -1, c, rows[:, :3], -1
            parts.append(t)
        tbl = np.vstack(parts)
        if len(tbl) == 0:
            assert out[b] is None
            continue
        tbl = detector.get_blobs_interior(tbl, shapes[b], *exclude)
        if len(tbl) == 0:
            assert out[b] is not None and out[b].shape == (0, 13)
            continue
        rel = tbl.copy()
        tbl[:, :3] += offsets3[b]
        tbl[:, 7:10] += offsets3[b]
        got = out[b]
        np.testing.assert_array_equal(got[:, :11], tbl)
        n = len(tbl)
        np.testing.assert_array_equal(arena.store[at:at + n, 13:], np.tile(grid_coords[b], (n, 1)))
        np.testing.assert_array_equal(arena.zyx[at:at + n], tbl[:, :3].astype(np.int32))
        np.testing.assert_array_equal(arena.abs[at:at + n], tbl[:, 7:10])
        a, e = seen["off"][b], seen["off"][b + 1]
        assert e - a == n
        np.testing.assert_array_equal(seen["rows5"][a:e], np.column_stack([np.full(n, b), rel[:, :3], rel[:, 6]]).astype(np.int32))
        want = colocalizer._flags_from_means(rel, seen["means"][:, a:e].T, shapes[b], 2)
        np.testing.assert_array_equal(got[:, 11:], want)
        at += n
    assert arena.n == at and arena.row_end[-1] == at and (arena.chan_lo, arena.chan_hi) == (0, 1)
    # a blob whose channel is no image channel: the reference raises IndexError, the native rule reports a bad argument
    rows5 = np.array([[0, 1, 1, 1, 5]], dtype=np.int32)
    off = np.array([0, 1], dtype=np.int64)
    m = np.zeros((1, 1))
    fl = np.zeros((1, 2))
    assert nat.lib().mmx_host_coloc_flags(m.ctypes.data, np.zeros(1, np.int32).ctypes.data, 1, 1, rows5.ctypes.data,
                                          off.ctypes.data, 1, np.array([[9, 9, 9]], np.int32).ctypes.data, 2,
                                          fl.ctypes.data, 2) == 1


def test_one_native_call_for_a_small_stack_equals_the_call_by_call_chain():
    """``mmx_host_finish_stack`` through ``stack_detect._StackFinisher``: a one-batch stack from its re-scored candidate
    table to the final table in one native call gives what the five calls give one after the other
    (``_resolve_peaks_native`` -> ``_prune_batch_native`` -> ``_ArenaSink`` -> ``prune_blobs_mp(final_form=True)``) --
    per-block tables, the final table, the pruning ratios, the counters -- hands its table out only when
    ``prune_blobs_mp`` is asked the planned way, and defers (changing nothing) when a block holds two equal peak values
    or the float32 values strayed beyond the band."""
    from magellanmapper_amd import _native as nat, blob_log as bl, stack_detect as sd
    config.setup_roi_profiles(None)
    config.resolutions = np.array([[1.0, 1.0, 1.0]])
    config.roi_profile.update(segment_size=20, denoise_size=None)
    shape = (38, 40, 36)
    blocks_g = sd.setup_blocks(config.roi_profile, shape)
    grid = blocks_g.sub_roi_slices.shape
    coords = list(np.ndindex(*grid))
    shapes = [tuple(len(range(*s.indices(n))) for s, n in zip(blocks_g.sub_roi_slices[c], shape)) for c in coords]
    nb, ns = len(coords), 3
    assert nb == 8
    rng = np.random.default_rng(5)
    space = bl.ScaleSpace.make(3.0, 5.0, ns)
    blocks = np.zeros(nb, dtype=nat.BLOCK_DTYPE)
    for k, shp in enumerate(shapes):
        blocks[k] = (0, shp[0], shp[1], shp[2], k, 32, 0)
    offsets3 = np.array([blocks_g.sub_rois_offsets[c] for c in coords], dtype=np.float64)

    class Img:
        pass
    Img.shape = shape
    args = (blocks_g.overlap, blocks_g.tol, blocks_g.sub_roi_slices, blocks_g.sub_rois_offsets, [0],
            blocks_g.overlap_padding)
    plan = sd.StackPruner._geometry(shape, blocks_g.overlap, blocks_g.tol, blocks_g.overlap_padding,
                                    blocks_g.sub_roi_slices, blocks_g.sub_rois_offsets)[0]

    def sink_for(arena):
        return sd._ArenaSink(arena, np.asarray(coords, dtype=np.int32), offsets3, shapes, None)

    def chain(table, n_c):
        stats = bl.BatchStats()
        pb = bl._prune_batch_native(bl._resolve_peaks_native(table, n_c, blocks, ns, 0.1, stats, 1e-3), space, 0.5, stats)
        arena = sd._TableArena(11, nb)
        tables = sink_for(arena)(list(range(nb)), pb, 0)
        seg = np.zeros(grid, dtype=object).view(sd._SegRois)
        for c, t in zip(coords, tables):
            seg[c] = t
        seg.arena = arena
        final, df = sd.StackPruner.prune_blobs_mp(Img, seg, *args, final_form=True, untouched=True)
        return tables, final, df, stats

    def one_call(table, n_c):
        stats = bl.BatchStats()
        arena = sd._TableArena(11, nb)
        fin = sd._StackFinisher(sink_for(arena), plan, [0])
        detector.Blobs(np.ones((1, 4))).format_blobs()
        tables = fin.run(list(range(nb)), table, n_c, blocks, space, 0.1, 1e-3, 0.5, stats, 0)
        return fin, arena, tables, stats

    done = 0
    for trial in range(10):
        table, n_c = _random_candidates(rng, shapes, ns, 60 if trial == 0 else 9)      # (dense blocks chain: deferred)
        want_tables, want_final, want_df, want_stats = chain(table, n_c)
        fin, arena, tables, stats = one_call(table, n_c)
        if tables is None:
            assert fin.deferred == 3 and arena.n == 0 and want_stats.n_order_fallbacks > 0       # a chain block: left alone
            continue
        done += 1
        for got, want in zip(tables, want_tables):
            assert (got is None) == (want is None)
            if want is not None:
                np.testing.assert_array_equal(got, want)
        seg = np.zeros(grid, dtype=object).view(sd._SegRois)
        for c, t in zip(coords, tables):
            seg[c] = t
        seg.arena, seg.pruner = arena, fin
        got_final, got_df = sd.StackPruner.prune_blobs_mp(Img, seg, *args, final_form=True, untouched=True)
        assert isinstance(got_final, sd._FinalTable) and got_final.col_names == want_final.col_names
        assert got_final.base is not None and np.shares_memory(got_final, fin.result[0])       # (its table, not a recomputation)
        np.testing.assert_array_equal(np.asarray(got_final), np.asarray(want_final))
        np.testing.assert_array_equal(got_df.to_numpy(), want_df.to_numpy())
        for f in ("n_peaks", "n_contested", "n_probes", "n_overlap_pairs", "n_blobs"):
            assert getattr(stats, f) == getattr(want_stats, f), f
        assert stats.max_f32_error == want_stats.max_f32_error
        # asked another way (not the final columns; another tolerance): the arena it filled is pruned as always
        seg.pruner = fin
        plain, _ = sd.StackPruner.prune_blobs_mp(Img, seg, *args)
        assert not isinstance(plain, sd._FinalTable) and plain.shape[1] == 11 and len(plain) == len(want_final)
        np.testing.assert_array_equal(plain[:, 7:10], np.asarray(want_final)[:, :3])
    assert 3 <= done < 10
    # equal peak values inside a block, a band that proved too narrow: deferred, nothing written
    table, n_c = _random_candidates(rng, shapes, ns, 40, tie_every=5)
    fin, arena, tables, _ = one_call(table, n_c)
    assert tables is None and fin.deferred == 1 and arena.n == 0 and fin.result is None
    table, n_c = _random_candidates(rng, shapes, ns, 40)
    table["v"][0] += np.float32(0.01)
    fin, arena, tables, _ = one_call(table, n_c)
    assert tables is None and fin.deferred == 2 and arena.n == 0
    # no candidates at all
    fin, arena, tables, _ = one_call(table[:0], 0)
    assert tables == [None] * nb and len(fin.result[0]) == 0


def test_stable_peak_order_switch_on_constructed_ties(monkeypatch):
    """``host_resolve.PEAK_ORDER = "0.19+"`` (PARITY UNPINNED: scikit-image >= 0.19 sorts peaks with a stable argsort; no
    fixture of the reference's pinned 0.25.2 can be made in this image): in a block that holds bit-equal float64
    responses the equal peaks keep np.nonzero order, everything else -- membership, values, blocks without ties -- is the
    default's; the default itself is untouched (NumPy's own order of equal keys), and negative thresholds are refused in
    the new mode (its 'nearest' peak mask equals this path's zero padding for thresholds >= 0 only)."""
    from magellanmapper_amd import _native as nat, blob_log as bl, host_resolve as hr
    rng = np.random.default_rng(23)
    shapes = [(6, 7, 8), (5, 5, 5), (4, 9, 6)]
    ns = 3
    table, n_c = _random_candidates(rng, shapes, ns, 60, tie_every=4)      # every fourth candidate: exactly 0.3125
    blocks = np.zeros(len(shapes), dtype=nat.BLOCK_DTYPE)
    for k, shp in enumerate(shapes):
        blocks[k] = (0, shp[0], shp[1], shp[2], k, 32, 0)
    assert hr.PEAK_ORDER == "0.18"
    default = bl._resolve_peaks_native(table, n_c, blocks, ns, 0.1, bl.BatchStats(), 1e-3)
    want_default = _resolve_with_numpy(table, n_c, shapes, ns, 0.1)
    monkeypatch.setattr(hr, "PEAK_ORDER", "0.19+")
    stable = bl._resolve_peaks_native(table, n_c, blocks, ns, 0.1, bl.BatchStats(), 1e-3)
    differs = 0
    for b, shp in enumerate(shapes):
        c_d, v_d = default.block(b)
        c_s, v_s = stable.block(b)
        np.testing.assert_array_equal(c_d, want_default[b][0])                 # (the default: np.argsort's own order)
        np.testing.assert_array_equal(v_s, v_d)                                # same values in the same (descending) order
        assert sorted(map(tuple, c_s)) == sorted(map(tuple, c_d))              # same peaks
        # stable: descending value, equal values by their place in the (z, y, x, sigma) cube
        lin = ((c_s[:, 0] * shp[1] + c_s[:, 1]) * shp[2] + c_s[:, 2]) * ns + c_s[:, 3]
        order = np.lexsort((lin, -v_s))
        np.testing.assert_array_equal(order, np.arange(len(v_s)))
        differs += int(not np.array_equal(c_s, c_d))
        assert np.count_nonzero(v_s == 0.3125) >= 2
    assert differs >= 1                 # (NumPy's default order of a dozen equal keys is not the stable one)
    with pytest.raises(NotImplementedError, match="nearest"):
        bl.Lane(0, 3.0, 3.0, 1, -0.5, 0.5)
    monkeypatch.setattr(hr, "PEAK_ORDER", "0.18")
    bl.Lane(0, 3.0, 3.0, 1, -0.5, 0.5)          # (the default takes any threshold)


def test_z_chunks_are_whole_block_layers_within_the_byte_limit():
    """``stack_detect._z_chunks``: the blocks of a share (z-major) cut into runs of whole layers whose planes fit half the
    limit; a layer that does not fit stands alone; every block is in exactly one chunk and the chunk's planes cover it."""
    from magellanmapper_amd import config, roi_prof, stack_detect as sd
    config.resolutions = np.array([[1.0, 1.0, 1.0]])
    shape = (230, 90, 100)
    bl_ = sd.setup_blocks(roi_prof.ROIProfile(segment_size=50, denoise_size=None), shape)
    coords = sd.StackDetector._grid_coords(bl_.sub_roi_slices.shape)
    for mine in (list(range(len(coords))), list(range(5, len(coords) - 3))):
        origins, shapes = sd.StackDetector._block_extents(bl_.sub_roi_slices, shape, mine)
        plane = 90 * 100 * 2
        for planes in (10, 70, 130, 10 ** 6):
            chunks = sd._z_chunks(coords, mine, origins, shapes, plane, 2 * planes * plane)
            assert chunks[0][0] == 0 and chunks[-1][1] == len(mine)
            assert all(a[1] == b[0] for a, b in zip(chunks, chunks[1:]))
            for k_lo, k_hi, z_lo, z_hi in chunks:
                layers = {coords[mine[k]][0] for k in range(k_lo, k_hi)}
                assert layers == set(range(min(layers), max(layers) + 1))
                # whole layers: no layer is shared with a neighbouring chunk
                assert not any(coords[mine[k]][0] in layers for k in list(range(0, k_lo)) + list(range(k_hi, len(mine))))
                for k in range(k_lo, k_hi):
                    assert z_lo <= origins[k][0] and origins[k][0] + shapes[k][0] <= z_hi
                assert len(layers) == 1 or (z_hi - z_lo) <= planes
            if planes == 10 ** 6:
                assert len(chunks) == 1
            if planes == 10:
                assert len(chunks) == len({coords[i][0] for i in mine})
