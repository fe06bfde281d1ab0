"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the golden
vectors captured from the real reference.  Needs a real MI355X (``-m gpu``).

Bar: LoG response within 1e-4 (float32 path; north_star), everything that decides integer
blob coordinates bit exact: float64 re-scored values, peak sets and order, pruned blobs,
magmap tables.
"""
import ast
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, lexsorted, load_golden

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

LOG_TOL = 1e-4


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: torch.cuda.is_available() is False")
    from magellanmapper_amd import _native
    assert os.path.exists(_native.LIB_PATH), "libmmx_hip.so must be built in-tree"
    assert _native.lib().mmx_device_count() >= 1, "no gfx950 device visible to libmmx_hip.so"
    return torch.device("cuda", 0)


@pytest.fixture(params=["native", "numpy"])
def host_path(request, monkeypatch):
    """Both host formulations of the per-batch decisions: the native one (``mmx_host_resolve_peaks`` & co., the
    default) and the NumPy one it replaced (kept as a cross-check and for ``exact_values=False``)."""
    from magellanmapper_amd import blob_log as bl
    monkeypatch.setattr(bl, "HOST_PATH", request.param)
    return request.param


BLOBLOG_CASES = sorted(os.path.basename(p)[len("bloblog_"):-4]
                       for p in glob.glob(os.path.join(GOLDEN, "bloblog_*.npz")))
DETECT_CASES = sorted(os.path.basename(p)[len("detect_"):-4]
                      for p in glob.glob(os.path.join(GOLDEN, "detect_*.npz")))
STACK_CASES = sorted(os.path.basename(p)[len("stack_"):-4]
                     for p in glob.glob(os.path.join(GOLDEN, "stack_*.npz")))


def _oracle_stages(g):
    from oracle import blob_log_oracle as blo
    return blo.blob_log(g["volume"], float(g["min_sigma"]), float(g["max_sigma"]),
                        int(g["num_sigma"]), float(g["threshold"]), float(g["overlap"]),
                        return_stages=True)


@pytest.mark.parametrize("case", BLOBLOG_CASES)
def test_log_cube_within_tolerance(gpu, case):
    """A0-A3: float32 device LoG vs the float64 oracle cube, fast and generic kernels."""
    from magellanmapper_amd import blob_log as bl
    g = load_golden("bloblog_%s.npz" % case)
    _, st = _oracle_stages(g)
    dvol = bl.DeviceVolume(g["volume"])
    space = bl.ScaleSpace.make(float(g["min_sigma"]), float(g["max_sigma"]), int(g["num_sigma"]))
    np.testing.assert_array_equal(space.sigmas, st["sigmas"][:, 0])
    shape = g["volume"].shape
    fast = bl.log_cube_blocks(dvol, 0, [(0, 0, 0)], [shape], space)[0]
    slow = bl.log_cube_blocks(dvol, 0, [(0, 0, 0)], [shape], space, generic=True)[0]
    ref = st["cube"].astype(np.float64)
    assert np.max(np.abs(fast - ref)) < LOG_TOL
    assert np.max(np.abs(slow - ref)) < LOG_TOL
    # and the golden crop from the real scikit-image run
    o = g["cube_crop_origin"]
    sl = tuple(slice(a, a + n) for a, n in zip(o, g["cube_crop"].shape[:3]))
    assert np.max(np.abs(fast[sl] - g["cube_crop"])) < LOG_TOL


@pytest.mark.parametrize("case", ["u16_5sigma", "u8_3sigma", "f32_2sigma", "u16_twoscale10"])
def test_fused_zx_path_gives_the_same_cube(gpu, case):
    """The fused Z+X kernels (zx_mode 2: wave-specialised packed math, 6: split-float16 matrix-core X+Z on operand-ordered
    voxels with tiled P / Q and its own Y pass, 7: the same with 16-bit tiles) must reproduce the three-pass result
    (mode 0)."""
    from magellanmapper_amd import _native as nat
    from magellanmapper_amd import blob_log as bl
    g = load_golden("bloblog_%s.npz" % case)
    _, st = _oracle_stages(g)
    dvol = bl.DeviceVolume(g["volume"])
    space = bl.ScaleSpace.make(float(g["min_sigma"]), float(g["max_sigma"]), int(g["num_sigma"]))
    shape = g["volume"].shape
    default = bl.ZX_MODE
    try:
        bl.ZX_MODE = nat.MMX_ZX_SEPARATE
        nat.timing_enable(True)
        sep = bl.log_cube_blocks(dvol, 0, [(0, 0, 0)], [shape], space)[0]
        kinds = nat.timing_read()
        assert kinds["zxpass"][1] == 0 and (kinds["zpass"][1] > 0 or kinds["generic"][1] > 0)
        assert bl.LAST_ZX_PATH == nat.MMX_ZX_SEPARATE
        for mode in (2, 6, 7):
            bl.ZX_MODE = mode
            fused = bl.log_cube_blocks(dvol, 0, [(0, 0, 0)], [shape], space)[0]
            kinds = nat.timing_read()
            # the fused kernels really ran (wherever the geometry lets any register-resident pass run)
            assert kinds["zxpass"][1] > 0 or kinds["generic"][1] > 0 or kinds["zpass"][1] > 0
            assert np.max(np.abs(fused - st["cube"].astype(np.float64))) < LOG_TOL, (mode, bl.LAST_ZX_PATH)
            # (mode 7 hands 16-bit intermediates to the Y pass: error <= 3.0e-5 of the value scale)
            assert np.max(np.abs(fused - sep)) < (4.5e-5 if bl.LAST_ZX_PATH == 7 else 2e-6) * max(1.0, float(np.abs(sep).max()))
    finally:
        bl.ZX_MODE = default
        nat.timing_enable(False)


@pytest.mark.parametrize("case", BLOBLOG_CASES)
def test_blob_log_identical_to_reference(gpu, case, host_path):
    """A4 + A5: ordered raw peaks with bit-exact float64 values, and the pruned blobs."""
    from magellanmapper_amd import blob_log as bl
    g = load_golden("bloblog_%s.npz" % case)
    res_o, st = _oracle_stages(g)
    dvol = bl.DeviceVolume(g["volume"])
    stats = bl.BatchStats()
    res, peaks = bl.blob_log_blocks(
        dvol, 0, [(0, 0, 0)], [g["volume"].shape], float(g["min_sigma"]), float(g["max_sigma"]),
        int(g["num_sigma"]), float(g["threshold"]), float(g["overlap"]), stats=stats,
        return_peaks=True)
    coords, vals = peaks[0]
    np.testing.assert_array_equal(coords, st["peaks"].reshape(-1, 4))
    np.testing.assert_array_equal(vals, st["peak_values"].astype(np.float64))   # bit exact
    np.testing.assert_array_equal(coords, g["peaks"].reshape(-1, 4))            # real skimage
    assert res[0].shape == g["pruned"].shape
    np.testing.assert_array_equal(res[0], g["pruned"])
    np.testing.assert_array_equal(res[0], res_o)
    if stats.n_candidates:
        # a quarter of the nomination band: 5e-6 for the float32 paths, 5e-5 where the default path hands 16-bit
        # intermediates to the Y pass (raw integer volumes, radii <= 24; their bound is 3.0e-5)
        q16 = bl.LAST_ZX_PATH == 7
        assert stats.max_f32_error < (4.4e-5 if q16 else 5e-6) * max(1.0, float(np.abs(g["volume"]).max())
                                                                     if g["volume"].dtype.kind == "f" else 1.0)


def test_rescore_bit_exact_at_arbitrary_points(gpu):
    """The float64 re-score equals SciPy (via the oracle) at random voxels, incl. borders."""
    import ctypes
    from magellanmapper_amd import _native as nat
    from magellanmapper_amd import blob_log as bl
    for case in ("u16_5sigma", "u8_3sigma", "f32_2sigma", "f64_2sigma", "u16_thin"):
        g = load_golden("bloblog_%s.npz" % case)
        _, st = _oracle_stages(g)
        cube = st["cube"]
        dvol = bl.DeviceVolume(g["volume"])
        space = bl.ScaleSpace.make(float(g["min_sigma"]), float(g["max_sigma"]), int(g["num_sigma"]))
        shape = g["volume"].shape
        rng = np.random.default_rng(3)
        n = 300
        pts = np.zeros(n, dtype=nat.CAND_DTYPE)
        pts["s"] = rng.integers(0, len(space.sigmas), n)
        for ax, name in enumerate("zyx"):
            pts[name] = rng.integers(0, shape[ax], n)
        pts["z"][:20] = 0
        pts["y"][10:30] = shape[1] - 1
        pts["x"][20:40] = 0
        blocks, _slot = bl._make_blocks(dvol, 0, [(0, 0, 0)], [shape])
        d_blocks = bl._to_device_bytes(blocks, gpu)
        d_pts = bl._to_device_bytes(pts, gpu)
        d_w0 = torch.from_numpy(space.w0_tab).to(gpu)
        d_w2 = torch.from_numpy(space.w2_tab).to(gpu)
        vol = dvol.view(0, False)
        nat.check(nat.lib().mmx_rescore_f64(
            ctypes.byref(vol), d_blocks.data_ptr(), 1, d_pts.data_ptr(), n, None, d_w0.data_ptr(),
            d_w2.data_ptr(), nat.as_int32_ptr(space.radii), nat.as_double_ptr(space.norms),
            len(space.sigmas), 1 if g["volume"].dtype == np.float32 else 0,
            torch.cuda.current_stream().cuda_stream), "rescore")
        got = d_pts.cpu().numpy().view(nat.CAND_DTYPE)["v64"]
        want = cube[pts["z"], pts["y"], pts["x"], pts["s"]].astype(np.float64)
        np.testing.assert_array_equal(got, want, err_msg=case)


def test_multi_block_batch_equals_per_block(gpu):
    """Blocks of one batch are independent images with reflect boundaries at their faces."""
    from magellanmapper_amd import blob_log as bl
    from oracle import blob_log_oracle as blo
    g = load_golden("stack_u16_2x3x3.npz")
    vol = g["roi"]
    dvol = bl.DeviceVolume(vol)
    origins = [(0, 0, 0), (20, 30, 40), (40, 50, 11), (3, 60, 0), (30, 0, 50)]
    shapes = [(45, 45, 45), (44, 40, 45), (24, 46, 45), (40, 36, 33), (5, 45, 46)]
    res = bl.blob_log_blocks(dvol, 0, origins, shapes, 3, 5, 5, 0.1, 0.5)
    for o, s, r in zip(origins, shapes, res):
        sub = vol[o[0]:o[0] + s[0], o[1]:o[1] + s[1], o[2]:o[2] + s[2]]
        want = blo.blob_log(sub, 3, 5, 5, 0.1, 0.5)
        assert r.shape == want.shape, (o, s)
        np.testing.assert_array_equal(r, want)
    # small workspace budget -> several batches, same answer
    res2 = bl.blob_log_blocks(dvol, 0, origins, shapes, 3, 5, 5, 0.1, 0.5, budget_bytes=8 << 20)
    for a, b in zip(res, res2):
        np.testing.assert_array_equal(a, b)


def test_the_tail_of_a_batch_on_its_own_stream_changes_nothing(gpu, monkeypatch):
    """Raw volumes: NMS, probe expansion and exact re-score of a batch run on a second stream beside the next batch's
    LoG kernels, the batches alternating between two workspaces (``blob_log.RESCORE_STREAM``).  Several batches of
    different sizes, with and without it: identical peaks, values included."""
    from magellanmapper_amd import blob_log as bl, synth
    vol = synth.make_volume(23, (70, 150, 160), 140)
    dvol = bl.DeviceVolume(vol)
    origins = [(z, y, x) for z in (0, 30) for y in (0, 50, 100) for x in (0, 55, 110)]
    shapes = [(40, 50, 50)] * len(origins)
    runs = []
    for on in (True, False, True):
        monkeypatch.setattr(bl, "RESCORE_STREAM", on)
        st = bl.BatchStats()
        res, peaks = bl.blob_log_blocks(dvol, 0, origins, shapes, 2, 4, 3, 0.08, 0.5, budget_bytes=24 << 20,
                                        stats=st, return_peaks=True)
        runs.append((res, peaks, st))
    assert runs[0][2].n_blobs > 100
    for res, peaks, st in runs[1:]:
        assert st.n_candidates == runs[0][2].n_candidates and st.n_blobs == runs[0][2].n_blobs
        for a, b in zip(res, runs[0][0]):
            np.testing.assert_array_equal(a, b)
        for (ca, va), (cb, vb) in zip(peaks, runs[0][1]):
            np.testing.assert_array_equal(ca, cb)
            np.testing.assert_array_equal(va, vb)


def test_large_sigma_takes_generic_path(gpu):
    """sigma 7.5 -> radius 30 > MMX_MAX_RADIUS_FAST: generic kernels, same exactness."""
    from magellanmapper_amd import blob_log as bl
    from magellanmapper_amd import synth
    from oracle import blob_log_oracle as blo
    vol = synth.make_volume(41, (40, 48, 52), 6, blob_sigma=7.0)
    want = blo.blob_log(vol, 7.0, 7.5, 2, 0.05, 0.5)
    got = bl.blob_log(vol, 7.0, 7.5, 2, 0.05, 0.5)
    assert len(want) > 0
    np.testing.assert_array_equal(got, want)


def test_contested_ties_resolved_exactly(gpu, host_path):
    """A mirror-symmetric volume has exact float64 ties between mirrored voxels: plateaus of
    equal maxima must come out exactly as scikit-image reports them."""
    from magellanmapper_amd import blob_log as bl
    from oracle import blob_log_oracle as blo
    rng = np.random.default_rng(9)
    half = rng.integers(300, 900, (24, 30, 16)).astype(np.uint16)
    zz, yy, xx = np.meshgrid(np.arange(24.), np.arange(30.), np.arange(16.), indexing="ij")
    for c in ((6, 8, 15.5), (16, 20, 15.5), (12, 14, 6.0)):
        d2 = (zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2
        half = np.maximum(half, (30000 * np.exp(-d2 / 18.0)).astype(np.uint16))
    vol = np.concatenate((half, half[:, :, ::-1]), axis=2)     # mirror about x = 15.5
    res_o, st = blo.blob_log(vol, 3, 4, 3, 0.05, 0.5, return_stages=True)
    stats = bl.BatchStats()
    res, peaks = bl.blob_log_blocks(bl.DeviceVolume(vol), 0, [(0, 0, 0)], [vol.shape], 3, 4, 3, 0.05,
                                    0.5, stats=stats, return_peaks=True)
    assert stats.n_contested > 0
    got = peaks[0][0]
    want = st["peaks"].reshape(-1, 4)
    # tied peaks may be ordered differently by an unstable sort only among exactly equal values
    np.testing.assert_array_equal(lexsorted(got), lexsorted(want))
    np.testing.assert_array_equal(np.sort(peaks[0][1]), np.sort(st["peak_values"]))
    np.testing.assert_array_equal(lexsorted(res[0]), lexsorted(res_o))


def _profiles_from(g):
    return ast.literal_eval(str(g["profiles"]))


def _apply_profiles(profs):
    from magellanmapper_amd import config
    config.setup_roi_profiles(["default"] * len(profs))
    for p, over in zip(config.roi_profiles, profs):
        p["isotropic"] = None
        p.update(over)
        p["denoise_size"] = None


@pytest.mark.parametrize("case", DETECT_CASES)
def test_detect_blobs_matches_reference(gpu, case):
    """A6/A7: ``detect_blobs`` 11-column table identical to the real reference's."""
    from magellanmapper_amd import config, detector
    g = load_golden("detect_%s.npz" % case)
    _apply_profiles(_profiles_from(g))
    unmix = ast.literal_eval(str(g["unmix"])) if "unmix" in g else None
    for p in config.roi_profiles:          # U1: the ROIProfile attribute the reference reads (:911)
        p.spectral_unmixing = unmix
    config.resolutions = g["resolutions"]
    channel = None if g["channel"].ndim == 0 else list(g["channel"])
    excl = None if g["exclude_border"].ndim == 0 else g["exclude_border"]
    table = detector.detect_blobs(g["roi"], channel, excl)
    if bool(g["is_none"]):
        assert table is None
        return
    assert table.dtype == np.float64
    np.testing.assert_array_equal(table, g["table"])


@pytest.fixture
def golden_preproc_env(monkeypatch):
    """The environment the preprocessing fixtures were made in: scikit-image 0.18.3 (RGB guess in
    filters.gaussian) and NumPy 1.26 (its np.exp gives the sigma-8 weights stored in preproc.npz)."""
    from magellanmapper_amd import config, preprocess
    w = load_golden("preproc.npz")["gauss8_weights"]
    monkeypatch.setattr(preprocess, "RGB_GUESS", True)
    monkeypatch.setattr(preprocess, "GAUSS_WEIGHTS_OVERRIDE", np.ascontiguousarray(w[32:]))
    yield
    config.near_max = [-1.0]


@pytest.mark.parametrize("case", STACK_CASES)
def test_detect_blobs_blocks_matches_reference(gpu, case, tmp_path, monkeypatch, golden_preproc_env, host_path):
    """A8-A14 (and P1-P3 for the ``denoise*`` cases): per-block tables, merged table and final
    8-column table identical to the real reference's ``detect_blobs_blocks``."""
    from magellanmapper_amd import chunking, config, stack_detect
    monkeypatch.chdir(tmp_path)
    g = load_golden("stack_%s.npz" % case)
    over = ast.literal_eval(str(g["overrides"]))
    config.setup_roi_profiles(None)
    config.roi_profile["denoise_size"] = None
    config.roi_profile.update(over)
    config.near_max = list(g["near_max"]) if "near_max" in g else [-1.0]
    config.resolutions = g["resolutions"]
    config.filename = "golden"
    channels = None if g["channels"].ndim == 0 else list(g["channels"])
    roi = g["roi"]
    chls = channels if channels is not None else (list(range(roi.shape[3])) if roi.ndim > 3 else [0])
    coloc = bool(g["coloc"]) if "coloc" in g else False
    bl = stack_detect.setup_blocks(config.roi_profile, roi.shape)
    seg = stack_detect.StackDetector.detect_blobs_sub_rois(
        None, roi, bl.sub_roi_slices, bl.sub_rois_offsets, bl.denoise_max_shape, bl.exclude_border,
        coloc, chls)
    assert seg.shape == tuple(g["grid"])
    for c in np.ndindex(*seg.shape):
        want = g["block_%d_%d_%d" % c]
        if want.shape[0] == 0:
            assert seg[c] is None
        else:
            np.testing.assert_array_equal(seg[c], want)
    merged = chunking.merge_blobs(seg)
    img5d = stack_detect.Image5d(roi[None])
    _, _, blobs = stack_detect.detect_blobs_blocks("golden", img5d, None, None, channels, False,
                                                   False, True, coloc)
    if g["final"].shape[0] == 0:
        assert merged is None and blobs.blobs is None
        return
    np.testing.assert_array_equal(merged, g["merged"])
    np.testing.assert_array_equal(blobs.blobs, g["final"])
    assert list(blobs.cols) == list(g["final_cols"])
    if coloc:         # C1 + the reference's column-offset quirk
        assert blobs.colocalizations.dtype == np.uint8
        np.testing.assert_array_equal(blobs.colocalizations, g["colocs"])
    else:
        assert blobs.colocalizations is None


def test_remove_close_blobs_device(gpu):
    """A13: device all-pairs search + host apply, vs the reference's outputs (incl. >1000-row
    chunking, multi-matches, round-half-even)."""
    from magellanmapper_amd import detector
    g = load_golden("prune.npz")
    for k in range(int(g["n_rc"])):
        detector.Blobs(np.ones((1, 4))).format_blobs()
        pruned, master = detector.remove_close_blobs(
            g["rc%d_check" % k].copy(), g["rc%d_master" % k].copy(), g["rc%d_tol" % k])
        np.testing.assert_array_equal(pruned, g["rc%d_pruned" % k])
        np.testing.assert_array_equal(master, g["rc%d_master_out" % k])


def test_subimage_offset_and_archive(gpu, tmp_path, monkeypatch):
    """C1 plumbing: 2-channel stand-in of sample_region (1,51,200,200,2), sub-image offset
    (30,30,8) size (70,70,10) as the reference's integration test uses (x,y,z order there),
    profile 4xnuc over the defaults; the archive round-trips."""
    from magellanmapper_amd import config, detector, stack_detect, synth
    from oracle import magmap_oracle as mmo
    monkeypatch.chdir(tmp_path)
    vol = np.stack((synth.make_volume(51, (51, 200, 200), 120),
                    synth.make_volume(52, (51, 200, 200), 90)), axis=-1)
    config.setup_roi_profiles(["4xnuc"])
    config.roi_profile.update(denoise_size=None, num_sigma=3)
    config.resolutions = np.array([[1.0, 1.0, 1.0]])
    config.filename = str(tmp_path / "sample_region.tif")
    config.channel = None
    img5d = stack_detect.Image5d(vol[None])
    offset, size = (8, 30, 30), (10, 70, 70)     # z, y, x
    _, _, blobs = stack_detect.detect_blobs_blocks(config.filename, img5d, offset, size, None,
                                                   False, True, False, False)
    sub = vol[8:18, 30:100, 30:100]
    want, _ = mmo.detect_blobs_blocks(sub, None, [dict(config.roi_profile)], config.resolutions)
    assert want is not None and len(want) > 0
    np.testing.assert_array_equal(lexsorted(blobs.blobs), lexsorted(want))
    assert os.path.exists("stack_detection_times.csv")
    # config.save_subimg: the ROI next to the archive, named as the reference names it (x,y,z in the file name)
    assert not os.path.exists(tmp_path / "sample_region_(30,30,8)x(70,70,10)_subimg.npy")
    monkeypatch.setattr(config, "save_subimg", True)
    _, _, again = stack_detect.detect_blobs_blocks(config.filename, img5d, offset, size, None,
                                                   False, False, False, False)
    monkeypatch.setattr(config, "save_subimg", False)
    np.testing.assert_array_equal(again.blobs, blobs.blobs)
    saved = np.load(tmp_path / "sample_region_(30,30,8)x(70,70,10)_subimg.npy")
    assert saved.dtype == vol.dtype
    np.testing.assert_array_equal(saved, sub)
    stats, fdbk, all_blobs = stack_detect.detect_blobs_stack(config.filename, img5d, offset, size)
    assert os.path.exists(all_blobs.path)
    loaded = detector.Blobs().load_blobs(all_blobs.path)
    np.testing.assert_array_equal(loaded.blobs, all_blobs.blobs)
    assert list(loaded.cols) == ["z", "y", "x", "radius", "confirmed", "truth", "channel", "region"]
    assert int(loaded.ver) == 5


def _oracle_final(vol, prof_over):
    from magellanmapper_amd import config
    from oracle import magmap_oracle as mmo
    prof = dict(config.roi_profile)
    return mmo.detect_blobs_blocks(vol, None, [prof], config.resolutions)[0]


def test_many_blocks_several_batches_identical_to_oracle(gpu, tmp_path, monkeypatch, host_path):
    """32 blocks through the pipelined batches (forced small workspace -> several batches, tapered
    tail, side-stream follow-ups) and the native host prune; final table identical to the oracle."""
    import functools
    from magellanmapper_amd import blob_log as bl
    from magellanmapper_amd import config, stack_detect, synth
    monkeypatch.chdir(tmp_path)
    vol = synth.make_volume(61, (96, 160, 168), 330)
    config.setup_roi_profiles(None)
    config.roi_profile.update(denoise_size=None, num_sigma=4, segment_size=44)
    config.resolutions = np.array([[1.0, 1.0, 1.0]])
    config.filename = "many"
    monkeypatch.setattr(bl, "blob_log_blocks", functools.partial(bl.blob_log_blocks, budget_bytes=96 << 20))
    img5d = stack_detect.Image5d(vol[None])
    _, _, blobs = stack_detect.detect_blobs_blocks("many", img5d, None, None, None, False, False, True, False)
    st = stack_detect.StackDetector.last_stats
    assert st.n_blocks == 3 * 4 * 4
    want = _oracle_final(vol, None)
    assert len(want) > 200
    np.testing.assert_array_equal(lexsorted(blobs.blobs), lexsorted(want))


def test_candidate_table_overflow_is_retried(gpu, monkeypatch, host_path):
    """A candidate table that is too small must be detected and the batch redone."""
    from magellanmapper_amd import blob_log as bl
    from oracle import blob_log_oracle as blo
    g = load_golden("stack_u16_2x3x3.npz")
    vol = g["roi"]
    want = blo.blob_log(vol, 3, 5, 3, 0.1, 0.5)
    real = bl._enqueue_detect
    calls = []

    def tiny_first(*args, **kwargs):
        if not calls:                       # first attempt: room for 8 candidates only
            kwargs["cap"] = 8
        calls.append(kwargs.get("cap"))
        return real(*args, **kwargs)

    monkeypatch.setattr(bl, "_enqueue_detect", tiny_first)
    got = bl.blob_log(vol, 3, 5, 3, 0.1, 0.5)
    assert len(calls) == 2 and calls[0] == 8 and calls[1] > 8
    np.testing.assert_array_equal(got, want)


def test_constant_image_plateaus(gpu, host_path):
    """Constant images: with one sigma every voxel equals its 3^4 maximum and scikit-image reports
    no peaks at all (peak.py:41-43); with two sigmas the brighter scale is one big plateau of
    peaks (all exact float64 ties) that the overlap prune then thins out -- both must match."""
    from magellanmapper_amd import blob_log as bl
    from oracle import blob_log_oracle as blo
    vol = np.full((20, 24, 40), 0.5, dtype=np.float64)
    want = blo.blob_log(vol, 2, 2, 1, -1.0, 0.5)
    got = bl.blob_log(vol, 2, 2, 1, -1.0, 0.5)
    assert want.shape == (0, 3) and got.shape == (0, 3)
    small = np.full((9, 10, 12), 0.5, dtype=np.float64)
    want, st = blo.blob_log(small, 2, 3, 2, -1.0, 0.5, return_stages=True)
    stats = bl.BatchStats()
    got, peaks = bl.blob_log_blocks(bl.DeviceVolume(small), 0, [(0, 0, 0)], [small.shape], 2, 3, 2, -1.0,
                                    0.5, stats=stats, return_peaks=True)
    assert len(st["peaks"]) == small.size and stats.n_contested == small.size
    np.testing.assert_array_equal(lexsorted(peaks[0][0]), lexsorted(st["peaks"]))
    np.testing.assert_array_equal(lexsorted(got[0]), lexsorted(want))


def test_two_ranks_share_one_volume(gpu, tmp_path):
    """bench.py's N > 1 path (self-launch of the ranks, block sharding, per-rank z-slab generation, distributed
    pruning) with two ranks on this one GPU over gloo must find exactly the blobs of the single-rank run."""
    import json
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, MMX_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    shape = ["96", "300", "300"]
    common = ["--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--shape", *shape]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", *common],
                         capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert one.returncode == 0, one.stderr[-2000:]
    # (no launcher: `python bench.py --gpus 2` starts its two ranks itself, before it touches the GPU)
    env.pop("WORLD_SIZE", None)
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", *common],
                         capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert two.returncode == 0, two.stderr[-2000:]
    r1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    r2 = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    assert r1["n_gpus"] == 1 and r2["n_gpus"] == 2
    assert r1["blobs"] == r2["blobs"] and r1["blobs"] > 100
    assert r1["table_sha1"] == r2["table_sha1"]          # the same final table, row for row
    assert r2["config"]["blocks_per_rank"] == 2          # 1 x 2 x 2 blocks over two ranks


def test_detect_blobs_stack_from_the_on_disk_image(gpu, tmp_path, monkeypatch):
    """importer.read_file (memory-mapped image5d.npy + meta.yml written by the real reference) ->
    detect_blobs_stack -> archive; equals detection on the in-memory array, and the metadata
    (resolutions, near_max) reaches the detection and the preprocessing."""
    import shutil
    from magellanmapper_amd import blob_log as bl, config, detector, importer, stack_detect, volume
    from oracle import magmap_oracle as mmo
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(volume, "_STREAM_MIN_BYTES", 1 << 10)    # take the streamed-upload path (slabs beside the detection)
    monkeypatch.setattr(volume, "_STREAM_CHUNK_BYTES", 40 << 10)     # ... in several slabs
    for fn in ("sample_image5d.npy", "sample_meta.yml"):
        shutil.copy(os.path.join(GOLDEN, fn), tmp_path / fn)
    config.setup_roi_profiles(None)
    config.roi_profile.update(num_sigma=3, segment_size=30)      # denoise_size stays at the default 25
    config.resolutions, config.near_max, config.channel = None, [-1.0], None
    try:
        img5d = importer.read_file(str(tmp_path / "sample.czi"))
        config.filename = str(tmp_path / "sample.czi")
        assert isinstance(img5d.img, np.memmap)
        np.testing.assert_array_equal(config.resolutions, [[5.0, 1.2, 1.2]])
        _, _, blobs = stack_detect.detect_blobs_stack(str(tmp_path / "sample"), img5d)
        want, _ = mmo.detect_blobs_blocks(np.asarray(img5d.img[0]), None, [dict(config.roi_profile)],
                                          config.resolutions, near_max=config.near_max)
        np.testing.assert_array_equal(blobs.blobs, want)
        back = detector.Blobs().load_blobs(str(tmp_path / "sample_blobs.npz"))
        np.testing.assert_array_equal(back.blobs, want)
        np.testing.assert_array_equal(back.resolutions, config.resolutions)
    finally:
        config.resolutions, config.near_max = None, [-1.0]
        detector.Blobs(np.ones((1, 4))).format_blobs()


@pytest.mark.parametrize("fused", [0, 2, 6, 7, "7 + Y on the VALU"])
def test_every_kernel_radius_matches_oracle(gpu, fused, monkeypatch):
    """Each compiled radius (1..24 register-resident, 25 generic) of the separable passes against the
    float64 oracle cube.  Regression: the X pass read its register window in pairs but sized it odd for
    odd radii (undefined behaviour that showed as NaNs for R = 3..11).  Mode 7 runs its Y pass on the matrix cores
    (ym_kernel: three row tiles here, both mirrored ends inside one k-block) and, with MMX_ZX_Y_VALU, on the VALU."""
    from magellanmapper_amd import _native as nat, blob_log as bl, synth
    from oracle import blob_log_oracle as blo
    if fused == "7 + Y on the VALU":
        fused = 7
        monkeypatch.setattr(bl, "ZX_FLAGS", nat.MMX_ZX_Y_VALU)
    vol = synth.make_volume(3, (35, 42, 48), 12)
    dvol = bl.DeviceVolume(vol)
    img = blo.img_as_float(vol)
    default, bl.ZX_MODE = bl.ZX_MODE, int(fused)
    try:
        for R in range(1, 26):
            sigma = (R + 0.2) / 4.0
            space = bl.ScaleSpace.make(sigma, sigma, 1)
            assert space.radii[0] == R
            got = np.squeeze(bl.log_cube_blocks(dvol, 0, [(0, 0, 0)], [vol.shape], space)[0])
            want = blo.log_cube(img, np.array([[sigma] * 3]))[..., 0]
            tol = LOG_TOL * 1e-2
            if fused == 7:      # 16-bit intermediates: the bound the library states for these weights
                tol = nat.lib().mmx_tiled_q16_error_bound(nat.as_double_ptr(space.w0[0]), nat.as_double_ptr(space.w2[0]),
                                                          R, float(space.norms[0]))
                assert 2.5e-5 < tol < 5.5e-5
            assert np.abs(got - want).max() < tol, R
            # the kernel asked for is the kernel that ran (its geometry conditions hold for this volume)
            if fused in (2, 3) and 1 <= R <= 24:
                assert bl.LAST_ZX_PATH == fused, (R, bl.LAST_ZX_PATH)
            if fused in (4, 5, 6, 7) and 1 <= R <= 24:
                assert bl.LAST_ZX_PATH == fused, (R, bl.LAST_ZX_PATH)
    finally:
        bl.ZX_MODE = default


@pytest.mark.parametrize("case", ["preprocessed_1.94", "unit_range_extremes", "float_range_40", "float64_range_3"])
def test_q16_error_bound_holds_across_value_ranges(gpu, case, monkeypatch):
    """The analytic bound of the 16-bit intermediates (``mmx_tiled_q16_error_bound`` x the stated value range) on FLOAT
    volumes, every radius 1..24: the range a stock preprocessing leaves (unsharp overshoot: [0, 1.94]), the full unit
    range with voxels AT both extremes (saturated plateaus next to zeros: what a clip leaves), a float image of range
    40 and a float64 one -- the cases on which the nomination band (4 x the bound) of preprocessed and float volumes
    rests.  The LoG contract of 1e-4 is relative to the value scale, as the band is."""
    from magellanmapper_amd import _native as nat, blob_log as bl, synth
    from oracle import blob_log_oracle as blo
    rng = np.random.default_rng(7)
    base = synth.make_volume(11, (35, 42, 48), 14).astype(np.float64) / 65535.0
    if case == "preprocessed_1.94":
        vmax, vol = 1.94, (base / base.max() * 1.94).astype(np.float32)
    elif case == "unit_range_extremes":
        vol = base / base.max()
        vol[rng.random(vol.shape) < 0.05] = 1.0            # saturated voxels scattered through the noise
        vol[10:20, 5:30, 8:40] = 1.0                       # and a saturated plateau
        vol[22:30, :, :20] = 0.0                           # next to clipped-to-zero regions
        vmax, vol = 1.0, vol.astype(np.float32)
    elif case == "float_range_40":
        vmax, vol = 40.0, (base / base.max() * 40.0).astype(np.float32)
    else:
        vmax, vol = 3.0, base / base.max() * 3.0           # float64: the float32 copy feeds the passes
    dvol = bl.DeviceVolume(vol)
    img = vol.astype(np.float64)
    monkeypatch.setattr(bl, "ZX_MODE", nat.MMX_ZX_TILED_Q16)
    worst = 0.0
    for R in range(1, 25):
        sigma = (R + 0.2) / 4.0
        space = bl.ScaleSpace.make(sigma, sigma, 1)
        got = np.squeeze(bl.log_cube_blocks(dvol, 0, [(0, 0, 0)], [vol.shape], space, value_range=vmax)[0])
        assert bl.LAST_ZX_PATH == nat.MMX_ZX_TILED_Q16, (case, R, bl.LAST_ZX_PATH)
        want = blo.log_cube(img, np.array([[sigma] * 3]))[..., 0]
        bound = nat.lib().mmx_tiled_q16_error_bound(nat.as_double_ptr(space.w0[0]), nat.as_double_ptr(space.w2[0]),
                                                    R, float(space.norms[0]))
        err = np.abs(got - want).max()
        assert err < bound * vmax, (case, R, err, bound * vmax)
        # what the band of 2.5e-4 covers fourfold (the few-tap kernels of sigma < 1 carry a little more)
        assert bound <= (bl.Q16_BOUND_ANY_SIGMA if R >= 4 else 5.4e-5), (R, bound)
        worst = max(worst, err / vmax)
    assert worst < 5.4e-5              # the north star's LoG tolerance of 1e-4, relative to the value scale, with room


@pytest.mark.parametrize("unsharp,clip_max,expect_q16", [(0.3, 1.0, True), (0.9, 1.0, False), (0.3, 1.6, False)])
def test_sixteen_bit_tiles_only_inside_the_absolute_log_tolerance(gpu, unsharp, clip_max, expect_q16):
    """AUTO picks 16-bit intermediates for preprocessed blocks only while their error bound IN VALUE UNITS
    (``mmx_tiled_q16_error_bound`` x the preprocessed range ``2 clip_max - s clip_min``) stays inside the 1e-4 LoG contract
    (``MMX_LOG_ABS_TOL``): the stock profile (range 1.94) does, a stronger unsharp mask or a wider clip falls back to
    float32 tiles with the narrow band -- and the detection equals the oracle either way."""
    from magellanmapper_amd import _native as nat, blob_log as bl, config, stack_detect, synth
    from oracle import magmap_oracle as mmo
    config.setup_roi_profiles(None)
    config.roi_profile.update(dict(num_sigma=3, denoise_size=25, segment_size=64, unsharp_strength=unsharp,
                                   clip_max=clip_max))
    config.resolutions = np.array([[1.0, 1.0, 1.0]])
    config.filename = "abs_tol"
    vol = synth.make_volume(21, (40, 56, 60), 30)
    try:
        img5d = stack_detect.Image5d(vol[None])
        _, _, blobs = stack_detect.detect_blobs_blocks("abs_tol", img5d, None, None, None, False, False, True, False)
        assert (bl.LAST_ZX_PATH == nat.MMX_ZX_TILED_Q16) == expect_q16, (bl.LAST_ZX_PATH, bl.LAST_Q16_BOUND)
        if expect_q16:
            assert 0 < bl.LAST_Q16_BOUND <= bl.LOG_ABS_TOL
        else:
            assert bl.LAST_ZX_PATH == nat.MMX_ZX_TILED          # float32 tiles of the same kernels
        want, _ = mmo.detect_blobs_blocks(vol, None, [dict(config.roi_profile)], config.resolutions)
        got = blobs.blobs
        assert want is not None and got is not None and got.shape == want.shape
        key = lambda t: t[np.lexsort(tuple(t[:, i] for i in range(t.shape[1] - 1, -1, -1)))]
        np.testing.assert_array_equal(key(got), key(want))
    finally:
        config.setup_roi_profiles(None)


def test_block_shape_and_dtype_sweep_matches_oracle(gpu):
    """Ragged extents (row lengths around the 8-float chunk, the 32-float pitch and the 64-lane wave),
    blocks thinner than the kernel radius (generic fallback per pass), every input dtype; several
    differently shaped blocks in ONE batch."""
    from magellanmapper_amd import blob_log as bl, synth
    from oracle import blob_log_oracle as blo
    rng = np.random.default_rng(21)
    shapes = [(9, 23, 17), (30, 31, 33), (12, 40, 63), (25, 26, 65), (40, 9, 100), (7, 50, 31),
              (33, 33, 129), (5, 6, 7), (20, 70, 257), (27, 11, 300), (26, 10, 330)]   # row pitch 320 / above it
    full = (max(s[0] for s in shapes), max(max(s[1] for s in shapes), 31), max(s[2] for s in shapes))
    all_shapes = shapes
    for dtype in (np.uint16, np.uint8, np.float32, np.float64):
        vol = synth.make_volume(int(rng.integers(1 << 30)), full, 30)
        if dtype == np.uint8:
            vol = (vol >> 8).astype(np.uint8)
        elif dtype != np.uint16:
            vol = (vol / 65535.0).astype(dtype)
        dvol = bl.DeviceVolume(vol)
        origins = [tuple(int(rng.integers(0, f - s + 1)) for f, s in zip(full, shp)) for shp in shapes]
        # (the 330-wide block sends its whole batch through the separate passes; without it the fused
        # kernel takes the batch, pitch-320 row included)
        roomy = [(30, 31, 33), (40, 26, 100), (33, 33, 129), (27, 30, 257), (27, 27, 300)]   # every pass register-resident
        for sigmas, shapes in (((2.3, 2.3), all_shapes), ((3.0, 4.75), all_shapes[:-1]), ((2.4, 4.75), roomy)):
            origins = [tuple(int(rng.integers(0, f - s + 1)) for f, s in zip(full, shp)) for shp in shapes]
            space = bl.ScaleSpace.make(sigmas[0], sigmas[1], 2)
            cubes = bl.log_cube_blocks(dvol, 0, origins, shapes, space)
            for o, shp, got in zip(origins, shapes, cubes):
                sub = vol[o[0]:o[0] + shp[0], o[1]:o[1] + shp[1], o[2]:o[2] + shp[2]]
                want = blo.log_cube(blo.img_as_float(sub), np.stack([space.sigmas] * 3, axis=1))
                err = np.abs(got - want).max()
                assert got.shape == want.shape and err < 5e-6, (dtype, shp, sigmas, err)


def test_blob_log_randomised_parameters_match_oracle(gpu):
    """blob_log end to end on seeded random volumes over the parameter space the profiles span (sigma
    ranges giving odd and even radii, one to many scales, thresholds, overlap limits, blob sizes and
    crowding): the same rows in the same order as the oracle."""
    from magellanmapper_amd import blob_log as bl, synth
    from oracle import blob_log_oracle as blo
    rng = np.random.default_rng(2024)
    total = 0
    for trial in range(14):
        shape = tuple(int(v) for v in rng.integers(18, 56, 3))
        n_blobs = int(rng.integers(3, 40))
        bs = float(rng.uniform(1.0, 4.0))
        vol = synth.make_volume(int(rng.integers(1 << 30)), shape, n_blobs, blob_sigma=bs,
                                amp=float(rng.uniform(8000, 50000)))
        lo = float(rng.uniform(0.8, 3.2))
        hi = lo + float(rng.uniform(0.0, 3.0))
        ns = int(rng.integers(1, 8))
        thr = float(rng.choice([0.02, 0.05, 0.1, 0.2]))
        ov = float(rng.choice([0.0, 0.3, 0.5, 0.9]))
        got = bl.blob_log(vol, lo, hi, ns, thr, ov)
        want = blo.blob_log(vol, lo, hi, ns, thr, ov)
        assert got.shape == want.shape, (trial, shape, lo, hi, ns, thr, ov)
        np.testing.assert_array_equal(got, want, err_msg=str((trial, shape, lo, hi, ns, thr, ov)))
        total += len(want)
    assert total > 100


def test_tail_column_widths_match_oracle(gpu):
    """Block widths just past a multiple of 64 (256 + the 5 overlap columns of the stock block size and
    its neighbours): the fused Z+X kernel gives the last <= 8 columns to its tail wave, and with 257..264
    columns deals the wave roles over the SIMDs (mmx_fused2.hip).  End to end through the NMS bit masks of
    the Y pass and the sparse NMS kernel: the same rows in the same order as the oracle."""
    from magellanmapper_amd import blob_log as bl, synth
    from oracle import blob_log_oracle as blo
    rng = np.random.default_rng(77)
    total = 0
    for width in (257, 261, 264, 265, 256, 65, 72, 129, 193, 200):
        shape = (int(rng.integers(26, 40)), int(rng.integers(26, 44)), width)
        vol = synth.make_volume(int(rng.integers(1 << 30)), shape, 60, blob_sigma=float(rng.uniform(1.5, 3.5)))
        if width % 2:
            vol = (vol >> 8).astype(np.uint8)
        lo = float(rng.uniform(1.0, 3.0))
        hi = lo + float(rng.uniform(0.5, 2.5))
        ns = int(rng.integers(2, 6))
        got = bl.blob_log(vol, lo, hi, ns, 0.05, 0.5)
        want = blo.blob_log(vol, lo, hi, ns, 0.05, 0.5)
        np.testing.assert_array_equal(got, want, err_msg=str((shape, lo, hi, ns)))
        total += len(want)
        space = bl.ScaleSpace.make(lo, hi, ns)
        cube = bl.log_cube_blocks(bl.DeviceVolume(vol), 0, [(0, 0, 0)], [shape], space)[0]
        ref = blo.log_cube(blo.img_as_float(vol), np.stack([space.sigmas] * 3, axis=1))
        assert np.abs(cube - ref).max() < 5e-6, (shape, lo, hi)
    assert total > 200


def test_selective_rescore_gives_the_exact_order(gpu):
    """``exact_values=False`` re-scores only the candidates whose decision needs float64 (contested ones, and
    candidates of a block whose float32 values lie within eps of each other); the others keep their float32
    value as a stand-in.  Peaks and their order must equal the all-exact run, on crowded volumes where
    near-ties are common, and the stand-ins must be the float32 roundings of the exact values' neighbourhood."""
    from magellanmapper_amd import blob_log as bl, synth
    rng = np.random.default_rng(31)
    n_sel = n_all = 0
    for trial in range(6):
        shape = tuple(int(v) for v in rng.integers(40, 90, 3))
        vol = synth.make_volume(int(rng.integers(1 << 30)), shape, int(rng.integers(150, 500)),
                                blob_sigma=float(rng.uniform(1.2, 2.5)), amp=float(rng.uniform(3000, 30000)))
        if trial % 3 == 2:     # many exactly equal blobs: plateaus of near-ties
            vol = np.tile(vol[:shape[0] // 2, :shape[1] // 2, :shape[2] // 2], (2, 2, 2))
        dvol = bl.DeviceVolume(vol)
        args = (dvol, 0, [(0, 0, 0)], [vol.shape], 1.5, 3.0, 4, 0.02, 0.5)
        st_a, st_b = bl.BatchStats(), bl.BatchStats()
        res_a, pk_a = bl.blob_log_blocks(*args, stats=st_a, return_peaks=True)                      # all exact
        res_b, pk_b = bl.blob_log_blocks(*args, stats=st_b, return_peaks=True, exact_values=False)
        if trial % 3 == 2:     # exact ties: an unstable sort may order equal values differently
            np.testing.assert_array_equal(lexsorted(pk_a[0][0]), lexsorted(pk_b[0][0]))
            np.testing.assert_array_equal(lexsorted(res_a[0]), lexsorted(res_b[0]))
        else:
            np.testing.assert_array_equal(pk_a[0][0], pk_b[0][0])
            np.testing.assert_array_equal(res_a[0], res_b[0])
        # (candidates that were not re-scored keep their float32 values: within a quarter of the band of the exact ones)
        assert np.abs(pk_a[0][1] - pk_b[0][1]).max() < (0.25 * bl.EPS_REL_Q16 if bl.LAST_ZX_PATH == 7 else 5e-6)
        assert st_b.n_rescored <= st_b.n_candidates
        n_sel += st_b.n_rescored
        n_all += st_b.n_candidates
    assert 0 < n_sel < n_all


@pytest.mark.parametrize("geometry", ["overlap_beyond_stride", "truncated_far_block", "all_excluded"])
def test_irregular_block_geometry_matches_oracle(gpu, tmp_path, monkeypatch, geometry):
    """Blocks not much larger than their overlap (small ``segment_size``, anisotropic voxels, a large
    ``exclude_border``): the reference's pruning regions then stop tiling the axis -- slabs overlap each other,
    passes are empty, a blob is listed once per region it falls into -- and blocks whose blobs were all
    excluded hold EMPTY tables.  Found by tools/soak_stack.py; the final table must equal the oracle's,
    duplicates included."""
    from magellanmapper_amd import config, stack_detect, synth
    from oracle import magmap_oracle as mmo
    monkeypatch.chdir(tmp_path)
    shape, res, over = {
        "overlap_beyond_stride": ((47, 110, 81), 3.0, dict(segment_size=30, num_sigma=4, detection_threshold=0.05,
                                                          overlap=0.8, exclude_border=(5, 0, 5))),
        "truncated_far_block": ((65, 69, 141), 3.0, dict(segment_size=44, num_sigma=2, detection_threshold=0.2,
                                                        overlap=0.3, exclude_border=(5, 3, 0),
                                                        prune_tol_factor=(0.5, 1.0, 1.5))),
        "all_excluded": ((40, 64, 60), 2.0, dict(segment_size=30, num_sigma=3, detection_threshold=0.5,
                                                 exclude_border=(0, 2, 1))),
    }[geometry]
    vol = synth.make_volume(7, shape, 160 if geometry != "all_excluded" else 2, blob_sigma=2.5)
    config.setup_roi_profiles(None)
    config.roi_profile.update(denoise_size=None, **over)
    config.resolutions = np.array([[res, 1.0, 1.0]])
    config.filename = "irregular"
    blocks = stack_detect.setup_blocks(config.roi_profile, shape)
    regular = all(stack_detect.StackPruner._axis_geometry(a, shape, blocks.overlap, blocks.overlap_padding,
                                                          blocks.sub_roi_slices, blocks.sub_rois_offsets)[1]
                  for a in range(3))
    if geometry != "all_excluded":
        assert not regular
    want, _ = mmo.detect_blobs_blocks(vol, None, [dict(config.roi_profile)], config.resolutions)
    img5d = stack_detect.Image5d(vol[None])
    _, _, blobs = stack_detect.detect_blobs_blocks("irregular", img5d, None, None, None, False, False, True, False)
    if want is None:
        assert blobs.blobs is None
    else:
        assert blobs.blobs is not None and blobs.blobs.shape == want.shape
        np.testing.assert_array_equal(lexsorted(blobs.blobs), lexsorted(want))


GROUPING_CASES = sorted(os.path.basename(p)[len("grouping_"):-4]
                        for p in glob.glob(os.path.join(GOLDEN, "grouping_*.npz")))


@pytest.mark.parametrize("case", GROUPING_CASES)
def test_detect_blobs_stack_channel_grouping_matches_reference(gpu, case, tmp_path, monkeypatch):
    """A15 end to end on the device path: ``detect_blobs_stack`` groups the channels by ``ROIProfile.BLOCK_SIZES``
    (equal ``segment_size``: one grid; unequal ``segment_size`` or ``prune_tol_factor``: one grid per channel),
    final table and saved archive identical to the real reference's."""
    from magellanmapper_amd import config, detector, stack_detect
    monkeypatch.chdir(tmp_path)
    g = load_golden("grouping_%s.npz" % case)
    over = ast.literal_eval(str(g["overrides"]))
    roi = g["roi"]
    config.setup_roi_profiles(None)
    config.roi_profiles = [type(config.roi_profile)(config.roi_profile) for _ in range(roi.shape[3])]
    for i, prof in enumerate(config.roi_profiles):
        prof["denoise_size"] = None
        for k, v in over.items():
            prof[k] = v["per_channel"][i] if isinstance(v, dict) and "per_channel" in v else v
    config.roi_profile = config.roi_profiles[0]
    config.resolutions, config.near_max, config.channel = np.array([[1.0, 1.0, 1.0]]), [-1.0] * roi.shape[3], None
    config.filename = str(tmp_path / "grp")
    try:
        grids = [stack_detect.setup_blocks(config.get_roi_profile(c), roi.shape[:3]).sub_roi_slices.shape
                 for c in range(roi.shape[3])]
        np.testing.assert_array_equal(np.array(grids), g["grids"])
        img5d = stack_detect.Image5d(roi[None])
        img5d.is_roi = True
        _, _, blobs = stack_detect.detect_blobs_stack(str(tmp_path / "grp"), img5d)
        np.testing.assert_array_equal(blobs.blobs, g["final"])
        back = detector.Blobs().load_blobs(str(tmp_path / "grp_blobs.npz"))
        np.testing.assert_array_equal(back.blobs, g["archive_segments"])
    finally:
        config.setup_roi_profiles(None)
        config.resolutions, config.near_max = None, [-1.0]
        detector.Blobs(np.ones((1, 4))).format_blobs()


def test_a_band_narrower_than_the_float32_error_widens_itself(gpu, monkeypatch):
    """The float32 passes only nominate; when their values stray from the exact ones by more than a quarter of the
    nomination band (forced here with an absurdly narrow band) the batch is nominated again with a wider one
    instead of failing -- and the blobs are still the reference's."""
    from magellanmapper_amd import _native as nat, blob_log as bl
    g = load_golden("bloblog_u16_5sigma.npz")
    monkeypatch.setattr(bl, "EPS_REL", 1e-9)
    monkeypatch.setattr(bl, "EPS_REL_Q16", 0.0)           # (no wider band for the 16-bit intermediates: float32 tiles)
    stats = bl.BatchStats()
    dvol = bl.DeviceVolume(g["volume"])
    got = bl.blob_log_blocks(dvol, 0, [(0, 0, 0)], [g["volume"].shape], float(g["min_sigma"]), float(g["max_sigma"]),
                             int(g["num_sigma"]), float(g["threshold"]), float(g["overlap"]), stats=stats)[0]
    assert stats.n_band_retries >= 1 and stats.n_blocks == 1
    np.testing.assert_array_equal(got, g["pruned"])
    # the 16-bit intermediates forced under a band their error (<= 3.0e-5) does not fit: same way out
    monkeypatch.setattr(bl, "EPS_REL_Q16", 2e-5)
    monkeypatch.setattr(bl, "ZX_MODE", nat.MMX_ZX_TILED_Q16)
    stats = bl.BatchStats()
    got = bl.blob_log_blocks(dvol, 0, [(0, 0, 0)], [g["volume"].shape], float(g["min_sigma"]), float(g["max_sigma"]),
                             int(g["num_sigma"]), float(g["threshold"]), float(g["overlap"]), stats=stats)[0]
    assert bl.LAST_ZX_PATH == nat.MMX_ZX_TILED_Q16 and stats.n_band_retries >= 1
    np.testing.assert_array_equal(got, g["pruned"])


def _abi_batch(vol, origins, shapes, sigmas, zx_mode, prepack, thr=0.1, eps=2e-5):
    """One batch straight through the C ABI: ``(path, mask layout, candidates sorted, LoG cubes)``."""
    import ctypes
    import torch
    from magellanmapper_amd import _native as nat, blob_log as bl, kernels1d as k1
    L = nat.lib()
    dvol = bl.DeviceVolume(vol)
    dev = dvol.tensor.device
    blocks, slot = bl._make_blocks(dvol, 0, origins, shapes)
    nb, ns = len(blocks), len(sigmas)
    ws = torch.zeros((4 + ns) * nb * slot, dtype=torch.float32, device=dev)
    d_blocks = bl._to_device_bytes(blocks, dev)
    v32 = dvol.view(0, True)
    stream = torch.cuda.current_stream().cuda_stream
    log_base = ws.data_ptr() + 4 * nb * slot * 4
    mask_words = (nb * slot) >> 5
    masks = torch.zeros(ns * mask_words * 2, dtype=torch.int64, device=dev)
    written, path = ctypes.c_int(0), ctypes.c_int(0)
    mode = zx_mode
    if prepack:
        nat.check(L.mmx_zx_pack(ctypes.byref(v32), d_blocks.data_ptr(), blocks.ctypes.data, nb, slot, ws.data_ptr(),
                                stream), "mmx_zx_pack")
        mode = (nat.MMX_ZX_TILED_Q16 if eps >= 1.5e-4 else nat.MMX_ZX_TILED) | nat.MMX_ZX_PREPACKED
    layouts, paths = set(), set()
    for i, s in enumerate(sigmas):
        R = k1.kernel_radius(s)
        w0, w2 = k1.gaussian_half_kernel(s, 0, R), k1.gaussian_half_kernel(s, 2, R)
        nat.check(L.mmx_log_batch_f32(ctypes.byref(v32), d_blocks.data_ptr(), blocks.ctypes.data, nb, slot,
                                      nat.as_double_ptr(w0), nat.as_double_ptr(w2), R, s * s,
                                      log_base + i * nb * slot * 4, ws.data_ptr(), masks.data_ptr() + i * mask_words * 16,
                                      thr - eps, eps, ctypes.byref(written), mode, ctypes.byref(path), stream), "log")
        layouts.add(written.value)
        paths.add(path.value)
    assert len(layouts) == 1 and len(paths) == 1
    cap = 1 << 16
    table = torch.zeros(cap * nat.CAND_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    layout = layouts.pop()
    nat.check(L.mmx_peaks_batch(log_base, masks.data_ptr(), layout, ns, d_blocks.data_ptr(), blocks.ctypes.data, nb,
                                slot, thr, eps, table.data_ptr(), cap, count.data_ptr(), stream), "peaks")
    torch.cuda.synchronize()
    n = int(count.item())
    assert 0 < n <= cap
    c = table[:n * nat.CAND_DTYPE.itemsize].cpu().numpy().view(nat.CAND_DTYPE)
    key = np.stack([c["slot"], c["s"], c["z"], c["y"], c["x"]], axis=1).astype(np.int64)
    order = np.lexsort(key.T[::-1])
    return paths.pop(), layout, key[order], c["v"][order], c["flags"][order]


def test_tiled_path_entries_and_prepacked_copy_through_the_abi(gpu):
    """``zx_mode`` 6 straight through ``include/mmx.h``: it reports the quad entry layout, ``mmx_peaks_batch`` finds
    exactly the candidates it finds from the packed kernel's row entries (same voxels, same flags, float32 values
    within rounding), and the voxel copy made once by ``mmx_zx_pack`` (``MMX_ZX_PREPACKED``) gives bit-identical
    candidates to the copy every call makes itself.  Ragged blocks of different widths and depths in one batch."""
    from magellanmapper_amd import _native as nat, synth
    vol = synth.make_volume(5, (70, 90, 150), 40)
    origins = [(0, 0, 0), (3, 5, 64), (20, 11, 7)]
    shapes = [(70, 90, 64), (67, 85, 86), (50, 61, 37)]
    sig = [3.0, 3.5, 4.0]
    p2, l2, k2, v2, f2 = _abi_batch(vol, origins, shapes, sig, nat.MMX_ZX_PACKED, False)
    p6, l6, k6, v6, f6 = _abi_batch(vol, origins, shapes, sig, nat.MMX_ZX_TILED, False)
    p6p, l6p, k6p, v6p, f6p = _abi_batch(vol, origins, shapes, sig, nat.MMX_ZX_TILED, True)
    pa, la, ka, va, fa = _abi_batch(vol, origins, shapes, sig, nat.MMX_ZX_AUTO, False)
    assert (p2, l2) == (nat.MMX_ZX_PACKED, nat.MMX_MASK_ROWS)
    assert (p6, l6) == (p6p, l6p) == (pa, la) == (nat.MMX_ZX_TILED, nat.MMX_MASK_QUADS)
    assert len(k2) >= 40
    assert np.array_equal(k2, k6) and np.array_equal(f2, f6)
    assert np.max(np.abs(v2 - v6)) < 2e-6
    assert np.array_equal(k6, k6p) and np.array_equal(v6, v6p) and np.array_equal(f6, f6p)
    assert np.array_equal(k6, ka) and np.array_equal(v6, va)
    # 16-bit intermediates: what AUTO picks once the band covers their rounding error fourfold.  Every candidate of the
    # narrow band is still nominated, its value within the bound the library states for these weights.
    p7, l7, k7, v7, f7 = _abi_batch(vol, origins, shapes, sig, nat.MMX_ZX_AUTO, True, eps=2.5e-4)
    assert (p7, l7) == (nat.MMX_ZX_TILED_Q16, nat.MMX_MASK_QUADS)
    from magellanmapper_amd import kernels1d as k1
    bound = max(nat.lib().mmx_tiled_q16_error_bound(nat.as_double_ptr(k1.gaussian_half_kernel(s_, 0, k1.kernel_radius(s_))),
                                                    nat.as_double_ptr(k1.gaussian_half_kernel(s_, 2, k1.kernel_radius(s_))),
                                                    k1.kernel_radius(s_), s_ * s_) for s_ in sig)
    assert 2.5e-5 < bound < 3.0e-5
    pos = {tuple(r): i for i, r in enumerate(k7)}
    idx = [pos.get(tuple(r), -1) for r in k2]
    assert min(idx) >= 0
    assert np.max(np.abs(v7[idx] - v2)) < bound
    assert np.all(f7 & nat.MMX_CAND_BAND)


def test_rotating_window_of_the_zx_march_against_the_float32_tiles(gpu):
    """Deep blocks, radius 8 and radius 17 / 20 / 24 on 16-bit tiles: the steady steps of ``zx4_kernel`` there keep the
    window's tiles in their slots and rotate the Z fragments (whole turns of 4 and 6 steps; 200 planes = 13 z tiles give
    two turns and one).  The float32 tiles of the same path shift their window as ever: every candidate they nominate in
    the narrow band is nominated from the 16-bit tiles too, with a value inside the bound the library states; and the
    16-bit result is the same from the prepacked copy and on a second run."""
    from magellanmapper_amd import _native as nat, synth, kernels1d as k1
    vol = synth.make_volume(21, (215, 70, 100), 110)
    origins = [(0, 0, 0), (9, 5, 33), (15, 20, 7)]
    shapes = [(200, 64, 64), (206, 60, 67), (187, 50, 37)]
    for sig in ([2.0], [4.25, 5.0, 6.0]):
        p6, l6, k6, v6, f6 = _abi_batch(vol, origins, shapes, sig, nat.MMX_ZX_TILED, False)
        p7, l7, k7, v7, f7 = _abi_batch(vol, origins, shapes, sig, nat.MMX_ZX_TILED_Q16, False, eps=2.5e-4)
        p7b, l7b, k7b, v7b, f7b = _abi_batch(vol, origins, shapes, sig, nat.MMX_ZX_TILED_Q16, True, eps=2.5e-4)
        assert (p6, l6) == (nat.MMX_ZX_TILED, nat.MMX_MASK_QUADS) and (p7, l7) == (p7b, l7b) == (nat.MMX_ZX_TILED_Q16, nat.MMX_MASK_QUADS)
        assert np.array_equal(k7, k7b) and np.array_equal(v7, v7b) and np.array_equal(f7, f7b)
        bound = max(nat.lib().mmx_tiled_q16_error_bound(
            nat.as_double_ptr(k1.gaussian_half_kernel(s_, 0, k1.kernel_radius(s_))),
            nat.as_double_ptr(k1.gaussian_half_kernel(s_, 2, k1.kernel_radius(s_))), k1.kernel_radius(s_), s_ * s_) for s_ in sig)
        assert len(k6) >= 60
        pos = {tuple(r): i for i, r in enumerate(k7)}
        idx = [pos.get(tuple(r), -1) for r in k6]
        assert min(idx) >= 0
        assert np.max(np.abs(v7[idx] - v6)) < bound
        # (both halves of every block's depth hold candidates: the steady steps' tiles are among those compared)
        assert (k6[:, 2] < 60).any() and (k6[:, 2] > 120).any()


def test_y_pass_on_the_matrix_cores_agrees_with_the_valu_kernel(gpu):
    """``MMX_ZX_TILED_Q16`` runs its Y pass on the matrix cores (``ym_kernel``); ``| MMX_ZX_Y_VALU`` asks for ``y6_kernel``
    on the same 16-bit tiles.  Both read the same tiles, so their values differ only by the float32 arithmetic and the low
    x low product the MFMA form leaves out (0.062 counts per unit of weight: under 1e-5 here); the candidates of either are
    those of the other up to the few whose value sits at the edge of the band.  Blocks whose y extent is below one k-block,
    not a multiple of 16, and long; radii of both fragment classes (R <= 16: 4 block offsets, R <= 24: 6)."""
    from magellanmapper_amd import _native as nat, synth
    vol = synth.make_volume(9, (60, 140, 120), 60)
    origins = [(0, 0, 0), (5, 3, 40), (11, 20, 7), (30, 0, 50)]
    shapes = [(40, 29, 64), (50, 37, 70), (33, 120, 37), (30, 140, 26)]
    for sig in ([2.0, 3.5], [4.25, 5.0, 6.0]):
        pm, lm, km, vm, fm = _abi_batch(vol, origins, shapes, sig, nat.MMX_ZX_TILED_Q16, False, eps=2.5e-4)
        pv, lv, kv, vv, fv = _abi_batch(vol, origins, shapes, sig, nat.MMX_ZX_TILED_Q16 | nat.MMX_ZX_Y_VALU, False, eps=2.5e-4)
        pp, lp, kp, vp, fp = _abi_batch(vol, origins, shapes, sig, nat.MMX_ZX_TILED_Q16, True, eps=2.5e-4)
        assert (pm, lm) == (pv, lv) == (pp, lp) == (nat.MMX_ZX_TILED_Q16, nat.MMX_MASK_QUADS)
        assert np.array_equal(km, kp) and np.array_equal(vm, vp) and np.array_equal(fm, fp)      # same kernel, prepacked copy
        a = {tuple(r): i for i, r in enumerate(km)}
        common = [(a[tuple(r)], i) for i, r in enumerate(kv) if tuple(r) in a]
        assert len(km) >= 30 and len(common) >= 0.97 * max(len(km), len(kv)), (len(km), len(kv), len(common))
        ia, ib = np.array(common).T
        assert np.max(np.abs(vm[ia] - vv[ib])) < 1e-5
        # whatever one nominates and the other does not is at the edge of the band: within 1e-5 of a limit either could
        # have put it on the other side of (checked on the values of the kernel that did nominate it)
        assert len(set(map(tuple, km)) ^ set(map(tuple, kv))) <= 0.03 * len(km) + 2


def test_zx_pack_refuses_what_the_tiled_path_cannot_take(gpu):
    """``mmx_zx_pack`` through the ABI: float64 voxels are refused (nothing written; callers hand the float32 copy),
    float32 voxels are split into float16 pieces; the log call takes the tiled path for them when it is asked for by
    name or when the volume states its value range (``mmx_volume.value_range``), and keeps the packed kernel when the
    library knows nothing about the values."""
    import ctypes
    import torch
    from magellanmapper_amd import _native as nat, blob_log as bl, synth
    vol = (synth.make_volume(2, (40, 48, 64), 10) / 65535.0).astype(np.float32)
    dvol = bl.DeviceVolume(vol)
    blocks, slot = bl._make_blocks(dvol, 0, [(0, 0, 0)], [vol.shape])
    ws = torch.zeros(5 * slot, dtype=torch.float32, device=dvol.tensor.device)
    d_blocks = bl._to_device_bytes(blocks, dvol.tensor.device)
    v32 = dvol.view(0, True)
    assert nat.lib().mmx_zx_pack(ctypes.byref(v32), d_blocks.data_ptr(), blocks.ctypes.data, 1, slot, ws.data_ptr(), None) == 0
    v64 = bl.DeviceVolume(vol.astype(np.float64)).view(0, False)
    assert v64.dtype == nat.MMX_F64
    rc = nat.lib().mmx_zx_pack(ctypes.byref(v64), d_blocks.data_ptr(), blocks.ctypes.data, 1, slot, ws.data_ptr(), None)
    assert rc == 5                                  # MMX_ERR_UNSUPPORTED
    assert nat.lib().mmx_zx_pack(None, None, None, 1, slot, None, None) == 1      # MMX_ERR_ARG
    space = bl.ScaleSpace.make(3.0, 3.0, 1)
    want = None
    default = bl.ZX_MODE
    try:
        for mode, path in ((nat.MMX_ZX_PACKED, nat.MMX_ZX_PACKED), (nat.MMX_ZX_AUTO, nat.MMX_ZX_PACKED),
                           (nat.MMX_ZX_TILED, nat.MMX_ZX_TILED)):
            bl.ZX_MODE = mode
            cube = bl.log_cube_blocks(dvol, 0, [(0, 0, 0)], [vol.shape], space)[0]
            assert bl.LAST_ZX_PATH == path, (mode, bl.LAST_ZX_PATH)
            want = cube if want is None else want
            assert np.abs(cube - want).max() < 2e-6
    finally:
        bl.ZX_MODE = default
    # the range stated in the volume: AUTO takes the tiled path, with 16-bit tiles once the band covers their error
    ws2 = torch.zeros(6 * slot + 64, dtype=torch.float32, device=dvol.tensor.device)
    mask = torch.zeros(slot // 2 + 64, dtype=torch.uint8, device=dvol.tensor.device)
    written, path = ctypes.c_int(0), ctypes.c_int(0)
    for rng_, eps, expect in ((1.0, 2e-5, nat.MMX_ZX_TILED), (1.0, 2.5e-4, nat.MMX_ZX_TILED_Q16), (-1.0, 2.5e-4, nat.MMX_ZX_TILED),
                              (0.0, 2.5e-4, nat.MMX_ZX_PACKED)):
        v32.value_range = rng_
        nat.check(nat.lib().mmx_log_batch_f32(
            ctypes.byref(v32), d_blocks.data_ptr(), blocks.ctypes.data, 1, slot, nat.as_double_ptr(space.w0[0]),
            nat.as_double_ptr(space.w2[0]), int(space.radii[0]), float(space.norms[0]), ws2.data_ptr() + 4 * slot * 4,
            ws2.data_ptr(), mask.data_ptr(), 0.1 - eps, eps, ctypes.byref(written), nat.MMX_ZX_AUTO,
            ctypes.byref(path), None), "mmx_log_batch_f32")
        torch.cuda.synchronize()
        assert path.value == expect, (rng_, eps, path.value)


def test_tiled_path_geometry_limits_and_interleaved_channels(gpu):
    """What the tiled kernels take and what they hand back to the packed kernel: a (z, y, x, c) image is read with
    its channel stride by the voxel copy (no alignment or stride rule left), more than eight distinct block widths
    in one batch exceed its fragment tables (the whole batch takes the packed kernel) -- the blobs equal the
    oracle's either way."""
    from magellanmapper_amd import _native as nat, blob_log as bl, synth
    from oracle import blob_log_oracle as blo
    a = synth.make_volume(11, (40, 50, 90), 25)
    b = synth.make_volume(12, (40, 50, 90), 25)
    img = np.stack([a, b], axis=-1)                               # channels interleaved along x in memory
    dvol = bl.DeviceVolume(img)
    for chl, vol in ((0, a), (1, b)):
        got = bl.blob_log_blocks(dvol, chl, [(0, 0, 0), (3, 4, 21)], [(40, 50, 64), (37, 46, 69)], 3.0, 4.0, 3, 0.05, 0.5)
        assert bl.LAST_ZX_PATH == nat.MMX_ZX_TILED_Q16
        for o, shp, res in zip([(0, 0, 0), (3, 4, 21)], [(40, 50, 64), (37, 46, 69)], got):
            sub = vol[o[0]:o[0] + shp[0], o[1]:o[1] + shp[1], o[2]:o[2] + shp[2]]
            np.testing.assert_array_equal(res, blo.blob_log(sub, 3.0, 4.0, 3, 0.05, 0.5))
    widths = [40, 44, 48, 52, 56, 60, 64, 68, 72, 76]             # ten width classes: one more than the tables hold
    shapes = [(40, 50, w) for w in widths]
    got = bl.blob_log_blocks(bl.DeviceVolume(a), 0, [(0, 0, 0)] * len(widths), shapes, 3.0, 4.0, 3, 0.05, 0.5)
    assert bl.LAST_ZX_PATH == nat.MMX_ZX_PACKED
    for shp, res in zip(shapes, got):
        np.testing.assert_array_equal(res, blo.blob_log(a[:, :, :shp[2]], 3.0, 4.0, 3, 0.05, 0.5))


def test_plateau_of_contested_candidates_in_one_batch(gpu, tmp_path, monkeypatch, host_path):
    """Found by tools/soak_stack.py (seed 202, trial 617): spectral unmixing clips whole regions of the second channel
    to 0, every voxel of such a plateau is a contested candidate, and one batch of 125 small blocks asked for the exact
    values of 19 million neighbours in one call -- more workgroups than a one-dimensional grid of 256-thread groups may
    have (2^32 threads): the launch wrapped silently, the missing values read as NaN and two thirds of the channel's
    blobs were dropped.  ``want`` in the fixture is the oracle's table for the dumped volume (two channels, isotropic
    rescale, unmixing, a profile of its own for channel 1); the oracle runs again here."""
    import ast
    from magellanmapper_amd import config, preprocess, stack_detect
    from oracle import magmap_oracle as mmo
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(preprocess, "RGB_GUESS", True)
    g = load_golden("soak_stack_202_617.npz")
    vol, res = g["vol"], g["res"]
    over, unmix, ch1 = (ast.literal_eval(str(g[k])) for k in ("over", "unmix", "ch1"))
    config.setup_roi_profiles(["default"] * 2)
    for p in config.roi_profiles:
        p.update(over)
    config.roi_profiles[1].update(ch1)
    config.roi_profile.update(over)
    monkeypatch.setattr(config, "resolutions", res)
    monkeypatch.setattr(config, "filename", "soak")
    monkeypatch.setattr(config, "near_max", [-1.0, -1.0])
    for p in config.roi_profiles:
        p.spectral_unmixing = unmix
    config.roi_profile.spectral_unmixing = unmix
    profs = [dict(p, spectral_unmixing=unmix) for p in config.roi_profiles]
    want, _ = mmo.detect_blobs_blocks(vol, None, profs, res, near_max=config.near_max, coloc=False)
    _, _, blobs = stack_detect.detect_blobs_blocks("soak", stack_detect.Image5d(vol[None]), None, None, None,
                                                   False, False, True, False)
    srt = lambda t: t[np.lexsort(t.T[::-1])]       # noqa: E731
    np.testing.assert_array_equal(srt(want), srt(g["want"]))
    assert stack_detect.StackDetector.last_stats.n_probes > (1 << 24)        # the case the fix is about
    np.testing.assert_array_equal(srt(blobs.blobs), srt(want))


def test_float_voxels_take_the_tiled_path(gpu):
    """Float images of ordinary magnitude (and every preprocessed block) run the matrix-core Z+X kernel on a copy that
    holds each voxel as two float16 pieces: every kernel radius against the float64 oracle cube, blob_log identical
    to the oracle with the path reported (also for a faint image: the nomination band is absolute, never below 2e-5);
    a float image whose values exceed what float16 pieces hold keeps the packed kernel."""
    from magellanmapper_amd import _native as nat, blob_log as bl, synth
    from oracle import blob_log_oracle as blo
    rng = np.random.default_rng(41)
    vol = (synth.make_volume(4, (38, 45, 70), 14).astype(np.float32) / 65535.0 * 1.3 - 0.05).astype(np.float32)
    vol += rng.normal(0, 1e-3, vol.shape).astype(np.float32)
    dvol = bl.DeviceVolume(vol)
    default, bl.ZX_MODE = bl.ZX_MODE, nat.MMX_ZX_TILED
    try:
        for R in range(1, 25):
            sigma = (R + 0.2) / 4.0
            space = bl.ScaleSpace.make(sigma, sigma, 1)
            got = np.squeeze(bl.log_cube_blocks(dvol, 0, [(0, 0, 0)], [vol.shape], space)[0])
            want = blo.log_cube(vol.astype(np.float64), np.array([[sigma] * 3]))[..., 0]
            assert bl.LAST_ZX_PATH == nat.MMX_ZX_TILED, R
            assert np.abs(got - want).max() < LOG_TOL * 1e-2 * 1.3, R
    finally:
        bl.ZX_MODE = default
    for scale, path in ((1.0, nat.MMX_ZX_TILED), (3.0e5, nat.MMX_ZX_PACKED), (1.0e-4, nat.MMX_ZX_TILED)):
        img = (vol * np.float32(scale)).astype(np.float32)
        want, st = blo.blob_log(img, 3, 5, 3, 0.1 * scale, 0.5, return_stages=True)
        stats = bl.BatchStats()
        res, peaks = bl.blob_log_blocks(bl.DeviceVolume(img), 0, [(0, 0, 0)], [img.shape], 3, 5, 3, 0.1 * scale, 0.5,
                                        stats=stats, return_peaks=True)
        assert bl.LAST_ZX_PATH == path, (scale, bl.LAST_ZX_PATH)
        assert len(want) > 5
        np.testing.assert_array_equal(peaks[0][0], st["peaks"].reshape(-1, 4))
        np.testing.assert_array_equal(peaks[0][1], st["peak_values"].astype(np.float64))
        np.testing.assert_array_equal(res[0], want)
        assert stats.max_f32_error < 5e-6 * max(1.0, scale * 1.3)


# ---------------------------------------------------------------- mmx_detect_batch: one native call per batch
def _peaks_of(bl, dvol, g, stats=None):
    res, peaks = bl.blob_log_blocks(
        dvol, 0, [(0, 0, 0)], [g["volume"].shape], float(g["min_sigma"]), float(g["max_sigma"]),
        int(g["num_sigma"]), float(g["threshold"]), float(g["overlap"]), stats=stats, return_peaks=True)
    return res[0], peaks[0]


@pytest.mark.parametrize("case", ["u16_5sigma", "u8_3sigma", "f32_2sigma", "f64_2sigma", "u16_thin", "u16_empty"])
def test_one_native_call_per_batch_equals_the_call_by_call_form(gpu, case, monkeypatch):
    """``mmx_detect_batch`` (SURVEY.md 8b's fused A0-A4 entry: voxel copy, every scale, NMS, probes, exact re-score and
    the copies enqueued by native code) against the same launches made one ctypes call at a time from Python, and
    against the real scikit-image: ordered peaks, bit-equal float64 values, pruned blobs."""
    from magellanmapper_amd import blob_log as bl
    g = load_golden("bloblog_%s.npz" % case)
    dvol = bl.DeviceVolume(g["volume"])
    monkeypatch.setattr(bl, "GRAPH_BLOCKS", 0)
    monkeypatch.setattr(bl, "NATIVE_BATCH", False)
    res_a, (coords_a, vals_a) = _peaks_of(bl, dvol, g)
    path_a = bl.LAST_ZX_PATH
    monkeypatch.setattr(bl, "NATIVE_BATCH", True)
    st = bl.BatchStats()
    res_b, (coords_b, vals_b) = _peaks_of(bl, dvol, g, st)
    assert bl.LAST_ZX_PATH == path_a
    np.testing.assert_array_equal(coords_b, coords_a)
    np.testing.assert_array_equal(vals_b, vals_a)
    np.testing.assert_array_equal(res_b, res_a)
    np.testing.assert_array_equal(coords_b, g["peaks"].reshape(-1, 4))            # real skimage
    np.testing.assert_array_equal(res_b, g["pruned"])


def test_small_batches_replay_a_captured_graph(gpu, monkeypatch):
    """A small volume detected again and again with the same buffers (``bench.py --config c2``'s step): the second
    sighting of a batch captures its launches as a hipGraph, later ones replay it -- same peaks every time, and the
    replay refuses to hide kernels from the per-kernel timing."""
    from magellanmapper_amd import _native as nat, blob_log as bl
    g = load_golden("bloblog_u16_5sigma.npz")
    dvol = bl.DeviceVolume(g["volume"])
    bl.release_buffers()
    nat.timing_enable(False)            # (another test's timing window must not be open: a replay would hide from it)
    monkeypatch.setattr(bl, "GRAPH_BLOCKS", 8)
    monkeypatch.setattr(bl, "NATIVE_BATCH", True)
    monkeypatch.setattr(bl, "HOST_PATH", "native")
    replays0 = bl.GRAPH_REPLAYS
    runs = [_peaks_of(bl, dvol, g) for _ in range(4)]
    bufs = bl._buffers_for(dvol.tensor.device)
    captured = [hit for hit in bufs.graphs.values() if hit]
    # one key: seen, captured, replayed twice
    assert len(captured) == 1 and captured[0][0], (len(bufs.graphs), nat.lib().mmx_timing_is_enabled())
    assert bl.GRAPH_REPLAYS - replays0 == 3
    for res, (coords, vals) in runs:
        np.testing.assert_array_equal(coords, g["peaks"].reshape(-1, 4))
        np.testing.assert_array_equal(vals, runs[0][1][1])
        np.testing.assert_array_equal(res, g["pruned"])
    # another band is another key: no stale replay
    monkeypatch.setattr(bl, "EPS_REL_Q16", 3e-4)
    res, (coords, _) = _peaks_of(bl, dvol, g)
    np.testing.assert_array_equal(coords, g["peaks"].reshape(-1, 4))
    assert len(bufs.graphs) == 2
    # with the per-kernel timing on the launches are made one by one (and show up in it)
    monkeypatch.setattr(bl, "EPS_REL_Q16", 2.5e-4)
    nat.timing_enable(True)
    try:
        _peaks_of(bl, dvol, g)
        assert nat.timing_read()["zxpass"][1] == int(g["num_sigma"])
    finally:
        nat.timing_enable(False)
    bl.release_buffers()


def test_detect_batch_through_the_abi(gpu):
    """``mmx_detect_batch`` driven from raw pointers, as a foreign binding would (no blob_log): two ragged uint16 blocks,
    three scales; the table it leaves in pinned host memory holds every peak of the real scikit-image with bit-equal
    float64 values; a too small table reports its overflow in the counters; bad arguments are refused."""
    import ctypes
    import torch
    from magellanmapper_amd import _native as nat, blob_log as bl
    from oracle import blob_log_oracle as blo
    L = nat.lib()
    g = load_golden("bloblog_u16_5sigma.npz")
    vol = g["volume"]
    dev = torch.device("cuda", 0)
    dvol = bl.DeviceVolume(vol)
    space = bl.ScaleSpace.make(float(g["min_sigma"]), float(g["max_sigma"]), 3)
    origins, shapes = [(0, 0, 0), (3, 2, 5)], [vol.shape, (vol.shape[0] - 3, vol.shape[1] - 4, vol.shape[2] - 9)]
    blocks, slot = bl._make_blocks(dvol, 0, origins, shapes)
    d_blocks = bl._to_device_bytes(blocks, dev)
    nb, ns = 2, 3
    ws = torch.empty(-(-int(L.mmx_workspace_bytes(nb, slot, ns, 1)) // 4), dtype=torch.float32, device=dev)
    d_w0, d_w2 = space.device_tables(dev)
    cap = 20000
    table = torch.zeros(cap * nat.CAND_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    count = torch.zeros(2, dtype=torch.int32, device=dev)
    h_count = torch.zeros(2, dtype=torch.int32).pin_memory()
    h_table = torch.zeros(cap * nat.CAND_DTYPE.itemsize, dtype=torch.uint8).pin_memory()
    v32, vex = dvol.view(0, True), dvol.view(0, False)
    thr, eps = float(g["threshold"]), 2.5e-4
    ev = bl._NativeEvent()
    stream = torch.cuda.current_stream().cuda_stream

    def args(cap_now):
        a = nat.DetectArgs()
        a.vol32, a.vol_exact = ctypes.pointer(v32), ctypes.pointer(vex)
        a.d_blocks, a.h_blocks, a.n_blocks, a.n_sigma, a.slot_elems = d_blocks.data_ptr(), blocks.ctypes.data, nb, ns, slot
        a.h_w0, a.h_w2, a.d_w0, a.d_w2 = space.w0_tab.ctypes.data, space.w2_tab.ctypes.data, d_w0.data_ptr(), d_w2.data_ptr()
        a.h_radius, a.h_norm = space.radii.ctypes.data, space.norms.ctypes.data
        a.d_work, a.work_bytes, a.thr, a.eps = ws.data_ptr(), ws.numel() * 4, thr, eps
        a.d_cands, a.cap, a.h_prefix, a.d_count = table.data_ptr(), cap_now, cap_now, count.data_ptr()
        a.h_count, a.h_cands = h_count.data_ptr(), h_table.data_ptr()
        a.zx_mode, a.zx_flags, a.store_f32, a.exact, a.expand = nat.MMX_ZX_AUTO, 0, 0, 1, 1
        a.stream = a.tail_stream = a.pack_stream = stream
        a.ev_done = ev.handle
        return a
    info = nat.DetectInfo()
    nat.check(L.mmx_detect_batch(ctypes.byref(args(cap)), ctypes.byref(info)), "mmx_detect_batch")
    ev.synchronize()
    assert info.zx_path == nat.MMX_ZX_TILED_Q16 and info.mask_layout == nat.MMX_MASK_QUADS and info.n_pass_rounds == 1
    assert 0 < info.q16_bound <= 3.0e-5
    n_all, n_cands = (int(v) for v in h_count.numpy().view(np.uint32))
    assert 0 < n_cands <= n_all <= cap
    cands = h_table.numpy()[:n_all * nat.CAND_DTYPE.itemsize].view(nat.CAND_DTYPE)
    for b, (o, shp) in enumerate(zip(origins, shapes)):
        sub = vol[o[0]:o[0] + shp[0], o[1]:o[1] + shp[1], o[2]:o[2] + shp[2]]
        cube = blo.log_cube(blo.img_as_float(sub), np.stack([space.sigmas] * 3, axis=1))
        want = blo.peak_coords(cube, blo.peak_mask(cube, thr))       # the real rule on the float64 cube
        mine = cands[:n_cands][cands["slot"][:n_cands] == b]
        have = {(int(c["z"]), int(c["y"]), int(c["x"]), int(c["s"])): float(c["v64"]) for c in mine}
        assert len(want) > 20
        for z, y, x, s_ in want:
            assert (z, y, x, s_) in have                              # every true peak was nominated ...
            assert have[(z, y, x, s_)] == cube[z, y, x, s_]          # ... and re-scored bit for bit
    # a table too small: the counters say how many entries there would have been
    nat.check(L.mmx_detect_batch(ctypes.byref(args(8)), ctypes.byref(info)), "mmx_detect_batch")
    ev.synchronize()
    assert int(h_count.numpy().view(np.uint32)[0]) > 8
    # refused: no workspace, a workspace too small
    bad = args(cap)
    bad.d_work = None
    assert L.mmx_detect_batch(ctypes.byref(bad), ctypes.byref(info)) == 1
    bad = args(cap)
    bad.work_bytes = 1024
    assert L.mmx_detect_batch(ctypes.byref(bad), ctypes.byref(info)) == 4      # MMX_ERR_WORKSPACE
    torch.cuda.synchronize()
