"""GPU parity tests of the per-block preprocessing: the HIP kernels
(through the C ABI) against the golden vectors of the real reference's ``plot_3d.saturate_roi`` /
``plot_3d.denoise_roi`` and against the CPU oracle.  Everything is float64 and compared BIT FOR BIT.
"""
import ast

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

PREPROC = load_golden("preproc.npz")
CASES = [str(n) for n in PREPROC["names"]]


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: torch.cuda.is_available() is False")
    from magellanmapper_amd import _native
    assert _native.lib().mmx_device_count() >= 1
    return torch.device("cuda", 0)


@pytest.fixture
def env(monkeypatch):
    """scikit-image 0.18.3 + NumPy 1.26 behaviour of the fixtures (see golden_preproc_env)."""
    from magellanmapper_amd import config, preprocess
    from oracle import preprocess_oracle as ppo
    monkeypatch.setattr(preprocess, "RGB_GUESS", True)
    monkeypatch.setattr(preprocess, "GAUSS_WEIGHTS_OVERRIDE",
                        np.ascontiguousarray(PREPROC["gauss8_weights"][32:]))
    monkeypatch.setattr(ppo, "GAUSS_WEIGHTS", PREPROC["gauss8_weights"])
    yield
    config.setup_roi_profiles(None)
    config.near_max = [-1.0]


def _set_profiles(over, n=1):
    from magellanmapper_amd import config
    config.setup_roi_profiles(["default"] * n)
    for i, p in enumerate(config.roi_profiles):
        for k, v in over.items():
            p[k] = v["per_channel"][i] if isinstance(v, dict) and "per_channel" in v else v
    return [dict(p) for p in config.roi_profiles]


@pytest.mark.parametrize("case", CASES)
def test_tile_matches_reference(gpu, env, case):
    """One tile == one call of saturate_roi + denoise_roi of the real reference."""
    from magellanmapper_amd import _native as nat, preprocess
    g = PREPROC
    roi = g[case + "_roi"]
    over = ast.literal_eval(str(g[case + "_over"]))
    _set_profiles(over, 2 if case == "2ch_perchl" else 1)
    near_max = list(g[case + "_near_max"])
    got, infos = preprocess.preprocess_roi(roi, roi.shape[:3], near_max=near_max, return_info=True)
    assert got.dtype == np.float64
    np.testing.assert_array_equal(got, g[case + "_den"])
    # the percentiles themselves, against this NumPy
    for c, (subs, info) in enumerate(infos):
        plane = roi[..., c] if roi.ndim == 4 else roi
        prof = _set_profiles(over, 2 if case == "2ch_perchl" else 1)[c if case == "2ch_perchl" else 0]
        vmin, vmax = np.percentile(plane, (prof["clip_vmin"], prof["clip_vmax"]))
        assert len(info) == 1 and info["vmin"][0] == vmin
        ident = bool(info["flags"][0] & nat.MMX_PP_IDENTITY)
        assert ident == (vmin == vmax)
        if not ident:
            assert info["vmax"][0] == max(vmax, near_max[c] * prof["max_thresh_factor"])
            sat = g[case + "_sat"][..., c] if roi.ndim == 4 else g[case + "_sat"]
            assert abs(info["mean"][0] - np.mean(sat)) < 1e-12


PREPROC_F64 = load_golden("preproc_f64.npz")


@pytest.mark.parametrize("case", [str(n) for n in PREPROC_F64["names"]])
def test_float64_tile_matches_reference(gpu, env, case):
    """FLOAT64 images (the part that raised until round 3): the order statistics come from
    an eight-level radix select on the doubles' bit patterns, an identity tile of negative values is left alone."""
    from magellanmapper_amd import _native as nat, preprocess
    g = PREPROC_F64
    roi = g[case + "_roi"]
    over = ast.literal_eval(str(g[case + "_over"]))
    prof = _set_profiles(over)[0]
    near_max = list(g[case + "_near_max"])
    got, infos = preprocess.preprocess_roi(roi, roi.shape[:3], near_max=near_max, return_info=True)
    assert got.dtype == np.float64
    np.testing.assert_array_equal(got, g[case + "_den"])
    (subs, info), = infos
    vmin, vmax = np.percentile(roi, (prof["clip_vmin"], prof["clip_vmax"]))
    assert len(info) == 1 and info["vmin"][0] == vmin
    assert bool(info["flags"][0] & nat.MMX_PP_IDENTITY) == (vmin == vmax)
    if vmin != vmax:
        assert info["vmax"][0] == max(vmax, near_max[0] * prof["max_thresh_factor"])


def test_float64_block_tiles_match_oracle(gpu, env):
    """A float64 block of several ragged tiles against the oracle's tile loop, bit for bit."""
    from magellanmapper_amd import preprocess
    from oracle import preprocess_oracle as ppo
    g = load_golden("stack_denoise_f64.npz")
    roi = np.ascontiguousarray(g["roi"][:40, :45, :52]) * 3.0 - 0.4
    profs = _set_profiles({})
    got = preprocess.preprocess_roi(roi, (25, 20, 30), near_max=[-1.0])
    want = ppo.preprocess_block(roi, (25, 20, 30), profs, [-1.0])
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("dms", [(25, 25, 25), (13, 25, 30), (7, 40, 9), (50, 64, 33), (64, 96, 96)])
def test_tiled_block_matches_oracle(gpu, env, dms):
    """A whole block, tiled like chunking.stack_splitter does: fast (LDS) and generic tiles mixed."""
    from magellanmapper_amd import preprocess
    from oracle import preprocess_oracle as ppo
    roi = load_golden("stack_denoise_dense.npz")["roi"][:50, :70, :66]
    profs = _set_profiles({})
    want = ppo.preprocess_block(roi, dms, profs, [30000.0])
    got = preprocess.preprocess_roi(roi, dms, near_max=[30000.0])
    np.testing.assert_array_equal(got, want)


def test_generic_kernel_equals_fast_kernel(gpu, env, monkeypatch):
    from magellanmapper_amd import preprocess
    roi = load_golden("stack_denoise.npz")["roi"][:40, :50, :52]
    _set_profiles({})
    fast = preprocess.preprocess_roi(roi, (25, 25, 25))
    for mode in (True, "big"):           # register-line kernel over a global scratch / one output per lane
        monkeypatch.setattr(preprocess, "FORCE_GENERIC", mode)
        slow = preprocess.preprocess_roi(roi, (25, 25, 25))
        np.testing.assert_array_equal(fast, slow)


def _synthetic_block(shape, seed, dtype=np.uint16, flat=()):
    """Background + blobs like the benchmark volume (tiles with and without erosion); ``flat`` = (z0, y0, x0, n)
    cubes of one value (vmin == vmax tiles: the reference leaves those voxels alone)."""
    rng = np.random.default_rng(seed)
    top = np.iinfo(dtype).max
    vol = rng.normal(500 * top / 65535, 50 * top / 65535, shape)
    zz, yy, xx = np.meshgrid(*[np.arange(n) for n in shape], indexing="ij")
    for _ in range(max(3, int(np.prod(shape) * 2e-4))):
        c = [rng.uniform(0, n) for n in shape]
        vol = np.maximum(vol, 0.6 * top * np.exp(-((zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2) / 18.0))
    vol = np.clip(vol, 0, top).astype(dtype)
    for z0, y0, x0, n in flat:
        vol[z0:z0 + n, y0:y0 + n, x0:x0 + n] = 777 % top
    return vol


@pytest.mark.parametrize("tpw", [1, 3, 8])
@pytest.mark.parametrize("case", ["u16", "u8", "no_unsharp", "no_erosion", "always_eroded", "ragged"])
def test_pipelined_kernels_equal_the_single_kernel_and_the_oracle(gpu, env, monkeypatch, case, tpw):
    """The statistics + blur kernels over a tile-major copy (`mmx_preproc_pipe.hip`) against one kernel per tile and the
    oracle, bit for bit: 25^3 tiles and the ragged ones a 58 x 61 x 83 block leaves, eroded / plain / identity tiles in
    one run (deferred and immediate output stores), runs of 1 / 3 / 8 tiles per workgroup."""
    from magellanmapper_amd import _native as nat, preprocess
    from oracle import preprocess_oracle as ppo
    shape, dms = ((58, 61, 83), (25, 25, 25)) if case != "ragged" else ((40, 57, 70), (18, 25, 31))
    roi = _synthetic_block(shape, 11, np.uint8 if case == "u8" else np.uint16, flat=((0, 0, 0, 25), (25, 25, 50, 25)))
    over = {"no_unsharp": {"unsharp_strength": 0}, "no_erosion": {"erosion_threshold": 0},
            "always_eroded": {"erosion_threshold": 1e-6}}.get(case, {})
    profs = _set_profiles(over)
    want = ppo.preprocess_block(roi, dms, profs, [-1.0])
    outs, infos = {}, {}
    for name, mode in (("single", nat.MMX_PP_SINGLE), ("pipelined", nat.MMX_PP_PIPELINED), ("auto", nat.MMX_PP_AUTO)):
        monkeypatch.setattr(preprocess, "KERNEL_MODE", mode)
        monkeypatch.setattr(preprocess, "TILES_PER_WG", tpw)
        outs[name], info = preprocess.preprocess_roi(roi, dms, return_info=True)
        infos[name] = np.concatenate([i[1] for i in info])
    np.testing.assert_array_equal(outs["single"], want)
    np.testing.assert_array_equal(outs["pipelined"], want)
    np.testing.assert_array_equal(outs["auto"], want)
    a, b = infos["single"], infos["pipelined"]
    np.testing.assert_array_equal(a["flags"], b["flags"])
    np.testing.assert_array_equal(a["vmin"], b["vmin"])
    np.testing.assert_array_equal(a["vmax"], b["vmax"])
    np.testing.assert_allclose(a["mean"], b["mean"], rtol=0, atol=1e-12)
    assert (a["flags"] & nat.MMX_PP_IDENTITY).any() or case in ("u8", "ragged")
    if case == "always_eroded":
        assert (a["flags"] & nat.MMX_PP_ERODED).all()
    if case == "u16":
        assert 0 < int((a["flags"] & nat.MMX_PP_ERODED != 0).sum()) < len(a)


def test_pipelined_kernels_with_caller_and_pool_workspace(gpu, env):
    """`mmx_preprocess_batch_mode` through the raw ABI: with the caller's workspace and records, with neither (both
    come from the stream's memory pool), and with a workspace that is too small (MMX_ERR_WORKSPACE)."""
    import ctypes
    from magellanmapper_amd import _native as nat, blob_log as bl, preprocess
    profs = _set_profiles({})
    roi = _synthetic_block((50, 50, 50), 3)
    want = preprocess.preprocess_roi(roi, (25, 25, 25))
    L = nat.lib()
    dev = torch.device("cuda", 0)
    dvol = bl.DeviceVolume(roi)
    pre = preprocess.Preprocessor((25, 25, 25), want_info=True)
    pre.run(dvol, 0, [(0, 0, 0)], [roi.shape], 0)              # builds the tile table and the outputs once
    subs = pre.last_subs
    params, _, _ = preprocess.channel_params(0, None)
    d_subs = bl._to_device_bytes(subs, dev)
    qc = np.array(pre._qc_rows, dtype=nat.QCLASS_DTYPE)
    d_qc = bl._to_device_bytes(qc, dev)
    d_w = torch.from_numpy(preprocess.gauss_weights()).to(dev)
    slot_pre, dst_sz, dst_sy, out64, out32 = pre.last_geometry
    vol = dvol.view(0, False)
    stream = torch.cuda.current_stream().cuda_stream

    def call(work, work_bytes, info):
        o64, o32 = torch.zeros_like(out64), torch.zeros_like(out32)
        rc = L.mmx_preprocess_batch_mode(
            ctypes.byref(vol), d_subs.data_ptr(), subs.ctypes.data, len(subs), d_qc.data_ptr(), len(qc),
            ctypes.byref(params), d_w.data_ptr(), dst_sy, dst_sz, o32.data_ptr(), o64.data_ptr(), info,
            nat.MMX_PP_PIPELINED, 0, work, work_bytes, stream)
        torch.cuda.synchronize()
        return rc, o64

    need = int(L.mmx_preprocess_work_bytes(subs.ctypes.data, len(subs)))
    assert need >= 2 * roi.size
    work = torch.empty(need, dtype=torch.uint8, device=dev)
    d_info = torch.zeros(len(subs) * nat.SUBINFO_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    for args in ((work.data_ptr(), need, d_info.data_ptr()), (None, 0, None)):
        rc, o64 = call(*args)
        assert rc == 0
        got = torch.as_strided(o64[:slot_pre], roi.shape, (dst_sz, dst_sy, 1)).cpu().numpy()
        np.testing.assert_array_equal(got, want)
    rc, _ = call(work.data_ptr(), need - 1024, d_info.data_ptr())
    assert rc == 4          # MMX_ERR_WORKSPACE


@pytest.mark.parametrize("force_generic", [False, True, "big"])
def test_knife_edge_mean_uses_numpys_summation_order(gpu, env, monkeypatch, force_generic):
    """erosion_threshold set exactly AT the tile mean (and one ulp below): the device must
    reproduce np.mean bit for bit to take the reference's branch."""
    from magellanmapper_amd import _native as nat, preprocess
    from oracle import preprocess_oracle as ppo
    monkeypatch.setattr(preprocess, "FORCE_GENERIC", force_generic)
    roi = PREPROC["dense_roi"]
    profs = _set_profiles({})
    mean = float(np.mean(ppo.saturate_roi(roi, profs, [-1.0])))
    for thr, eroded in ((mean, False), (np.nextafter(mean, 0.0), True), (np.nextafter(mean, 1.0), False)):
        profs = _set_profiles({"erosion_threshold": thr})
        want = ppo.denoise_roi(ppo.saturate_roi(roi, profs, [-1.0]), profs)
        got, infos = preprocess.preprocess_roi(roi, roi.shape, return_info=True)
        info = infos[0][1]
        assert info["flags"][0] & nat.MMX_PP_EXACT_MEAN
        assert bool(info["flags"][0] & nat.MMX_PP_ERODED) == eroded
        assert info["mean"][0] == mean
        np.testing.assert_array_equal(got, want)


def test_exact_mean_on_awkward_sizes(gpu, env):
    """NumPy's pairwise summation has three regimes (n < 8, n <= 128, recursive halving)."""
    from magellanmapper_amd import _native as nat, preprocess
    from oracle import preprocess_oracle as ppo
    rng = np.random.default_rng(5)
    for shape in ((1, 1, 5), (1, 7, 9), (3, 5, 9), (2, 8, 8), (5, 11, 13), (9, 17, 23), (25, 25, 25)):
        roi = rng.integers(0, 4000, shape).astype(np.uint16)
        profs = _set_profiles({})
        sat = ppo.saturate_roi(roi, profs, [-1.0])
        if sat.dtype != np.float64:
            continue
        mean = float(np.mean(sat))
        _set_profiles({"erosion_threshold": mean})
        _, infos = preprocess.preprocess_roi(roi, shape, return_info=True)
        info = infos[0][1]
        assert info["flags"][0] & nat.MMX_PP_EXACT_MEAN, shape
        assert info["mean"][0] == mean, shape


def test_unsupported_inputs_fail_loudly(gpu, env):
    from magellanmapper_amd import preprocess
    _set_profiles({})
    with pytest.raises(NotImplementedError):
        preprocess.preprocess_roi(np.zeros((4, 4, 4), np.float32), (4, 4, 4))
    _set_profiles({"clip_vmax": 101})
    with pytest.raises(ValueError):
        preprocess.preprocess_roi(np.zeros((4, 4, 4), np.uint16), (4, 4, 4))
    _set_profiles({})
    with pytest.raises(IndexError):                 # one near_max per channel, as in the reference
        preprocess.preprocess_roi(np.zeros((4, 4, 4, 2), np.uint16), (4, 4, 4), near_max=[-1.0])


def test_detect_sub_roi_with_denoise(gpu, env):
    """The single-block entry (StackDetector.detect_sub_roi) with preprocessing on."""
    from magellanmapper_amd import config, stack_detect
    from oracle import magmap_oracle as mmo
    g = load_golden("stack_denoise.npz")
    roi = g["roi"][:44, :60, :62]
    config.setup_roi_profiles(None)
    config.roi_profile.update(num_sigma=4)
    config.resolutions = np.array([[1.0, 1.0, 1.0]])
    excl = np.array([1, 2, 2])
    coord, got = stack_detect.StackDetector.detect_sub_roi(
        (0, 1, 0), (0, 10, 0), (1, 1, 1), np.array([25, 25, 25]), excl, None, roi, None)
    want = mmo.detect_sub_roi((0, 1, 0), (0, 10, 0), (1, 1, 1), excl, roi, None,
                              [dict(config.roi_profile)], config.resolutions,
                              denoise_max_shape=np.array([25, 25, 25]))
    assert coord == (0, 1, 0) and want is not None
    np.testing.assert_array_equal(got, want)


def test_unmixing_on_preprocessed_blocks_matches_oracle(gpu, env, tmp_path, monkeypatch):
    """Spectral unmixing (detector.py:910-921) after the per-block preprocessing, whole stack, 3 channels:
    the detected channel is (preprocessed c1) - 0.5 * (preprocessed c0), clipped at 0."""
    from magellanmapper_amd import config, stack_detect
    from oracle import magmap_oracle as mmo
    monkeypatch.chdir(tmp_path)
    roi = load_golden("stack_coloc_3ch.npz")["roi"]
    unmix = {1: {0: 0.5}, 2: {1: 0.25, 0: 0.25}}
    config.setup_roi_profiles(None)
    config.roi_profile.update(num_sigma=3, segment_size=30, denoise_size=25)
    config.roi_profile.spectral_unmixing = unmix
    config.resolutions = np.array([[1.0, 1.0, 1.0]])
    config.near_max = [-1.0, -1.0, -1.0]
    config.filename = "unmix"
    try:
        _, _, blobs = stack_detect.detect_blobs_blocks("unmix", stack_detect.Image5d(roi[None]), None, None,
                                                       None, False, False, True, False)
        prof = dict(config.roi_profile, spectral_unmixing=unmix)
        want, _ = mmo.detect_blobs_blocks(roi, None, [prof], config.resolutions, near_max=config.near_max)
        assert want is not None and len(np.unique(want[:, 6])) == 3
        np.testing.assert_array_equal(blobs.blobs, want)
    finally:
        config.roi_profile.spectral_unmixing = None


ISO = load_golden("isotropic.npz")


@pytest.mark.parametrize("case", [str(n) for n in ISO["names"]])
def test_make_isotropic_matches_reference(gpu, case):
    """R1: the device rescale == the real reference's cv_nd.make_isotropic (uint16 truncation, float64 bit
    for bit, clip to the range of the whole multichannel block)."""
    from magellanmapper_amd import preprocess
    got = preprocess.make_isotropic(ISO[case + "_roi"], ISO[case + "_scale"], ISO[case + "_res"])
    want = ISO[case + "_out"]
    assert got.dtype == want.dtype and got.shape == want.shape
    np.testing.assert_array_equal(got, want)


def test_isotropic_z_only_single_channel_matches_scipy_zoom(gpu):
    """The stock lightsheet shape (single channel, z rescaled only): scikit-image 0.18.3 cannot pin it
    (it takes its 2-D warp there), the oracle's SciPy call -- the pinned release's code path -- can."""
    from magellanmapper_amd import preprocess
    from oracle import isotropic_oracle
    rng = np.random.default_rng(11)
    roi = rng.integers(0, 65535, (13, 40, 37)).astype(np.uint16)
    for scale, res in (((0.96, 1, 1), (5.0, 1.0, 1.0)), ((1, 1, 1), (2.0, 1.0, 1.0)), ((0.96, 1, 1), (1.0, 1.0, 1.0))):
        got = preprocess.make_isotropic(roi, scale, np.array(res))
        np.testing.assert_array_equal(got, isotropic_oracle.make_isotropic(roi, scale, np.array(res)))
    f = rng.random((9, 21, 19)) * 2.5 - 0.4
    np.testing.assert_array_equal(preprocess.make_isotropic(f, (1, 1, 1), np.array((3.3, 1.0, 1.0))),
                                  isotropic_oracle.make_isotropic(f, (1, 1, 1), np.array((3.3, 1.0, 1.0))))


@pytest.mark.parametrize("case", ["2ch_z", "2ch_f64", "2ch_slight"])
def test_isotropic_z_only_single_channel_equals_each_channel_of_the_fixtures(gpu, case):
    """The device rescale of ONE channel along z (the stock ``lightsheet`` shape) against the real reference's
    two-channel results, channel by channel: the interpolation never mixes channels, and for two-channel blocks
    scikit-image 0.18.3 runs the pinned release's code path (see the oracle test of the same name)."""
    from magellanmapper_amd import preprocess
    roi, want = ISO[case + "_roi"], ISO[case + "_out"]
    for c in range(roi.shape[3]):
        got = preprocess.make_isotropic(np.ascontiguousarray(roi[..., c]), ISO[case + "_scale"], ISO[case + "_res"])
        assert got.dtype == want.dtype
        np.testing.assert_array_equal(got, want[..., c])


def test_stock_anisotropic_lightsheet_geometry_matches_oracle(gpu, env, tmp_path, monkeypatch):
    """The geometry real light-sheet data gives the default profile: 6.6 x 1.1 x 1.1 um voxels (the
    reference's own test resolution) -> 4 x 23 x 23-voxel denoise tiles (256-lane small-tile kernel),
    sigma 2.7..4.5 (kernel radii 11..18, odd ones included), overlap (1, 5, 5); whole stack vs the oracle."""
    from magellanmapper_amd import config, stack_detect, synth
    from oracle import magmap_oracle as mmo
    monkeypatch.chdir(tmp_path)
    vol = synth.make_volume(77, (22, 150, 160), 60, blob_sigma=2.6)
    config.setup_roi_profiles(None)
    config.roi_profile.update(num_sigma=5, segment_size=70)          # denoise_size 25 as shipped
    config.resolutions = np.array([[6.6, 1.1, 1.1]])
    config.near_max = [-1.0]
    config.filename = "aniso"
    blocks = stack_detect.setup_blocks(config.roi_profile, vol.shape)
    assert list(blocks.denoise_max_shape) == [4, 23, 23] and list(blocks.overlap) == [1, 5, 5]
    _, _, blobs = stack_detect.detect_blobs_blocks("aniso", stack_detect.Image5d(vol[None]), None, None,
                                                   None, False, False, True, False)
    want, st = mmo.detect_blobs_blocks(vol, None, [dict(config.roi_profile)], config.resolutions)
    assert st["seg_rois"].size >= 8 and len(want) > 20
    np.testing.assert_array_equal(blobs.blobs, want)


def test_isotropic_unit_thick_blocks_use_edge_mode(gpu):
    """A block with an axis of length 1 (the remainder block of a stack, a one-plane ROI) is resized in
    scikit-image's 'edge' mode by the reference (cv_nd.py:1096-1101) = SciPy's 'nearest': the coordinate
    is left alone and the sample indices are clamped.  Against the oracle's SciPy call (no fixture from the
    real reference exists for this shape: scikit-image 0.18.3 takes its 2-D warp there)."""
    from magellanmapper_amd import preprocess
    from oracle import isotropic_oracle
    rng = np.random.default_rng(12)
    for shape, scale, res in (((1, 30, 33), (0.96, 1, 1), (3.0, 1.0, 1.0)),
                              ((1, 17, 40), (1, 1, 1), (2.5, 1.2, 1.0)),
                              ((12, 1, 28), (1, 1, 1), (2.0, 1.0, 1.5)),
                              ((1, 1, 9), (1, 1, 1), (2.0, 3.0, 1.0))):
        for dtype in (np.uint16, np.float64):
            roi = (rng.integers(0, 65535, shape).astype(np.uint16) if dtype == np.uint16
                   else rng.random(shape) * 3 - 0.5)
            got = preprocess.make_isotropic(roi, scale, np.array(res))
            want = isotropic_oracle.make_isotropic(roi, scale, np.array(res))
            assert got.shape == want.shape and got.dtype == want.dtype
            np.testing.assert_array_equal(got, want, err_msg=str((shape, scale, res, dtype)))


@pytest.mark.parametrize("denoise", [None, 20])
def test_unmixing_of_isotropically_rescaled_blocks(gpu, env, denoise):
    """The reference resizes the whole multichannel block first (detector.py:893-897) and unmixes the resized
    channels (:910-921): rescale + unmixing (+ preprocessing) together, against the oracle."""
    from magellanmapper_amd import config, detector, synth
    from oracle import magmap_oracle as mmo
    rng = np.random.default_rng(5)
    shape = (22, 60, 57)
    roi = np.stack([synth.make_volume(int(rng.integers(1 << 30)), shape, 40, blob_sigma=2.2) for _ in range(3)], axis=-1)
    unmix = {1: {0: 0.3, 2: 0.1}, 2: {0: 0.5}}
    config.setup_roi_profiles(["default"] * 3)
    for p in config.roi_profiles:
        p.update(isotropic=(0.96, 1, 1), denoise_size=None, num_sigma=3, detection_threshold=0.1)
        p.spectral_unmixing = unmix
    config.resolutions = np.array([[2.5, 1.0, 1.0]])
    config.near_max = [-1.0] * 3
    profs = [dict(p, spectral_unmixing=unmix) for p in config.roi_profiles]
    if denoise is None:
        want = mmo.detect_blobs(roi, None, profs, config.resolutions)
        got = detector.detect_blobs(roi, None)
    else:
        from magellanmapper_amd import blob_log as bl
        dms = (int(np.ceil(denoise / 2.5)), denoise, denoise)
        want = mmo.detect_sub_roi((0, 0, 0), (0, 0, 0), (0, 0, 0), None, roi, None, profs, config.resolutions,
                                  denoise_max_shape=dms, near_max=config.near_max)
        got = detector.detect_blobs_blocks_device(bl.DeviceVolume(roi), None, [(0, 0, 0)], [shape],
                                                  denoise_max_shape=dms)[0]
    for p in config.roi_profiles:
        p.spectral_unmixing = None
    assert want is not None and len(want) > 20
    np.testing.assert_array_equal(got, want)


def test_isotropic_rescale_of_float32_images(gpu):
    """A float32 image is interpolated in double and stored into a float32 array (SciPy), clipped to its own
    range, and detected in float32 like the reference's float32 cube: rescale and detection vs the oracle."""
    from magellanmapper_amd import config, detector, preprocess, synth
    from oracle import isotropic_oracle, magmap_oracle as mmo
    rng = np.random.default_rng(8)
    roi = (synth.make_volume(5, (18, 50, 47), 30, blob_sigma=2.0) / 65535.0).astype(np.float32)
    for scale, res in (((0.96, 1, 1), (3.0, 1.0, 1.0)), ((1, 1, 1), (2.0, 1.5, 1.0))):
        got = preprocess.make_isotropic(roi, scale, np.array(res))
        want = isotropic_oracle.make_isotropic(roi, scale, np.array(res))
        assert got.dtype == np.float32 and got.shape == want.shape
        np.testing.assert_array_equal(got, want)
    config.setup_roi_profiles(None)
    config.roi_profile.update(isotropic=(0.96, 1, 1), denoise_size=None, num_sigma=3)
    config.resolutions = np.array([[2.5, 1.0, 1.0]])
    want = mmo.detect_blobs(roi, None, [dict(config.roi_profile)], config.resolutions)
    got = detector.detect_blobs(roi, None)
    assert want is not None and len(want) > 10
    np.testing.assert_array_equal(got, want)
    roi2 = np.stack([roi, (roi * rng.random(roi.shape)).astype(np.float32)], axis=-1)
    want = mmo.detect_blobs(roi2, None, [dict(config.roi_profile)], config.resolutions)
    np.testing.assert_array_equal(detector.detect_blobs(roi2, None), want)


def test_isotropic_anti_aliased_down_sampling(gpu):
    """Shrinking axes: scikit-image (>= 0.19) smooths with a Gaussian of sigma (factor - 1) / 2 before the zoom
    (``anti_aliasing`` default).  Device path vs the oracle's SciPy calls, bit for bit: uint16 (truncated
    back), float64, multichannel (clip range over all channels), mixed up / down axes, strong and mild factors."""
    from magellanmapper_amd import preprocess
    from oracle import isotropic_oracle
    rng = np.random.default_rng(21)
    cases = [((24, 40, 37), (1, 1, 1), (1.0, 2.0, 2.0)),          # z halves relative to ... (factor 0.5 on z? no: y,x grow)
             ((24, 40, 37), (0.5, 1, 1), (1.0, 1.0, 1.0)),        # z shrinks by 2
             ((30, 33, 41), (0.7, 0.6, 1.3), (1.0, 1.0, 1.0)),    # two axes shrink, one grows
             ((16, 50, 20), (1, 0.3, 1), (1.0, 1.0, 1.0)),        # strong: sigma 1.17, radius 5
             ((21, 22, 23), (0.9, 0.95, 0.85), (1.0, 1.0, 1.0))]  # mild: radius 0 or 1
    for shape, scale, res in cases:
        for kind in ("u16", "f64", "u16x2", "f32"):
            if kind == "u16":
                roi = rng.integers(0, 65535, shape).astype(np.uint16)
            elif kind == "f32":
                roi = (rng.random(shape) * 3 - 0.5).astype(np.float32)
            elif kind == "f64":
                roi = rng.random(shape) * 3 - 0.5
            else:
                roi = rng.integers(0, 40000, shape + (2,)).astype(np.uint16)
            got = preprocess.make_isotropic(roi, scale, np.array(res))
            want = isotropic_oracle.make_isotropic(roi, scale, np.array(res))
            assert got.shape == want.shape and got.dtype == want.dtype
            np.testing.assert_array_equal(got, want, err_msg=str((shape, scale, res, kind)))


def test_anti_aliasing_of_a_batch_mixing_unit_thick_and_regular_blocks(gpu):
    """One batch with a regular block and a one-plane remainder block, both down-sampled (anti-aliased) along y / x:
    the reference resizes the former in 'reflect' mode and the latter in 'edge' mode (cv_nd.py:1095-1101); the
    anti-aliasing pass takes the mode per block.  Each block against the oracle's SciPy calls, bit for bit."""
    from magellanmapper_amd import blob_log as bl, preprocess
    from oracle import isotropic_oracle
    rng = np.random.default_rng(23)
    for dtype in (np.uint16, np.float64):
        vol = (rng.integers(0, 65535, (14, 40, 44)).astype(np.uint16) if dtype == np.uint16
               else rng.random((14, 40, 44)) * 3 - 0.5)
        scale, res = (1, 0.6, 0.7), np.array((1.0, 1.0, 1.0))
        factor = preprocess.calc_isotropic_factor(scale, res)
        origins = [(0, 0, 0), (13, 0, 0), (2, 5, 4), (12, 3, 0)]
        shapes = [(13, 40, 44), (1, 40, 44), (9, 30, 33), (1, 5, 40)]
        new = [preprocess.isotropic_shape(s_, factor) for s_ in shapes]
        dvol = bl.DeviceVolume(vol)
        rs = preprocess.Rescaler(factor, [0], None)
        rs.set_blocks(origins, shapes, new)
        rs.run(dvol, 0, origins, new, 0)
        for o, s_, got in zip(origins, shapes, rs.fetch(new)):
            sub = vol[o[0]:o[0] + s_[0], o[1]:o[1] + s_[1], o[2]:o[2] + s_[2]]
            want = isotropic_oracle.make_isotropic(sub, scale, res)
            assert got.shape == want.shape and got.dtype == want.dtype, (o, s_)
            np.testing.assert_array_equal(got, want, err_msg=str((o, s_, dtype)))


# ------------------------------------------------------------------------------- total-variation denoising
TV = load_golden("tv.npz")


@pytest.mark.parametrize("case", [str(n) for n in TV["names"]])
def test_tile_with_tv_denoising_matches_reference(gpu, env, case):
    """``tot_var_denoise`` on (settings of the stock profiles 'minpreproc' and '2p20x', other weights, ragged /
    uint8 / constant / tiny tiles): the device iteration (``mmx_preprocess_batch_generic``) stops at the
    reference's iteration and returns its float64 values bit for bit -- the real ``plot_3d.denoise_roi`` with the
    real ``skimage.restoration.denoise_tv_chambolle``."""
    from magellanmapper_amd import preprocess
    roi = TV[case + "_roi"]
    over = ast.literal_eval(str(TV[case + "_over"]))
    _set_profiles(over, 1)
    got = preprocess.preprocess_roi(roi, roi.shape[:3], near_max=list(TV[case + "_near_max"]))
    assert got.dtype == np.float64
    np.testing.assert_array_equal(got, TV[case + "_den"])


def test_tv_denoised_block_of_several_tiles_matches_oracle(gpu, env):
    """Tiles of different sizes in one batch (25-voxel tiles over a 40 x 45 x 52 block), weight 0.05 + unsharp."""
    from magellanmapper_amd import preprocess
    from oracle import preprocess_oracle as ppo
    roi = load_golden("stack_denoise.npz")["roi"][:40, :45, :52]
    profs = _set_profiles(dict(tot_var_denoise=0.05), 1)
    got = preprocess.preprocess_roi(roi, (25, 25, 25))
    want = ppo.preprocess_block(roi, (25, 25, 25), profs, [-1.0])
    np.testing.assert_array_equal(got, want)
