"""Device ``blob_log``: Laplacian-of-Gaussian scale-space blob detection on MI355X.

Drop-in for the one third-party call on the reference's hot path,
``skimage.feature.blob_log`` as ``magmap.cv.detector.detect_blobs`` uses it
(reference magmap/cv/detector.py:931-933), for batches of blocks of one volume:

=====  ==========================================  ================================
row    reference step                              here
=====  ==========================================  ================================
A0     ``img_as_float`` (skimage dtype.py:310-328)  folded into the Z-pass weights
A1     sigma ladder (skimage blob.py:473-497)       :func:`kernels1d.sigma_ladder`
A2,A3  ``-gaussian_laplace * sigma**2`` per scale   ``mmx_log_batch_f32`` (HIP)
A4     ``peak_local_max`` 3^4 NMS (peak.py)         ``mmx_peaks_batch`` (HIP) +
                                                    ``mmx_rescore_f64`` (HIP, exact)
                                                    + tie resolution below
A5     ``_prune_blobs`` (blob.py:146-187)           ``mmx_overlap_pairs`` (HIP) +
                                                    the sequential rule below
=====  ==========================================  ================================

Exactness.  The float32 passes only *nominate* candidates (within ``eps`` of being a
maximum / of the threshold).  Wherever a decision of the reference could depend on more
than float32 resolves -- a candidate within ``eps`` of the threshold or of a neighbour, two
candidates of a block within ``eps`` of each other (their order) -- the float64 cube values
are recomputed on the device with SciPy's exact operation order (contested candidates also
get their 80 neighbours' exact values); by default every candidate is re-scored.  Peak
membership, the descending-response order and therefore the integer blob coordinates are
decided as the reference's float64 values decide them.  Nothing here falls back to a CPU implementation: without the HIP library and a
GPU the calls raise.
"""
from __future__ import annotations

import ctypes
import os
import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _native as nat
from . import kernels1d as k1

try:  # torch is the device-memory / stream plumbing
    import torch
except Exception as exc:  # pragma: no cover
    raise ImportError("magellanmapper_amd needs PyTorch-ROCm for device memory") from exc
#: half-width of the float32 "contested" band, relative to the input's value scale
EPS_REL = float(os.environ.get("MMX_EPS_REL", 2e-5))
#: the band for raw integer volumes, whose default kernels hand the Z+X results to the Y pass as 16-bit fixed point
#: (``MMX_ZX_TILED_Q16``: error <= 3.0e-5 of the value range from radius 4 on, ``mmx_tiled_q16_error_bound``); 0 keeps
#: float32 intermediates
EPS_REL_Q16 = float(os.environ.get("MMX_EPS_REL_Q16", 2.5e-4))
#: the rounding error of the 16-bit intermediates, relative to the value range, for kernel radii >= 4 (sigma >= 0.875;
#: largest at radius 5: 2.97e-5 -- swept over sigma in tests/test_host_logic.py; rounds 3-5 carried 5.3e-5: Q was
#: quantised over twice the range it can take, see ``q16_bounds`` in csrc/mmx_api.hip).  Per call the library states the
#: bound of the sigmas at hand (``mmx_tiled_q16_error_bound``); the kernels of radius 1..3 carry up to 5.4e-5.
Q16_BOUND_ANY_SIGMA = 3.0e-5
#: ``MMX_LOG_ABS_TOL`` of include/mmx.h: the LoG contract in value units (BASELINE.json: "LoG response within 1e-4")
LOG_ABS_TOL = 1e-4
if 0.0 < EPS_REL_Q16 < 4.0 * Q16_BOUND_ANY_SIGMA:
    # exactness rests on the band covering the error fourfold: an environment variable may widen it or switch the
    # 16-bit intermediates off (0), not narrow it below what the bound needs
    raise ValueError(f"MMX_EPS_REL_Q16={EPS_REL_Q16:g} is narrower than 4 x the 16-bit intermediates' error bound "
                     f"({4.0 * Q16_BOUND_ANY_SIGMA:g}); use 0 to keep float32 intermediates")
#: raw volumes on the native host path: one ``mmx_detect_batch`` call enqueues a whole batch (voxel copy, the passes of
#: every scale, NMS, probes, exact re-score, copies) instead of a dozen calls from here (``False``: the call-by-call
#: form, which tests keep as a cross-check)
NATIVE_BATCH = True
#: batches of at most this many blocks that come back with the very same arguments (a small volume detected step
#: after step) are captured as a hipGraph and replayed with one launch (0: never)
GRAPH_BLOCKS = 8
#: with the per-kernel timing on: (before, after) event pairs around the LoG stream's wait for a batch's preprocessing
PRE_WAITS: list = []
#: batches replayed from a captured graph so far (bench.py reports the count of its timed region)
GRAPH_REPLAYS = 0
#: bound of the 16-bit intermediates of the most recent batch that used them (value units), else 0: bench.py prints it
LAST_Q16_BOUND = 0.0
#: nomination band (value units) of that batch
LAST_NMS_BAND = 0.0
#: host clock (time.perf_counter) at which the most recent batch's last kernel was seen complete
LAST_BATCH_DONE_T = 0.0
#: host clock at which the first batch of the most recent ``blob_log_blocks`` call had been handed to the GPU (the
#: "start" of a step: block lists, plans, tables -> first launch), and the batch sizes of that call (bench.py --share)
FIRST_ENQUEUED_T = 0.0
LAST_BATCH_SIZES: list = []
#: ``mmx_zx_mode`` passed with every ``mmx_log_batch_f32`` call (``MMX_FUSE`` in the environment overrides the
#: default for kernel experiments; tests set it to cross-check the kernels against each other)
ZX_MODE = int(os.environ.get("MMX_FUSE", nat.MMX_ZX_AUTO))
#: flags or-ed into the mode of the tiled calls: ``nat.MMX_ZX_Y_VALU`` runs the Y pass of the 16-bit tiles on the VALU
#: (``y6_kernel``) instead of the matrix cores (``ym_kernel``) -- cross-checks and A/B timing
ZX_FLAGS = 0
#: the ``mmx_zx_mode`` the most recent ``mmx_log_batch_f32`` call of this process actually ran
LAST_ZX_PATH = None
#: who takes the per-batch decisions on the re-scored candidates: "native" (``mmx_host_resolve_peaks`` /
#: ``mmx_host_overlap_prune``: threaded, outside the GIL, no second device round trip) or "numpy" (the same rules as
#: array expressions; kept as a cross-check -- tests run both -- and for ``exact_values=False``)
HOST_PATH = os.environ.get("MMX_HOST_PATH", "native")
#: value scales (largest |voxel|, at least 1) up to which float voxels take the tiled matrix-core path: its float16
#: pieces overflow at 65504 (faint images are no problem: the nomination band never shrinks below EPS_REL x 1)
FLOAT_TILED_RANGE = (0.0, 2.0 ** 12)
#: per-block preprocessing on a stream of its own (beside the previous batch's LoG kernels)
PRE_STREAM = True
#: ... with the NMS / re-score tail of a preprocessed batch on the second stream too.  Off:
#: measured on the box it LOSES -- 283.6 against 278.9 ms (two-channel tile), 214.6 against 212.4 ms (--denoise 25): the
#: preprocessing kernel already fills every CU it can, a third stream only adds to the time-sharing
PRE_SIDE_TAIL = False
#: ... and this many batches ahead of the one the host is finishing (``preprocess.N_BUFFER_SETS`` - 1 buffer sets allow
#: it): the first batch is then the only one whose LoG passes wait for their preprocessing
PRE_AHEAD = 2
#: ... and this many BATCHES (x the number of lanes) when the preprocessed blocks are retained per block
PRE_AHEAD_RETAINED = 2
#: raw volumes: everything after a batch's last LoG kernel -- NMS, probe expansion, exact re-score, the copies of the
#: results to the host -- on a stream of its own, beside the LoG kernels of the next batch, which then work in a second
#: workspace (``_Buffers.workspace(n, 1)``: +16 GiB with the default budget).  Measured on the benchmark volume,
#: alternating runs on one box: 115.2 / 113.5 ms per volume without, 110.2 / 109.4 with (the NMS takes 8.3 instead of
#: 4.5 ms and the Z+X kernel 59 instead of 53 when they run beside each other; the re-score alone on the side stream,
#: one workspace: -2.8 ms)
RESCORE_STREAM = True
PACK_STREAM = False

from .volume import (DeviceVolume, _SlabUpload, _img_as_float_host, _require_gpu, _stream_ptr,   # noqa: F401,E402
                     _NP_TO_MMX, _TORCH_DTYPES)
from .buffers import (_Buffers, _NativeEvent, _UploadRing, _buffers_for, _stream_wait, _to_device_bytes,   # noqa: F401,E402
                      release_buffers, to_device, _PREFIX_ENTRIES)
from .host_resolve import (OVERLAP_BAND, PeakBatch, VERIFIED_SCIPY, _BandTooNarrow, _apply_pairs,   # noqa: F401,E402
                           _check_f32_error, _exact_overlap, _prune_batch, _prune_batch_native,
                           _reference_pair_order, _resolve_peaks, _resolve_peaks_native)


@dataclass
class ScaleSpace:
    """Per-call filter parameters (never cached across calls: the reference re-reads its
    profile on every call and the grid search mutates it)."""
    sigmas: np.ndarray
    norms: np.ndarray
    radii: np.ndarray
    w0: List[np.ndarray]
    w2: List[np.ndarray]
    w0_tab: np.ndarray = field(default=None)
    w2_tab: np.ndarray = field(default=None)
    _dev: Optional[dict] = field(default=None, repr=False)
    _q16: Optional[float] = field(default=None, repr=False)

    @classmethod
    def make(cls, min_sigma: float, max_sigma: float, num_sigma: int) -> "ScaleSpace":
        """The filter parameters of ``blob_log(min_sigma, max_sigma, num_sigma)``: a pure function of the three
        numbers, so the last few are kept (the PROFILE is re-read at every call; what its numbers imply is not)."""
        if np.isscalar(min_sigma) and np.isscalar(max_sigma):
            key = (float(min_sigma), float(max_sigma), int(num_sigma))
            hit = _SPACES.get(key)
            if hit is None:
                if len(_SPACES) > 64:
                    _SPACES.clear()
                hit = _SPACES[key] = cls._make(min_sigma, max_sigma, num_sigma)
            return hit
        return cls._make(min_sigma, max_sigma, num_sigma)

    def device_tables(self, dev):
        """``(w0, w2)`` half-kernel tables on ``dev`` (for the exact re-score), uploaded once per device."""
        key = str(dev)
        if self._dev is None:
            self._dev = {}
        if key not in self._dev:
            self._dev[key] = (torch.from_numpy(self.w0_tab).to(dev), torch.from_numpy(self.w2_tab).to(dev))
        return self._dev[key]

    def q16_bound(self) -> float:
        """``mmx_tiled_q16_error_bound`` of the worst scale, relative to the value range (cached: the scale space is)."""
        if self._q16 is None:
            L = nat.lib()
            b = [float(L.mmx_tiled_q16_error_bound(nat.as_double_ptr(self.w0[s]), nat.as_double_ptr(self.w2[s]),
                                                   int(self.radii[s]), float(self.norms[s]))) for s in range(len(self.sigmas))]
            self._q16 = float("inf") if (not b or min(b) < 0) else max(b)
        return self._q16

    @classmethod
    def _make(cls, min_sigma: float, max_sigma: float, num_sigma: int) -> "ScaleSpace":
        if not (np.isscalar(min_sigma) and np.isscalar(max_sigma)):
            raise NotImplementedError("per-axis sigmas are not on the reference's path "
                                      "(detector.py:926-927 passes scalars)")
        sigmas, norms = k1.sigma_ladder(float(min_sigma), float(max_sigma), int(num_sigma))
        radii = np.array([k1.kernel_radius(s) for s in sigmas], dtype=np.int32)
        if radii.max(initial=0) > nat.MMX_MAX_RADIUS_GENERIC:
            raise nat.MmxError("sigma too large for the device kernels (radius > 255)")
        w0, w2 = [], []
        tab0 = np.zeros((len(sigmas), nat.MMX_MAX_RADIUS_GENERIC + 1))
        tab2 = np.zeros_like(tab0)
        for i, (s, r) in enumerate(zip(sigmas, radii)):
            if s <= 1e-15:
                raise NotImplementedError("sigma = 0 is not supported on the device path")
            a = k1.gaussian_half_kernel(float(s), 0, int(r))
            b = k1.gaussian_half_kernel(float(s), 2, int(r))
            w0.append(a)
            w2.append(b)
            tab0[i, :r + 1] = a
            tab2[i, :r + 1] = b
        return cls(sigmas, norms, radii, w0, w2, tab0, tab2)


_SPACES: Dict[tuple, ScaleSpace] = {}


@dataclass
class BatchStats:
    """Counters of one :func:`blob_log_blocks` call (also feeds bench.py)."""
    n_blocks: int = 0
    n_voxels: int = 0
    n_candidates: int = 0
    n_contested: int = 0
    n_probes: int = 0
    n_rescored: int = 0         # candidates re-scored in float64 because a decision needed it
    n_peaks: int = 0
    n_blobs: int = 0
    n_overlap_pairs: int = 0
    n_order_fallbacks: int = 0
    n_band_retries: int = 0       # batches nominated again because float32 strayed too far from float64
    max_f32_error: float = 0.0


#: debugging aid: cap on the blocks of one batch (``MMX_MAX_BATCH``; bisecting a batch-size dependent failure)
_MAX_BATCH = 1 << 30
#: the tail of the block list is re-split into batches that shrink by this factor (nothing hides the host work of the
#: last batches), the first batch into a ramp that starts at this many blocks (nothing hides its GPU time from the host)
TAPER = 4
RAMP = 16
#: experiments only: cut the block list into batches of exactly these sizes (when they add up to the number of blocks)
FORCED_BATCH_SIZES: list = []


def plan_batches(shapes: Sequence[Tuple[int, int, int]], num_sigma: int,
                 budget_bytes: int, extra_bytes_per_voxel: int = 0) -> List[List[int]]:
    """Group block indices into batches whose workspace fits ``budget_bytes``.

    Workspace = (4 + num_sigma) float32 arrays of ``n_blocks * slot`` voxels, slot = the
    largest block of the batch (+ ``extra_bytes_per_voxel`` for the preprocessed copies).
    Blocks keep their order (z-major grid order).
    """
    per_vox = (4 + num_sigma) * 4 + (num_sigma + 1) // 2 + extra_bytes_per_voxel     # + the NMS bit masks
    if not len(shapes):
        return []
    if FORCED_BATCH_SIZES and sum(FORCED_BATCH_SIZES) == len(shapes):       # (experiments: bench.py --batches)
        cuts = np.concatenate(([0], np.cumsum(FORCED_BATCH_SIZES)))
        return [list(range(int(a), int(b))) for a, b in zip(cuts[:-1], cuts[1:])]
    batches: List[List[int]] = []
    cur: List[int] = []
    cur_slot = 0
    for i, shp in enumerate(shapes):
        vox = int(shp[0]) * int(shp[1]) * (-(-int(shp[2]) // nat.MMX_ROW_ALIGN) * nat.MMX_ROW_ALIGN)
        slot = max(cur_slot, vox)
        if cur and ((len(cur) + 1) * slot * per_vox > budget_bytes or len(cur) >= _MAX_BATCH):
            batches.append(cur)
            cur, slot = [], vox
        cur.append(i)
        cur_slot = slot
    if cur:
        batches.append(cur)
    # The host finishes batch k (candidates -> peaks -> per-block prune -> tables: ~0.45 ms per block of the
    # benchmark volume) while the GPU runs batch k + 1 (~0.46 ms per block with the tiled kernels), so it trails the
    # GPU by the host time of one batch, and nothing hides that of the LAST batches: the tail of the block list is
    # re-split into batches that shrink geometrically towards the end (... 18 / 11 / 7 / 4 blocks) -- each still
    # gives the GPU about as much work as the host has left from the batch before.  Measured (tools/steptrace.py):
    # 89 / 89 / 59 / 19 blocks left the host 8 ms behind the GPU at the end and 18 ms of tail, 89 / 89 / 39 / 20 /
    # 10 / 5 / 4 did not; with 22-block batches a tail of 22 / 7 / 7 against 18 / 18 / 11 / 7 / 4.
    taper = TAPER
    vox = [int(s_[0]) * int(s_[1]) * int(s_[2]) for s_ in shapes]

    def fits(batch):
        slot = max(int(shapes[i][0]) * int(shapes[i][1]) * (-(-int(shapes[i][2]) // nat.MMX_ROW_ALIGN) * nat.MMX_ROW_ALIGN)
                   for i in batch)
        return len(batch) == 1 or (len(batch) * slot * per_vox <= budget_bytes and len(batch) <= _MAX_BATCH)

    full = max(len(b_) for b_ in batches)
    tail = batches.pop()
    while batches and len(tail) < 2 * full and _MAX_BATCH >= (1 << 30):      # (a capped batch size: no re-merging)
        tail = batches.pop() + tail
    # (only where there is real work: below ~64 Mvoxel a batch is a few hundred microseconds of kernels)
    if taper > 0 and len(tail) > taper and sum(vox[i] for i in tail) > (64 << 20):
        sizes, step = [], float(taper)
        while sum(sizes) + int(step) <= len(tail) and int(step) < full:
            sizes.append(int(step))
            step *= 1.6
        rest = len(tail) - sum(sizes)
        n_front = -(-rest // full) if rest else 0
        front = [rest // n_front + (1 if j < rest % n_front else 0) for j in range(n_front)] if n_front else []
        at, pieces = 0, []
        for n in front + sizes[::-1]:
            pieces.append(tail[at:at + n])
            at += n
        tail_batches = []
        for piece in pieces:                    # (blocks of different sizes: a piece may exceed the budget its count suggests)
            while not fits(piece):
                cut = len(piece) // 2
                tail_batches.append(piece[:cut])
                piece = piece[cut:]
            tail_batches.append(piece)
        batches.extend(tail_batches)
    else:
        stack = [tail]
        while stack:                             # (the merged tail of small batches must still fit the budget)
            piece = stack.pop(0)
            if fits(piece):
                batches.append(piece)
            else:
                stack[0:0] = [piece[:len(piece) // 2], piece[len(piece) // 2:]]
    # ... and nothing hides the GPU time of the FIRST batch from the host, which has nothing to do until its
    # candidates arrive: with the tiled kernels the host work per block (~0.5 ms) is as long as the kernels'
    # (~0.55 ms), so a first batch of 89 blocks put the host 50 ms behind for the whole step (tail after the last
    # kernel 24 - 38 ms).  The first batch is therefore split into a ramp of `MMX_RAMP`, 2 x, 4 x ... blocks.
    ramp = RAMP
    first = batches[0]
    if ramp > 0 and len(batches) > 1 and len(first) > 2 * ramp and sum(vox[i] for i in first) > (64 << 20):
        head = []
        while len(first) > 2 * ramp:
            head.append(first[:ramp])
            first = first[ramp:]
            ramp *= 2
        batches[0:1] = head + [first]
    return batches


_SELF_TESTED = set()


def self_test(dev) -> None:
    """Known-answer check of the matrix-core kernels, once per process and device (~30 ms): a small block tall
    enough for their steady-state loop through ``MMX_ZX_TILED`` and ``MMX_ZX_TILED_Q16`` (integer and float voxels,
    one radius per kernel geometry class) against the library's generic one-thread-per-voxel kernels.

    Why it exists: the exactness machinery compares float32 with float64 values AT THE CANDIDATES a batch
    nominates -- a kernel that is wrong everywhere nominates nothing and the batch comes back empty without an
    error.  That happened during development (an inline-assembly instruction behind an MFMA that still read its
    destination register: zeros for radius <= 8); the parity tests caught it, and this makes a production process
    fail as loudly should a compiler or driver ever produce it again."""
    key = str(dev)
    if key in _SELF_TESTED:
        return
    L = nat.lib()
    rng = np.random.default_rng(1234)
    vol = rng.integers(0, 65536, (70, 12, 80)).astype(np.uint16)
    vol[20:50, 3:9, 30:60] //= 16
    shape = vol.shape
    for dtype in (np.uint16, np.float32):
        dv = DeviceVolume(vol if dtype == np.uint16 else (vol.astype(np.float32) / np.float32(65535.0)), dev)
        blocks, slot = _make_blocks(dv, 0, [(0, 0, 0)], [shape])
        d_blocks = _to_device_bytes(blocks, dev)
        v32 = dv.view(0, True)
        v32.value_range = 1.0 if dtype == np.float32 else 0.0
        ws = torch.zeros(6 * slot, dtype=torch.float32, device=dev)
        for sigma in (1.6, 3.2, 4.6):                      # radii 6, 13, 18: the three kernel geometry classes
            space = ScaleSpace.make(sigma, sigma, 1)
            args = (ctypes.byref(v32), d_blocks.data_ptr(), blocks.ctypes.data, 1, slot, nat.as_double_ptr(space.w0[0]),
                    nat.as_double_ptr(space.w2[0]), int(space.radii[0]), float(space.norms[0]))
            nat.check(L.mmx_log_batch_f32_generic(*args, ws.data_ptr() + 5 * slot * 4, ws.data_ptr(), _stream_ptr()),
                      "mmx_log_batch_f32_generic")
            want = ws[5 * slot:6 * slot].clone()
            for mode, tol in ((nat.MMX_ZX_TILED, 2e-6), (nat.MMX_ZX_TILED_Q16, 6e-5)):
                path = ctypes.c_int(0)
                nat.check(L.mmx_log_batch_f32(*args, ws.data_ptr() + 4 * slot * 4, ws.data_ptr(), None, 0.0, 0.0, None,
                                              mode, ctypes.byref(path), _stream_ptr()), "mmx_log_batch_f32")
                got = ws[4 * slot:5 * slot]
                n = shape[0] * shape[1] * int(blocks["px"][0])
                g = got[:n].view(shape[0], shape[1], -1)[..., :shape[2]]
                w = want[:n].view(shape[0], shape[1], -1)[..., :shape[2]]
                err = float((g - w).abs().max().item())
                if path.value == mode and not err < tol:
                    raise nat.MmxError(f"self-test of the device kernels failed (zx_mode {mode}, radius "
                                       f"{int(space.radii[0])}, {np.dtype(dtype).name} voxels: off by {err:.3g}); "
                                       "this build of libmmx_hip.so must not be used")
    _SELF_TESTED.add(key)


def log_cube_blocks(dvol: DeviceVolume, channel: int, origins, shapes, space: ScaleSpace,
                    generic: bool = False, value_range: float = 0.0) -> List[np.ndarray]:
    """Float32 ``(z, y, x, sigma)`` cubes of the given blocks (A0-A3 only; used by tests
    for the 1e-4 LoG tolerance and by the profiling scripts).  ``value_range``: what the caller knows about float
    voxels (``mmx_volume.value_range``: > 0 = values in [0, m], < 0 = |v| <= m, 0 = unknown), which decides whether
    they may take the tiled matrix-core kernels."""
    dvol.wait_all()
    dev = dvol.tensor.device
    L = nat.lib()
    blocks, slot = _make_blocks(dvol, channel, origins, shapes)
    nb, ns = len(blocks), len(space.sigmas)
    ws = torch.empty((4 + ns) * nb * slot, dtype=torch.float32, device=dev)
    d_blocks = _to_device_bytes(blocks, dev)
    vol32 = dvol.view(channel, True)
    if vol32.dtype == nat.MMX_F32:
        vol32.value_range = float(value_range)
    fn = L.mmx_log_batch_f32_generic if generic else L.mmx_log_batch_f32
    log_base = ws.data_ptr() + 4 * nb * slot * 4
    global LAST_ZX_PATH
    path = ctypes.c_int(0)
    for s in range(ns):
        nat.check(fn(ctypes.byref(vol32), d_blocks.data_ptr(), blocks.ctypes.data, nb, slot,
                     nat.as_double_ptr(space.w0[s]), nat.as_double_ptr(space.w2[s]),
                     int(space.radii[s]), float(space.norms[s]),
                     log_base + s * nb * slot * 4, ws.data_ptr(),
                     *(() if generic else (None, 0.0, 0.0, None, ZX_MODE | ZX_FLAGS if ZX_MODE >= 0 else ZX_MODE, ctypes.byref(path))),
                     _stream_ptr()), "mmx_log_batch_f32")
        LAST_ZX_PATH = None if generic else path.value
    torch.cuda.synchronize()
    logs = ws[4 * nb * slot:].view(ns, nb, slot).cpu().numpy()
    out = []
    for i, shp in enumerate(shapes):
        px = int(blocks["px"][i])
        n = int(shp[0]) * int(shp[1]) * px
        cube = logs[:, i, :n].reshape((ns, shp[0], shp[1], px))[..., :shp[2]]
        out.append(np.moveaxis(cube, 0, -1).copy())
    return out


def _make_blocks(dvol: DeviceVolume, channel: int, origins, shapes):
    """``mmx_block`` records of one batch (block i owns slot i) and the workspace slot size they need."""
    t = dvol.tensor
    strides = np.array([t.stride()[0], t.stride()[1], t.stride()[2]], dtype=np.int64)
    blocks = np.zeros(len(shapes), dtype=nat.BLOCK_DTYPE)
    if not len(shapes):
        return blocks, 1
    o = np.asarray(origins, dtype=np.int64).reshape(-1, 3)
    shp = np.asarray(shapes, dtype=np.int64).reshape(-1, 3)
    if (o < 0).any() or (shp < 1).any() or (o + shp > np.asarray(dvol.shape[:3], dtype=np.int64)).any():
        raise ValueError("block outside the volume")
    px = -(-shp[:, 2] // nat.MMX_ROW_ALIGN) * nat.MMX_ROW_ALIGN      # 128-B aligned rows
    blocks["src_off"] = o @ strides
    blocks["nz"], blocks["ny"], blocks["nx"] = shp[:, 0], shp[:, 1], shp[:, 2]
    blocks["slot"] = np.arange(len(shp))
    blocks["px"] = px
    return blocks, max(1, int((shp[:, 0] * shp[:, 1] * px).max()))


def blob_log_blocks(dvol: DeviceVolume, channel: int, origins: Sequence[Sequence[int]],
                    shapes: Sequence[Sequence[int]], min_sigma: float, max_sigma: float,
                    num_sigma: int, threshold: float, overlap: float, *,
                    budget_bytes: Optional[int] = None, stats: Optional[BatchStats] = None,
                    return_peaks: bool = False, on_batch=None, pre=None,
                    exact_values: Optional[bool] = None, sink=None, finisher=None, plan_num_sigma: Optional[int] = None):
    """``blob_log`` of every block -> list of ``(n, 4)`` float64 ``[z, y, x, sigma]`` arrays.

    ``exact_values`` (default True): re-score EVERY candidate in float64 in the batch's own kernel queue, so
    that the peak values are the reference's bit for bit.  ``False`` re-scores only the candidates whose
    decision needs it (see ``_resolve_peaks``) -- fewer points, but as a second, host-synchronous launch
    per batch: on the benchmark volume, whose blobs are all alike, 68 % of the candidates have a block mate
    within eps and the step is slower (243 vs 227 ms); it pays on images with a wide intensity range.

    Each block is an independent image exactly as each reference worker's sub-ROI is
    (reference magmap/cv/stack_detect.py:79): reflect boundaries at the block faces,
    coordinates relative to the block.  Blocks without blobs give ``np.empty((0, 3))`` like
    scikit-image does (blob.py:516-517).  Row order equals the reference's.
    ``on_batch(indices, results)`` is called as each batch finishes, while the GPU is busy
    with the next one.  ``pre`` (a ``preprocess.Preprocessor``) saturates + denoises every block
    on the device first (reference stack_detect.py:122-150); detection then runs on the
    float64 result exactly as the reference's ``blob_log`` does.
    ``sink(indices, peak_batch)`` (native host path only) takes each batch as a :class:`PeakBatch` -- rows, ``alive``
    flags, block offsets -- instead of per-block arrays; the call then returns ``None`` entries for those blocks.
    ``finisher(indices, cands, n_cands, blocks, space, thr, eps, overlap, stats) -> bool`` (native host path, only when
    ALL blocks are one batch): takes the batch's re-scored candidate table and does everything behind it itself
    (``stack_detect._StackFinisher``: one native call up to the stack's final table); ``False``: it left the batch to
    the usual steps.
    """
    lane = Lane(channel, min_sigma, max_sigma, num_sigma, threshold, overlap, stats=stats, on_batch=on_batch, pre=pre,
                exact_values=exact_values, sink=sink, finisher=finisher)
    out = blob_log_lanes(dvol, [lane], origins, shapes, budget_bytes=budget_bytes, return_peaks=return_peaks,
                         plan_num_sigma=plan_num_sigma)
    return (out[0][0], out[1][0]) if return_peaks else out[0]


class Lane:
    """One detection over the blocks of a volume -- a channel with its scales, threshold, preprocessing and the callbacks
    that take its batches (the arguments of :func:`blob_log_blocks`); :func:`blob_log_lanes` runs several of them over the
    same blocks in ONE pipeline."""

    def __init__(self, channel: int, min_sigma: float, max_sigma: float, num_sigma: int, threshold: float, overlap: float,
                 *, stats: Optional[BatchStats] = None, on_batch=None, pre=None, exact_values: Optional[bool] = None,
                 sink=None, finisher=None):
        self.channel, self.threshold, self.overlap = channel, float(threshold), float(overlap)
        from . import host_resolve as _hr
        if _hr.PEAK_ORDER == "0.19+" and self.threshold < 0:
            raise NotImplementedError("PEAK_ORDER '0.19+' pads the peak mask with 'nearest': identical to the zero padding "
                                      "of this path for thresholds >= 0 only")
        self.space = ScaleSpace.make(min_sigma, max_sigma, num_sigma)
        self.stats = stats if stats is not None else BatchStats()
        self.on_batch, self.pre, self.sink, self.finisher = on_batch, pre, sink, finisher
        self.exact = bool(True if exact_values is None else exact_values)
        self.results: list = []
        self.peaks_out: list = []

    def bind(self, dvol: DeviceVolume, n_blocks: int) -> None:
        """What follows from the volume: value scale, nomination band, value range, the weight tables on its device."""
        space, pre, channel = self.space, self.pre, self.channel
        self.results = [None] * n_blocks
        self.peaks_out = [None] * n_blocks
        self.vscale = float(dvol.value_scale() if pre is None else pre.value_scale([channel]))
        eps = EPS_REL * self.vscale
        # what is known about the voxels the passes will read: raw integer images [0, 1]; float images their measured
        # range; preprocessed / unmixed / rescaled blocks the bounds their arithmetic implies
        self.vrange = vrange = dvol.value_range(channel) if pre is None else pre.value_range([channel])
        integer_voxels = pre is None and dvol.np_dtype in (np.dtype(np.uint8), np.dtype(np.uint16))
        # volumes whose voxels are known to be non-negative and of ordinary magnitude take 16-bit intermediates: the
        # band must cover their rounding error fourfold
        # -- and their error in value units must stay inside the LoG contract (a profile whose unsharp / clip settings
        # stretch the preprocessed range past ~3.4 keeps float32 intermediates and the narrow band)
        if (vrange is not None and vrange[0] >= 0.0 and vrange[1] <= FLOAT_TILED_RANGE[1] and EPS_REL_Q16 > EPS_REL
                and ZX_MODE in (nat.MMX_ZX_AUTO, nat.MMX_ZX_TILED_Q16)
                and (ZX_MODE == nat.MMX_ZX_TILED_Q16 or space.q16_bound() * (vrange[1] if not integer_voxels else 1.0)
                     <= LOG_ABS_TOL)):
            eps = EPS_REL_Q16 * self.vscale
        self.eps = eps
        self.d_w0, self.d_w2 = space.device_tables(dvol.tensor.device)


#: workspace budget per batch of a call that does not say (``budget_bytes=None``); bench.py sets it from the free HBM
BUDGET_BYTES = 24 << 30
#: several channels of a stack in one pipeline, batch-major (both channels of a batch of blocks before the next batch):
#: "uploading" (default) = while the volume is still on its way to the device -- a host tile is then needed layer by
#: layer by all channels instead of whole by the first (one C5 tile from the host: 240 against 285 ms); True = always;
#: False = never: the channels one after the other, each through all its batches.  For a RESIDENT volume the two orders
#: cost the same on average (213-217 ms per C5 step) but batch-major scatters more (p95 / p50 1.08 against 1.02; as a
#: sub-record behind other workloads 232-254 against 223-225): profiles/r06_experiments.txt section 7
BATCH_MAJOR = "uploading"


def blob_log_lanes(dvol: DeviceVolume, lanes: Sequence[Lane], origins: Sequence[Sequence[int]],
                   shapes: Sequence[Sequence[int]], *, budget_bytes: Optional[int] = None, return_peaks: bool = False,
                   plan_num_sigma: Optional[int] = None):
    """:func:`blob_log_blocks` for several lanes (channels) over the SAME blocks in one pipeline, batch-major: the
    batches are planned once, and batch b goes through lane 0, lane 1, ... before batch b + 1 -- one queue of launches
    for the whole call (no drain between channels), a block's tables of every channel complete together (so that a stack's
    rows land, and its regions are pruned, from the first batches on), and a host tile still on its way up is needed
    layer by layer by ALL channels instead of whole by the first.  All lanes preprocess or none does.
    ``plan_num_sigma``: plan the batches as for this many scales (calls for the channels of one stack made one after
    the other cut the blocks into the SAME batches that way, whatever each channel's own number of scales).
    Returns ``[lane results]`` (and ``[lane peaks]`` with ``return_peaks``)."""
    _require_gpu()
    budget_bytes = int(BUDGET_BYTES if budget_bytes is None else budget_bytes)
    lanes = list(lanes)
    if not lanes:
        return ([], []) if return_peaks else []
    if len({lane.pre is None for lane in lanes}) > 1:
        raise ValueError("lanes of one pipeline either all preprocess their blocks or none does")
    any_pre = lanes[0].pre is not None
    bufs = _buffers_for(dvol.tensor.device)
    # The very same tuple objects step after step (stack_detect hands over its cached block lists): their checked copies
    # and their content key are remembered -- converting and hashing 256 blocks is 0.3 ms before the first launch.
    # (tuples only: a list could have been edited in place since)
    immutable = type(origins) is tuple and type(shapes) is tuple
    seen = bufs.plan_lists.get((id(origins), id(shapes))) if immutable else None
    if seen is not None and seen[0] is origins and seen[1] is shapes:
        _, _, lists_key, origins, shapes = seen
    else:
        given = (origins, shapes)
        shapes = [tuple(int(v) for v in s) for s in shapes]
        origins = [tuple(int(v) for v in o) for o in origins]
        lists_key = (tuple(origins), tuple(shapes))
        if immutable:
            if len(bufs.plan_lists) >= 8:
                bufs.plan_lists.clear()
            bufs.plan_lists[(id(given[0]), id(given[1]))] = (given[0], given[1], lists_key, origins, shapes)   # (kept alive: ids stay theirs)
    for lane in lanes:
        lane.bind(dvol, len(shapes))
    n_sig = [len(lane.space.sigmas) for lane in lanes]
    ns_max = max(n_sig + [int(plan_num_sigma or 0)])
    # the batches and their block tables on the device: remembered for the same block lists, volume layout and budget (a
    # stack detected step after step) -- a millisecond of planning, record building and upload per step otherwise, before
    # the first kernel can start
    plan_key = None
    planned = None
    if not any_pre:
        t_ = dvol.tensor
        # (block records hold element offsets, not addresses: any volume of this layout can use them; the block lists
        #  are compared by content -- `lists_key` above -- so that a caller who builds them afresh finds the same device
        #  tables, which is also what lets a small batch's captured graph be found again)
        plan_key = (lists_key, tuple(t_.stride()), tuple(t_.shape), str(t_.dtype), ns_max,
                    int(budget_bytes), _MAX_BATCH, RAMP, TAPER)
        planned = bufs.plans.get(plan_key)
    if planned is not None:
        batches = planned[0]
    else:
        batches = plan_batches(shapes, ns_max, budget_bytes,
                               0 if not any_pre else max(lane.pre.bytes_per_voxel() for lane in lanes))
    n_b = len(batches)
    n_l = len(lanes)
    items = [(b, l) for b in range(n_b) for l in range(n_l)]       # batch-major
    n_items = len(items)
    # How far the GPU queue runs ahead of the host.  Raw volumes: EVERY batch is enqueued before the host looks at
    # the first result, so the GPU runs the batches back to back whatever the host is doing (measured: enqueueing
    # batch k + 1 only after the host work of batch k - 1 left the GPU idle ~4 ms per batch once the host work of
    # a batch outlasted the kernels of the next).  Each batch in flight owns a candidate table; the workspace is
    # shared (stream order).  With preprocessing the float64 tiles are double-buffered, so one batch ahead.
    ahead = n_items if not any_pre else PRE_AHEAD
    if any_pre and all(getattr(lane.pre, "retains", lambda c: False)(lane.channel) for lane in lanes):
        # every lane's preprocessed blocks have slots of their own (Preprocessor.retain: the co-localisation's): no buffer
        # set limits how far the queue may run ahead -- and with several lanes PRE_AHEAD items are less than PRE_AHEAD
        # batches
        ahead = max(ahead, min(n_items, PRE_AHEAD_RETAINED * n_l))
    bufs.slots(ahead + 1)
    prepared = None
    if planned is not None:
        prepared = planned[1]
    elif not any_pre:
        # block tables of every batch go to the device BEFORE the first kernel: a pageable host -> device copy
        # waits for everything queued on the stream before it
        prepared = [_make_blocks(dvol, lanes[0].channel, [origins[i] for i in b], [shapes[i] for i in b]) for b in batches]
        if prepared:        # (one upload for all of them: fourteen small copies were 0.4 ms of idle GPU at the start of a step)
            sizes = [blk.nbytes for blk, _ in prepared]
            allrec = _to_device_bytes(np.concatenate([blk.view(np.uint8).reshape(-1) for blk, _ in prepared]),
                                      dvol.tensor.device)
            offs = np.concatenate([[0], np.cumsum(sizes)])
            prepared = [(blk, slot, allrec[int(offs[i]):int(offs[i + 1])]) for i, (blk, slot) in enumerate(prepared)]
        if len(bufs.plans) >= 8:
            bufs.plans.clear()
        bufs.plans[plan_key] = (batches, prepared)
    if any_pre and PRE_SIDE_TAIL and RESCORE_STREAM and all(lane.exact for lane in lanes) and n_items > 1:
        # preprocessed batches alternate between two workspaces as well (the tail of a batch on the second stream beside
        # the next batch's passes): both at their final size before anything is queued on them
        need = 0
        for b in batches:
            slot_b = max(int(shapes[i][0]) * int(shapes[i][1]) * (-(-int(shapes[i][2]) // nat.MMX_ROW_ALIGN) * nat.MMX_ROW_ALIGN)
                         for i in b)
            need = max(need, -(-int(nat.lib().mmx_workspace_bytes(len(b), slot_b, ns_max, 1)) // 4))
        bufs.workspace(need)
        bufs.workspace(need, 1)
        bufs.ws_free = [None, None]
    if not any_pre:        # ... and the shared workspace has its final size before anything is queued on it
        if prepared:
            need = max(-(-int(nat.lib().mmx_workspace_bytes(len(blk), slot, ns_max, 1)) // 4)
                       for blk, slot, _ in prepared)
            bufs.workspace(need)
            if RESCORE_STREAM and n_items > 1:
                bufs.workspace(need, 1)
            bufs.ws_free = [None, None]
    jobs: List[Optional[dict]] = [None] * n_items
    done_events: List = []
    enq = 0
    uploading = dvol._upload is not None
    for k, (b_k, l_k) in enumerate(items):
        while enq < min(n_items, k + 1 + ahead):       # item k itself and `ahead` items behind it
            b_e, l_e = items[enq]
            batch, ln = batches[b_e], lanes[l_e]
            if uploading and enq > k and not dvol.upload_ready(
                    [(origins[i][0], origins[i][0] + shapes[i][0], origins[i][1], origins[i][1] + shapes[i][1]) for i in batch]):
                # a volume still on its way up: this batch's voxels have not been queued for copying yet.  Enqueueing it
                # would block the host until they are -- with finished batches waiting for their host work (all of it
                # then piled up behind the upload's last region: 30 ms at the end of a from-host step).  Item k first.
                break
            jobs[enq] = _enqueue_detect(dvol, ln.channel, [origins[i] for i in batch], [shapes[i] for i in batch],
                                        ln.space, ln.threshold, ln.eps, bufs, enq % (ahead + 1), ln.d_w0, ln.d_w2,
                                        pre=ln.pre, exact=ln.exact, prepared=None if prepared is None else prepared[b_e],
                                        vscale=ln.vscale, vrange=ln.vrange,
                                        buffer_free=(None if enq < ahead + 1 else done_events[enq - (ahead + 1)])
                                        if ln.pre is not None else False, parity=enq)
            done_events.append(jobs[enq]["done"])
            jobs[enq]["batch"] = batch
            if enq == 0:
                global FIRST_ENQUEUED_T, LAST_BATCH_SIZES
                FIRST_ENQUEUED_T, LAST_BATCH_SIZES = time.perf_counter(), [len(b_) for b_ in batches]
            enq += 1
        pending, jobs[k] = jobs[k], None
        ln = lanes[l_k]
        # host + side-stream work of item k, the GPU busy with the items behind it
        peaks = _finish_detect(pending, dvol, ln.space, ln.threshold, ln.eps, bufs, ln.d_w0, ln.d_w2, ln.stats,
                               finisher=ln.finisher if n_items == 1 else None, overlap=ln.overlap)
        if peaks is _FINISHED:
            continue
        if isinstance(peaks, PeakBatch):
            pb = _prune_batch_native(peaks, ln.space, ln.overlap, ln.stats)
            if ln.sink is not None:        # the caller builds its tables from the arrays (native, no per-block lists)
                with torch.cuda.stream(bufs.side):      # (whatever it launches -- co-localisation means -- beside the next batch)
                    ln.sink(pending["batch"], pb)
                continue
            pruned = [pb.blobs(b) for b in range(len(pb))]
            peaks = [pb.block(b) for b in range(len(pb))] if return_peaks else [None] * len(pb)
        else:
            with torch.cuda.stream(bufs.side):
                pruned = _prune_batch(peaks, ln.space, ln.overlap, dvol.tensor.device, ln.stats)
        for i, pk, res in zip(pending["batch"], peaks, pruned):
            ln.results[i] = res
            ln.peaks_out[i] = pk
        if ln.on_batch is not None:    # caller's per-block post-processing, still overlapped: whatever it launches
            with torch.cuda.stream(bufs.side):           # (co-localisation means) runs beside the next batch's kernels
                ln.on_batch(pending["batch"], pruned)
    results = [ln.results for ln in lanes]
    return (results, [ln.peaks_out for ln in lanes]) if return_peaks else results


# --------------------------------------------------------------------------- A0-A4
def _enqueue_detect(dvol, channel, origins, shapes, space: ScaleSpace, thr: float, eps: float,
                    bufs: _Buffers, which: int, d_w0, d_w2, cap: Optional[int] = None, pre=None,
                    exact: bool = False, prepared=None, vscale: Optional[float] = None, vrange=None,
                    buffer_free=False, parity: Optional[int] = None):
    """Enqueue (P1-P3,) A0-A4 of one batch on the current stream; nothing here waits for the GPU.
    ``exact``: also re-score every candidate in float64 (otherwise ``_resolve_peaks`` re-scores the few
    whose decision depends on it)."""
    L = nat.lib()
    dev = dvol.tensor.device
    d_blocks = None
    # a volume still on its way up (`DeviceVolume.stream_wait`): this batch waits for the slabs its blocks touch, on
    # every stream that reads voxels (the passes, the voxel copy of the tiled path, the exact re-score)
    boxes = ([(int(o[0]), int(o[0]) + int(s_[0]), int(o[1]), int(o[1]) + int(s_[1])) for o, s_ in zip(origins, shapes)]
             if dvol._upload is not None else None)
    if pre is None:
        if dvol._upload is not None:
            dvol.stream_wait(None, [torch.cuda.current_stream(), bufs.pack_stream, bufs.rescore_stream, bufs.side], boxes)
        if prepared is not None:
            blocks, slot, d_blocks = prepared
        else:
            blocks, slot = _make_blocks(dvol, channel, origins, shapes)
        vol32 = dvol.view(channel, True)
        vol_exact = dvol.view(channel, False)
        store_f32 = 1 if dvol.np_dtype == np.float32 else 0
    else:
        if PRE_STREAM and buffer_free is not False:
            # on its own stream: it may start as soon as the batch that last used this buffer set is done
            # (`buffer_free`: that batch's completion event), i.e. while the batch before this one is still filtering.
            # (None: a buffer set no batch of this call has used yet -- behind whatever is queued on the main stream.
            #  Letting the first batches' preprocessing start at once instead measured nothing on C3 --denoise 25
            #  and +10 ms on C5, round 5.)
            main = torch.cuda.current_stream()
            bufs.pre_stream.wait_stream(main) if buffer_free is None else bufs.pre_stream.wait_event(buffer_free)
            with torch.cuda.stream(bufs.pre_stream):
                dvol.stream_wait(None, [torch.cuda.current_stream(), bufs.side], boxes)     # (side: co-localisation means)
                blocks, slot, vol32, vol_exact = pre.run(dvol, channel, origins, shapes, which)
                ready = torch.cuda.Event()
                ready.record()
            timed = bool(L.mmx_timing_is_enabled())
            if timed:       # how long the LoG stream waits for this batch's preprocessing (bench.py: pre_stream_wait_ms)
                before = torch.cuda.Event(enable_timing=True)
                before.record()
            main.wait_event(ready)
            if timed:
                after = torch.cuda.Event(enable_timing=True)
                after.record()
                PRE_WAITS.append((before, after))
        else:
            dvol.stream_wait(None, [torch.cuda.current_stream(), bufs.side], boxes)
            blocks, slot, vol32, vol_exact = pre.run(dvol, channel, origins, shapes, which)
        store_f32 = int(getattr(pre, "store_f32", 0))
    nb, ns = len(blocks), len(space.sigmas)
    if slot >= (1 << 29):
        raise nat.MmxError("block too large for one workspace slot (>= 2^29 voxels)")
    mask_words = (nb * slot) >> 5            # 16-byte entries per sigma (include/mmx.h: d_nms_mask)
    # raw volumes: the batches alternate between two workspaces, and everything after a batch's last LoG kernel -- NMS,
    # probe expansion, exact re-score, copies -- runs on a second stream beside the next batch's LoG kernels
    # (preprocessed batches too with PRE_SIDE_TAIL -- measured and left off; their two workspaces are sized by
    #  blob_log_blocks before anything is queued)
    side_tail = bool(RESCORE_STREAM and exact and ((pre is None and prepared is not None) or
                                                   (pre is not None and PRE_SIDE_TAIL and bufs.ws2 is not None)))
    ws_i = ((which if parity is None else parity) & 1) if (side_tail and bufs.ws2 is not None) else 0
    ws = bufs.workspace(-(-int(L.mmx_workspace_bytes(nb, slot, ns, 1)) // 4), ws_i)
    # ... and the voxel copy of the tiled path, the first kernel of a batch, on a third: it only needs the workspace
    pack_side = side_tail and PACK_STREAM and bufs.ws2 is not None
    native_batch = bool(NATIVE_BATCH and pre is None and exact and HOST_PATH == "native")
    if bufs.ws_free[ws_i] is not None and not native_batch:   # (also a batch that is nominated again, on the main stream: same workspace)
        _stream_wait(bufs.pack_stream if pack_side else torch.cuda.current_stream(), bufs.ws_free[ws_i])
    if d_blocks is None:
        d_blocks = _to_device_bytes(blocks, dev)
    stream = _stream_ptr()
    log_base = ws.data_ptr() + 4 * nb * slot * 4
    # NMS pre-filter masks, [ns][nb][slot / 32] uint64: written by the Y pass of the fused path
    mask_base = (log_base + ns * nb * slot * 4 + 15) & ~15
    if native_batch:
        # ---- the whole batch in one native call (mmx_detect_batch): voxel copy, every scale, NMS, probes, re-score, copies
        global LAST_ZX_PATH, LAST_Q16_BOUND, LAST_NMS_BAND
        is_float = vol32.dtype == nat.MMX_F32
        if is_float:
            float_ok = vrange is not None and max(abs(vrange[0]), abs(vrange[1])) <= FLOAT_TILED_RANGE[1]
            vol32.value_range = 0.0 if not float_ok else (max(vrange[1], 1e-30) if vrange[0] >= 0.0
                                                          else -max(abs(vrange[0]), abs(vrange[1])))
        n_vox = int(sum(int(np.prod(s)) for s in shapes))
        if cap is None:
            cap = max(4096, min(n_vox * ns, n_vox // 2000 * ns + 65536))
        table = bufs.cand_table(which, cap)
        count = bufs.counts[which]
        ev_read, ev_done = bufs.events(which)
        side = bufs.rescore_stream if side_tail else None
        a = nat.DetectArgs()
        a.vol32, a.vol_exact = ctypes.pointer(vol32), ctypes.pointer(vol_exact)
        a.d_blocks, a.h_blocks = d_blocks.data_ptr(), blocks.ctypes.data
        a.n_blocks, a.n_sigma, a.slot_elems = nb, ns, slot
        a.h_w0, a.h_w2 = space.w0_tab.ctypes.data, space.w2_tab.ctypes.data
        a.d_w0, a.d_w2 = d_w0.data_ptr(), d_w2.data_ptr()
        a.h_radius, a.h_norm = space.radii.ctypes.data, space.norms.ctypes.data
        a.d_work, a.work_bytes = ws.data_ptr(), ws.numel() * 4
        a.thr, a.eps = thr, eps
        a.d_cands, a.cap, a.d_count = table.data_ptr(), cap, count.data_ptr()
        a.h_count = bufs.host_counts[which].data_ptr()
        a.h_cands, a.h_prefix = bufs.host_table(which).data_ptr(), min(cap, _PREFIX_ENTRIES)
        a.zx_mode, a.zx_flags, a.store_f32, a.exact, a.expand = ZX_MODE, ZX_FLAGS, store_f32, 1, 1
        a.stream = stream
        a.tail_stream = side.cuda_stream if side is not None else stream
        a.pack_stream = bufs.pack_stream.cuda_stream if pack_side else stream
        prev = bufs.ws_free[ws_i]
        if prev is not None and not isinstance(prev, _NativeEvent):     # (left by the call-by-call form)
            _stream_wait(bufs.pack_stream if pack_side else torch.cuda.current_stream(), prev)
            prev = None
        a.ev_work_free = prev.handle if prev is not None else None
        a.ev_work_read = ev_read.handle if side_tail else None
        a.ev_done = ev_done.handle
        info = nat.DetectInfo()
        rc = _launch_batch(L, a, info, bufs, nb, blocks, space, vol32, vol_exact)
        if rc != 0:
            detail = L.mmx_detect_last_error().decode()
            nat.check(rc, "mmx_detect_batch" + (f" [{detail}]" if detail and rc == 2 else ""))
        LAST_ZX_PATH = info.zx_path
        if info.zx_path == nat.MMX_ZX_TILED_Q16:
            LAST_Q16_BOUND, LAST_NMS_BAND = info.q16_bound, eps
        if side_tail:
            bufs.ws_free[ws_i] = ev_read
        return dict(blocks=blocks, d_blocks=d_blocks, shapes=shapes, origins=origins, channel=channel,
                    nb=nb, ns=ns, n_vox=n_vox, cap=cap, which=which, done=ev_done, store_f32=store_f32,
                    vol_exact=vol_exact, pre=pre, exact=exact, eps=eps, native=True, vscale=vscale, vrange=vrange)
    written = ctypes.c_int(0)
    path = ctypes.c_int(0)

    def passes(with_mask: bool, mode: int):
        """Every scale of the batch -> the set of entry layouts the calls reported (0 = no entries)."""
        global LAST_ZX_PATH, LAST_Q16_BOUND, LAST_NMS_BAND
        layouts = set()
        # the tiled path works from an operand-ordered copy of the voxels that does not depend on sigma: made
        # once here, trusted by the calls below for as long as every call so far ran the tiled path (any other
        # path uses the same part of the workspace for something else)
        packed = False
        tiled_mode = nat.MMX_ZX_TILED
        is_float = vol32.dtype == nat.MMX_F32
        # float voxels (float images, preprocessed blocks): the tiled path holds each as two float16 pieces, which
        # suits values of ordinary magnitude -- the range is known here, not in the library (mmx_volume.value_range)
        float_ok = is_float and vrange is not None and max(abs(vrange[0]), abs(vrange[1])) <= FLOAT_TILED_RANGE[1]
        nonneg = not is_float or (float_ok and vrange[0] >= 0.0)
        if is_float:
            vol32.value_range = 0.0 if not float_ok else (max(vrange[1], 1e-30) if vrange[0] >= 0.0
                                                          else -max(abs(vrange[0]), abs(vrange[1])))
        if mode in (nat.MMX_ZX_AUTO, nat.MMX_ZX_TILED_Q16) and nonneg:
            # 16-bit intermediates when the band covers their rounding error fourfold (or when asked for by name)
            bound = max(float(L.mmx_tiled_q16_error_bound(nat.as_double_ptr(space.w0[s]), nat.as_double_ptr(space.w2[s]),
                                                          int(space.radii[s]), float(space.norms[s]))) for s in range(ns))
            bound *= 1.0 if not is_float else float(vol32.value_range)
            # (by name: taken whatever the band -- kernel experiments and the band-retry tests ask for it with narrow
            #  bands; the run-time check |float32 - float64| < eps / 4 on every re-scored candidate then widens the band)
            if mode == nat.MMX_ZX_TILED_Q16 or (0 <= 4.0 * bound <= eps and bound <= LOG_ABS_TOL):
                tiled_mode = nat.MMX_ZX_TILED_Q16
                LAST_Q16_BOUND, LAST_NMS_BAND = bound, eps
        if (mode in (nat.MMX_ZX_AUTO, nat.MMX_ZX_TILED, nat.MMX_ZX_TILED_Q16) and not is_float) or \
                (mode in (nat.MMX_ZX_AUTO, nat.MMX_ZX_TILED) and float_ok):
            nonlocal pack_side
            if pack_side:
                with torch.cuda.stream(bufs.pack_stream):
                    rc = L.mmx_zx_pack(ctypes.byref(vol32), d_blocks.data_ptr(), blocks.ctypes.data, nb, slot,
                                       ws.data_ptr(), bufs.pack_stream.cuda_stream)
                torch.cuda.current_stream().wait_stream(bufs.pack_stream)
                pack_side = False                # (a second round of passes, should one be needed: in stream order)
            else:
                rc = L.mmx_zx_pack(ctypes.byref(vol32), d_blocks.data_ptr(), blocks.ctypes.data, nb, slot,
                                   ws.data_ptr(), stream)
            if rc not in (0, 5):                 # MMX_OK, MMX_ERR_UNSUPPORTED (float voxels, workspace shape)
                nat.check(rc, "mmx_zx_pack")
            packed = rc == 0
        for s in range(ns):
            nat.check(L.mmx_log_batch_f32(
                ctypes.byref(vol32), d_blocks.data_ptr(), blocks.ctypes.data, nb, slot,
                nat.as_double_ptr(space.w0[s]), nat.as_double_ptr(space.w2[s]), int(space.radii[s]),
                float(space.norms[s]), log_base + s * nb * slot * 4, ws.data_ptr(),
                (mask_base + s * mask_words * 16) if with_mask else None, thr - eps, eps,
                ctypes.byref(written), (tiled_mode | nat.MMX_ZX_PREPACKED | ZX_FLAGS) if packed else mode,
                ctypes.byref(path), stream),
                "mmx_log_batch_f32")
            LAST_ZX_PATH = path.value
            packed = packed and path.value == tiled_mode
            layouts.add(written.value if with_mask else 0)
        return layouts

    # With the entries the Y pass leaves whole segments of the cube unwritten, so it is all scales, in one
    # layout, or none: if one scale cannot produce them (a radius outside the fused kernels, tiny blocks) or
    # the scales ran different kernels, every scale is computed again -- with the packed kernel's entries if a
    # scale produced those, else in full.
    layouts = passes(True, ZX_MODE)
    if layouts == {nat.MMX_MASK_ROWS, nat.MMX_MASK_QUADS}:
        layouts = passes(True, nat.MMX_ZX_PACKED)
    if len(layouts) > 1:
        layouts = passes(False, ZX_MODE)
    mask_layout = layouts.pop()
    mask_ok = mask_layout > 0
    n_vox = int(sum(int(np.prod(s)) for s in shapes))
    if cap is None:
        cap = max(4096, min(n_vox * ns, n_vox // 2000 * ns + 65536))
    table = bufs.cand_table(which, cap)
    count = bufs.counts[which]
    native = bool(exact and HOST_PATH == "native")

    def tail(stream_ptr):
        count.zero_()
        nat.check(L.mmx_peaks_batch(log_base, mask_base if mask_ok else None, mask_layout, ns, d_blocks.data_ptr(),
                                    blocks.ctypes.data, nb, slot, thr, eps, table.data_ptr(), cap,
                                    count.data_ptr(), stream_ptr),
                  "mmx_peaks_batch")
        if side_tail:                    # (the workspace may be written again once the NMS has read it)
            bufs.ws_free[ws_i] = torch.cuda.Event()
            bufs.ws_free[ws_i].record()
        if native:
            # the neighbours that can out-vote the contested candidates join the table: one re-score, one copy
            nat.check(L.mmx_expand_probes(table.data_ptr(), cap, count.data_ptr(), count.data_ptr() + 4,
                                          d_blocks.data_ptr(), nb, ns, stream_ptr), "mmx_expand_probes")
        if exact:
            nat.check(L.mmx_rescore_f64(
                ctypes.byref(vol_exact), d_blocks.data_ptr(), nb, table.data_ptr(), cap,
                count.data_ptr(), d_w0.data_ptr(), d_w2.data_ptr(), nat.as_int32_ptr(space.radii),
                nat.as_double_ptr(space.norms), ns, store_f32, stream_ptr), "mmx_rescore_f64")
        bufs.host_counts[which].copy_(count, non_blocking=True)
        if native:
            n_pre = min(cap, _PREFIX_ENTRIES) * nat.CAND_DTYPE.itemsize
            bufs.host_table(which)[:n_pre].copy_(table[:n_pre], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return ev
    if side_tail:
        # (every batch of a raw volume owns its candidate table; the volume is never written)
        side = bufs.rescore_stream
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            done = tail(side.cuda_stream)
    else:
        done = tail(stream)
    return dict(blocks=blocks, d_blocks=d_blocks, shapes=shapes, origins=origins, channel=channel,
                nb=nb, ns=ns, n_vox=n_vox, cap=cap, which=which, done=done, store_f32=store_f32,
                vol_exact=vol_exact, pre=pre, exact=exact, eps=eps, native=native, vscale=vscale, vrange=vrange)


def _launch_batch(L, a, info, bufs: _Buffers, nb: int, blocks, space, vol32, vol_exact) -> int:
    """``mmx_detect_batch(a)`` -- or, for a small batch that has come by with the very same arguments before (a small
    volume detected step after step: same buffers, geometry, scales, band), the replay of its captured hipGraph: one
    launch instead of a dozen.  The second sighting of a key captures, later ones replay; a capture is refused while
    per-kernel timing is on (its events cannot live inside one) and the plain call is made instead."""
    if not (0 < nb <= GRAPH_BLOCKS) or a.ev_work_free:
        return L.mmx_detect_batch(ctypes.byref(a), ctypes.byref(info))
    if L.mmx_timing_is_enabled():                   # (a replay would hide the kernels from the per-kernel timing, and
        return L.mmx_detect_batch(ctypes.byref(a), ctypes.byref(info))      # a capture cannot hold its events)
    caller = None
    if not a.stream:
        # the caller works on the default stream, which cannot be captured: the batch runs on a stream of this module's
        # instead, ordered after what the caller has queued and before what it queues next
        caller = torch.cuda.current_stream()
        if bufs.graph_stream is None:
            bufs.graph_stream = torch.cuda.Stream(device=bufs.dev)
        gs = bufs.graph_stream
        origin = a.stream
        for name in ("stream", "tail_stream", "pack_stream"):
            if getattr(a, name) == origin:
                setattr(a, name, gs.cuda_stream)
        gs.wait_stream(caller)
    rc = _launch_batch_graph(L, a, info, bufs, blocks, space, vol32, vol_exact)
    if caller is not None:
        caller.wait_stream(bufs.graph_stream)
    return rc


def _launch_batch_graph(L, a, info, bufs: _Buffers, blocks, space, vol32, vol_exact) -> int:
    # everything a node of the graph would freeze (the two volume records by content: their addresses change per call)
    key = (bytes(vol32), bytes(vol_exact), blocks.tobytes(), space.sigmas.tobytes(), a.d_blocks, a.slot_elems, a.d_w0,
           a.d_w2, a.d_work, a.work_bytes, a.thr, a.eps, a.d_cands, a.cap, a.h_prefix, a.d_count, a.h_count, a.h_cands,
           a.zx_mode, a.zx_flags, a.store_f32, a.stream, a.tail_stream, a.pack_stream)
    hit = bufs.graphs.get(key)
    if hit is None:                                 # first sighting: remember it, run it plainly
        if len(bufs.graphs) >= 16:
            bufs.drop_graphs()
        bufs.graphs[key] = ()
        return L.mmx_detect_batch(ctypes.byref(a), ctypes.byref(info))
    if hit == "plain":                              # a capture of this batch failed once: launched call by call
        return L.mmx_detect_batch(ctypes.byref(a), ctypes.byref(info))
    if not hit:
        # second sighting: capture (this only records the launches; the replay below runs them)
        graph = ctypes.c_void_p()
        rc = L.mmx_detect_batch_capture(ctypes.byref(a), ctypes.byref(info), ctypes.byref(graph))
        if rc != 0:
            # not capturable (the library says why in mmx_detect_last_error): never tried again for this key
            bufs.graphs[key] = "plain"
            return L.mmx_detect_batch(ctypes.byref(a), ctypes.byref(info))
        hit = bufs.graphs[key] = (graph.value, nat.DetectInfo.from_buffer_copy(info), (blocks, space))
    graph, saved, _ = hit
    global GRAPH_REPLAYS
    GRAPH_REPLAYS += 1
    ctypes.memmove(ctypes.byref(info), ctypes.byref(saved), ctypes.sizeof(info))
    rc = L.mmx_graph_launch(graph, a.stream, a.ev_done, None)
    if rc == 0 and a.ev_work_read:
        # (the graph is ordered as a whole on `stream`: "the NMS has read the workspace" holds once it is through)
        rc = L.mmx_event_record(a.ev_work_read, a.stream)
    return rc


_FINISHED = object()        # what `_finish_detect` returns when a finisher has taken the whole batch


def _finish_detect(job, dvol, space: ScaleSpace, thr: float, eps: float, bufs: _Buffers, d_w0, d_w2,
                   stats: BatchStats, finisher=None, overlap: float = 0.5):
    """Wait for one batch's candidates and turn them into ordered raw peaks
    ``(coords int64 (n, 4), values float64 (n,))`` per block."""
    job["done"].synchronize()
    global LAST_BATCH_DONE_T
    LAST_BATCH_DONE_T = time.perf_counter()      # (bench.py: what a step still does after its last kernel)
    eps = job.get("eps", eps)
    which, cap, ns = job["which"], job["cap"], job["ns"]
    native = job.get("native", False)
    words = bufs.host_counts[which].numpy().view(np.uint32)
    count = int(words[0])
    n_cands = int(words[1]) if native else count
    if native and n_cands >= job["n_vox"] * ns:
        count = n_cands = 0                 # constant cubes (below)
    if count > cap:
        if count >= job["n_vox"] * ns and not native:
            # every voxel of every block "equals its maximum": only possible for constant
            # cubes, which scikit-image treats as having no peaks (peak.py:41-43)
            count = 0
        else:
            # table overflow (rare): redo this batch with a table that fits; the pipeline
            # already reused the workspace, so the passes run again
            torch.cuda.current_stream().synchronize()
            redo = _enqueue_detect(dvol, job["channel"], job["origins"], job["shapes"], space, thr,
                                   eps, bufs, which, d_w0, d_w2, cap=count + 1024, pre=job.get("pre"),
                                   exact=job.get("exact", False), vscale=job.get("vscale"), vrange=job.get("vrange"))
            redo["batch"] = job.get("batch")
            return _finish_detect(redo, dvol, space, thr, eps, bufs, d_w0, d_w2, stats)
    with torch.cuda.stream(bufs.side):
        table = bufs.cands[which]
        try:
            if native:
                if count <= _PREFIX_ENTRIES:
                    cands = bufs.host_table(which).numpy()[:count * nat.CAND_DTYPE.itemsize].view(nat.CAND_DTYPE)
                else:
                    cands = table[:count * nat.CAND_DTYPE.itemsize].cpu().numpy().view(nat.CAND_DTYPE)
                if finisher is not None and not job.get("retries") and finisher(
                        job.get("batch"), cands, n_cands, job["blocks"], space, thr, eps, overlap, stats):
                    stats.n_blocks += job["nb"]
                    stats.n_voxels += job["n_vox"]
                    stats.n_candidates += n_cands
                    return _FINISHED
                out = _resolve_peaks_native(cands, n_cands, job["blocks"], ns, thr, stats, eps)
            else:
                cands = (table[:count * nat.CAND_DTYPE.itemsize].cpu().numpy().view(nat.CAND_DTYPE)
                         if count else np.zeros(0, dtype=nat.CAND_DTYPE))
                out = _resolve_peaks(cands, job["blocks"], job["shapes"], ns, thr, dvol, job["vol_exact"],
                                     job["d_blocks"], d_w0, d_w2, space, job["store_f32"], stats, eps,
                                     job.get("exact", False))
        except _BandTooNarrow as exc:
            out = None
            err = exc.err
            wider = max(2.0 * eps, 8.0 * err)
        if out is not None:
            stats.n_blocks += job["nb"]
            stats.n_voxels += job["n_vox"]
            stats.n_candidates += n_cands
            return out
    # the float32 values were further from the exact ones than the band allows: nominate this batch again
    # with a band of 8 x the deviation found (the exact re-score then decides as always).  The pipeline has
    # reused the workspace, so the passes run again.
    if job.get("retries", 0) >= 6:
        raise nat.MmxError(f"float32 LoG deviates from the exact values by {err:.3g}: no usable band")
    stats.n_band_retries += 1
    torch.cuda.current_stream().synchronize()
    redo = _enqueue_detect(dvol, job["channel"], job["origins"], job["shapes"], space, thr, wider, bufs, which,
                           d_w0, d_w2, cap=None, pre=job.get("pre"), exact=True, vscale=job.get("vscale"), vrange=job.get("vrange"))
    redo["batch"] = job.get("batch")
    redo["retries"] = job.get("retries", 0) + 1
    return _finish_detect(redo, dvol, space, thr, wider, bufs, d_w0, d_w2, stats)


def blob_log(image, min_sigma=1, max_sigma=50, num_sigma=10, threshold=.2, overlap=.5):
    """``skimage.feature.blob_log`` signature for one 3-D image (host array or tensor)."""
    dvol = image if isinstance(image, DeviceVolume) else DeviceVolume(image)
    if dvol.multichannel:
        raise ValueError("blob_log takes a single-channel (z, y, x) image")
    return blob_log_blocks(dvol, 0, [(0, 0, 0)], [dvol.shape[:3]], min_sigma, max_sigma, num_sigma,
                           threshold, overlap)[0]
