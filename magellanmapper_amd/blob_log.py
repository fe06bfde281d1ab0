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
maximum / of the threshold).  Every candidate gets its float64 cube value recomputed on
the device with SciPy's exact operation order; contested candidates also get their 80
neighbours' exact values.  Peak membership, the descending-response order and therefore
the integer blob coordinates are decided on float64 values that equal the reference's bit
for bit.  Nothing here falls back to a CPU implementation: without the HIP library and a
GPU the calls raise.
"""
from __future__ import annotations

import ctypes
import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _native as nat
from . import kernels1d as k1

try:  # torch is the device-memory / stream plumbing
    import torch
except Exception as exc:  # pragma: no cover
    raise ImportError("magellanmapper_amd needs PyTorch-ROCm for device memory") from exc

#: half-width of the float32 "contested" band, relative to the input's value scale
EPS_REL = 2e-5
#: band around the overlap limit inside which the host re-evaluates the fraction exactly
OVERLAP_BAND = 1e-9

_NP_TO_MMX = {np.dtype(np.uint8): nat.MMX_U8, np.dtype(np.uint16): nat.MMX_U16,
              np.dtype(np.float32): nat.MMX_F32, np.dtype(np.float64): nat.MMX_F64}
_TORCH_DTYPES = {np.dtype(np.uint8): torch.uint8, np.dtype(np.uint16): torch.uint16,
                 np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}


def _require_gpu() -> "torch.device":
    if not torch.cuda.is_available():
        raise nat.MmxError("no GPU visible: the blob-detection path is HIP-only (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def _stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


class DeviceVolume:
    """A ``(z, y, x[, c])`` image resident in HBM.

    Integer images other than uint8/uint16 and float16 are converted like
    ``skimage.img_as_float`` would (to float64) on the host first.  For a float64 image a
    float32 copy feeds the float32 passes; the exact re-score reads the float64 original.
    """

    def __init__(self, image, device: Optional["torch.device"] = None):
        dev = device or _require_gpu()
        if isinstance(image, torch.Tensor):
            t = image
            np_dtype = np.dtype(str(t.dtype).replace("torch.", ""))
        else:
            arr = np.asarray(image)
            if arr.dtype not in _NP_TO_MMX:
                arr = _img_as_float_host(arr)
            np_dtype = arr.dtype
            t = torch.from_numpy(np.ascontiguousarray(arr))
        if np_dtype not in _NP_TO_MMX:
            raise TypeError(f"unsupported voxel type {np_dtype}")
        if t.ndim not in (3, 4):
            raise ValueError("image must be (z, y, x) or (z, y, x, c)")
        self.tensor = t.to(dev).contiguous()
        self.np_dtype = np_dtype
        self.shape = tuple(self.tensor.shape)
        self.n_channels = self.shape[3] if self.tensor.ndim == 4 else 1
        self._f32 = None
        self._scale = None

    @property
    def multichannel(self) -> bool:
        return self.tensor.ndim == 4

    def value_scale(self) -> float:
        """Magnitude of the image values after ``img_as_float`` (1 for integer images)."""
        if self._scale is None:
            if self.np_dtype.kind == "f":
                m = float(self.tensor.abs().max().item()) if self.tensor.numel() else 1.0
                self._scale = max(1.0, m)
            else:
                self._scale = 1.0
        return self._scale

    def _strides(self, t) -> Tuple[int, int, int]:
        st = t.stride()
        return st[0], st[1], st[2]

    def view(self, channel: int, for_f32_passes: bool) -> nat.Volume:
        t = self.tensor
        if for_f32_passes and self.np_dtype == np.float64:
            if self._f32 is None:
                self._f32 = t.to(torch.float32)
            t = self._f32
            code = nat.MMX_F32
        else:
            code = _NP_TO_MMX[self.np_dtype]
        sz, sy, sx = self._strides(t)
        ptr = int(t.data_ptr()) + (int(channel) if self.multichannel else 0) * t.element_size()
        return nat.Volume(ptr, code, 0, int(sz), int(sy), int(sx))


def _img_as_float_host(arr: np.ndarray) -> np.ndarray:
    """``skimage.util.img_as_float`` for the dtypes the device path does not read natively
    (skimage/util/dtype.py:310-328)."""
    kind = arr.dtype.kind
    if kind == "b":
        return arr.astype(np.float64)
    if kind == "u":
        return np.multiply(arr, 1.0 / np.iinfo(arr.dtype).max, dtype=np.float64)
    if kind == "i":
        info = np.iinfo(arr.dtype)
        out = np.add(arr, 0.5, dtype=np.float64)
        out *= 2 / (float(info.max) - float(info.min))
        return out
    if kind == "f":
        return arr.astype(np.float32 if arr.dtype.itemsize < 4 else arr.dtype)
    raise TypeError(f"cannot use {arr.dtype} as an image")


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

    @classmethod
    def make(cls, min_sigma: float, max_sigma: float, num_sigma: int) -> "ScaleSpace":
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


@dataclass
class BatchStats:
    """Counters of one :func:`blob_log_blocks` call (also feeds bench.py)."""
    n_blocks: int = 0
    n_voxels: int = 0
    n_candidates: int = 0
    n_contested: int = 0
    n_probes: int = 0
    n_peaks: int = 0
    n_blobs: int = 0
    n_overlap_pairs: int = 0
    n_order_fallbacks: int = 0
    max_f32_error: float = 0.0


def plan_batches(shapes: Sequence[Tuple[int, int, int]], num_sigma: int,
                 budget_bytes: int) -> List[List[int]]:
    """Group block indices into batches whose workspace fits ``budget_bytes``.

    Workspace = (4 + num_sigma) float32 arrays of ``n_blocks * slot`` voxels, slot = the
    largest block of the batch.  Blocks keep their order (z-major grid order).
    """
    per_vox = (4 + num_sigma) * 4
    batches: List[List[int]] = []
    cur: List[int] = []
    cur_slot = 0
    for i, shp in enumerate(shapes):
        vox = int(shp[0]) * int(shp[1]) * (-(-int(shp[2]) // nat.MMX_ROW_ALIGN) * nat.MMX_ROW_ALIGN)
        slot = max(cur_slot, vox)
        if cur and (len(cur) + 1) * slot * per_vox > budget_bytes:
            batches.append(cur)
            cur, slot = [], vox
        cur.append(i)
        cur_slot = slot
    if cur:
        batches.append(cur)
    return batches


class _Buffers:
    """Device scratch that is reused across batches of one call."""

    def __init__(self, dev):
        self.dev = dev
        self.ws = None
        self.cands = None
        self.count = torch.zeros(1, dtype=torch.int32, device=dev)

    def workspace(self, n_floats: int):
        if self.ws is None or self.ws.numel() < n_floats:
            self.ws = None
            self.ws = torch.empty(n_floats, dtype=torch.float32, device=self.dev)
        return self.ws

    def cand_table(self, cap: int):
        if self.cands is None or self.cands.numel() < cap * nat.CAND_DTYPE.itemsize:
            self.cands = None
            self.cands = torch.empty(cap * nat.CAND_DTYPE.itemsize, dtype=torch.uint8, device=self.dev)
        return self.cands


def _to_device_bytes(arr: np.ndarray, dev) -> "torch.Tensor":
    return torch.from_numpy(np.ascontiguousarray(arr).view(np.uint8).reshape(-1)).to(dev)


def log_cube_blocks(dvol: DeviceVolume, channel: int, origins, shapes, space: ScaleSpace,
                    generic: bool = False) -> List[np.ndarray]:
    """Float32 ``(z, y, x, sigma)`` cubes of the given blocks (A0-A3 only; used by tests
    for the 1e-4 LoG tolerance and by the profiling scripts)."""
    dev = dvol.tensor.device
    L = nat.lib()
    blocks, slot = _make_blocks(dvol, channel, origins, shapes)
    nb, ns = len(blocks), len(space.sigmas)
    ws = torch.empty((4 + ns) * nb * slot, dtype=torch.float32, device=dev)
    d_blocks = _to_device_bytes(blocks, dev)
    vol32 = dvol.view(channel, True)
    fn = L.mmx_log_batch_f32_generic if generic else L.mmx_log_batch_f32
    log_base = ws.data_ptr() + 4 * nb * slot * 4
    for s in range(ns):
        nat.check(fn(ctypes.byref(vol32), d_blocks.data_ptr(), blocks.ctypes.data, nb, slot,
                     nat.as_double_ptr(space.w0[s]), nat.as_double_ptr(space.w2[s]),
                     int(space.radii[s]), float(space.norms[s]),
                     log_base + s * nb * slot * 4, ws.data_ptr(), _stream_ptr()), "mmx_log_batch_f32")
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
    t = dvol.tensor
    sz, sy, sx = t.stride()[0], t.stride()[1], t.stride()[2]
    blocks = np.zeros(len(shapes), dtype=nat.BLOCK_DTYPE)
    slot = 1
    for i, (o, shp) in enumerate(zip(origins, shapes)):
        for ax in range(3):
            if o[ax] < 0 or shp[ax] < 1 or o[ax] + shp[ax] > dvol.shape[ax]:
                raise ValueError("block outside the volume")
        px = -(-int(shp[2]) // nat.MMX_ROW_ALIGN) * nat.MMX_ROW_ALIGN   # 128-B aligned rows
        blocks[i] = (int(o[0]) * sz + int(o[1]) * sy + int(o[2]) * sx, shp[0], shp[1], shp[2], i, px, 0)
        slot = max(slot, int(shp[0]) * int(shp[1]) * px)
    return blocks, slot


def blob_log_blocks(dvol: DeviceVolume, channel: int, origins: Sequence[Sequence[int]],
                    shapes: Sequence[Sequence[int]], min_sigma: float, max_sigma: float,
                    num_sigma: int, threshold: float, overlap: float, *,
                    budget_bytes: int = 24 << 30, stats: Optional[BatchStats] = None,
                    return_peaks: bool = False):
    """``blob_log`` of every block -> list of ``(n, 4)`` float64 ``[z, y, x, sigma]`` arrays.

    Each block is an independent image exactly as each reference worker's sub-ROI is
    (reference magmap/cv/stack_detect.py:79): reflect boundaries at the block faces,
    coordinates relative to the block.  Blocks without blobs give ``np.empty((0, 3))`` like
    scikit-image does (blob.py:516-517).  Row order equals the reference's.
    """
    _require_gpu()
    space = ScaleSpace.make(min_sigma, max_sigma, num_sigma)
    shapes = [tuple(int(v) for v in s) for s in shapes]
    origins = [tuple(int(v) for v in o) for o in origins]
    stats = stats if stats is not None else BatchStats()
    results: List[Optional[np.ndarray]] = [None] * len(shapes)
    peaks_out: List[Optional[Tuple[np.ndarray, np.ndarray]]] = [None] * len(shapes)
    bufs = _Buffers(dvol.tensor.device)
    eps = EPS_REL * dvol.value_scale()
    for batch in plan_batches(shapes, len(space.sigmas), budget_bytes):
        peaks = _detect_batch(dvol, channel, [origins[i] for i in batch], [shapes[i] for i in batch],
                              space, float(threshold), eps, bufs, stats)
        pruned = _prune_batch(peaks, space, float(overlap), dvol.tensor.device, stats)
        for i, pk, res in zip(batch, peaks, pruned):
            results[i] = res
            peaks_out[i] = pk
    return (results, peaks_out) if return_peaks else results


# --------------------------------------------------------------------------- A0-A4
def _detect_batch(dvol, channel, origins, shapes, space: ScaleSpace, thr: float, eps: float,
                  bufs: _Buffers, stats: BatchStats):
    """Ordered raw peaks ``(coords int64 (n, 4), values float64 (n,))`` per block."""
    L = nat.lib()
    dev = dvol.tensor.device
    blocks, slot = _make_blocks(dvol, channel, origins, shapes)
    nb, ns = len(blocks), len(space.sigmas)
    if slot >= (1 << 29):
        raise nat.MmxError("block too large for one workspace slot (>= 2^29 voxels)")
    ws = bufs.workspace((4 + ns) * nb * slot)
    d_blocks = _to_device_bytes(blocks, dev)
    vol32 = dvol.view(channel, True)
    vol_exact = dvol.view(channel, False)
    stream = _stream_ptr()
    log_base = ws.data_ptr() + 4 * nb * slot * 4
    for s in range(ns):
        nat.check(L.mmx_log_batch_f32(
            ctypes.byref(vol32), d_blocks.data_ptr(), blocks.ctypes.data, nb, slot,
            nat.as_double_ptr(space.w0[s]), nat.as_double_ptr(space.w2[s]), int(space.radii[s]),
            float(space.norms[s]), log_base + s * nb * slot * 4, ws.data_ptr(), stream),
            "mmx_log_batch_f32")
    n_vox = int(sum(int(np.prod(s)) for s in shapes))
    d_w0 = torch.from_numpy(space.w0_tab).to(dev)
    d_w2 = torch.from_numpy(space.w2_tab).to(dev)
    store_f32 = 1 if dvol.np_dtype == np.float32 else 0
    cap = max(4096, min(n_vox * ns, n_vox // 2000 * ns + 65536))
    while True:
        table = bufs.cand_table(cap)
        bufs.count.zero_()
        nat.check(L.mmx_peaks_batch(log_base, ns, d_blocks.data_ptr(), blocks.ctypes.data, nb, slot,
                                    thr, eps, table.data_ptr(), cap, bufs.count.data_ptr(), stream),
                  "mmx_peaks_batch")
        nat.check(L.mmx_rescore_f64(
            ctypes.byref(vol_exact), d_blocks.data_ptr(), nb, table.data_ptr(), cap,
            bufs.count.data_ptr(), d_w0.data_ptr(), d_w2.data_ptr(), nat.as_int32_ptr(space.radii),
            nat.as_double_ptr(space.norms), ns, store_f32, stream), "mmx_rescore_f64")
        count = int(bufs.count.item()) & 0xFFFFFFFF
        if count <= cap:
            break
        if count >= n_vox * ns:
            # every voxel of every block "equals its maximum": only possible for constant
            # cubes, which scikit-image treats as having no peaks (peak.py:41-43)
            count = 0
            break
        cap = count + 1024
    cands = (table[:count * nat.CAND_DTYPE.itemsize].cpu().numpy().view(nat.CAND_DTYPE)
             if count else np.zeros(0, dtype=nat.CAND_DTYPE))
    stats.n_blocks += nb
    stats.n_voxels += n_vox
    stats.n_candidates += count
    if count:
        err = float(np.max(np.abs(cands["v"].astype(np.float64) - cands["v64"])))
        stats.max_f32_error = max(stats.max_f32_error, err)
        if not err < 0.25 * eps:
            raise nat.MmxError(
                f"float32 LoG deviates from the exact value by {err:.3g} (band {eps:.3g}): "
                "refusing to decide peaks on it")
    return _resolve_peaks(cands, blocks, shapes, ns, thr, dvol, vol_exact, d_blocks, d_w0, d_w2,
                          space, store_f32, stats)


def _resolve_peaks(cands, blocks, shapes, ns, thr, dvol, vol_exact, d_blocks, d_w0, d_w2,
                   space, store_f32, stats):
    """Exact peak membership + the reference's ordering, per block."""
    L = nat.lib()
    dev = dvol.tensor.device
    nb = len(blocks)
    keep = np.ones(len(cands), dtype=bool)
    contested = np.nonzero(cands["flags"] & nat.MMX_CAND_CONTESTED)[0]
    stats.n_contested += len(contested)
    if len(contested):
        # exact values of the (up to) 80 neighbours of every contested candidate
        offs = np.array([(ds, dz, dy, dx) for ds in (-1, 0, 1) for dz in (-1, 0, 1)
                         for dy in (-1, 0, 1) for dx in (-1, 0, 1)
                         if (ds, dz, dy, dx) != (0, 0, 0, 0)], dtype=np.int32)
        c = cands[contested]
        dims = np.array([shapes[i] for i in c["slot"]], dtype=np.int32)      # (m, 3)
        ss = c["s"][:, None] + offs[None, :, 0]
        zz = c["z"][:, None] + offs[None, :, 1]
        yy = c["y"][:, None] + offs[None, :, 2]
        xx = c["x"][:, None] + offs[None, :, 3]
        inside = ((ss >= 0) & (ss < ns) & (zz >= 0) & (zz < dims[:, 0:1]) &
                  (yy >= 0) & (yy < dims[:, 1:2]) & (xx >= 0) & (xx < dims[:, 2:3]))
        owner, which = np.nonzero(inside)
        probes = np.zeros(len(owner), dtype=nat.CAND_DTYPE)
        probes["slot"] = c["slot"][owner]
        probes["s"] = ss[owner, which]
        probes["z"] = zz[owner, which]
        probes["y"] = yy[owner, which]
        probes["x"] = xx[owner, which]
        probes["v64"] = np.nan
        stats.n_probes += len(probes)
        if len(probes):
            d_probes = _to_device_bytes(probes, dev)
            nat.check(L.mmx_rescore_f64(
                ctypes.byref(vol_exact), d_blocks.data_ptr(), nb, d_probes.data_ptr(), len(probes),
                None, d_w0.data_ptr(), d_w2.data_ptr(), nat.as_int32_ptr(space.radii),
                nat.as_double_ptr(space.norms), ns, store_f32, _stream_ptr()), "mmx_rescore_f64")
            vals = d_probes.cpu().numpy().view(nat.CAND_DTYPE)["v64"]
        else:
            vals = np.zeros(0)
        nbr_max = np.full(len(contested), -np.inf)
        np.maximum.at(nbr_max, owner, vals)
        border = ~inside.all(axis=1)
        nbr_max[border] = np.maximum(nbr_max[border], 0.0)   # mode='constant', cval=0
        keep[contested] = c["v64"] >= nbr_max
    keep &= cands["v64"] > thr
    cands = cands[keep]
    out = []
    for i, shp in enumerate(shapes):
        mine = cands[cands["slot"] == i]
        if len(mine) == 0:
            out.append((np.zeros((0, 4), dtype=np.int64), np.zeros(0)))
            continue
        # C order of np.nonzero on the (z, y, x, sigma) cube, then argsort(-values)
        lin = ((mine["z"].astype(np.int64) * shp[1] + mine["y"]) * shp[2] + mine["x"]) * ns + mine["s"]
        mine = mine[np.argsort(lin, kind="stable")]
        vals = mine["v64"].copy()
        order = np.argsort(-vals)
        coords = np.stack([mine["z"], mine["y"], mine["x"], mine["s"]], axis=1).astype(np.int64)[order]
        out.append((coords, vals[order]))
        stats.n_peaks += len(order)
    return out


# ------------------------------------------------------------------------------ A5
def _exact_overlap(b1: np.ndarray, b2: np.ndarray) -> float:
    """``_blob_overlap`` with the reference's exact libm calls, for the knife-edge pairs
    (skimage/feature/blob.py:84-143, 3-D branch :55-81)."""
    root = math.sqrt(3)
    if b1[-1] == b2[-1] == 0:
        return 0.0
    if b1[-1] > b2[-1]:
        ms, r1, r2 = b1[-1:], 1, b2[-1] / b1[-1]
    else:
        ms, r2, r1 = b2[-1:], 1, b1[-1] / b2[-1]
    p1 = b1[:3] / (ms * root)
    p2 = b2[:3] / (ms * root)
    d = np.sqrt(np.sum((p2 - p1) ** 2))
    if d > r1 + r2:
        return 0.0
    if d <= abs(r1 - r2):
        return 1.0
    vol = (math.pi / (12 * d) * (r1 + r2 - d) ** 2 *
           (d ** 2 + 2 * d * (r1 + r2) - 3 * (r1 ** 2 + r2 ** 2) + 6 * r1 * r2))
    return vol / (4. / 3 * math.pi * min(r1, r2) ** 3)


def _reference_pair_order(lm: np.ndarray) -> np.ndarray:
    """The visiting order ``_prune_blobs`` uses (blob.py:169-172): iteration order of the
    Python ``set`` returned by SciPy's ``cKDTree.query_pairs``.  It is implementation
    defined, so when the outcome depends on it the only faithful source is the same call."""
    from scipy import spatial
    sigma = lm[:, -1].max()
    distance = 2 * sigma * math.sqrt(lm.shape[1] - 1)
    tree = spatial.cKDTree(lm[:, :-1])
    return np.array(list(tree.query_pairs(distance)))


def _prune_batch(peaks, space: ScaleSpace, overlap: float, dev, stats: BatchStats):
    """Sphere-overlap prune of every block of the batch."""
    L = nat.lib()
    lms = []
    for coords, _vals in peaks:
        if len(coords) == 0:
            lms.append(None)
            continue
        lm = coords.astype(np.float64)
        lm[:, 3] = space.sigmas[coords[:, 3]]
        lms.append(lm)
    sizes = np.array([0 if lm is None else len(lm) for lm in lms], dtype=np.int32)
    offsets = np.zeros(len(lms) + 1, dtype=np.int32)
    np.cumsum(sizes, out=offsets[1:])
    total = int(offsets[-1])
    results = []
    pairs = np.zeros((0, 2), dtype=np.int32)
    frac = np.zeros(0)
    if total:
        allb = np.concatenate([lm for lm in lms if lm is not None])
        d_blobs = torch.from_numpy(allb).to(dev)
        d_off = torch.from_numpy(offsets).to(dev)
        cap = max(1024, 4 * total)
        while True:
            d_pairs = torch.empty((cap, 2), dtype=torch.int32, device=dev)
            d_frac = torch.empty(cap, dtype=torch.float64, device=dev)
            d_count = torch.zeros(1, dtype=torch.int32, device=dev)
            nat.check(L.mmx_overlap_pairs(d_blobs.data_ptr(), d_off.data_ptr(), len(lms), overlap,
                                          OVERLAP_BAND, d_pairs.data_ptr(), d_frac.data_ptr(), cap,
                                          d_count.data_ptr(), _stream_ptr()), "mmx_overlap_pairs")
            n = int(d_count.item()) & 0xFFFFFFFF
            if n <= cap:
                break
            cap = n + 64
        if n:
            pairs = d_pairs[:n].cpu().numpy()
            frac = d_frac[:n].cpu().numpy()
    stats.n_overlap_pairs += len(pairs)
    owner = np.searchsorted(offsets, pairs[:, 0], side="right") - 1 if len(pairs) else np.zeros(0, int)
    for b, lm in enumerate(lms):
        if lm is None:
            results.append(np.empty((0, 3)))
            continue
        mine = owner == b
        if not mine.any():
            results.append(lm)
            stats.n_blobs += len(lm)
            continue
        loc = pairs[mine] - offsets[b]
        fr = frac[mine].copy()
        edge = np.abs(fr - overlap) <= OVERLAP_BAND
        for k in np.nonzero(edge)[0]:
            fr[k] = _exact_overlap(lm[loc[k, 0]], lm[loc[k, 1]])
        act = loc[fr > overlap]
        sig = lm[:, 3].copy()
        if len(act):
            uses = np.bincount(act.ravel(), minlength=len(lm))
            if uses.max() > 1:
                # chains: the result depends on the visiting order -> take the reference's
                stats.n_order_fallbacks += 1
                active = {(int(i), int(j)) for i, j in act}
                for i, j in _reference_pair_order(lm):
                    i, j = int(i), int(j)
                    if (i, j) in active and sig[i] > 0 and sig[j] > 0:
                        if sig[i] > sig[j]:
                            sig[j] = 0
                        else:
                            sig[i] = 0
            else:
                i, j = act[:, 0], act[:, 1]
                first_bigger = sig[i] > sig[j]
                sig[j[first_bigger]] = 0
                sig[i[~first_bigger]] = 0
        res = lm[sig > 0]
        results.append(res)
        stats.n_blobs += len(res)
    return results


def blob_log(image, min_sigma=1, max_sigma=50, num_sigma=10, threshold=.2, overlap=.5):
    """``skimage.feature.blob_log`` signature for one 3-D image (host array or tensor)."""
    dvol = image if isinstance(image, DeviceVolume) else DeviceVolume(image)
    if dvol.multichannel:
        raise ValueError("blob_log takes a single-channel (z, y, x) image")
    return blob_log_blocks(dvol, 0, [(0, 0, 0)], [dvol.shape[:3]], min_sigma, max_sigma, num_sigma,
                           threshold, overlap)[0]
