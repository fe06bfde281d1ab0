"""Host side of a batch: the decisions on the re-scored candidates (A4: ``_resolve_peaks`` / ``_resolve_peaks_native``) and the
per-block sphere-overlap prune (A5), with the reference's own pair order where the outcome depends on it.
Split out of ``blob_log.py`` (round 5)."""
from __future__ import annotations

import ctypes
import math
from typing import Tuple

import numpy as np

from . import _native as nat

try:
    import torch
except Exception as exc:  # pragma: no cover
    raise ImportError("magellanmapper_amd needs PyTorch-ROCm for device memory and streams") from exc
try:
    # A declared dependency of the product path (not of the kernels): when two overlapping blobs of a block
    # each win one pair and lose another, scikit-image's outcome depends on the order in which
    # ``cKDTree.query_pairs`` returns the pairs (skimage/feature/blob.py:169-172) -- implementation defined, so
    # the only faithful source is the same call (``_reference_pair_order``; 13 of the 256 benchmark blocks).
    # Fixtures were made with SciPy 1.7.1, the reference pins 1.15.3 (envs/requirements.txt:47); both agree on
    # every golden case.  SciPy is the reference's own dependency, so it is present wherever this drops in.
    from scipy import spatial as _scipy_spatial
except Exception as exc:  # pragma: no cover
    raise ImportError("magellanmapper_amd needs SciPy (scipy.spatial.cKDTree) for the reference's pair "
                      "order in chained overlap prunes") from exc

from .volume import _stream_ptr
from .buffers import _to_device_bytes

#: band around the overlap limit inside which the host re-evaluates the fraction exactly
OVERLAP_BAND = 1e-9
#: which scikit-image the order of EXACTLY EQUAL peak values follows: "0.18" (default; what every fixture of this
#: repository was made with: `argsort(-values)`, NumPy's default sort -- its order of equal keys is its own) or "0.19+"
#: (`argsort(..., kind="stable")`, scikit-image >= 0.19, incl. the reference's pinned 0.25.2, envs/requirements.txt:46:
#: equal values keep np.nonzero order; its peak mask pads with 'nearest' instead of zeros, which is the same mask for the
#: non-negative thresholds of this path and is refused for negative ones).  PARITY UNPINNED for "0.19+": no fixture of
#: the pinned release can be made in this image; it differs from the default only in blocks that
#: hold two bit-equal float64 responses.
PEAK_ORDER = "0.18"


def _pipeline():
    from . import blob_log
    return blob_log


class _BandTooNarrow(Exception):
    """The float32 values of a batch deviate from the exact ones by more than a quarter of the nomination
    band: the batch is nominated again with a wider band (``_finish_detect``)."""

    def __init__(self, err: float):
        super().__init__(err)
        self.err = err


def _check_f32_error(v32, v64, eps, stats):
    """Every candidate the reference would find is nominated as long as |float32 - float64| < eps / 4 (a true
    maximum then stays within eps of its float32 neighbours and of the threshold).  A larger deviation -- a float
    image with a huge dynamic range, say -- is not fatal: the caller widens the band and nominates again."""
    if len(v32):
        err = float(np.max(np.abs(v32.astype(np.float64) - v64)))
        if not np.isfinite(err):
            raise nat.MmxError("non-finite LoG values: the image holds NaN or infinite voxels")
        if not err < 0.25 * eps:
            raise _BandTooNarrow(err)
        stats.max_f32_error = max(stats.max_f32_error, err)


def _resolve_peaks(cands, blocks, shapes, ns, thr, dvol, vol_exact, d_blocks, d_w0, d_w2,
                   space, store_f32, stats, eps, exact):
    """Exact peak membership + the reference's ordering, per block.

    What the reference's float64 values decide: (i) whether a candidate within ``eps`` of the threshold or of
    a neighbour is a peak -- the NMS kernel flags those CONTESTED; (ii) the order of the peaks of a block
    (``argsort(-value)``, peak.py:17), which only two candidates whose float32 values lie within ``eps`` of
    each other can swap.  Those candidates (and the neighbours of the contested ones) are re-scored in
    float64 here; every other candidate keeps its float32 value as a stand-in, which leaves every comparison
    the reference makes unchanged (|float32 - float64| < eps / 4 is checked on all re-scored values).
    ``exact``: all candidates were re-scored by ``_enqueue_detect`` already.
    """
    L = nat.lib()
    dev = dvol.tensor.device
    nb = len(blocks)
    keep = np.ones(len(cands), dtype=bool)
    contested = np.nonzero(cands["flags"] & nat.MMX_CAND_CONTESTED)[0]
    v32 = cands["v"].astype(np.float64)
    if exact:
        _check_f32_error(cands["v"], cands["v64"], eps, stats)      # (raises before any counter moves)
        selves = np.zeros(0, dtype=np.int64)
    stats.n_contested += len(contested)
    if exact:
        pass
    else:
        # candidates of one block whose float32 values are within eps of each other: their order is open
        cslot = cands["slot"].astype(np.int64)
        o = np.lexsort((-v32, cslot))
        close = (cslot[o][1:] == cslot[o][:-1]) & ((v32[o][:-1] - v32[o][1:]) < eps)
        need = np.zeros(len(cands), dtype=bool)
        need[o[1:][close]] = True
        need[o[:-1][close]] = True
        need[contested] = True
        selves = np.nonzero(need)[0]
        stats.n_rescored += len(selves)
        cands = cands.copy() if not cands.flags.writeable else cands
        cands["v64"] = v32                      # stand-ins; the re-scored ones are overwritten below
    if len(contested) or len(selves):
        # exact values of the (up to) 80 neighbours of every contested candidate
        offs = np.array([(ds, dz, dy, dx) for ds in (-1, 0, 1) for dz in (-1, 0, 1)
                         for dy in (-1, 0, 1) for dx in (-1, 0, 1)
                         if (ds, dz, dy, dx) != (0, 0, 0, 0)], dtype=np.int32)
        c = cands[contested]
        dims = np.array([shapes[i] for i in c["slot"]], dtype=np.int32).reshape(-1, 3)      # (m, 3)
        ss = c["s"][:, None] + offs[None, :, 0]
        zz = c["z"][:, None] + offs[None, :, 1]
        yy = c["y"][:, None] + offs[None, :, 2]
        xx = c["x"][:, None] + offs[None, :, 3]
        inside = ((ss >= 0) & (ss < ns) & (zz >= 0) & (zz < dims[:, 0:1]) &
                  (yy >= 0) & (yy < dims[:, 1:2]) & (xx >= 0) & (xx < dims[:, 2:3]))
        probe = inside
        if len(c) and np.all(c["flags"] & nat.MMX_CAND_BAND):
            # the sparse NMS kernel recorded which neighbours have float32 values within eps below the candidate's
            # (or above it): with |float32 - float64| < eps / 4 every other neighbour is below it in float64 too
            bits = np.arange(64, dtype=np.uint64)
            in_band = np.concatenate([(c["band"][:, None] >> bits[None, :]) & np.uint64(1),
                                      ((c["flags"][:, None] >> np.arange(16, 32, dtype=np.uint32)[None, :]) & 1)
                                      .astype(np.uint64)], axis=1).astype(bool)
            probe = inside & in_band
        owner, which = np.nonzero(probe)
        n_nb = len(owner)
        probes = np.zeros(n_nb + len(selves), dtype=nat.CAND_DTYPE)
        probes["slot"][:n_nb] = c["slot"][owner]
        probes["s"][:n_nb] = ss[owner, which]
        probes["z"][:n_nb] = zz[owner, which]
        probes["y"][:n_nb] = yy[owner, which]
        probes["x"][:n_nb] = xx[owner, which]
        for f in ("slot", "s", "z", "y", "x"):
            probes[f][n_nb:] = cands[f][selves]
        probes["v64"] = np.nan
        stats.n_probes += n_nb
        if len(probes):
            d_probes = _to_device_bytes(probes, dev)
            nat.check(L.mmx_rescore_f64(
                ctypes.byref(vol_exact), d_blocks.data_ptr(), nb, d_probes.data_ptr(), len(probes),
                None, d_w0.data_ptr(), d_w2.data_ptr(), nat.as_int32_ptr(space.radii),
                nat.as_double_ptr(space.norms), ns, store_f32, _stream_ptr()), "mmx_rescore_f64")
            vals = d_probes.cpu().numpy().view(nat.CAND_DTYPE)["v64"]
        else:
            vals = np.zeros(0)
        if len(selves):
            _check_f32_error(cands["v"][selves], vals[n_nb:], eps, stats)
            cands["v64"][selves] = vals[n_nb:]
            vals = vals[:n_nb]
        nbr_max = np.full(len(contested), -np.inf)
        np.maximum.at(nbr_max, owner, vals)
        border = ~inside.all(axis=1)
        nbr_max[border] = np.maximum(nbr_max[border], 0.0)   # mode='constant', cval=0
        keep[contested] = cands["v64"][contested] >= nbr_max
    keep &= cands["v64"] > thr
    sel = np.nonzero(keep)[0]
    # plain contiguous columns from here on (record-array field access walks 48-byte strides): the five leading
    # int32 fields (slot, s, z, y, x) in one gather
    ints = np.ascontiguousarray(cands).view(np.int32).reshape(-1, nat.CAND_DTYPE.itemsize // 4)[sel, :5].astype(np.int64)
    slot, cs, cz, cy, cx = (ints[:, j] for j in range(5))
    v64 = cands["v64"][sel]
    # group by block; inside a block the C order of np.nonzero on the (z, y, x, sigma) cube: one sort on one key
    # (a voxel of a block appears once, so the key is unique)
    dims = np.asarray(shapes, dtype=np.int64)
    lin = ((cz * dims[slot, 1] + cy) * dims[slot, 2] + cx) * ns + cs
    span = int(np.max(dims[:, 0] * dims[:, 1] * dims[:, 2])) * ns
    order = np.argsort(slot * span + lin, kind="stable") if span * len(shapes) < (1 << 62) else np.lexsort((lin, slot))
    slot = slot[order]
    coords_all = ints[order][:, [2, 3, 4, 1]]
    vals_all = v64[order]
    bounds = np.searchsorted(slot, np.arange(len(shapes) + 1))
    out = []
    for i in range(len(shapes)):
        a, b = bounds[i], bounds[i + 1]
        if a == b:
            out.append((np.zeros((0, 4), dtype=np.int64), np.zeros(0)))
            continue
        cube_size = int(shapes[i][0]) * int(shapes[i][1]) * int(shapes[i][2]) * ns
        if b - a == cube_size and cube_size > 1:
            # every voxel equals its 3^4 maximum (a constant cube): "no peak for a trivial image"
            # (skimage peak.py:41-43)
            out.append((np.zeros((0, 4), dtype=np.int64), np.zeros(0)))
            continue
        vals = vals_all[a:b].copy()
        # the reference's call on the reference's array (peak.py:17; scikit-image >= 0.19: kind="stable")
        rank = np.argsort(-vals, kind="stable") if PEAK_ORDER == "0.19+" else np.argsort(-vals)
        out.append((coords_all[a:b][rank], vals[rank]))
        stats.n_peaks += b - a
    return out


class PeakBatch:
    """The raw peaks of one batch as the native host code leaves them: ``coords[offsets[b]:offsets[b + 1]]`` are
    block ``b``'s ``[z, y, x, sigma index]`` rows (int32) by descending float64 response ``vals`` -- the
    reference's order (``argsort(-values)`` of the ``np.nonzero`` rows, peak.py:17).  After the overlap prune
    ``alive`` marks the surviving rows and ``sigmas`` maps the last column to the blob's sigma."""
    __slots__ = ("coords", "vals", "offsets", "alive", "sigmas")

    def __init__(self, coords, vals, offsets):
        self.coords, self.vals, self.offsets = coords, vals, offsets
        self.alive = None
        self.sigmas = None

    def __len__(self):
        return len(self.offsets) - 1

    def block(self, b: int) -> Tuple[np.ndarray, np.ndarray]:
        """``(coords int64 (n, 4), values float64 (n,))`` of block ``b``."""
        lo, hi = self.offsets[b], self.offsets[b + 1]
        return self.coords[lo:hi].astype(np.int64), self.vals[lo:hi]

    def blobs(self, b: int) -> np.ndarray:
        """Block ``b``'s pruned ``[z, y, x, sigma]`` rows (float64), ``np.empty((0, 3))`` without peaks."""
        lo, hi = self.offsets[b], self.offsets[b + 1]
        if lo == hi:
            return np.empty((0, 3))
        rows = self.coords[lo:hi][self.alive[lo:hi].view(bool)]
        out = rows.astype(np.float64)
        out[:, 3] = self.sigmas[rows[:, 3]]
        return out


def _resolve_peaks_native(cands, n_cands: int, blocks, ns: int, thr: float, stats: BatchStats, eps: float) -> PeakBatch:
    """``mmx_host_resolve_peaks``: the decisions of ``_resolve_peaks`` for a table whose candidates AND probes
    (``mmx_expand_probes``) were re-scored on the device; blocks whose peaks tie take their order from NumPy."""
    L = nat.lib()
    nb = len(blocks)
    n_total = len(cands)
    nz_coords = np.empty((max(n_cands, 1), 4), dtype=np.int32)
    nz_vals = np.empty(max(n_cands, 1))
    coords = np.empty_like(nz_coords)
    vals = np.empty_like(nz_vals)
    offsets = np.zeros(nb + 1, dtype=np.int32)
    ties = np.zeros(nb, dtype=np.uint8)
    st = np.zeros(4)
    nat.check(L.mmx_host_resolve_peaks(cands.ctypes.data if n_total else None, n_cands, n_total, blocks.ctypes.data,
                                       nb, ns, float(thr), nz_coords.ctypes.data, nz_vals.ctypes.data,
                                       coords.ctypes.data, vals.ctypes.data, offsets.ctypes.data, ties.ctypes.data,
                                       st.ctypes.data), "mmx_host_resolve_peaks")
    err = float(st[2])
    if n_cands:
        if not np.isfinite(err):
            raise nat.MmxError("non-finite LoG values: the image holds NaN or infinite voxels")
        if not err < 0.25 * eps:
            raise _BandTooNarrow(err)                      # (before any counter moves)
        stats.max_f32_error = max(stats.max_f32_error, err)
    stats.n_contested += int(st[0])
    stats.n_probes += n_total - n_cands
    stats.n_peaks += int(st[1])
    if PEAK_ORDER == "0.19+":
        ties = ties[:0]                 # (stable order = descending value, equal values in np.nonzero order: what the native step left)
    for b in np.nonzero(ties)[0]:
        # equal float64 responses inside one block: the reference's order is whatever np.argsort makes of them
        lo, hi = offsets[b], offsets[b + 1]
        rank = np.argsort(-nz_vals[lo:hi])                 # the reference's call on the reference's array (peak.py:17)
        coords[lo:hi] = nz_coords[lo:hi][rank]
        vals[lo:hi] = nz_vals[lo:hi][rank]
    n = int(offsets[-1])
    return PeakBatch(coords[:n], vals[:n], offsets)


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


#: SciPy releases whose ``cKDTree.query_pairs`` + CPython set order are known to reproduce the real ``_prune_blobs`` on
#: the chain-heavy fixture (tests/golden/overlap_prune.npz: made with 1.7.1; 1.15.3 is the reference's pin and this
#: image's).  Any other release is used all the same -- it is the reference's own call -- with one warning.
VERIFIED_SCIPY = ("1.7.1", "1.15.3")
_warned_scipy = False


def _reference_pair_order(lm: np.ndarray) -> np.ndarray:
    """The visiting order ``_prune_blobs`` uses (blob.py:169-172): iteration order of the
    Python ``set`` returned by SciPy's ``cKDTree.query_pairs``.  It is implementation
    defined, so when the outcome depends on it the only faithful source is the same call."""
    global _warned_scipy
    if not _warned_scipy:
        _warned_scipy = True
        import scipy
        if scipy.__version__ not in VERIFIED_SCIPY:
            import warnings
            warnings.warn(f"SciPy {scipy.__version__}: the pair order of cKDTree.query_pairs decides blocks with "
                          f"pruning chains and was verified against the real _prune_blobs for {VERIFIED_SCIPY} only "
                          "(tests/test_host_logic.py::test_overlap_prune_reproduces_scikit_image_on_every_fixture)")
    return np.array(list(_reference_pair_set(lm)))


def _reference_pair_set(lm: np.ndarray):
    """The ``set`` itself (tuples ``(i, j)``, ``i < j``): callers that only need the order of a few known pairs
    filter it while iterating -- a block of 3 000 blobs has ~10 000 pairs, and a Python loop over all of them cost
    the co-localisation run (C5, chains in every other batch of its second channel) 2-6 ms per block."""
    sigma = lm[:, -1].max()
    distance = 2 * sigma * math.sqrt(lm.shape[1] - 1)
    tree = _scipy_spatial.cKDTree(lm[:, :-1])
    return tree.query_pairs(distance)


def _apply_pairs(allb, sig, offsets, pairs, frac, overlap: float, stats: BatchStats, only_blocks=None) -> None:
    """The sequential rule of ``_prune_blobs`` (blob.py:172-186) on the over-limit pairs: ``sig`` of the losers is
    zeroed in place.  ``pairs`` are global rows (i < j) in any order, ``frac`` their overlap fractions; fractions
    within ``OVERLAP_BAND`` of the limit are re-evaluated with the reference's exact libm calls first.
    ``only_blocks``: leave every other block alone (its outcome is already known)."""
    frac = frac.copy()
    for k in np.nonzero(np.abs(frac - overlap) <= OVERLAP_BAND)[0]:   # knife edge: exact libm
        frac[k] = _exact_overlap(allb[pairs[k, 0]], allb[pairs[k, 1]])
    act = pairs[frac > overlap]
    if not len(act):
        return
    i, j = act[:, 0], act[:, 1]
    block_of_pair = np.searchsorted(offsets, i, side="right") - 1
    if only_blocks is not None:
        sel = np.isin(block_of_pair, only_blocks)
        act, i, j, block_of_pair = act[sel], i[sel], j[sel], block_of_pair[sel]
        if not len(act):
            return
    first_bigger = sig[i] > sig[j]
    loser = np.where(first_bigger, j, i)
    winner = np.where(first_bigger, i, j)
    chained = np.intersect1d(loser, winner)
    chain_blocks = np.unique(np.searchsorted(offsets, chained, side="right") - 1)
    simple = ~np.isin(block_of_pair, chain_blocks)
    sig[loser[simple]] = 0
    for b in chain_blocks:
        stats.n_order_fallbacks += 1
        lo, hi = offsets[b], offsets[b + 1]
        mine = block_of_pair == b
        active = {(int(a_) - lo, int(b_) - lo) for a_, b_ in act[mine]}
        bs = sig[lo:hi]
        for a_, b_ in [p for p in _reference_pair_set(allb[lo:hi]) if p in active]:     # (the set's own order)
            if bs[a_] > 0 and bs[b_] > 0:
                if bs[a_] > bs[b_]:
                    bs[b_] = 0
                else:
                    bs[a_] = 0


def _prune_batch_native(pb: PeakBatch, space: ScaleSpace, overlap: float, stats: BatchStats) -> PeakBatch:
    """``mmx_host_overlap_prune`` on the host's own peaks (no upload, no kernel, no wait): ``pb.alive`` per row.
    Blocks whose outcome depends on the order scikit-image visits the pairs in, and batches with a fraction on the
    knife edge, go through :func:`_apply_pairs` with the pairs the native search found."""
    L = nat.lib()
    nb = len(pb)
    n = len(pb.coords)
    pb.sigmas = np.ascontiguousarray(space.sigmas, dtype=np.float64)
    pb.alive = np.ones(n, dtype=np.uint8)
    if n == 0:
        return pb
    open_blocks = np.zeros(nb, dtype=np.uint8)
    cap = max(1024, 4 * n)
    n_pairs, n_knife = ctypes.c_int64(0), ctypes.c_int64(0)
    while True:
        pairs = np.empty((cap, 2), dtype=np.int32)
        frac = np.empty(cap)
        nat.check(L.mmx_host_overlap_prune(pb.coords.ctypes.data, pb.offsets.ctypes.data, nb, pb.sigmas.ctypes.data,
                                           len(pb.sigmas), float(overlap), OVERLAP_BAND, pb.alive.ctypes.data,
                                           open_blocks.ctypes.data, pairs.ctypes.data, frac.ctypes.data, cap,
                                           ctypes.byref(n_pairs), ctypes.byref(n_knife)), "mmx_host_overlap_prune")
        if n_pairs.value <= cap:
            break
        cap = n_pairs.value + 64
    stats.n_overlap_pairs += n_pairs.value
    todo = None if n_knife.value else np.nonzero(open_blocks)[0]
    if todo is None or len(todo):
        allb = pb.coords.astype(np.float64)
        allb[:, 3] = pb.sigmas[pb.coords[:, 3]]
        sig = allb[:, 3].copy()
        if todo is not None:
            sig[pb.alive == 0] = 0         # (the closed blocks' outcome stands)
        _apply_pairs(allb, sig, pb.offsets, pairs[:n_pairs.value].astype(np.int64), frac[:n_pairs.value], overlap,
                     stats, only_blocks=todo)
        pb.alive = (sig > 0).astype(np.uint8)
    stats.n_blobs += int(pb.alive.sum())
    return pb


def _prune_batch(peaks, space: ScaleSpace, overlap: float, dev, stats: BatchStats):
    """Sphere-overlap prune of every block of the batch (skimage blob.py:146-187).

    The device returns every pair whose overlap fraction exceeds the limit.  The reference
    visits pairs one by one and zeroes the smaller sigma (first of the pair on ties); a dead
    blob never kills another.  The outcome is independent of the visiting order unless some
    blob loses one over-limit pair and wins another (a chain): only blocks with such a blob
    take the reference's own order from ``cKDTree.query_pairs``.
    """
    L = nat.lib()
    sizes = np.array([len(c) for c, _ in peaks], dtype=np.int32)
    offsets = np.zeros(len(peaks) + 1, dtype=np.int32)
    np.cumsum(sizes, out=offsets[1:])
    total = int(offsets[-1])
    if total == 0:
        return [np.empty((0, 3)) for _ in peaks]
    coords = np.concatenate([c for c, _ in peaks if len(c)])
    allb = coords.astype(np.float64)
    allb[:, 3] = space.sigmas[coords[:, 3]]
    d_blobs = torch.from_numpy(allb).to(dev)
    d_off = torch.from_numpy(offsets).to(dev)
    cap = max(1024, 4 * total)
    while True:
        d_pairs = torch.empty((cap, 2), dtype=torch.int32, device=dev)
        d_frac = torch.empty(cap, dtype=torch.float64, device=dev)
        d_count = torch.zeros(1, dtype=torch.int32, device=dev)
        nat.check(L.mmx_overlap_pairs(d_blobs.data_ptr(), d_off.data_ptr(), len(peaks), overlap,
                                      OVERLAP_BAND, float(space.sigmas.max()), d_pairs.data_ptr(),
                                      d_frac.data_ptr(), cap,
                                      d_count.data_ptr(), _stream_ptr()), "mmx_overlap_pairs")
        n = int(d_count.item()) & 0xFFFFFFFF
        if n <= cap:
            break
        cap = n + 64
    sig = allb[:, 3].copy()
    stats.n_overlap_pairs += n
    if n:
        _apply_pairs(allb, sig, offsets, d_pairs[:n].cpu().numpy().astype(np.int64), d_frac[:n].cpu().numpy(),
                     overlap, stats)
    results = []
    for b in range(len(peaks)):
        lo, hi = offsets[b], offsets[b + 1]
        if lo == hi:
            results.append(np.empty((0, 3)))
            continue
        res = allb[lo:hi][sig[lo:hi] > 0]
        results.append(res)
        stats.n_blobs += len(res)
    return results


