"""Per-block preprocessing on the device: contrast stretch, unsharp mask, erosion.

Mirror of the sub-sub-block loop of ``StackDetector.detect_sub_roi`` (reference
magmap/cv/stack_detect.py:122-150): every detection block is cut into
``denoise_max_shape`` tiles (``chunking.stack_splitter`` without overlap), each tile goes through
``plot_3d.saturate_roi`` (magmap/plot/plot_3d.py:55-112) and ``plot_3d.denoise_roi`` (:115-172)
on its own, and the float64 results are merged back into the block that ``detect_blobs`` sees.
Here one HIP workgroup does all of that for one tile (``csrc/mmx_preproc.hip``), bit for bit.

Host side of the exactness contract (the parts that are cheaper to state in NumPy than on the
device, all O(number of distinct tile sizes)):

* :func:`quantile_ranks` -- the index arithmetic of ``np.percentile(..., method="linear")``
  (numpy/lib/_function_base_impl.py ``_quantile`` / ``_get_indexes`` / ``_get_gamma``);
* the sigma-8 Gaussian weights, built like ``scipy.ndimage._filters._gaussian_kernel1d``
  (:mod:`kernels1d`).

Scope: uint8 / uint16 voxels (what microscopes write) and float64 images (the same float64 arithmetic once the
order statistics are found); float32 images, whose result in the reference depends on the NumPy release, raise
``NotImplementedError``.  scikit-image < 0.19 treats a 3-D array whose last axis has length 3
as RGB inside ``filters.gaussian``; the reference pins 0.25 (no such guess), :data:`RGB_GUESS`
switches the old behaviour on for the golden fixtures made with 0.18.3.
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

try:
    import torch
except ImportError:  # pragma: no cover
    torch = None

from . import _native as nat
from . import config, kernels1d

GAUSS_SIGMA = 8.0                    # hard-coded in the reference (plot_3d.py:151)
#: scikit-image < 0.19 ``filters.gaussian`` RGB guess for ``(M, N, 3)`` arrays
RGB_GUESS = False
#: test hook: half kernel ``w[0..32]`` to use instead of this NumPy's (``np.exp`` is not bit-stable
#: across NumPy releases and the golden fixtures were made under NumPy 1.26)
GAUSS_WEIGHTS_OVERRIDE: Optional[np.ndarray] = None
#: test hook: send every tile through the global-scratch kernels (sides <= 64: register-line kernel,
#: larger: one output per lane) instead of the LDS kernel; "big" forces the one-output-per-lane kernel
FORCE_GENERIC = False
#: kernel form of the LDS-resident tiles (``nat.MMX_PP_AUTO`` / ``_SINGLE`` / ``_PIPELINED``) and the tiles one
#: workgroup of the pipelined blur kernel walks (0 = the library's default): what tests and tools switch
KERNEL_MODE = nat.MMX_PP_AUTO
TILES_PER_WG = 0


def _wait_upload(dvol, origins, shapes) -> None:
    """A volume still on its way to the device (``DeviceVolume.stream_wait``): the current stream waits for the slabs
    these blocks touch."""
    if getattr(dvol, "_upload", None) is not None:
        dvol.stream_wait(None, None, [(int(o[0]), int(o[0]) + int(s_[0]), int(o[1]), int(o[1]) + int(s_[1]))
                                      for o, s_ in zip(origins, shapes)])


def gauss_weights() -> np.ndarray:
    if GAUSS_WEIGHTS_OVERRIDE is not None:
        return np.ascontiguousarray(GAUSS_WEIGHTS_OVERRIDE, dtype=np.float64)
    return kernels1d.gaussian_half_kernel(GAUSS_SIGMA, 0, kernels1d.kernel_radius(GAUSS_SIGMA))


def quantile_ranks(n: int, pct) -> Tuple[int, int, float]:
    """``(prev, next, gamma)`` of ``np.percentile(a, pct)`` for ``a.size == n`` (method "linear"):
    the result is ``_lerp(sorted(a)[prev], sorted(a)[next], gamma)``."""
    if not 0 <= pct <= 100:
        raise ValueError("Percentiles must be in the range [0, 100]")
    q = np.true_divide(np.asanyarray(pct, dtype=np.float64), 100)
    virt = (n - 1) * q
    prev = np.floor(virt)
    nxt = prev + 1
    if virt >= n - 1:
        prev = nxt = -1.0
    gamma = float(virt - prev)              # NumPy subtracts the *fixed* index (gamma is unused then)
    prev_i, next_i = int(prev), int(nxt)
    if prev_i < 0:
        prev_i = next_i = n - 1
    return prev_i, next_i, gamma


def _tile_grid(shape: Sequence[int], dms: Sequence[int]):
    """Tile origins and extents of one block, C order (chunking.stack_splitter, no overlap)."""
    axes = []
    for n, d in zip(shape, dms):
        starts = np.arange(0, n, d, dtype=np.int64)
        axes.append((starts, np.minimum(d, n - starts)))
    oz, oy, ox = np.meshgrid(axes[0][0], axes[1][0], axes[2][0], indexing="ij")
    ez, ey, ex = np.meshgrid(axes[0][1], axes[1][1], axes[2][1], indexing="ij")
    return (np.stack([oz.ravel(), oy.ravel(), ox.ravel()], axis=1),
            np.stack([ez.ravel(), ey.ravel(), ex.ravel()], axis=1))


def channel_params(chl: int, near_max: Optional[Sequence[float]] = None) -> Tuple[nat.PreprocParams, float, float]:
    """C-ABI parameters of one channel from its ROI profile + ``config.near_max``, read at call
    time like the reference does (plot_3d.py:84-99, 141-163)."""
    settings = config.get_roi_profile(chl)
    # total-variation denoising (plot_3d.py:147-149): `if tot_var_denoise:` then weight = the value itself
    # (True -> 1.0 in NumPy arithmetic); tau / weight as Python evaluates it in the reference's expression
    tv = settings["tot_var_denoise"]
    tv_weight = float(tv) if tv else 0.0
    tv_factor = (1. / (2. * 3)) / tv if tv else 0.0
    near = config.near_max if near_max is None else near_max
    max_thresh = near[chl] * settings["max_thresh_factor"]       # IndexError like the reference
    strength = settings["unsharp_strength"] or 0.0
    ero = settings["erosion_threshold"] or 0.0
    p = nat.PreprocParams(float(settings["clip_min"]), float(settings["clip_max"]), float(max_thresh),
                          float(strength), float(ero), kernels1d.kernel_radius(GAUSS_SIGMA),
                          1 if RGB_GUESS else 0, tv_weight, float(tv_factor))
    return p, settings["clip_vmin"], settings["clip_vmax"]


#: output buffer sets a preprocessing stage cycles through (``run(..., which)``): batch k + 2 may be preprocessed while
#: batch k + 1 waits for its LoG passes and batch k's results are still being re-scored (``blob_log``: two batches ahead)
N_BUFFER_SETS = 3
#: bytes per block voxel of one stage's output slots: a float32 and a float64 copy in every buffer set
_SLOT_BYTES = (4 + 8) * N_BUFFER_SETS
#: device -> {"f64_<channel>" / "f32": tensor}: the resident preprocessed blocks of `Preprocessor.retain`
_RETAINED: Dict[str, Dict[str, "torch.Tensor"]] = {}


def release_retained() -> None:
    """Drop the preprocessed blocks `Preprocessor.retain` keeps between calls (tens of GB for a whole tile)."""
    _RETAINED.clear()


class Preprocessor:
    """Preprocesses the blocks of a batch into uniform-stride float32 / float64 slots."""

    def __init__(self, denoise_max_shape: Sequence[int], near_max: Optional[Sequence[float]] = None,
                 want_info: bool = False):
        dms = [int(v) for v in denoise_max_shape]
        if len(dms) != 3 or min(dms) < 1:
            raise ValueError("denoise_max_shape must be three positive sizes")
        self.dms = dms
        self.near_max = near_max
        self.want_info = want_info
        self._out64 = [None] * N_BUFFER_SETS
        self._out32 = [None] * N_BUFFER_SETS
        self._scratch = None
        self._info = [None] * N_BUFFER_SETS
        self._work = None
        self._last_use = None           # (stream, event) of the most recent run: orders the reuse of `_work` / `_info`
        self._weights = None
        self._tmpl_key = None
        self._tmpl, self._qc_rows, self._qc_index = {}, [], {}
        self._stage = [None] * N_BUFFER_SETS              # pinned staging of the tile tables, one per buffer set
        self._stage_free = [None] * N_BUFFER_SETS
        self._tiles: Dict[Tuple[int, int, int], Tuple[np.ndarray, np.ndarray]] = {}
        self.last_info: Optional[np.ndarray] = None
        self.last_subs: Optional[np.ndarray] = None
        self._retain = None

    def retain(self, dvol, origins, shapes, channels, budget_fraction: float = 0.45) -> bool:
        """Keep the preprocessed float64 blocks of ``channels`` resident for the whole call (every block its own slot)
        instead of recycling two batch-sized buffers: the intensity co-localisation, which the reference runs on the
        image detection saw (stack_detect.py:159-162), then reads them instead of preprocessing every channel a
        second time.  MI355X's 288 GB make that affordable for whole tiles (2 channels x 128 blocks of 261^3:
        40 GB + 20 GB of float32); returns False -- nothing kept, callers preprocess again -- when the free HBM does not."""
        nb = len(shapes)
        if nb == 0:
            return False
        shp = np.asarray(shapes, dtype=np.int64).reshape(nb, 3)
        sx = int(-(-shp[:, 2].max() // nat.MMX_ROW_ALIGN) * nat.MMX_ROW_ALIGN)
        sy_rows = int(shp[:, 1].max())
        slot = sx * sy_rows * int(shp[:, 0].max())
        need = nb * slot * (8 + 4) * len(channels)
        free_b, _ = torch.cuda.mem_get_info(dvol.tensor.device)
        dev = dvol.tensor.device
        held = sum(t.numel() * t.element_size() for t in _RETAINED.get(str(dev), {}).values() if t is not None)
        if need > budget_fraction * (free_b + held):        # (what an earlier call left here is ours to reuse)
            return False
        # The buffers outlive this object: a stack detected step after step (or channel group after channel group)
        # takes the same tens of GB every time, and handing them back to the caching allocator in between lets smaller
        # requests carve pieces out of them -- the next call then pays a fresh hipMalloc of 50 GB (measured: 160 ms of
        # a 305 ms step of the two-channel benchmark tile).  `release_retained()` drops them.
        pool = _RETAINED.setdefault(str(dev), {})

        def take(name, dtype):
            t = pool.get(name)
            if t is None or t.numel() < nb * slot or t.dtype != dtype:
                pool[name] = None
                t = pool[name] = torch.empty(nb * slot, dtype=dtype, device=dev)
            return t[:nb * slot]
        wanted = {f"f64_{int(c)}" for c in channels} | {f"f32_{int(c)}" for c in channels}
        for name in [k for k in pool if k not in wanted]:
            del pool[name]                  # (channels of an earlier call that this one does not keep)
        # (a float32 copy per channel too: with the channels of a batch detected back to back -- blob_log.blob_log_lanes
        #  -- channel 1's preprocessing runs ahead, beside channel 0's LoG passes, and one shared copy would be overwritten
        #  under them)
        self._retain = dict(
            gid={(tuple(int(v) for v in o), tuple(int(v) for v in s_)): i for i, (o, s_) in enumerate(zip(origins, shapes))},
            sx=sx, sy_rows=sy_rows, slot=slot, nb=nb,
            out64={int(c): take(f"f64_{int(c)}", torch.float64) for c in channels},
            out32={int(c): take(f"f32_{int(c)}", torch.float32) for c in channels})
        return True

    def retains(self, channel: int) -> bool:
        """Whether the preprocessed blocks of ``channel`` go to slots of their own (``retain``) instead of the buffer sets."""
        return self._retain is not None and int(channel) in self._retain["out64"]

    def retained_view(self, channel: int, origins, shapes):
        """``(blocks, vol64)`` of already preprocessed blocks of ``channel`` (``None`` when nothing is retained)."""
        r = self._retain
        if r is None or int(channel) not in r["out64"]:
            return None
        blocks = np.zeros(len(shapes), dtype=nat.BLOCK_DTYPE)
        for i, (o, s_) in enumerate(zip(origins, shapes)):
            g = r["gid"].get((tuple(int(v) for v in o), tuple(int(v) for v in s_)))
            if g is None:
                return None
            px = -(-int(s_[2]) // nat.MMX_ROW_ALIGN) * nat.MMX_ROW_ALIGN
            blocks[i] = (g * r["slot"], s_[0], s_[1], s_[2], i, px, 0)
        dst_sy, dst_sz = r["sx"], r["sx"] * r["sy_rows"]
        return blocks, nat.Volume(r["out64"][int(channel)].data_ptr(), nat.MMX_F64, 0, dst_sz, dst_sy, 1)

    # ---- geometry
    @staticmethod
    def bytes_per_voxel() -> int:
        """Extra HBM per block voxel: a float32 + a float64 slot in each of the ``N_BUFFER_SETS`` buffer sets, the
        tile-major uint16 copy of the pipelined kernels (+ the 7-double scratch of the total-variation iteration when
        any profile switches it on)."""
        tv = any(p["tot_var_denoise"] for p in [config.roi_profile, *config.roi_profiles] if p)
        return _SLOT_BYTES + 2 + (7 * 8 if tv else 0)

    def value_scale(self, channels: Sequence[int]) -> float:
        """Bound on |preprocessed voxel|: den + (den - s*blur) with den, blur in [clip_min, clip_max]."""
        m = 1.0
        for chl in channels:
            s = config.get_roi_profile(chl)
            c = max(abs(float(s["clip_min"])), abs(float(s["clip_max"])))
            m = max(m, 2 * c + abs(float(s["unsharp_strength"] or 0.0)) * c)
        return m

    def value_range(self, channels: Sequence[int]):
        """``(lo, hi)`` bounds on the preprocessed voxels: ``clip`` puts them into [clip_min, clip_max], the unsharp
        mask ``den + (den - s * blur)`` (blur a convex mean of ``den``) into [2 clip_min - s clip_max, 2 clip_max -
        s clip_min], the erosion takes minima.  ``None`` with total-variation denoising on (its iterate may leave the
        clipped range by a rounding)."""
        lo, hi = np.inf, -np.inf
        for chl in channels:
            s = config.get_roi_profile(chl)
            if s["tot_var_denoise"]:
                return None
            c_lo, c_hi = float(s["clip_min"]), float(s["clip_max"])
            k = float(s["unsharp_strength"] or 0.0)
            if k:
                c_lo, c_hi = 2 * c_lo - max(k * c_hi, k * c_lo), 2 * c_hi - min(k * c_lo, k * c_hi)
            lo, hi = min(lo, c_lo), max(hi, c_hi)
        return (float(lo), float(hi)) if lo <= hi else None

    def _grid(self, shape):
        shape = tuple(int(v) for v in shape)
        if shape not in self._tiles:
            self._tiles[shape] = _tile_grid(shape, self.dms)
        return self._tiles[shape]

    def _template(self, shape, vstrides, dstrides, pct_lo, pct_hi, force=None):
        """``(fast, mid, big)`` sub-block tables of one block shape with block-relative offsets."""
        force = FORCE_GENERIC if force is None else force
        tkey = (shape, vstrides, dstrides)
        hit = self._tmpl.get(tkey)
        if hit is not None:
            return hit
        L = nat.lib()
        o, e = self._grid(shape)
        t = np.zeros(len(o), dtype=nat.SUBBLOCK_DTYPE)
        t["src_off"] = o[:, 0] * vstrides[0] + o[:, 1] * vstrides[1] + o[:, 2] * vstrides[2]
        t["dst_off"] = o[:, 0] * dstrides[0] + o[:, 1] * dstrides[1] + o[:, 2]
        t["nz"], t["ny"], t["nx"] = e[:, 0], e[:, 1], e[:, 2]
        uniq, inv = np.unique(e, axis=0, return_inverse=True)
        inv = inv.reshape(-1)
        cls = np.zeros(len(uniq), dtype=np.int32)
        fast = np.zeros(len(uniq), dtype=bool)
        for j, u in enumerate(uniq):
            n = int(u[0]) * int(u[1]) * int(u[2])
            if n not in self._qc_index:
                lp, ln, lg = quantile_ranks(n, pct_lo)
                hp, hn, hg = quantile_ranks(n, pct_hi)
                self._qc_index[n] = len(self._qc_rows)
                self._qc_rows.append((lp, ln, hp, hn, lg, hg))
            cls[j] = self._qc_index[n]
            fast[j] = (not force) and L.mmx_preprocess_fast_lds(int(u[0]), int(u[1]), int(u[2])) != 0
        t["qclass"] = cls[inv]
        is_fast = fast[inv]
        # tiles that do not fit LDS: every side <= 64 -> register-line kernel over a global scratch,
        # anything larger -> one output per lane (both behind mmx_preprocess_batch_generic)
        is_mid = ~is_fast & (e.max(axis=1) <= 64) & (force != "big")
        hit = (t[is_fast], t[is_mid], t[~is_fast & ~is_mid])
        self._tmpl[tkey] = hit
        return hit

    def _buffer(self, name, which, n_elems, dtype, dev):
        cur = getattr(self, name)
        t = cur if which is None else cur[which]
        if t is None or t.numel() < n_elems or t.device != dev:
            t = torch.empty(int(n_elems), dtype=dtype, device=dev)
            if which is None:
                setattr(self, name, t)
            else:
                cur[which] = t
        return t

    def run(self, dvol, channel: int, origins, shapes, which: int = 0):
        """Enqueue the preprocessing of the given blocks of channel ``channel`` on the current
        stream.  Returns ``(blocks, slot_elems, vol32, vol64)``: the ``mmx_block`` table whose
        ``src_off`` point into the preprocessed slots, the LoG workspace slot size and the two
        ``mmx_volume`` views (float32 for the passes, float64 for the exact re-score)."""
        _wait_upload(dvol, origins, shapes)
        L = nat.lib()
        dev = dvol.tensor.device
        # float64 images: the reference's arithmetic is the same float64 arithmetic once np.percentile has found its
        # order statistics (a radix select on the doubles, in the one-output-per-lane kernel).  float32 images: what the
        # reference computes depends on the NumPy release (float32 throughout under 1.26, float64 after the clip under
        # 2.2: DESIGN.md section 6) -- not built.
        f64_voxels = dvol.np_dtype == np.dtype(np.float64)
        if dvol.np_dtype not in (np.dtype(np.uint8), np.dtype(np.uint16), np.dtype(np.float64)):
            raise NotImplementedError(
                f"device preprocessing reads uint8 / uint16 / float64 voxels, not {dvol.np_dtype}")
        params, pct_lo, pct_hi = channel_params(int(channel), self.near_max)
        nb = len(shapes)
        shp = np.asarray(shapes, dtype=np.int64).reshape(nb, 3)
        org = np.asarray(origins, dtype=np.int64).reshape(nb, 3)
        for ax in range(3):
            if (org[:, ax] < 0).any() or (shp[:, ax] < 1).any() or (org[:, ax] + shp[:, ax] > dvol.shape[ax]).any():
                raise ValueError("block outside the volume")
        # uniform slot geometry: rows 128-byte aligned like the LoG workspace
        sx = int(-(-shp[:, 2].max() // nat.MMX_ROW_ALIGN) * nat.MMX_ROW_ALIGN)
        sy_rows = int(shp[:, 1].max())
        dst_sy, dst_sz = sx, sx * sy_rows
        slot_pre = dst_sz * int(shp[:, 0].max())
        # retained blocks (retain()): every block of the call has its own slot, batch after batch
        slots_of = np.arange(nb, dtype=np.int64)
        kept = self._retain if (self._retain is not None and int(channel) in self._retain["out64"]) else None
        if kept is not None:
            try:
                slots_of = np.array([kept["gid"][(tuple(int(v) for v in o), tuple(int(v) for v in s_))]
                                     for o, s_ in zip(org, shp)], dtype=np.int64)
                sx, sy_rows, slot_pre = kept["sx"], kept["sy_rows"], kept["slot"]
                dst_sy, dst_sz = sx, sx * sy_rows
            except KeyError:
                kept = None
        t = dvol.tensor
        vsz, vsy, vsx = (int(v) for v in t.stride()[:3])
        # sub-block table: per distinct block shape a cached template (tile extents, offsets relative
        # to the block, quantile class, fast / generic split); a block only adds its two base offsets
        # total-variation denoising iterates in the one-output-per-lane kernel over a global scratch
        tv_on = params.tv_weight != 0.0
        force = "big" if (tv_on or f64_voxels) else FORCE_GENERIC
        key = (pct_lo, pct_hi, force)
        if self._tmpl_key != key:
            self._tmpl_key, self._tmpl, self._qc_rows, self._qc_index = key, {}, [], {}
        # blocks of one shape share a template: their tables are filled with two broadcast additions,
        # straight into a pinned staging buffer (a 25 um tile of anisotropic data is ~2 000 voxels, so a
        # batch easily has 1e6 tiles)
        # (one staging buffer per buffer set: the copy out of it is queued behind the preprocessing of the batch
        #  before, so with ONE buffer the host waited here until the GPU had reached this batch's predecessor --
        #  never more than a batch ahead, tools/steptrace.py --denoise 25)
        sset = int(which) % N_BUFFER_SETS
        if self._stage_free[sset] is not None:
            self._stage_free[sset].synchronize()             # this set's previous copy out of the staging buffer
        groups: Dict[Tuple[int, int, int], List[int]] = {}
        for i in range(nb):
            groups.setdefault(tuple(int(v) for v in shp[i]), []).append(i)
        plan = []                                            # (class, template, block indices)
        counts = [0, 0, 0]
        for shape_key, members in groups.items():
            tmpls = self._template(shape_key, (vsz, vsy, vsx), (dst_sz, dst_sy), pct_lo, pct_hi, force)
            for cls, tmpl in enumerate(tmpls):
                if len(tmpl):
                    plan.append((cls, tmpl, members))
                    counts[cls] += len(tmpl) * len(members)
        n_fast, n_mid = counts[0], counts[1]
        n_gen = counts[1] + counts[2]
        total = n_fast + n_gen
        item = nat.SUBBLOCK_DTYPE.itemsize
        if self._stage[sset] is None or self._stage[sset].numel() < max(1, total) * item:
            self._stage[sset] = torch.empty(max(1, total) * item * 5 // 4, dtype=torch.uint8).pin_memory()
        stage = self._stage[sset]
        subs = stage.numpy()[:total * item].view(nat.SUBBLOCK_DTYPE)
        at = [0, n_fast, n_fast + n_mid]
        base_src_all = org[:, 0] * vsz + org[:, 1] * vsy + org[:, 2] * vsx
        for cls, tmpl, members in plan:
            m, nt = len(members), len(tmpl)
            view = subs[at[cls]:at[cls] + m * nt].reshape(m, nt)
            view[...] = tmpl[None, :]
            idx = np.asarray(members, dtype=np.int64)
            view["src_off"] += base_src_all[idx][:, None]
            view["dst_off"] += (slots_of[idx] * slot_pre)[:, None]
            at[cls] += m * nt
        qc = np.array(self._qc_rows, dtype=nat.QCLASS_DTYPE)
        if n_gen:
            gen_n = (subs["nz"][n_fast:].astype(np.int64) * subs["ny"][n_fast:] * subs["nx"][n_fast:])
            offs = np.concatenate([[0], np.cumsum((7 if tv_on else 2) * gen_n)])
            subs["scratch_off"][n_fast:] = offs[:-1]
            scratch = self._buffer("_scratch", None, int(offs[-1]), torch.float64, dev)
        if kept is not None:
            out32, out64 = kept["out32"][int(channel)], kept["out64"][int(channel)]
        else:
            out32 = self._buffer("_out32", which, nb * slot_pre, torch.float32, dev)
            out64 = self._buffer("_out64", which, nb * slot_pre, torch.float64, dev)
        d_subs = torch.empty(max(1, total) * item, dtype=torch.uint8, device=dev)
        d_subs[:total * item].copy_(stage[:total * item], non_blocking=True)
        self._stage_free[sset] = torch.cuda.Event()
        self._stage_free[sset].record()
        from . import blob_log as _bl
        d_qc = _bl._to_device_bytes(qc, dev)
        # per-tile records: the statistics kernel hands them to the blur kernel (and `info()` reads them)
        n_info = max(1, len(subs)) * nat.SUBINFO_DTYPE.itemsize
        if self.want_info:
            d_info = torch.zeros(n_info, dtype=torch.uint8, device=dev)
        else:                       # (reused in stream order: every run of this object is queued on one stream)
            d_info = self._buffer("_info", int(which) % N_BUFFER_SETS, n_info, torch.uint8, dev)
        weights = gauss_weights()
        if len(weights) != params.radius + 1:
            raise nat.MmxError("Gaussian half kernel has the wrong length")
        wkey = weights.tobytes()
        if self._weights is None or self._weights[0] != wkey or self._weights[1].device != dev:
            self._weights = (wkey, torch.from_numpy(weights).to(dev))
        d_w = self._weights[1]
        vol = dvol.view(channel, False)
        stream = torch.cuda.current_stream().cuda_stream
        # `_work` and the `_info` sets are reused in STREAM order: a run queued on another stream than the one before it
        # (blob_log.PRE_STREAM toggled between calls, a caller's own stream) first waits for that run's kernels
        last = self._last_use
        if last is not None and last[0] != stream:
            torch.cuda.current_stream().wait_event(last[1])
        item = nat.SUBBLOCK_DTYPE.itemsize
        info_ptr = d_info.data_ptr() if d_info is not None else None
        if n_fast:
            # scratch of the pipelined form (a tile-major uint16 copy of the voxels + per-tile offsets): one buffer,
            # reused in stream order like the per-tile records
            wb = int(L.mmx_preprocess_work_bytes(subs.ctypes.data, n_fast)) if KERNEL_MODE != nat.MMX_PP_SINGLE else 0
            work = self._buffer("_work", None, max(1, wb), torch.uint8, dev)
            nat.check(L.mmx_preprocess_batch_mode(
                ctypes.byref(vol), d_subs.data_ptr(), subs.ctypes.data, n_fast, d_qc.data_ptr(), len(qc),
                ctypes.byref(params), d_w.data_ptr(), dst_sy, dst_sz,
                out32.data_ptr(), out64.data_ptr(), info_ptr, int(KERNEL_MODE), int(TILES_PER_WG),
                work.data_ptr() if wb else None, wb, stream),
                "mmx_preprocess_batch_mode")
        for first, count in ((n_fast, n_mid), (n_fast + n_mid, n_gen - n_mid)):
            if count:      # one call per kernel class: the entry point picks the kernel from the extents
                nat.check(L.mmx_preprocess_batch_generic(
                    ctypes.byref(vol), d_subs.data_ptr() + first * item, subs.ctypes.data + first * item,
                    count, d_qc.data_ptr(), len(qc), ctypes.byref(params), d_w.data_ptr(),
                    dst_sy, dst_sz, out32.data_ptr(), out64.data_ptr(),
                    (info_ptr + first * nat.SUBINFO_DTYPE.itemsize) if info_ptr else None,
                    scratch.data_ptr(), int(scratch.numel()), stream), "mmx_preprocess_batch_generic")
        used = torch.cuda.Event()
        used.record()
        self._last_use = (stream, used)
        self.last_subs = subs.copy() if self.want_info else None
        self._keep = (d_subs, d_qc, d_info)
        if self.want_info:
            self.last_info = d_info[:len(subs) * nat.SUBINFO_DTYPE.itemsize]        # device bytes; see info()
        # block table over the preprocessed slots
        blocks = np.zeros(nb, dtype=nat.BLOCK_DTYPE)
        slot = 1
        for i in range(nb):
            px = -(-int(shp[i, 2]) // nat.MMX_ROW_ALIGN) * nat.MMX_ROW_ALIGN
            blocks[i] = (int(slots_of[i]) * slot_pre, shp[i, 0], shp[i, 1], shp[i, 2], i, px, 0)
            slot = max(slot, int(shp[i, 0]) * int(shp[i, 1]) * px)
        vol32 = nat.Volume(out32.data_ptr(), nat.MMX_F32, 0, dst_sz, dst_sy, 1)
        vol64 = nat.Volume(out64.data_ptr(), nat.MMX_F64, 0, dst_sz, dst_sy, 1)
        self.last_geometry = (slot_pre, dst_sz, dst_sy, out64, out32) if kept is None else None
        return blocks, slot, vol32, vol64

    def info(self) -> np.ndarray:
        """Per-sub-block diagnostics of the last :meth:`run` (``want_info=True``), in ``last_subs`` order."""
        if self.last_info is None:
            raise ValueError("run() was not asked for diagnostics")
        return self.last_info.cpu().numpy().view(nat.SUBINFO_DTYPE)

    def fetch(self, shapes, which: int = 0) -> List[np.ndarray]:
        """The float64 preprocessed blocks of the last :meth:`run` as host arrays."""
        slot_pre, dst_sz, dst_sy, out64, _ = self.last_geometry
        torch.cuda.current_stream().synchronize()
        res = []
        for i, s in enumerate(shapes):
            nz, ny, nx = (int(v) for v in s)
            flat = out64[i * slot_pre:(i + 1) * slot_pre]
            view = torch.as_strided(flat, (nz, ny, nx), (dst_sz, dst_sy, 1))
            res.append(view.cpu().numpy().copy())
        return res


class Unmixer:
    """Spectral unmixing of one detected channel for the blocks of a batch (reference
    magmap/cv/detector.py:910-921): ``x = x - fac * roi[..., k]; x[x < 0] = 0`` for every
    ``(k, fac)`` in turn, in float64, on the block as detection sees it -- the raw voxels, or the
    preprocessed channels when ``denoise_max_shape`` is set (the reference preprocesses the block in
    ``detect_sub_roi`` before ``detect_blobs`` unmixes it).  Same ``run`` contract as
    :class:`Preprocessor`, so ``blob_log_blocks`` takes either."""

    def __init__(self, subtract: Sequence[Tuple[int, float]], denoise_max_shape=None, near_max=None,
                 rescale=None):
        """``rescale = (isotropic factor, channels of the detection)``: the profile's isotropic rescale comes
        first (the reference resizes the whole multichannel block, then unmixes the resized channels:
        detector.py:893-921), so every channel involved goes through its own :class:`Rescaler`."""
        self.subtract = [(int(k), float(f)) for k, f in subtract]
        self.dms = None if denoise_max_shape is None else [int(v) for v in denoise_max_shape]
        self.near_max = near_max
        self.rescale = rescale
        self._rs: Dict[int, "Rescaler"] = {}
        self._blocks_args = None
        self._pres: Dict[int, Preprocessor] = {}
        self._out64 = [None] * N_BUFFER_SETS
        self._out32 = [None] * N_BUFFER_SETS

    def set_blocks(self, origins, shapes, new_shapes) -> None:
        self._blocks_args = (origins, shapes, new_shapes)

    def bytes_per_voxel(self) -> int:
        n_src = 1 + len({k for k, _ in self.subtract})
        pre = _SLOT_BYTES + 2                  # one Preprocessor per source channel (slots + its tile-major copy)
        if self.rescale is not None:
            return _SLOT_BYTES + n_src * (_SLOT_BYTES + (0 if self.dms is None else pre * len(self.rescale[1])))
        return _SLOT_BYTES + (0 if self.dms is None else pre * n_src)

    def value_scale(self, channels: Sequence[int]) -> float:
        if self.dms is not None:
            involved = list(channels) + [k for k, _ in self.subtract]
            return Preprocessor(self.dms).value_scale(involved)
        return self._raw_scale

    def value_range(self, channels: Sequence[int]):
        """The subtraction is clipped at 0: ``[0, largest voxel]`` whatever is subtracted (negative factors add)."""
        if any(f < 0 for _, f in self.subtract):
            return None
        return 0.0, float(self.value_scale(channels))

    _buffer = Preprocessor._buffer

    def run(self, dvol, channel: int, origins, shapes, which: int = 0):
        _wait_upload(dvol, origins, shapes)
        from . import blob_log as bl
        L = nat.lib()
        dev = dvol.tensor.device
        if not dvol.multichannel:
            raise IndexError("spectral unmixing needs a (z, y, x, c) image")
        for k, _ in self.subtract:
            if not 0 <= k < dvol.n_channels:
                raise IndexError(f"index {k} is out of bounds for axis 3 with size {dvol.n_channels}")
        if self.rescale is not None:
            views = {}
            for c in [channel] + [k for k, _ in self.subtract]:
                if c not in views:
                    rs = self._rs.get(c)
                    if rs is None:
                        rs = self._rs[c] = Rescaler(self.rescale[0], self.rescale[1], self.dms, self.near_max)
                    rs.set_blocks(*self._blocks_args)
                    blocks_src, _, _, views[c] = rs.run(dvol, c, origins, shapes, 0)
            main = views[channel]
            subs = [views[k] for k, _ in self.subtract]
        elif self.dms is None:
            blocks_src, _ = bl._make_blocks(dvol, 0, origins, shapes)
            main = dvol.view(channel, False)
            subs = [dvol.view(k, False) for k, _ in self.subtract]
        else:
            views = {}
            for c in [channel] + [k for k, _ in self.subtract]:
                if c not in views:
                    pre = self._pres.setdefault(c, Preprocessor(self.dms, self.near_max))
                    blocks_src, _, _, views[c] = pre.run(dvol, c, origins, shapes, 0)
            main = views[channel]
            subs = [views[k] for k, _ in self.subtract]
        nb = len(shapes)
        shp = np.asarray(shapes, dtype=np.int64).reshape(nb, 3)
        sx = int(-(-shp[:, 2].max() // nat.MMX_ROW_ALIGN) * nat.MMX_ROW_ALIGN)
        dst_sy, dst_sz = sx, sx * int(shp[:, 1].max())
        slot_pre = dst_sz * int(shp[:, 0].max())
        out32 = self._buffer("_out32", which, nb * slot_pre, torch.float32, dev)
        out64 = self._buffer("_out64", which, nb * slot_pre, torch.float64, dev)
        d_src = torch.from_numpy(blocks_src.view(np.uint8).reshape(-1)).to(dev)
        sub_arr = (nat.Volume * max(1, len(subs)))(*subs)
        facs = np.array([f for _, f in self.subtract], dtype=np.float64)
        nat.check(L.mmx_unmix_batch(
            ctypes.byref(main), sub_arr, nat.as_double_ptr(facs) if len(facs) else None, len(subs),
            d_src.data_ptr(), blocks_src.ctypes.data, nb, slot_pre, dst_sy, dst_sz,
            out32.data_ptr(), out64.data_ptr(), torch.cuda.current_stream().cuda_stream), "mmx_unmix_batch")
        self._keep = d_src
        blocks = np.zeros(nb, dtype=nat.BLOCK_DTYPE)
        slot = 1
        for i in range(nb):
            px = -(-int(shp[i, 2]) // nat.MMX_ROW_ALIGN) * nat.MMX_ROW_ALIGN
            blocks[i] = (i * slot_pre, shp[i, 0], shp[i, 1], shp[i, 2], i, px, 0)
            slot = max(slot, int(shp[i, 0]) * int(shp[i, 1]) * px)
        self._raw_scale = (float(np.iinfo(dvol.np_dtype).max) if dvol.np_dtype.kind in "ui"
                           else dvol.value_scale())
        self.last_geometry = (slot_pre, dst_sz, dst_sy, out64, out32)
        return (blocks, slot, nat.Volume(out32.data_ptr(), nat.MMX_F32, 0, dst_sz, dst_sy, 1),
                nat.Volume(out64.data_ptr(), nat.MMX_F64, 0, dst_sz, dst_sy, 1))

    fetch = Preprocessor.fetch


def calc_isotropic_factor(scale, res=None) -> np.ndarray:
    """``cv_nd.calc_isotropic_factor`` (reference cv_nd.py:1040-1067)."""
    if res is None:
        res = config.resolutions[0]
    resize_factor = np.divide(res, np.amin(res))
    resize_factor = resize_factor * scale
    return resize_factor


_AXIS_TABLES: Dict[Tuple[int, int], Tuple[np.ndarray, np.ndarray]] = {}


def zoom_axis_table(n_in: int, n_out: int, mode: str = "mirror") -> Tuple[np.ndarray, np.ndarray]:
    """``(index (n_out, 2) int32, weight (n_out, 2) float64)`` of one axis of
    ``scipy.ndimage.zoom(order=1, mode=mode, grid_mode=True)``: SciPy's NI_ZoomShift computes, per
    output index k, ``cc = (k + 0.5) * (n_in / n_out) - 0.5``, extends it into the array -- ``mirror``
    ("whole-sample symmetric": -c -> c, beyond the end 2(n-1) - c) or ``nearest`` (what scikit-image's
    ``mode='edge'`` becomes, used by the reference for blocks one voxel thick, cv_nd.py:1096-1101: the
    coordinate stays as it is, only the sample indices are clamped to [0, n-1], so an outside point is
    ``v * w0 + v * w1`` of the edge sample, not ``v``) -- takes ``floor(cc)`` and the next sample (border
    indices extended the same way) with weights ``w0 = 1 - frac``, ``w1 = 1 - w0``.  Same double arithmetic
    here (checked against SciPy bit for bit, tests/test_host_logic.py)."""
    key = (int(n_in), int(n_out), mode)
    hit = _AXIS_TABLES.get(key)
    if hit is not None:
        return hit
    n_in, n_out, _ = key
    if mode not in ("mirror", "nearest"):
        raise ValueError(mode)
    if n_in < 2 and mode == "mirror":
        raise ValueError("SciPy's mirror extension needs at least two samples")
    zoom = np.divide(np.float64(n_in), np.float64(n_out))
    sz2 = 2 * n_in - 2
    idx = np.zeros((n_out, 2), dtype=np.int32)
    wts = np.zeros((n_out, 2), dtype=np.float64)
    for k in range(n_out):
        cc = np.float64(k)
        cc = cc + 0.5
        cc = cc * zoom
        cc = cc - 0.5
        if mode == "nearest":
            pass            # SciPy leaves the coordinate alone and clamps the two sample indices below
        elif cc < 0:
            cc = sz2 * int(-cc / sz2) + cc
            cc = cc + sz2 if cc <= 1 - n_in else -cc
        elif cc > n_in - 1:
            cc = cc - sz2 * int(cc / sz2)
            if cc >= n_in:
                cc = sz2 - cc
        start = int(np.floor(cc))
        for ll in range(2):
            j = start + ll
            if mode == "nearest":
                j = 0 if j < 0 else (n_in - 1 if j >= n_in else j)
            elif j < 0:
                j = sz2 * int(-j / sz2) + j
                j = j + sz2 if j <= 1 - n_in else -j
            elif j >= n_in:
                j -= sz2 * int(j / sz2)
                if j >= n_in:
                    j = sz2 - j
            idx[k, ll] = j
        frac = cc - np.floor(cc)
        wts[k, 0] = 1.0 - frac
        wts[k, 1] = 1.0 - wts[k, 0]
    _AXIS_TABLES[key] = (idx, wts)
    return idx, wts


def isotropic_shape(shape3, factor) -> Tuple[int, int, int]:
    """``(shape * factor).astype(int)`` (cv_nd.py:1091-1092)."""
    out = (np.array(shape3, dtype=int) * np.asarray(factor)).astype(int)
    if (out < 1).any():
        raise ValueError(f"isotropic rescale of a {tuple(shape3)} block gives an empty array")
    return tuple(int(v) for v in out)


class Rescaler:
    """The profile's ``isotropic`` rescale of the blocks of a batch (reference detector.py:893-897 ->
    cv_nd.make_isotropic -> skimage.transform.resize), after the optional per-block preprocessing, as
    a ``run`` source for ``blob_log_blocks``.  ``run`` is called with the RESIZED shapes (those are the
    images the detection sees); :meth:`set_blocks` tells it the original extents.  scikit-image clips the
    result to the value range of the whole multichannel block, so every channel of the ROI is scanned."""

    def __init__(self, factor, channels: Sequence[int], denoise_max_shape=None, near_max=None):
        self.factor = np.asarray(factor, dtype=np.float64)
        self.channels = [int(c) for c in channels]
        self.dms = None if denoise_max_shape is None else [int(v) for v in denoise_max_shape]
        self.near_max = near_max
        self._pres: Dict[int, Preprocessor] = {}
        self._orig: Dict[Tuple[Tuple[int, ...], Tuple[int, ...]], Tuple[int, int, int]] = {}
        self._out = [None] * N_BUFFER_SETS
        self._out32 = [None] * N_BUFFER_SETS
        self._aa0 = self._aa1 = None          # ping-pong buffers of the anti-aliasing passes
        self._scale = 1.0

    def set_blocks(self, origins, shapes, new_shapes) -> None:
        for o, s, n in zip(origins, shapes, new_shapes):
            s, n = tuple(int(v) for v in s), tuple(int(v) for v in n)
            self._orig[(tuple(int(v) for v in o), n)] = s

    @staticmethod
    def aa_radius(n_in: int, n_out: int) -> Tuple[float, int]:
        """``(sigma, radius)`` of scikit-image's anti-aliasing Gaussian along one axis (``resize`` with the default
        ``anti_aliasing``: sigma = max(0, (n_in / n_out - 1) / 2), SciPy's ``truncate=4``)."""
        sigma = max(0.0, (n_in / n_out - 1) / 2)
        return sigma, int(4.0 * sigma + 0.5)

    def bytes_per_voxel(self) -> int:
        # resized copies (float64 + float32, or uint16, per buffer set) + the preprocessed sources of all channels
        return _SLOT_BYTES + (0 if self.dms is None else (_SLOT_BYTES + 2) * len(self.channels))

    def value_scale(self, channels: Sequence[int]) -> float:
        if self.dms is not None:
            return Preprocessor(self.dms).value_scale(self.channels)
        return self._scale

    def value_range(self, channels: Sequence[int]):
        """Interpolation and the anti-aliasing Gaussian are convex combinations: the source's range."""
        if self.dms is not None:
            return Preprocessor(self.dms).value_range(self.channels)
        return None

    _buffer = Preprocessor._buffer

    def run(self, dvol, channel: int, origins, shapes, which: int = 0):
        _wait_upload(dvol, origins, shapes)
        from . import blob_log as bl
        L = nat.lib()
        dev = dvol.tensor.device
        stream = torch.cuda.current_stream().cuda_stream
        nb = len(shapes)
        new_shp = np.asarray(shapes, dtype=np.int64).reshape(nb, 3)
        orig = [self._orig[(tuple(int(v) for v in o), tuple(int(v) for v in n))]
                for o, n in zip(origins, new_shp)]
        # ---- sources: the block as detect_blobs receives it, every channel (for the clip range)
        if self.dms is None:
            if dvol.np_dtype not in (np.dtype(np.uint8), np.dtype(np.uint16), np.dtype(np.float32),
                                     np.dtype(np.float64)):
                raise NotImplementedError(f"isotropic rescale of {dvol.np_dtype} images is not built")
            blocks_src, _ = bl._make_blocks(dvol, 0, origins, orig)
            views = {c: dvol.view(c, False) for c in range(dvol.n_channels)}
            zero_channels = False
            self._scale = dvol.value_scale()
        else:
            views = {}
            for c in self.channels:
                pre = self._pres.setdefault(c, Preprocessor(self.dms, self.near_max))
                blocks_src, _, _, views[c] = pre.run(dvol, c, origins, orig, 0)
            zero_channels = len(self.channels) < dvol.n_channels      # unselected channels stay 0
        d_src = torch.from_numpy(blocks_src.view(np.uint8).reshape(-1)).to(dev)
        mm = np.empty((nb, 2))
        mm[:, 0], mm[:, 1] = (0.0, 0.0) if zero_channels else (np.inf, -np.inf)
        d_mm = torch.from_numpy(mm).to(dev)
        for c in sorted(views):
            nat.check(L.mmx_minmax_batch(ctypes.byref(views[c]), d_src.data_ptr(), blocks_src.ctypes.data, nb,
                                         d_mm.data_ptr(), stream), "mmx_minmax_batch")
        # ---- anti-aliasing: a block that shrinks along any axis is first smoothed along its shrinking axes
        # (skimage.transform.resize -> ndi.gaussian_filter(image.astype(float), (factor - 1) / 2, mode=...)),
        # axis by axis in SciPy's order, exact float64; the resize below then reads the smoothed copy
        src = views[channel]
        out_code = src.dtype
        aa = [[self.aa_radius(orig[i][a], int(new_shp[i, a])) if any(int(new_shp[i, b]) < orig[i][b] for b in range(3))
               else (0.0, 0) for a in range(3)] for i in range(nb)]
        if any(r > 0 for blk in aa for _, r in blk):
            from . import kernels1d as k1
            f32 = src.dtype == nat.MMX_F32        # SciPy filters a float32 image into float32 arrays
            oshp = np.asarray(orig, dtype=np.int64).reshape(nb, 3)
            asx = int(-(-oshp[:, 2].max() // nat.MMX_ROW_ALIGN) * nat.MMX_ROW_ALIGN)
            a_sy, a_sz = asx, asx * int(oshp[:, 1].max())
            a_slot = a_sz * int(oshp[:, 0].max())
            cur_vol, cur_blocks, cur_dblocks = src, blocks_src, d_src
            keep = []
            for a in range(3):
                radii = np.array([aa[i][a][1] for i in range(nb)], dtype=np.int32)
                if not radii.any():
                    continue
                pitch = int(radii.max()) + 1
                wts = np.zeros((nb, pitch))
                for i in range(nb):
                    # (radius 0: SciPy's one-tap kernel is exactly [1.0]: the pass copies)
                    wts[i, :radii[i] + 1] = (k1.gaussian_half_kernel(aa[i][a][0], 0, int(radii[i])) if radii[i]
                                             else [1.0])
                edge = np.array([min(orig[i]) == 1 or (dvol.multichannel and dvol.n_channels == 1)
                                 for i in range(nb)])
                # (a batch may mix unit-thick remainder blocks -- 'nearest' -- with regular ones -- 'mirror': the
                #  block table carries the choice per block)
                mixed = bool(edge.any() and not edge.all())
                if mixed:
                    cur_blocks = cur_blocks.copy()
                    cur_blocks["_pad"] = edge.astype(np.int32)
                    cur_dblocks = torch.from_numpy(cur_blocks.view(np.uint8).reshape(-1)).to(dev)
                buf = self._buffer("_aa%d" % (len(keep) & 1), None, nb * a_slot,
                                   torch.float32 if f32 else torch.float64, dev)
                if buf.dtype != (torch.float32 if f32 else torch.float64):
                    buf = torch.empty(nb * a_slot, dtype=torch.float32 if f32 else torch.float64, device=dev)
                    setattr(self, "_aa%d" % (len(keep) & 1), buf)
                d_w = torch.from_numpy(wts.reshape(-1)).to(dev)
                d_r = torch.from_numpy(radii).to(dev)
                nat.check(L.mmx_gauss_axis_batch(ctypes.byref(cur_vol), cur_dblocks.data_ptr(), cur_blocks.ctypes.data,
                                                 nb, a, d_w.data_ptr(), d_r.data_ptr(), pitch, 2 if mixed else int(edge.all()),
                                                 a_slot, a_sy, a_sz, buf.data_ptr(), stream), "mmx_gauss_axis_batch")
                nxt = np.zeros(nb, dtype=nat.BLOCK_DTYPE)
                for i in range(nb):
                    nxt[i] = (i * a_slot, oshp[i, 0], oshp[i, 1], oshp[i, 2], i, asx, 0)
                cur_vol = nat.Volume(buf.data_ptr(), nat.MMX_F32 if f32 else nat.MMX_F64, 0, a_sz, a_sy, 1)
                cur_blocks = nxt
                cur_dblocks = torch.from_numpy(nxt.view(np.uint8).reshape(-1)).to(dev)
                keep.append((buf, d_w, d_r, cur_dblocks))
            src, blocks_src = cur_vol, cur_blocks
            self._keep_aa = keep
        # ---- axis tables + block descriptors
        tabs_i, tabs_w, at, offs = [], [], 0, {}
        rb = np.zeros(nb, dtype=nat.RESIZE_DTYPE)
        for i in range(nb):
            t3 = []
            # a block with an axis of length 1 (channels included) is resized in 'edge' mode (cv_nd.py:1096-1101)
            edge = min(orig[i]) == 1 or (dvol.multichannel and dvol.n_channels == 1)
            for a in range(3):
                key = (int(orig[i][a]), int(new_shp[i, a]), "nearest" if edge else "mirror")
                if key not in offs:
                    ix, w = zoom_axis_table(*key)
                    offs[key] = at
                    tabs_i.append(ix)
                    tabs_w.append(w)
                    at += len(ix)
                t3.append(offs[key])
            rb[i] = (blocks_src["src_off"][i], orig[i][0], orig[i][1], orig[i][2],
                     new_shp[i, 0], new_shp[i, 1], new_shp[i, 2], i, t3[0], t3[1], t3[2])
        d_idx = torch.from_numpy(np.concatenate(tabs_i).reshape(-1)).to(dev)
        d_wts = torch.from_numpy(np.concatenate(tabs_w).reshape(-1)).to(dev)
        d_rb = torch.from_numpy(rb.view(np.uint8).reshape(-1)).to(dev)
        # ---- output slots (uniform strides, 128-byte rows)
        sx = int(-(-new_shp[:, 2].max() // nat.MMX_ROW_ALIGN) * nat.MMX_ROW_ALIGN)
        dst_sy, dst_sz = sx, sx * int(new_shp[:, 1].max())
        slot_pre = dst_sz * int(new_shp[:, 0].max())
        if out_code == nat.MMX_F64:
            out = self._buffer("_out", which, nb * slot_pre, torch.float64, dev)
            out32 = self._buffer("_out32", which, nb * slot_pre, torch.float32, dev)
            vol32 = nat.Volume(out32.data_ptr(), nat.MMX_F32, 0, dst_sz, dst_sy, 1)
            vol_exact = nat.Volume(out.data_ptr(), nat.MMX_F64, 0, dst_sz, dst_sy, 1)
            o32 = out32.data_ptr()
        else:
            # integer images come back in their own type; float32 images as float32 (SciPy interpolates in
            # double and stores into a float32 array; detection then runs in float32 like the reference's)
            tdt = {nat.MMX_U8: torch.uint8, nat.MMX_U16: torch.uint16, nat.MMX_F32: torch.float32}[out_code]
            out = self._buffer("_out", which, nb * slot_pre, tdt, dev)
            if out.dtype != tdt:
                self._out[which] = out = torch.empty(nb * slot_pre, dtype=tdt, device=dev)
            vol32 = vol_exact = nat.Volume(out.data_ptr(), out_code, 0, dst_sz, dst_sy, 1)
            o32 = None
        nat.check(L.mmx_resize_batch_as(ctypes.byref(src), d_rb.data_ptr(), rb.ctypes.data, nb, d_idx.data_ptr(),
                                        d_wts.data_ptr(), d_mm.data_ptr(), slot_pre, dst_sy, dst_sz, int(out_code),
                                        out.data_ptr(), o32, stream), "mmx_resize_batch_as")
        self._keep = (d_src, d_mm, d_idx, d_wts, d_rb)
        blocks = np.zeros(nb, dtype=nat.BLOCK_DTYPE)
        slot = 1
        for i in range(nb):
            px = -(-int(new_shp[i, 2]) // nat.MMX_ROW_ALIGN) * nat.MMX_ROW_ALIGN
            blocks[i] = (i * slot_pre, new_shp[i, 0], new_shp[i, 1], new_shp[i, 2], i, px, 0)
            slot = max(slot, int(new_shp[i, 0]) * int(new_shp[i, 1]) * px)
        self.last_geometry = (slot_pre, dst_sz, dst_sy, out, None)
        self.store_f32 = 1 if out_code == nat.MMX_F32 else 0       # the cube of a float32 image is float32
        return blocks, slot, vol32, vol_exact

    fetch = Preprocessor.fetch


def make_isotropic(roi, scale, res=None, denoise_max_shape=None) -> np.ndarray:
    """``cv_nd.make_isotropic`` of a whole ``(z, y, x[, c])`` ROI through the device path (every channel)."""
    from . import blob_log as bl
    dvol = roi if isinstance(roi, bl.DeviceVolume) else bl.DeviceVolume(roi)
    factor = calc_isotropic_factor(scale, res)
    shape3 = tuple(dvol.shape[:3])
    new = isotropic_shape(shape3, factor)
    chans = list(range(dvol.n_channels))
    rs = Rescaler(factor, chans, denoise_max_shape)
    rs.set_blocks([(0, 0, 0)], [shape3], [new])
    outs = []
    for c in chans:
        rs.run(dvol, c, [(0, 0, 0)], [new], 0)
        outs.append(rs.fetch([new])[0])
    return np.stack(outs, axis=-1) if dvol.multichannel else outs[0]


def preprocess_roi(roi, denoise_max_shape, channel: Optional[Sequence[int]] = None,
                   near_max: Optional[Sequence[float]] = None, return_info: bool = False):
    """Saturate + denoise a ``(z, y, x[, c])`` block tile by tile -> float64 array of the same
    shape: what ``detect_sub_roi`` hands to ``detect_blobs`` (stack_detect.py:122-150).
    Channels not selected stay zero, as in the reference."""
    from . import blob_log as bl
    from .detector import _channels_of
    dvol = roi if isinstance(roi, bl.DeviceVolume) else bl.DeviceVolume(roi)
    multichannel, channels = _channels_of(dvol.tensor.ndim, dvol.n_channels, channel)
    out = np.zeros(dvol.shape, dtype=np.float64)
    infos = []
    for chl in channels:
        pre = Preprocessor(denoise_max_shape, near_max, want_info=return_info)
        pre.run(dvol, chl if multichannel else 0, [(0, 0, 0)], [dvol.shape[:3]])
        block = pre.fetch([dvol.shape[:3]])[0]
        if multichannel:
            out[..., chl] = block
        else:
            out = block
        if return_info:
            infos.append((pre.last_subs, pre.info()))
    return (out, infos) if return_info else out
