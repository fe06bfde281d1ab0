"""The image on the device: ``DeviceVolume`` (a resident ``(z, y, x[, c])`` tensor, or a host image on its way up z-slab by
z-slab beside the detection: ``_SlabUpload``) and the host-side ``img_as_float`` for the dtypes the kernels do not read.
Split out of ``blob_log.py`` (round 5); ``blob_log`` re-exports the public names."""
from __future__ import annotations

import atexit
import bisect
import os
import threading
import weakref
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import _native as nat

try:
    import torch
except Exception as exc:  # pragma: no cover
    raise ImportError("magellanmapper_amd needs PyTorch-ROCm for device memory and streams") from exc


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

    ``streamed``: how a large host image gets there.  ``False``: one synchronous copy -- the constructor returns with
    the voxels read.  ``True``: z-slab by z-slab on a copy stream beside the detection (:class:`_SlabUpload`); the
    constructor returns BEFORE the source has been read, and the caller promises to leave the source untouched until
    :meth:`wait_all` / :meth:`close` (``stack_detect`` opts in for the images it is handed for the length of one call,
    ``Image5d.prefetch`` for the tile it announces).  ``None`` (default): streamed only where nobody can write behind
    the copy -- read-only arrays and read-only memory maps (what the reference's importer hands over,
    importer.py:794) -- and synchronous for ordinary writeable arrays and pinned tensors, which a caller may well
    refill right after this returns (a double-buffered tile).  ``cells = (z ends, y ends)`` of the caller's block rows
    (optional, streamed uploads only): the image then goes up y-band by y-band within each z-layer of blocks instead of
    in whole z-slabs, and a block can start once the bands it touches have landed.

    ``z_off`` / ``full_shape``: ``image`` holds planes ``[z_off, z_off + nz)`` of a larger ``full_shape = (Z, Y, X)``
    volume and the object answers for the WHOLE volume -- ``shape`` is the full one, blocks are addressed by their
    coordinates in it (the views' base pointers are shifted back by ``z_off`` planes), and touching a plane it does not
    hold is the caller's error.  What a rank's share of a stack (``bench.py``) and a z-chunk of an image too large for
    the device (``stack_detect``) are.  ``cells`` are then relative to ``image``.
    """

    _upload = None          # the z-slab upload still in flight (`_SlabUpload`), if any
    z_off = 0

    def __init__(self, image, device: Optional["torch.device"] = None, streamed: Optional[bool] = None, cells=None,
                 z_off: int = 0, full_shape=None):
        dev = device or _require_gpu()
        want_stream = STREAM_UPLOAD and streamed is not False
        if isinstance(image, torch.Tensor):
            t = image
            np_dtype = np.dtype(str(t.dtype).replace("torch.", ""))
        else:
            arr = np.asarray(image)
            if arr.dtype not in _NP_TO_MMX:
                arr = _img_as_float_host(arr)
            np_dtype = arr.dtype
            if np_dtype not in _NP_TO_MMX:
                raise TypeError(f"unsupported voxel type {np_dtype}")
            if arr.ndim not in (3, 4):
                raise ValueError("image must be (z, y, x) or (z, y, x, c)")
            if arr.nbytes > _STREAM_MIN_BYTES and want_stream and (streamed or not arr.flags.writeable):
                # large host images (the reference's callers hand a memory-mapped image5d.npy, importer.py:794) go up
                # z-slab by z-slab on a copy stream; detection starts on the blocks whose slabs have landed
                t = None
                self._upload = _SlabUpload(arr, dev, cells)
                self.tensor = self._upload.out
            else:
                t = torch.from_numpy(np.array(arr) if not arr.flags.writeable else np.ascontiguousarray(arr))
        if np_dtype not in _NP_TO_MMX:
            raise TypeError(f"unsupported voxel type {np_dtype}")
        if t is not None and t.device.type == "cpu" and t.is_pinned() and want_stream and streamed and \
                t.numel() * t.element_size() > _STREAM_MIN_BYTES and t.is_contiguous():
            self._upload = _SlabUpload(t, dev, cells)   # (pinned source: DMA straight from it, no staging thread)
            self.tensor = self._upload.out
            t = None
        if t is not None:
            if t.ndim not in (3, 4):
                raise ValueError("image must be (z, y, x) or (z, y, x, c)")
            self.tensor = t.to(dev).contiguous()
        elif self.tensor.ndim not in (3, 4):
            raise ValueError("image must be (z, y, x) or (z, y, x, c)")
        self.np_dtype = np_dtype
        self.shape = tuple(self.tensor.shape)
        if z_off or full_shape is not None:
            full = tuple(int(v) for v in (full_shape if full_shape is not None else self.shape[:3]))
            if len(full) != 3 or full[1:] != self.shape[1:3] or z_off < 0 or z_off + self.shape[0] > full[0]:
                raise ValueError(f"planes [{z_off}, {z_off + self.shape[0]}) of shape {self.shape[:3]} do not lie in a "
                                 f"volume of shape {full}")
            self.z_off = int(z_off)
            self.shape = full + self.shape[3:]
        self.n_channels = self.shape[3] if self.tensor.ndim == 4 else 1
        self._f32 = None
        self._scale = None
        self._ranges = {}

    @property
    def multichannel(self) -> bool:
        return self.tensor.ndim == 4

    def stream_wait(self, z_hi: Optional[int] = None, streams=None, boxes=None) -> None:
        """Order ``streams`` (default: the current one) after the upload of planes ``[0, z_hi)`` (all planes when
        ``None``) or, with ``boxes`` = ``(z_lo, z_hi, y_lo, y_hi)`` extents of blocks, of the regions holding those.
        Nothing to do for a resident volume.  With a staging thread behind the upload the host waits until that
        region's copy has been QUEUED (its event recorded), never for the copy itself."""
        up = self._upload
        if up is None:
            return
        if boxes is not None:
            ev = up.event_for_boxes(self._local_boxes(boxes))
        else:
            ev = up.event_for(self.tensor.shape[0] if z_hi is None else int(z_hi) - self.z_off)
        from .buffers import _stream_wait
        for st in (streams or [torch.cuda.current_stream()]):
            if st is not None:
                _stream_wait(st, ev)
        if up.all_queued() and up.events[-1].query():
            up.finish()
            self._upload = None             # everything has landed: later calls cost nothing

    def upload_ready(self, boxes) -> bool:
        """True when the copies of every region holding voxels of ``boxes`` have been QUEUED (a ``stream_wait`` for
        them would not block the host); always True for a resident volume."""
        up = self._upload
        return up is None or up.queued_for_boxes(self._local_boxes(boxes))

    def _local_boxes(self, boxes):
        """``(z_lo, z_hi, y_lo, y_hi)`` extents in the whole volume -> in the planes this object holds."""
        if not self.z_off:
            return boxes
        return [(b[0] - self.z_off, b[1] - self.z_off, b[2], b[3]) for b in boxes]

    def wait_all(self) -> None:
        """Host-side wait for the whole upload (readers of the voxels outside the batched detection); from here on the
        source is no longer read."""
        up = self._upload
        if up is not None:
            up.event_for(self.tensor.shape[0]).synchronize()
            up.finish()
            self._upload = None

    def close(self) -> None:
        """Done with this volume, whether or not its upload has finished (a share of the blocks that ends early, a
        partial ROI, a failed detection, an abandoned tile): slabs not yet staged are dropped, the staging thread is
        joined, and the copies already queued keep the device allocation alive until they have run (the allocation
        is recorded on the copy stream), so that the block can go back to the allocator at once without a DMA still
        writing into it.  The source is not read after this returns unless copies of a PINNED source were queued --
        those are waited for here (the DMA reads the caller's buffer directly)."""
        up, self._upload = self._upload, None
        if up is not None:
            up.cancel()

    def __del__(self):
        try:
            self.close()
        except Exception:       # (interpreter shutdown)
            pass

    def value_scale(self) -> float:
        """Magnitude of the image values after ``img_as_float`` (1 for integer images)."""
        if self._scale is None:
            if self.np_dtype.kind == "f":
                self.wait_all()
                m = float(self.tensor.abs().max().item()) if self.tensor.numel() else 1.0
                self._scale = max(1.0, m)
            else:
                self._scale = 1.0
        return self._scale

    def value_range(self, channel: int = 0) -> Tuple[float, float]:
        """``(min, max)`` of one channel's voxels after ``img_as_float`` (``(0, 1)`` for integer images)."""
        if self.np_dtype.kind != "f":
            return 0.0, 1.0
        key = int(channel) if self.multichannel else 0
        if key not in self._ranges:
            self.wait_all()
            t = self.tensor[..., key] if self.multichannel else self.tensor
            if t.numel() == 0:
                self._ranges[key] = (0.0, 1.0)
            else:
                lo, hi = torch.aminmax(t)
                self._ranges[key] = (float(lo.item()), float(hi.item()))
        return self._ranges[key]

    def _strides(self, t) -> Tuple[int, int, int]:
        st = t.stride()
        return st[0], st[1], st[2]

    def view(self, channel: int, for_f32_passes: bool) -> nat.Volume:
        t = self.tensor
        if for_f32_passes and self.np_dtype == np.float64:
            if self._f32 is None:
                self.wait_all()
                self._f32 = t.to(torch.float32)
            t = self._f32
            code = nat.MMX_F32
        else:
            code = _NP_TO_MMX[self.np_dtype]
        sz, sy, sx = self._strides(t)
        ptr = int(t.data_ptr()) + (int(channel) if self.multichannel else 0) * t.element_size()
        ptr -= self.z_off * int(sz) * t.element_size()        # (plane z of the whole volume is plane z - z_off here)
        return nat.Volume(ptr, code, 0, int(sz), int(sy), int(sx))


#: host images above this size go to the device z-slab by z-slab on a copy stream (`_SlabUpload`) when the caller allows
#: it (`DeviceVolume(streamed=...)`); 0 / False keeps the one synchronous copy (tests compare the two)
_STREAM_MIN_BYTES = 64 << 20
_STREAM_CHUNK_BYTES = 128 << 20
STREAM_UPLOAD = True
#: threads that fill the pinned staging buffers from a pageable / memory-mapped source.  One memcpy stream reads
#: ~10 GB/s, the link takes 55-57: four threads (round 5) left a memory-mapped C3 volume staging-bound (8.6 GB / ~40
#: GB/s = 215 ms against 150 ms of DMA), sixteen take memory bandwidth and cores from the detection's own host side
#: (a preprocessed tile from a memory map: 292 ms against 206 with six; raw C3: 176 with eight, 174 with six); 0 = six
#: where the machine has them
_STAGE_THREADS = 0
#: the staging loop of plain arrays and memory maps as one native call (False: the Python loop, which array subclasses
#: and non-contiguous sources always take)
NATIVE_STAGING = True
#: pinned staging buffers of one upload in flight (each one slab): with three the threads fill slab k + 2 while the
#: DMA reads slab k and slab k + 1 waits its turn
_STAGE_DEPTH = 3
#: page-locked staging memory kept between uploads (bytes); buffers beyond it are freed when their upload ends
_STAGING_KEEP_BYTES = 1 << 30
_UPLOAD_STREAMS: Dict[str, "torch.cuda.Stream"] = {}
_STAGING: List["torch.Tensor"] = []                 # free pinned staging buffers (uint8), any size
_STAGING_LOCK = threading.Lock()
_LIVE_UPLOADS: "weakref.WeakSet" = weakref.WeakSet()
_LAST_STAGED: Dict[str, "weakref.ref"] = {}         # per device: the staged upload the next one queues up behind
_QUEUES_CHECKED = [False]


def _stage_threads() -> int:
    n = int(_STAGE_THREADS)
    if n <= 0:
        n = max(2, min(6, (os.cpu_count() or 8) // 4))
    return n


def _take_staging(nbytes: int) -> "torch.Tensor":
    """A pinned uint8 buffer of at least ``nbytes``: the smallest free one that is large enough, else a new one
    (pinning 128 MiB takes tens of ms, which is why they are kept)."""
    with _STAGING_LOCK:
        fit = [b for b in _STAGING if b.numel() >= nbytes]
        if fit:
            buf = min(fit, key=lambda b: b.numel())
            _STAGING[:] = [b for b in _STAGING if b is not buf]         # (by identity: `==` on tensors compares elements)
            return buf
    return torch.empty(int(nbytes), dtype=torch.uint8).pin_memory()


def _give_staging(bufs) -> None:
    with _STAGING_LOCK:
        for b in bufs:
            if sum(x.numel() for x in _STAGING) + b.numel() <= _STAGING_KEEP_BYTES:
                _STAGING.append(b)


def release_staging() -> None:
    """Drop the pinned staging buffers and the copy streams kept between uploads (``buffers.release_buffers``)."""
    with _STAGING_LOCK:
        _STAGING.clear()
    _UPLOAD_STREAMS.clear()
    _LAST_STAGED.clear()


def _check_hw_queues() -> None:
    """Once per process, at the first streamed upload: the copy stream needs a hardware queue of its own, which takes
    ``GPU_MAX_HW_QUEUES`` >= 8 read WHEN THE HIP RUNTIME STARTS (magellanmapper_amd/__init__ sets it at import; a host
    application that initialised the GPU before importing this package, or that set a smaller value itself, gets the
    default 4 and kernels queue behind the copies: 302 against 254 ms per tile)."""
    if _QUEUES_CHECKED[0]:
        return
    _QUEUES_CHECKED[0] = True
    import warnings
    from . import GPU_WAS_INITIALISED_AT_IMPORT
    try:
        queues = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    except ValueError:
        queues = 4
    if GPU_WAS_INITIALISED_AT_IMPORT or queues < 8:
        warnings.warn(
            "magellanmapper_amd: uploads that run beside the detection want GPU_MAX_HW_QUEUES >= 8 set before the HIP "
            f"runtime starts (now: {queues}{', and the GPU was initialised before this package was imported' if GPU_WAS_INITIALISED_AT_IMPORT else ''}); "
            "the copy stream may share a hardware queue with a kernel stream, which costs up to 20 % per streamed tile",
            RuntimeWarning, stacklevel=3)


class _SlabUpload:
    """A host ``(z, y, x[, c])`` image on its way to the device region by region, on a stream of its own: one event per
    region, so that the detection of the blocks a region completes can start while the rest is still in flight.
    Regions are z-slabs (blocks are consumed in z-major order) or, when the caller says where its block rows end
    (``cells = (z ends, y ends)``), y-bands of z-layers in (z, y) order -- block row j of layer l can then start once
    band j of that layer has landed instead of the whole layer, and behind the LAST band only one row of blocks is left
    to detect instead of a whole layer (a quarter of the benchmark volume).  A pinned source is read by the DMA engine
    directly -- every copy is queued at once; a pageable or memory-mapped one goes through a small ring of pinned
    staging buffers filled by a few host threads.

    Lifetime: the device allocation is recorded on the copy stream (``record_stream``), so dropping the volume while
    copies are in flight cannot hand the block to another allocation before they have run; :meth:`cancel` drops the
    regions not yet staged and joins the staging thread; the source must stay untouched until the upload has finished
    or been cancelled."""

    def __init__(self, src, dev, cells=None):
        _check_hw_queues()
        if isinstance(src, torch.Tensor):
            shape, tdtype, itemsize = tuple(src.shape), src.dtype, src.element_size()
        else:
            shape, tdtype, itemsize = tuple(src.shape), getattr(torch, str(src.dtype)), src.dtype.itemsize
        import time
        self._t0 = time.perf_counter()
        self.out = torch.empty(shape, dtype=tdtype, device=dev)
        self.dev = dev
        self.nz = shape[0]
        self.ny = shape[1] if len(shape) > 1 else 1
        self.itemsize = itemsize
        self.row_bytes = max(1, int(np.prod(shape[2:])) * itemsize)
        plane = max(1, self.ny * self.row_bytes)
        self.slab = max(1, min(self.nz, _STREAM_CHUNK_BYTES // plane))
        self.regions = self._plan(cells)      # (z0, z1, y0, y1) in upload order
        self._boxes = np.asarray(self.regions, dtype=np.int64).reshape(-1, 4)
        self._events: List = []              # one per region: appended as queued, or all made up front (native staging)
        self._nq = None                      # native staging: [regions queued so far], written by the staging call
        self._cancel = np.zeros(1, dtype=np.int32)
        self.n_slabs = len(self.regions)
        self.cv = threading.Condition()
        self.error: Optional[BaseException] = None
        self.cancelled = False
        self.thread: Optional[threading.Thread] = None
        self._keep = None
        # ONE copy stream per device for every upload: streams are dealt to the hardware queues round-robin as they are
        # made, so a stream per volume would sooner or later share a queue with a kernel stream (magellanmapper_amd/__init__)
        self.stream = _UPLOAD_STREAMS.get(str(dev))
        if self.stream is None:
            self.stream = _UPLOAD_STREAMS[str(dev)] = torch.cuda.Stream(dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))        # (the allocation above)
        # the block is written from the copy stream: the allocator must not reuse it before that stream is through with it
        self.out.record_stream(self.stream)
        _LIVE_UPLOADS.add(self)
        if self.nz == 0 or not self.regions:
            ev = torch.cuda.Event()
            ev.record(self.stream)
            self._events.append(ev)
            self.regions, self.n_slabs = [(0, 0, 0, self.ny)], 1
            self._boxes = np.asarray(self.regions, dtype=np.int64).reshape(-1, 4)
        elif isinstance(src, torch.Tensor) and src.is_pinned():
            base = src.data_ptr()
            for z0, z1, y0, y1 in self.regions:
                self._queue(z0, z1, y0, y1, src, base)
            self._keep = src                       # (the source must outlive the copies)
        else:
            arr = src.numpy() if isinstance(src, torch.Tensor) else src
            # one staged upload at a time per device: the copy stream is a queue -- the next tile's regions queued
            # between this tile's would delay the blocks waiting for them (two uploads interleaved: 347 against 283 ms
            # per C5 tile) -- so this one starts staging once the one before it has queued its last region
            prev = _LAST_STAGED.get(str(dev))
            prev = prev() if prev is not None else None
            _LAST_STAGED[str(dev)] = weakref.ref(self)
            self.thread = threading.Thread(target=self._stage, args=(arr, tdtype, prev), daemon=True, name="mmx-upload")
            self.thread.start()

    def _plan(self, cells):
        """Upload order: z-slabs of at most ``_STREAM_CHUNK_BYTES``; with ``cells = (z ends, y ends)`` (ascending, the
        last ones the image's extent) the y-bands of every z-layer in turn, a band cut along z where it exceeds that."""
        nz, ny = self.nz, self.ny
        if nz == 0:
            return []
        banded = None
        if cells is not None:
            z_ends = sorted({int(v) for v in cells[0] if 0 < int(v) < nz} | {nz})
            y_ends = sorted({int(v) for v in cells[1] if 0 < int(v) < ny} | {ny})
            if len(y_ends) > 1:
                banded = (z_ends, y_ends)
        if banded is None:
            return [(z0, min(z0 + self.slab, nz), 0, ny) for z0 in range(0, nz, self.slab)]
        regions = []
        za = 0
        for zb in banded[0]:
            ya = 0
            for yb in banded[1]:
                step = max(1, _STREAM_CHUNK_BYTES // max(1, (yb - ya) * self.row_bytes))
                for z0 in range(za, zb, step):
                    regions.append((z0, min(z0 + step, zb), ya, yb))
                ya = yb
            za = zb
        return regions

    def _queue(self, z0, z1, y0, y1, src, src_base, src_pitch=None):
        """One region's copy on the copy stream + its event.  ``src``: a pinned tensor holding the whole image
        (``src_pitch`` None: the image's own plane pitch) or the region alone, packed (``src_pitch`` = its plane bytes)."""
        full_rows = y0 == 0 and y1 == self.ny
        with torch.cuda.stream(self.stream):
            if full_rows and src_pitch is None:
                self.out[z0:z1].copy_(src[z0:z1], non_blocking=True)
            elif full_rows:
                self.out[z0:z1].copy_(src, non_blocking=True)
            else:
                plane = self.ny * self.row_bytes
                width = (y1 - y0) * self.row_bytes
                dst = self.out.data_ptr() + z0 * plane + y0 * self.row_bytes
                if src_pitch is None:
                    ptr, pitch = src_base + z0 * plane + y0 * self.row_bytes, plane
                else:
                    ptr, pitch = src_base, src_pitch
                nat.check(nat.lib().mmx_copy_rect_h2d(dst, plane, ptr, pitch, width, z1 - z0, self.stream.cuda_stream),
                          "mmx_copy_rect_h2d")
            ev = torch.cuda.Event()
            ev.record()
        with self.cv:
            self._events.append(ev)
            self.cv.notify_all()
        return ev

    @property
    def n_queued(self) -> int:
        return len(self._events) if self._nq is None else int(self._nq[0])

    @property
    def events(self) -> List:
        """The events of the regions queued so far, in upload order."""
        return self._events[:self.n_queued]

    @property
    def bounds(self) -> List[int]:
        """z end of every region queued so far."""
        return [r[1] for r in self.regions[:self.n_queued]]

    def all_queued(self) -> bool:
        return self.n_queued >= max(1, self.n_slabs)

    def finish(self) -> None:
        """Every copy has completed (the caller has seen the last event): the source is released, the thread gone."""
        th, self.thread = self.thread, None
        if th is not None and th is not threading.current_thread():
            th.join()
        self._keep = None

    def cancel(self) -> None:
        """Give the upload up: regions not yet staged are dropped (waiters are told), the staging thread is joined;
        copies of a pinned source that are already queued cannot be recalled and are waited for, because the DMA reads
        the caller's buffer.  Safe to call more than once and after the upload has finished."""
        with self.cv:
            self.cancelled = True
            self._cancel[0] = 1
            self.cv.notify_all()
        th, self.thread = self.thread, None
        if th is not None and th is not threading.current_thread():
            th.join()
        if self._keep is not None:
            queued = self.events
            if queued:
                queued[-1].synchronize()
            self._keep = None

    def _wait_queued(self, waiter) -> None:
        """Block until this upload has queued its last region (or gave up); ``waiter`` stops waiting when cancelled."""
        with self.cv:
            while not (self.all_queued() or self.cancelled or self.error is not None or waiter.cancelled):
                self.cv.wait(0.002)

    def _stage(self, arr, tdtype, prev=None):
        stage: List = []
        done: List = []
        try:
            if prev is not None:
                prev._wait_queued(self)
            torch.cuda.set_device(self.dev)
            if NATIVE_STAGING and type(arr) in (np.ndarray, np.memmap) and arr.flags.c_contiguous:
                return self._stage_native(arr)
            inner = tuple(arr.shape[2:])
            itemsize = self.itemsize
            need = max((z1 - z0) * (y1 - y0) for z0, z1, y0, y1 in self.regions) * self.row_bytes
            depth = max(2, min(int(_STAGE_DEPTH), len(self.regions)))
            stage = [_take_staging(need) for _ in range(depth)]
            done = [None] * depth
            n_thr = _stage_threads()
            if isinstance(arr, np.memmap):
                _advise_sequential(arr)
            with ThreadPoolExecutor(n_thr, thread_name_prefix="mmx-stage") as pool:
                for k, (z0, z1, y0, y1) in enumerate(self.regions):
                    if self.cancelled:
                        break
                    which = k % depth
                    n_el = (z1 - z0) * (y1 - y0) * int(np.prod(inner, dtype=np.int64))
                    buf = stage[which][:n_el * itemsize].view(tdtype).view((z1 - z0, y1 - y0) + inner)
                    if done[which] is not None:
                        done[which].synchronize()          # the DMA that last used this buffer
                    host = buf.numpy()
                    cuts = np.linspace(0, z1 - z0, min(n_thr, z1 - z0) + 1).astype(int)
                    list(pool.map(lambda ab: np.copyto(host[ab[0]:ab[1]], arr[z0 + ab[0]:z0 + ab[1], y0:y1]),
                                  zip(cuts[:-1], cuts[1:])))
                    if self.cancelled:
                        break
                    done[which] = self._queue(z0, z1, y0, y1, buf, buf.data_ptr(), (y1 - y0) * self.row_bytes)
        except BaseException as exc:               # (reported by whoever waits for a region)
            with self.cv:
                self.error = exc
                self.cv.notify_all()
        finally:
            try:
                for ev in done:
                    if ev is not None:
                        ev.synchronize()           # (the buffers go back only when the DMA has read them)
                _give_staging(stage)
            except BaseException:                  # (interpreter shutdown / a lost device: the buffers are dropped)
                pass

    def _stage_native(self, arr) -> None:
        """The whole staging loop in ONE native call (``mmx_host_stage_upload``): nothing in it needs the interpreter lock,
        which a busy detection holds most of the time -- the Python loop above waited for it between every two regions
        (a two-channel tile from a memory map: 413 against 289 ms from pinned memory)."""
        from .buffers import _NativeEvent
        need = max((z1 - z0) * (y1 - y0) for z0, z1, y0, y1 in self.regions) * self.row_bytes
        depth = max(2, min(int(_STAGE_DEPTH), len(self.regions)))
        stage = [_take_staging(need) for _ in range(depth)]
        try:
            events = [_NativeEvent() for _ in self.regions]
            nq = np.zeros(1, dtype=np.int64)
            with self.cv:
                self._events, self._nq = events, nq
                self.cv.notify_all()
            if isinstance(arr, np.memmap):
                _advise_sequential(arr)
            import ctypes
            regions = np.ascontiguousarray(self.regions, dtype=np.int64)
            if os.environ.get("MMX_STAGE_PROF"):
                import sys, time
                print(f"_SlabUpload: {(time.perf_counter() - self._t0) * 1e3:.2f} ms from the constructor to the staging call",
                      file=sys.stderr)
            rc = nat.lib().mmx_host_stage_upload(
                arr.ctypes.data, self.out.data_ptr(), regions.ctypes.data, len(regions), self.nz, self.ny, self.row_bytes,
                (ctypes.c_void_p * depth)(*[b.data_ptr() for b in stage]), need, depth,
                (ctypes.c_void_p * len(events))(*[e.handle for e in events]), self.stream.cuda_stream,
                self.dev.index if self.dev.index is not None else torch.cuda.current_device(),
                nq.ctypes.data, self._cancel.ctypes.data, _stage_threads())
            nat.check(rc, "mmx_host_stage_upload")
        finally:
            try:
                for ev in self.events[-depth:]:
                    ev.synchronize()               # (the buffers go back only when the DMA has read them)
                _give_staging(stage)
            except BaseException:
                pass
            with self.cv:
                self.cv.notify_all()

    def event_for(self, z_hi: int):
        """The event after which planes ``[0, z_hi)`` are on the device (waits until its copy has been queued)."""
        return self.event_for_boxes([(0, int(z_hi), 0, self.ny)])

    def _last_region(self, boxes) -> int:
        last = 0
        rg = self._boxes
        for z_lo, z_hi, y_lo, y_hi in boxes:
            z_lo, z_hi = max(0, int(z_lo)), min(int(z_hi), self.nz)
            y_lo, y_hi = max(0, int(y_lo)), min(int(y_hi), self.ny)
            if z_hi <= z_lo or y_hi <= y_lo:
                continue
            hit = np.flatnonzero((rg[:, 0] < z_hi) & (rg[:, 1] > z_lo) & (rg[:, 2] < y_hi) & (rg[:, 3] > y_lo))
            if len(hit):
                last = max(last, int(hit[-1]))
        return last

    def queued_for_boxes(self, boxes) -> bool:
        """Whether ``event_for_boxes(boxes)`` would return without waiting (errors and a cancelled upload count as
        ready: the wait itself reports them)."""
        if self.error is not None or self.cancelled:
            return True
        n = self.n_queued
        return self._last_region(boxes) < n or n >= max(1, self.n_slabs)

    def event_for_boxes(self, boxes):
        """The event after which every voxel of the ``(z_lo, z_hi, y_lo, y_hi)`` boxes (all x) is on the device: that
        of the last region, in upload order, that holds any of them (the copy stream runs the regions in order).  Waits
        until that region's copy has been QUEUED, never for the copy itself."""
        last = self._last_region(boxes)
        with self.cv:
            while True:
                if self.error is not None:
                    raise nat.MmxError(f"upload of the image failed: {self.error!r}") from self.error
                n = self.n_queued
                if last < n:
                    return self._events[last]
                if n >= max(1, self.n_slabs):
                    return self._events[n - 1]
                if self.cancelled:
                    raise nat.MmxError("upload of the image was cancelled (DeviceVolume.close) before these planes went up")
                # (the native staging loop publishes its progress in a counter, not through this condition: polled)
                self.cv.wait(0.5 if (self._nq is None and self.thread is None) else 0.0003)


def _advise_sequential(arr) -> None:
    """Tell the kernel a memory map is about to be read front to back (read-ahead on a file-backed map; harmless on
    tmpfs).  Best effort: a map this Python cannot advise is read as it is."""
    try:
        import mmap as _mmap
        mm = getattr(arr, "_mmap", None)
        if mm is not None and hasattr(mm, "madvise"):
            mm.madvise(_mmap.MADV_SEQUENTIAL)
            if hasattr(_mmap, "MADV_WILLNEED"):
                mm.madvise(_mmap.MADV_WILLNEED)
    except (OSError, ValueError, AttributeError):
        pass


def _cancel_live_uploads() -> None:
    """At interpreter exit: no staging thread may still be inside HIP calls when the runtime is torn down."""
    for up in list(_LIVE_UPLOADS):
        try:
            up.cancel()
        except BaseException:
            pass


atexit.register(_cancel_live_uploads)


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


