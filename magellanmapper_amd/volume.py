"""The image on the device: ``DeviceVolume`` (a resident ``(z, y, x[, c])`` tensor, or a host image on its way up z-slab by
z-slab beside the detection: ``_SlabUpload``) and the host-side ``img_as_float`` for the dtypes the kernels do not read.
Split out of ``blob_log.py`` (round 5); ``blob_log`` re-exports the public names."""
from __future__ import annotations

import bisect
import threading
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
    """

    _upload = None          # the z-slab upload still in flight (`_SlabUpload`), if any

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
            if np_dtype not in _NP_TO_MMX:
                raise TypeError(f"unsupported voxel type {np_dtype}")
            if arr.ndim not in (3, 4):
                raise ValueError("image must be (z, y, x) or (z, y, x, c)")
            if arr.nbytes > _STREAM_MIN_BYTES and STREAM_UPLOAD:
                # large host images (the reference's callers hand a memory-mapped image5d.npy, importer.py:794) go up
                # z-slab by z-slab on a copy stream; detection starts on the blocks whose slabs have landed
                t = None
                self._upload = _SlabUpload(arr, dev)
                self.tensor = self._upload.out
            else:
                t = torch.from_numpy(np.array(arr) if not arr.flags.writeable else np.ascontiguousarray(arr))
        if np_dtype not in _NP_TO_MMX:
            raise TypeError(f"unsupported voxel type {np_dtype}")
        if t is not None and t.device.type == "cpu" and t.is_pinned() and STREAM_UPLOAD and \
                t.numel() * t.element_size() > _STREAM_MIN_BYTES and t.is_contiguous():
            self._upload = _SlabUpload(t, dev)          # (pinned source: DMA straight from it, no staging thread)
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
        self.n_channels = self.shape[3] if self.tensor.ndim == 4 else 1
        self._f32 = None
        self._scale = None
        self._ranges = {}

    @property
    def multichannel(self) -> bool:
        return self.tensor.ndim == 4

    def stream_wait(self, z_hi: Optional[int] = None, streams=None) -> None:
        """Order ``streams`` (default: the current one) after the upload of planes ``[0, z_hi)`` (all planes when
        ``None``).  Nothing to do for a resident volume.  With a staging thread behind the upload the host waits until
        that slab's copy has been QUEUED (its event recorded), never for the copy itself."""
        up = self._upload
        if up is None:
            return
        ev = up.event_for(self.shape[0] if z_hi is None else int(z_hi))
        for st in (streams or [torch.cuda.current_stream()]):
            if st is not None:
                st.wait_event(ev)
        if up.all_queued() and up.events[-1].query():
            self._upload = None             # everything has landed: later calls cost nothing

    def wait_all(self) -> None:
        """Host-side wait for the whole upload (readers of the voxels outside the batched detection)."""
        up = self._upload
        if up is not None:
            up.event_for(self.shape[0]).synchronize()
            self._upload = None

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
        return nat.Volume(ptr, code, 0, int(sz), int(sy), int(sx))


#: host images above this size go to the device z-slab by z-slab on a copy stream (`_SlabUpload`); 0 / False keeps
#: the one synchronous copy (tests compare the two)
_STREAM_MIN_BYTES = 64 << 20
_STREAM_CHUNK_BYTES = 128 << 20
STREAM_UPLOAD = True
#: threads that fill a pinned staging buffer from a pageable / memory-mapped source (one memcpy stream reads ~10 GB/s,
#: the link takes 57)
_STAGE_THREADS = 4
_UPLOAD_STREAMS: Dict[str, "torch.cuda.Stream"] = {}
_STAGING: Dict[Tuple[str, int], list] = {}          # (dtype, elements) -> free pairs of pinned staging buffers
_STAGING_LOCK = threading.Lock()


class _SlabUpload:
    """A host ``(z, y, x[, c])`` image on its way to the device, z-slab by z-slab, on a stream of its own: one event
    per slab, so that the detection of the blocks a slab completes can start while the rest is still in flight (blocks
    are consumed in z-major order).  A pinned source is read by the DMA engine directly -- every copy is queued at once;
    a pageable or memory-mapped one goes through two pinned staging buffers filled by a few host threads."""

    def __init__(self, src, dev):
        if isinstance(src, torch.Tensor):
            shape, tdtype, itemsize = tuple(src.shape), src.dtype, src.element_size()
        else:
            shape, tdtype, itemsize = tuple(src.shape), getattr(torch, str(src.dtype)), src.dtype.itemsize
        self.out = torch.empty(shape, dtype=tdtype, device=dev)
        self.dev = dev
        self.nz = shape[0]
        plane = max(1, int(np.prod(shape[1:])) * itemsize)
        self.slab = max(1, min(self.nz, _STREAM_CHUNK_BYTES // plane))
        self.bounds: List[int] = []          # z end of every queued slab
        self.events: List = []
        self.n_slabs = -(-self.nz // self.slab) if self.nz else 0
        self.cv = threading.Condition()
        self.error: Optional[BaseException] = None
        # ONE copy stream per device for every upload: streams are dealt to the hardware queues round-robin as they are
        # made, so a stream per volume would sooner or later share a queue with a kernel stream (magellanmapper_amd/__init__)
        self.stream = _UPLOAD_STREAMS.get(str(dev))
        if self.stream is None:
            self.stream = _UPLOAD_STREAMS[str(dev)] = torch.cuda.Stream(dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))        # (the allocation above)
        if self.nz == 0:
            ev = torch.cuda.Event()
            ev.record(self.stream)
            self.bounds.append(0)
            self.events.append(ev)
        elif isinstance(src, torch.Tensor) and src.is_pinned():
            with torch.cuda.stream(self.stream):
                for z0 in range(0, self.nz, self.slab):
                    z1 = min(z0 + self.slab, self.nz)
                    self.out[z0:z1].copy_(src[z0:z1], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record()
                    self.bounds.append(z1)
                    self.events.append(ev)
            self._keep = src                       # (the source must outlive the copies)
        else:
            arr = src.numpy() if isinstance(src, torch.Tensor) else src
            self.thread = threading.Thread(target=self._stage, args=(arr, tdtype), daemon=True)
            self.thread.start()

    def all_queued(self) -> bool:
        return len(self.events) >= max(1, self.n_slabs)

    def _stage(self, arr, tdtype):
        try:
            torch.cuda.set_device(self.dev)
            shape1 = (self.slab,) + tuple(arr.shape[1:])
            key = (str(tdtype), int(np.prod(shape1)))
            # (pinning 128 MiB takes tens of ms: a pair of buffers is kept for the next volume; two uploads at once --
            #  tile k + 1 behind tile k -- each take their own pair)
            with _STAGING_LOCK:
                free = _STAGING.setdefault(key, [])
                stage = free.pop() if free else None
            if stage is None:
                stage = [torch.empty(key[1], dtype=tdtype).pin_memory() for _ in range(2)]
            done = [None, None]
            with ThreadPoolExecutor(_STAGE_THREADS) as pool:
                for k, z0 in enumerate(range(0, self.nz, self.slab)):
                    z1 = min(z0 + self.slab, self.nz)
                    buf = stage[k & 1][:(z1 - z0) * int(np.prod(arr.shape[1:]))].view((z1 - z0,) + tuple(arr.shape[1:]))
                    if done[k & 1] is not None:
                        done[k & 1].synchronize()          # the DMA that last used this buffer
                    host = buf.numpy()
                    cuts = np.linspace(0, z1 - z0, min(_STAGE_THREADS, z1 - z0) + 1).astype(int)
                    list(pool.map(lambda ab: np.copyto(host[ab[0]:ab[1]], arr[z0 + ab[0]:z0 + ab[1]]),
                                  zip(cuts[:-1], cuts[1:])))
                    with torch.cuda.stream(self.stream):
                        self.out[z0:z1].copy_(buf, non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record()
                    done[k & 1] = ev
                    with self.cv:
                        self.bounds.append(z1)
                        self.events.append(ev)
                        self.cv.notify_all()
            for ev in done:
                if ev is not None:
                    ev.synchronize()               # (the buffers go back only when the DMA has read them)
            with _STAGING_LOCK:
                if len(_STAGING) > 4:
                    _STAGING.clear()
                if len(_STAGING.setdefault(key, [])) < 2:
                    _STAGING[key].append(stage)
        except BaseException as exc:               # (reported by whoever waits for a slab)
            with self.cv:
                self.error = exc
                self.cv.notify_all()

    def event_for(self, z_hi: int):
        """The event after which planes ``[0, z_hi)`` are on the device (waits until its copy has been queued)."""
        z_hi = max(0, min(int(z_hi), self.nz))
        with self.cv:
            while True:
                if self.error is not None:
                    raise nat.MmxError(f"upload of the image failed: {self.error!r}") from self.error
                i = bisect.bisect_left(self.bounds, z_hi)
                if i < len(self.events):
                    return self.events[i]
                if self.all_queued():
                    return self.events[-1]
                self.cv.wait(0.5)


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


