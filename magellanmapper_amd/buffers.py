"""Per-device buffers, streams and events of the batched detection (``_Buffers``), the pinned upload ring for small
host -> device copies and the event wrappers.  Split out of ``blob_log.py`` (round 5)."""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional

import numpy as np

from . import _native as nat

try:
    import torch
except Exception as exc:  # pragma: no cover
    raise ImportError("magellanmapper_amd needs PyTorch-ROCm for device memory and streams") from exc


def _pipeline():
    from . import blob_log
    return blob_log


#: candidate-table entries copied to pinned host memory together with the counts, before the host knows how many
#: there are (a batch of the benchmark volume holds ~3e4; more entries cost a second, synchronous copy)
_PREFIX_ENTRIES = 1 << 16


class _NativeEvent:
    """A HIP event the library records (``mmx_detect_batch``: ``ev_done`` / ``ev_work_read``); what this host needs of
    ``torch.cuda.Event``: ``synchronize`` and being waited for by a stream (:func:`_stream_wait`)."""
    __slots__ = ("handle",)

    def __init__(self):
        h = ctypes.c_void_p()
        nat.check(nat.lib().mmx_event_create(ctypes.byref(h)), "mmx_event_create")
        self.handle = h.value

    def synchronize(self) -> None:
        nat.check(nat.lib().mmx_event_synchronize(self.handle), "mmx_event_synchronize")

    def query(self) -> bool:
        rc = nat.lib().mmx_event_query(self.handle)
        if rc > 1:
            nat.check(rc, "mmx_event_query")
        return rc == 0

    def __del__(self):
        try:
            if self.handle:
                nat.lib().mmx_event_destroy(self.handle)
        except Exception:       # (interpreter shutdown)
            pass


def _stream_wait(stream, event) -> None:
    """``stream.wait_event(event)`` for torch events and for the library's own."""
    if isinstance(event, _NativeEvent):
        nat.check(nat.lib().mmx_stream_wait_event(stream.cuda_stream, event.handle), "mmx_stream_wait_event")
    else:
        stream.wait_event(event)


class _Buffers:
    """Device scratch that is reused across the batches of one call.

    ``main`` is the caller's stream: the float32 passes, the NMS and the bulk re-score of
    batch k are enqueued there back to back.  ``side`` is a high-priority stream for the
    small follow-up work of batch k-1 (copying its candidates out, re-scoring the neighbours
    of contested candidates, the overlap-pair search), which therefore overlaps the heavy
    kernels of batch k instead of queueing behind them.
    """

    def __init__(self, dev):
        self.dev = dev
        self.ws = None
        self.ws2 = None
        self.ws_free = [None, None]        # events: the last reader of each workspace (the NMS of a batch) is done
        self.cands = []
        self.counts = []
        self.host_counts = []
        self.host_tabs = []
        self.side = torch.cuda.Stream(device=dev, priority=-1)
        # per-block preprocessing (float64 vector arithmetic) of batch k + 1 runs here, beside the LoG kernels of
        # batch k (bound by memory requests) on the caller's stream
        self.pre_stream = torch.cuda.Stream(device=dev)
        self.rescore_stream = torch.cuda.Stream(device=dev, priority=0)
        self.pack_stream = torch.cuda.Stream(device=dev)
        self.native_events = []            # per candidate-table slot: (workspace read, batch done)
        self.graphs = {}                   # captured small batches: key -> (graph handle, mmx_detect_info, keep-alives)
        self.graph_stream = None           # where they run when the caller is on the (uncapturable) default stream
        self.plans = {}                    # batch plans + uploaded block tables of recent (block lists, volume layout)
        self.plan_lists = {}               # (id(origins), id(shapes)) -> (the lists, their content key)
        self.slots(2)

    def slots(self, n: int):
        """At least ``n`` candidate-table slots (one per batch in flight)."""
        while len(self.cands) < n:
            self.cands.append(None)
            # [0]: entries in the table (candidates + probes, counts past the capacity); [1]: candidates among them
            self.counts.append(torch.zeros(2, dtype=torch.int32, device=self.dev))
            self.host_counts.append(torch.zeros(2, dtype=torch.int32).pin_memory())
            self.host_tabs.append(None)
            self.native_events.append(None)

    def workspace(self, n_floats: int, which: int = 0):
        """Workspace ``which`` (0: the only one of most paths; 1: the second of the two that batches of a raw volume
        alternate between, so that the NMS and re-score of one batch run beside the LoG kernels of the next)."""
        if which == 0:
            if self.ws is None or self.ws.numel() < n_floats:
                self.ws = None
                self.ws = torch.empty(n_floats, dtype=torch.float32, device=self.dev)
            return self.ws
        if self.ws2 is None or self.ws2.numel() < n_floats:
            self.ws2 = None
            self.ws2 = torch.empty(n_floats, dtype=torch.float32, device=self.dev)
        return self.ws2

    def events(self, which: int):
        """``(workspace read, batch done)`` events of slot ``which`` (made once, recorded again by every batch that takes
        the slot -- which happens only after the previous holder has been waited for)."""
        if self.native_events[which] is None:
            self.native_events[which] = (_NativeEvent(), _NativeEvent())
        return self.native_events[which]

    def drop_graphs(self) -> None:
        for hit in self.graphs.values():
            if hit and hit != "plain":
                nat.lib().mmx_graph_destroy(hit[0])
        self.graphs = {}

    def host_table(self, which: int):
        """Pinned staging for the first ``_PREFIX_ENTRIES`` entries of slot ``which``'s candidate table."""
        if self.host_tabs[which] is None:
            self.host_tabs[which] = torch.empty(_PREFIX_ENTRIES * nat.CAND_DTYPE.itemsize, dtype=torch.uint8).pin_memory()
        return self.host_tabs[which]

    def cand_table(self, which: int, cap: int):
        need = cap * nat.CAND_DTYPE.itemsize
        if self.cands[which] is None or self.cands[which].numel() < need:
            self.cands[which] = None
            self.cands[which] = torch.empty(need, dtype=torch.uint8, device=self.dev)
        return self.cands[which]


_BUFFERS: Dict[str, _Buffers] = {}


def _buffers_for(dev) -> _Buffers:
    """The per-device scratch, created once: the side stream, the pinned count words and the workspace
    cost ~25 ms to set up (pinned allocations, stream creation), a tenth of a whole benchmark volume."""
    key = str(dev)
    if key not in _BUFFERS:
        if os.environ.get("MMX_SELF_TEST", "1") != "0":
            _pipeline().self_test(dev)
        _BUFFERS[key] = _Buffers(dev)
    return _BUFFERS[key]



def release_buffers() -> None:
    """Drop the cached device scratch (workspace, candidate tables, captured graphs) of every device."""
    for b in _BUFFERS.values():
        b.drop_graphs()
    _BUFFERS.clear()
    from . import preprocess, volume
    preprocess.release_retained()
    volume.release_staging()        # (pinned staging buffers of host-volume uploads, the copy streams)


class _UploadRing:
    """Small host -> device uploads (block tables, quantile classes, row offsets) that do not stall the host: a
    pageable ``tensor.to(device)`` waits for everything queued on the stream before it -- with kernels of a few
    milliseconds queued that is a few milliseconds per table, a dozen times per batch on the preprocessing and
    co-localisation paths, and the GPU then idles while the host catches up.  Here the bytes go through a ring of
    pinned slots and an asynchronous copy on the current stream; a slot is reused only after its copy has completed."""
    SLOTS, SLOT_BYTES = 64, 1 << 17

    def __init__(self):
        self.stage = torch.empty(self.SLOTS * self.SLOT_BYTES, dtype=torch.uint8).pin_memory()
        self.host = self.stage.numpy()
        self.busy = [None] * self.SLOTS
        self.at = 0

    #: largest upload taken (a quarter of the ring: the co-localisation's blob rows of a batch are ~0.7 MB)
    MAX_BYTES = SLOTS // 4 * SLOT_BYTES

    def put(self, raw: np.ndarray, dev) -> "torch.Tensor":
        n = raw.size
        need = -(-n // self.SLOT_BYTES)                 # consecutive slots
        if self.at + need > self.SLOTS:
            self.at = 0
        k = self.at
        self.at = (k + need) % self.SLOTS
        for j in range(k, k + need):
            if self.busy[j] is not None:
                self.busy[j].synchronize()
        lo = k * self.SLOT_BYTES
        self.host[lo:lo + n] = raw
        out = torch.empty(n, dtype=torch.uint8, device=dev)
        out.copy_(self.stage[lo:lo + n], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        for j in range(k, k + need):
            self.busy[j] = ev
        return out


_UPLOAD: Optional[_UploadRing] = None


def _to_device_bytes(arr: np.ndarray, dev) -> "torch.Tensor":
    """``arr``'s bytes as a uint8 device tensor, uploaded on the current stream."""
    global _UPLOAD
    raw = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
    if 0 < raw.size <= _UploadRing.MAX_BYTES and getattr(dev, "type", str(dev)[:4]) == "cuda":
        if _UPLOAD is None:
            _UPLOAD = _UploadRing()
        return _UPLOAD.put(raw, dev)
    return torch.from_numpy(raw).to(dev)


def to_device(arr: np.ndarray, dev) -> "torch.Tensor":
    """``torch.from_numpy(arr).to(dev)`` without the stall of a pageable copy (small arrays: :class:`_UploadRing`)."""
    arr = np.ascontiguousarray(arr)
    t = _to_device_bytes(arr, dev)
    return t.view(getattr(torch, str(arr.dtype))).view(arr.shape) if arr.size else torch.from_numpy(arr).to(dev)


