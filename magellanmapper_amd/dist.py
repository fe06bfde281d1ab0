"""Multi-GPU sharding of blocks and the blob-table gather (new; the reference has a TODO
at magmap/cv/stack_detect.py:406-407 where distribution would go).

One process per GPU, launched with ``torch.distributed.run``.  Blocks are independent
units (each is filtered on its own extent), so rank ``r`` of ``N``
takes a contiguous z-major range of the block grid and no data-path collective is needed
until the end, when every rank's per-block tables are exchanged with ONE padded
``all_gather`` (RCCL over xGMI on GPUs; a few MB, latency bound) so that the overlap
de-duplication can run on the merged table.  Without an initialised process group all
functions degrade to the single-process case.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

try:
    import torch
    import torch.distributed as tdist
except Exception:  # pragma: no cover
    torch = None
    tdist = None


def _active() -> bool:
    return tdist is not None and tdist.is_available() and tdist.is_initialized()


_solo = 0


def rank() -> int:
    if _solo:
        return 0
    if _loopback is not None:
        return _loopback.me
    return tdist.get_rank() if _active() else 0


def world_size() -> int:
    if _solo:
        return 1
    if _loopback is not None:
        return _loopback.world
    return tdist.get_world_size() if _active() else 1


class Loopback:
    """ONE process standing in for rank ``me`` of ``world`` with a recording in place of the wire (``bench.py --share
    k/N``, tests): ``rank()`` / ``world_size()`` / ``my_share()`` answer as that rank's would, and every collective of a
    step -- the i-th ``all_reduce_sum`` / row exchange since :meth:`begin_step` -- hands back what the OTHER ranks
    contributed to their i-th collective when this same process played them (mode ``"record"``: the own contribution is
    stored; ranks not played yet contribute nothing).  Playing every rank twice fills the tape with what a real run
    would put on the wire (the first round records the rows near the seams, which the second round's pruning needs);
    ``"replay"`` then measures one rank's step with everything a rank does except the transfer itself -- and its merged
    table is the whole stack's.  Installed with :func:`set_loopback`; never active unless a caller installs it."""

    def __init__(self, me: int, world: int):
        if not 0 <= int(me) < int(world):
            raise ValueError(f"rank {me} of {world}")
        self.me, self.world, self.mode, self.at = int(me), int(world), "record", 0
        self.tape = {q: [] for q in range(self.world)}
        self._recv = {}                 # replay: the receive buffer of the i-th exchange, put together once

    def begin_step(self, me: Optional[int] = None) -> None:
        if me is not None:
            self.me = int(me)
        self.at = 0

    def exchange(self, own: np.ndarray):
        """The i-th collective of the step: every rank's contribution, in rank order (``None``: not recorded yet)."""
        i, self.at = self.at, self.at + 1
        if self.mode == "record":
            mine = self.tape[self.me]
            while len(mine) <= i:
                mine.append(None)
            mine[i] = np.array(own, copy=True)
        return [own if q == self.me else (self.tape[q][i] if i < len(self.tape[q]) else None)
                for q in range(self.world)]


    def gather_rows(self, rows: np.ndarray, n_cols: int):
        """A row exchange as ``_gather_rows`` returns it: ``(blocks (n_ranks, most, width) or None, counts, width)``.
        In replay the receive buffer of each exchange of the step is put together once and only this rank's own rows
        are copied into it afterwards -- what the real path does on the host too (own rows into the pinned send buffer;
        the other ranks' rows arrive by DMA)."""
        i = self.at
        parts = self.exchange(rows)
        counts = [0 if p is None else int(p.shape[0]) for p in parts]
        width = max([n_cols] + [int(p.shape[1]) for p in parts if p is not None and p.shape[0]])
        most = max(counts)
        if most == 0 or width == 0:
            return None, counts, width
        key = (i, tuple(c for r, c in enumerate(counts) if r != self.me), most, width)
        hit = self._recv.get(i) if self.mode == "replay" else None
        if hit is not None and hit[0] == key:
            bufs = hit[1]
            if counts[self.me]:
                bufs[self.me, :counts[self.me], :rows.shape[1]] = rows
            bufs[self.me, counts[self.me]:] = 0.0
            return bufs, counts, width
        bufs = np.zeros((len(parts), most, width))
        for r, p in enumerate(parts):
            if counts[r]:
                bufs[r, :counts[r], :p.shape[1]] = p
        if self.mode == "replay":
            self._recv[i] = (key, bufs)
        return bufs, counts, width


_loopback: Optional[Loopback] = None


def set_loopback(wire: Optional[Loopback]) -> None:
    global _loopback
    if wire is not None and _active():
        raise RuntimeError("a loop-back wire stands in for a process group: not with one initialised")
    _loopback = wire


class solo:
    """``with dist.solo():`` -- inside, this process works as if there were no process group (rank 0 of 1: all blocks
    are its own, no collective is entered).  For work ONE rank of a group does by itself while the others wait, e.g.
    the parity sample of ``bench.py`` on rank 0."""

    def __enter__(self):
        global _solo
        _solo += 1
        return self

    def __exit__(self, *exc):
        global _solo
        _solo -= 1
        return False


def share_bounds(n_items: int, r: int, n_ranks: int) -> Tuple[int, int]:
    """Contiguous, balanced range of ``n_items`` for rank ``r`` (first ranks get the extras)."""
    base, extra = divmod(n_items, n_ranks)
    lo = r * base + min(r, extra)
    return lo, lo + base + (1 if r < extra else 0)


def my_share(n_items: int) -> List[int]:
    lo, hi = share_bounds(n_items, rank(), world_size())
    return list(range(lo, hi))


def tile_share(n_tiles: int, r: Optional[int] = None, n_ranks: Optional[int] = None) -> List[int]:
    """Tiles of a tiled stack that rank ``r`` detects when the stack is sharded BY TILE (``stack_detect.
    detect_blobs_tiles(shard="tiles")``, BASELINE.json configs[4]): tiles r, r + N, r + 2N, ... -- round robin, so that
    the split needs no tile count up front (tiles may come from an iterator) and consecutive tiles of a rank are N
    apart on the stage: while one rank detects tile k the next ranks upload and detect k + 1 ... k + N - 1 on their own
    PCIe links.  Every tile is an image of its own (the reference makes one ``detect_blobs_blocks`` call per file,
    stack_detect.py:338-517), so this split has no exchange step at all."""
    r = rank() if r is None else int(r)
    n_ranks = world_size() if n_ranks is None else int(n_ranks)
    return list(range(r, int(n_tiles), max(1, n_ranks)))


def _device_for_collectives():
    if tdist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


_last_gather_ms = 0.0


def last_gather_ms() -> float:
    """Wall milliseconds this process spent exchanging tables since the last call: the most recent
    :func:`gather_tables`, or the collectives of a distributed pruning (:func:`all_gather_rows`,
    :func:`all_reduce_sum`); 0 without a group.  Reading resets the exchange clock."""
    global _exchange_ms
    out = _last_gather_ms + _exchange_ms
    _exchange_ms = 0.0
    return out


_exchange_ms = 0.0


def _timed(fn):
    """Add the wall time of a collective helper to the exchange clock."""
    import functools
    import time

    @functools.wraps(fn)
    def inner(*args, **kwargs):
        global _exchange_ms
        t0 = time.perf_counter()
        try:
            return fn(*args, **kwargs)
        finally:
            if _multi_rank():
                _exchange_ms += (time.perf_counter() - t0) * 1e3
    return inner


def gather_tables(local: Sequence[Tuple[int, Optional[np.ndarray]]], n_items: int, decode_on: Optional[int] = None,
                  raw: bool = False):
    """All ranks' ``(block_index, table | None)`` lists, merged and sorted by block index.

    Wire format: every table row is prefixed with its block index; ranks exchange row counts, column counts
    and one flag per block first (one small all_gather), then one all_gather of the row-padded float64 tables
    (two collectives per call, a few MB: latency bound).  ``decode_on``: only that rank unpacks the tables (the
    rank that prunes); the others get ``(index, None)`` placeholders back and skip the host work.
    ``raw`` (several ranks only): the decoding rank gets ``(block index per row, all rows in block order, indices
    of EMPTY blocks)`` instead of per-block tables -- what the pruning rank turns into its arena with a handful of
    whole-array copies instead of one Python iteration per block; other ranks get ``None``.
    """
    global _last_gather_ms
    _last_gather_ms = 0.0
    if not _multi_rank():
        return sorted(local, key=lambda e: e[0])
    _no_loopback("gather_tables")
    import time
    t_start = time.perf_counter()
    dev = _device_for_collectives()
    n_ranks = world_size()
    rows = [np.concatenate((np.full((len(t), 1), i, dtype=np.float64), t), axis=1)
            for i, t in local if t is not None and len(t)]
    mine = np.concatenate(rows) if rows else np.zeros((0, 0))
    # meta: [rows, wire columns, table columns, one flag per block of the share: 0 = None, 1 = EMPTY table
    # (a block whose blobs were all excluded: the reference keeps it, and a stack of nothing but such blocks
    # merges to an empty table, not to None), 2 = rows follow]
    max_share = -(-n_items // n_ranks)
    lo = share_bounds(n_items, rank(), n_ranks)[0]
    widths = {t.shape[1] for _, t in local if t is not None}
    if len(widths) > 1:
        raise ValueError(f"block tables of one rank differ in width: {sorted(widths)}")
    tcols = max(widths or [0])
    meta_np = np.zeros(3 + max_share, dtype=np.int64)
    meta_np[:3] = (mine.shape[0], mine.shape[1], tcols)
    for i, t in local:
        meta_np[3 + i - lo] = 0 if t is None else (2 if len(t) else 1)
    meta = torch.from_numpy(meta_np).to(dev)
    metas = [torch.zeros_like(meta) for _ in range(n_ranks)]
    tdist.all_gather(metas, meta)
    metas = torch.stack(metas).cpu().numpy()
    max_rows = int(metas[:, 0].max())
    n_cols = int(metas[:, 1].max())
    # every rank that holds rows must hold them at the same width (co-localisation columns are appended to
    # every block table or to none): a narrower table would come back padded with zero columns
    have = metas[metas[:, 0] > 0, 1]
    if len(have) and not np.all(have == have[0]):
        raise ValueError(f"ranks hold block tables of different widths: {sorted(set(int(v) - 1 for v in have))}")
    decode = decode_on is None or decode_on == rank()
    if raw:
        if recv_needed := (max_rows and n_cols):
            padded = np.zeros((max_rows, n_cols))
            padded[:mine.shape[0], :mine.shape[1]] = mine
            send = torch.from_numpy(padded).to(dev)
            recv = [torch.empty_like(send) for _ in range(n_ranks)]
            tdist.all_gather(recv, send)
        result = None
        if decode:
            empties = [share_bounds(n_items, r, n_ranks)[0] + int(pos)
                       for r, m in enumerate(metas) for pos in np.nonzero(m[3:] == 1)[0]]
            width = int(metas[:, 2].max())
            if recv_needed:
                bufs = torch.stack(recv).cpu().numpy()
                parts = [b[:int(m[0]), :int(m[1])] for m, b in zip(metas, bufs) if m[0]]
                rows = np.concatenate(parts) if len(parts) > 1 else parts[0]
                result = (rows[:, 0].astype(np.int64), np.ascontiguousarray(rows[:, 1:]), empties)
            else:
                result = (np.zeros(0, dtype=np.int64), np.zeros((0, width)), empties)
        _last_gather_ms = (time.perf_counter() - t_start) * 1e3
        return result
    out: List[Tuple[int, Optional[np.ndarray]]] = []
    recv = None
    if max_rows and n_cols:
        padded = np.zeros((max_rows, n_cols))
        padded[:mine.shape[0], :mine.shape[1]] = mine
        send = torch.from_numpy(padded).to(dev)
        recv = [torch.empty_like(send) for _ in range(n_ranks)]
        tdist.all_gather(recv, send)
    if decode:
        for r, m in enumerate(metas):
            lo_r = share_bounds(n_items, r, n_ranks)[0]
            for pos in np.nonzero(m[3:] == 1)[0]:
                out.append((lo_r + int(pos), np.zeros((0, int(m[2])))))
        if recv is not None:
            bufs = torch.stack(recv).cpu().numpy()                  # one device -> host copy
            for m, tbl in zip(metas, bufs):
                tbl = tbl[:int(m[0]), :int(m[1])]                   # this rank's own rows and width
                if len(tbl):
                    idx = tbl[:, 0].astype(np.int64)
                    # rows of one block are contiguous and in order
                    cuts = np.nonzero(np.diff(idx))[0] + 1
                    for part in np.split(tbl, cuts):
                        out.append((int(part[0, 0]), np.ascontiguousarray(part[:, 1:])))
    present = {i for i, _ in out}
    out.extend((i, None) for i in range(n_items) if i not in present)
    _last_gather_ms = (time.perf_counter() - t_start) * 1e3
    return sorted(out, key=lambda e: e[0])


def raise_together(failure: Optional[BaseException], what: str = "a rank") -> None:
    """Agree on whether any rank failed before the next collective: re-raises ``failure`` on the rank that holds it
    and raises ``RuntimeError`` on all the others, instead of leaving them waiting in a collective the failed rank
    never enters (one tiny all_reduce; nothing without a process group beyond re-raising)."""
    if _multi_rank() and _loopback is None:
        flag = torch.tensor([0 if failure is None else 1], dtype=torch.int64, device=_device_for_collectives())
        tdist.all_reduce(flag, op=tdist.ReduceOp.MAX)
        if int(flag.item()) and failure is None:
            raise RuntimeError(f"{what} failed on another rank; see its log")
    if failure is not None:
        raise failure


def _no_loopback(what: str) -> None:
    if _loopback is not None:
        raise NotImplementedError(f"{what} has no loop-back form: a Loopback wire stands in for the distributed pruning "
                                  "of a regular block geometry (row exchanges and all_reduce_sum) only")


def _multi_rank() -> bool:
    """True when collectives have somebody to talk to.  ``_force_collectives`` (a test hook, never set by the
    product) sends a ONE-rank group through the collective code as well, so that a single GPU can execute the RCCL
    branches: pinned staging, ``all_gather_into_tensor`` on device tensors, the stream ordering around them."""
    return (_loopback is not None and not _solo) or (_active() and (world_size() > 1 or _force_collectives))


_force_collectives = False


@_timed
def _gather_rows(rows: np.ndarray, n_cols: Optional[int], failure: Optional[BaseException] = None,
                 what: str = "a rank"):
    """``(buffers (n_ranks, most, width) or None, per-rank row counts, width)``: every rank's rows, padded to the
    largest count (two collectives: the counts, then the rows).  ``failure``: what this rank's preparation of the
    rows raised, if anything -- it travels with the counts, so that every rank raises before the second collective
    instead of waiting in it for a rank that never comes."""
    if _loopback is not None:
        if failure is not None:
            raise failure
        return _loopback.gather_rows(rows, int(n_cols or 0))
    dev = _device_for_collectives()
    n_ranks = world_size()
    meta = torch.tensor([rows.shape[0], rows.shape[1] if rows.shape[0] else int(n_cols or 0),
                         0 if failure is None else 1], dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(n_ranks)]
    tdist.all_gather(metas, meta)
    metas = torch.stack(metas).cpu().numpy()
    if failure is not None:
        raise failure
    if metas[:, 2].any():
        raise RuntimeError(f"{what} failed on rank {int(np.flatnonzero(metas[:, 2])[0])}; see its log")
    width = int(metas[:, 1].max())
    have = metas[metas[:, 0] > 0, 1]
    if len(have) and not np.all(have == have[0]):
        raise ValueError(f"ranks hold rows of different widths: {sorted(set(int(v) for v in have))}")
    counts = [int(v) for v in metas[:, 0]]
    most = max(counts)
    if most == 0 or width == 0:
        return None, counts, width
    if dev.type == "cpu":
        padded = np.zeros((most, width))
        padded[:rows.shape[0], :rows.shape[1]] = rows
        send = torch.from_numpy(padded)
        recv = torch.empty((n_ranks * most, width), dtype=torch.float64)      # (concatenated along the rows)
        tdist.all_gather_into_tensor(recv, send)
        bufs = recv.numpy().reshape(n_ranks, most, width)
    else:
        # device collectives (RCCL): pinned staging both ways -- pageable copies of tens of MB would cost more than
        # the exchange itself -- and one gather into one tensor
        stage = _pinned("send", most * width)
        host = stage.numpy()[:most * width].reshape(most, width)
        host[:rows.shape[0], :rows.shape[1]] = rows
        host[rows.shape[0]:] = 0.0
        send = torch.empty((most, width), dtype=torch.float64, device=dev)
        send.copy_(stage[:most * width].view(most, width), non_blocking=True)
        recv = torch.empty((n_ranks * most, width), dtype=torch.float64, device=dev)
        tdist.all_gather_into_tensor(recv, send)
        back = _pinned("recv", n_ranks * most * width)
        back[:n_ranks * most * width].view(n_ranks * most, width).copy_(recv, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        bufs = back.numpy()[:n_ranks * most * width].reshape(n_ranks, most, width)
    return bufs, counts, width


def _as_rows(rows: np.ndarray) -> np.ndarray:
    rows = np.ascontiguousarray(rows, dtype=np.float64)
    if rows.ndim != 2:
        rows = rows.reshape(len(rows), -1) if rows.size else np.zeros((0, 0))
    return rows


def all_gather_rows(rows: Optional[np.ndarray], n_cols: Optional[int] = None,
                    failure: Optional[BaseException] = None, what: str = "a rank") -> List[np.ndarray]:
    """Every rank's ``(n_r, c)`` float64 array, in rank order (two collectives: the row counts, then the rows padded
    to the largest count).  Ranks without rows may pass ``n_cols=None`` / a ``(0, 0)`` array: the width is agreed on
    first.  ``failure``: an exception this rank met while making ``rows`` (which may then be ``None``); every rank
    raises.  Without a process group: ``[rows]``."""
    rows = _as_rows(np.zeros((0, 0)) if rows is None else rows)
    if not _multi_rank():
        if failure is not None:
            raise failure
        return [rows]
    bufs, counts, width = _gather_rows(rows, n_cols, failure, what)
    if bufs is None:
        return [np.zeros((0, width)) for _ in counts]
    return [np.array(bufs[r, :n]) for r, n in enumerate(counts)]


def all_gather_rows_concat(rows: Optional[np.ndarray], n_cols: Optional[int] = None,
                           failure: Optional[BaseException] = None, what: str = "a rank") -> np.ndarray:
    """The same exchange with every rank's rows back to back in ONE array, rank after rank (one copy out of the
    receive buffer instead of a copy per rank and a concatenation: 2 ms for the 21 MB of a pruned benchmark table)."""
    rows = _as_rows(np.zeros((0, 0)) if rows is None else rows)
    if not _multi_rank():
        if failure is not None:
            raise failure
        return rows
    bufs, counts, width = _gather_rows(rows, n_cols, failure, what)
    if bufs is None:
        return np.zeros((0, width))
    return np.concatenate([bufs[r, :n] for r, n in enumerate(counts)])


def all_gather_rows_padded(rows: Optional[np.ndarray], n_cols: Optional[int] = None,
                           failure: Optional[BaseException] = None, what: str = "a rank"):
    """The same exchange, handed over as it arrives: ``(blocks (n_ranks, most, width) or None, per-rank row counts,
    width)`` -- rank ``r``'s rows are ``blocks[r, :counts[r]]``.  The blocks live in this module's receive buffer and
    are valid until the next exchange: for a consumer that reads them once (the merge by key takes them as they are,
    which spares the copy that makes them one array: 20 MB for a pruned benchmark table)."""
    rows = _as_rows(np.zeros((0, 0)) if rows is None else rows)
    if not _multi_rank():
        if failure is not None:
            raise failure
        return (rows[None] if rows.size else None), [len(rows)], rows.shape[1] if rows.size else int(n_cols or 0)
    return _gather_rows(rows, n_cols, failure, what)


def gather_tile_tables(local: Sequence[Tuple], failure: Optional[BaseException] = None):
    """Every rank's per-tile tables on every rank: ``local`` = this rank's ``(tile index, table | None[, tag])`` items
    (float64 2-D tables of any number of rows, one common width among the tables that hold rows; ``tag``: one integer
    per tile that travels with it, 0 if absent) -> the items of ALL ranks, sorted by tile index.  ``None`` stays
    ``None`` and a table without rows keeps its width.  Two exchanges (one row of three numbers per tile, then the rows
    prefixed with their tile index); without a process group: ``local``, sorted.  Collective -- every rank calls it, also
    a rank that detected no tile (``local = []``); ``failure``: what this rank's detection raised, if anything -- it
    travels with the first exchange, so that every rank raises instead of waiting for a rank that never comes."""
    items = []
    for it in local:
        idx, tbl = int(it[0]), it[1]
        tag = int(it[2]) if len(it) > 2 else 0
        if tbl is not None:
            tbl = np.ascontiguousarray(tbl, dtype=np.float64)
            if tbl.ndim != 2:
                raise ValueError(f"tile {idx}: a table must be 2-D, not {tbl.shape}")
        items.append((idx, tbl, tag))
    items.sort(key=lambda e: e[0])
    if not _multi_rank():
        if failure is not None:
            raise failure
        return items
    meta = np.array([[i, -1 if t is None else len(t), 0 if t is None else t.shape[1], tag] for i, t, tag in items],
                    dtype=np.float64).reshape(-1, 4)
    metas = all_gather_rows(meta, 4, failure, "detection of a rank's tiles")
    widths = sorted({int(m[2]) for part in metas for m in part if m[1] > 0})
    if len(widths) > 1:         # (every rank sees the same metas: all raise, nobody waits in the second exchange)
        raise ValueError(f"tile tables differ in width: {widths}")
    width = widths[0] if widths else 0
    rows = [np.concatenate((np.full((len(t), 1), float(i)), t), axis=1) for i, t, _ in items if t is not None and len(t)]
    mine = np.concatenate(rows) if rows else np.zeros((0, 0))
    parts = all_gather_rows(mine, width + 1)
    out = []
    for part_meta, rows_r in zip(metas, parts):
        at = 0
        for i, n, w, tag in part_meta:
            i, n, w = int(i), int(n), int(w)
            if n < 0:
                out.append((i, None, int(tag)))
            elif n == 0:
                out.append((i, np.zeros((0, w)), int(tag)))
            else:
                block = rows_r[at:at + n]
                if len(block) != n or not np.all(block[:, 0] == i):
                    raise RuntimeError(f"tile {i}: the exchange delivered other rows than announced")
                out.append((i, np.ascontiguousarray(block[:, 1:1 + w]), int(tag)))
                at += n
    out.sort(key=lambda e: e[0])
    seen = [i for i, _, _ in out]
    if len(set(seen)) != len(seen):
        raise ValueError(f"tile indices held by more than one rank: {sorted(i for i in set(seen) if seen.count(i) > 1)}")
    return out


_pinned_bufs = {}


def _pinned(name: str, n: int):
    """A pinned float64 staging buffer of at least ``n`` elements (kept between calls)."""
    buf = _pinned_bufs.get(name)
    if buf is None or buf.numel() < n:
        buf = _pinned_bufs[name] = torch.empty(max(int(n * 1.25), 1024), dtype=torch.float64).pin_memory()
    return buf


@_timed
def all_reduce_sum(values: np.ndarray) -> np.ndarray:
    """Element-wise sum of an int64 array over the ranks (the array itself without a process group)."""
    values = np.ascontiguousarray(values, dtype=np.int64)
    if not _multi_rank():
        return values
    if _loopback is not None:
        return np.sum([p for p in _loopback.exchange(values) if p is not None and p.shape == values.shape], axis=0)
    t = torch.from_numpy(values.copy()).to(_device_for_collectives())
    tdist.all_reduce(t, op=tdist.ReduceOp.SUM)
    return t.cpu().numpy()


def broadcast_table(table: Optional[np.ndarray], src: int = 0) -> Optional[np.ndarray]:
    """``table`` of rank ``src`` (a float64 2-D array or ``None``) on every rank."""
    if not _multi_rank():
        return table
    _no_loopback("broadcast_table")
    dev = _device_for_collectives()
    is_src = rank() == src
    meta = torch.tensor([-1, 0] if (not is_src or table is None) else list(table.shape), dtype=torch.int64, device=dev)
    tdist.broadcast(meta, src)
    rows, cols = (int(v) for v in meta.cpu())
    if rows < 0:
        return None
    buf = (torch.from_numpy(np.ascontiguousarray(table, dtype=np.float64)).to(dev) if is_src
           else torch.empty((rows, cols), dtype=torch.float64, device=dev))
    if rows * cols:
        tdist.broadcast(buf, src)
    return table if is_src else buf.cpu().numpy()
