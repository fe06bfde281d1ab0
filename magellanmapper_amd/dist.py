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


def rank() -> int:
    return tdist.get_rank() if _active() else 0


def world_size() -> int:
    return tdist.get_world_size() if _active() else 1


def share_bounds(n_items: int, r: int, n_ranks: int) -> Tuple[int, int]:
    """Contiguous, balanced range of ``n_items`` for rank ``r`` (first ranks get the extras)."""
    base, extra = divmod(n_items, n_ranks)
    lo = r * base + min(r, extra)
    return lo, lo + base + (1 if r < extra else 0)


def my_share(n_items: int) -> List[int]:
    lo, hi = share_bounds(n_items, rank(), world_size())
    return list(range(lo, hi))


def _device_for_collectives():
    if tdist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def gather_tables(local: Sequence[Tuple[int, Optional[np.ndarray]]], n_items: int
                  ) -> List[Tuple[int, Optional[np.ndarray]]]:
    """All ranks' ``(block_index, table | None)`` lists, merged and sorted by block index.

    Wire format: every table row is prefixed with its block index; ranks exchange row
    counts and column counts first (one small all_gather), then one all_gather of the
    row-padded float64 tables.
    """
    if not _active() or world_size() == 1:
        return sorted(local, key=lambda e: e[0])
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
    tcols = max([t.shape[1] for _, t in local if t is not None] or [0])
    meta_np = np.zeros(3 + max_share, dtype=np.int64)
    meta_np[:3] = (mine.shape[0], mine.shape[1], tcols)
    for i, t in local:
        meta_np[3 + i - lo] = 0 if t is None else (2 if len(t) else 1)
    meta = torch.from_numpy(meta_np).to(dev)
    metas = [torch.zeros_like(meta) for _ in range(n_ranks)]
    tdist.all_gather(metas, meta)
    metas = [m.cpu().numpy() for m in metas]
    max_rows = int(max(m[0] for m in metas))
    n_cols = int(max(m[1] for m in metas))
    out: List[Tuple[int, Optional[np.ndarray]]] = []
    for r, m in enumerate(metas):
        lo_r = share_bounds(n_items, r, n_ranks)[0]
        for pos in np.nonzero(m[3:] == 1)[0]:
            out.append((lo_r + int(pos), np.zeros((0, int(m[2])))))
    if max_rows and n_cols:
        padded = np.zeros((max_rows, n_cols))
        padded[:mine.shape[0], :mine.shape[1]] = mine
        send = torch.from_numpy(padded).to(dev)
        recv = [torch.empty_like(send) for _ in range(n_ranks)]
        tdist.all_gather(recv, send)
        for m, buf in zip(metas, recv):
            tbl = buf.cpu().numpy()[:int(m[0])]
            if len(tbl):
                idx = tbl[:, 0].astype(np.int64)
                # rows of one block are contiguous and in order
                cuts = np.nonzero(np.diff(idx))[0] + 1
                for part in np.split(tbl, cuts):
                    out.append((int(part[0, 0]), np.ascontiguousarray(part[:, 1:])))
    present = {i for i, _ in out}
    out.extend((i, None) for i in range(n_items) if i not in present)
    return sorted(out, key=lambda e: e[0])
