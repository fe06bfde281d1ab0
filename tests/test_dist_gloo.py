"""world_size-2 test of block sharding + the blob-table gather on the gloo backend (CPU).
The GPU run uses the same code over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.multiprocessing as tmp  # noqa: E402

from conftest import ROOT  # noqa: E402
from magellanmapper_amd import dist  # noqa: E402


def test_share_bounds_cover_everything():
    for n in (0, 1, 7, 75, 256):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                lo, hi = dist.share_bounds(n, r, world)
                got += list(range(lo, hi))
            assert got == list(range(n))
            sizes = [np.diff(dist.share_bounds(n, r, world))[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def _tables_for(i):
    rng = np.random.default_rng(100 + i)
    n = int(rng.integers(0, 5))
    if i % 5 == 3:
        return np.zeros((0, 11))       # a block whose blobs were all excluded: EMPTY, not None
    return None if n == 0 else rng.integers(0, 50, (n, 11)).astype(np.float64) + 0.25 * i


def _worker(rank, world, port, n_items, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as td
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from magellanmapper_amd import dist as d
        mine = d.my_share(n_items)
        local = [(i, _tables_for(i)) for i in mine]
        merged = d.gather_tables(local, n_items)
        assert [i for i, _ in merged] == list(range(n_items))
        for i, t in merged:
            want = _tables_for(i)
            if want is None:
                assert t is None
            else:
                assert t is not None and t.shape == want.shape      # (0, 11) tables survive the gather
                np.testing.assert_array_equal(t, want)
        open(os.path.join(out_dir, f"ok{rank}"), "w").write(str(len(mine)))
    finally:
        td.destroy_process_group()


def test_gather_tables_two_ranks(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    tmp.spawn(_worker, args=(2, port, 11, str(tmp_path)), nprocs=2, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]
    assert int(open(tmp_path / "ok0").read()) + int(open(tmp_path / "ok1").read()) == 11


def test_single_process_passthrough():
    local = [(2, None), (0, np.ones((1, 11)))]
    out = dist.gather_tables(local, 3)
    assert [i for i, _ in out] == [0, 2]
    assert dist.rank() == 0 and dist.world_size() == 1 and dist.my_share(4) == [0, 1, 2, 3]
