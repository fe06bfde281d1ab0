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


def _worker_rows(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as td
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from magellanmapper_amd import dist as d

        def rows_of(r):          # rank 1 holds nothing; the others different numbers of rows
            n = 0 if r == 1 else 3 + 2 * r
            return (np.arange(n * 5, dtype=np.float64).reshape(n, 5) + 1000.0 * r) if n else np.zeros((0, 0))
        parts = d.all_gather_rows(rows_of(rank), 5)
        joined = d.all_gather_rows_concat(rows_of(rank), 5)
        assert len(parts) == world
        for r, p in enumerate(parts):
            want = rows_of(r)
            assert p.shape == ((0, 5) if want.size == 0 else want.shape)
            if want.size:
                np.testing.assert_array_equal(p, want)
        np.testing.assert_array_equal(joined, np.concatenate([p for p in parts]))
        assert joined.flags["C_CONTIGUOUS"] and joined.shape[1] == 5
        # nobody holds anything: an empty table of the agreed width
        assert d.all_gather_rows_concat(np.zeros((0, 0)), 7).shape == (0, 7)
        open(os.path.join(out_dir, f"rows{rank}"), "w").write("ok")
    finally:
        td.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,n_items", [(2, 11), (3, 10), (4, 13)])
def test_gather_tables_uneven_shares(tmp_path, world, n_items):
    """Uneven shares (10 blocks over 3 ranks, 13 over 4), blocks without a table and EMPTY tables."""
    tmp.spawn(_worker, args=(world, _free_port(), n_items, str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == [f"ok{r}" for r in range(world)]
    assert sum(int(open(tmp_path / f"ok{r}").read()) for r in range(world)) == n_items


# ---------------------------------------------------------------- the whole N-rank tail: gather -> prune on rank 0
GRID_SHAPE, SEGMENT = (70, 96, 100), 40


def _block_tables(seed=11, n_cols=11):
    """Per-block 11(+2)-column tables as detection would leave them: seeded blob centres, every block lists the
    centres inside its (overlapping) extent, so that duplicates sit in the overlaps; one block EMPTY, one None."""
    from magellanmapper_amd import config, stack_detect
    config.setup_roi_profiles(None)
    config.resolutions = np.array([[1.0, 1.0, 1.0]])
    config.roi_profile.update(segment_size=SEGMENT, denoise_size=None)
    blocks = stack_detect.setup_blocks(config.roi_profile, GRID_SHAPE)
    rng = np.random.default_rng(seed)
    centres = rng.integers(0, GRID_SHAPE, (1500, 3))
    tables = []
    grid = blocks.sub_roi_slices.shape
    for k, c in enumerate(np.ndindex(*grid)):
        slc = blocks.sub_roi_slices[c]
        lo = np.array([s.indices(n)[0] for s, n in zip(slc, GRID_SHAPE)])
        hi = np.array([s.indices(n)[1] for s, n in zip(slc, GRID_SHAPE)])
        inside = np.all((centres >= lo) & (centres < hi), axis=1)
        pts = centres[inside] + rng.integers(-1, 2, (int(inside.sum()), 3))      # each block sees it a voxel off
        pts = np.clip(pts, lo, hi - 1)
        if k == 4:
            tables.append(np.zeros((0, n_cols)))
            continue
        if k == 7 or not len(pts):
            tables.append(None)
            continue
        t = np.full((len(pts), n_cols), -1.0)
        t[:, 0:3] = pts
        t[:, 3] = 5.0
        t[:, 6] = 0
        t[:, 7:10] = pts
        if n_cols > 11:
            t[:, 11:] = rng.integers(0, 2, (len(pts), n_cols - 11))
        tables.append(t)
    return blocks, tables


def _prune(blocks, seg_rois):
    from magellanmapper_amd import stack_detect

    class Img:
        shape = GRID_SHAPE
    return stack_detect.StackPruner.prune_blobs_mp(Img, seg_rois, blocks.overlap, blocks.tol, blocks.sub_roi_slices,
                                                   blocks.sub_rois_offsets, [0], blocks.overlap_padding)[0]


def _worker_prune(rank, world, port, n_cols, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as td
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from magellanmapper_amd import dist as d, stack_detect
        blocks, tables = _block_tables(n_cols=n_cols)
        mine = d.my_share(len(tables))
        seg = stack_detect.StackDetector.assemble_seg_rois([(i, tables[i]) for i in mine],
                                                           blocks.sub_roi_slices.shape, n_cols - 11)
        pruned = _prune(blocks, seg) if rank == 0 else None
        pruned = d.broadcast_table(pruned)
        np.save(os.path.join(out_dir, f"pruned{rank}.npy"), pruned)
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize("world,n_cols", [(2, 11), (3, 11), (4, 13)])
def test_ranks_gather_then_rank0_prunes_like_one_process(tmp_path, world, n_cols):
    """Block tables sharded over 2-4 ranks (uneven shares, an EMPTY and a missing block, 13-column tables with
    co-localisation flags), gathered, pruned on rank 0 and broadcast: the same table, row for row, as one
    process pruning all blocks -- on every rank."""
    from magellanmapper_amd import stack_detect
    blocks, tables = _block_tables(n_cols=n_cols)
    seg = stack_detect.StackDetector.assemble_seg_rois(list(enumerate(tables)), blocks.sub_roi_slices.shape,
                                                       n_cols - 11)
    want = _prune(blocks, seg)
    assert want is not None and 200 < len(want) < sum(len(t) for t in tables if t is not None)
    tmp.spawn(_worker_prune, args=(world, _free_port(), n_cols, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / f"pruned{r}.npy"), want)


def test_gather_rejects_tables_of_different_widths():
    from magellanmapper_amd import dist as d
    with pytest.raises(ValueError):
        # (single process: the per-rank check; across ranks the meta exchange makes the same check)
        import torch.distributed as td
        td.init_process_group("gloo", rank=0, world_size=1, init_method=f"tcp://127.0.0.1:{_free_port()}")
        try:
            d._active()
            # world_size 1 short-circuits: call the width check through a 1-rank "multi-rank" path
            old = d.world_size
            d.world_size = lambda: 2
            try:
                d.gather_tables([(0, np.ones((1, 11))), (1, np.ones((1, 13)))], 2)
            finally:
                d.world_size = old
        finally:
            td.destroy_process_group()


def test_single_process_passthrough():
    local = [(2, None), (0, np.ones((1, 11)))]
    out = dist.gather_tables(local, 3)
    assert [i for i, _ in out] == [0, 2]
    assert dist.rank() == 0 and dist.world_size() == 1 and dist.my_share(4) == [0, 1, 2, 3]


# ---------------------------------------------------------------- distributed pruning: every rank prunes its own rows
def _worker_dist_prune(rank, world, port, case, out_dir, regions=False):
    sys.path.insert(0, ROOT)
    import torch.distributed as td
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from magellanmapper_amd import dist as d, stack_detect as sd
        if regions:     # every rank prunes its blocks region by region whatever the number of rows
            sd.StackPruner.REGION_MIN_ROWS = 0
        blocks, tables, shape, channels, n_extra = _dist_case(case)
        grid = blocks.sub_roi_slices.shape
        coords = list(np.ndindex(*grid))
        mine = d.my_share(len(coords))
        arena = sd._TableArena(11 + n_extra, len(mine))
        local = []
        for i in mine:
            tbl = tables[i]
            if tbl is not None and len(tbl):
                arena.add(coords[i], tbl)
            arena.landed()
            local.append((i, tbl))
        seg = sd.StackDetector.assemble_seg_rois(local, grid, n_extra, arena, local_only=True)
        assert seg.local_only and all(seg[coords[i]] is None for i in range(len(coords)) if i not in mine)
        region_runs = []
        run_all = sd._RegionPruner.run_all
        sd._RegionPruner.run_all = lambda self: (region_runs.append(len(self.regions)), run_all(self))[1]

        class Img:
            pass
        Img.shape = shape
        pruned, df = sd.StackPruner.prune_blobs_mp(Img, seg, blocks.overlap, blocks.tol, blocks.sub_roi_slices,
                                                   blocks.sub_rois_offsets, channels, blocks.overlap_padding)
        np.save(os.path.join(out_dir, f"pruned{rank}.npy"), np.zeros((0, 0)) if pruned is None else pruned)
        np.save(os.path.join(out_dir, f"ratios{rank}.npy"), np.zeros((0, 0)) if df is None else df.to_numpy())
        # the same collective with the table asked for in its final columns (what stack_detect._StackRun asks for)
        final, df2 = sd.StackPruner.prune_blobs_mp(Img, seg, blocks.overlap, blocks.tol, blocks.sub_roi_slices,
                                                   blocks.sub_rois_offsets, channels, blocks.overlap_padding,
                                                   final_form=True, untouched=True)
        assert (df2 is None) == (df is None) and (df is None or np.array_equal(df2.to_numpy(), df.to_numpy()))
        is_final = isinstance(final, sd._FinalTable)
        np.save(os.path.join(out_dir, f"final{rank}.npy"), np.zeros((0, 0)) if final is None else np.asarray(final))
        with open(os.path.join(out_dir, f"final{rank}.txt"), "w") as f:
            f.write(",".join(final.col_names) if is_final else "")
        if n_extra:     # ... and with the co-localisation columns announced: final columns + the flags' columns beside them
            fl, _ = sd.StackPruner.prune_blobs_mp(Img, seg, blocks.overlap, blocks.tol, blocks.sub_roi_slices,
                                                  blocks.sub_rois_offsets, channels, blocks.overlap_padding,
                                                  final_form=True, untouched=True, n_flag_cols=n_extra)
            assert fl is None or (isinstance(fl, sd._FinalTable) and fl.coloc_cols is not None)
            if fl is not None:
                np.save(os.path.join(out_dir, f"flagged{rank}.npy"), np.hstack((np.asarray(fl), fl.coloc_cols)))
        with open(os.path.join(out_dir, f"regions{rank}.txt"), "w") as f:
            f.write(" ".join(str(v) for v in region_runs))
    finally:
        td.destroy_process_group()


def _dist_case(case):
    """Block tables of a synthetic stack (test_host_logic._synthetic_block_tables: near-duplicates in the overlaps)
    with the oddities a stack can have: EMPTY and missing blocks, co-localisation columns, two channels, more ranks
    than z-layers, a stack whose blocks are all EMPTY / all missing."""
    from magellanmapper_amd import config, stack_detect as sd
    from test_host_logic import _synthetic_block_tables
    config.setup_roi_profiles(None)
    config.resolutions = np.array([[1.0, 1.0, 1.0]])
    shape, seg, channels, n_extra, n_blobs = (96, 150, 170), 40, [0], 0, 6000
    if case == "extra_columns_two_channels":
        channels, n_extra = [0, 1], 2
    elif case == "far_from_seam":
        shape, n_blobs = (150, 100, 120), 12000         # four z-layers of blocks: one or two per rank
    elif case == "flat":
        shape, seg, n_blobs = (30, 200, 260), 50, 3000
    elif case.startswith("c4_grid"):
        # BASELINE.json configs[3]'s partition: a 4 x 8 x 8 grid of blocks over 8 ranks = 32 blocks = HALF a z-layer
        # per rank, so rank seams run along y as well as z
        shape, seg, n_blobs = (160, 320, 320), 40, 60000
    config.roi_profile.update(segment_size=seg, denoise_size=None)
    blocks = sd.setup_blocks(config.roi_profile, shape)
    rng = np.random.default_rng(31)
    by_coord = _synthetic_block_tables(rng, shape, blocks, n_blobs, channels, n_extra=n_extra)
    tables = [by_coord[c] for c in np.ndindex(*blocks.sub_roi_slices.shape)]
    if case in ("holes", "extra_columns_two_channels"):
        tables[3] = np.zeros((0, 11 + n_extra))          # all blobs excluded: EMPTY
        tables[5] = None
        tables[len(tables) - 1] = None
    elif case == "second_half_none":
        # only the first rank(s) hold blobs: they receive no halo rows and their own rows ARE the whole local table
        for k in range(len(tables) // 2, len(tables)):
            tables[k] = None
    elif case == "one_rank_all_empty":
        # the last third of the blocks found blobs and excluded them all: EMPTY tables, no rows
        for k in range(2 * len(tables) // 3, len(tables)):
            tables[k] = np.zeros((0, 11))
    elif case == "far_from_seam":
        # no blob within reach of another rank's blocks: nobody sends or receives a halo row
        reach = 4 * int(np.max(blocks.tol)) + int(np.max(blocks.overlap)) + 2
        grid = blocks.sub_roi_slices.shape
        z_seams = [int(blocks.sub_rois_offsets[(j, 0, 0)][0]) for j in range(1, grid[0])]
        for k, tbl in enumerate(tables):
            if tbl is None:
                continue
            keep = np.all([np.abs(tbl[:, 0] - zs) > reach for zs in z_seams], axis=0)
            tables[k] = tbl[keep] if keep.any() else None
    elif case == "c4_grid_one_rank_without_blobs":
        for k in range(5 * 32, 6 * 32):                  # rank 5 of 8 finds nothing at all
            tables[k] = None
    elif case == "all_empty":
        tables = [np.zeros((0, 11)) if k % 2 else None for k in range(len(tables))]
    elif case == "nothing":
        tables = [None] * len(tables)
    return blocks, tables, shape, channels, n_extra


@pytest.mark.parametrize("world,case", [(2, "plain"), (3, "holes"), (4, "extra_columns_two_channels"), (4, "flat"),
                                        (3, "all_empty"), (2, "nothing"), (2, "second_half_none"),
                                        (3, "second_half_none"), (3, "one_rank_all_empty"), (2, "far_from_seam"),
                                        (4, "far_from_seam"), (2, "plain+regions"), (3, "holes+regions"),
                                        (2, "extra_columns_two_channels+regions"), (2, "flat+regions"),
                                        (3, "second_half_none+regions"), (2, "far_from_seam+regions"),
                                        (8, "c4_grid"), (8, "c4_grid_one_rank_without_blobs"), (8, "c4_grid+regions")])
def test_distributed_pruning_equals_one_process(tmp_path, world, case):
    """Every rank holds the tables of its own blocks only, prunes its own rows (three passes on its rows plus the
    other ranks' rows within reach of its blocks) and merges everybody's survivors by key: the table -- rows, order,
    averaged coordinates -- and the pruning-ratio statistics one process gets from the whole table, on every rank;
    uneven shares, EMPTY / missing blocks, 13-column two-channel tables, nothing at all.  ``+regions``: every rank prunes
    its blocks region by region on a few threads (what it does from ``StackPruner.REGION_MIN_ROWS`` own rows on)."""
    from magellanmapper_amd import stack_detect as sd
    case, _, regions = case.partition("+")
    blocks, tables, shape, channels, n_extra = _dist_case(case)
    seg = sd.StackDetector.assemble_seg_rois(list(enumerate(tables)), blocks.sub_roi_slices.shape, n_extra)

    class Img:
        pass
    Img.shape = shape
    want, df = sd.StackPruner.prune_blobs_mp(Img, seg, blocks.overlap, blocks.tol, blocks.sub_roi_slices,
                                             blocks.sub_rois_offsets, channels, blocks.overlap_padding)
    if case.startswith("c4_grid"):
        assert blocks.sub_roi_slices.shape == (4, 8, 8)             # 32 blocks a rank: half z-layers
    if case == "nothing":
        assert want is None
    elif case == "all_empty":
        assert want.shape == (0, 11)
    elif case == "far_from_seam":
        assert 300 < len(want)
    else:
        assert 1000 < len(want) < sum(len(t) for t in tables if t is not None)
    tmp.spawn(_worker_dist_prune, args=(world, _free_port(), case, str(tmp_path), bool(regions)), nprocs=world, join=True)
    runs = [(tmp_path / f"regions{r}.txt").read_text().split() for r in range(world)]
    if regions:         # (some rank did prune several regions side by side, in both collectives)
        assert any(len(v) == (3 if n_extra else 2) and int(v[0]) > 1 for v in runs), runs
    elif not case.startswith("c4_grid"):      # (tables of that size prune by regions on their own)
        assert not any(runs), runs
    for r in range(world):
        got = np.load(tmp_path / f"pruned{r}.npy")
        ratios = np.load(tmp_path / f"ratios{r}.npy")
        if want is None:
            assert got.size == 0 and got.shape == (0, 0)
            continue
        assert got.shape == want.shape
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(ratios.reshape(df.shape), df.to_numpy())
        # final_form: the reference's last two steps folded into the exchange (not with co-localisation columns)
        final = np.load(tmp_path / f"final{r}.npy")
        names = (tmp_path / f"final{r}.txt").read_text()
        if n_extra:
            assert names == ""
            np.testing.assert_array_equal(final, want)
            flagged = np.load(tmp_path / f"flagged{r}.npy")
            rel = want.copy()
            rel[:, 0:3] = rel[:, 7:10]
            np.testing.assert_array_equal(flagged[:, :8], rel[:, [0, 1, 2, 3, 4, 5, 6, 10]])
            np.testing.assert_array_equal(flagged[:, 8:], want[:, 10:10 + n_extra])
        else:
            from magellanmapper_amd import detector
            bb = detector.Blobs(want.copy())
            bb.replace_rel_with_abs_blob_coords(bb.blobs)
            want_final = bb.remove_abs_blob_coords(True)
            assert names.split(",") == list(bb.cols)
            np.testing.assert_array_equal(final, want_final.reshape(final.shape) if not len(want_final) else want_final)


@pytest.mark.parametrize("world", [2, 4])
def test_rows_of_all_ranks_back_to_back(tmp_path, world):
    """``dist.all_gather_rows`` / ``all_gather_rows_concat``: uneven shares, a rank without rows, nobody with rows -- the
    concatenated form is the per-rank form back to back (what the distributed pruning merges by key)."""
    tmp.spawn(_worker_rows, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"rows{r}").exists() for r in range(world))


# ---------------------------------------------------------------- a rank-local failure inside the collective pruning
def _worker_dist_failure(rank, world, port, stage, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as td
    from datetime import timedelta
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(seconds=60))
    try:
        from magellanmapper_amd import dist as d, stack_detect as sd
        blocks, tables, shape, channels, n_extra = _dist_case("plain")
        grid = blocks.sub_roi_slices.shape
        coords = list(np.ndindex(*grid))
        mine = d.my_share(len(coords))
        arena = sd._TableArena(11, len(mine))
        local = []
        for i in mine:
            if tables[i] is not None and len(tables[i]):
                arena.add(coords[i], tables[i])
            arena.landed()
            local.append((i, tables[i]))
        seg = sd.StackDetector.assemble_seg_rois(local, grid, 0, arena, local_only=True)
        if rank == 1:           # this rank alone fails, in the stage before the given collective
            def boom(*a, **k):
                raise sd.nat.MmxError("injected failure")
            setattr(sd.StackPruner, {"seam": "_seam_rows", "own": "_prune_own_rows"}[stage], staticmethod(boom))

        class Img:
            pass
        Img.shape = shape
        try:
            sd.StackPruner.prune_blobs_mp(Img, seg, blocks.overlap, blocks.tol, blocks.sub_roi_slices,
                                          blocks.sub_rois_offsets, channels, blocks.overlap_padding)
            verdict = "returned"
        except sd.nat.MmxError as exc:
            verdict = f"own:{exc}"
        except RuntimeError as exc:
            verdict = f"peer:{exc}"
        open(os.path.join(out_dir, f"verdict{rank}"), "w").write(verdict)
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize("stage", ["seam", "own"])
def test_a_failing_rank_makes_every_rank_raise(tmp_path, stage):
    """One rank fails inside the collective pruning -- before the first exchange, before the second: every rank
    raises at the next collective (the failed rank its own exception, the others a RuntimeError naming the cause)
    instead of waiting in a collective the failed rank never enters."""
    world = 3
    tmp.spawn(_worker_dist_failure, args=(world, _free_port(), stage, str(tmp_path)), nprocs=world, join=True)
    verdicts = [open(tmp_path / f"verdict{r}").read() for r in range(world)]
    assert verdicts[1].startswith("own:") and "injected failure" in verdicts[1]
    assert verdicts[0].startswith("peer:") and verdicts[2].startswith("peer:")


# ---------------------------------------------------------------- configs[4] sharded BY TILE (no exchange while detecting)
def _tile_result(k):
    """What the detection of tile ``k`` leaves: ``(final 8-column table | None, colocs | None)`` -- every fifth tile
    without blobs, every seventh with an EMPTY table, the others with a seeded number of rows and two flag columns."""
    rng = np.random.default_rng(900 + k)
    if k % 5 == 3:
        return None, None
    n = 0 if k % 7 == 5 else int(rng.integers(1, 40))
    table = rng.integers(0, 500, (n, 8)).astype(np.float64) + k
    return table, rng.integers(0, 2, (n, 2)).astype(np.uint8)


class _FakeTile:
    """Stands in for ``stack_detect.Image5d`` where there is no GPU: records who prefetched / released it."""
    def __init__(self, k, log):
        self.k, self.log, self.device_volume = k, log, None

    def prefetch(self, own_planes=False):
        self.log.append(("prefetch", self.k))
        self.device_volume = object()
        return self

    def release(self):
        self.log.append(("release", self.k))
        self.device_volume = None


def _worker_tiles(rank, world, port, n_tiles, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as td
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from magellanmapper_amd import config, detector, dist as d, stack_detect as sd
        config.resolutions = np.array([[1.0, 1.0, 1.0]])
        mine = list(range(rank, n_tiles, world))
        assert d.tile_share(n_tiles) == mine
        log = []
        tiles = [_FakeTile(k, log) for k in range(n_tiles)]
        seen_names = []

        def fake_detect(base, tile, offset, size, channels, verify, save_dfs, full_roi, coloc):
            # one process' path: inside the call this rank is alone (no collective can be entered by mistake)
            assert d.world_size() == 1 and d.rank() == 0 and full_roi and tile.device_volume is not None
            seen_names.append(base)
            table, colocs = _tile_result(tile.k)
            blobs = detector.Blobs(None)
            if table is not None:
                blobs.cols = [c.value for c in detector.Blobs.Cols if not c.name.startswith("ABS_")]
                blobs.blobs, blobs.colocalizations = table, colocs
            return None, None, blobs
        sd.detect_blobs_blocks = fake_detect
        names = (f"tile{k}" for k in range(n_tiles))                 # an ITERATOR of names: advanced past foreign tiles
        got = list(sd.detect_blobs_tiles(names, iter(tiles), coloc=True, shard="tiles"))
        assert d.world_size() == world                               # (solo has been left)
        assert [k for k, _ in got] == mine and seen_names == [f"tile{k}" for k in mine]
        # only this rank's tiles were touched, tile k + N announced before tile k was detected, each released once
        assert {k for _, k in log} == set(mine)
        assert [e for e in log if e[0] == "release"] == [("release", k) for k in mine]
        for a, b in zip(mine, mine[1:]):
            assert log.index(("prefetch", b)) < log.index(("release", a))
        everything = sd.gather_tiles(got)
        assert [k for k, _ in everything] == list(range(n_tiles))
        for k, blobs in everything:
            table, colocs = _tile_result(k)
            if table is None:
                assert blobs.blobs is None and blobs.colocalizations is None
            else:
                assert blobs.blobs.shape == table.shape and blobs.blobs.shape[1] == 8     # EMPTY tables keep their width
                np.testing.assert_array_equal(blobs.blobs, table)
                assert blobs.colocalizations.dtype == np.uint8
                np.testing.assert_array_equal(blobs.colocalizations, colocs)
                assert list(blobs.cols)[-1] == "region" and len(blobs.cols) == 8
            if k in mine:
                assert blobs is dict(got)[k]                         # own tiles come back as they are
        # a rank whose detection failed: everybody raises at the gather instead of waiting for it
        try:
            sd.gather_tiles([], failure=ValueError("tile unreadable") if rank == 1 % world else None)
            verdict = "returned"
        except ValueError as exc:
            verdict = f"own:{exc}"
        except RuntimeError as exc:
            verdict = f"peer:{exc}"
        open(os.path.join(out_dir, f"tiles{rank}"), "w").write(f"{len(mine)} {verdict}")
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize("world,n_tiles", [(8, 11), (8, 5), (3, 7), (2, 1)])
def test_tiles_sharded_over_ranks_and_gathered(tmp_path, world, n_tiles):
    """BASELINE.json configs[4] across the ranks BY TILE: ``detect_blobs_tiles(shard="tiles")`` gives rank r the tiles
    r, r + N, ... (uneven counts, ranks without any tile at 5 tiles over 8 ranks), detects each as one process would
    (inside the call the rank is alone), touches no other rank's tile, and ``gather_tiles`` puts every table and its
    flags on every rank -- ``None`` and EMPTY tables included; a failing rank makes all ranks raise."""
    tmp.spawn(_worker_tiles, args=(world, _free_port(), n_tiles, str(tmp_path)), nprocs=world, join=True)
    notes = [open(tmp_path / f"tiles{r}").read().split(" ", 1) for r in range(world)]
    assert sum(int(n) for n, _ in notes) == n_tiles
    failing = 1 % world
    for r, (_, verdict) in enumerate(notes):
        assert verdict.startswith("own:tile unreadable" if r == failing else "peer:"), (r, verdict)


def test_tile_gather_without_a_group_and_bad_arguments():
    from magellanmapper_amd import detector, stack_detect as sd
    assert dist.tile_share(5) == [0, 1, 2, 3, 4] and dist.tile_share(7, 2, 3) == [2, 5] and dist.tile_share(2, 5, 8) == []
    a, b = detector.Blobs(None), detector.Blobs(None)
    assert sd.gather_tiles([(3, a), (1, b)]) == [(1, b), (3, a)]
    items = dist.gather_tile_tables([(2, np.ones((3, 4))), (0, None, 7)])
    assert [i for i, _, _ in items] == [0, 2] and items[0][1] is None and items[0][2] == 7
    with pytest.raises(ValueError, match="2-D"):
        dist.gather_tile_tables([(0, np.ones(3))])
    with pytest.raises(ValueError, match="shard"):
        list(sd.detect_blobs_tiles("x", [], shard="rows"))


@pytest.mark.parametrize("world,case", [(8, "c4_grid"), (8, "c4_grid+16 region threads"), (3, "holes"),
                                        (4, "extra_columns_two_channels"), (2, "nothing")])
def test_loopback_wire_replays_a_rank_with_the_whole_stacks_table(world, case, monkeypatch):
    """``dist.Loopback`` (what ``bench.py --share k/N`` measures a rank's step with): ONE process plays every rank of the
    distributed pruning twice -- the first round records the rows near the seams, the second the survivors pruned with
    them -- and then replays single ranks: every replay merges the WHOLE stack's table, equal to the one-process
    passes, and the statistics sum up as over a real group."""
    from magellanmapper_amd import stack_detect as sd
    case, _, wide = case.partition("+")
    blocks, tables, shape, channels, n_extra = _dist_case(case)
    grid = blocks.sub_roi_slices.shape
    coords = list(np.ndindex(*grid))
    if wide:
        # a machine with many cores: a rank's 32 blocks are pruned in as many regions as there are region threads --
        # quarter rows of the block grid (2 blocks each) instead of whole x-rows
        from concurrent.futures import ThreadPoolExecutor
        monkeypatch.setattr(sd.stack_prune, "_REGION_POOL", [ThreadPoolExecutor(max_workers=16), os.getpid()])
        monkeypatch.setattr(sd.StackPruner, "REGION_MIN_ROWS", 0)
        seen = []
        init = sd._RegionPruner.__init__
        monkeypatch.setattr(sd._RegionPruner, "__init__", lambda self, *a, **k: (init(self, *a, **k), seen.append(len(self.regions)))[0])

    class Img:
        pass
    Img.shape = shape
    args = (blocks.overlap, blocks.tol, blocks.sub_roi_slices, blocks.sub_rois_offsets, channels, blocks.overlap_padding)
    want, df = sd.StackPruner.prune_blobs_mp(Img, sd.StackDetector.assemble_seg_rois(list(enumerate(tables)), grid, n_extra),
                                             *args)

    def play(q):
        wire.begin_step(q)
        mine = dist.my_share(len(coords))
        assert mine == list(range(*dist.share_bounds(len(coords), q, world)))
        arena = sd._TableArena(11 + n_extra, len(mine))
        local = []
        for i in mine:
            if tables[i] is not None and len(tables[i]):
                arena.add(coords[i], tables[i])
            arena.landed()
            local.append((i, tables[i]))
        seg = sd.StackDetector.assemble_seg_rois(local, grid, n_extra, arena, local_only=True)
        return sd.StackPruner.prune_blobs_mp(Img, seg, *args)

    wire = dist.Loopback(0, world)
    dist.set_loopback(wire)
    try:
        assert dist.world_size() == world and dist._multi_rank()
        for _ in range(2):
            for q in range(world):
                play(q)
        wire.mode = "replay"
        for q in sorted({0, world // 2, world - 1}):
            got, df_q = play(q)
            if want is None:
                assert got is None
                continue
            np.testing.assert_array_equal(got, want)
            np.testing.assert_array_equal(df_q.to_numpy(), df.to_numpy())
        with pytest.raises(NotImplementedError):
            dist.broadcast_table(np.zeros((1, 1)))
        if wide:
            assert seen and max(seen) == 16            # 32 blocks in 16 regions of two blocks
    finally:
        dist.set_loopback(None)
    assert dist.world_size() == 1 and not dist._multi_rank()
