"""The RCCL (``nccl`` backend) branches of ``magellanmapper_amd.dist`` executed on the one GPU a test box has.

With one rank every helper returns before its collective, and the gloo tests take the CPU branches, so until an
8-GPU node runs ``bench.py --gpus 8`` the device branches -- pinned staging buffers, ``all_gather_into_tensor`` on
device tensors, the copies back and the stream ordering around them -- would never have executed.  Here a ONE-rank
``nccl`` process group is set up on ``cuda:0`` and ``dist._force_collectives`` (a test hook: nothing in the product
sets it) sends that single rank through the collective code: the first RCCL calls of this repository."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

_SCRIPT = textwrap.dedent('''
    import os, sys
    import numpy as np
    sys.path.insert(0, {root!r})
    sys.path.insert(0, os.path.join({root!r}, "tests"))
    import torch
    import torch.distributed as td
    from magellanmapper_amd import config, dist, stack_detect as sd
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    td.init_process_group("nccl", rank=0, world_size=1, device_id=dev,
                          init_method="tcp://127.0.0.1:{port}")
    assert td.get_backend() == "nccl" and dist.world_size() == 1
    dist._force_collectives = True          # the test hook: one rank goes through the collective code
    try:
        rng = np.random.default_rng(1)
        rows = rng.normal(size=(1234, 10))
        # pinned staging -> device all_gather_into_tensor -> pinned copy back, twice (the buffers are reused)
        for _ in range(2):
            parts = dist.all_gather_rows(rows, 10)
            assert len(parts) == 1 and parts[0].shape == rows.shape and np.array_equal(parts[0], rows)
            joined = dist.all_gather_rows_concat(rows, 10)
            assert joined.flags["C_CONTIGUOUS"] and np.array_equal(joined, rows)
        assert "send" in dist._pinned_bufs and dist._pinned_bufs["send"].is_pinned()
        assert dist.all_gather_rows_concat(np.zeros((0, 0)), 7).shape == (0, 7)      # nobody holds rows
        big = rng.normal(size=(200000, 12))                                           # 19 MB: the survivors' exchange
        assert np.array_equal(dist.all_gather_rows_concat(big, 12), big)
        v = np.arange(7, dtype=np.int64) * 3
        assert np.array_equal(dist.all_reduce_sum(v), v)
        dist.raise_together(None, "nothing")
        try:
            dist.raise_together(ValueError("mine"), "a stage")
            raise SystemExit("raise_together swallowed the failure")
        except ValueError as exc:
            assert str(exc) == "mine"
        try:
            dist.all_gather_rows(None, 10, failure=KeyError("stage failed"))
            raise SystemExit("the failure did not travel with the row counts")
        except KeyError:
            pass
        tbl = rng.normal(size=(321, 11))
        assert np.array_equal(dist.broadcast_table(tbl), tbl) and dist.broadcast_table(None) is None
        local = [(0, rng.integers(0, 9, (5, 11)).astype(float)), (1, None), (2, np.zeros((0, 11)))]
        merged = dist.gather_tables(local, 3)
        assert [i for i, _ in merged] == [0, 1, 2] and np.array_equal(merged[0][1], local[0][1])
        assert merged[1][1] is None and merged[2][1].shape == (0, 11)
        idx, allrows, empties = dist.gather_tables(local, 3, decode_on=0, raw=True)
        assert list(idx) == [0] * 5 and np.array_equal(allrows, local[0][1]) and empties == [2]
        assert dist.last_gather_ms() > 0

        # the whole distributed pruning of a stack (both exchanges, the merge by key, the counts' all_reduce) over
        # the one-rank RCCL group == the single-process passes
        from test_host_logic import _synthetic_block_tables
        config.setup_roi_profiles(None)
        config.resolutions = np.array([[1.0, 1.0, 1.0]])
        shape = (96, 150, 170)
        config.roi_profile.update(segment_size=40, denoise_size=None)
        blocks = sd.setup_blocks(config.roi_profile, shape)
        by_coord = _synthetic_block_tables(np.random.default_rng(31), shape, blocks, 6000, [0])
        grid = blocks.sub_roi_slices.shape
        coords = list(np.ndindex(*grid))

        class Img:
            pass
        Img.shape = shape

        def seg_of(local_only):
            arena = sd._TableArena(11, len(coords))
            for c in coords:
                if by_coord[c] is not None:
                    arena.add(c, by_coord[c])
                arena.landed()
            seg = sd.StackDetector.assemble_seg_rois([(i, by_coord[c]) for i, c in enumerate(coords)], grid, 0, arena)
            seg.local_only = local_only
            return seg
        dist._force_collectives = False
        want, df_want = sd.StackPruner.prune_blobs_mp(Img, seg_of(False), blocks.overlap, blocks.tol,
                                                      blocks.sub_roi_slices, blocks.sub_rois_offsets, [0],
                                                      blocks.overlap_padding)
        dist._force_collectives = True
        dist.last_gather_ms()
        got, df_got = sd.StackPruner.prune_blobs_mp(Img, seg_of(True), blocks.overlap, blocks.tol,
                                                    blocks.sub_roi_slices, blocks.sub_rois_offsets, [0],
                                                    blocks.overlap_padding)
        assert dist.last_gather_ms() > 0                      # the collectives really ran
        assert 1000 < len(want) and np.array_equal(got, want)
        assert np.array_equal(df_got.to_numpy(), df_want.to_numpy())
        torch.cuda.synchronize()
        print("RCCL_ONE_RANK_OK", len(got))
    finally:
        td.destroy_process_group()
''')


def test_one_rank_rccl_group_executes_the_device_collectives(tmp_path):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: torch.cuda.is_available() is False")
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "rccl_one_rank.py"
    script.write_text(_SCRIPT.format(root=ROOT, port=port))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    run = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=env,
                         cwd=str(tmp_path))
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-4000:])
    assert "RCCL_ONE_RANK_OK" in run.stdout
