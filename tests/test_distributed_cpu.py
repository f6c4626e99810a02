"""world_size-2 gloo test of the sharded path on CPU: two processes each hold an interleaved shard of
the cloud, all_reduce(MIN) on the per-sample distance buffer, result == the unsharded reference golden."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import load_e2e, dict_values

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, name, out_dir, mode):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from flooder_amd.distributed import flood_complex_sharded, shard_points
        from helpers import load_e2e

        z, kw, keys = load_e2e(name)
        pts = torch.as_tensor(z["points"])
        lms = torch.as_tensor(z["landmarks"])
        # only rank 0 holds the seed the golden was drawn with: flood_complex_sharded broadcasts its generator state
        torch.manual_seed(int(z["weight_seed"]) if rank == 0 else 987654321 + rank)
        if mode == "points":
            fc = flood_complex_sharded(shard_points(pts, rank, world), lms, mode="points", **kw)
        else:
            fc = flood_complex_sharded(pts, lms, mode="simplices", **kw)
        np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array([fc[k] for k in keys]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["simplices", "points"])
@pytest.mark.parametrize("name", ["torus3d_grid", "eight2d_rand"])
def test_two_rank_gloo_matches_unsharded(name, mode, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, name, str(tmp_path), mode), nprocs=2, join=True)
    z, kw, keys = load_e2e(name)
    r0 = np.load(tmp_path / "r0.npy")
    r1 = np.load(tmp_path / "r1.npy")
    assert np.array_equal(r0, r1)  # every rank returns the full result
    assert np.abs(r0 - z["filtration_f32"]).max() < 5e-7


def _mismatch_worker(rank, world, port, name, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from flooder_amd.distributed import flood_complex_sharded
        from helpers import load_e2e

        z, kw, _ = load_e2e(name)
        pts = torch.as_tensor(z["points"])
        lms = torch.as_tensor(z["landmarks"])
        if rank == 1:
            lms = lms[:-1]          # another complex on this rank
        try:
            flood_complex_sharded(pts, lms, mode="simplices", **kw)
            msg = "no error"
        except RuntimeError as e:
            msg = str(e)
        with open(os.path.join(out_dir, f"r{rank}.txt"), "w") as f:
            f.write(msg)
    finally:
        dist.destroy_process_group()


def test_ranks_holding_different_complexes_stop_before_the_collective(tmp_path):
    """A rank whose Delaunay differs (other landmarks; a native library that loaded on one rank only) would make the
    face all-reduce hang or mix rows: every rank raises instead (``core._assert_ranks_agree``)."""
    port = _free_port()
    mp.spawn(_mismatch_worker, args=(2, port, "torus3d_grid", str(tmp_path)), nprocs=2, join=True)
    for r in (0, 1):
        assert "do not hold the same complex" in (tmp_path / f"r{r}.txt").read_text()
