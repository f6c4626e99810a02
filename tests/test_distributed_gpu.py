"""Two ranks, ONE GPU, gloo: the sharded path on ROCm tensors through the HIP kernels and both reduction hooks.

RCCL refuses two ranks on one device, so the collective here is gloo's; everything else - the per-rank HIP sweeps,
the (S, R) bit-pattern MIN of ``mode="points"``, the (S, F) MIN of ``mode="simplices"``, the generator-state
broadcast for random weights - is the code an 8-GPU RCCL run executes.  Results must equal the unsharded GPU
result bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import load_e2e

pytestmark = pytest.mark.gpu
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
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from flooder_amd import _native
        from flooder_amd.distributed import flood_complex_sharded, shard_points
        from helpers import load_e2e

        _native.load()  # no fallback: the HIP library must be there
        dev = torch.device("cuda:0")
        z, kw, keys = load_e2e(name)
        pts = torch.as_tensor(z["points"], device=dev)
        lms = torch.as_tensor(z["landmarks"], device=dev)
        torch.manual_seed(int(z["weight_seed"]) if rank == 0 else 4242 + rank)
        if mode == "points":
            fc = flood_complex_sharded(shard_points(pts, rank, world), lms, mode="points", **kw)
        else:
            fc = flood_complex_sharded(pts, lms, mode="simplices", **kw)
        np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array([fc[k] for k in keys]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["simplices", "points"])
@pytest.mark.parametrize("name", ["torus3d_grid30", "eight2d_rand", "gauss6d_maxdim2"])
def test_two_ranks_one_gpu_match_unsharded(name, mode, tmp_path):
    import flooder_amd as fa

    assert torch.cuda.is_available()
    z, kw, keys = load_e2e(name)
    dev = torch.device("cuda:0")
    torch.manual_seed(int(z["weight_seed"]))
    full = fa.flood_complex(torch.as_tensor(z["points"], device=dev), torch.as_tensor(z["landmarks"], device=dev), **kw)
    want = np.array([full[k] for k in keys])
    mp.spawn(_worker, args=(2, _free_port(), name, str(tmp_path), mode), nprocs=2, join=True)
    r0 = np.load(tmp_path / "r0.npy")
    r1 = np.load(tmp_path / "r1.npy")
    assert np.array_equal(r0, r1)      # every rank returns the full result
    assert np.array_equal(r0, want)    # and it is the unsharded result, bit for bit
    assert np.abs(r0 - z["filtration_f32"]).max() < 5e-6 * max(1.0, float(np.abs(z["points"]).max()))
