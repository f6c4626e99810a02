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
        if name.startswith("gauss6d"):   # (small: let the sorted-sample sweep and its tile sharding run all the same)
            from flooder_amd import core
            core.BVH_SORTED_MIN_SAMPLES = 0
        dev = torch.device("cuda:0")
        z, kw, keys = load_e2e(name)
        pts = torch.as_tensor(z["points"], device=dev)
        lms = torch.as_tensor(z["landmarks"], device=dev)
        torch.manual_seed(int(z["weight_seed"]) if rank == 0 else 4242 + rank)
        if mode == "points":
            fc = flood_complex_sharded(shard_points(pts, rank, world), lms, mode="points", **kw)
        else:
            fc = flood_complex_sharded(pts, lms, mode=mode, **kw)
        np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array([fc[k] for k in keys]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["simplices", "points", "blocks"])
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


@pytest.mark.parametrize("cloud,world", [("gauss", 3), ("torus", 8), ("gauss", 700)])
def test_block_shards_one_after_the_other_equal_unsharded(cloud, world):
    """``shard_blocks``: every rank's block swept against the index of ITS sub-cloud (the rows inside the block's
    bounding balls) - all ranks of a ``world``-rank run one after the other on this GPU, combined by the minimum that
    ``all_reduce(MIN)`` would take.  Landmarks are cloud points (FPS), so the values must be the unsharded ones bit
    for bit; a rank sees only part of the cloud; more ranks than simplices leaves some without work."""
    import flooder_amd as fa
    from flooder_amd import core
    from oracle import flood_oracle as fo

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    pts = (torch.randn(150_000, 3, generator=g) if cloud == "gauss" else torch.as_tensor(fo.noisy_torus(150_000, seed=5))).to(dev)
    lms = fa.generate_landmarks(pts, 120 if world > 100 else 300, start_idx=0)
    full = fa.flood_complex(pts, lms)
    keys = sorted(full)
    want = np.array([full[k] for k in keys], dtype=np.float32)
    got = np.full(len(keys), np.inf, dtype=np.float32)
    rows_seen = []
    orig = core.block_subcloud

    def spy(points32, verts, d, box=None):
        sub = orig(points32, verts, d, box=box)
        rows_seen.append(sub.shape[0])
        return sub

    core.block_subcloud = spy
    parts = []   # every rank's (S, F) matrix as it would enter the all-reduce

    def collect(full):
        parts.append(full.clone())

    def reduced(full):   # what all_reduce(MIN) leaves on every rank
        full.copy_(torch.stack(parts + [full]).amin(dim=0))

    try:
        ranks = list(range(world)) if world <= 8 else [0, 1, world // 2, world - 1]
        for r in ranks[1:]:
            fa.flood_complex(pts, lms, simplex_shard=(r, world), shard_blocks=True, face_reduce_hook=collect)
        part = fa.flood_complex(pts, lms, simplex_shard=(ranks[0], world), shard_blocks=True, face_reduce_hook=reduced)
        assert sorted(part) == keys
        got = np.array([part[k] for k in keys], dtype=np.float32)
    finally:
        core.block_subcloud = orig
    if world <= 8:
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        assert min(rows_seen) < pts.shape[0], "every block saw the whole cloud"
    else:   # a few ranks of many: what they produced is final, the rest still +inf
        done = np.isfinite(got)
        assert done.any() and np.array_equal(got[done].view(np.uint32), want[done].view(np.uint32))


@pytest.mark.parametrize("dim,world,kw", [(6, 2, dict(max_dimension=2, points_per_edge=6)), (5, 3, dict(max_dimension=2, num_rand=40)),
                                          (4, 8, dict(max_dimension=3, points_per_edge=4)), (6, 5000, dict(max_dimension=1, points_per_edge=5))])
def test_tile_shards_of_the_sorted_sweep_one_after_the_other_equal_unsharded(dim, world, kw, monkeypatch):
    """Above 3D a simplex-sharded run shards the TILES of the sorted sample order (core.shards_sorted_tiles): all
    ranks of a ``world``-rank run one after the other on this GPU, each holding the face maxima over its own samples,
    combined by what the all-reduce does (MIN of the negated matrices).  Bit for bit the unsharded values - also with
    more ranks than tiles, and whether tiles or simplices are sharded."""
    import flooder_amd as fa
    from flooder_amd import core
    from oracle import flood_oracle as fo

    dev = torch.device("cuda:0")
    rng = np.random.default_rng(dim + world)
    P = rng.normal(size=(50_000, dim)).astype(np.float32)
    tp = torch.as_tensor(P, device=dev)
    lms = tp[torch.as_tensor(fo.exact_fps(P, 26, 0), device=dev)]
    monkeypatch.setattr(core, "BVH_SORTED_MIN_SAMPLES", 0)
    torch.manual_seed(7)
    want = fa.flood_complex(tp, lms, method="bvh", **kw)
    keys = sorted(want)
    ranks = list(range(world)) if world <= 8 else [0, 1, world // 2, world - 1]

    def run(tiles):
        monkeypatch.setattr(core, "SHARD_SORTED_TILES", tiles)
        used = []
        orig = core._sweep_dimension_bvh

        def spy(*a, **k):
            used.append(k.get("tile_shard"))
            return orig(*a, **k)

        monkeypatch.setattr(core, "_sweep_dimension_bvh", spy)
        parts = {}   # per dimension pass (in call order): the matrices as they enter the all-reduce

        def collect_for(r):
            calls = [0]

            def hook(full):
                parts.setdefault(calls[0], []).append(full.clone())
                calls[0] += 1
            return hook

        def reduce_hook():
            calls = [0]

            def hook(full):
                full.copy_(torch.stack(parts.get(calls[0], []) + [full]).amin(dim=0))
                calls[0] += 1
            return hook

        for r in ranks[1:]:
            torch.manual_seed(7)
            fa.flood_complex(tp, lms, method="bvh", simplex_shard=(r, world), face_reduce_hook=collect_for(r), **kw)
        torch.manual_seed(7)
        got = fa.flood_complex(tp, lms, method="bvh", simplex_shard=(ranks[0], world), face_reduce_hook=reduce_hook(), **kw)
        monkeypatch.setattr(core, "_sweep_dimension_bvh", orig)
        return got, used

    got, used = run(True)
    assert any(u is not None for u in used), "no dimension pass was tile-sharded"
    if world <= 8:
        assert sorted(got) == keys and [got[k] for k in keys] == [want[k] for k in keys]
        got2, used2 = run(False)
        assert all(u is None for u in used2)
        assert [got2[k] for k in keys] == [want[k] for k in keys]


def _rccl_worker(rank, world, port, out_dir):
    """One rank, backend "nccl" (= RCCL on ROCm), the device bound at init as bench.py does."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        from flooder_amd import _native, core
        from flooder_amd.distributed import flood_complex_sharded, shard_points
        from helpers import load_e2e

        _native.load()
        assert dist.get_backend() == "nccl"
        # the collectives of the three modes, on the dtypes they are issued with, through RCCL itself
        for t, op in ((torch.tensor([3, 0x7f800000, 5], dtype=torch.int32, device=dev), dist.ReduceOp.MIN),
                      (torch.tensor([3, 0x7ff0000000000000], dtype=torch.int64, device=dev), dist.ReduceOp.MIN),
                      (torch.tensor([-1.5, -0.0, float("-inf")], dtype=torch.float32, device=dev), dist.ReduceOp.MIN),
                      (torch.tensor([1.5, float("inf")], dtype=torch.float32, device=dev), dist.ReduceOp.MAX)):
            want = t.clone()
            dist.all_reduce(t, op=op)
            assert torch.equal(t.view(torch.int64 if t.dtype == torch.int64 else torch.int32),
                               want.view(torch.int64 if t.dtype == torch.int64 else torch.int32)), (t, want)
        res = {}
        for name in ("torus3d_grid30", "eight2d_rand", "gauss6d_maxdim2"):
            if name.startswith("gauss6d"):   # (small: let the sorted-sample sweep and its tile sharding run all the same)
                core.BVH_SORTED_MIN_SAMPLES = 0
            z, kw, keys = load_e2e(name)
            pts = torch.as_tensor(z["points"], device=dev)
            lms = torch.as_tensor(z["landmarks"], device=dev)
            for mode in ("simplices", "points", "blocks"):
                torch.manual_seed(int(z["weight_seed"]))
                p = shard_points(pts, rank, world) if mode == "points" else pts
                fc = flood_complex_sharded(p, lms, mode=mode, always_reduce=True, **kw)
                res[f"{name}/{mode}"] = np.array([fc[k] for k in keys])
        # float64 inputs: the (S, R) int64 bit patterns through MIN
        z, kw, keys = load_e2e("torus3d_grid30")
        fc = flood_complex_sharded(torch.as_tensor(z["points"], device=dev).double(), torch.as_tensor(z["landmarks"], device=dev).double(),
                                   mode="points", always_reduce=True, **kw)
        res["torus3d_grid30/points_f64"] = np.array([fc[k] for k in keys])
        np.savez(os.path.join(out_dir, "rccl.npz"), **res)
    finally:
        dist.destroy_process_group()


def test_rccl_world_size_one(tmp_path):
    """RCCL itself, on the one GPU there is: a world of ONE rank with backend "nccl" and ``device_id`` at init (what
    ``bench.py`` does on an 8-GPU node), every collective of the sharded path forced to run (``always_reduce``) -
    ``all_reduce(MIN)`` on int32 / int64 bit patterns (``mode="points"``, float32 and float64 inputs), on the float32
    (S, F) matrix (``"simplices"``, ``"blocks"``) and on the negated matrix of the tile shards above 3D, MIN / MAX of
    the cloud's extent, the generator-state broadcast.  Results must be the unsharded ones bit for bit."""
    import warnings

    import flooder_amd as fa
    from flooder_amd import core

    assert torch.cuda.is_available()
    mp.spawn(_rccl_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    got = np.load(tmp_path / "rccl.npz")
    dev = torch.device("cuda:0")
    keep = core.BVH_SORTED_MIN_SAMPLES
    try:
        for name in ("torus3d_grid30", "eight2d_rand", "gauss6d_maxdim2"):
            if name.startswith("gauss6d"):
                core.BVH_SORTED_MIN_SAMPLES = 0
            z, kw, keys = load_e2e(name)
            torch.manual_seed(int(z["weight_seed"]))
            full = fa.flood_complex(torch.as_tensor(z["points"], device=dev), torch.as_tensor(z["landmarks"], device=dev), **kw)
            want = np.array([full[k] for k in keys])
            for mode in ("simplices", "points", "blocks"):
                assert np.array_equal(got[f"{name}/{mode}"], want), (name, mode)
        z, kw, keys = load_e2e("torus3d_grid30")
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            full = fa.flood_complex(torch.as_tensor(z["points"], device=dev).double(), torch.as_tensor(z["landmarks"], device=dev).double(), **kw)
        assert np.array_equal(got["torus3d_grid30/points_f64"], np.array([full[k] for k in keys]))
    finally:
        core.BVH_SORTED_MIN_SAMPLES = keep


def test_block_shards_refuse_landmarks_off_the_cloud():
    """``shard_blocks`` is exact only for landmarks that are rows of the cloud (a block's sub-cloud holds the rows inside
    its bounding balls): anything else is refused instead of answered with values that are too large."""
    import flooder_amd as fa

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    pts = torch.randn(20_000, 3, generator=g).to(dev)
    lms = fa.generate_landmarks(pts, 40, start_idx=0)
    fa.flood_complex(pts, lms, points_per_edge=6, simplex_shard=(0, 2), shard_blocks=True)
    with pytest.raises(ValueError, match="rows of `points`"):
        fa.flood_complex(pts, lms + 1e-3, points_per_edge=6, simplex_shard=(0, 2), shard_blocks=True)
    # landmarks_in_cloud: the caller vouches (True: the O(n log n) membership test and its host synchronisation are
    # skipped - same values), or says they are not (False: refused without looking)
    from flooder_amd import core

    calls = []
    orig = core._rows_are_subset
    core._rows_are_subset = lambda *a, **k: calls.append(1) or orig(*a, **k)
    try:
        a = fa.flood_complex(pts, lms, points_per_edge=6, simplex_shard=(0, 2), shard_blocks=True)
        n_checked = len(calls)
        b = fa.flood_complex(pts, lms, points_per_edge=6, simplex_shard=(0, 2), shard_blocks=True, landmarks_in_cloud=True)
        assert n_checked >= 1 and len(calls) == n_checked and a == b
    finally:
        core._rows_are_subset = orig
    with pytest.raises(ValueError, match="rows of `points`"):
        fa.flood_complex(pts, lms, points_per_edge=6, simplex_shard=(0, 2), shard_blocks=True, landmarks_in_cloud=False)
