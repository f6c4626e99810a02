"""Console front end (flooder_amd/cli.py) - the reference's CLI contract (flooder/cli.py:185-297, 404-500)."""
import json
import pickle
import subprocess
import sys

import numpy as np
import pytest
import torch

import flooder_amd as fa
from flooder_amd import cli


def _cloud(tmp_path, n=400, seed=0):
    pts = fa.generate_noisy_torus_points_3d(n, seed=seed).numpy()
    path = tmp_path / "cloud.npy"
    np.save(path, pts.astype(np.float64))  # any float dtype on disk; the CLI casts to float32
    return path, pts


def test_cli_cpu_end_to_end(tmp_path):
    path, pts = _cloud(tmp_path)
    out = tmp_path / "res" / "diagrams"          # no suffix: ".pkl" is appended, the folder is created
    stats = tmp_path / "stats.json"
    rc = cli.main(["--input-file", str(path), "--num-landmarks", "40", "--points-per-edge", "6", "--device", "cpu",
                   "--output-file", str(out), "--stats-json", str(stats), "--max-dimension", "2"])
    assert rc == 0
    payload = pickle.load(open(str(out) + ".pkl", "rb"))
    assert set(payload) == {"diagrams", "meta"}
    meta = payload["meta"]
    assert meta["n_points"] == 400 and meta["ambient_dim"] == 3 and meta["max_dimension"] == 2
    assert meta["points_per_edge"] == 6 and meta["num_rand"] is None and meta["seed"] is None
    assert meta["num_landmarks"] == 40 and meta["device"] == "cpu" and meta["batch_size"] == 64
    assert len(payload["diagrams"]) == 2
    # the diagrams are those of flood_complex + the simplex tree's persistence
    st = fa.flood_complex(torch.from_numpy(pts), 40, max_dimension=2, points_per_edge=6, return_simplex_tree=True)
    st.compute_persistence()
    for d in range(2):
        want = np.asarray(st.persistence_intervals_in_dimension(d)).reshape(-1, 2)
        got = np.asarray(payload["diagrams"][d]).reshape(-1, 2)
        np.testing.assert_allclose(np.sort(got, axis=0), np.sort(want, axis=0), rtol=0, atol=0)
    rows = json.load(open(stats))
    assert [r["name"] for r in rows] == ["Loading", "Flood complex", "Persistence"]
    assert set(rows[0]) == {"name", "wall_s", "cpu_s", "ram_delta_mib", "gpu_peak_mib", "cuda_ms"}
    assert all(r["gpu_peak_mib"] is None and r["cuda_ms"] is None for r in rows)


def test_cli_argument_rules(tmp_path):
    path, _ = _cloud(tmp_path, n=50)
    p = cli.build_parser()
    a = p.parse_args(["--input-file", "x.npy"])
    assert (a.num_landmarks, a.fps_height, a.batch_size, a.device) == (2000, 9, 64, "cuda:0")
    assert cli.resolve_simplex_representation(None, None) == (30, None)
    assert cli.resolve_simplex_representation(None, 12) == (None, 12)
    with pytest.raises(SystemExit):
        p.parse_args(["--input-file", "x.npy", "--points-per-edge", "5", "--num-rand", "7"])
    with pytest.raises(SystemExit):
        p.parse_args(["--input-file", "x.npy", "--device", "gpu0"])
    with pytest.raises(SystemExit):
        p.parse_args([])
    assert cli.effective_max_dim(None, 3) == 3
    with pytest.raises(ValueError):
        cli.effective_max_dim(0, 3)
    with pytest.raises(ValueError):
        cli.effective_max_dim(4, 3)
    with pytest.raises(FileNotFoundError):
        cli.load_point_cloud(tmp_path / "missing.npy")
    bad = tmp_path / "bad.npy"
    np.save(bad, np.zeros(7))
    with pytest.raises(ValueError):
        cli.load_point_cloud(bad)
    junk = tmp_path / "junk.npy"
    junk.write_bytes(b"not a numpy file")
    with pytest.raises(ValueError):
        cli.load_point_cloud(junk)
    t, n, d = cli.load_point_cloud(path)
    assert t.dtype == torch.float32 and (n, d) == (50, 3)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            cli.validate_device("cuda:0")


def test_cli_num_rand_is_seeded(tmp_path):
    path, _ = _cloud(tmp_path, n=200)
    outs = []
    for k in range(2):
        out = tmp_path / f"r{k}.pkl"
        cli.main(["--input-file", str(path), "--num-landmarks", "25", "--num-rand", "20", "--seed", "7",
                  "--device", "cpu", "--output-file", str(out)])
        outs.append(pickle.load(open(out, "rb")))
    assert outs[0]["meta"]["seed"] == 7 and outs[0]["meta"]["num_rand"] == 20
    for a, b in zip(outs[0]["diagrams"], outs[1]["diagrams"]):
        np.testing.assert_array_equal(np.asarray(a), np.asarray(b))


def test_cli_as_module(tmp_path):
    path, _ = _cloud(tmp_path, n=120)
    r = subprocess.run([sys.executable, "-m", "flooder_amd.cli", "--input-file", str(path), "--num-landmarks", "20",
                        "--points-per-edge", "4", "--device", "cpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "Flooder runtime statistics" in r.stdout and "Persistence" in r.stdout


def test_save_to_disk(tmp_path):
    f = tmp_path / "x.pt"
    fa.save_to_disk({"a": torch.arange(3)}, f)
    back = torch.load(f, weights_only=False)
    assert set(back) == {"a", "_meta"} and back["_meta"]["keys"] == ["a"]
    with pytest.raises(FileExistsError):
        fa.save_to_disk({"a": 1}, f)
    fa.save_to_disk({"a": 1, "_meta": "mine"}, f, overwrite=True)
    assert torch.load(f, weights_only=False)["_meta"] == "mine"
    fa.save_to_disk([1, 2], f, overwrite=True)
    assert torch.load(f, weights_only=False) == [1, 2]
    fa.save_to_disk({"a": 1}, f, metadata=False, overwrite=True)
    assert set(torch.load(f, weights_only=False)) == {"a"}


@pytest.mark.gpu
def test_cli_gpu_matches_cpu(tmp_path):
    path, _ = _cloud(tmp_path, n=5000, seed=3)
    res = {}
    for dev in ("cpu", "cuda:0"):
        out = tmp_path / f"{dev.replace(':', '_')}.pkl"
        stats = tmp_path / f"{dev.replace(':', '_')}.json"
        cli.main(["--input-file", str(path), "--num-landmarks", "100", "--points-per-edge", "10", "--device", dev,
                  "--output-file", str(out), "--stats-json", str(stats), "--cuda-events"])
        res[dev] = (pickle.load(open(out, "rb")), json.load(open(stats)))
    assert res["cuda:0"][1][1]["cuda_ms"] is not None and res["cuda:0"][1][1]["gpu_peak_mib"] > 0
    for a, b in zip(res["cpu"][0]["diagrams"], res["cuda:0"][0]["diagrams"]):
        a = np.asarray(a).reshape(-1, 2)
        b = np.asarray(b).reshape(-1, 2)
        assert a.shape == b.shape
        fin = np.isfinite(a)
        np.testing.assert_array_equal(fin, np.isfinite(b))
        np.testing.assert_allclose(np.sort(b[fin]), np.sort(a[fin]), rtol=1e-5, atol=1e-5)
