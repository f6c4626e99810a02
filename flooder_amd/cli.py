"""Command line front end: ``.npy`` point cloud in, persistence diagrams (pickle) and per-step statistics out.

Mirrors the reference's console script (``flooder/cli.py``: options :185-297, steps :427-500, output payload
:404-424, statistics record :41-48) so existing invocations keep working:

    python -m flooder_amd.cli --input-file cloud.npy --num-landmarks 1000 --output-file out.pkl \\
        --device cuda:0 --stats-json stats.json --cuda-events

Differences, all on the device side: ``cuda:N`` means ROCm device N and is accepted only when it is a gfx950
(MI355X) whose HIP library loads - the reference's compute-capability gate (:313-317) has no meaning here;
``--no-triton`` is accepted and ignored (there is one device path); the tables are plain text (``rich`` is not a
dependency).  Persistence comes from gudhi when it is importable and from the package's own Z/2 reduction
otherwise (``flooder_amd/simplex_tree.py``).
"""
from __future__ import annotations

import argparse
import dataclasses
import json
import os
import pickle
import re
import sys
import time
from pathlib import Path
from typing import List, Optional, Sequence

import numpy as np
import torch

try:  # optional, as in the reference
    import psutil
except Exception:  # pragma: no cover
    psutil = None


@dataclasses.dataclass
class StepStats:
    """One row of the statistics table / one object of ``--stats-json`` (fields as ``cli.py:41-48``)."""
    name: str
    wall_s: float
    cpu_s: float
    ram_delta_mib: Optional[float]
    gpu_peak_mib: Optional[float]
    cuda_ms: Optional[float]


@dataclasses.dataclass
class RunMeta:
    """Metadata stored next to the diagrams (fields as ``cli.py:52-66``)."""
    input_file: str
    output_file: Optional[str]
    num_landmarks: int
    max_dimension: int
    fps_height: int
    batch_size: int
    device: str
    points_per_edge: Optional[int]
    num_rand: Optional[int]
    seed: Optional[int]
    use_triton: bool
    n_points: int
    ambient_dim: int


class StepTimer:
    """Context manager measuring wall and process-CPU seconds, the RSS delta, the peak device memory of the
    torch allocator and - with ``use_cuda_events`` on a device - the elapsed device time between two events
    on the current stream."""

    def __init__(self, name: str, device: torch.device, use_cuda_events: bool = False):
        self.name = name
        self.device = device
        self.on_gpu = device.type == "cuda"
        self.with_events = bool(use_cuda_events) and self.on_gpu
        self.stats: Optional[StepStats] = None

    @staticmethod
    def _rss() -> Optional[int]:
        if psutil is None:
            return None
        try:
            return psutil.Process(os.getpid()).memory_info().rss
        except Exception:
            return None

    def __enter__(self) -> "StepTimer":
        self._rss0 = self._rss()
        if self.on_gpu:
            torch.cuda.reset_peak_memory_stats(self.device)
            if self.with_events:
                self._ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                self._ev[0].record()
        self._cpu0 = time.process_time()
        self._wall0 = time.perf_counter()
        return self

    def __exit__(self, exc_type, exc, tb) -> None:
        wall = time.perf_counter() - self._wall0
        cpu = time.process_time() - self._cpu0
        rss1 = self._rss()
        ram = (rss1 - self._rss0) / 2 ** 20 if (rss1 is not None and self._rss0 is not None) else None
        peak = ms = None
        if self.on_gpu:
            peak = torch.cuda.max_memory_allocated(self.device) / 2 ** 20
            if self.with_events:
                self._ev[1].record()
                torch.cuda.synchronize(self.device)
                ms = self._ev[0].elapsed_time(self._ev[1])
        self.stats = StepStats(self.name, wall, cpu, ram, peak, ms)


def format_stats_table(steps: Sequence[StepStats]) -> str:
    head = ("Step", "Wall (s)", "CPU (s)", "GPU peak (MiB)", "RAM delta (MiB)", "Device (ms)")

    def cell(v) -> str:
        return "-" if v is None or not np.isfinite(v) else f"{v:.3f}"

    rows = [head] + [(s.name, cell(s.wall_s), cell(s.cpu_s), cell(s.gpu_peak_mib), cell(s.ram_delta_mib),
                      cell(s.cuda_ms)) for s in steps]
    width = [max(len(r[i]) for r in rows) for i in range(len(head))]
    lines = ["Flooder runtime statistics"]
    for k, r in enumerate(rows):
        lines.append("  ".join(r[0].ljust(width[0]) if i == 0 else r[i].rjust(width[i]) for i in range(len(head))))
        if k == 0:
            lines.append("  ".join("-" * w for w in width))
    return "\n".join(lines)


def device_type(value: str) -> str:
    """argparse type: ``cpu`` or ``cuda:<id>`` (``cli.py:167-176``)."""
    if value == "cpu" or re.fullmatch(r"cuda:\d+", value):
        return value
    raise argparse.ArgumentTypeError(f"Invalid device '{value}'. Must be 'cpu' or 'cuda:<id>' with <id> an integer.")


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog="flooder", description="Flood complex persistent homology of a point cloud "
                                                           "(MI355X-native build)")
    g = p.add_argument_group("Flooder options")
    g.add_argument("--num-landmarks", metavar="INT", type=int, default=2000,
                   help="Number of landmarks for Flood complex (default: %(default)s)")
    g.add_argument("--max-dimension", metavar="INT", type=int, default=None,
                   help="Compute PH up to max. dimension (exclusive) (default: ambient dim)")
    g.add_argument("--fpsh", dest="fps_height", metavar="INT", type=int, default=9,
                   help="Farthest-Point Sampling height (accepted for compatibility; the selection is exact FPS)")
    g.add_argument("--batch-size", metavar="INT", type=int, default=64,
                   help="Batch size for Flood complex (accepted; does not change results)")
    g.add_argument("--device", type=device_type, default="cuda:0", help='Device: "cpu", or "cuda:N" (default: %(default)s)')
    g.add_argument("--seed", metavar="INT", type=int, default=None, help="Random seed (only used when --num-rand is set)")
    g.add_argument("--no-triton", action="store_true", help="Accepted for compatibility (one device path here)")
    mex = g.add_mutually_exclusive_group(required=False)
    mex.add_argument("--points-per-edge", metavar="INT", type=int, default=None,
                     help="Points per edge for Flood PH (default: 30 if neither option given)")
    mex.add_argument("--num-rand", metavar="INT", type=int, default=None,
                     help="Number of random points per simplex (default: None)")
    io = p.add_argument_group("Input/Output options")
    io.add_argument("--input-file", metavar="FILE", type=str, required=True, help="NumPy .npy file with a (N, D) point cloud")
    io.add_argument("--output-file", metavar="FILE", type=str, default=None,
                    help="Output pickle (.pkl) with persistence diagrams + metadata")
    io.add_argument("-v", "--verbose", action="store_true", help="Print parsed arguments")
    io.add_argument("--stats-json", metavar="FILE", type=str, default=None, help="Write runtime statistics to JSON")
    io.add_argument("--cuda-events", action="store_true", help="Also measure device time with stream events")
    return p


def validate_device(device_str: str) -> torch.device:
    """``RuntimeError`` when the device is missing or is not an MI355X with a loadable HIP library."""
    dev = torch.device(device_str)
    if dev.type == "cuda":
        if not torch.cuda.is_available():
            raise RuntimeError("CUDA requested but not available. Use --device cpu.")
        if dev.index is not None and dev.index >= torch.cuda.device_count():
            raise RuntimeError(f"Device {device_str} requested but only {torch.cuda.device_count()} device(s) present.")
        from . import _native
        lib = _native.load()  # ImportError if the HIP library is missing: no silent CPU fallback
        torch.cuda.set_device(dev)
        import ctypes
        buf = ctypes.create_string_buffer(64)
        _native.check(lib.flooder_device_arch(dev.index or 0, buf, 64), "flooder_device_arch")
        arch = buf.value.decode()
        if not arch.startswith("gfx950"):
            raise RuntimeError(f"Device architecture {arch or '?'} detected; the kernels are built for gfx950 (MI355X).")
    return dev


def load_point_cloud(path: Path):
    """(tensor float32 (N, D), N, D); ``FileNotFoundError`` / ``ValueError`` as ``cli.py:322-349``."""
    if not path.exists():
        raise FileNotFoundError(f"Input file does not exist: {path}")
    try:
        arr = np.load(path, mmap_mode="r")
    except Exception as e:
        raise ValueError(f"Failed to load NumPy file '{path}': {e}") from e
    if arr.ndim != 2:
        raise ValueError(f"Expected a 2D array (N, D); got shape {arr.shape}")
    t = torch.from_numpy(np.array(arr, dtype=np.float32, copy=True))
    return t, int(t.shape[0]), int(t.shape[1])


def effective_max_dim(user_max: Optional[int], ambient_dim: int) -> int:
    if user_max is None:
        return ambient_dim
    if user_max < 1:
        raise ValueError("--max-dimension must be >= 1")
    if user_max > ambient_dim:
        raise ValueError(f"--max-dimension ({user_max}) cannot exceed ambient dimension ({ambient_dim})")
    return user_max


def resolve_simplex_representation(points_per_edge: Optional[int], num_rand: Optional[int]):
    return (30, None) if (points_per_edge is None and num_rand is None) else (points_per_edge, num_rand)


def maybe_seed(seed: Optional[int]) -> None:
    if seed is None:
        return
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def save_output(path: Path, diagrams, meta: RunMeta) -> Path:
    """``{"diagrams": [...], "meta": {...}}`` pickled atomically (``cli.py:404-424``); ``.pkl`` is appended to
    a suffix-less path."""
    if path.suffix == "":
        path = path.with_suffix(".pkl")
    path.parent.mkdir(parents=True, exist_ok=True)
    tmp = path.with_suffix(path.suffix + ".tmp")
    with tmp.open("wb") as f:
        pickle.dump({"diagrams": diagrams, "meta": dataclasses.asdict(meta)}, f, protocol=pickle.HIGHEST_PROTOCOL)
    tmp.replace(path)
    return path


def dump_stats_json(steps: Sequence[StepStats], out_path: Optional[str]) -> None:
    if not out_path:
        return
    p = Path(out_path)
    p.parent.mkdir(parents=True, exist_ok=True)
    with p.open("w") as f:
        json.dump([dataclasses.asdict(s) for s in steps], f, indent=2)


def main(argv: Optional[Sequence[str]] = None) -> int:
    from . import flood_complex

    args = build_parser().parse_args(argv)
    if args.verbose:
        print(vars(args))
    device = validate_device(args.device)
    steps: List[StepStats] = []

    with StepTimer("Loading", device, args.cuda_events) as t:
        cloud_cpu, n_pts, dim = load_point_cloud(Path(args.input_file))
    steps.append(t.stats)
    print(f"Loading point cloud ({n_pts},{dim}) done")

    max_dim = effective_max_dim(args.max_dimension, dim)
    points_per_edge, num_rand = resolve_simplex_representation(args.points_per_edge, args.num_rand)
    maybe_seed(args.seed if num_rand is not None else None)
    use_triton = not args.no_triton

    with StepTimer("Flood complex", device, args.cuda_events) as t:
        cloud = cloud_cpu.to(device, non_blocking=True)
        stree = flood_complex(cloud, args.num_landmarks, max_dimension=max_dim, points_per_edge=points_per_edge,
                              batch_size=args.batch_size, fps_h=args.fps_height,
                              use_triton=use_triton if device.type == "cuda" else None,
                              return_simplex_tree=True, num_rand=num_rand)
    steps.append(t.stats)
    print(f"Building Flood complex with {stree.num_simplices()} simplices done")

    with StepTimer("Persistence", device, args.cuda_events) as t:
        stree.compute_persistence()
        diagrams = [stree.persistence_intervals_in_dimension(i) for i in range(max_dim)]
    steps.append(t.stats)
    print(f"Computing persistence up to max. dim {max_dim} done\n")

    if args.output_file:
        meta = RunMeta(input_file=args.input_file, output_file=args.output_file, num_landmarks=args.num_landmarks,
                       max_dimension=max_dim, fps_height=args.fps_height, batch_size=args.batch_size,
                       device=str(device), points_per_edge=points_per_edge, num_rand=num_rand,
                       seed=args.seed if num_rand is not None else None, use_triton=use_triton,
                       n_points=n_pts, ambient_dim=dim)
        save_output(Path(args.output_file), diagrams, meta)

    print(format_stats_table(steps))
    dump_stats_json(steps, args.stats_json)
    return 0


if __name__ == "__main__":
    try:
        sys.exit(main())
    except Exception as e:  # same contract as the reference: report, then re-raise
        print(f"Error: {e}", file=sys.stderr)
        raise
