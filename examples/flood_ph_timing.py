"""Flood PH runtime on the reference's three example clouds, measured with its protocol.

Counterpart of the reference's `examples/example_01_cheese_3d.py`, `example_02_torus_3d.py` and
`example_03_figure_eight_2d.py` (warm-up call on the first 10 000 points, `torch.cuda.synchronize()`,
`time.perf_counter()` around complex construction and around persistence: example_01:77-107), using
`flooder_amd` for everything.  The Alpha-complex comparison of those scripts needs gudhi and is run only when
gudhi is importable.

    python examples/flood_ph_timing.py [cheese|torus|eight] [--sizes 10000 100000 1000000] [--reps 3] [--cpu]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # run from a source checkout

import numpy as np
import torch

import flooder_amd as fa

try:
    from gudhi import AlphaComplex  # noqa: F401
    HAS_GUDHI = True
except Exception:
    HAS_GUDHI = False


def make_cloud(kind: str, n: int, device):
    if kind == "cheese":
        return fa.generate_swiss_cheese_points(n, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), 6, (0.1, 0.2), device=device)[0]
    if kind == "torus":
        return fa.generate_noisy_torus_points_3d(n, device=device)
    if kind == "eight":
        return fa.generate_figure_eight_points_2d(n, noise_std=0.005).to(device)
    raise ValueError(kind)


def timed_flood_ph(points: torch.Tensor, n_lms: int):
    on_gpu = points.is_cuda
    fa.flood_complex(points[:10000], min(n_lms, 10000))            # warm-up, as the reference's examples do
    if on_gpu:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = fa.flood_complex(points, n_lms, return_simplex_tree=True)
    if on_gpu:
        torch.cuda.synchronize()
    t_complex = time.perf_counter() - t0
    st.compute_persistence()
    t_ph = time.perf_counter() - t0
    dim = points.shape[1]
    return t_complex, t_ph, st.num_simplices(), np.asarray(st.persistence_intervals_in_dimension(dim - 1))


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("cloud", nargs="?", default="cheese", choices=["cheese", "torus", "eight"])
    ap.add_argument("--sizes", type=int, nargs="+", default=[10_000, 100_000, 1_000_000])
    ap.add_argument("--landmarks", type=int, default=1000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--cpu", action="store_true", help="run the CPU (kd-tree) path instead of the MI355X")
    args = ap.parse_args()
    device = torch.device("cpu") if args.cpu else torch.device("cuda")
    print(f"Flood PH timing on {args.cloud} ({device}, {args.landmarks} landmarks)")
    for n in args.sizes:
        rows = []
        for rep in range(args.reps):
            pts = make_cloud(args.cloud, n, device)
            t_c, t_ph, n_simplices, top = timed_flood_ph(pts, args.landmarks)
            rows.append((t_c, t_ph))
            long_bars = int((top[:, 1] - top[:, 0] > 0.05).sum()) if top.size else 0
            print(f"{n:9d} points (try {rep}) | complex {t_c:7.3f} s | complex + PH {t_ph:7.3f} s | "
                  f"{n_simplices} simplices | top-dimensional bars longer than 0.05: {long_bars}")
            if HAS_GUDHI and n <= 1_000_000 and rep == 0:
                t0 = time.perf_counter()
                alpha = AlphaComplex(points=pts.cpu().numpy()).create_simplex_tree(output_squared_values=False)
                alpha.compute_persistence()
                print(f"{n:9d} points         | Alpha complex + PH (gudhi, CPU) {time.perf_counter() - t0:7.3f} s")
        m = np.mean(rows, axis=0)
        print(f"{n:9d} points mean     | complex {m[0]:7.3f} s | complex + PH {m[1]:7.3f} s")


if __name__ == "__main__":
    main()
