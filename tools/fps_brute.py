"""Brute-force exact FPS (flooder_fps_f32: one sweep over the cloud per landmark) for the profiler:
python tools/fps_brute.py <n_points> <dim> <n_landmarks>"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flooder_amd import core
n, dim, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
torch.manual_seed(0)
pts = torch.rand(n, dim, device="cuda:0")
core.fps_indices(pts, 16, 0, method="brute"); torch.cuda.synchronize()
t0 = time.perf_counter(); core.fps_indices(pts, k, 0, method="brute"); torch.cuda.synchronize()
t = time.perf_counter() - t0
b = (4 * dim + 8) * n
print(f"brute FPS {n} x {dim}D, {k} landmarks: {t / k * 1e6:.2f} us per landmark, {b * k / t / 1e9:.0f} GB/s of {b / 1e6:.0f} MB per landmark")
