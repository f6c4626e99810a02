"""Landmark selection timings on one MI355X: brute force vs bucketed.  usage: python tools/time_fps.py"""
import sys, time, torch
sys.path.insert(0, '.')
import flooder_amd as fa
from flooder_amd import core
dev = torch.device('cuda:0')
torch.manual_seed(42)
for name, pts, k in (("1M gaussian / 1k", torch.randn(1_000_000, 3), 1000),
                     ("16M cheese / 4k", fa.generate_swiss_cheese_points(16_000_000, k=6, seed=42)[0], 4000)):
    tp = pts.to(dev)
    for method in ("brute", "bucket"):
        core.fps_indices(tp, 16, 0, method=method); torch.cuda.synchronize()
        t0 = time.perf_counter(); core.fps_indices(tp, k, 0, method=method); torch.cuda.synchronize()
        t = time.perf_counter() - t0
        index = core.PointIndex(tp); torch.cuda.synchronize()
        t0 = time.perf_counter(); core.fps_indices(tp, k, 0, method=method, index=index if method == "bucket" else None); torch.cuda.synchronize()
        t2 = time.perf_counter() - t0
        print(f"{name:20s} {method:7s} {t * 1e3:9.3f} ms   (with a ready index {t2 * 1e3:9.3f} ms)  {t2 / k * 1e6:6.2f} us/landmark", flush=True)
