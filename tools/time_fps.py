"""Landmark selection timings on one MI355X: brute force, bucketed one landmark per launch (flooder_fps_indexed_f32,
dim <= 3) and bucketed with several landmarks per launch (flooder_fps_batched_f32).  usage: python tools/time_fps.py [quick]"""
import sys, time, torch
sys.path.insert(0, '.')
import flooder_amd as fa
from flooder_amd import core, _native
dev = torch.device('cuda:0')
torch.manual_seed(42)
lib = _native.load()
cases = [("1M gaussian 3D / 1k", torch.randn(1_000_000, 3), 1000),
         ("2M gaussian 6D / 2k", torch.randn(2_000_000, 6), 2000)]
if "quick" not in sys.argv:
    cases.append(("16M cheese 3D / 4k", fa.generate_swiss_cheese_points(16_000_000, k=6, seed=42)[0], 4000))
for name, pts, k in cases:
    tp = pts.to(dev)
    index = core.PointIndex(tp); torch.cuda.synchronize()
    ref = None
    for method, batched, opts in (("brute", False, {}), ("bucket", False, {}), ("bucket", True, {}),
                                  ("bucket", True, {"fps_switch": 4}), ("bucket", True, {"fps_switch": 16}),
                                  ("bucket", True, {"fps_switch": 32}), ("bucket", True, {"fps_switch": 64}),
                                  ("bucket", True, {"fps_switch": 128}), ("bucket", True, {"fps_rounds": 1})):
        if method == "bucket" and not batched and pts.shape[1] > 3:
            continue
        if method == "brute" and pts.shape[0] > 4_000_000 and "full" not in sys.argv:
            k_run = 600          # (200 ms per 4000 landmarks: a prefix is enough for the rate)
        else:
            k_run = k
        core.FPS_BATCHED = batched
        for o, v in opts.items():
            lib.flooder_set_option(o.encode(), v)
        try:
            core.fps_indices(tp, 128, 0, method=method, index=index if method == "bucket" else None); torch.cuda.synchronize()
            t0 = time.perf_counter()
            got = core.fps_indices(tp, k_run, 0, method=method, index=index if method == "bucket" else None)
            torch.cuda.synchronize()
            t2 = time.perf_counter() - t0
        finally:
            for o in opts:
                lib.flooder_set_option(o.encode(), 0)
        got = got.cpu()
        if ref is None:
            ref = got
        same = bool((got[:min(len(got), len(ref))] == ref[:min(len(got), len(ref))]).all())
        tag = f"{method}{' batched' if batched else ''} {opts if opts else ''}"
        print(f"{name:22s} {tag:36s} {t2 * 1e3:9.3f} ms ready index  {t2 / k_run * 1e6:6.2f} us/landmark  launches "
              f"{core.LAST_FPS_LAUNCHES if batched else k_run:5d}  same indices as the first row: {same}", flush=True)
    core.FPS_BATCHED = True
