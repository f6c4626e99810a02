import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from flooder_amd import core
dev = torch.device('cuda:0')
torch.manual_seed(42)
for n, k in [(1_000_000, 1000), (1_000_000, 2000), (100_000, 1000), (16_000_000, 1000)]:
    pts = torch.randn(n, 3, device=dev)
    core.fps_indices(pts, 8, 0); torch.cuda.synchronize()
    t0 = time.perf_counter(); idx = core.fps_indices(pts, k, 0); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"n={n} landmarks={k}: {dt*1e3:.2f} ms  ({dt/k*1e6:.2f} us/iter, {(16*n+4*n)*k/dt/1e12:.2f} TB/s eff. of 20 B/pt)")
