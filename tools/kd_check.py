"""k-d order of the point index (above 3D): permutation check, build time against the curve order, tightness of the
1024-point nodes.  python tools/kd_check.py [n] [dim]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flooder_amd import core

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
pts = torch.randn(n, dim, generator=g).to(dev)
for above in (3, 8):
    core.KD_ORDER_ABOVE_DIM = above
    idx = core.PointIndex(pts)
    torch.cuda.synchronize()
    t = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); idx = core.PointIndex(pts); b.record(); torch.cuda.synchronize(); t.append(a.elapsed_time(b))
    o = idx.order32.long()
    perm_ok = bool((torch.sort(o).values == torch.arange(n, device=dev)).all())
    rows_ok = bool((idx.pts[:n, :dim] == pts[o]).all())
    ext = {}
    for gsz in (16, 1024, 65536):
        m = n // gsz * gsz
        grp = idx.pts[:m, :dim].reshape(-1, gsz, dim)
        ext[gsz] = round(float((grp.amax(1) - grp.amin(1)).mean()), 4)
    print("kd" if idx.kd else "curve", "n", n, "dim", dim, "build ms", round(sorted(t)[2], 3), "permutation", perm_ok, "rows", rows_ok,
          "mean extent per axis of 16 / 1024 / 65536-row groups", ext)
