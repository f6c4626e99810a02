"""Time of the point index build (bbox, curve codes, radix sort, gather, box tree) by cloud size, HIP events, median of
reps.  With FLOODER_HIP_LIB pointing at a library built with -DFLOODER_OS_SB=.. -DFLOODER_OS_SI=.. this is the tuning
run of the radix sort's block shape (flood_index.hip: flooder_index_sort_zeroed)."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from flooder_amd import _native, core  # noqa: E402

dev = torch.device("cuda:0")
label = sys.argv[1] if len(sys.argv) > 1 else "default"
if len(sys.argv) > 2:   # block shape of the radix passes forced (option "sort_shape")
    assert _native.load().flooder_set_option(b"sort_shape", int(sys.argv[2])) == 0
out = []
for n in (100_000, 300_000, 1_000_000, 2_000_000, 4_000_000, 8_000_000, 16_000_000):
    g = torch.Generator().manual_seed(1)
    pts = torch.randn(n, 3, generator=g).to(dev)
    for _ in range(3):
        idx = core.PointIndex(pts)
    torch.cuda.synchronize()
    ts = []
    for _ in range(15):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        idx = core.PointIndex(pts)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    out.append(f"{n/1e6:g}M {np.median(ts)*1e3:7.1f} us")
    del pts, idx
print(f"{label:12s}", "  ".join(out))
