"""Diagnostic (library built with -DFLOODER_WAVE_END_FIN): duration of every item of the finish's last pass, and what
list scheduling on the chip's waves would make of other orders.   usage: python tools/item_times.py [cfg3|cfg5] [W]"""
import sys, heapq, torch, numpy as np
sys.path.insert(0, '.')
import flooder_amd as fa
from flooder_amd import _native, core
core.CELL_SUPER = False
_native.check(_native.load().flooder_set_option(b"finish_budget", 0), "set_option")
which = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1
torch.manual_seed(42)
dev = torch.device('cuda:0')
if which == "cfg5":
    from flooder_amd.synthetic import generate_swiss_cheese_points
    pts = generate_swiss_cheese_points(16_000_000, k=6, seed=42)[0].to(dev)
    nl = 4000
else:
    pts = fa.generate_noisy_torus_points_3d(1_000_000, seed=42).to(dev)
    nl = 1000
lms = fa.generate_landmarks(pts, nl, start_idx=0)
stree, simplices = core._build_complex(lms, 3)
verts = lms[torch.as_tensor(simplices[3], device=dev)]
verts = verts[torch.argsort(verts.mean(1)[:, 0])][0::W].contiguous()
weights, vi, fi = core.generate_grid(30, 3, dev, torch.float32)
faces = core._FaceTable(fi, weights.shape[0], dev)
index = core.PointIndex(pts)
stats = torch.zeros(40000 + 9 + 2 * 200000 + 64, dtype=torch.int64, device=dev)
for _ in range(2):
    stats.zero_()
    core._sweep_dimension_cell(index, verts, weights, faces, None, stats=stats)
torch.cuda.synchronize()
rec = stats[9 + 40000: 9 + 40000 + 2 * 200000].cpu().numpy().reshape(-1, 2)
rec = rec[rec[:, 0] != 0]
dur = (rec[:, 0] & 0xffffffff).astype(np.float64)          # 10 ns ticks
rounds = (rec[:, 0] >> 32) & 0xfff
evals = (rec[:, 0] >> 44) & 0xfffff
M = (rec[:, 1] & 0xffffffff).astype(np.uint32).view(np.float32).astype(np.float64)
nlive = (rec[:, 1] >> 32) & 0xff
print(f"{which} W={W}: {len(dur)} items, total {dur.sum() / 1e5:.1f} wave-ms, mean {dur.mean() / 100:.1f} us, max {dur.max() / 100:.1f} us")
for name, key in (("bound M", M), ("live samples", nlive.astype(float)), ("M * live", M * nlive), ("evals (oracle)", evals.astype(float))):
    print(f"  rank correlation of the duration with {name}: {np.corrcoef(np.argsort(np.argsort(key)), np.argsort(np.argsort(dur)))[0, 1]:.3f}")

def span(order, workers=4096):
    h = [0.0] * workers
    heapq.heapify(h)
    for i in order:
        heapq.heappush(h, heapq.heappop(h) + dur[i])
    return max(h)

n = len(dur)
print(f"  list scheduling on 4096 waves (ticks): as queued {span(range(n)):.0f}; by bound {span(np.argsort(-M)):.0f}; by live count "
      f"{span(np.argsort(-nlive, kind='stable')):.0f}; by M*live {span(np.argsort(-(M * nlive))):.0f}; by true duration {span(np.argsort(-dur)):.0f}; "
      f"perfect balance {dur.sum() / 4096:.0f}")
