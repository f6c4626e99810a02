"""Diagnostic: per-chunk duration of the cell sweep (library built with -DFLOODER_PHASE_TIMERS): where is the tail?"""
import sys, torch, numpy as np
sys.path.insert(0, '.')
import flooder_amd as fa
from flooder_amd import _native, core
lib = _native.load()
torch.manual_seed(42)
dev = torch.device('cuda:0')
pts = torch.randn(1_000_000, 3).to(dev)
lms = fa.generate_landmarks(pts, 1000, start_idx=0)
stree, simplices = core._build_complex(lms, 3)
simp = torch.as_tensor(simplices[3], device=dev)
verts = lms[simp]
weights, vi, fi = core.generate_grid(30, 3, dev, torch.float32)
faces = core._FaceTable(fi, weights.shape[0], dev)
index = core.PointIndex(pts)
S, R = verts.shape[0], weights.shape[0]
chunks = (R + 255) // 256
stats = torch.zeros(64 + S * chunks, dtype=torch.int64, device=dev)
for _ in range(2):
    stats.zero_()
    core._sweep_dimension_cell(index, verts, weights, faces, None, stats=stats)
torch.cuda.synchronize()
t = stats[64:].cpu().numpy().astype(float)
print("chunks", t.size, "sum/2048 waves (cycles)", t.sum() / 2048, "max", t.max(), "mean", t.mean(), "median", np.median(t))
for q in (50, 90, 99, 99.9, 99.99):
    print("pct", q, np.percentile(t, q))
order = np.argsort(-t)[:30]
vol = torch.linalg.det((verts[:, 1:] - verts[:, :1])).abs().cpu().numpy() / 6
cen = verts.mean(1).norm(dim=1).cpu().numpy()
for g in order:
    s, q = divmod(int(g), chunks)
    print(f"chunk {g} simplex {s} q {q} cycles {t[g]:.0f} vol {vol[s]:.4f} |centre| {cen[s]:.2f}")
# greedy list schedule in queue order on 2048 waves -> makespan vs ideal
import heapq
h = [0.0] * 2048
heapq.heapify(h)
for v in t:
    heapq.heappush(h, heapq.heappop(h) + v)
print("list-schedule makespan (cycles)", max(h), "ideal", t.sum() / 2048)
h = [0.0] * 2048
heapq.heapify(h)
for v in np.sort(t)[::-1]:
    heapq.heappush(h, heapq.heappop(h) + v)
print("LPT makespan (cycles)", max(h))
