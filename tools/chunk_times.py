"""Diagnostic: per-chunk start/end cycle stamps of the cell sweep (library built with -DFLOODER_PHASE_TIMERS):
where is the tail?   usage: python tools/chunk_times.py [W] [cfg2|cfg3|cfg5]   (W: keep every W-th simplex, as one rank of W would)"""
import sys, torch, numpy as np
sys.path.insert(0, '.')
import flooder_amd as fa
from flooder_amd import _native, core
W = int(sys.argv[1]) if len(sys.argv) > 1 else 1
lib = _native.load()
core.CELL_SUPER = False  # (the per-chunk records below assume one work item per chunk)
torch.manual_seed(42)
dev = torch.device('cuda:0')
which = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
n_l = 1000
if which == "cfg3":
    pts = fa.generate_noisy_torus_points_3d(1_000_000, seed=42).to(dev)
elif which == "cfg5":
    from flooder_amd.synthetic import generate_swiss_cheese_points
    pts, n_l = generate_swiss_cheese_points(16_000_000, k=6, seed=42)[0].to(dev), 4000
else:
    pts = torch.randn(1_000_000, 3).to(dev)
lms = fa.generate_landmarks(pts, n_l, start_idx=0)
stree, simplices = core._build_complex(lms, 3)
simp = torch.as_tensor(simplices[3], device=dev)
verts = lms[simp]
order = torch.argsort(verts.mean(1)[:, 0])
verts = verts[order][0::W].contiguous()
weights, vi, fi = core.generate_grid(30, 3, dev, torch.float32)
faces = core._FaceTable(fi, weights.shape[0], dev)
index = core.PointIndex(pts)
S, R = verts.shape[0], weights.shape[0]
chunks = (R + 255) // 256
stats = torch.zeros(64 + 16 * S * chunks, dtype=torch.int64, device=dev)
for _ in range(3):
    stats.zero_()
    core._sweep_dimension_cell(index, verts, weights, faces, None, stats=stats)
torch.cuda.synchronize()
t = stats[64:].cpu().numpy().reshape(-1, 16)
ok = t[:, 1] > 0
t0, t1 = t[ok, 0].min(), t[ok, 1].max()
dur = ((t[:, 1] - t[:, 0]) * ok).astype(np.float64)
print(f"W={W} chunks {ok.sum()}  span {t1 - t0:.0f} ticks(10ns)  sum/2048 {dur.sum() / 2048:.0f}  max {dur.max():.0f} mean {dur[ok].mean():.0f}")
first_start = np.sort(t[ok, 0] - t0)
print("chunk starts (ticks(10ns) after the first): 1st..5th", first_start[:5], " 2048th", first_start[min(2047, len(first_start) - 1)])
late = np.argsort(-t[:, 1])[:12]
for g in late:
    s, q = divmod(int(g), chunks)
    info = int(t[g, 2])
    print(f"  ends at {t[g, 1] - t0:9.0f}  started {t[g, 0] - t0:9.0f}  dur {dur[g]:8.0f}  simplex {s} q {q}  n_cand {info & 0xfffff} "
          f"n_keep {(info >> 20) & 0xfffff} attempts {(info >> 40) & 15} exhaustive {(info >> 44) & 1}  gather steps {t[g, 3] >> 40} kcycles waiting for streamed rows (exhaustive) {(t[g, 3] & ((1 << 40) - 1)) // 1000} in flush {t[g, 4] // 1000}")
    names = ["pop", "samples", "gather0", "density", "gather1", "count", "prefix", "scatter", "query", "brute", "output"]
    print("      kcycles: " + " ".join(f"{n}={int(v) // 1000}" for n, v in zip(names, t[g, 5:16])))
# occupancy profile: how many waves are busy over time
edges = np.linspace(t0, t1, 21)
busy = [(np.minimum(t[ok, 1], b) - np.maximum(t[ok, 0], a)).clip(0).sum() / (b - a) for a, b in zip(edges[:-1], edges[1:])]
print("busy waves per 5% time slice:", np.round(busy).astype(int).tolist())

# where in the queue do the long chunks sit?  (queue position = simplex-major chunk id)
pos = np.nonzero(ok)[0]
d = dur[ok]
for a, b in zip(np.linspace(0, len(pos), 11)[:-1].astype(int), np.linspace(0, len(pos), 11)[1:].astype(int)):
    seg = d[a:b]
    print(f"queue {a:7d}..{b:7d}: mean {seg.mean():7.0f} p50 {np.percentile(seg, 50):7.0f} p99 {np.percentile(seg, 99):8.0f} max {seg.max():8.0f} ticks; started {(t[pos[a:b], 0] - t0).min():8.0f}..{(t[pos[a:b], 0] - t0).max():8.0f}")
# chunks still running in the last 15 % of the kernel: how long are they, when did they start?
late = ok & (t[:, 1] > t0 + 0.85 * (t1 - t0))
print(f"chunks ending in the last 15%: {late.sum()}, of them longer than 10x the mean: {(dur[late] > 10 * dur[ok].mean()).sum()}, "
      f"total wave-time {dur[late].sum() / (t1 - t0):.0f} wave-spans; exhaustive {((t[late, 2] >> 44) & 1).sum()}")
hist, edges2 = np.histogram(np.log10(np.maximum(d, 1)), bins=12)
print("log10(duration ticks) histogram:", [(round(float(e), 2), int(h)) for e, h in zip(edges2[:-1], hist)])
print("share of total wave-time by duration class (ticks): " + ", ".join(
    f"{lo}-{hi}: {d[(d >= lo) & (d < hi)].sum() / d.sum() * 100:.1f}%" for lo, hi in ((0, 2000), (2000, 5000), (5000, 10000), (10000, 20000), (20000, 10**9))))

print("longest chunks:")
names = ["pop", "samples", "gather0", "density", "gather1", "count", "prefix", "scatter", "query", "brute", "output"]
for g in np.argsort(-dur)[:40]:
    s_, q = divmod(int(g), chunks)
    info = int(t[g, 2])
    ph = " ".join(f"{n}={int(v) // 1000}" for n, v in zip(names, t[g, 5:16]) if int(v) >= 20000)
    print(f"  dur {dur[g]:7.0f} start {t[g, 0] - t0:8.0f} simplex {s_:5d} q {q:2d} n_cand {info & 0xfffff:6d} n_keep {(info >> 20) & 0xfffff:6d} "
          f"att {(info >> 40) & 15} exh {(info >> 44) & 1} steps {t[g, 3] >> 40:4d} | kcyc {ph}")
# how the long chunks split by kind
longm = ok & (dur > 10000)
exh = ((t[:, 2] >> 44) & 1).astype(bool)
print(f"chunks > 10000 ticks: {longm.sum()} (exhaustive {np.sum(longm & exh)}), wave-time share {dur[longm].sum() / dur[ok].sum() * 100:.1f}%; "
      f"phase shares inside them: " + " ".join(f"{n}={t[longm, 5 + i].sum() / t[longm, 5:16].sum() * 100:.0f}%" for i, n in enumerate(names)))

# exhaustive chunks (kept set larger than the LDS stage) against the others
for name, m in (("exhaustive", ok & exh), ("others", ok & ~exh)):
    if m.sum() == 0:
        continue
    print(f"{name}: {m.sum()} chunks, {dur[m].sum() / dur[ok].sum() * 100:.1f}% of the wave-time, mean {dur[m].mean():.0f} ticks, n_keep mean "
          f"{((t[m, 2] >> 20) & 0xfffff).mean():.0f}; phases: " + " ".join(f"{n}={t[m, 5 + i].sum() / t[m, 5:16].sum() * 100:.0f}%" for i, n in enumerate(names)))
att = (t[:, 2] >> 40) & 15
for a in range(1, 6):
    m = ok & (att == a)
    if m.sum():
        print(f"attempts {a}: {m.sum()} chunks, {dur[m].sum() / dur[ok].sum() * 100:.1f}% of the wave-time, mean {dur[m].mean():.0f} ticks")
