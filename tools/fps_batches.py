"""Diagnostic of the batched landmark selection (flooder_fps_batched_f32): landmarks per launch, what closed each batch,
time inside a launch and to the next one - from the records block 0 leaves (core.FPS_KEEP_DIAG).
usage: python tools/fps_batches.py [cfg2|cfg5|cfg4]"""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
import flooder_amd as fa
from flooder_amd import core

which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
dev = torch.device('cuda:0')
torch.manual_seed(42)
if which == "cfg5":
    pts, k = fa.generate_swiss_cheese_points(16_000_000, k=6, seed=42)[0], 4000
elif which == "cfg4":
    pts, k = torch.randn(2_000_000, 6), 2000
else:
    pts, k = torch.randn(1_000_000, 3), 1000
pts = pts.to(dev)
index = core.PointIndex(pts)
core.FPS_BATCHED = True
core.FPS_KEEP_DIAG = True
for _ in range(2):
    core.fps_indices(pts, k, 0, method="bucket", index=index)
torch.cuda.synchronize()
d = core.LAST_FPS_DIAG
words, blocks = d["words"], d["blocks"]
ctr = d["ctr"].cpu().numpy()
rec = d["rec"].cpu().numpy().view(np.uint32).reshape(-1, blocks, words)
L = core.LAST_FPS_LAUNCHES
its = ctr[:rec.shape[0]]
done = [i for i in range(1, rec.shape[0]) if its[i] > its[i - 1]]
batch = np.array([its[i] - its[i - 1] for i in done])
why = np.array([rec[i, 0, 3] for i in done])
t0 = np.array([rec[i, 0, 8] for i in done]).astype(np.int64)
t1 = np.array([rec[i, 0, 9] for i in done]).astype(np.int64)
inside = ((t1 - t0) & 0xffffffff) / 100.0                      # 100 MHz ticks -> us
gap = ((t0[1:] - t1[:-1]) & 0xffffffff) / 100.0
print(f"{which}: {L} launches in all, {len(done)} batched launches that selected something, {batch.sum()} landmarks in them: {batch.mean():.2f} per launch")
print("landmarks per launch:", {int(b): int((batch == b).sum()) for b in np.unique(batch)})
names = {1: "nothing above the bound of the other points", 2: "a skipped (lowered) candidate may come first", 3: "points hidden behind an accepted one", 4: "KMAX / all selected"}
print("what closed the batch:", {names.get(int(w), int(w)): int((why == w).sum()) for w in np.unique(why)})
print(f"inside a launch: mean {inside.mean():.2f} us (p10 {np.percentile(inside, 10):.2f}, p90 {np.percentile(inside, 90):.2f}); to the next launch: mean {gap.mean():.2f} us")
for b in np.unique(batch):
    print(f"   batches of {int(b)}: {inside[batch == b].mean():.2f} us inside")
