"""Diagnostic: batch sizes of the batched FPS and what closed each batch.  usage: python tools/fps_batches.py [1m|6d|16m]"""
import sys, torch, numpy as np
sys.path.insert(0, '.')
import flooder_amd as fa
from flooder_amd import core
dev = torch.device('cuda:0')
torch.manual_seed(42)
which = sys.argv[1] if len(sys.argv) > 1 else "1m"
if which == "16m":
    pts, k = fa.generate_swiss_cheese_points(16_000_000, k=6, seed=42)[0], 4000
elif which == "6d":
    pts, k = torch.randn(2_000_000, 6), 2000
else:
    pts, k = torch.randn(1_000_000, 3), 1000
tp = pts.to(dev)
index = core.PointIndex(tp)
core.FPS_KEEP_DIAG = True
core.fps_indices(tp, k, 0, method="bucket", index=index)
torch.cuda.synchronize()
d = core.LAST_FPS_DIAG
ctr = d["ctr"].cpu().numpy()
rec = d["rec"].cpu().numpy().reshape(-1, d["blocks"], d["words"])
L = np.arange(1, len(ctr) - 1)
step = ctr[L + 1] - ctr[L]
used = step > 0
sizes = step[used]
why = rec[L[used] + 1, 0, 3]
print(which, "launches that selected:", int(used.sum()), "enqueued:", core.LAST_FPS_LAUNCHES, "mean batch", sizes.mean())
print("batch size histogram", np.bincount(sizes))
print("closed by: 1 nothing above B / 2 conflict / 3 hidden / 4 KMAX-or-end", np.bincount(why, minlength=5))
for part in range(4):
    q = slice(part * len(sizes) // 4, (part + 1) * len(sizes) // 4)
    print(" quarter", part, "mean batch %.2f" % sizes[q].mean(), "why", np.bincount(why[q], minlength=5))
t0 = rec[L[used] + 1, 0, 8].astype(np.int64) & 0xffffffff
t1 = rec[L[used] + 1, 0, 9].astype(np.int64) & 0xffffffff
dur = ((t1 - t0) & 0xffffffff) / 100.0
gap = ((t0[1:] - t1[:-1]) & 0xffffffff) / 100.0
print("block 0: in-kernel us per launch: mean %.2f median %.2f; gap to the next launch: mean %.2f median %.2f; sum %.2f ms"
      % (dur.mean(), np.median(dur), gap.mean(), np.median(gap), (dur.sum() + gap.sum()) / 1e3))
for part in range(4):
    q = slice(part * len(sizes) // 4, (part + 1) * len(sizes) // 4)
    print(" quarter", part, "in-kernel %.2f us" % dur[q].mean(), "gap %.2f us" % gap[q][:-1].mean(), "batch %.2f" % sizes[q].mean())
for nbv in (1, 2, 4, 8, 12, 16):
    sel = sizes == nbv
    if sel.sum() > 2:
        print("  batch size", nbv, "in-kernel %.2f us (n=%d)" % (dur[sel].mean(), sel.sum()))
