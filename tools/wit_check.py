"""Witness sweep A/B on the GPU box: face values with CELL_WITNESS on / off (must be bit-identical), the witness
sweep's counters, step times.  usage: python tools/wit_check.py [cfg2|cfg3|cfg5|small] [steps]"""
import sys, time
import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flooder_amd as fa
from flooder_amd import core, _native
from flooder_amd import synthetic

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
opts = [a for a in sys.argv[3:] if "=" in a]
dev = torch.device("cuda:0")
lib = _native.load()
for o in opts:
    k, v = o.split("=")
    assert lib.flooder_set_option(k.encode(), int(v)) == 0, o
torch.manual_seed(42)
if wl == "cfg2":
    pts, n_l = torch.randn(1_000_000, 3), 1000
elif wl == "small":
    pts, n_l = torch.randn(100_000, 3), 300
elif wl == "cfg3":
    from oracle import flood_oracle as fo
    pts, n_l = torch.as_tensor(fo.noisy_torus(1_000_000, seed=42)), 1000
pts = pts.to(dev).float().contiguous()
lms = fa.generate_landmarks(pts, n_l, start_idx=0)
d = 3
stree, simplices = core._build_complex(lms, d)
simp = torch.as_tensor(simplices[d], device=dev)
verts = lms[simp]
axis = int(torch.argmax(pts.max(0).values - pts.min(0).values).item())
order_s = torch.argsort(verts.mean(1)[:, axis])
verts = verts[order_s].contiguous()
weights, vertex_idxs, face_idxs = core.generate_grid(30, d, dev, torch.float32)
faces = core._FaceTable(face_idxs, weights.shape[0], dev)
plan = core.SamplePlan(weights, faces)
rows = stree._locate(d, np.sort(simp[order_s].cpu().numpy(), axis=1))
slots = core.shared_face_slots(stree, d, rows, [v.cpu().numpy() for v in vertex_idxs], dev)
index = core.PointIndex(pts)
S, R = verts.shape[0], weights.shape[0]
print("S", S, "R", R, "coarse", plan.wit[2] if plan.wit else None)


def run(wit, with_stats=False, shared=True):
    core.CELL_WITNESS = wit
    st = torch.zeros(96 + 12 * S, dtype=torch.int64, device=dev) if with_stats else None   # (timer builds also keep a record per simplex)
    out, _ = core._sweep_dimension_cell(index, verts, weights, faces, None, stats=st, plan=plan,
                                        face_slots=slots[:2] if shared else None)
    torch.cuda.synchronize()
    return out, (st.cpu().numpy() if st is not None else None)


for shared in (True, False):
    a, _ = run(False, shared=shared)
    b, st = run(True, with_stats=True, shared=shared)
    same = torch.equal(a.view(torch.int32), b.view(torch.int32))
    nd = int((a.view(torch.int32) != b.view(torch.int32)).sum().item())
    print(f"shared_slots={shared}: bit-identical {same} ({nd} of {a.numel()} differ); max abs diff {float((a - b).abs().max()):.3e}")
    if not same:
        bad = torch.nonzero(a.view(torch.int32).reshape(-1) != b.view(torch.int32).reshape(-1)).reshape(-1)[:10]
        print("  first diffs", [(int(i), float(a.reshape(-1)[i]), float(b.reshape(-1)[i])) for i in bad])
    names = ["handled", "too_heavy", "gather_over", "too_dense", "staged", "coarse_cert", "live", "rounds", "unresolved",
             "tiles_flagged", "pairs", "bins"]
    w = st[16:28]
    print("  exact passes", int(st[38]), "gather overflow there", int(st[39]))
    print("  wit:", {n: int(v) for n, v in zip(names, w)})
    if w[0] > 0:
        print(f"  per handled simplex: staged {w[4] / w[0]:.0f} coarse_cert {w[5] / w[0]:.0f} live {w[6] / w[0]:.0f} rounds {w[7] / w[0]:.2f} "
              f"unresolved {w[8] / w[0]:.1f} tiles {w[9] / w[0]:.2f} bins {w[11] / w[0]:.1f}")
    if st[28:38].sum() > 0:
        ph = st[28:38].astype(float)
        print("  wit phases % (setup, gather, hist, stage, coarse, fine, rounds, flag, pop, -):", [round(100 * v / ph.sum(), 1) for v in ph], "Mcycles/item", round(ph.sum() / max(w[0], 1) / 1e6, 3))
    print("  cell:", [int(v) for v in st[:9]], "finish:", [int(v) for v in st[9:16]])

res = {}
for wit in (False, True, False, True):
    core.CELL_WITNESS = wit
    tm = core._KernelTimer()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record()
    for i in range(steps):
        core._sweep_dimension_cell(index, verts, weights, faces, None, plan=plan, face_slots=slots[:2], timer=tm)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(steps)])
    res[wit] = (ms.mean(), ms.min(), {k: round(v / steps, 4) for k, v in tm.totals_ms().items()})
    print(f"witness={wit}: sweep-only step {ms.mean():.4f} ms (min {ms.min():.4f})", {k: round(v / steps, 4) for k, v in tm.totals_ms().items()})
print("SUMMARY", wl, " ".join(opts), f"| wit {res[True][0]:.4f} (min {res[True][1]:.4f}) {res[True][2]} | off {res[False][0]:.4f}")
