"""Host-parallel Delaunay of the cfg 4 landmarks (2000 FPS landmarks of 2 M 6-D Gaussian points) and its face tables
at several thread counts.  usage: time_delaunay.py [threads ...]   (FLOODER_DELAUNAY_VERBOSE=1 for the routine's own split)"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import flooder_amd as fa
from flooder_amd import core, simplex_tree as stm

torch.manual_seed(42)
pts = torch.randn(2_000_000, 6)
if torch.cuda.is_available():
    lms = fa.generate_landmarks(pts.to("cuda:0"), 2000, start_idx=0).cpu()
else:
    lms = pts[torch.randperm(2_000_000)[:2000]]
L = lms.numpy().astype(np.float64)
for t in [int(a) for a in sys.argv[1:]] or [0]:
    stm.DELAUNAY_THREADS = t
    best = None
    for rep in range(3):
        t0 = time.perf_counter(); cells = stm.delaunay_cells(L); t1 = time.perf_counter()
        st, simp = core._build_complex(lms, 2); t2 = time.perf_counter()
        best = min(best, (t1 - t0, t2 - t1)) if best else (t1 - t0, t2 - t1)
    print(f"threads {t or stm._host_threads()}: delaunay_cells {best[0]*1e3:.0f} ms ({len(cells)} cells, exact calls {stm.LAST_DELAUNAY.get('exact_calls')}), "
          f"_build_complex(max_dimension=2) {best[1]*1e3:.0f} ms", flush=True)
