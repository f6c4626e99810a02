"""Host-side profile of flood_complex at cfg 2 (cProfile over 20 calls after warm-up): where the milliseconds around
the 1.2 ms device step go.  python tools/e2e_profile.py [tree|dict]"""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flooder_amd as fa
mode = sys.argv[1] if len(sys.argv) > 1 else "tree"
dev = torch.device("cuda:0")
torch.manual_seed(42)
pts = torch.randn(1_000_000, 3).to(dev)
lms = fa.generate_landmarks(pts, 1000, start_idx=0)
for _ in range(3):
    fa.flood_complex(pts, lms, return_simplex_tree=(mode == "tree"))
torch.cuda.synchronize()
t = []
for _ in range(10):
    t0 = time.perf_counter(); fa.flood_complex(pts, lms, return_simplex_tree=(mode == "tree")); torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
print("median ms", sorted(t)[5] * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    fa.flood_complex(pts, lms, return_simplex_tree=(mode == "tree"))
torch.cuda.synchronize()
pr.disable()
ps = pstats.Stats(pr); ps.sort_stats("cumulative").print_stats(45)
