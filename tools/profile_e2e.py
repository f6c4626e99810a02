import sys, time, cProfile, pstats, torch
sys.path.insert(0, '.')
import flooder_amd as fa
dev = torch.device('cuda:0')
torch.manual_seed(42)
pts = torch.randn(1_000_000, 3).to(dev)
lms = fa.generate_landmarks(pts, 1000, start_idx=0)
fa.flood_complex(pts[:10000], lms); torch.cuda.synchronize()
for rs in (False, True):
    t0 = time.perf_counter(); out = fa.flood_complex(pts, lms, return_simplex_tree=rs); torch.cuda.synchronize(); print("return_simplex_tree", rs, "wall", time.perf_counter() - t0)
pr = cProfile.Profile(); pr.enable(); out = fa.flood_complex(pts, lms); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
