"""float64 device path: wall time of flood_complex on float64 tensors vs float32.  usage: python tools/time_f64.py"""
import sys, time, warnings, torch
sys.path.insert(0, '.')
warnings.simplefilter("ignore")
import flooder_amd as fa
dev = torch.device('cuda:0')
torch.manual_seed(42)
pts = torch.randn(1_000_000, 3)
for dt in (torch.float32, torch.float64):
    tp = pts.to(dev, dtype=dt)
    lms = fa.generate_landmarks(tp, 1000, start_idx=0)
    fa.flood_complex(tp[:10000], lms); torch.cuda.synchronize()
    t0 = time.perf_counter(); st = fa.flood_complex(tp, lms, return_simplex_tree=True); torch.cuda.synchronize()
    print(dt, f"{(time.perf_counter() - t0) * 1e3:.1f} ms end to end", flush=True)
