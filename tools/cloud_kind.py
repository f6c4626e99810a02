"""The density grid's cloud-kind words (flood_common.hpp: cloud_kind_kernel) for the BASELINE clouds:
points in interior cells / points in occupied cells.  usage: cloud_kind.py"""
import sys, torch
sys.path.insert(0, '.')
import flooder_amd as fa
from flooder_amd import core
dev = torch.device("cuda:0")
torch.manual_seed(42)
clouds = [("cfg2 gauss 1M", torch.randn(1_000_000, 3)), ("cfg3 torus 1M", fa.generate_noisy_torus_points_3d(1_000_000, seed=42)),
          ("cheese 1M", fa.generate_swiss_cheese_points(1_000_000, k=6, seed=42)[0]),
          ("cfg5 cheese 16M", fa.generate_swiss_cheese_points(16_000_000, k=6, seed=42)[0]),
          ("gauss 100k", torch.randn(100_000, 3)), ("torus 100k", fa.generate_noisy_torus_points_3d(100_000, seed=1)),
          ("annulus 2D 200k", fa.generate_annulus_points_2d(200_000, torch.tensor([0.0, 0.0]), 1.5, 0.4, seed=3)),
          ("gauss 2D 200k", torch.randn(200_000, 2))]
for name, p in clouds:
    idx = core.PointIndex(p.to(dev))
    nf = 256 * 256 if p.shape[1] == 2 else 64 ** 3
    k = idx.dens[nf:nf + 4].cpu().tolist()
    print(f"{name:18s} interior {k[2]:9d} of {k[3]:9d} points: {100.0 * k[2] / max(k[3], 1):5.1f} %")
