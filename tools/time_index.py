import sys, time, torch
sys.path.insert(0, '.')
from flooder_amd import core, _native
lib = _native.load()
dev = torch.device('cuda:0')
torch.manual_seed(0)
pts = torch.randn(1_000_000, 3, device=dev)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("PointIndex total", t(lambda: core.PointIndex(pts)))
print("min+max", t(lambda: (pts.min(dim=0).values, pts.max(dim=0).values)))
print("min+max+cpu", t(lambda: torch.stack([pts.min(dim=0).values, pts.max(dim=0).values]).cpu()))
codes = torch.randint(0, 2**62, (1_000_000,), device=dev)
print("argsort int64", t(lambda: torch.argsort(codes)))
c32 = torch.randint(0, 2**30, (1_000_000,), device=dev, dtype=torch.int32)
print("argsort int32", t(lambda: torch.argsort(c32)))
print("sort int64 (values+idx)", t(lambda: torch.sort(codes)))
order = torch.argsort(codes)
print("gather rows", t(lambda: pts[order]))
print("full(inf) 1M x4", t(lambda: torch.full((1_000_000, 4), float('inf'), device=dev)))
def build_padded():
    p = torch.full((1_000_000, 4), float('inf'), device=dev); p[:, :3] = pts[order]; p[:, 3:] = 0; return p
print("padded build", t(build_padded))
