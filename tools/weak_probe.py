"""Per-rank step time of the weak-scaling job (W x 1M spots) for ONE virtual rank on one GPU: communication is
stubbed (halo rows arrive as zeros, reductions are local), so this is the compute + replicated-graph-build part."""
import os, sys, time, json
os.environ["FDX_DIST_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flashdeconv_amd.distributed import ShardedFlashDeconv

class FakeComm:
    def __init__(self, rank, world): self.rank, self.world = rank, world
    def all_reduce_max(self, t): pass
    def all_reduce_sum(self, t): pass
    def all_gather_rows(self, nbr, cnt, bounds):          # other ranks' rows: empty lists (valid input, fewer edges)
        lo, hi = int(bounds[self.rank]), int(bounds[self.rank + 1])
        nbr[:lo] = -1; nbr[hi:] = -1; cnt[:lo] = 0; cnt[hi:] = 0
    def exchange(self, send, recv):
        for t in recv.values(): t.zero_()

dev = torch.device("cuda", 0)
W = int(os.environ.get("W", 8)); rank = int(os.environ.get("R", 3)); per = int(os.environ.get("PER", 1_000_000))
n, G, K = W * per, 2000, 30
g = torch.Generator(device=dev); g.manual_seed(12345)
coords = torch.rand(n, 2, generator=g, device=dev, dtype=torch.float64) * float(np.sqrt(n))
X = torch.randn(K, G, generator=g, device=dev, dtype=torch.float64)
m = ShardedFlashDeconv(sketch_dim=512, preprocess="raw", n_hvg=G, comm=FakeComm(rank, W))
Xh = X.cpu().numpy()
own = m.plan(coords, Xh)
print("n_total", n, "n_own", m.n_own, "n_halo", m.n_halo, flush=True)
Y = torch.empty((m.n_own, G), device=dev, dtype=torch.float32)
for r0 in range(0, m.n_own, 1 << 17):
    r1 = min(m.n_own, r0 + (1 << 17))
    B = torch.rand(r1 - r0, K, generator=g, device=dev, dtype=torch.float64); B /= B.sum(dim=1, keepdim=True)
    Y[r0:r1] = (B @ X + 0.1 * torch.randn(r1 - r0, G, generator=g, device=dev, dtype=torch.float64)).to(torch.float32)
for _ in range(2):
    m.plan(coords, Xh); m.fit_transform(Y, Xh)
for _ in range(2):
    m.timings_ = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.plan(coords, Xh); m.fit_transform(Y, Xh)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(json.dumps({"wall_ms": round((t1 - t0) * 1e3, 2), **{k: round(v, 2) for k, v in m.timings_.items()}, "iters": m.info_["n_iterations"]}), flush=True)
m._profile = False
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.plan(coords, Xh); m.fit_transform(Y, Xh)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("unprofiled wall_ms", round((t1 - t0) * 1e3, 2), flush=True)
