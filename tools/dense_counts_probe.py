"""dense log-CPM sketch on small-count data (1M x 2000, ~0.75 counts per entry) with and without the log1p table"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flashdeconv_amd import FlashDeconv
dev = torch.device("cuda", 0)
n, G, K = 1_000_000, 2000, 30
g = torch.Generator(device=dev); g.manual_seed(0)
X = torch.exp(torch.randn(K, G, generator=g, device=dev, dtype=torch.float64) * 1.0 - 0.5)
coords = torch.rand(n, 2, generator=g, device=dev, dtype=torch.float64) * 1000
Y = torch.empty((n, G), device=dev, dtype=torch.float32)
for r0 in range(0, n, 1 << 16):
    r1 = min(n, r0 + (1 << 16))
    B = torch.rand(r1 - r0, K, generator=g, device=dev, dtype=torch.float64); B /= B.sum(dim=1, keepdim=True)
    lam = B @ X; lam *= 1500.0 / lam.sum(dim=1, keepdim=True)
    Y[r0:r1] = torch.poisson(lam, generator=g).to(torch.float32)
print("mean count", float(Y[:1000].mean()), "max", float(Y[:100000].max()))
m = FlashDeconv(sketch_dim=512, preprocess="log_cpm", n_hvg=G, max_iter=5)
for rep in range(3): m.fit(Y, X.cpu().numpy(), coords, output="torch")
print(json.dumps({k: round(v, 2) for k, v in m.timings_.items()}), "no_table" if os.environ.get("FDX_NO_LOG_TABLE") else "table")
