#!/usr/bin/env python3
"""Selected stored entries per spot of bench.py's CSR family (what the fused CSR sketch keeps in LDS between its two passes):
distribution against the keep buffer's capacity.  usage: python tools/csr_kept_probe.py [n_spots]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from flashdeconv_amd import FlashDeconv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
dev = torch.device("cuda:0")
Y, X, coords = bench.gen_sparse(torch, n, 20000, 30, dev, seed=0)
m = FlashDeconv(sketch_dim=512, preprocess="log_cpm", n_hvg=2000, max_iter=2)
m.fit(Y, X, coords, output="torch")
sel = torch.zeros(20000, dtype=torch.bool, device=dev)
sel[torch.as_tensor(m.gene_idx_, device=dev)] = True
hit = sel[Y.col_indices()].to(torch.int32)
cs = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), hit.cumsum(0)])
per_row = (cs[Y.crow_indices()[1:]] - cs[Y.crow_indices()[:-1]]).float()
stored = (Y.crow_indices()[1:] - Y.crow_indices()[:-1]).float()
q = torch.tensor([0.0, 0.5, 0.9, 0.99, 1.0], device=dev)
print("selected genes", len(m.gene_idx_), "stored per spot", stored.mean().item(), "selected per spot: mean", per_row.mean().item(),
      "quantiles", torch.quantile(per_row, q).tolist())
for cap in (384, 512, 576, 640, 768, 1024):
    print("cap", cap, "rows over", (per_row > cap).float().mean().item())
