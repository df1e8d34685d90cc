"""CSR input at scale: n spots x G_all genes (~5 % stored), HVG to 2000 genes, log_cpm; stage times vs the dense path."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flashdeconv_amd import FlashDeconv, _lib
dev = torch.device("cuda", 0)
n = int(os.environ.get("N", 1_000_000)); G = int(os.environ.get("G", 20000)); K = 30
g = torch.Generator(device=dev); g.manual_seed(0)
X = torch.exp(torch.randn(K, G, generator=g, device=dev, dtype=torch.float64) * 1.2 - 1.0)
for k in range(K):
    idx = torch.randperm(G, generator=g, device=dev)[:40]
    X[k, idx] *= 8
side = int(np.ceil(np.sqrt(n)))
ii = torch.arange(n, device=dev)
coords = torch.stack([(ii % side).double(), (ii // side).double()], dim=1)
coords += torch.randn(n, 2, generator=g, device=dev, dtype=torch.float64) * 0.1
centres = torch.rand(K, 2, generator=g, device=dev, dtype=torch.float64) * side
crow, col, val = [torch.zeros(1, dtype=torch.int64, device=dev)], [], []
step = 1 << 15
nnz = 0
for r0 in range(0, n, step):
    r1 = min(n, r0 + step)
    B = torch.exp(-torch.cdist(coords[r0:r1], centres) / (side / 2)); B /= B.sum(dim=1, keepdim=True)
    lam = (B @ X)
    lam *= (float(os.environ.get("DEPTH", 1500)) / lam.sum(dim=1, keepdim=True))
    Yc = torch.poisson(lam, generator=g).to(torch.float32).to_sparse_csr()
    crow.append(Yc.crow_indices()[1:] + nnz); col.append(Yc.col_indices().to(torch.int32)); val.append(Yc.values())
    nnz += int(Yc.values().numel())
Y = torch.sparse_csr_tensor(torch.cat(crow), torch.cat(col), torch.cat(val), size=(n, G), device=dev)
del crow, col, val
print(f"n={n} G={G} nnz={nnz} density={nnz / n / G:.4f} nnz/row={nnz / n:.0f}", flush=True)
Xh = X.cpu().numpy()
for pre in ("log_cpm", "raw"):
    m = FlashDeconv(sketch_dim=512, preprocess=pre, n_hvg=2000, max_iter=int(os.environ.get("ITERS", 20)))
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.fit(Y, Xh, coords, output="torch")
        torch.cuda.synchronize(); t1 = time.perf_counter()
    print(pre, "CSR fit wall_ms", round((t1 - t0) * 1e3, 2), {k: round(v, 2) for k, v in m.timings_.items()}, "genes", len(m.gene_idx_), "iters", m.info_["n_iterations"], flush=True)
csr = _lib.CsrOnDevice.from_torch(Y)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); csr.gene_moments(); t1 = time.perf_counter()
print("csr gene moments ms", round((t1 - t0) * 1e3, 2))
# dense path on the selected genes for comparison
gi = torch.as_tensor(m.gene_idx_, device=dev)
Yd = torch.empty((n, len(gi)), dtype=torch.float32, device=dev)
for r0 in range(0, n, step):
    r1 = min(n, r0 + step)
    rows = torch.sparse_csr_tensor(Y.crow_indices()[r0:r1 + 1] - Y.crow_indices()[r0], Y.col_indices()[Y.crow_indices()[r0]:Y.crow_indices()[r1]],
                                   Y.values()[Y.crow_indices()[r0]:Y.crow_indices()[r1]], size=(r1 - r0, G), device=dev)
    Yd[r0:r1] = rows.to_dense()[:, gi]
Xs = Xh[:, m.gene_idx_]
for pre in ("log_cpm", "raw"):
    md = FlashDeconv(sketch_dim=512, preprocess=pre, n_hvg=2000, max_iter=int(os.environ.get("ITERS", 20)))
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        md.fit(Yd, Xs, coords, output="torch")
        torch.cuda.synchronize(); t1 = time.perf_counter()
    print(pre, "dense(selected) fit wall_ms", round((t1 - t0) * 1e3, 2), {k: round(v, 2) for k, v in md.timings_.items()}, flush=True)
