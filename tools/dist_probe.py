"""Stage times of the sharded driver with ONE rank (nccl), against the single-GPU fit.  FDX_DIST_TIMING=1 is set here."""
import os, sys, time, json
os.environ["FDX_DIST_TIMING"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
import bench
from flashdeconv_amd.distributed import ShardedFlashDeconv
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
Y, X, coords = bench.gen_gaussian(torch, 1_000_000, 2000, 30, dev, 0)
coords_d = torch.as_tensor(coords, device=dev, dtype=torch.float64) if not torch.is_tensor(coords) else coords.to(dev, torch.float64)
Xh = X if isinstance(X, np.ndarray) else X.cpu().numpy()
m = ShardedFlashDeconv(sketch_dim=512, preprocess="raw", n_hvg=2000)
own = m.plan(coords_d)
Yo = Y[own] if torch.is_tensor(Y) else torch.as_tensor(Y, device=dev)[own]
for _ in range(2):
    m.plan(coords_d, Xh); m.fit_transform(Yo, Xh)
for _ in range(3):
    m.timings_ = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.plan(coords_d, Xh); m.fit_transform(Yo, Xh)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(json.dumps({"wall_ms": round((t1 - t0) * 1e3, 2), **{k: round(v, 2) for k, v in m.timings_.items()}, "iters": m.info_["n_iterations"]}))
m._profile = False
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.plan(coords_d, Xh); m.fit_transform(Yo, Xh)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("unprofiled wall_ms", round((t1 - t0) * 1e3, 2))
dist.destroy_process_group()
