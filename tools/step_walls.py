"""Per-step wall of one bench family, step by step (which warm steps stall?).

    python tools/step_walls.py sparse|counts|gaussian|lattice [steps] [--trace]

Every step is bracketed by torch.cuda.synchronize(); with --trace FDX_TRACE_HOST=1 is set for the whole run, so the host marks
of a slow step sit right above its line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fam = sys.argv[1] if len(sys.argv) > 1 else "sparse"
steps = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 20
if "--trace" in sys.argv:
    os.environ["FDX_TRACE_HOST"] = "1"
import numpy as np
import torch
import bench
from flashdeconv_amd import FlashDeconv

dev = torch.device("cuda", 0)
n, G, K, d = 1_000_000, 2000, 30, 512
if fam == "sparse":
    Y, X, coords = bench.gen_sparse(torch, n, 20000, K, dev, 0)
    kw = dict(sketch_dim=d, preprocess="log_cpm", n_hvg=G, max_iter=20)
elif fam == "gaussian":
    Y, X, coords = bench.gen_gaussian(torch, n, G, K, dev, 0)
    kw = dict(sketch_dim=d, preprocess="raw", n_hvg=G)
else:
    Y, X, coords = bench.gen_counts(torch, n, G, K, dev, 0)
    kw = dict(sketch_dim=d, preprocess="log_cpm", n_hvg=G)
    if fam == "lattice":
        side = int(np.ceil(np.sqrt(n)))
        ii = torch.arange(n, device=dev)
        coords = torch.stack([(ii % side).double(), (ii // side).double()], dim=1)
import warnings
warnings.simplefilter("ignore")
m = FlashDeconv(**kw)
walls = []
for s in range(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.fit(Y, X, coords, output="torch")
    torch.cuda.synchronize()
    w = (time.perf_counter() - t0) * 1e3
    walls.append(w)
    t = m.timings_
    acc = t["host_pre_ms"] + t["span_ms"] + t["host_post_ms"]
    print(json.dumps({"step": s, "wall_ms": round(w, 2), "unaccounted": round(w - acc, 2),
                      **{k: round(v, 2) for k, v in t.items() if k in ("host_pre_ms", "span_ms", "host_post_ms", "select_ms", "prologue_ms",
                                                                      "sketch_ms", "solve_ms", "finish_ms", "leverage_wait_ms", "graph_ms",
                                                                      "ties_remedy_ms")}}), flush=True)
ws = sorted(walls[2:])
print(json.dumps({"family": fam, "steps": steps, "min": round(ws[0], 2), "median": round(ws[len(ws) // 2], 2), "max": round(ws[-1], 2),
                  "mean": round(sum(ws) / len(ws), 2)}))
