import os, sys, time
os.environ["FDX_TRACE_HOST"] = "1"
sys.path.insert(0, "/root/repo")
import torch, bench
from flashdeconv_amd import FlashDeconv
dev = torch.device("cuda:0")
Y, X, coords = bench.gen_gaussian(torch, 1_000_000, 2000, 30, dev, 0)
m = FlashDeconv(sketch_dim=512, preprocess="raw", n_hvg=2000)
for i in range(4):
    m.fit(Y, X, coords, output="torch")
torch.cuda.synchronize()

for i in range(3):
    print("---- fit", i, file=sys.stderr)
    m.fit(Y, X, coords, output="torch")
    torch.cuda.synchronize()
    print({k: round(v, 3) for k, v in m.timings_.items()}, file=sys.stderr)
