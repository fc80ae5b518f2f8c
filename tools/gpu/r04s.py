import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import xeve_amd
from xeve_amd import encode, lib
import bench
xeve_amd.init(0)
dev = torch.device("cuda", 0)
W, H, F = int(sys.argv[1]), int(sys.argv[2]), 8
G = int(sys.argv[3])
cfg = encode.config(W, H, qp=32, keyint=8, bframes=15, closed_gop=True, preset="medium", threads=8)
e = encode.BatchEncoder(cfg, 2, F)
e.begin(); total = e.advance(0); e.close()
per_picture = total // F
print("fused?", lib.load().xeve_hip_walk_fused(G * min(8, (H + 63) // 64)), "per_picture", per_picture, flush=True)
r = bench.class_profile(torch, dev, cfg, G, F, per_picture, W * H * 3 // 2)
print(json.dumps(r["kernels"]), flush=True)
