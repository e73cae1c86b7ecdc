"""Kernel time of the thinned-feed chains at C3 size (5M SNPs x 5k individuals), sizes 50 100 200 300 on one resident
panel: lod_feed_kernel's HIP-event time per size (what bench.py's also.c3_multi_winsize.thinned_feed reports)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from garlic_amd import abi, synth
import bench
nloci, nind = int(os.environ.get("SNPS", 5000000)), int(os.environ.get("INDS", 5000))
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
spec = synth.PanelSpec(nloci, seed=20260101 + 2, max_gap=200000)
ctx = abi.Context(0)
panel, _ = bench.load_panel(ctx, spec, nind, dev)
res = {}
for W in (50, 100, 200, 300):
    panel.lod_feed(W, 0.001, 200000, W, copy=False)
    ks = []
    for _ in range(4):
        panel.lod_feed(W, 0.001, 200000, W, copy=False)
        ks.append(round(panel.stats()["chain_kernel_ms"], 3))
    res[W] = ks
res["sum_of_means"] = round(sum(float(np.mean(v)) for v in res.values()), 2)
print(json.dumps(res), flush=True)
