#!/usr/bin/env python3
"""variance under rocprofv3 --pmc: several allocations of the output, 12 passes each; the profiler's CSV
gives per-dispatch counters and timestamps, tools/exp/variance_pmc_join.py groups them by allocation."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch
from garlic_amd import abi, synth
import bench

PASSES = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
nloci, nind, W = 1_000_000, 1000, 100
spec = synth.PanelSpec(nloci, seed=20260102, max_gap=200000)
ctx = abi.Context(0)
panel, _ = bench.load_panel(ctx, spec, nind, dev)
base, pitch, total = panel.out_layout(32, nind)
pads = []
for trial in range(8):
    if trial % 2 == 1:
        pads.append(torch.empty(int(np.random.default_rng(trial).integers(1, 1 << 28)), dtype=torch.uint8, device=dev))
    out = torch.empty(total, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    for _ in range(PASSES):
        panel.lod_windows_device(out.data_ptr(), W, 0.001, 200000)
    torch.cuda.synchronize()
    del out
    torch.cuda.empty_cache()
