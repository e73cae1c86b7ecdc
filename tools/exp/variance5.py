#!/usr/bin/env python3
"""variance, part 5: ONE output buffer, the panel (packed genotypes, term table) re-created several times
with perturbing allocations in between: does the placement of the READ buffers decide fast / slow?"""
import os, sys, json
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch
from garlic_amd import abi, synth
import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
nloci, nind, W = 1_000_000, 1000, 100
spec = synth.PanelSpec(nloci, seed=20260102, max_gap=200000)
ctx = abi.Context(0)
ctx.set_async(True)
outs = []
pads = []
panel, _ = bench.load_panel(ctx, spec, nind, dev)
base, pitch, total = panel.out_layout(32, nind)
for k in range(2):
    outs.append(torch.empty(total, dtype=torch.float64, device=dev))
    pads.append(torch.empty(77777777 * (k + 1), dtype=torch.uint8, device=dev))
for trial in range(6):
    row = {"panel": trial}
    for k, out in enumerate(outs):
        for _ in range(4):
            panel.lod_windows_device(out.data_ptr(), W, 0.001, 200000)
        torch.cuda.synchronize()
        for _ in range(16):
            panel.lod_windows_device(out.data_ptr(), W, 0.001, 200000)
        torch.cuda.synchronize()
        row[f"out{k}"] = round(float(np.mean(ctx.recent_kernel_ms(16))), 4)
    print(json.dumps(row), flush=True)
    panel.close()
    pads.append(torch.empty(int(np.random.default_rng(trial).integers(1 << 20, 1 << 29)), dtype=torch.uint8, device=dev))
    panel, _ = bench.load_panel(ctx, spec, nind, dev)
