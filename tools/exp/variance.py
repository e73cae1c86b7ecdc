#!/usr/bin/env python3
"""Where does the run-to-run spread of the C2 chain kernel (1.42 .. 1.65 ms on one box) come from?
Several allocations of the output in one process, many passes each, per-pass HIP-event times."""
import os, sys, time, json
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
panel, _ = bench.load_panel(ctx, spec, nind, dev)
base, pitch, total = panel.out_layout(32, nind)
res = []
pads = []
for trial in range(8):
    if trial % 2 == 1:
        pads.append(torch.empty(int(np.random.default_rng(trial).integers(1, 1 << 28)), dtype=torch.uint8, device=dev))
    out = torch.empty(total, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    for rep in range(3):
        for _ in range(5):
            panel.lod_windows_device(out.data_ptr(), W, 0.001, 200000)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            panel.lod_windows_device(out.data_ptr(), W, 0.001, 200000)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 30 * 1e3
        k = ctx.recent_kernel_ms(30)
        res.append({"trial": trial, "rep": rep, "ptr": hex(out.data_ptr()), "ms_per_pass": dt, "k_mean": float(np.mean(k)),
                    "k_min": float(np.min(k)), "k_max": float(np.max(k))})
        print(json.dumps(res[-1]), flush=True)
        if rep == 1:
            time.sleep(2.0)      # idle gap: does the clock state matter?
    del out
    torch.cuda.empty_cache()
