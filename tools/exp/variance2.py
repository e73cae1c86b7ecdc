#!/usr/bin/env python3
"""variance, part 2: is the slow/fast mode of the C2 chain kernel a property of the output buffer's
allocation (physical placement) or of its alignment?  One big allocation, sub-buffers at several byte
offsets; then fresh allocations obtained in different ways."""
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

def measure(ptr, tag):
    for _ in range(5):
        panel.lod_windows_device(ptr, W, 0.001, 200000)
    torch.cuda.synchronize()
    for _ in range(30):
        panel.lod_windows_device(ptr, W, 0.001, 200000)
    torch.cuda.synchronize()
    k = ctx.recent_kernel_ms(30)
    print(json.dumps({"tag": tag, "ptr": hex(ptr), "k_mean": round(float(np.mean(k)), 4), "k_min": round(float(np.min(k)), 4)}), flush=True)

for trial in range(3):
    big = torch.empty(total + (64 << 20), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    for off in (0, 32, 512, 8192, 262144, 1 << 20, 3 << 20, 32 << 20):     # in doubles: 0, 256 B, 4 KB, 64 KB, 2 MB, 8 MB, 24 MB, 256 MB
        measure(big.data_ptr() + off * 8, f"big{trial}+{off * 8}")
    del big
    torch.cuda.empty_cache()
    # a different allocation in between changes what the next one gets
    pad = torch.empty((trial + 1) * 123456789, dtype=torch.uint8, device=dev)
    out = torch.empty(total, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    measure(out.data_ptr(), f"after_pad{trial}")
    del out
    torch.cuda.empty_cache()
    out = torch.empty(total, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    measure(out.data_ptr(), f"realloc{trial}")
    del out, pad
    torch.cuda.empty_cache()
