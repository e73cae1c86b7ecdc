#!/usr/bin/env python3
"""variance, part 4: is the slow/fast mode a property of the allocation alone?  Per allocation: the chain
kernel, a plain streaming fill (torch) and the C2-shaped store pattern of tools/ubench (via hipModule? no:
torch strided writes) on the same buffer."""
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

def ev_time(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

pads = []
for trial in range(8):
    if trial % 2 == 1:
        pads.append(torch.empty(int(np.random.default_rng(trial).integers(1, 1 << 28)), dtype=torch.uint8, device=dev))
    out = torch.empty(total, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    for _ in range(5):
        panel.lod_windows_device(out.data_ptr(), W, 0.001, 200000)
    torch.cuda.synchronize()
    for _ in range(30):
        panel.lod_windows_device(out.data_ptr(), W, 0.001, 200000)
    torch.cuda.synchronize()
    k = float(np.mean(ctx.recent_kernel_ms(30)))
    fill_ms = ev_time(lambda: out.fill_(1.0))
    # column-strided writes: a [rows][pitch] view, 32 columns of all rows per call (like one tile of every item)
    n0, p0 = int(spec.chr_nloci[0]), int(pitch[0])
    v = out[base[0]: base[0] + 1024 * p0].view(1024, p0)
    def tiles():
        for s in range(0, 4096, 32):
            v[:, s:s + 32] = 2.0
    tile_ms = ev_time(tiles, reps=3)
    print(json.dumps({"trial": trial, "chain_ms": round(k, 4), "fill_TBps": round(total * 8 / fill_ms / 1e9, 3),
                      "tile_writes_ms": round(tile_ms, 4)}), flush=True)
    del out, v
    torch.cuda.empty_cache()
