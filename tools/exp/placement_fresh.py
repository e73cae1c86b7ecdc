#!/usr/bin/env python3
"""Score-buffer placement across re-allocations inside one process (and across processes: run it twice): several
rounds of `n` plain buffers allocated, the C2 kernel timed into each, all freed again (really: empty_cache)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from garlic_amd import abi, synth
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
nloci, nind, W = 1_000_000, 1000, 100
spec = synth.PanelSpec(nloci, seed=20260102, max_gap=200000)
ctx = abi.Context(0)
panel, _ = bench.load_panel(ctx, spec, nind, dev)
base, pitch, total = panel.out_layout(32, nind)
def t3(ptr):
    for _ in range(2): panel.lod_windows_device(ptr, W, 0.001, 200000, pitch_align=32)
    ctx.synchronize()
    for _ in range(3): panel.lod_windows_device(ptr, W, 0.001, 200000, pitch_align=32)
    return float(np.mean(ctx.recent_kernel_ms(3)))
keep = []
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    bufs = [torch.empty(total, dtype=torch.float64, device=dev) for _ in range(6)]
    torch.cuda.synchronize()
    print("round", rnd, [round(t3(b.data_ptr()), 3) for b in bufs], flush=True)
    del bufs
    torch.cuda.empty_cache()
    if rnd == 1:      # a spoiler that stays: shifts what the next rounds get
        keep.append(torch.empty(int(1.3 * 2**30 / 8), dtype=torch.float64, device=dev))
