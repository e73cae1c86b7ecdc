#!/usr/bin/env python3
"""does the kernel time of the buffer garlic_panel_alloc_scores picked stay what the probe measured?  C2 shape."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from garlic_amd import abi, synth
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
nloci, nind, W = 1_000_000, 1000, 100
spec = synth.PanelSpec(nloci, seed=20260102, max_gap=200000)
ctx = abi.Context(0)
panel, _ = bench.load_panel(ctx, spec, nind, dev)
ctx.set_async(True)
def passes(ptr, n):
    for _ in range(2): panel.lod_windows_device(ptr, W, 0.001, 200000)
    ctx.synchronize()
    for _ in range(n): panel.lod_windows_device(ptr, W, 0.001, 200000)
    return [round(x, 3) for x in ctx.recent_kernel_ms(n)]
if os.environ.get("PLAIN"):
    base, pitch, total = panel.out_layout(32, nind)
    plain = torch.empty(total, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    print("plain torch buffer", passes(plain.data_ptr(), 3))
    del plain
    torch.cuda.empty_cache()
ctx.set_async(False)
buf, times = panel.alloc_scores(W, 0.001, 200000, candidates=int(os.environ.get("CANDS", 8)))
ctx.set_async(True)
print("probe", [round(t, 3) for t in times], "kept", round(min(times), 3))
if os.environ.get("TENSOR"):
    out = buf.tensor()
    print("through .tensor().data_ptr()", passes(out.data_ptr(), 5), hex(out.data_ptr()), hex(buf.ptr))
print("after, 5 passes", passes(buf.ptr, 5))
print("after, 20 passes", passes(buf.ptr, 20))
ctx.trim()
print("after trim, 10 passes", passes(buf.ptr, 10))
