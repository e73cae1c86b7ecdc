#!/usr/bin/env python3
"""C2 chain kernel into score buffers allocated different ways: torch (hipMalloc) vs virtual ranges backed by
separately created physical chunks, mapped in order or shuffled."""
import ctypes as C, os, sys, time, json
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
L = abi.lib()
L.garlic_debug_alloc_scattered.argtypes = [C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.POINTER(C.c_void_p)]
L.garlic_debug_free_scattered.argtypes = [C.c_void_p]

def run(ptr):
    for _ in range(5):
        panel.lod_windows_device(ptr, W, 0.001, 200000)
    torch.cuda.synchronize()
    for _ in range(20):
        panel.lod_windows_device(ptr, W, 0.001, 200000)
    torch.cuda.synchronize()
    return float(np.mean(ctx.recent_kernel_ms(20)))

for trial in range(3):
    out = torch.empty(total, dtype=torch.float64, device=dev)
    print(json.dumps({"alloc": "torch", "trial": trial, "kernel_ms": run(out.data_ptr())}), flush=True)
    del out
    torch.cuda.empty_cache()
for chunk_mb in (2, 16, 128, 1024):
    for seed in (0, 1, 2):
        p = C.c_void_p()
        rc = L.garlic_debug_alloc_scattered(0, total * 8, chunk_mb << 20, seed, C.byref(p))
        if rc:
            print("alloc failed", chunk_mb, seed, L.garlic_hip_last_error()); continue
        print(json.dumps({"alloc": f"vmm chunk {chunk_mb} MB", "shuffle": seed, "kernel_ms": run(p.value)}), flush=True)
        L.garlic_debug_free_scattered(p)
# one large allocation cut into slices (slow in every slice when this was looked at before), then VMM again
big = torch.empty(total * 4 + 4096, dtype=torch.float64, device=dev)
for k in range(4):
    print(json.dumps({"alloc": "slice of one 32-GB torch allocation", "slice": k, "kernel_ms": run(big.data_ptr() + k * (total * 8 + 1024))}), flush=True)
p = C.c_void_p()
L.garlic_debug_alloc_scattered(0, total * 8, 2 << 20, 0, C.byref(p))
print(json.dumps({"alloc": "vmm chunk 2 MB while the big one lives", "kernel_ms": run(p.value)}), flush=True)
L.garlic_debug_free_scattered(p)
del big
torch.cuda.empty_cache()
outs = [torch.empty(total, dtype=torch.float64, device=dev) for _ in range(6)]
for k, o in enumerate(outs):
    print(json.dumps({"alloc": "six torch buffers side by side", "k": k, "kernel_ms": run(o.data_ptr())}), flush=True)
ps = []
for k in range(4):
    p = C.c_void_p()
    L.garlic_debug_alloc_scattered(0, total * 8, 2 << 20, 0, C.byref(p))
    ps.append(p)
    print(json.dumps({"alloc": "vmm side by side with the six", "k": k, "kernel_ms": run(p.value)}), flush=True)
