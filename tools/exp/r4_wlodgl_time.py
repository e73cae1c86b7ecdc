#!/usr/bin/env python3
"""GL-weighted wLOD (wlod_strip_gl_kernel) alone: 2M x 1280 (or NLOCI x NIND), W from the WS list; prints one JSON line
per W with the kernel's HIP-event mean and a checksum of the scores (variants must agree bit for bit).
VARIANT tags the line (tools/exp/r4_variants.sh swaps the library)."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
import bench
from garlic_amd import synth, abi

nloci = int(os.environ.get("NLOCI", 2_000_000))
nind = int(os.environ.get("NIND", 1280))
sizes = [int(x) for x in os.environ.get("WS", "100").split(",")]
steps = int(os.environ.get("STEPS", 8))
dev = torch.device("cuda:0")
ctx = abi.Context(0)
spec = synth.PanelSpec(nloci, seed=20260101 + 3, max_gap=bench.MAX_GAP)
panel, _ = bench.load_panel(ctx, spec, nind, dev, gq=True)
base, pitch, total = panel.out_layout(32, nind)
out = ctx.alloc_scores(total)
for W in sizes:
    panel.compute_ld(W, want_output=False)
    call = lambda: panel.wlod_windows_device(out.data_ptr(), W, bench.ERROR, bench.MAX_GAP, bench.M_GEN, bench.MU, use_gl=True)
    dt, k = bench.timed_passes(ctx, call, steps, 2, torch.cuda.synchronize)
    crc = int(out.tensor().view(torch.int64).sum().item())      # wrapping sum of the score bits, whole buffer
    st = panel.stats()
    clk = bench.clock_under_load(call, torch.cuda.synchronize, 2.0) if os.environ.get("CLOCK") else None
    flops = 2.0 * nloci * nind * W
    print(json.dumps({"variant": os.environ.get("VARIANT", "shipped"), "W": W, "kernel_ms": k, "ms_per_pass": dt / steps * 1e3,
                      "frac_fp64": flops / (k * 1e-3) / 1e12 / bench.FP64_PEAK_TFLOPS, "crc": crc,
                      "reruns": int(st["n_stall_reruns"]), "clock": clk}), flush=True)
