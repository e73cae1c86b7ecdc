#!/usr/bin/env python3
"""Narrow-window wLOD (GARLIC's default --winsize 10) alone: plain and with likelihoods, NLOCI x NIND (2M x 1280), kernel ms
of a few passes each.  No child processes (usable under rocprofv3 --pmc)."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from garlic_amd import synth, abi

nloci = int(os.environ.get("NLOCI", 2_000_000))
nind = int(os.environ.get("NIND", 1280))
W = int(os.environ.get("W", 10))
steps = int(os.environ.get("STEPS", 6))
dev = torch.device("cuda:0")
ctx = abi.Context(0)
spec = synth.PanelSpec(nloci, seed=20260101 + 3, max_gap=bench.MAX_GAP)
panel, _ = bench.load_panel(ctx, spec, nind, dev, gq=True)
base, pitch, total = panel.out_layout(32, nind)
out = ctx.alloc_scores(total)
panel.compute_ld(W, want_output=False)
for name, gl, bytes_per in (("wlod", False, 8.25), ("wlod_gl", True, 16.25)):
    call = lambda: panel.wlod_windows_device(out.data_ptr(), W, bench.ERROR, bench.MAX_GAP, bench.M_GEN, bench.MU, use_gl=gl)
    dt, k = bench.timed_passes(ctx, call, steps, 2, torch.cuda.synchronize)
    chk = int(out.tensor().view(torch.int64).sum().item())
    print(json.dumps({"variant": os.environ.get("VARIANT", "shipped"), "leg": name, "W": W, "kernel_ms": k,
                      "frac_hbm": bytes_per * nloci * nind / (k * 1e-3) / 8e12, "checksum": chk}), flush=True)
