#!/usr/bin/env python3
"""wall time of garlic_lod_feed_multi vs single garlic_lod_feed calls at C3 size (5M x 5k, W = 50 100 200 300)"""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from garlic_amd import abi, synth
nloci, nind = int(os.environ.get("SNPS", 5000000)), int(os.environ.get("INDS", 5000))
sizes = [50, 100, 200, 300]
dev = torch.device("cuda:0")
spec = synth.PanelSpec(nloci, seed=20260103, max_gap=200000)
with abi.Context(0) as ctx, abi.Panel(ctx, spec.chr_nloci, nind) as panel:
    panel.set_map(spec.pos, spec.centro_start, spec.centro_end)
    panel.set_freq(spec.freq)
    for l0, g in synth.genotype_chunks(spec, nind, dev):
        torch.cuda.synchronize()
        panel.set_genotypes_device(g.data_ptr(), g.shape[1], l0, g.shape[0])
    res = {}
    for W in sizes:
        panel.lod_feed(W, 0.001, 200000, W, copy=False)
    t0 = time.perf_counter()
    for W in sizes:
        panel.lod_feed(W, 0.001, 200000, W, copy=False)
    res["single_calls_ms"] = (time.perf_counter() - t0) * 1e3
    panel.lod_feed_multi(sizes, 0.001, 200000, copy=False)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        f, _ = panel.lod_feed_multi(sizes, 0.001, 200000, copy=False)
        ts.append((time.perf_counter() - t0) * 1e3)
    res["multi_call_ms"] = ts
    res["feed_gb"] = 8e-9 * sum(int(x.shape[0]) for x in f)
    print(json.dumps(res))
