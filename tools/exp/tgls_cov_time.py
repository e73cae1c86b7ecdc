import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from garlic_amd import abi, synth
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
nloci, nind, W = 2_000_000, 1280, 100
spec = synth.PanelSpec(nloci, seed=3, max_gap=200000)
ctx = abi.Context(0)
panel, _ = bench.load_panel(ctx, spec, nind, dev, gq=True)
base, pitch, total = panel.out_layout(32, nind)
out = torch.empty(total, dtype=torch.float64, device=dev)
panel.lod_windows_device(out.data_ptr(), W, 0.001, 200000, use_gl=True)
torch.cuda.synchronize()
print("tgls chain ms", panel.stats()["chain_kernel_ms"])
b8, p8, t8 = panel.out_layout(8, nind)
cov8 = torch.empty(t8, dtype=torch.int16, device=dev)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    panel.roh_coverage_fused_device(W, 0.001, 200000, 2.5, cov8.data_ptr(), pitch_align=8, use_gl=True)
    torch.cuda.synchronize()
    print("tgls fused coverage call ms", (time.perf_counter() - t0) * 1e3)
b1, p1, t1 = panel.out_layout(1, nind)
cov = torch.empty(t1, dtype=torch.int16, device=dev)
panel.lod_windows_device(out.data_ptr(), W, 0.001, 200000, use_gl=True)
panel.roh_coverage_device(out.data_ptr(), W, 2.5, cov.data_ptr())
ok = all(bool(torch.equal(cov[b1[c]: b1[c] + nind * p1[c]].view(nind, p1[c])[:, :spec.chr_nloci[c]], cov8[b8[c]: b8[c] + nind * p8[c]].view(nind, p8[c])[:, :spec.chr_nloci[c]])) for c in range(len(b1)))
print("equal:", ok)
