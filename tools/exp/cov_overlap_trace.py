"""Per chain item of lod_bits_kernel: when it ended (ms after the kernel's first item began), with the counts in the
queue and without (GARLIC_TRACE dumps of garlic_roh_coverage_fused)."""
import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from garlic_amd import abi, synth
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
nloci, nind, W = 10_000_000, 1250, 100
spec = synth.PanelSpec(nloci, seed=20260101 + 3, max_gap=200000)
ctx = abi.Context(0)
panel, _ = bench.load_panel(ctx, spec, nind, dev)
b8, p8, t8 = panel.out_layout(8, nind)
cov = torch.zeros(t8, dtype=torch.int16, device=dev)
for name, env in (("overlap", "1"), ("own_launch", None)):
    if env: os.environ["GARLIC_COVERAGE_OVERLAP"] = env
    else: os.environ.pop("GARLIC_COVERAGE_OVERLAP", None)
    os.environ.pop("GARLIC_TRACE", None)
    for _ in range(2):
        panel.roh_coverage_fused_device(W, 0.001, 200000, 2.5, cov.data_ptr(), pitch_align=8)
    os.environ["GARLIC_TRACE"] = f"/tmp/trace_{name}.txt"
    panel.roh_coverage_fused_device(W, 0.001, 200000, 2.5, cov.data_ptr(), pitch_align=8)
    t = np.loadtxt(f"/tmp/trace_{name}.txt", dtype=np.int64)
    t0 = t[:, 2].min()
    beg, tb, end, ntiles = (t[:, 2] - t0) / 1e5, (t[:, 3] - t0) / 1e5, (t[:, 4] - t0) / 1e5, t[:, 8]
    per_win = (t[:, 4] - t[:, 3]) * 10.0 / (32.0 * ntiles)      # ns per window
    idx = [0, 5, 10, 20, 40, 80, 160, 240, 320]
    print(name, "items", len(t), "last end ms", round(end.max(), 2))
    for i in idx:
        if i < len(t):
            print(f"  item {i}: tiles {ntiles[i]}, begin {beg[i]:.2f} end {end[i]:.2f} ms, {per_win[i]:.1f} ns per window")
