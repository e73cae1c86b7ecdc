"""Run-to-run determinism of the round-4 feed / bits chains at a size where many workgroups share every CU: the feed of four
window sizes, the coverage counts from bits and the ROH segments, N launches each, compared with the first bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from garlic_amd import abi, synth
import bench
N = int(os.environ.get("REPS", 120))
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = abi.Context(0)
spec = synth.PanelSpec(1_000_000, seed=20260109, max_gap=200000)
nind = 2500
panel, _ = bench.load_panel(ctx, spec, nind, dev)
bad = 0
for W in (50, 100, 200, 300):
    first = None
    for rep in range(N // 4):
        f, _ = panel.lod_feed(W, 0.001, 200000, W)
        if first is None: first = f.copy()
        elif not np.array_equal(f.view(np.uint64), first.view(np.uint64)): bad += 1; print("feed differs", W, rep, flush=True)
_, _, t8 = panel.out_layout(8, nind)
cov = torch.empty(t8, dtype=torch.int16, device=dev)
first = None
for rep in range(N):
    cov.fill_(-1); torch.cuda.synchronize()
    panel.roh_coverage_fused_device(100, 0.001, 200000, 2.5, cov.data_ptr(), pitch_align=8)
    torch.cuda.synchronize()
    if first is None: first = cov.clone()
    elif not torch.equal(cov, first): bad += 1; print("coverage differs", rep, flush=True)
segs0 = None
for rep in range(N):
    s = panel.roh_segments(100, 0.001, 200000, 2.5, 0.25)
    if segs0 is None: segs0 = s
    elif not np.array_equal(s, segs0): bad += 1; print("segments differ", rep, flush=True)
print("launches", N + 2 * N, "differences", bad, "segments", segs0.shape[0], flush=True)
sys.exit(1 if bad else 0)
