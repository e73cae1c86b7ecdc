"""The pace of ONE chain wave: a panel that is a single run of windows (one chromosome, no hole, no centromere), so that
every work item is as long as the kernel: kernel time / windows = cycles per window on the critical path.
Run under rocprofv3 --kernel-trace --stats for the per-kernel durations; the calls' wall times are printed too."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from garlic_amd import abi, synth
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
nind, W = 1280, 100
ctx = abi.Context(0)
for nloci in [int(x) for x in (sys.argv[1:] or ["250000", "500000"])]:
    spec = synth.PanelSpec(nloci, seed=5, max_gap=200000, nchr=1)
    spec.pos = (np.arange(1, nloci + 1, dtype=np.int64) * 100).astype(np.int32)
    spec.gpos = spec.pos * 1e-6
    spec.centro_start[:] = 0
    spec.centro_end[:] = 0
    panel, _ = bench.load_panel(ctx, spec, nind, dev, gq=True)
    base, pitch, total = panel.out_layout(32, nind)
    out = torch.empty(total, dtype=torch.float64, device=dev)
    res = {"nloci": nloci}
    for name, gl in (("lod", False), ("tgls", True)):
        for _ in range(3):
            panel.lod_windows_device(out.data_ptr(), W, 0.001, 200000, use_gl=gl)
        torch.cuda.synchronize()
        res[name + "_kernel_ms"] = round(panel.stats()["chain_kernel_ms"], 3)
    b8, p8, t8 = panel.out_layout(8, nind)
    cov8 = torch.empty(t8, dtype=torch.int16, device=dev)
    for name, gl in (("lod_fused", False), ("tgls_fused", True)):
        best = 1e9
        for rep in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            panel.roh_coverage_fused_device(W, 0.001, 200000, 2.5, cov8.data_ptr(), pitch_align=8, use_gl=gl)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3)
        res[name + "_call_ms"] = round(best, 3)
    print(res, flush=True)
    del out, cov8
    panel.close()
