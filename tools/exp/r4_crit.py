"""The pace of one chain wave of lod_bits_kernel / lod_feed_kernel and their throughput:
  (a) a panel that is ONE run of windows (one chromosome, no hole, no centromere), 1280 individuals: every work item is as
      long as the kernel, kernel time / windows = ns per window on the critical path;
  (b) 1M SNPs x 5000 individuals, 22 chromosomes (C3's shape, a fifth of its SNPs): the chip full of chains."""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from garlic_amd import abi, synth
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = abi.Context(0)
W = 100
res = {"lib": os.environ.get("VARIANT", "")}
nloci, nind = int(os.environ.get("CRIT_SNPS", 500000)), 1280
spec = synth.PanelSpec(nloci, seed=5, max_gap=200000, nchr=1)
spec.pos = (np.arange(1, nloci + 1, dtype=np.int64) * 100).astype(np.int32)
spec.gpos = spec.pos * 1e-6
spec.centro_start[:] = 0
spec.centro_end[:] = 0
panel, _ = bench.load_panel(ctx, spec, nind, dev)
cap = 4_000_000
buf = np.empty((cap, 4), dtype=np.int32)
n = ctypes.c_int64()
args = (panel.handle, W, 0.001, 200000, 0, 0, 7, 1e-9, 2.5, 0.25, ctypes.c_void_p(buf.ctypes.data), cap, ctypes.byref(n))
ks, ts = [], []
for _ in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    abi.check(abi.lib().garlic_roh_segments(*args))
    ts.append((time.perf_counter() - t0) * 1e3)
    ks.append(ctx.recent_kernel_ms(1)[0])       # the span memset + lod_bits_kernel
res["single_run_bits_kernel_ms"] = round(min(ks[1:]), 3)
res["single_run_bits_ns_per_window"] = round(min(ks[1:]) * 1e6 / nloci, 2)
res["single_run_segments_call_ms"] = round(min(ts[1:]), 3)
ks = []
for _ in range(4):
    panel.lod_feed(W, 0.001, 200000, W, copy=False)
    ks.append(panel.stats()["chain_kernel_ms"])
res["single_run_feed_kernel_ms"] = round(min(ks[1:]), 3)
res["single_run_feed_ns_per_window"] = round(min(ks[1:]) * 1e6 / nloci, 2)
panel.close()
nloci, nind = 1_000_000, 5000
spec = synth.PanelSpec(nloci, seed=20260101 + 2, max_gap=200000)
panel, _ = bench.load_panel(ctx, spec, nind, dev)
for Wf in (50, 100, 300):
    ks = []
    for _ in range(4):
        panel.lod_feed(Wf, 0.001, 200000, Wf, copy=False)
        ks.append(panel.stats()["chain_kernel_ms"])
    res["feed_1Mx5k_W%d_kernel_ms" % Wf] = round(min(ks[1:]), 3)
_, _, t8 = panel.out_layout(8, nind)
cov = torch.empty(t8, dtype=torch.int16, device=dev)
ts = []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    panel.roh_coverage_fused_device(W, 0.001, 200000, 2.5, cov.data_ptr(), pitch_align=8)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
res["fused_1Mx5k_call_ms"] = round(min(ts), 3)
print(json.dumps(res), flush=True)
