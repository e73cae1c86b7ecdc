"""The final pass to the host, 10M SNPs x 1250 individuals, W = 100: ROH segments from the device (garlic_roh_segments)
against the coverage counts to the host (garlic_roh_coverage_fused, where = host: 25 GB over PCIe, and the per-individual
walk still to do there) and against the counts left on the device."""
import ctypes, os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from garlic_amd import abi, synth
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
nloci, nind, W = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, int(os.environ.get("NIND", 1250)), 100
spec = synth.PanelSpec(nloci, seed=20260101 + 3, max_gap=200000)
ctx = abi.Context(0)
panel, _ = bench.load_panel(ctx, spec, nind, dev, gq=len(sys.argv) > 2)
cap = 16_000_000
buf = np.empty((cap, 4), dtype=np.int32)
n = ctypes.c_int64()
for name, gl, wt in (("unweighted", 0, 0),) + ((("tgls", 1, 0),) if len(sys.argv) > 2 else ()):
    args = (panel.handle, W, 0.001, 200000, gl, wt, 7, 1e-9, 2.5, 0.25, ctypes.c_void_p(buf.ctypes.data), cap, ctypes.byref(n))
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        abi.check(abi.lib().garlic_roh_segments(*args))
        ts.append(round((time.perf_counter() - t0) * 1e3, 2))
    seg = buf[:n.value]
    print(name, "garlic_roh_segments call ms", ts, "segments", n.value, "bytes", n.value * 16,
          "mean SNPs per segment", round(float((seg[:, 3] - seg[:, 2] + 1).mean()), 1), flush=True)
    b8, p8, t8 = panel.out_layout(8, nind)
    cov = torch.empty(t8, dtype=torch.int16, device=dev)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        panel.roh_coverage_fused_device(W, 0.001, 200000, 2.5, cov.data_ptr(), pitch_align=8, use_gl=bool(gl))
        torch.cuda.synchronize()
        ts.append(round((time.perf_counter() - t0) * 1e3, 2))
    print(name, "garlic_roh_coverage_fused (counts stay on the device) call ms", ts, flush=True)
    del cov
    if nloci <= 10_000_000:
        t0 = time.perf_counter()
        host = panel.roh_coverage_fused(W, 0.001, 200000, 2.5, pitch_align=8, use_gl=bool(gl))
        print(name, "garlic_roh_coverage_fused (counts to the host:", sum(h.nbytes for h in host) >> 20, "MiB) call ms",
              round((time.perf_counter() - t0) * 1e3, 1), flush=True)
        del host
