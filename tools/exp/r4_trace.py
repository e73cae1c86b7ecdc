"""per-item time stamps (GARLIC_TRACE) of lod_bits_kernel: a single-run panel (500k x 1280) and the 10M x 1250 shard;
tools/exp/feed_trace.py prints pace (cycles per window) and the shader clock the items ran at"""
import ctypes, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from garlic_amd import abi, synth
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = abi.Context(0)
W = 100
out_dir = sys.argv[1]
def run(name, spec, nind):
    panel, _ = bench.load_panel(ctx, spec, nind, dev)
    cap = 16_000_000
    buf = np.empty((cap, 4), dtype=np.int32)
    n = ctypes.c_int64()
    args = (panel.handle, W, 0.001, 200000, 0, 0, 7, 1e-9, 2.5, 0.25, ctypes.c_void_p(buf.ctypes.data), cap, ctypes.byref(n))
    for _ in range(3):
        abi.check(abi.lib().garlic_roh_segments(*args))
    path = os.path.join(out_dir, "trace_%s.txt" % name)
    os.environ["GARLIC_TRACE"] = path
    abi.check(abi.lib().garlic_roh_segments(*args))
    del os.environ["GARLIC_TRACE"]
    panel.close()
    print("==", name, flush=True)
    subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "feed_trace.py"), path])
nloci = 500000
spec = synth.PanelSpec(nloci, seed=5, max_gap=200000, nchr=1)
spec.pos = (np.arange(1, nloci + 1, dtype=np.int64) * 100).astype(np.int32)
spec.gpos = spec.pos * 1e-6
spec.centro_start[:] = 0
spec.centro_end[:] = 0
run("single", spec, 1280)
run("shard", synth.PanelSpec(10_000_000, seed=20260101 + 3, max_gap=200000), 1250)
