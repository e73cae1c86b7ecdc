"""CPU: the bench line committed under profiles/ (printed by bench.py under rocprofv3 on an MI355X) carries
every field of the driver's contract, and its roofline object agrees with the rocprofv3 summary next to it
(tools/profile_round.sh -> tools/summarize_profiles.py, one lease)."""
import csv
import glob
import json
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
# every round that committed the profiled bench line (r02 on): the newest one is what DESIGN.md quotes
TAGS = sorted(re.match(r"(r\d+)_", os.path.basename(p)).group(1) for p in glob.glob(os.path.join(PROF, "r*_bench_under_rocprof.json")))
TAGS = [t for t in TAGS if t >= "r02"]


@pytest.mark.parametrize("TAG", TAGS)
def test_committed_bench_line_and_profile_agree(TAG):
    line = json.load(open(os.path.join(PROF, f"{TAG}_bench_under_rocprof.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert line["vs_baseline"] is None and line["dtype"] == "f64" and line["data"] == "synthetic"
    assert "workload" in line["config"] and "model" not in line["config"]
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert roof["traffic"] is None and "traffic_source" in roof          # not measured inside a bench run
    # achieved = algorithmic bytes per launch / the kernel's HIP-event time
    assert abs(roof["achieved"] - roof["algorithmic_bytes_per_launch"] / (roof["kernel_ms"] * 1e-3) / 1e9) < 1e-3 * roof["achieved"]
    # ... and the rocprofv3 kernel trace of the same run agrees with that time over the timed region
    # (the last `steps` dispatches: bench.py first tries several score buffers, see output_placement)
    rows = list(csv.DictReader(open(os.path.join(PROF, f"{TAG}_kernel_stats.csv"))))
    timed = [r for r in rows if "timed region only" in r["Name"]][0]
    assert int(timed["Calls"]) == line["steps"]
    assert abs(float(timed["AverageNs"]) * 1e-6 - roof["kernel_ms"]) < 0.03 * roof["kernel_ms"]
    pmc = json.load(open(os.path.join(PROF, f"{TAG}_pmc_traffic.json")))
    assert abs(pmc["hbm_bytes_per_launch"] - roof["algorithmic_bytes_per_launch"]) < 0.05 * roof["algorithmic_bytes_per_launch"]
    assert "candidates_kernel_ms" in line["output_placement"]


@pytest.mark.parametrize("TAG", TAGS)
def test_committed_tgls_profile(TAG):
    """the TGLS chain at the shard shape: every term row fetched once, >= 0.60 of the HBM peak on 16.25 B per window"""
    d = json.load(open(os.path.join(PROF, f"{TAG}_tgls_pmc_traffic.json")))
    assert 0.95 < d["fetch_over_terms_once"] < 1.10
    assert abs(d["hbm_bytes_per_launch"] - d["algorithmic_bytes_per_launch"]) < 0.05 * d["algorithmic_bytes_per_launch"]
    assert d["algorithmic_bytes_per_launch"] / (d["kernel_trace_avg_ns"] * 1e-9) / 8e12 >= 0.60


def test_gpus_n_without_a_launcher_starts_one_rank_per_gpu():
    """python bench.py --gpus 2 with WORLD_SIZE unset must not run one GPU and call it two: it starts itself under
    torch.distributed.run as a child process.  Here (no GPU) both ranks then stop at the missing device."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "small", "--no-cpu"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert "torch.distributed.run" in r.stderr and "--nproc-per-node=2" in r.stderr
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and "no HIP device visible" in r.stderr
    else:
        assert r.returncode != 0 or json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 2
