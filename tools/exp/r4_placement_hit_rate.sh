#!/bin/bash
# How often does a fresh process land in the fast placement of the score buffer?  N fresh processes of the headline bench
# (C2: 1M SNPs x 1000 individuals, W = 100; no CPU baseline, no `also` legs), each printing the kept / median / worst
# candidate, one plain hipMalloc buffer, and what garlic_panel_alloc_scores drew.  -> gpurun_out/r4/placement_hit_rate.txt
O=gpurun_out/r4; mkdir -p $O
N=${N:-8}
: > $O/placement_runs.jsonl
for i in $(seq 1 $N); do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu --also none 2>> $O/placement_runs.err >> $O/placement_runs.jsonl || break
done
python - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/r4/placement_runs.jsonl") if l.startswith("{")]
out = open("gpurun_out/r4/placement_hit_rate.txt", "w")
def p(*a):
    s = " ".join(str(x) for x in a); print(s); out.write(s + "\n")
p("# tools/exp/r4_placement_hit_rate.sh: %d fresh processes of `python bench.py --steps 20 --warmup 5 --no-cpu --also none` on one MI355X" % len(rows))
p("# (C2: 1M SNPs x 1000 individuals, W = 100, 8.25 GB per pass).  ms = HIP-event time of lod_chain_kernel; frac = of 8 TB/s.")
p("# kept: the buffer garlic_panel_alloc_scores kept (what the timed region writes); median / worst: over the candidates of the kept")
p("# buffer's round; plain: one hipMalloc buffer; drawn / rounds: candidates timed until one reached 0.74 of the HBM peak (1.394 ms) or")
p("# three rounds / 2 s were spent; timed: the kernel over the 20 timed steps")
p("%4s %9s %9s %9s %9s %6s %6s %8s %9s %7s" % ("run", "kept_ms", "median_ms", "worst_ms", "plain_ms", "drawn", "rounds", "reached", "timed_ms", "frac"))
hits = 0
for i, r in enumerate(rows):
    pl = r["output_placement"]; d = pl["drawn"]
    hits += bool(d["reached_target"])
    p("%4d %9.3f %9.3f %9.3f %9.3f %6d %6d %8s %9.3f %7.3f" % (i + 1, pl["kept_ms"], pl["median_ms"], pl["worst_ms"], pl["one_plain_hipmalloc_buffer_ms"],
      d["candidates_drawn"], d["rounds"], d["reached_target"], r["roofline"]["kernel_ms"], r["roofline"]["frac"]))
import statistics as st
p("# fast placement (kept <= 1.394 ms) found in %d of %d fresh processes; timed kernel: median %.3f ms = %.3f of HBM, worst %.3f = %.3f; plain hipMalloc median %.3f ms = %.3f"
  % (hits, len(rows), st.median(r["roofline"]["kernel_ms"] for r in rows), 8.25e9 / (st.median(r["roofline"]["kernel_ms"] for r in rows) * 1e-3) / 8e12,
     max(r["roofline"]["kernel_ms"] for r in rows), 8.25e9 / (max(r["roofline"]["kernel_ms"] for r in rows) * 1e-3) / 8e12,
     st.median(r["output_placement"]["one_plain_hipmalloc_buffer_ms"] for r in rows),
     8.25e9 / (st.median(r["output_placement"]["one_plain_hipmalloc_buffer_ms"] for r in rows) * 1e-3) / 8e12))
PY
