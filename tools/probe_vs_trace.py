"""Cross-check of bench.py's live HIP-event timing of the probed FFN launch against a rocprofv3 kernel trace of the same
run: python tools/probe_vs_trace.py <kernel_trace.csv> <bench line json> <out json>.  The aligned K-contiguous/K-contiguous
kernel is launched `per_step` times per step in forward order; the positions are averaged separately over the timed steps."""
import csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("void gemm_f32_fast_kernel<true, true, false>")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
line = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
steps, warm = line["steps"], line["warmup"]
total_steps = 2 * (steps + warm) + steps + min(warm, 2)   # resident run, fp32-MFMA-only run, run with collation in the loop
per_step = len(rows) // total_steps
assert per_step * total_steps == len(rows), (len(rows), total_steps)
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0 for r in rows]
pos = []
for p in range(per_step):
    d = [dur[s * per_step + p] for s in range(warm, warm + steps)]          # the timed steps of the first (resident) run
    pos.append({"position": p, "avg_us": round(sum(d) / len(d), 1), "min_us": round(min(d), 1), "max_us": round(max(d), 1)})
probe = line["roofline"]["avg_us"]
best = pos[per_step - 2]                             # forward order ends with FFN linear1 (the probed launch), FFN linear2
out = {"command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-graph-stress",
       "kernel": "gemm_f32_fast_kernel<true, true, false>", "launches_per_step": per_step,
       "positions_in_forward_order": pos, "roofline_avg_us_hip_events_same_run": probe,
       "probed_position": best["position"], "trace_avg_us_of_that_position": best["avg_us"],
       "relative_difference": round(abs(best["avg_us"] - probe) / probe, 4)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
