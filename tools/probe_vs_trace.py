"""The probed FFN linear1 launch of bench.py, HIP events against the kernel trace of the SAME run:
    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 bench.py --no-cpu-baseline --no-graph-stress > <bench.json>
    python tools/probe_vs_trace.py <dir> <bench.json> <warmup> <steps> [out.json]
A step ends with the optimizer launch (sgd_kernel); in the first timed region the probed launch is the first gemm_x3w_kernel
of the step with the step's largest grid (N = 2048: FFN linear1 forward; the second one is the data gradient of linear2)."""
import csv, glob, json, sys

d, bench, warm, steps = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)))))
rows.sort()
step_list, cur = [], []
for s, e, name, grid in rows:
    cur.append((s, e, name, grid))
    if name.startswith("sgd_kernel"):
        step_list.append(cur)
        cur = []
dur = []
for st in step_list[warm:warm + steps]:
    x = [(e - s, g) for s, e, n, g in st if n.startswith("gemm_x3w_kernel")]
    if not x:
        continue
    gmax = max(g for _, g in x)
    dur.append(next(t for t, g in x if g == gmax) / 1e3)
line = json.loads(open(bench).read().strip().splitlines()[-1])
probe = line["roofline"]["avg_us"]
avg = sum(dur) / max(len(dur), 1)
doc = {"command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-graph-stress",
       "kernel": "gemm_x3w_kernel (FFN linear1 forward: the step's first launch of this kernel with the largest grid)",
       "steps_seen_in_trace": len(step_list), "timed_steps_used": len(dur),
       "trace_avg_us": round(avg, 1), "trace_min_us": round(min(dur), 1) if dur else None, "trace_max_us": round(max(dur), 1) if dur else None,
       "roofline_avg_us_hip_events_same_run": probe,
       "relative_difference": round(abs(probe - avg) / avg, 4) if avg else None,
       "note": "S varies over the 8 resident batches, hence the spread; both figures are from ONE run under rocprofv3"}
out = sys.argv[5] if len(sys.argv) > 5 else "profiles/r03_ffn1_probe_vs_trace.json"
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps(doc))
