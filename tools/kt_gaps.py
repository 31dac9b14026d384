"""Where the GPU idles inside a traced run: python tools/kt_gaps.py <dir> <warm_steps> <steps> [top]
Reads a rocprofv3 --kernel-trace csv of tools/trace_step.py, finds the step boundaries from the optimizer launch
(one per step), and for the timed steps prints span, busy time (union of the kernel intervals), idle time per step and
the largest gaps with the kernels on both sides."""
import csv
import glob
import sys
from collections import defaultdict

files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
warm, steps = int(sys.argv[2]), int(sys.argv[3])
top = int(sys.argv[4]) if len(sys.argv) > 4 else 25
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
rows.sort()
ends = [i for i, r in enumerate(rows) if r[2].startswith(("optim_kernel", "sgd_kernel", "adamax_kernel"))]
assert len(ends) >= warm + steps, "found %d optimizer launches" % len(ends)
first = ends[warm - 1] + 1 if warm else 0
last = ends[warm + steps - 1]
sel = rows[first:last + 1]
span = sel[-1][1] - sel[0][0]
busy, cur_end, gaps = 0, sel[0][0], []
for k, (s, e, n) in enumerate(sel):
    if s > cur_end:
        gaps.append((s - cur_end, sel[k - 1][2] if k else "", n))
        busy += e - s
        cur_end = e
    else:
        busy += max(0, e - cur_end)
        cur_end = max(cur_end, e)
print("steps %d: span %.1f us/step, busy %.1f us/step, idle %.1f us/step in %.1f gaps/step (%d launches/step)"
      % (steps, span / steps / 1e3, busy / steps / 1e3, (span - busy) / steps / 1e3, len(gaps) / steps, len(sel) / steps))
by = defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    by[(a, b)][0] += 1
    by[(a, b)][1] += g
for (a, b), (c, g) in sorted(by.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%8.1f us/step  %5.1f /step  %6.2f us avg   %s  ->  %s" % (g / steps / 1e3, c / steps, g / c / 1e3, a, b))
hist = defaultdict(int)
for g, _, _ in gaps:
    hist[min(int(g / 1e3), 20)] += g
print("idle by gap size (us : us/step):", {k: round(v / steps / 1e3, 1) for k, v in sorted(hist.items())})
