"""Per-kernel and per-(kernel, grid) summary of a rocprofv3 --kernel-trace csv: python tools/kt_groups.py <dir> [steps]
Prints time per step when `steps` is given (timed + warm-up steps of the traced run)."""
import csv
import glob
import sys
from collections import defaultdict

files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = defaultdict(lambda: [0, 0.0])
for f in files:
    for r in csv.DictReader(open(f)):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        key = (r["Kernel_Name"][:90], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
        agg[key][0] += 1
        agg[key][1] += d
tot = sum(v[1] for v in agg.values())
print("total %.1f us (%.1f us / step)" % (tot, tot / steps))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 60]:
    print("%9.1f us/step %6.1f calls/step %8.1f us avg  %5.1f%%  grid %s,%s,%s  %s" % (v[1] / steps, v[0] / steps, v[1] / v[0], 100 * v[1] / tot, k[1], k[2], k[3], k[0]))
