"""Which launches the in-stream zero fills of one training step belong to: for one step of a rocprofv3 --kernel-trace csv
(steps end with sgd_kernel), every zero_quads_kernel launch with its grid, duration and the next kernel on the stream:
python tools/kt_zero_context.py <dir> [step index]"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size", ""))))
rows.sort()
steps, cur = [], []
for r in rows:
    cur.append(r)
    if r[2].startswith("sgd_kernel"):
        steps.append(cur)
        cur = []
st = steps[int(sys.argv[2]) if len(sys.argv) > 2 else 5]
tot = 0.0
for i, (s, e, n, g) in enumerate(st):
    if n.startswith("zero_quads") or "FillFunctor" in n or "copyBuffer" in n or n.startswith("copy2d"):
        nxt = st[i + 1][2][:70] if i + 1 < len(st) else "-"
        tot += (e - s) / 1e3
        print("%-14s grid %8s %6.1f us  -> %s" % (n[:14], g, (e - s) / 1e3, nxt))
print("launches in the step: %d, GPU time of the step's kernels %.1f us, fills/copies %.1f us" % (len(st), sum(e - s for s, e, _, _ in st) / 1e3, tot))
