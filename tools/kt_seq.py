"""Start-to-start and end-to-start times of consecutive launches of one kernel in a rocprofv3 --kernel-trace csv:
python tools/kt_seq.py <dir> <kernel substring>"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
prev = None
for k, (s, e, n) in enumerate(rows):
    if sys.argv[2] in n:
        if prev is not None and k == prev[2] + 1:
            print("dur %.1f us   gap after previous end %.1f us" % ((e - s) / 1e3, (s - prev[1]) / 1e3))
        prev = (s, e, k)
