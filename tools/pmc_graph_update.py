"""Refreshes FETCH_SIZE / WRITE_SIZE / traffic of the named kernels in profiles/r01_pmc_graph.json from two rocprofv3
--pmc passes over tools/one_graph.py (directories given on the command line: fetch pass, write pass)."""
import collections
import csv
import glob
import json
import sys

KERNELS = ("graphpool_fwd_v4_kernel", "graphpool_bwd_v5_kernel", "graphconv_bwd_dx_v4_kernel")


def mean_per_kernel(directory, counter):
    f = glob.glob(directory + "/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v[1:]) / max(len(v) - 1, 1), len(v)) for k, v in agg.items()}


fetch = mean_per_kernel(sys.argv[1], "FETCH_SIZE")
write = mean_per_kernel(sys.argv[2], "WRITE_SIZE")
path = "profiles/r01_pmc_graph.json"
d = json.load(open(path))
for name in KERNELS:
    fk = [k for k in fetch if k.startswith(name)]
    if not fk:
        print("no launches of", name)
        continue
    k = fk[0]
    ent = d["kernels"].setdefault(name, {})
    ent["FETCH_SIZE"], ent["WRITE_SIZE"] = round(fetch[k][0], 1), round(write[k][0], 1)
    ent["launches_seen"] = fetch[k][1]
    ent["traffic_bytes_per_launch"] = int(2 * fetch[k][0] * 1024 + write[k][0] * 1024)
    ent["note"] = "FETCH/WRITE re-measured with the XCD-aware block order (other counters: earlier row-order run)"
    print(name, ent["traffic_bytes_per_launch"])
json.dump(d, open(path, "w"), indent=1)
