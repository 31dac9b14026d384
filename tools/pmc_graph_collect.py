"""Builds profiles/r02_pmc_graph.json from three rocprofv3 --pmc passes over tools/one_graph.py:
python tools/pmc_graph_collect.py <fetch_dir> <write_dir> <sq_dir> [out.json]
(passes: FETCH_SIZE | WRITE_SIZE | GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT)"""
import collections
import csv
import glob
import json
import sys


def per_kernel(directory):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(directory + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            out[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def mean(v):
    return sum(v[1:]) / max(len(v) - 1, 1) if len(v) > 1 else v[0]


fetch, write, sq = per_kernel(sys.argv[1]), per_kernel(sys.argv[2]), per_kernel(sys.argv[3])
dst = sys.argv[4] if len(sys.argv) > 4 else "profiles/r02_pmc_graph.json"
kernels = {}
for name in fetch:
    if not name.startswith(("graph", "gather2d", "nonsat")):
        continue
    ent = {"FETCH_SIZE": round(mean(fetch[name]["FETCH_SIZE"]), 1), "launches_seen": len(fetch[name]["FETCH_SIZE"])}
    if name in write and "WRITE_SIZE" in write[name]:
        ent["WRITE_SIZE"] = round(mean(write[name]["WRITE_SIZE"]), 1)
        ent["traffic_bytes_per_launch"] = int(2 * ent["FETCH_SIZE"] * 1024 + ent["WRITE_SIZE"] * 1024)
    for c, v in sq.get(name, {}).items():
        ent[c] = round(mean(v), 1)
    if "GRBM_GUI_ACTIVE" in ent and "SQ_VALU_MFMA_BUSY_CYCLES" in ent and ent["GRBM_GUI_ACTIVE"] > 0:
        # GRBM_GUI_ACTIVE sums the 8 XCDs; MFMA busy cycles sum the 1024 SIMDs
        ent["mfma_busy_fraction"] = round((ent["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (ent["GRBM_GUI_ACTIVE"] / 8.0), 3)
    kernels[name] = ent
doc = {
    "workload": "graph4k (config 4): 4096 molecules, 246506 atoms, 528580 directed edges; kernels as launched by "
                "bench.graph_stress (tools/one_graph.py)",
    "command": "rocprofv3 --pmc <counters> --output-format csv -- python3 tools/one_graph.py   (three separate passes: "
               "FETCH_SIZE | WRITE_SIZE | GRBM_GUI_ACTIVE SQ_*; 4 launches per kernel, first dropped, mean of the rest)",
    "units": "FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them; traffic_bytes_per_launch = 2 * FETCH_SIZE * 1024 + "
             "WRITE_SIZE * 1024 (MI355X_MICROARCH.md, HBM section: gfx950 FETCH_SIZE tallies 128-byte requests of "
             "16-byte-per-lane reads at 64 B -> doubled; WRITE_SIZE taken as is). Fetch counts Infinity-Cache hits too: it "
             "is L2-fabric traffic, an upper bound of HBM bytes.",
    "kernels": kernels,
}
json.dump(doc, open(dst, "w"), indent=1)
for k, e in kernels.items():
    print("%-60s traffic %s  mfma %s" % (k, e.get("traffic_bytes_per_launch"), e.get("mfma_busy_fraction")))
