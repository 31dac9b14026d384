"""Counter sums per kernel name of a rocprofv3 --pmc run (all dispatches), sorted by the first counter:
python tools/pmc_by_kernel.py <dir> [top n]"""
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
names = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] not in names:
            names.append(r["Counter_Name"])
        cnt[(k, r["Counter_Name"])] += 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
print("%-70s %6s " % ("kernel", "calls") + " ".join("%16s" % c[-16:] for c in names))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][names[0]])[:n]:
    print("%-70s %6d " % (k, cnt[(k, names[0])]) + " ".join("%16.0f" % v[c] for c in names))
