"""Mean counter values per kernel from rocprofv3 --pmc output directories (first dispatch of each kernel dropped):
python tools/pmc_dump.py <substring of the kernel name> <dir> [<dir> ...]  ->  JSON on stdout."""
import collections, csv, glob, json, sys

want = sys.argv[1]
out = collections.OrderedDict()
for d in sys.argv[2:]:
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if want in r["Kernel_Name"]:
                vals[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in vals.items():
        o = out.setdefault(k, collections.OrderedDict())
        for c, v in cs.items():
            o[c] = round(sum(v[1:]) / max(len(v) - 1, 1), 1)
            o.setdefault("dispatches", len(v))
json.dump(out, sys.stdout, indent=1)
print()
