"""Prints value / ms_per_step / roofline.frac / collation ratio of a bench JSON line: python tools/kt_bench_line.py <file>"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("with_collation", {}).get("ratio_to_resident"))
