"""Print the graph_stress lines (config-4 batch) — lab script."""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench
r = bench.graph_stress(torch.device('cuda:0'), reps=20, full=True, row_order_lines=True)
for k in r["kernels"]:
    print("%-52s %8.1f us  frac %.3f (hbm %.3f)" % (k["kernel"], k["us"], k["frac"], k["frac_hbm"]))
