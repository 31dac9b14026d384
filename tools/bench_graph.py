"""Graph-kernel micro-benchmark on the config-4 batch (dev tool; GPU box only)."""
import sys, json
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
dev = torch.device('cuda:0')
r = bench.graph_stress(dev, reps=10, full=True) if 'full' in sys.argv else bench.graph_stress(dev, reps=10)
print(r['workload'])
for k in r['kernels']:
    print("%-44s %8.1f us  %7.1f GB/s  %5.1f%% of 8 TB/s" % (k['kernel'], k['us'], k['achieved'], 100 * k['frac']))
