"""PMC target: the graph kernels of the config-4 batch, 4 launches each (run under rocprofv3 --pmc; GPU box only)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
bench.graph_stress(torch.device('cuda:0'), reps=3, full=True, row_order_lines=False)
