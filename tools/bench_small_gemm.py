"""The small / thin products of a Davis step (view projections of the 194- and 22-wide views, the graph stack's Linear,
ProteinCNN's last layer, the FC head), forward, data gradient and weight gradient: microseconds per launch, `reps` launches
back to back between two HIP events.  python tools/bench_small_gemm.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from jova_amd.nn import functional as F_            # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
# (rows, in, out, what)
LAYERS = [(5760, 194, 512, "view projection, graph view"), (23168, 22, 512, "view projection, protein view"),
          (128, 1024, 512, "view projection, fingerprints"), (4168, 64, 194, "graph stack Linear 64->194"),
          (23168, 242, 22, "ProteinCNN last layer 242->22"), (128, 1536, 267, "head 1536->267"),
          (128, 267, 637, "head 267->637"), (128, 637, 1, "head 637->1")]


def timed(fn):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


tot = [0.0, 0.0, 0.0]
print("%-34s %7s %5s %5s | %8s %8s %8s   (us per launch)" % ("layer", "rows", "in", "out", "forward", "dX", "dW+db"))
for rows, k, n, what in LAYERS:
    x = torch.randn(rows, k, device=dev)
    w = torch.randn(n, k, device=dev)
    b = torch.randn(n, device=dev)
    y = torch.empty(rows, n, device=dev)
    dy = torch.randn(rows, n, device=dev)
    dx = torch.empty(rows, k, device=dev)
    dw, db = torch.zeros(n, k, device=dev), torch.zeros(n, device=dev)
    t_f = timed(lambda: F_.linear_fwd(x, w, b, y))
    t_x = timed(lambda: F_.linear_bwd_data(dy, w, dx))
    t_w = timed(lambda: F_.linear_bwd_weight(dy, x, dw, db))
    for i, t in enumerate((t_f, t_x, t_w)):
        tot[i] += t
    print("%-34s %7d %5d %5d | %8.1f %8.1f %8.1f" % (what, rows, k, n, t_f, t_x, t_w))
print("%-34s %19s | %8.1f %8.1f %8.1f   sum %.1f" % ("total", "", tot[0], tot[1], tot[2], sum(tot)))
