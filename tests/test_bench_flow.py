"""The bench's N-rank flow, executed: ``python -m torch.distributed.run --nproc-per-node 1 ... bench.py`` with the RCCL
("nccl") backend on a one-rank group (JOVA_BENCH_FORCE_DIST=1), i.e. the exact command line the driver uses for
``--gpus N`` with N = 1 ranks: process-group setup on 127.0.0.1, ShardSampler, the hook-started two-piece gradient
all-reduce inside every step, barriers, the max-over-ranks timing, the JSON line.  So the driver's first multi-GPU run is
not the first execution of that code.  One rank only: a development box has one GPU and RCCL refuses
two ranks on one device."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_n_rank_flow_under_torchrun_one_rccl_rank(device):
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, JOVA_BENCH_FORCE_DIST="1", JOVA_BENCH_WATCHDOG="150", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)                                   # (an earlier test of this process may have set them)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
           "--no-cpu-baseline", "--no-graph-stress"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=170)
    assert res.returncode == 0, "bench under torchrun failed:\n%s\n%s" % (res.stdout[-2000:], res.stderr[-4000:])
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["steps"] == 4 and line["value"] > 0 and line["scaling"] == "weak"
    comm = line["comm"]
    assert comm["backend"] == "nccl" and comm["rccl_ranks"] == 1 and comm["overlapped"] is True
    assert comm["allreduce_bytes"] >= 4 * line["config"]["trainable_params"]
    assert line["rehearsal"].startswith("N-rank flow")
    assert line["work"]["flops_per_step"] > 1e11 and line["work"]["tokens_per_step"] > 1e4
