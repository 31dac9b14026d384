#!/usr/bin/env python3
"""Benchmark of the JOVA multi-view DTI training hot path on MI355X (contract: see the task statement / DESIGN.md).

  python bench.py --gpus N --steps K --warmup W          (N > 1: starts N rank processes itself, see launch_ranks)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A step = zero_grad -> forward -> MSELoss(out*w, y) -> backward -> (gradient all-reduce) -> SGD update of the tuned
Davis joint-view model (ecfp8-gconv__pcnn2d, B = 128 per GPU) on a synthetic Davis-shaped batch that is already
resident in HBM.  Rank 0 prints ONE JSON line: training pairs/s of the whole job, plus
  roofline      the dominant kernel of the step (fp32 MFMA GEMM), timed live with HIP events on its stream;
  graph_roofline  the graph kernels on the config-4 stress batch (4096 molecules), HBM roofline;
  cpu_baseline  the oracle (CPU restatement of the reference path) timed on this box's host cores, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from jova_amd import synth                                      # noqa: E402
from jova_amd.data.collate import BatchPrefetcher, collate_pairs   # noqa: E402
from jova_amd.feat.mol_graphs import ConvMol                    # noqa: E402
from jova_amd.nn import functional as F_                        # noqa: E402
from jova_amd.train import TUNED_DAVIS_HPARAMS, Trainer, count_parameters, create_integrated_net   # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 matrix (v_mfma_f32_32x32x2_f32) dense peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 matrix dense peak (v_mfma_f32_32x32x16_bf16)


# name -> (synthetic shape, protein views, description)
WORKLOADS = {
    "davis": ("davis", ["pcnn2d"], "Davis DTI joint-view model (ecfp8-gconv__pcnn2d, tuned hparams) fwd+bwd+SGD, "
                                   "B=128 per GPU, synthetic Davis-shaped batches resident in HBM"),
    "kiba": ("kiba", ["pcnn2d"], "KIBA-shaped DTI joint-view model (ecfp8-gconv__pcnn2d, tuned hparams) fwd+bwd+SGD, "
                                 "B=128 per GPU, synthetic KIBA-shaped batches (proteins to 376 segments) resident in HBM"),
    "metz_psc": ("metz", ["pcnn2d", "psc"], "Metz-shaped DTI joint-view model with the PSC view "
                                            "(ecfp8-gconv__pcnn2d-psc, tuned hparams) fwd+bwd+SGD, B=128 per GPU, "
                                            "synthetic batches resident in HBM"),
    # proj/dti/train_joint_attn.py:1022-1068 (default_hparams_bopt): B = 256, adamax, three wide head layers
    "davis_b256_default": ("davis", ["pcnn2d"], "Davis-shaped DTI joint-view model (ecfp8-gconv__pcnn2d, the reference's "
                                                "DEFAULT hparams: gconv 512, lin_dims 1033/1481/1800, adamax) fwd+bwd+"
                                                "Adamax, B=256 per GPU, synthetic batches resident in HBM"),
    # the long-protein stress variant (one protein of 1024 segments in every batch)
    "long1024": ("long1024", ["pcnn2d"], "long-protein stress variant (P_max = 1024 segments in every batch, tuned Davis "
                                         "hparams) fwd+bwd+SGD, B=128 per GPU, synthetic batches resident in HBM"),
}

DEFAULT_HPARAMS_OVERRIDE = {
    "lin_dims": [1033, 1481, 1800], "dprob": 0.2, "tr_batch_size": 256, "optimizer": "adamax",
    "optimizer__global__weight_decay": 0.0007, "optimizer__global__lr": 0.0004, "gconv": {"dim": 512},
    "prot": {"model_types": ["pcnn2d"], "vocab_size": 9048, "window": 11, "pcnn_num_layers": 2, "embedding_dim": 33,
             "psc_dim": 8421},
}

EPOCH_PAIRS = {"davis": 25459, "kiba": 118000, "metz": 28245, "long1024": 25459}     # training pairs of one epoch


def build_batches(pool, conv_mols, hp, device, n_batches, prot2vec, shape, world, rank):
    """The first ``n_batches`` batches of this rank's shard of a fixed synthetic epoch (ShardSampler: one seeded
    permutation, disjoint shards, tail dropped), collated and resident in HBM."""
    from jova_amd.data.sampler import ShardSampler
    mi, pi, y, w = pool.epoch_pairs(EPOCH_PAIRS[shape], seed=1000)
    sampler = ShardSampler(len(mi), world, rank, hp["tr_batch_size"], seed=1)
    out = []
    shard = sampler.indices()
    if shape == "long1024":                         # every batch holds the 1024-segment protein (pool protein 0)
        pi = pi.copy()
        pi[shard[:, 0]] = 0
    for idx in shard[:n_batches]:
        out.append(collate_pairs(pool, conv_mols, mi[idx], pi[idx], y[idx], w[idx], device, prot2vec,
                                 with_psc="psc" in hp["prot"]["model_types"]))

    def make_batch(k, stream):                      # batch k (mod n_batches) of this rank's shard, collated now
        idx = shard[(k % n_batches) % len(shard)]
        return collate_pairs(pool, conv_mols, mi[idx], pi[idx], y[idx], w[idx], device, prot2vec,
                             with_psc="psc" in hp["prot"]["model_types"], stream=stream)
    return out, sampler, make_batch


def step_flops(hp, batch):
    """Algorithmic FLOPs (2 per multiply-add) of one training step on `batch` and its token count S * B.

    Forward products only are enumerated; backward = 2 x forward (one data and one weight gradient per product; the few
    input gradients the step does not need -- fingerprints, descriptors -- are counted all the same: < 0.1 %).  Padded
    tokens count: the reference's attention has no padding mask (models.py:952), they are real work.
      JointAttention (models.py:939-973), T = S B tokens: view projections 2 B L sum_v n_v d_v, QKV 6 T L^2, scores and PV
      4 B S^2 L, out-proj 2 T L^2, FFN 4 T L I;  ProteinCNN2D (:760-771) 2 P_max B (layers-1) (11e)^2 + 2 P_max B 11e e;
      GraphConv x2 4 N F_in F_out (layers.py:312-339) + Linear 2 N 64 g;  FC head 2 B sum p_in p_out."""
    B = batch.batch_size
    L, I = hp["latent_dim"], 2048
    g, e, win = hp["gconv"]["dim"], hp["prot"]["embedding_dim"], hp["prot"]["window"]
    n_atoms = batch.glist[0].shape[0]
    a_max = max(batch.glist[3])
    p_max = int(batch.prot_input[2])
    views = [(1, hp["ecfp8"]["dim"]), (a_max, g), (p_max, e)] + ([(1, hp["prot"]["psc_dim"])] if batch.psc is not None else [])
    S = sum(n for n, _ in views)
    T = S * B
    ja = 2.0 * B * L * sum(n * d for n, d in views) + hp["attn_layers"] * (8.0 * T * L * L + 4.0 * T * L * I + 4.0 * B * S * S * L)
    d = win * e
    pcnn = 2.0 * p_max * B * ((hp["prot"]["pcnn_num_layers"] - 1) * d * d + d * e)
    gconv = 4.0 * n_atoms * (75 * 64 + 64 * 64) + 2.0 * n_atoms * 64 * g
    head, p_in = 0.0, len(views) * L
    for p_out in list(hp["lin_dims"]) + [hp["output_dim"]]:
        head += 2.0 * B * p_in * p_out
        p_in = p_out
    return 3.0 * (ja + pcnn + gconv + head), T


def launch_ranks(n):
    """``python bench.py --gpus N`` from a bare shell: this parent never touches the GPU; it starts N rank processes
    (torch.distributed.run, one per GPU, rendezvous on 127.0.0.1) that run this same file, and exits with their code.
    JOVA_BENCH_SHARE_GPU=1 (development boxes with one GPU): the ranks share GPU 0 and reduce gradients over gloo
    through pinned host memory -- a rehearsal of the N-rank flow, flagged as such in the JSON, not a measurement."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()                       # (counting devices does not initialise the GPU)
    if n_dev < n and not os.environ.get("JOVA_BENCH_SHARE_GPU"):
        raise SystemExit("--gpus %d but %d GPU(s) visible: one process per GPU (JOVA_BENCH_SHARE_GPU=1 rehearses the "
                         "N-rank flow on one GPU over gloo)" % (n, n_dev))
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def measured_traffic(names=("r02_pmc_ffn1.json", "r01_pmc_ffn1.json")):
    """HBM-side bytes per launch of the probed GEMM from the committed PMC passes (profiles/r03_pmc_x3w_ffn1.json for the
    bf16-pieces kernel, r02_pmc_ffn1.json for the fp32 MFMA kernel: FETCH_SIZE / WRITE_SIZE collected in separate rocprofv3
    --pmc runs of tools/one_gemm.py, gfx950 corrections applied as MI355X_MICROARCH.md prescribes).  It is quoted for the
    S = 279 launch those passes ran, not re-measured here."""
    for name in names:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                return int(json.load(f)["traffic_bytes_per_launch"])
        except (OSError, ValueError, KeyError):
            continue
    return None


def cpu_baseline(pool, hp, batch_ids, state_dict, n_steps, threads):
    """Oracle (fp32 CPU restatement of the reference path): fwd + bwd + SGD on the same kind of batch."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import mol_graphs as og
    from oracle import nn_ref as R
    torch.set_num_threads(threads)
    mi, pi, y, w = batch_ids
    t_col = time.perf_counter()
    b = og.agglomerate([og.SortedMol(*pool.mols[i]) for i in mi])
    gl = og.gconv_input_list(b)
    glist = [torch.from_numpy(gl[0]), torch.from_numpy(gl[1]), torch.from_numpy(gl[2]), gl[3]] + \
            [torch.from_numpy(a) for a in gl[4:]]
    idx = og.embedding_indices([pool.prot_names[i] for i in pi], pool.protein_profile, fill_val=pool.vocab)
    t_col = time.perf_counter() - t_col
    batch = dict(ecfp=torch.from_numpy(pool.ecfp[mi]), glist=glist, prot_idx=torch.from_numpy(idx),
                 psc=torch.from_numpy(pool.psc[pi]) if "psc" in hp["prot"]["model_types"] else None)
    p = {k: (v.detach().cpu().clone().requires_grad_(v.is_floating_point() and "running" not in k))
         for k, v in state_dict.items()}
    yt, wt = torch.from_numpy(y).float(), torch.from_numpy(w).float()
    lr, wd = hp["optimizer__global__lr"], hp["optimizer__global__weight_decay"]
    times = []
    for step in range(n_steps + 1):
        t0 = time.perf_counter()
        out, _ = R.dti_forward(p, batch, hp["attn_heads"], hp["prot"]["pcnn_num_layers"], len(hp["lin_dims"]))
        loss = R.mse_loss(out, wt, yt)
        loss.backward()
        with torch.no_grad():
            for v in p.values():
                if v.requires_grad:
                    if v.grad is not None:
                        v -= lr * (v.grad + wd * v)
                        v.grad = None
        if step > 0:
            times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    cpu_model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return dict(value=round(hp["tr_batch_size"] / med, 2), unit="pairs/s", cores=threads, kind="port", cpu=cpu_model,
                host_cpus=os.cpu_count(),
                sample="%d timed fwd+bwd+SGD steps (after 1 warm-up) of the fp32 oracle on one Davis-shaped batch of %d "
                       "pairs, median %.2f s/step; oracle collate %.1f ms/batch not included"
                       % (n_steps, hp["tr_batch_size"], med, t_col * 1e3))


def graph_stress(device, reps=20, full=True, row_order_lines=True):
    """Config 4: 4096 molecules (~246 k atoms, degree <= 10): graph kernels against the HBM roofline.

    Every kernel is launched through the C ABI (jova_amd._lib.call) on pre-allocated buffers, `reps` launches back to
    back between two HIP events on torch's current stream (the stream the library launches on), so the figure is the
    kernel's own duration, not the Python / autograd dispatch around it (which is longer than these kernels)."""
    from jova_amd.data.collate import process_gconv_view_data
    from jova_amd.nn.layers import GraphConvLayer
    from jova_amd._lib import P, call, lib, ptr_array, stream
    import ctypes
    pool = synth.SyntheticPool("graph4k", seed=1)
    mols = [ConvMol(f, a, smiles=str(i)) for i, (f, a) in enumerate(pool.mols)]
    glist = process_gconv_view_data(mols, device)
    topo = F_.topology_of(glist)
    n = glist[0].shape[0]
    e = sum(int(a.numel()) for a in glist[4:])
    conv1, conv2 = GraphConvLayer(75, 64).to(device), GraphConvLayer(64, 64).to(device)
    x1 = topo.atoms_f32(glist[0])                       # fp32 copy of the int64 features, once per batch
    x2 = torch.randn(n, 64, device=device).relu_()
    res = []

    pmc = {}
    for name in ("r01_pmc_graph.json", "r02_pmc_graph.json", "r03_pmc_graph.json"):       # later rounds override per kernel
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                pmc.update(json.load(f)["kernels"])
        except (OSError, ValueError, KeyError):
            pass

    def traffic_of(*kernels):
        """L2-fabric bytes per launch from the committed PMC passes (profiles/r0N_pmc_graph.json), summed over the
        kernels of one entry; None when a kernel is missing there."""
        try:
            return int(sum(pmc[k]["traffic_bytes_per_launch"] for k in kernels))
        except KeyError:
            return None

    def timed(name, fn, nbytes, launches=1, kernels=(), flops=0):
        """`flops` (the matrix products of the entry, 2 per multiply-add) adds the fp32-MFMA fraction; the entry's
        `bound` is the roofline that binds (the larger of the two lower bounds on its duration) and `frac` the
        fraction of THAT one; frac_hbm / frac_mfma are both printed."""
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        gbs = nbytes / (ms * 1e-3) / 1e9
        tfl = flops / (ms * 1e-3) / 1e12
        frac_hbm, frac_mfma = gbs / HBM_PEAK_GBS, tfl / MFMA_F32_PEAK_TFLOPS
        mfma_bound = frac_mfma > frac_hbm
        res.append(dict(kernel=name, bound="mfma" if mfma_bound else "hbm",
                        achieved=round(tfl if mfma_bound else gbs, 1),
                        peak=MFMA_F32_PEAK_TFLOPS if mfma_bound else HBM_PEAK_GBS, unit="TFLOP/s" if mfma_bound else "GB/s",
                        frac=round(max(frac_hbm, frac_mfma), 4), frac_hbm=round(frac_hbm, 4),
                        frac_mfma=round(frac_mfma, 4) if flops else None, us=round(ms * 1e3, 1),
                        algorithmic_bytes=int(nbytes), flops=int(flops) if flops else None,
                        launches=launches, traffic=traffic_of(*kernels) if kernels else None))

    def wb(layer):
        w = [l.weight for l in layer.linear_layers]
        bias = [l.bias for l in layer.linear_layers]
        return w, bias, ptr_array(w), ptr_array(bias)

    st = stream()
    ds, adj = P(topo.deg_slice), topo.adj_ptrs
    dsh = ctypes.c_void_p(topo.deg_slice_host.ctypes.data)
    w1, b1, pw1, pb1 = wb(conv1)
    w2, b2, pw2, pb2 = wb(conv2)
    out = torch.empty(n, 64, device=device)
    arg = torch.empty(n, 64, dtype=torch.uint8, device=device)
    ws1 = torch.empty(int(lib().jova_graphconv_workspace_floats(75, 64)), device=device)
    ws2 = torch.empty(int(lib().jova_graphconv_workspace_floats(64, 64)), device=device)
    wbytes = 4 * 21 * 65 * 64
    # layer 1: the batch's int64 atom features are converted to fp32 once per batch (cached on the topology), so the
    # per-step kernel reads 4-byte features: algorithmic bytes as SURVEY.md section 8d
    timed("graphconv_fwd_l1(75->64)", lambda: call("jova_graphconv_fwd", P(x1), 0, x1.shape[1], n, 75, 64, ds, adj, pw1,
                                                   pb1, 1, P(out), None, P(ws1), dsh, st),
          4 * (n * 75 + n * 64 + e) + 4 * 21 * 76 * 64, launches=1, kernels=("graphconv_fwd_wave_kernel<10, 3, false>",),
          flops=4 * n * 75 * 64)
    timed("graphconv_fwd_l2(64->64)", lambda: call("jova_graphconv_fwd", P(x2), 0, 64, n, 64, 64, ds, adj, pw2, pb2, 1,
                                                   P(out), None, P(ws2), dsh, st),
          4 * (n * 64 + n * 64 + e) + wbytes, launches=1, kernels=("graphconv_fwd_wave_kernel<8, 4, false>",),
          flops=4 * n * 64 * 64)
    order = P(topo.block_order)
    table = P(topo.row_table)
    timed("graphpool_fwd(64)", lambda: call("jova_graphpool_fwd", P(x2), n, 64, ds, adj, P(out), P(arg), order, table, st),
          4 * (2 * n * 64 + e) + n * 64, kernels=("graphpool_fwd_t_kernel",))
    if row_order_lines:                                    # the same work without the row table / the block order
        timed("graphpool_fwd(64), no row table",
              lambda: call("jova_graphpool_fwd", P(x2), n, 64, ds, adj, P(out), P(arg), order, None, st),
              4 * (2 * n * 64 + e) + n * 64, kernels=("graphpool_fwd_v4_kernel",))
        timed("graphpool_fwd(64), blocks in row order",
              lambda: call("jova_graphpool_fwd", P(x2), n, 64, ds, adj, P(out), P(arg), None, None, st),
              4 * (2 * n * 64 + e) + n * 64, kernels=())
    if full:
        # backward kernels accumulate into gradient buffers (as in the trainer).  In the model the BatchNorm behind
        # each convolution applies the backward of the convolution's ReLU (it reads that output anyway), so the
        # convolution's backward receives dZ: y_or_null = NULL here.
        dw1 = [torch.zeros_like(t) for t in w1]; db1 = [torch.zeros_like(t) for t in b1]
        dw2 = [torch.zeros_like(t) for t in w2]; db2 = [torch.zeros_like(t) for t in b2]
        pdw1, pdb1, pdw2, pdb2 = ptr_array(dw1), ptr_array(db1), ptr_array(dw2), ptr_array(db2)
        dz = torch.randn(n, 64, device=device)
        dx, gbuf = torch.empty(n, 64, device=device), torch.empty(n, 64, device=device)
        sym = int(topo.symmetric)
        dws = torch.empty(int(lib().jova_graphconv_bwd_scratch_floats()), device=device)
        timed("graphconv_bwd_l1(dW only)",
              lambda: call("jova_graphconv_bwd", P(x1), 0, x1.shape[1], n, 75, 64, ds, adj, pw1, P(dz), None, pdw1, pdb1,
                           None, None, P(topo.sym_flag), dsh, P(dws), None, st),
              4 * (n * 64 + n * 75 + e) + 2 * 4 * 21 * 76 * 64,
              kernels=("graphconv_bwd_split_kernel<10>", "graphconv_dw_fold_kernel"), launches=3,
              flops=4 * n * 75 * 64)
        # 64 -> 64: graphconv_bwd_wave_kernel (dW_self, dW_rel, G = dZ W_rel) + graphconv_bwd_dx_wave_kernel
        # (dx = dZ W_self + gathered G) + the scatter kernel that returns at once on a bond graph
        timed("graphconv_bwd_l2(dW + dX, 3 launches)",
              lambda: call("jova_graphconv_bwd", P(x2), 0, 64, n, 64, 64, ds, adj, pw2, P(dz), None, pdw2, pdb2, P(dx),
                           P(gbuf), P(topo.sym_flag), dsh, None, order, st),
              4 * (n * 64 + 3 * n * 64 + e) + 2 * wbytes, launches=3, flops=8 * n * 64 * 64,
              kernels=("graphconv_bwd_wave_kernel<8, true, 2>", "graphconv_bwd_dx_wave_kernel"))
        timed("graphpool_bwd(64)", lambda: call("jova_graphpool_bwd", P(dz), P(arg), n, 64, ds, adj, P(dx),
                                                P(topo.sym_flag), sym, P(topo.rev_pos), order, table, st),
              4 * (2 * n * 64 + e) + n * 64, kernels=("graphpool_bwd_t_kernel",))
        if row_order_lines:
            timed("graphpool_bwd(64), no row table",
                  lambda: call("jova_graphpool_bwd", P(dz), P(arg), n, 64, ds, adj, P(dx), P(topo.sym_flag), sym,
                               P(topo.rev_pos), order, None, st),
                  4 * (2 * n * 64 + e) + n * 64, kernels=("graphpool_bwd_v5_kernel",))
            timed("graphpool_bwd(64), blocks in row order",
                  lambda: call("jova_graphpool_bwd", P(dz), P(arg), n, 64, ds, adj, P(dx), P(topo.sym_flag), sym,
                               P(topo.rev_pos), None, None, st),
                  4 * (2 * n * 64 + e) + n * 64, kernels=())
    if full:
        # GraphGather2D (contiguous chunks -> zero-padded [A_max, B, g]) and the nonsat activation on it, g = 128
        g = 128
        lst = topo.n_atoms_lst
        a_max, bsz = max(lst), len(lst)
        xg = torch.randn(n, g, device=device)
        og = torch.empty(a_max, bsz, g, device=device)
        dxg = torch.zeros(n, g, device=device)
        timed("gather2d_fwd(g=128)", lambda: call("jova_gather2d_fwd", P(xg), P(topo.mol_off), a_max, bsz, g, P(og), st),
              4 * (n * g + a_max * bsz * g), kernels=("gather2d_fwd_v4_kernel",))
        timed("gather2d_bwd(g=128)", lambda: call("jova_gather2d_bwd", P(og), P(topo.mol_off), a_max, bsz, g, P(dxg), st),
              4 * (2 * n * g), kernels=("gather2d_bwd_v4_kernel",))
        ne = og.numel()
        on, df = torch.empty_like(og), torch.empty_like(og)
        nev = torch.empty(3, dtype=torch.int32, device=device)
        wsn = torch.empty(int(lib().jova_nonsat_workspace_bytes(ne)), dtype=torch.uint8, device=device)
        call("jova_nonsat_fwd", P(og), ne, 1e-4, 100, P(on), P(df), P(nev), P(wsn), st)
        evals = int(nev[0].item())
        # per evaluation: read x and y, write y' (12 B / element); final pass: read 2, write 2 (16 B / element)
        timed("nonsat_fwd(%d evaluations, one persistent launch)" % evals,
              lambda: call("jova_nonsat_fwd", P(og), ne, 1e-4, 100, P(on), P(df), P(nev), P(wsn), st),
              ne * (12 * evals + 16), launches=3, kernels=("nonsat_persistent_kernel<4>",))
    return dict(workload="graph4k: 4096 molecules, %d atoms, %d directed edges" % (n, e), kernels=res,
                note="kernel durations: reps launches through the C ABI between two HIP events; conv backward takes dZ "
                     "(ReLU mask applied by the BatchNorm backward, as in the model); frac_mfma against the dense fp32 "
                     "matrix peak (157.3 TFLOP/s), frac_hbm against 8 TB/s, frac = the binding one")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph-stress", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="davis",
                    help="davis = BASELINE.json configs[1] (the bench line); kiba / metz_psc = configs[3] / [5] shapes")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    # a stuck run dumps every thread's stack and exits instead of hanging: always on for a multi-rank run (a rank that
    # never reaches a collective would otherwise hold the others forever), on request for one rank
    watchdog = os.environ.get("JOVA_BENCH_WATCHDOG") or ("900" if world > 1 else "")
    if watchdog and float(watchdog) > 0:
        import faulthandler
        faulthandler.dump_traceback_later(float(watchdog), exit=True)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but the launcher started %d ranks" % (args.gpus, world))
    n_dev = torch.cuda.device_count()
    shared = bool(os.environ.get("JOVA_BENCH_SHARE_GPU")) and world > 1
    if local_rank >= n_dev and not shared:
        raise SystemExit("rank %d has no GPU of its own (%d visible): one process per GPU" % (local_rank, n_dev))
    dev_id = 0 if shared else local_rank
    torch.cuda.set_device(dev_id)
    device = torch.device("cuda", dev_id)
    # JOVA_BENCH_FORCE_DIST=1 (under torch.distributed.run --nproc-per-node 1): the N-rank flow -- RCCL group, overlapped
    # gradient all-reduce, barriers, max over ranks -- on a one-rank group; a rehearsal for one-GPU boxes
    dist_on = world > 1 or bool(os.environ.get("JOVA_BENCH_FORCE_DIST"))
    pg = None
    if dist_on:
        if shared:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=device)
        pg = torch.distributed.group.WORLD

    hp = dict(TUNED_DAVIS_HPARAMS)
    if args.workload == "davis_b256_default":
        hp.update(DEFAULT_HPARAMS_OVERRIDE)
    shape, pviews, workload_text = WORKLOADS[args.workload]
    hp["prot"] = dict(hp["prot"], model_types=list(pviews))
    pool = synth.SyntheticPool(shape, seed=1)
    conv_mols = [ConvMol(f, a, smiles=str(i)) for i, (f, a) in enumerate(pool.mols)]
    torch.manual_seed(1)                                   # identical initial weights on every rank
    model = create_integrated_net(hp, pool.protein_profile).to(device).train()
    sd0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    trainer = Trainer(model, hp, world_size=world, process_group=pg, force_sync=dist_on, rank=rank)
    prot2vec = model[0].models[2][0]
    # eight batches of this rank's shard, resident in HBM, cycled (the set BENCH_r01 was measured on); the with-collation
    # run below collates the SAME eight batches, in the same order, inside its timed region, so the two rates differ by
    # the cost of collation only (batches differ in padded length: over 35 distinct batches of the epoch the same code
    # measures 17.9 k pairs/s resident and 0.997 of it with collation)
    n_res = 8
    batches, sampler, make_batch = build_batches(pool, conv_mols, hp, device, n_res, prot2vec, shape, world, rank)

    def sync_all():
        torch.cuda.synchronize()
        if dist_on:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        trainer.step(batches[i % len(batches)])
    F_.PROBE["ffn1_fwd"] = []
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss, _ = trainer.step(batches[i % len(batches)])
    sync_all()
    elapsed = time.perf_counter() - t0
    probe = F_.PROBE.pop("ffn1_fwd")
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(loss.item())
    if not np.isfinite(final_loss):
        raise SystemExit("non-finite loss")
    trainer.check_health()
    # the same K steps with the backward products kept on the fp32 MFMA kernel (jova_gemm_set_bf16x3(0)): what the
    # bf16-pieces backward GEMM is worth, and the number to take if that form is not accepted as fp32 arithmetic
    from jova_amd._lib import lib as _jova_lib
    x3_was = _jova_lib().jova_gemm_get_bf16x3()
    _jova_lib().jova_gemm_set_bf16x3(0)
    for i in range(min(args.warmup, 2)):
        trainer.step(batches[i % len(batches)])
    F_.PROBE["ffn1_fwd"] = []
    sync_all()
    t3 = time.perf_counter()
    for i in range(args.steps):
        trainer.step(batches[i % len(batches)])
    sync_all()
    elapsed_f = time.perf_counter() - t3
    probe_f = F_.PROBE.pop("ffn1_fwd")
    _jova_lib().jova_gemm_set_bf16x3(x3_was)
    if dist_on:
        tf_ = torch.tensor([elapsed_f], dtype=torch.float64, device=device)
        if shared:
            trainer.reduce(tf_)
            elapsed_f = float(tf_.item()) / world
        else:
            torch.distributed.all_reduce(tf_, op=torch.distributed.ReduceOp.MAX)
            elapsed_f = float(tf_.item())
    # the same K steps with collation INSIDE the timed region: every batch is merged (native jova_collate_gconv), packed
    # into pinned memory and uploaded by the prefetch thread while the previous step runs
    n_cyc = len(batches)
    first = (n_cyc - args.warmup % n_cyc) % n_cyc        # so that the timed region starts at batch 0, as above
    pf = iter(BatchPrefetcher(lambda k, st: make_batch(first + k, st), args.warmup + args.steps, device, depth=3))
    for _ in range(args.warmup):                         # untimed: pinned staging buffers, copy stream, thread start
        trainer.step(next(pf))
    sync_all()
    t2 = time.perf_counter()
    for b in pf:
        loss2, _ = trainer.step(b)
    sync_all()
    elapsed_c = time.perf_counter() - t2
    if dist_on:
        tc_ = torch.tensor([elapsed_c], dtype=torch.float64, device=device)
        if shared:
            trainer.reduce(tc_)
            elapsed_c = float(tc_.item()) / world
        else:
            torch.distributed.all_reduce(tc_, op=torch.distributed.ReduceOp.MAX)
            elapsed_c = float(tc_.item())
    comm = None
    if dist_on:
        # the step's one exchange, alone: all-reduce of the flat gradient buffer, 10 back to back between barriers
        sync_all()
        t1 = time.perf_counter()
        for _ in range(10):
            trainer.reduce(trainer.flat.grad)
        sync_all()
        t_ar = torch.tensor([(time.perf_counter() - t1) / 10], dtype=torch.float64, device=device)
        trainer.reduce(t_ar) if shared else torch.distributed.all_reduce(t_ar, op=torch.distributed.ReduceOp.MAX)
        t_ar = float(t_ar.item()) / (world if shared else 1)
        nbytes = trainer.flat.grad.numel() * 4
        comm = dict(backend=torch.distributed.get_backend(pg), rccl_ranks=world if not shared else 0,
                    allreduce_bytes=nbytes, allreduce_us=round(t_ar * 1e6, 1),
                    bus_gbs=round(2.0 * (world - 1) / max(world, 1) * nbytes / t_ar / 1e9, 2),
                    overlapped=trainer.sync is not None)

    bsz = hp["tr_batch_size"]
    # work of the timed steps, so that pairs/s can be compared across batch sets and rounds
    per_batch = [step_flops(hp, b) for b in batches]
    timed = [per_batch[i % len(batches)] for i in range(args.steps)]
    flops_per_step = float(np.mean([f for f, _ in timed]))
    tokens_per_step = float(np.mean([t for _, t in timed]))
    pairs_per_s = args.steps * bsz * world / elapsed
    # dominant kernel: the aligned GEMM; probe = the FFN linear1 launch (largest single launch of the step), timed with
    # HIP events on its stream in both timed regions.  On the fp32 matrix pipe (fp32_mfma_only run) the roofline is the
    # fp32 MFMA peak over the algorithmic 2 M N K.  In the default run the same product executes as SIX bf16 MFMA products
    # of three-way split operands: its roofline is the bf16 matrix peak over the 6 x 2 M N K the pipe really executes
    # (`algorithmic_tflops` = 2 M N K / time, the fp32-equivalent rate, rides along).
    def probe_stats(pr):
        fl = sum(f for _, _, f in pr)
        t_ms = sum(a.elapsed_time(b) for a, b, _ in pr)
        return fl, t_ms, (fl / (t_ms * 1e-3) / 1e12 if t_ms > 0 else 0.0)
    flops, ms, tfl = probe_stats(probe)
    flops_f, ms_f, tfl_f = probe_stats(probe_f)
    shape_txt = "(JointAttention FFN linear1, [S*B,512]x[512,2048]+bias+ReLU)"
    traffic_note = ("NOT re-measured in this run: bytes per launch of the S=279 launch that profiles/r02_pmc_ffn1.json "
                    "profiled on the fp32 MFMA kernel (2*FETCH_SIZE + WRITE_SIZE, separate --pmc passes; algorithmic bytes "
                    "of that launch 369.9e6); this run's launches average S*B = %d tokens" % int(tokens_per_step))
    roofline_f32 = dict(kernel="gemm_f32_fast_kernel<K-contig,K-contig> " + shape_txt, bound="mfma", achieved=round(tfl_f, 2),
                        peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s", frac=round(tfl_f / MFMA_F32_PEAK_TFLOPS, 4),
                        traffic=measured_traffic(), traffic_note=traffic_note, launches=len(probe_f),
                        avg_us=round(ms_f * 1e3 / max(len(probe_f), 1), 1),
                        note="the probed launch in the fp32_mfma_only run: fp32 MFMA (v_mfma_f32_32x32x2_f32), algorithmic flops")
    if x3_was >= 2:
        roofline = dict(kernel="gemm_x3w_kernel: fp32 from bf16 pieces, W split beforehand " + shape_txt, bound="mfma",
                        achieved=round(6.0 * tfl, 1), peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s",
                        frac=round(6.0 * tfl / MFMA_BF16_PEAK_TFLOPS, 4), algorithmic_tflops=round(tfl, 2),
                        traffic=measured_traffic(("r03_pmc_x3w_ffn1.json",)),
                        traffic_note="NOT re-measured in this run: bytes per launch of the S=279 launch that "
                                     "profiles/r03_pmc_x3w_ffn1.json profiled (2*FETCH_SIZE + WRITE_SIZE, separate --pmc "
                                     "passes; algorithmic bytes of that launch 372.0e6: x, the 6-byte-per-element image of "
                                     "W, bias, the output); this run's launches average S*B = %d tokens" % int(tokens_per_step),
                        launches=len(probe), avg_us=round(ms * 1e3 / max(len(probe), 1), 1),
                        note="six v_mfma_f32_16x16x32_bf16 piece products per fp32 product: achieved = 6 x algorithmic "
                             "flops / time against the dense bf16 matrix peak; algorithmic_tflops = 2MNK / time "
                             "(fp32-equivalent); the probe brackets the product's launch only (the image of W is a "
                             "5-us launch of its own in front of it).  The same launch on the fp32 matrix pipe: "
                             "roofline_fp32_mfma_only")
    else:
        roofline = dict(roofline_f32, achieved=round(tfl, 2), frac=round(tfl / MFMA_F32_PEAK_TFLOPS, 4), launches=len(probe),
                        avg_us=round(ms * 1e3 / max(len(probe), 1), 1), note="the probed launch in the default run")
    line = dict(metric="DTI training pairs/sec (fwd+bwd, whole node) + achieved HBM GB/s fraction",
                value=round(pairs_per_s, 1), unit="pairs/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                ms_per_step=round(elapsed / args.steps * 1e3, 3), higher_is_better=True, scaling="weak",
                vs_baseline=None, dtype="f32", data="synthetic",
                dtype_note="fp32 operands and results everywhere; aligned products of a GFLOP or more (forward since round 3, backward dX / dW since round 2) as six bf16 x bf16 piece products of three-way split fp32 operands with fp32 accumulation (error vs fp64 equal to an fp32 FMA chain: tests/test_gpu_ops.py::test_backward_products_from_bf16_pieces_are_fp32_accurate, operand range: ::test_bf16_pieces_operand_range; why the forward is admitted: tests/test_full_size_parity.py::test_reference_fp32_scatter_of_the_joint_attention_stage); fp32_mfma_only = the same steps with every product on fp32 MFMA (bf16x3_mode %d here)" % x3_was,
                config=dict(workload=workload_text,
                            global_batch=bsz * world, parallelism="dp%d" % world,
                            trainable_params=count_parameters(model)),
                final_loss=round(final_loss, 6), roofline=roofline)
    line["config"]["sampler"] = ("ShardSampler: %d-pair synthetic epoch, %d disjoint shards, %d steps/epoch, tail dropped"
                                 % (sampler.n_pairs, world, sampler.steps_per_epoch))
    line["work"] = dict(
        tokens_per_step=round(tokens_per_step, 1), flops_per_step=round(flops_per_step, 0),
        tflops_per_gpu=round(flops_per_step / (elapsed / args.steps) / 1e12, 2),
        frac_of_fp32_matrix_peak=round(flops_per_step / (elapsed / args.steps) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
        note="per GPU, mean over the timed steps; tokens = S*B with S = 1 + A_max + P_max (+1 with the psc view), padded "
             "tokens included (no padding mask in the reference); flops = algorithmic, 2 per multiply-add, backward = 2 x "
             "forward (bench.py:step_flops); tflops_per_gpu = flops_per_step / ms_per_step -- with the backward products "
             "on the bf16 pipe this is an fp32-EQUIVALENT rate, the fp32-pipe figure is fp32_mfma_only.tflops_per_gpu")
    line["roofline_fp32_mfma_only"] = roofline_f32
    line["fp32_mfma_only"] = dict(value=round(args.steps * bsz * world / elapsed_f, 1), unit="pairs/s",
                                  ms_per_step=round(elapsed_f / args.steps * 1e3, 3),
                                  tflops_per_gpu=round(flops_per_step / (elapsed_f / args.steps) / 1e12, 2),
                                  frac_of_fp32_matrix_peak=round(flops_per_step / (elapsed_f / args.steps) / 1e12 /
                                                                 MFMA_F32_PEAK_TFLOPS, 4))
    line["with_collation"] = dict(
        value=round(args.steps * bsz * world / elapsed_c, 1), unit="pairs/s", ms_per_step=round(elapsed_c / args.steps * 1e3, 3),
        ratio_to_resident=round(elapsed / elapsed_c, 4),
        note="same steps on the same batches, every batch collated inside the timed region (native merge + pinned upload on a prefetch thread)")
    if comm is not None:
        line["comm"] = comm
    if dist_on and world == 1:
        line["rehearsal"] = "N-rank flow on a one-rank RCCL group"
    if shared:
        line["rehearsal"] = "%d ranks SHARING one GPU, gradients reduced over gloo through host memory: not a measurement" % world
    if rank == 0 and world == 1:
        if not args.no_graph_stress:
            line["graph_roofline"] = graph_stress(device)
        if not args.no_cpu_baseline:
            r = np.random.RandomState(1000)
            # the GPU box gives a one-GPU job a 16-CPU share (OMP_NUM_THREADS = 16 there): those are "all host cores"
            threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
            line["cpu_baseline"] = cpu_baseline(pool, hp, pool.sample_pairs(r, bsz), sd0, args.cpu_steps, threads)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist_on:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
