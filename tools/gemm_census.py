"""Per-call census of the GEMMs of one Davis training step: shape, batch, the form the library picks (fast / generic),
duration (HIP events) and achieved TFLOP/s.  GPU box only: python tools/gemm_census.py [workload]"""
import os, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import bench
from jova_amd import synth
from jova_amd.feat.mol_graphs import ConvMol
from jova_amd.nn import functional as F_
from jova_amd.train import TUNED_DAVIS_HPARAMS, Trainer, create_integrated_net

dev = torch.device('cuda:0')
hp = dict(TUNED_DAVIS_HPARAMS)
pool = synth.SyntheticPool("davis", seed=1)
mols = [ConvMol(f, a, smiles=str(i)) for i, (f, a) in enumerate(pool.mols)]
torch.manual_seed(1)
model = create_integrated_net(hp, pool.protein_profile).to(dev).train()
tr = Trainer(model, hp)
batches, _, _ = bench.build_batches(pool, mols, hp, dev, 2, model[0].models[2][0], "davis", 1, 0)
for _ in range(3):
    tr.step(batches[0])
torch.cuda.synchronize()
rec = []
orig = F_.gemm


def traced(A, sam, sak, B, sbk, sbn, C, ldc, M, N, K, alpha=1.0, bias=None, residual=None, ldr=0, relu=False, splitk=1,
           nb1=1, nb2=1, sA=(0, 0), sB=(0, 0), sC=(0, 0), tag=None):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig(A, sam, sak, B, sbk, sbn, C, ldc, M, N, K, alpha, bias, residual, ldr, relu, splitk, nb1, nb2, sA, sB, sC)
    e1.record()
    a_kc, b_kc = sak == 1, sbk == 1
    lda, ldb = (sam if a_kc else sak), (sbn if b_kc else sbk)
    rec.append((e0, e1, M, N, K, nb1 * nb2, int(a_kc), int(b_kc), lda, ldb, splitk))


F_.gemm = traced
tr.step(batches[0])
torch.cuda.synchronize()
F_.gemm = orig
agg = collections.OrderedDict()
for e0, e1, M, N, K, nb, akc, bkc, lda, ldb, sk in rec:
    key = (M, N, K, nb, akc, bkc, lda, ldb, sk)
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += e0.elapsed_time(e1) * 1e3
tot = sum(v[1] for v in agg.values())
print("%d gemm calls, %.0f us between events in all" % (len(rec), tot))
print("     M      N      K    nb a_kc b_kc    lda    ldb  sk calls   us/call  TFLOP/s  form")
for key, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    M, N, K, nb, akc, bkc, lda, ldb, sk = key
    fast = nb == 1 and M % 128 == 0 and N % 128 == 0 and K % 16 == 0 and lda % 4 == 0 and ldb % 4 == 0
    print("%6d %6d %6d %5d %4d %4d %6d %6d %3d %5d %9.1f %8.1f  %s" % (M, N, K, nb, akc, bkc, lda, ldb, sk, n, us / n,
          2.0 * M * N * K * nb / (us / n) / 1e6, "fast" if fast else "generic"))
