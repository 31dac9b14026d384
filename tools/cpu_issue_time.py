"""Is the Davis step bound by the GPU or by the host issuing it?  Times 30 steps twice: the host time until the last
launch has been issued (no synchronisation inside) and the time until the GPU has finished.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
from jova_amd import synth
from jova_amd.feat.mol_graphs import ConvMol
from jova_amd.train import TUNED_DAVIS_HPARAMS, Trainer, create_integrated_net

dev = torch.device('cuda:0')
hp = dict(TUNED_DAVIS_HPARAMS)
pool = synth.SyntheticPool("davis", seed=1)
mols = [ConvMol(f, a, smiles=str(i)) for i, (f, a) in enumerate(pool.mols)]
torch.manual_seed(1)
model = create_integrated_net(hp, pool.protein_profile).to(dev).train()
tr = Trainer(model, hp)
batches, _, _ = bench.build_batches(pool, mols, hp, dev, 8, model[0].models[2][0], "davis", 1, 0)
for i in range(5):
    tr.step(batches[i % 8])
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(30):
        tr.step(batches[i % 8])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("issue %.3f ms/step   finished %.3f ms/step   (host ahead of the GPU by %.1f ms at the end)"
          % ((t1 - t0) / 30 * 1e3, (t2 - t0) / 30 * 1e3, (t2 - t1) * 1e3), flush=True)
