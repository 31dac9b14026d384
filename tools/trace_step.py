"""N training steps of one workload and nothing else, for `rocprofv3 --kernel-trace`:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python3 tools/trace_step.py davis 3 10
(python tools/kt_groups.py gpurun_out/kt 13 then prints time and launches per step).  JOVA_GEMM_NO_BF16X3=1 traces the
fp32-MFMA-only step."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B                                              # noqa: E402
from jova_amd import synth                                     # noqa: E402
from jova_amd.feat.mol_graphs import ConvMol                   # noqa: E402
from jova_amd.train import TUNED_DAVIS_HPARAMS, Trainer, create_integrated_net   # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "davis"
warm, steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3, int(sys.argv[3]) if len(sys.argv) > 3 else 10
device = torch.device("cuda:0")
hp = dict(TUNED_DAVIS_HPARAMS)
if workload == "davis_b256_default":
    hp.update(B.DEFAULT_HPARAMS_OVERRIDE)
shape, pviews, _ = B.WORKLOADS[workload]
hp["prot"] = dict(hp["prot"], model_types=list(pviews))
pool = synth.SyntheticPool(shape, seed=1)
cms = [ConvMol(f, a, smiles=str(i)) for i, (f, a) in enumerate(pool.mols)]
torch.manual_seed(1)
model = create_integrated_net(hp, pool.protein_profile).to(device).train()
tr = Trainer(model, hp)
batches, _, _ = B.build_batches(pool, cms, hp, device, 8, model[0].models[2][0], shape, 1, 0)
for i in range(warm + steps):
    tr.step(batches[i % 8])
torch.cuda.synchronize()
print("done", float(tr.loss_sum.item()))
