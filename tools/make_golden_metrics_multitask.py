"""Writes tests/golden/metrics_multitask.npz: multi-task (output_dim = len(tasks) > 1, ToxCast-shaped) metric cases whose
expected values come from the REFERENCE's own ``Metric`` class (jova/metrics/all_metrics.py:120-610), i.e. its per-task
filter, its exclusion of constant-label tasks for the concordance index and its observation-weighted nan-mean
(arithmetic_mean=True), driven exactly as the DTI driver drives it (Metric(fn, np.nanmean),
proj/dti/train_joint_attn.py:299-301; Metric.compute_metric as called by jova/metrics/evaluate.py:49-59).

Build-container only.  Import recipe (packages registered without running their __init__, as tools/make_golden.py): two
names the image cannot provide are bound, and the header of the fixture records it:
  * ``rdkit`` -- an empty module: jova/utils/io.py imports it at module top, all_metrics.py only takes ``log`` from there;
    nothing of rdkit is called.
  * ``jova.metrics.swapped.count_swapped`` -- the Cython-generated glue (ext_src/swapped.c, Cython 0.17 of 2013) does
    not compile against this image's numpy; the name is bound to a three-line wrapper (sort both arrays by prediction,
    as the wrapper does) around the reference's OWN pair counter swapped_pairs(), ext_src/c_swapped.c compiled in place
    into oracle/_ref/ by oracle/Makefile.
jova/metrics/evaluate.py itself is not imported (its jova.trans import pulls the rdkit-based splitters); the un-z-scoring it
applies first (NormalizationTransformer.untransform: z * y_std + y_mean) is done here.

  make -C oracle && python tools/make_golden_metrics_multitask.py
"""
import os
import sys
import types
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import metrics_ref as MR                                      # noqa: E402


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


_pkg("jova", REF + "/jova").cuda = False
for sub in ("utils", "metrics"):
    _pkg("jova." + sub, "%s/jova/%s" % (REF, sub))
_rd = types.ModuleType("rdkit")
_rd.Chem = types.ModuleType("rdkit.Chem")
sys.modules["rdkit"], sys.modules["rdkit.Chem"] = _rd, _rd.Chem
_sw = types.ModuleType("jova.metrics.swapped")
_sw.count_swapped = lambda correct, predictions: MR.swapped_pairs_ref(np.asarray(correct), np.asarray(predictions))
sys.modules["jova.metrics.swapped"] = _sw
sys.modules["jova.metrics"].swapped = _sw
import jova.metrics.all_metrics as AM                                      # noqa: E402


def expected(y, p, w, std, mean):
    metrics = [AM.Metric(AM.rms_score, np.nanmean), AM.Metric(AM.concordance_index, np.nanmean),
               AM.Metric(AM.pearson_r2_score, np.nanmean)]
    yt, pt = y.astype(np.float64) * std + mean, p.astype(np.float64) * std + mean
    tasks = ["task%d" % t for t in range(y.shape[1])]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rms, ci, r2 = (float(m.compute_metric(yt, pt, w.astype(np.float64), per_task_metrics=False, n_classes=2, tasks=tasks))
                       for m in metrics)
    return np.array([rms, ci, r2, np.nanmean([ci, r2]) - rms], dtype=np.float64)          # train_joint_attn.py:335-343


def main():
    assert MR.have_reference_build(), "run `make -C oracle` first"
    r = np.random.RandomState(20261111)
    out, cases = {}, []
    # (n, T, fraction of zero weights, ties in labels (decimals or 0), special)
    for k, (n, T, frac0, dec, special) in enumerate([
            (128, 12, 0.7, 0, None), (128, 12, 0.7, 1, "empty_task"), (256, 5, 0.3, 1, "constant_task"),
            (64, 3, 0.0, 0, None), (128, 1, 0.2, 2, None), (200, 17, 0.9, 1, "empty_and_constant")]):
        y = r.normal(size=(n, T)).astype(np.float32)
        p = (0.6 * y + 0.8 * r.normal(size=(n, T))).astype(np.float32)
        if dec:
            y = np.round(y, dec).astype(np.float32)
        w = (r.uniform(size=(n, T)) >= frac0).astype(np.float32)
        if special in ("empty_task", "empty_and_constant"):
            w[:, 1] = 0                                        # a task without a single labelled sample
        if special in ("constant_task", "empty_and_constant"):
            y[:, T - 1] = 0.25                                 # all kept labels equal: no pair -> excluded from the CI
        std, mean = (0.87, 5.45) if k % 2 else (1.0, 0.0)
        tag = "m%d" % k
        out[tag + ".y"], out[tag + ".p"], out[tag + ".w"] = y, p, w
        out[tag + ".scale"] = np.array([std, mean], dtype=np.float64)
        out[tag + ".want"] = expected(y, p, w, std, mean)
        cases.append(tag)
        print(tag, (n, T, frac0, dec, special), out[tag + ".want"])
    out["cases"] = np.array(cases)
    out["made_by"] = np.array("jova.metrics.all_metrics.Metric.compute_metric of the reference; bound names: rdkit (empty, "
                              "import only), jova.metrics.swapped.count_swapped (-> the reference's swapped_pairs() C)")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "metrics_multitask.npz"), **out)


if __name__ == "__main__":
    main()
