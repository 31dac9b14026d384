"""Writes tests/golden/metrics_cases.npz: inputs and expected metric values for §8(f2).

Expected values come from the pieces the reference itself calls: its own swapped_pairs() (ext_src/c_swapped.c, compiled
in place by oracle/Makefile into oracle/_ref/), scipy.stats.pearsonr and sklearn's mean_squared_error
(all_metrics.py:80-95), combined as cindex_measure.py:15-38 and train_joint_attn.py:335-343 combine them.  (The
reference's Python metric module itself is not imported: it loads a prebuilt Cython binary shipped in the tree.)
Run in the build container:  make -C oracle && python tools/make_golden_metrics.py
"""
import os
import sys

import numpy as np
from scipy.stats import pearsonr
from sklearn.metrics import mean_squared_error

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import metrics_ref as MR                                      # noqa: E402


def expected(y, p, w, std, mean):
    keep = w != 0
    yt = y[keep].astype(np.float64) * std + mean
    pt = p[keep].astype(np.float64) * std + mean
    c = np.sort(yt)
    pairs, ties = 0, 0
    for i in range(1, len(c)):
        ties = 0 if c[i] != c[i - 1] else ties + 1
        pairs += i - ties
    sw = MR.swapped_pairs_ref(yt, pt)
    ci = 1 - sw / float(pairs)
    r2 = pearsonr(yt, pt)[0] ** 2
    rms = np.sqrt(mean_squared_error(yt, pt))
    return np.array([rms, ci, r2, np.nanmean([ci, r2]) - rms, sw, pairs], dtype=np.float64)


def main():
    assert MR.have_reference_build(), "run `make -C oracle` first"
    r = np.random.RandomState(20260211)
    out = {}
    cases = []
    for k, (n, tie_y, tie_p, frac_w0, std, mean) in enumerate([
            (2, 0, 0, 0.0, 1.0, 0.0), (7, 1, 1, 0.0, 1.0, 0.0), (128, 1, 0, 0.0, 0.87, 5.45), (128, 2, 2, 0.1, 0.87, 5.45),
            (256, 0, 0, 0.0, 0.83, 11.7), (257, 1, 1, 0.3, 1.0, 0.0), (1000, 2, 1, 0.05, 12.3, 54.0),
            (3000, 1, 2, 0.0, 1.0, 0.0)]):
        y = r.normal(size=n).astype(np.float32)
        p = (0.6 * y + 0.8 * r.normal(size=n)).astype(np.float32)
        if tie_y:
            y = np.round(y, tie_y).astype(np.float32)
        if tie_p:
            p = np.round(p, tie_p).astype(np.float32)
        w = (r.uniform(size=n) >= frac_w0).astype(np.float32)
        if n == 2:
            y = np.array([0.5, -0.25], np.float32)
        tag = "c%d" % k
        out[tag + ".y"], out[tag + ".p"], out[tag + ".w"] = y, p, w
        out[tag + ".scale"] = np.array([std, mean], dtype=np.float64)
        out[tag + ".want"] = expected(y, p, w, std, mean)
        cases.append(tag)
    out["cases"] = np.array(cases)
    dst = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "metrics_cases.npz")
    np.savez_compressed(dst, **out)
    for c in cases:
        print(c, out[c + ".want"])


if __name__ == "__main__":
    main()
