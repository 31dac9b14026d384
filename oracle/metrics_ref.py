"""ORACLE (test infrastructure only) -- the reference's per-batch regression metrics.

Restates, in numpy:
  * rms_score             /root/reference/jova/metrics/all_metrics.py:93-95   sqrt(mean((y - p)^2))
  * pearson_r2_score      /root/reference/jova/metrics/all_metrics.py:80-82   scipy.stats.pearsonr(y, p)[0] ** 2
  * concordance_index     /root/reference/jova/metrics/cindex_measure.py:15-38 (cindex_singletask) on top of
    count_swapped (Cython wrapper: argsort by prediction, then swapped_pairs, ext_src/c_swapped.c:324-365):
        swapped = #{f_a < f_b and s_a > s_b} + 0.5 #{f_a == f_b and s_a != s_b},  pairs = #{s_a != s_b}
        cindex = 1 - swapped / pairs
  * the score of Jova.evaluate, proj/dti/train_joint_attn.py:335-343: nanmean([ci, r2]) - rms
  * the filter of Metric.compute_singletask_metric (all_metrics.py:651-652): only samples with w != 0 count; labels and
    predictions are un-z-scored first (NormalizationTransformer.untransform: z * y_std + y_mean).
Pinned: swapped_pairs_ref() below calls the reference's own C (oracle/_ref/libcswapped.so, built by oracle/Makefile from
the reference's source in place); tests/test_metrics.py checks this restatement against it on data with ties.
"""
import ctypes
import os

import numpy as np

_REF_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libcswapped.so")


def have_reference_build():
    return os.path.exists(_REF_SO)


def swapped_pairs_ref(correct, predictions):
    """count_swapped of the reference: its Cython wrapper's argsort + the reference's compiled swapped_pairs()."""
    lib = ctypes.CDLL(_REF_SO)
    lib.swapped_pairs.restype = ctypes.c_double
    lib.swapped_pairs.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    order = np.argsort(predictions)
    a = np.ascontiguousarray(correct[order], dtype=np.float64)
    b = np.ascontiguousarray(predictions[order], dtype=np.float64)
    return float(lib.swapped_pairs(len(a), a.ctypes.data, len(b), b.ctypes.data))


def swapped_pairs(correct, predictions):
    s = np.asarray(correct, dtype=np.float64).reshape(-1)
    f = np.asarray(predictions, dtype=np.float64).reshape(-1)
    lt = f[:, None] < f[None, :]
    strict = np.count_nonzero(lt & (s[:, None] > s[None, :]))
    iu = np.triu_indices(len(s), k=1)
    ties = np.count_nonzero((f[iu[0]] == f[iu[1]]) & (s[iu[0]] != s[iu[1]]))
    return strict + 0.5 * ties


def n_pairs(correct):
    c = np.sort(np.asarray(correct, dtype=np.float64).reshape(-1))
    pairs, ties = 0, 0
    for i in range(1, len(c)):                      # cindex_measure.py:24-30
        ties = 0 if c[i] != c[i - 1] else ties + 1
        pairs += i - ties
    return pairs


def concordance_index(y, p):
    pairs = n_pairs(y)
    if pairs == 0:
        return float("nan")                          # (the reference calls sys.exit here)
    return 1.0 - swapped_pairs(y, p) / float(pairs)


def rms_score(y, p):
    y, p = np.asarray(y, np.float64).reshape(-1), np.asarray(p, np.float64).reshape(-1)
    return float(np.sqrt(np.mean((y - p) ** 2)))


def pearson_r2_score(y, p):
    y, p = np.asarray(y, np.float64).reshape(-1), np.asarray(p, np.float64).reshape(-1)
    ym, pm = y - y.mean(), p - p.mean()
    return float((np.dot(ym, pm) / np.sqrt(np.dot(ym, ym) * np.dot(pm, pm))) ** 2)


def dti_metrics(y, p, w=None, y_std=1.0, y_mean=0.0):
    """-> (rms, ci, r2, score) as Jova.evaluate computes them for one batch."""
    y, p = np.asarray(y, np.float64).reshape(-1), np.asarray(p, np.float64).reshape(-1)
    if w is not None:
        keep = np.asarray(w).reshape(-1) != 0
        y, p = y[keep], p[keep]
    y, p = y * y_std + y_mean, p * y_std + y_mean
    rms, ci, r2 = rms_score(y, p), concordance_index(y, p), pearson_r2_score(y, p)
    return rms, ci, r2, float(np.nanmean([ci, r2]) - rms)


def dti_metrics_multitask(y, p, w=None, y_std=1.0, y_mean=0.0):
    """[n, T] labels / predictions / weights -> (rms, ci, r2, score): every task measured on its samples with w != 0, then
    combined as Metric.compute_metric does with the constructor default arithmetic_mean=True and task_averager=np.nanmean
    (all_metrics.py:147, :256-296, :517-600; the driver's Metric(fn, np.nanmean), train_joint_attn.py:299-301):
        task value   = NaN without samples; for the concordance index also NaN -- and the task EXCLUDED -- when all kept
                       labels are equal (:271-279)
        coefficient  = included_tasks * n_t / N,  N = sum of n_t over the tasks that are not excluded (:560-581)
        result       = np.nanmean(coefficient * value) = sum over finite values of n_t * value_t / N
    Pinned by tools/make_golden_metrics.py against the reference's own Metric class (metrics_multitask.npz)."""
    y, p = np.asarray(y, np.float64), np.asarray(p, np.float64)
    n, T = y.shape
    w = np.ones_like(y) if w is None else np.asarray(w, np.float64).reshape(n, T)
    y, p = y * y_std + y_mean, p * y_std + y_mean
    res = []
    for fn, excludes_constant_labels in ((rms_score, False), (concordance_index, True), (pearson_r2_score, False)):
        vals, nobs, total = [], [], 0
        for t in range(T):
            keep = w[:, t] != 0
            n_t = int(keep.sum())
            if n_t == 0:
                vals.append(np.nan)
                nobs.append(0)
                continue
            yt, pt = y[keep, t], p[keep, t]
            if excludes_constant_labels and not (yt != yt[0]).any():
                vals.append(np.nan)                 # excluded: its samples leave N as well
                nobs.append(0)
                continue
            with np.errstate(all="ignore"):
                vals.append(fn(yt, pt))
            nobs.append(n_t)
            total += n_t
        vals, nobs = np.array(vals), np.array(nobs, np.float64)
        ok = ~np.isnan(vals)
        res.append(float((vals[ok] * nobs[ok]).sum() / total) if ok.any() and total > 0 else float("nan"))
    rms, ci, r2 = res
    return rms, ci, r2, float(np.nanmean([ci, r2]) - rms)

