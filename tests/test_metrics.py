"""§8(f2): evaluation metrics.  CPU part: the numpy restatement (oracle/metrics_ref.py) against the reference's own
compiled swapped_pairs() (oracle/_ref, when built) and against the committed golden cases (made by
tools/make_golden_metrics.py from that C code + scipy + sklearn).  GPU part: jova_dti_metrics against both; pair counts
are compared as exact integers, the fp64 quantities to 1e-12 relative."""
import os

import numpy as np
import pytest
import torch

from oracle import metrics_ref as MR

GOLD = os.path.join(os.path.dirname(__file__), "golden", "metrics_cases.npz")
REL = 1e-12


def _cases():
    z = np.load(GOLD)
    for c in z["cases"]:
        c = str(c)
        yield c, z[c + ".y"], z[c + ".p"], z[c + ".w"], z[c + ".scale"], z[c + ".want"]


def _close(got, want, rel=REL):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    both_nan = np.isnan(got) & np.isnan(want)
    return bool(np.all(both_nan | (np.abs(got - want) <= rel * np.maximum(1.0, np.abs(want)))))


def test_oracle_against_golden_cases():
    for c, y, p, w, sc, want in _cases():
        keep = w != 0
        yt, pt = y[keep].astype(np.float64) * sc[0] + sc[1], p[keep].astype(np.float64) * sc[0] + sc[1]
        assert MR.swapped_pairs(yt, pt) == want[4] and MR.n_pairs(yt) == want[5], c
        assert _close(MR.dti_metrics(y, p, w, sc[0], sc[1]), want[:4], 1e-10), c


@pytest.mark.skipif(not MR.have_reference_build(), reason="oracle/_ref not built (needs the reference tree: make -C oracle)")
def test_oracle_pair_count_equals_reference_c():
    r = np.random.RandomState(5)
    for n in (1, 2, 3, 17, 200, 777):
        for dec in (0, 1, 3):
            y, p = np.round(r.normal(size=n), dec), np.round(r.normal(size=n), dec)
            assert MR.swapped_pairs(y, p) == MR.swapped_pairs_ref(y, p), (n, dec)


def test_oracle_degenerate_inputs():
    assert np.isnan(MR.concordance_index(np.ones(4), np.arange(4.0)))            # no pair with different labels
    rms, ci, r2, score = MR.dti_metrics(np.arange(4.0), np.arange(4.0))
    assert rms == 0.0 and ci == 1.0 and abs(r2 - 1.0) < 1e-15 and abs(score - 1.0) < 1e-15
    rms, ci, r2, score = MR.dti_metrics(np.arange(4.0), -np.arange(4.0))         # fully reversed
    assert ci == 0.0 and abs(r2 - 1.0) < 1e-15


MULTI = os.path.join(os.path.dirname(__file__), "golden", "metrics_multitask.npz")


def _multi_cases():
    z = np.load(MULTI)
    for c in z["cases"]:
        c = str(c)
        yield c, z[c + ".y"], z[c + ".p"], z[c + ".w"], z[c + ".scale"], z[c + ".want"]


def test_oracle_multitask_equals_reference_metric_class():
    """metrics_multitask.npz holds what the reference's Metric.compute_metric returned (tools/make_golden_metrics_multitask.py):
    12- and 17-task batches with 70 - 90 % of the labels missing (w = 0), a task without samples, a task whose kept labels
    are all equal (excluded from the concordance index), and the single-task limit."""
    n = 0
    for c, y, p, w, sc, want in _multi_cases():
        assert _close(MR.dti_metrics_multitask(y, p, w, sc[0], sc[1]), want, 1e-10), c
        if y.shape[1] == 1:                                                   # T = 1: the single-task function agrees
            assert _close(MR.dti_metrics(y, p, w, sc[0], sc[1]), want, 1e-10), c
        n += 1
    assert n == 6


@pytest.mark.gpu
def test_device_multitask_metrics_equal_reference_metric_class(device):
    """jova_dti_metrics_tasks on [n, T] batches against the reference-made values; per-task rows against the oracle."""
    from jova_amd.metrics import dti_metrics
    for c, y, p, w, sc, want in _multi_cases():
        n, T = y.shape
        per_task = torch.full((T, 6), float("nan"), dtype=torch.float64, device=device)
        if T == 1:                                                            # [n, 1] takes the single-task entry point
            got = dti_metrics(torch.from_numpy(p).to(device), torch.from_numpy(y).to(device), torch.from_numpy(w).to(device),
                              sc[0], sc[1]).cpu().numpy()
            assert _close(got[:4], want), (c, got, want)
            continue
        got = dti_metrics(torch.from_numpy(p).to(device), torch.from_numpy(y).to(device), torch.from_numpy(w).to(device),
                          sc[0], sc[1], per_task=per_task).cpu().numpy()
        assert _close(got[:4], want), (c, got, want)
        rows = per_task.cpu().numpy()
        for t in range(T):
            keep = w[:, t] != 0
            assert rows[t, 3] == keep.sum()
            if keep.sum() == 0:
                assert np.isnan(rows[t, :3]).all()
                continue
            yt, pt = y[keep, t].astype(np.float64) * sc[0] + sc[1], p[keep, t].astype(np.float64) * sc[0] + sc[1]
            assert rows[t, 4] == MR.swapped_pairs(yt, pt) and rows[t, 5] == MR.n_pairs(yt), (c, t)
            assert _close(rows[t, 0], MR.rms_score(yt, pt)), (c, t)
        assert got[4] == rows[:, 4].sum() and got[5] == rows[:, 5].sum()


@pytest.mark.gpu
def test_device_metrics_equal_golden_cases(device):
    from jova_amd.metrics import dti_metrics
    for c, y, p, w, sc, want in _cases():
        got = dti_metrics(torch.from_numpy(p).to(device), torch.from_numpy(y).to(device), torch.from_numpy(w).to(device),
                          sc[0], sc[1]).cpu().numpy()
        assert got[4] == want[4] and got[5] == want[5], (c, got, want)            # exact counts
        assert _close(got[:4], want[:4]), (c, got, want)


@pytest.mark.gpu
def test_device_metrics_vs_oracle_random_ties_weights_and_shapes(device):
    from jova_amd.metrics import dti_metrics
    r = np.random.RandomState(11)
    for n in (1, 2, 5, 63, 64, 255, 256, 257, 511, 1300, 5000):
        for dec in (1, 4):
            y = np.round(r.normal(size=(n, 1)), dec).astype(np.float32)
            p = np.round(0.5 * y + r.normal(size=(n, 1)), dec).astype(np.float32)
            w = (r.uniform(size=(n, 1)) > 0.2).astype(np.float32) if n > 2 else None
            got = dti_metrics(torch.from_numpy(p).to(device), torch.from_numpy(y).to(device),
                              None if w is None else torch.from_numpy(w).to(device), 0.87, 5.45).cpu().numpy()
            keep = np.ones(n, bool) if w is None else (w.reshape(-1) != 0)
            yt, pt = y.reshape(-1)[keep].astype(np.float64) * 0.87 + 5.45, p.reshape(-1)[keep].astype(np.float64) * 0.87 + 5.45
            if keep.sum() >= 1:
                assert got[4] == MR.swapped_pairs(yt, pt) and got[5] == MR.n_pairs(yt), (n, dec, got)
            if keep.sum() >= 2 and MR.n_pairs(yt) > 0:
                assert _close(got[:4], MR.dti_metrics(y, p, w, 0.87, 5.45)), (n, dec, got)


@pytest.mark.gpu
def test_device_metrics_degenerate_inputs(device):
    from jova_amd.metrics import dti_metrics
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=device)
    got = dti_metrics(t([1.0, 2.0, 3.0]), t([4.0, 4.0, 4.0])).cpu().numpy()       # constant labels
    assert np.isnan(got[1]) and np.isnan(got[2]) and np.isnan(got[3]) and got[5] == 0 and abs(got[0] - np.sqrt(14 / 3)) < 1e-12
    got = dti_metrics(t([1.0, 2.0]), t([1.0, 2.0]), t([0.0, 0.0])).cpu().numpy()  # every sample masked
    assert np.isnan(got[:4]).all() and got[4] == 0 and got[5] == 0
    got = dti_metrics(t([]), t([])).cpu().numpy()
    assert np.isnan(got[:4]).all()
    got = dti_metrics(t([3.0, 3.0, 1.0]), t([1.0, 2.0, 3.0])).cpu().numpy()       # tied predictions: half swaps
    assert got[4] == MR.swapped_pairs(np.array([1.0, 2.0, 3.0]), np.array([3.0, 3.0, 1.0])) == 2.5 and got[5] == 3
    with pytest.raises(ValueError):
        dti_metrics(t([1.0, 2.0]), t([1.0]))


@pytest.mark.gpu
def test_metrics_log_collects_batches_without_host_reads(device):
    from jova_amd.metrics import MetricsLog

    class B:
        pass
    r = np.random.RandomState(3)
    log = MetricsLog(device, y_std=2.0, y_mean=1.0, capacity=2)
    want = []
    for k in range(5):                                                             # grows past its capacity
        b = B()
        y, p = r.normal(size=40).astype(np.float32), r.normal(size=40).astype(np.float32)
        b.y, b.w = torch.from_numpy(y).to(device), torch.ones(40, device=device)
        s = log(torch.from_numpy(p).to(device), b)
        assert s.is_cuda and s.numel() == 1
        want.append(MR.dti_metrics(y, p, None, 2.0, 1.0))
    got = log.summary()
    want = np.array(want)
    assert _close([got["nanmean-rms_score"], got["nanmean-concordance_index"], got["nanmean-pearson_r2_score"], got["score"]],
                  want.mean(0))
