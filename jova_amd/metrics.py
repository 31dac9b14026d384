"""Evaluation metrics on the device -- §8(f2).

Mirrors the metric set of the reference's DTI driver (proj/dti/train_joint_attn.py:303-343: ``rms_score``,
``concordance_index``, ``pearson_r2_score`` wrapped in ``Metric(..., np.nanmean)`` for a single task, and the score
``nanmean([ci, r2]) - rms``), evaluated by ``jova_dti_metrics`` on the batch where it lies in HBM.  The reference moves
every batch's predictions to the host and counts concordant pairs there (jova/metrics/cindex_measure.py:15-38); here the
numbers stay on the device and the epoch loop reads them once per phase.
"""
import torch

from . import _lib

COLUMNS = ("nanmean-rms_score", "nanmean-concordance_index", "nanmean-pearson_r2_score", "score")


def dti_metrics(outputs, y, w=None, y_std=1.0, y_mean=0.0, out=None, ws=None, per_task=None):
    """-> float64 device tensor [rms, ci, r2, score, swapped_pairs, pairs] of one batch (no host synchronisation).

    outputs / y / w: device tensors of n elements (any shape; fp32).  w == 0 drops a sample (all_metrics.py:651-652);
    y_std / y_mean undo the label z-scoring of NormalizationTransformer before the metrics are taken, as
    compute_model_performance does with ``transformers``.

    Multi-task models (``output_dim = len(tasks) > 1``, train_joint_attn.py:1027): ``y`` of shape [n, T] with T > 1 selects
    the per-task form -- every task measured on its own samples with w != 0, the tasks then combined the way
    Metric.compute_metric does (observation-weighted nan-mean, all_metrics.py:517-600); ``per_task`` (optional float64
    device tensor [T, 6]) receives rms, ci, r2, n_t, swapped, pairs of every task."""
    _lib.require_gpu(outputs, y, w)
    if y.dim() == 2 and y.shape[1] > 1:
        n, T = y.shape
        p2 = outputs.detach().reshape(n, T).float().contiguous()
        t2 = y.detach().float().contiguous()
        w2 = None if w is None else w.detach().reshape(n, T).float().contiguous()
        if out is None:
            out = torch.empty(6, dtype=torch.float64, device=p2.device)
        if ws is None or ws.numel() < 16 * T:
            ws = torch.empty(16 * T, dtype=torch.float64, device=p2.device)
        assert per_task is None or (per_task.dtype == torch.float64 and per_task.numel() >= 6 * T and per_task.is_contiguous())
        _lib.call("jova_dti_metrics_tasks", _lib.P(t2), _lib.P(p2), _lib.P(w2), int(n), int(T), float(y_std), float(y_mean),
                  _lib.P(ws), _lib.P(per_task), _lib.P(out), _lib.stream())
        return out
    p = outputs.detach().reshape(-1).float().contiguous()
    t = y.detach().reshape(-1).float().contiguous()
    if p.numel() != t.numel():
        raise ValueError("dti_metrics: %d predictions for %d labels" % (p.numel(), t.numel()))
    wv = None
    if w is not None:
        wv = w.detach().reshape(-1).float().contiguous()
        if wv.numel() != t.numel():
            raise ValueError("dti_metrics: %d weights for %d labels" % (wv.numel(), t.numel()))
    if out is None:
        out = torch.empty(6, dtype=torch.float64, device=p.device)
    if ws is None:
        ws = torch.empty(16, dtype=torch.float64, device=p.device)
    assert out.dtype == torch.float64 and out.numel() >= 6 and ws.dtype == torch.float64 and ws.numel() >= 16
    _lib.call("jova_dti_metrics", _lib.P(t), _lib.P(p), _lib.P(wv), int(t.numel()), float(y_std), float(y_mean),
              _lib.P(ws), _lib.P(out), _lib.stream())
    return out


class MetricsLog:
    """Collects one metrics row per evaluation batch in a device table; ``summary()`` is the single host read.

    Usable directly as ``Trainer.fit(evaluate=log)``: calling it with ``(outputs, batch)`` records the batch and returns
    its score as a device scalar, the way Jova.evaluate returns the score for the epoch statistics
    (train_joint_attn.py:474-506)."""

    def __init__(self, device, y_std=1.0, y_mean=0.0, capacity=1024):
        self.y_std, self.y_mean = float(y_std), float(y_mean)
        self.table = torch.full((capacity, 6), float("nan"), dtype=torch.float64, device=device)
        self.ws = torch.empty(16, dtype=torch.float64, device=device)
        self.n = 0

    def reset(self):
        self.n = 0

    def __call__(self, outputs, batch):
        if self.n == self.table.shape[0]:
            self.table = torch.cat([self.table, torch.full_like(self.table, float("nan"))])
        row = self.table[self.n]
        dti_metrics(outputs, batch.y, batch.w, self.y_std, self.y_mean, out=row, ws=self.ws)
        self.n += 1
        return row[3]

    def rows(self):
        return self.table[:self.n]

    def summary(self):
        """dict of per-metric np.nanmean over the recorded batches plus the plain mean score (train_joint_attn.py:531)."""
        if self.n == 0:
            return {k: float("nan") for k in COLUMNS}
        r = self.rows()[:, :4]
        ok = ~torch.isnan(r)
        mean = (torch.where(ok, r, torch.zeros_like(r)).sum(0) / ok.sum(0)).tolist()
        res = dict(zip(COLUMNS[:3], mean[:3]))
        res["score"] = float(self.rows()[:, 3].mean().item())
        return res
