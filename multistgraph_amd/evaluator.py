"""The evaluator's metric table on the device.

``DeviceEvaluator`` mirrors the surface of the reference's ``TrafficStateEvaluator`` (libcity/evaluator/
traffic_state_evaluator.py:11-174: ``collect`` / ``evaluate`` / ``clear``, the ten metrics of TrafficStateEvaluator.json,
modes "single" and "average") but never copies predictions to the host: ``collect`` reduces a batch to 14 sums per
horizon on the device (matgcn_metric_sums), ``evaluate`` reads one small table back.  ``groupstd_table`` is the
per-horizon table TrafficStateExecutor.evaluate writes after re-transforming with the per-tract mean / std
(libcity/executor/traffic_state_executor.py:293-322).

HIP path only: CPU tensors raise.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from . import _lib
from .ops import metric_sums, metric_table

ALLOWED_METRICS = list(_lib.METRICS)


class DeviceEvaluator:
    """config keys as the reference: ``metrics`` (default ['MAE']), ``evaluator_mode`` ('single' | 'average'),
    ``min_s`` (1e-4).  Every ``collect`` call is one batch of the reference's evaluator (``evaluate`` averages the
    batches' metrics, :123-131); with ``streaming=True`` all calls accumulate into ONE batch - the executor collects
    the whole test set in one call (traffic_state_executor.py:289), and this is that call fed batch by batch."""

    def __init__(self, config, streaming: bool = False):
        self.metrics: List[str] = list(config.get("metrics", ["MAE"]))
        self.mode = str(config.get("evaluator_mode", "single")).lower()
        self.min_s = float(config.get("min_s", 1e-4))
        self.streaming = bool(streaming)
        if not isinstance(config.get("metrics", ["MAE"]), list):
            raise TypeError("Evaluator type is not list")
        for m in self.metrics:
            if m not in ALLOWED_METRICS:
                raise ValueError("the metric {} is not allowed in TrafficStateEvaluator".format(str(m)))
        if self.mode not in ("single", "average"):
            raise ValueError("Error parameter evaluator_mode={}, please set `single` or `average`.".format(self.mode))
        self.clear()

    def clear(self):
        self.result: Dict[str, float] = {}
        self._tables: List[torch.Tensor] = []      # one (2, out, 10) table per closed batch
        self._sums: Optional[torch.Tensor] = None  # streaming: the running sums
        self.len_timeslots = 0

    def collect(self, batch):
        """batch['y_true'], batch['y_pred']: (B, timeslots, N, od) CUDA float32, already de-scaled (what the executor
        hands over, traffic_state_executor.py:268-273)."""
        if not isinstance(batch, dict):
            raise TypeError("evaluator.collect input is not a dict of user")
        y_true, y_pred = batch["y_true"], batch["y_pred"]
        if y_true.shape != y_pred.shape:
            raise ValueError("batch['y_true'].shape is not equal to batch['y_pred'].shape")
        self.collect_scaled(y_pred, y_true)

    def collect_scaled(self, pred, y, y_start: int = 0, mean=None, std=None, label_start=None):
        """The same from the model's raw output: ``pred`` (B, out, N, od) in scaled units, ``y`` the batch's labels
        (B, y_steps, N, F) - or, with ``label_start``, the device-resident series (T, N, F) -, de-scaled in the kernel
        by x*std + mean (the scaler's inverse_transform; scalar or one pair per node)."""
        self.len_timeslots = int(pred.shape[1])
        sums = metric_sums(pred, y, y_start, mean, std, min_s=self.min_s, label_start=label_start,
                           sums=self._sums if self.streaming else None)
        if self.streaming:
            self._sums = sums
        else:
            self._tables.append(metric_table(sums))

    def table(self) -> torch.Tensor:
        """(out, 10) float64 CPU tensor of the selected mode, metric order ``ALLOWED_METRICS``"""
        if self.streaming:
            if self._sums is None:
                raise RuntimeError("nothing collected")
            t = metric_table(self._sums)
        else:
            if not self._tables:
                raise RuntimeError("nothing collected")
            t = torch.stack(self._tables, 0).mean(0)   # the mean of the batches' metrics (:123-131)
        return t[0 if self.mode == "single" else 1].cpu()

    def evaluate(self) -> Dict[str, float]:
        t = self.table()
        for i in range(1, self.len_timeslots + 1):
            for m in self.metrics:
                self.result[m + "@" + str(i)] = float(t[i - 1, ALLOWED_METRICS.index(m)])
        return self.result


def groupstd_table(pred, y, group_mean, group_std, y_start: int = 0, mean=None, std=None, label_start=None,
                   s_small: float = 10.0, sums: Optional[torch.Tensor] = None):
    """The per-horizon table of the group-std re-transform (traffic_state_executor.py:293-322): values de-scaled by
    the scaler (mean / std), re-transformed per node (x * All_std + All_m, :307-308), predictions below 0 set to 0
    (:312), only elements with truth_t > s_small kept (:316-317), then MAE / MSE / RMSE (loss.masked_*_np with their
    default NaN null value: plain means), R2 / EVAR with prediction and truth EXCHANGED as the reference passes them to
    sklearn (:318-319), MAPE.  Returns (columns dict of (out,) float64 CPU tensors, the running sums to pass back in
    for the next batch)."""
    sums = metric_sums(pred, y, y_start, mean, std, group_mean, group_std, clamp_min=0.0, truth_min=s_small, min_s=-1.0,
                       label_start=label_start, sums=sums)
    t = metric_table(sums, swap_r2=True)[0].cpu()
    col = ALLOWED_METRICS.index
    return {"MAE": t[:, col("MAE")], "MSE": t[:, col("MSE")], "RMSE": t[:, col("RMSE")], "R2": t[:, col("R2")],
            "EVAR": t[:, col("EVAR")], "MAPE": t[:, col("MAPE")]}, sums
