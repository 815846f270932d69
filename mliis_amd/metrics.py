"""Host-side metrics of the evaluation surface: per-image IoU (reptile.py:526-549), Shaban-style counts
(reptile.py:552-566), CI95 (utils/util.py:133-136) and the EarlyStopper (hyperparam_search.py:24-68).
Values pinned by tests/golden/host_logic.json."""
import operator
from typing import Optional, Sequence

import numpy as np


def iou(prediction: np.ndarray, label: np.ndarray, epsilon: float = 1e-7, class_of_interest_channel: Optional[int] = 1,
        round_labels: bool = True) -> float:
    prediction, label = np.asarray(prediction), np.asarray(label)
    if prediction.ndim > 3:
        raise ValueError("Function is intended for single image masks, not batches.")
    if prediction.shape != label.shape:
        raise ValueError("prediction shape and label shape must be equal but are: {} and {} respectively.".format(
            prediction.shape, label.shape))
    if class_of_interest_channel is not None:
        prediction, label = prediction[..., class_of_interest_channel], label[..., class_of_interest_channel]
    p = np.round(prediction).astype(bool)
    l = (np.round(label) if round_labels else label).astype(bool)
    return (np.count_nonzero(p & l) + epsilon) / (np.count_nonzero(p | l) + epsilon)


def measure(y_in, pred_in, thresh: float = 0.5):
    y, p = np.asarray(y_in) > thresh, np.asarray(pred_in) > thresh
    return (int((y & p).sum()), int((~y & ~p).sum()), int((~y & p).sum()), int((y & ~p).sum()))


def iou_img(tp, fp, fn) -> float:
    return tp / float(max(tp + fp + fn, 1))


def ci95(a: Sequence[float]) -> float:
    a = np.asarray(a, dtype=np.float64)
    return float(1.96 * a.std() / np.sqrt(len(a)))


class EarlyStopper:
    def __init__(self, patience: int = 10, metric_should_increase: bool = True, min_steps: int = 0):
        self.patience, self.min_steps = patience, min_steps
        self._better = operator.gt if metric_should_increase else operator.lt
        self._best_metric = None
        self._best_num_steps = min_steps if min_steps > 0 else None
        self.num_evals_without_improving = 0

    def continue_training(self, metric, total_steps_taken) -> bool:
        if total_steps_taken <= self.min_steps:
            self._best_metric = metric
            return True
        if self._best_metric is None or self._better(metric, self._best_metric):
            self.num_evals_without_improving = 0
            self._best_metric, self._best_num_steps = metric, total_steps_taken
            return True
        self.num_evals_without_improving += 1
        return self.num_evals_without_improving <= self.patience

    def best_metric(self):
        return self._best_metric

    def best_num_steps(self):
        return self._best_num_steps
