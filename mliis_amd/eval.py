"""`evaluate_gecko` -- re-statement of meta_learners/supervised_reptile/supervised_reptile/eval.py:18-90: `num_samples` evaluation passes
(each fine-tunes on num_shots examples of a sampled task -- or of every task when serially_eval_all_tasks -- and scores the held-out
examples), mean IoU over passes, 95 % CI over all task splits (utils/util.py:133-136).  TF handles replaced by the learner."""
from __future__ import annotations

import itertools
from typing import Dict, List, Optional, Tuple

import numpy as np

from .metrics import ci95
from .reptile import Gecko


def evaluate_gecko(learner, dataset, num_classes=1, num_shots=5, eval_inner_batch_size=5, eval_inner_iters=50, replacement=False,
                   num_samples=100, transductive=False, weight_decay_rate=1, meta_fn=Gecko, visualize_predicted_segmentations=False,
                   save_fine_tuned_checkpoints=False, save_fine_tuned_checkpoints_dir: Optional[str] = None, lr_scheduler=None, lr=None,
                   augment=False, serially_eval_all_tasks: bool = False, aug_rate: Optional[float] = None,
                   **_ignored) -> Tuple[float, Dict[str, List[float]]]:
    print("Evaluating with eval_inner_iters: {}".format(eval_inner_iters))
    print("Evaluating with lr: {}".format(lr))
    pre_step_op = weight_decay_rate if weight_decay_rate != 1 else None
    gecko = meta_fn(learner, transductive=transductive, pre_step_op=pre_step_op, lr_scheduler=lr_scheduler, augment=augment,
                    aug_rate=aug_rate, rng_mode="reference", dist=_Single())
    mean_ious, task_iou_map = [], {}
    for i in range(num_samples):
        mean_iou, m = gecko.evaluate(dataset, num_classes=num_classes, num_shots=num_shots, inner_batch_size=eval_inner_batch_size,
                                     inner_iters=eval_inner_iters, replacement=replacement, eval_all_tasks=serially_eval_all_tasks,
                                     save_fine_tuned_checkpoints=save_fine_tuned_checkpoints,
                                     save_fine_tuned_checkpoints_dir=save_fine_tuned_checkpoints_dir, eval_sample_num=i, lr=lr)
        for k, v in m.items():
            task_iou_map.setdefault(k, []).append(v)
        mean_ious.append(mean_iou)
    all_ious = list(itertools.chain(*task_iou_map.values()))
    ci = ci95(all_ious)
    print("Mean of all {} task-splits: {} +/- 95% CI: {}".format(len(all_ious), float(np.nanmean(all_ious)), ci))
    print("{} NaN values out of total number of samples: {}".format(int(np.count_nonzero(np.isnan(mean_ious))), num_samples))
    mean_iou = float(np.nanmean(mean_ious))
    print("Mean of samples:")
    print("{} mean IoU, +/- 95% CI: {}".format(mean_iou, ci))
    return mean_iou, task_iou_map


class _Single:
    """Evaluation always runs on one rank (the reference has no distributed evaluation)."""
    rank, world = 0, 1

    @staticmethod
    def all_reduce_sum(t):
        return t

    @staticmethod
    def barrier():
        return None
