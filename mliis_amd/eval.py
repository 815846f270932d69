"""Evaluation drivers: `evaluate_gecko`, `optimize_update_hyperparams` (UHO) and `run_k_shot_learning_curves_experiment`.

`evaluate_gecko` -- re-statement of meta_learners/supervised_reptile/supervised_reptile/eval.py:18-90: `num_samples` evaluation passes
(each fine-tunes on num_shots examples of a sampled task -- or of every task when serially_eval_all_tasks -- and scores the held-out
examples), mean IoU over passes, 95 % CI over all task splits (utils/util.py:133-136).  TF handles replaced by the learner."""
from __future__ import annotations

import itertools
from typing import Dict, List, Optional, Tuple

import numpy as np

from .metrics import ci95
from .reptile import Gecko, SingleRank


def evaluate_gecko(learner, dataset, num_classes=1, num_shots=5, eval_inner_batch_size=5, eval_inner_iters=50, replacement=False,
                   num_samples=100, transductive=False, weight_decay_rate=1, meta_fn=Gecko, visualize_predicted_segmentations=False,
                   save_fine_tuned_checkpoints=False, save_fine_tuned_checkpoints_dir: Optional[str] = None, lr_scheduler=None, lr=None,
                   augment=False, serially_eval_all_tasks: bool = False, aug_rate: Optional[float] = None, aug_pool=None,
                   lanes=(), **_ignored) -> Tuple[float, Dict[str, List[float]]]:
    print("Evaluating with eval_inner_iters: {}".format(eval_inner_iters))
    print("Evaluating with lr: {}".format(lr))
    pre_step_op = weight_decay_rate if weight_decay_rate != 1 else None
    gecko = meta_fn(learner, transductive=transductive, pre_step_op=pre_step_op, lr_scheduler=lr_scheduler, augment=augment,
                    aug_rate=aug_rate, rng_mode="reference", dist=_Single(), aug_pool=aug_pool, lanes=lanes)
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


_Single = SingleRank   # evaluation always runs on one rank (the reference has no distributed evaluation)


def optimize_update_hyperparams(learner, dataset, num_classes=1, num_shots=5, eval_inner_batch_size=5, eval_inner_iters=5, replacement=False,
                                num_samples=100, transductive=False, weight_decay_rate=1, meta_fn=Gecko, save_fine_tuned_checkpoints=False,
                                save_fine_tuned_checkpoints_dir: Optional[str] = None, lr_scheduler=None, lr=None,
                                lr_search_range_low: float = 0.0005, lr_search_range_high: float = 0.05, drop_rate=None,
                                drop_rate_search_range_low: float = 0.1, drop_rate_search_range_high: float = 0.8, aug_rate: float = 0.5,
                                aug_rate_search_range_low: float = 0.5, aug_rate_search_range_high: float = 0.5,
                                batch_size_search_range_low: int = 8, batch_size_search_range_high: int = 8, augment=False,
                                serially_eval_all_tasks: bool = True, min_steps: int = 0, max_steps: int = 80, num_configs_to_sample=100,
                                num_train_val_data_splits_to_sample_per_config=1, save_dir: Optional[str] = None,
                                results_csv_name: str = "GP_val-set_hyper_param_search_results.csv",
                                eval_tasks_with_median_early_stopping_iterations: bool = False, estimator: str = "GP", aug_pool=None,
                                **_ignored):
    """Update-hyperparameter optimisation on a set of (validation) tasks -- eval.py:93-182: every sampled configuration
    (lr, drop rate, aug rate, inner batch size) is scored by Gecko.evaluate_with_early_stopping over the tasks; returns
    (best lr, expected best number of fine-tuning steps) and writes `<save_dir>/<results_csv_name>_<shots>-shot.csv`.
    (run_metasegnet.py:143-150 also passes `b=args.uho_outer_iters`, which the reference function does not accept -- the reference
    call fails with a TypeError; the stray argument is ignored here.)"""
    import os
    from . import hyperparam_search as hs
    supported_estimators = {"GP"}
    assert estimator in supported_estimators
    if save_fine_tuned_checkpoints:
        print("Saving fine-tuned checkpoints to {}".format(save_fine_tuned_checkpoints_dir))
    pre_step_op = weight_decay_rate if weight_decay_rate != 1 else None
    gecko = meta_fn(learner, transductive=transductive, pre_step_op=pre_step_op, lr_scheduler=lr_scheduler, augment=augment,
                    rng_mode="reference", dist=_Single(), aug_pool=aug_pool)
    params = {"dataset": dataset, "num_classes": num_classes, "num_shots": num_shots, "inner_batch_size": eval_inner_batch_size,
              "replacement": replacement, "eval_all_tasks": serially_eval_all_tasks, hs.LEARNING_RATE_NAME: lr, hs.DROPOUT_RATE_NAME: drop_rate,
              hs.AUG_RATE_NAME: aug_rate, "eval_tasks_with_median_early_stopping_iterations": eval_tasks_with_median_early_stopping_iterations,
              "min_steps": min_steps, "max_steps": max_steps}
    if eval_tasks_with_median_early_stopping_iterations:
        print("Evaluating val-set tasks with median iterations returned by early stopping.")
    before_ext, ext = os.path.splitext(results_csv_name)
    results_csv_name = before_ext + "_{}-shot".format(num_shots) + ext
    save_results_to = os.path.join(save_dir, results_csv_name) if save_dir is not None else results_csv_name
    if save_dir is not None:
        os.makedirs(save_dir, exist_ok=True)
    best_lr, expected_best_step_num = hs.lr_droprate_aug_rate_batch_size_gp_search(
        gecko.evaluate_with_early_stopping, params, lr_search_range_low=lr_search_range_low, lr_search_range_high=lr_search_range_high,
        drop_rate_search_range_low=drop_rate_search_range_low, drop_rate_search_range_high=drop_rate_search_range_high,
        aug_rate_search_range_low=aug_rate_search_range_low, aug_rate_search_range_high=aug_rate_search_range_high,
        batch_size_search_range_low=batch_size_search_range_low, batch_size_search_range_high=batch_size_search_range_high,
        n=num_configs_to_sample, m=num_train_val_data_splits_to_sample_per_config, save_results_to=save_results_to)
    return best_lr, expected_best_step_num


DEFAULT_K_RANGE = [1, 5, 10, 50, 100, 200, 400]


def run_k_shot_learning_curves_experiment(learner, dataset, num_classes=1, num_shots=5, eval_inner_batch_size=8, eval_inner_iters=5,
                                          replacement=False, num_samples=100, transductive=True, weight_decay_rate=1, meta_fn=Gecko,
                                          lr_scheduler=None, lr=None, augment=True, aug_rate: float = 0.5,
                                          csv_outpath: Optional[str] = "k-shot-results.csv", iter_range=None, k_range=None,
                                          test_samples: int = 20, aug_pool=None, **_ignored):
    """k-shot learning curves (eval.py:187-241): `num_samples` repetitions per task of Gecko.evaluate_k_shot_range over
    k = 1, 5, 10, 50, 100, 200, 400; the (k, mIoU) table goes to `csv_outpath` (the reference ends by rewriting the file with the
    current table only, so that is what is written).  `k_range` / `test_samples` are extensions for smaller tasks."""
    from .reptile import DEFAULT_ITER_RANGE
    k_range = DEFAULT_K_RANGE if k_range is None else list(k_range)
    print("Running k-shot learning curves experiment over k-ranges {} and dataset {}".format(k_range, [x.name for x in dataset]))
    if iter_range is None:
        iter_range = DEFAULT_ITER_RANGE[:len(k_range)]
    print("Using iter range {}".format(iter_range))
    gecko = meta_fn(learner, transductive=transductive, pre_step_op=weight_decay_rate, lr_scheduler=lr_scheduler, augment=augment,
                    aug_rate=aug_rate, rng_mode="reference", dist=_Single(), aug_pool=aug_pool)
    ks, results = gecko.evaluate_m_k_shot_ranges_all_tasks(tasks=dataset, k_range=k_range, m=num_samples, inner_batch_size=eval_inner_batch_size,
                                                           inner_iters=eval_inner_iters, replacement=replacement, lr=lr, test_samples=test_samples,
                                                           iter_range=iter_range, aug_rate=aug_rate)
    print("k-shot learning curve results:")
    print("ks:")
    print(ks)
    print("IoUs")
    print(results)
    if csv_outpath is not None:
        import csv
        with open(csv_outpath, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["k", "mIoU"])
            w.writerows(zip(ks, results))
    return ks, results
