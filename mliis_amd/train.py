"""Outer meta-training loop -- re-statement of meta_learners/supervised_reptile/supervised_reptile/train.py:18-135
(`train_gecko`): linear meta-step-size anneal (:90-92), one meta-learner train_step per iteration (:94-98), periodic
evaluation on train/test tasks (:100-121), checkpoint every 100 meta-iterations and at the end (:129-131), ETA print
(utils/util.py:94-98).  The TF session/model handles are replaced by a mliis_amd.learner.Learner; TensorBoard event files are
replaced by `<save_dir>/{train,test}/scalars.jsonl` (same tags: IoU, meta_step_size).
"""
from __future__ import annotations

import json
import os
import time
from typing import Optional

import numpy as np

from .checkpoint import Saver
from .reptile import Gecko


def log_estimated_time_remaining(start_time, cur_step, total_steps, unit_name="meta-step"):
    elapsed = (time.time() - start_time) / 60.0
    print("This {} took:".format(unit_name), elapsed, "minutes.")
    print("Estimated training hours remaining:%.4f" % ((total_steps - cur_step) * elapsed / 60.0))
    return elapsed


def meta_step_size_at(i: int, meta_iters: int, meta_step_size: float, meta_step_size_final: float) -> float:
    frac_done = i / meta_iters
    return frac_done * meta_step_size_final + (1 - frac_done) * meta_step_size


class _ScalarWriter:
    def __init__(self, d):
        os.makedirs(d, exist_ok=True)
        self.f = open(os.path.join(d, "scalars.jsonl"), "a")

    def add(self, step, **tags):
        self.f.write(json.dumps(dict(step=step, **tags)) + "\n")
        self.f.flush()


def train_gecko(learner, train_set, test_set, save_dir, num_classes=5, num_shots=5, inner_batch_size=5, inner_iters=20, replacement=False,
                meta_step_size=0.1, meta_step_size_final=0.1, meta_batch_size=1, meta_iters=10000, eval_inner_batch_size=5,
                eval_inner_iters=50, eval_interval=10, weight_decay_rate=1, time_deadline=None, train_shots=None, transductive=False,
                meta_fn=Gecko, log_fn=print, save_checkpoint_every_n_meta_iters=100, max_checkpoints_to_keep=2, augment=False,
                lr_scheduler=None, lr=None, save_best_seen=False, num_tasks_to_eval=100, aug_rate: Optional[float] = None, dist=None,
                seed: int = 0, verbose: bool = True, checkpoint_format: str = "npz", aug_pool=None, lanes=()):
    os.makedirs(save_dir, exist_ok=True)
    saver = Saver(max_to_keep=max_checkpoints_to_keep, fmt=checkpoint_format)
    best_saver = Saver(max_to_keep=1, fmt=checkpoint_format) if save_best_seen else None
    best_eval_iou = -np.inf
    pre_step_op = weight_decay_rate if weight_decay_rate != 1 else None
    reptile = meta_fn(learner, transductive=transductive, pre_step_op=pre_step_op, lr_scheduler=lr_scheduler, augment=augment,
                      aug_rate=aug_rate, dist=dist, seed=seed, aug_pool=aug_pool, lanes=lanes)
    D = reptile.dist
    rank0 = D.rank == 0
    writers = {"train": _ScalarWriter(os.path.join(save_dir, "train")), "test": _ScalarWriter(os.path.join(save_dir, "test"))} if rank0 else {}
    for i in range(meta_iters):
        begin = time.time()
        cur = meta_step_size_at(i, meta_iters, meta_step_size, meta_step_size_final)
        if verbose and rank0:
            print("Reptile training step {} of {}".format(i + 1, meta_iters))
            print("Current meta-step size: {}".format(cur))
        reptile.train_step(train_set, num_classes=num_classes, num_shots=(train_shots or num_shots), inner_batch_size=inner_batch_size,
                           inner_iters=inner_iters, replacement=replacement, meta_step_size=cur, meta_batch_size=meta_batch_size, lr=lr)
        if eval_interval and i % eval_interval == 0 and hasattr(reptile, "evaluate"):
            # Multi-rank: EVERY rank runs the evaluation (same parameters after the all-reduce, same task lists, same draws from the
            # global generator), so the ranks stay in lock-step: nobody waits in the next all-reduce for a rank that is still
            # evaluating (RCCL watchdog), and the host generators stay identical on all ranks.  Rank 0 records the result.
            ious = []
            for name, dataset in (("train", train_set), ("test", test_set)):
                mean_iou, _ = reptile.evaluate(dataset, num_classes=num_classes, num_shots=num_shots, inner_batch_size=eval_inner_batch_size,
                                               inner_iters=eval_inner_iters, replacement=replacement, eval_all_tasks=False,
                                               num_tasks_to_sample=num_tasks_to_eval)
                if rank0:
                    writers[name].add(i, IoU=float(mean_iou), meta_step_size=float(cur))
                ious.append(mean_iou)
            # rank 0's numbers on every rank: the per-rank device mask generators (and augmenter draws) may differ, and the
            # best-checkpoint decision below must be the same everywhere
            ious = D.broadcast_floats(ious, device=getattr(learner, "device", None))
            if rank0:
                log_fn("Train step %d: train=%f test=%f" % (i, ious[0], ious[1]))
            if save_best_seen and ious[1] > best_eval_iou:
                best_eval_iou = ious[1]
                if rank0:
                    learner.synchronize()   # evaluate() restores the variables asynchronously on the learner's stream
                    best_saver.save(learner.named_numpy(), os.path.join(save_dir, "best_eval"), i)
        if rank0 and (i % save_checkpoint_every_n_meta_iters == 0 or i == meta_iters - 1):
            learner.synchronize()
            saver.save(learner.named_numpy(), save_dir, i)
        # the deadline is a per-rank wall clock: the exit is taken by ALL ranks in the same iteration (collective OR), otherwise
        # the ranks that go on would hang in the next all-reduce
        if time_deadline is not None and D.any_true(time.time() > time_deadline, device=getattr(learner, "device", None)):
            break
        if verbose and rank0:
            log_estimated_time_remaining(begin, i, meta_iters)
    return reptile
