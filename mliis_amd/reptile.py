"""Reptile ("Gecko") and first-order MAML ("FOMLIS") meta-learners over a device-resident learner.

Re-statement of meta_learners/supervised_reptile/supervised_reptile/reptile.py:23-125 (Gecko.__init__/train_step) and
:569-663 (FOMLIS) with the TensorFlow handles (`session`, `input_ph`, `label_ph`, `minimize_op`, `lr_ph`) replaced by one
`learner` object (mliis_amd.learner.Learner on the GPU; any object with the same methods elsewhere).  What changed, on purpose:

  * parameters never leave the device: `export_variables`/`import_variables` (variables.py:70-80) are arena clones/copies and the
    outer update `theta <- theta_old + eps * mean_tasks(delta)` (variables.py:9-45; reptile.py:124-125,646-647) is one axpby;
  * the tasks of a meta-batch are sharded over ranks (task t -> rank t mod P); every rank sums its task deltas and ONE
    all-reduce(sum) over [delta | BN-moving contribution] combines them (RCCL over xGMI on the GPU node);
  * BN moving statistics are not reset between tasks in the reference (reptile.py:34 vs :35-36) and therefore follow the
    SEQUENTIAL exponential moving average over all tasks' steps.  Training-mode statistics never read the moving average, so
    that sequence is reproduced exactly under sharding: each task accumulates S_t from a zero start and
    m <- 0.99^(B*T) * m0 + sum_t 0.99^((B-1-t)*T) * S_t   (SURVEY.md 8(e)-1);
  * randomness: rng_mode="reference" consumes Python's global `random` exactly like the reference (valid on one rank);
    rng_mode="per_task" derives an independent generator per (meta_iter, task) so results do not depend on the rank count.
Kept quirks: Gecko.train_step runs TWO optimizer steps per batch when `lr` is given and no scheduler is set
(reptile.py:114-121); FOMLIS.train_step ignores the lr scheduler (reptile.py:639-643, E9).
"""
from __future__ import annotations

import random
from typing import List, Optional, Sequence

import torch

from . import metaseg
from .spec import BN_MOMENTUM


class Dist:
    """Thin view of torch.distributed (backend "nccl" == RCCL on ROCm, "gloo" on CPU).  World size 1 when uninitialised."""

    def __init__(self):
        import torch.distributed as dist
        self._d = dist if (dist.is_available() and dist.is_initialized()) else None
        self.rank = self._d.get_rank() if self._d else 0
        self.world = self._d.get_world_size() if self._d else 1
        self.timed = False     # bench.py: events around every all_reduce_sum on the calling stream (see timing())
        self._events = []

    def all_reduce_sum(self, t: torch.Tensor):
        if self._d is not None and self.world > 1:
            if self.timed and t.is_cuda:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self._d.all_reduce(t, op=self._d.ReduceOp.SUM)
                e1.record()
                self._events.append((e0, e1))
            else:
                self._d.all_reduce(t, op=self._d.ReduceOp.SUM)
        return t

    def timing(self):
        """(mean ms inside the collectives, mean ms of this rank's own work between two collectives) over the timed all_reduce_sum calls
        since the last call; the events are dropped.  The collective's span on the calling stream contains the wait for the slowest
        rank, the span between two collectives is this rank's adaptation work: the two decompose a meta-step of an N > 1 run."""
        ev, self._events = self._events, []
        if not ev:
            return None, None
        ev[-1][1].synchronize()
        inside = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
        between = [ev[i][1].elapsed_time(ev[i + 1][0]) for i in range(len(ev) - 1)]
        return inside, (sum(between) / len(between) if between else None)

    def barrier(self):
        if self._d is not None and self.world > 1:
            self._d.barrier()

    def any_true(self, flag: bool, device=None) -> bool:
        """Collective OR of a per-rank decision (loop exits must be taken by all ranks in the same iteration, or the next
        all-reduce hangs): one 4-byte all-reduce(MAX)."""
        if self._d is None or self.world == 1:
            return bool(flag)
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device if self._d.get_backend() == "nccl" else "cpu")
        self._d.all_reduce(t, op=self._d.ReduceOp.MAX)
        return bool(int(t.item()))

    def broadcast_floats(self, values, device=None, src: int = 0):
        """Rank `src`'s list of floats on every rank (evaluation results computed on one rank only)."""
        if self._d is None or self.world == 1:
            return [float(v) for v in values]
        t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device if self._d.get_backend() == "nccl" else "cpu")
        self._d.broadcast(t, src=src)
        return [float(v) for v in t.tolist()]


class SingleRank:
    """A Dist that is always world size 1: for work one rank does alone while a process group exists (rank-0 evaluation-time
    fine-tuning), so it neither shards its tasks nor issues collectives the other ranks never join."""
    rank, world = 0, 1

    def all_reduce_sum(self, t):
        return t

    def barrier(self):
        pass

    def any_true(self, flag, device=None):
        return bool(flag)

    def broadcast_floats(self, values, device=None, src=0):
        return [float(v) for v in values]


DEFAULT_ITER_RANGE = [1, 5, 10, 25, 50, 100, 200]   # reptile.py:21


def _task_rng(seed: int, meta_iter: int, task: int) -> random.Random:
    return random.Random((seed * 1000003 + meta_iter) * 1000003 + task)


def _to_numpy(a):
    import numpy as np
    return a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)


class Gecko:
    """A meta-learning session for image segmentation that extends Reptile (reference class of the same name)."""

    meta_fn = "Reptile"

    def __init__(self, learner, variables=None, transductive: bool = False, pre_step_op=None, lr_scheduler=None, augment: bool = False,
                 aug_rate: Optional[float] = None, dist: Optional[Dist] = None, rng_mode: Optional[str] = None, seed: int = 0,
                 aug_pool=None, lanes: Sequence = ()):
        self.learner = learner
        # lanes: further Learners of the same architecture (own arenas, own streams).  The tasks of a meta-batch that fall on this
        # rank are then adapted len(lanes)+1 at a time, their inner steps issued round-robin so the launches of one task fill the
        # compute units the other leaves idle (one 8-image step alone does not fill 256 CUs: profiles/r01_notes.md).  Same update as
        # task-by-task -- the tasks are independent and the deltas are accumulated in task order -- except that every lane draws its
        # drop-connect masks from its own generator.
        self.lanes = list(lanes)
        # Adam (the reference's default inner optimizer): every lane keeps its own second-moment slots and step count, exactly as every
        # RANK does under multi-GPU sharding -- the reference's single sequential history over all tasks is not
        # reproduced then (Adam with concurrent lanes is NOT reference-parity; parity runs use --sgd or one lane); the SGD path (--sgd)
        # stays bit-identical to the task-by-task loop.  What IS kept: a restored / imported optimizer state (checkpoint resume,
        # evaluation's restore) reaches every lane, so lanes never start cold beside a warm main learner.
        self._lane_adam_epoch = None
        for ln in self.lanes:
            if ln.n_trainable != learner.n_trainable or getattr(ln, "optimizer", "sgd") != getattr(learner, "optimizer", "sgd"):
                raise ValueError("lanes must share the learner's architecture and inner optimizer")
        # (Lanes keep the split-product kernels of the default fp32 path.  Round 5 switched them to the native instruction because small
        # kernels of one learner returned wrong values while the split-product kernels of another were on the chip; round 6 named the
        # mechanism -- a packed fp32 instruction with op_sel:[0,1] on a CU that also holds a wave mixing bf16 matrix instructions with
        # memory instructions, profiles/r06_notes.md -- and fenced it on both sides: conv_x3_k / conv_filter_x3_batched_k occupy their
        # CUs alone, and the library is built without that instruction form, tests/test_build_cpu.py.)
        self._transductive = transductive
        # pre_step_op: the reference passes a TF op that multiplies all trainables by `weight_decay_rate`
        # (variables.py:48-55); here it is that rate (float) or None.
        self._pre_step_rate = 1.0 if pre_step_op is None else float(pre_step_op)
        self.lr_scheduler = lr_scheduler
        self.aug_rate = aug_rate
        self.eval_sample_number = 0
        self.dist = dist or Dist()
        self.rng_mode = rng_mode or ("reference" if self.dist.world == 1 else "per_task")
        # host augmentation of the inner-loop batches (reptile.py:44-49, augmenters/np_augmenters.py).  rng_mode "reference": the
        # global `random` / `numpy.random` streams, like the reference; "per_task": private streams re-seeded per (meta-iter, task).
        # aug_pool (augment.AugmentPool, optional): worker processes for the pixel half of the augmentation; the draws stay here.
        self.augmenter = None
        self.aug_pool = aug_pool
        # augment: False | True (pixels on the host, draw-identical to the reference) | "device" (same draws of the operations and their
        # parameters; the pixel work -- and the two per-pixel noise fields -- on the device: csrc/augment.hip, Learner.augment_batch)
        self.device_aug = augment == "device"
        if augment:
            from .augment import Augmenter
            self.augmenter = Augmenter(fields=not self.device_aug)
            if self.device_aug and not getattr(learner, "aug_capacity", 0):
                raise ValueError("augment='device' needs a learner built with augment_batch_capacity >= the inner batch size")
        print("Augmentation rate {}".format(self.aug_rate))
        if self.rng_mode == "reference" and self.dist.world > 1:
            raise ValueError("rng_mode='reference' consumes the global generator sequentially and is only valid on one rank")
        self.seed = seed
        self.meta_iter = 0
        self._comm = None
        print("{} meta-learning session instantiated ({} rank(s), rng_mode={}).".format(self.meta_fn, self.dist.world, self.rng_mode))

    # ------------------------------------------------------------------------------------------------ helpers
    def _rng(self, task_idx: int):
        return None if self.rng_mode == "reference" else _task_rng(self.seed, self.meta_iter, task_idx)

    def _batches(self, n_shots: int, inner_batch_size: int, inner_iters: int, replacement: bool, rng) -> List[List[int]]:
        return [list(b) for b in metaseg.mini_batch_indices(n_shots, inner_batch_size, inner_iters, replacement, rng)]

    def _sample(self, dataset, num_shots, rng):
        (images, labels) = metaseg.sample_task(dataset, num_shots, rng)
        if self.augmenter is None or self.device_aug:
            self.learner.load_task(images, labels)
            self._host_task = (images, labels)     # (device path: only the shapes are read when the schedule is drawn)
        else:   # augmented batches are built on the host from these arrays and uploaded one inner step at a time
            self._host_task = (_to_numpy(images), _to_numpy(labels))
        return int(images.shape[0])

    _train_aug_rate_from_self = False   # Reptile's train_step does not forward aug_rate to _mini_batches (reptile.py:108); FOMAML does

    def _augmented_task_schedule(self, inner_batch_size, inner_iters, replacement, rng, task_idx):
        """Draws one task's augmented inner-loop schedule (all generator consumption happens here, in the reference's order) and hands
        the pixel work to the pool, if there is one."""
        if rng is not None:   # per-task mode: private, reproducible streams for the augmenter as well
            import numpy as np
            from .augment import PRISTINE_ORDER
            self.augmenter.py = rng
            self.augmenter.npr = np.random.RandomState(_task_rng(self.seed, self.meta_iter, task_idx).getrandbits(32))
            self.augmenter.order = list(PRISTINE_ORDER)   # the reference's persistent shuffled order would make a task's draws depend on
            #                                              which tasks this rank saw before (i.e. on the rank count): start every task fresh
        x, y = self._host_task
        wr = getattr(self, "sample_train_val_with_replacement", False)
        sched = metaseg.AugmentedSchedule(x, y, inner_batch_size, inner_iters, replacement, self.augmenter,
                                          self.aug_rate if self._train_aug_rate_from_self else None, rng,
                                          tail_shots=getattr(self, "tail_shots", None), fomaml=self._train_aug_rate_from_self,
                                          with_replacement_train_shots=self.train_shots if wr else None,
                                          npr=self._npr(rng) if wr else None)
        return sched.submit(self.aug_pool) if (self.aug_pool is not None and not self.device_aug) else sched

    def _run_meta_batch(self, dataset, num_shots, inner_batch_size, inner_iters, replacement, meta_step_size, meta_batch_size, lr,
                        fomaml: bool):
        L, D = self.learner, self.dist
        with L.comm_context():   # every arena-sized temporary is created/used in the learner's stream order
            old = L.export_trainable()
            bn0 = L.export_bn()
            nt, nb = old.numel(), bn0.numel()
            if self._comm is None or self._comm.numel() != nt + nb:
                self._comm = torch.zeros(nt + nb, dtype=old.dtype, device=old.device)
                self._bn_zero = torch.zeros_like(bn0)
            comm = self._comm
            comm.zero_()
            delta, bn_acc = comm[:nt], comm[nt:]
            decay = BN_MOMENTUM
            # moving-average updates one task issues: one per optimizer step, and Gecko runs TWO steps per batch when `lr` is given
            # (quirk E1, _steps_per_batch) -- the sequential-average weights below count updates, not batches
            T = inner_iters * self._steps_per_batch(lr)
            # With augmentation every task of the meta-batch is sampled and its schedule DRAWN up front (same generator order as
            # task-by-task, since the inner steps consume no host randomness); the pixel work of later tasks then runs on the worker
            # pool while the device trains on the earlier ones.
            mine = [t for t in range(meta_batch_size) if self.rng_mode == "reference" or t % D.world == D.rank]
            ahead = {}
            if self.augmenter is not None and not self.device_aug:
                for t in mine:
                    rng = self._rng(t)
                    self._sample(dataset, num_shots, rng)
                    ahead[t] = self._augmented_task_schedule(inner_batch_size, inner_iters, replacement, rng, t)
            if self.lanes and self.augmenter is None:
                self._adapt_concurrently(mine, dataset, num_shots, inner_batch_size, inner_iters, replacement, meta_batch_size, lr, fomaml,
                                         old, delta, bn_acc)
                mine = []
            for t in mine:
                if self.augmenter is None:
                    rng = self._rng(t)
                    n_shots = self._sample(dataset, num_shots, rng)
                    batches = self._task_batches(n_shots, inner_batch_size, inner_iters, replacement, rng)
                elif self.device_aug:   # draws on the host (same order as the host path), pixels on the device, batch by batch
                    rng = self._rng(t)
                    self._sample(dataset, num_shots, rng)
                    batches = self._augmented_task_schedule(inner_batch_size, inner_iters, replacement, rng, t).device_batches()
                else:
                    batches = ahead.pop(t).batches()
                L.import_bn(self._bn_zero)
                last_backup = None
                for j, idx in enumerate(batches):
                    if fomaml and j == inner_iters - 1:
                        last_backup = L.export_trainable()
                    if self.device_aug:              # idx = (recipes, shot indices): augmented on the device into the batch slots
                        idx = self._device_batch(L, idx[1], idx[0])
                    elif self.augmenter is not None:   # idx is an augmented (images, labels) batch: make it the resident "task"
                        L.load_task(idx[0], idx[1])
                        idx = list(range(int(idx[0].shape[0])))
                    self._step(idx, j, lr)
                # delta += theta_task - (theta_before_last_step | theta_old)
                L.axpby(1.0, L.export_trainable(), 1.0, delta)
                L.axpby(-1.0, last_backup if fomaml else old, 1.0, delta)
                L.axpby(decay ** ((meta_batch_size - 1 - t) * T), L.export_bn(), 1.0, bn_acc)
                L.import_trainable(old)
            D.all_reduce_sum(comm)
            # theta <- old + (eps / B) * sum_t delta_t ;  bn <- decay^(B*T) * bn0 + sum_t decay^((B-1-t)T) S_t
            L.axpby(meta_step_size / meta_batch_size, delta, 1.0, old)
            L.import_trainable(old)
            L.axpby(decay ** (meta_batch_size * T), bn0, 1.0, bn_acc)
            L.import_bn(bn_acc)
        self.meta_iter += 1

    def _adapt_concurrently(self, mine, dataset, num_shots, inner_batch_size, inner_iters, replacement, meta_batch_size, lr, fomaml,
                            old, delta, bn_acc):
        """The task loop of _run_meta_batch over several learners at once (called inside the main learner's comm_context, so the
        export_* / import_* / axpby calls below order the lanes' streams against the main learner's)."""
        L = self.learner
        lanes = [L] + self.lanes
        self._sync_lane_optimizer_state()
        decay, T = BN_MOMENTUM, inner_iters * self._steps_per_batch(lr)
        for g0 in range(0, len(mine), len(lanes)):
            group = []
            for lane, t in zip(lanes, mine[g0:g0 + len(lanes)]):   # host draws in task order, like the sequential loop
                rng = self._rng(t)
                images, labels = metaseg.sample_task(dataset, num_shots, rng)
                if lane is not L:   # sampled on the main learner's stream, consumed on the lane's
                    for a in (images, labels):
                        if torch.is_tensor(a) and a.is_cuda:
                            a.record_stream(lane.stream)
                lane.load_task(images, labels)
                batches = self._task_batches(int(images.shape[0]), inner_batch_size, inner_iters, replacement, rng)
                if lane is not L:
                    lane.import_trainable(old)
                lane.import_bn(self._bn_zero)
                group.append([lane, t, list(batches), None])
            for j in range(max(len(g[2]) for g in group)):
                for g in group:
                    lane, _, batches, _ = g
                    if j < len(batches):
                        if fomaml and j == inner_iters - 1:
                            g[3] = lane.export_trainable()
                        self._step(batches[j], j, lr, lane)
            for lane, t, _, last_backup in group:
                L.axpby(1.0, lane.export_trainable(), 1.0, delta)
                L.axpby(-1.0, last_backup if fomaml else old, 1.0, delta)
                L.axpby(decay ** ((meta_batch_size - 1 - t) * T), lane.export_bn(), 1.0, bn_acc)
            L.import_trainable(old)

    def _sync_lane_optimizer_state(self):
        """Adam only: after the main learner's slots were replaced from outside (load_named / import_all) copy them into every lane."""
        L = self.learner
        ep = getattr(L, "adam_epoch", None)
        if getattr(L, "adam_v", None) is None or ep is None or ep == self._lane_adam_epoch:
            return
        # Called inside L.comm_context(): the current stream IS L.stream, so each lane's copy is ordered after L's pending writes of its
        # slots (import_adam -> _in()).  The other direction needs an edge of its own: L's next Adam step rewrites adam_v IN PLACE and must
        # not start before the lanes' copies (on their streams) have read it (write-after-read).
        assert torch.cuda.current_stream(L.device) == L.stream, "_sync_lane_optimizer_state must run inside the main learner's comm_context()"
        for ln in self.lanes:
            ln.import_adam(L.adam_v, L.adam_t)
            L.stream.wait_stream(ln.stream)
        self._lane_adam_epoch = ep

    def _task_batches(self, n_shots, inner_batch_size, inner_iters, replacement, rng):
        return self._batches(n_shots, inner_batch_size, inner_iters, replacement, rng)

    @staticmethod
    def _device_batch(L, shot_idx, recipes):
        """Indices of one inner step's samples: the shots themselves when no sample of the batch is augmented (FOMAML's raw tail batch,
        or every draw kept the original), else the batch slots the device augmenter fills."""
        if all(r is None for r in recipes):
            return [int(i) for i in shot_idx]
        return L.augment_batch(shot_idx, recipes)

    def _steps_per_batch(self, lr) -> int:
        """Optimizer steps (= BN moving-average updates) `_step` issues per mini-batch."""
        return 2 if lr is not None else 1

    def _step(self, idx, j, lr, L=None):
        L = L or self.learner
        wd = self._pre_step_rate
        if lr is not None:  # reptile.py:114-116 -- first optimizer step with the given lr
            L.inner_step(idx, lr=lr, weight_decay_rate=wd)
            wd = 1.0
        if self.lr_scheduler is not None:  # :117-119
            L.inner_step(idx, lr=self.lr_scheduler.cur_lr(cur_step=j), weight_decay_rate=wd)
        else:  # :120-121 -- runs also when lr was given (quirk E1)
            L.inner_step(idx, weight_decay_rate=wd)

    # ------------------------------------------------------------------------------------------------ public surface
    def train_step(self, dataset, num_classes=1, num_shots=5, inner_batch_size=8, inner_iters=8, replacement=False, meta_step_size=0.1,
                   meta_batch_size=1, lr=None, verbose=False, **_unused_tf_handles):
        """Perform one Reptile training step (reptile.py:64-125).  `num_classes` is ignored (binary Gecko)."""
        self._run_meta_batch(dataset, num_shots, inner_batch_size, inner_iters, replacement, meta_step_size, meta_batch_size, lr, False)


    # ------------------------------------------------------------------------------------------------ evaluation
    DEFAULT_NUM_TEST_EXAMPLES = 5

    def evaluate(self, dataset, num_classes=1, num_shots=5, inner_batch_size=8, inner_iters=4, replacement=False, eval_all_tasks=False,
                 num_tasks_to_sample=1, test_shots=DEFAULT_NUM_TEST_EXAMPLES, verbose=False, save_fine_tuned_checkpoints=False,
                 save_fine_tuned_checkpoints_dir: Optional[str] = None, eval_sample_num: Optional[int] = None, lr: Optional[float] = None,
                 drop_rate: Optional[float] = None, aug_rate: Optional[float] = None, **_unused_tf_handles):
        """One evaluation pass (reptile.py:127-233): fine-tune on `num_shots` examples of each sampled task, predict the held-out
        `test_shots`, return (mean IoU, {task_name: IoU}).  Shuffles the caller's task list in place like the reference (E18)."""
        import numpy as np
        print("Evaluating {} meta-learning.".format(self.meta_fn))
        if eval_all_tasks:
            sampled = dataset
        else:
            random.shuffle(dataset)
            sampled = dataset[:num_tasks_to_sample]
        ious, task_iou_map = [], {}
        if self.lanes and self.augmenter is None and not save_fine_tuned_checkpoints:
            for name, iou in self._evaluate_concurrently(sampled, num_shots, test_shots, inner_batch_size, inner_iters, replacement, lr,
                                                         drop_rate):
                ious.append(iou)
                task_iou_map[name] = iou
            sampled = []
        for task in sampled:
            (images, labels), name = metaseg.sample_task([task], num_shots + test_shots, None, return_task_name=True)
            n = int(images.shape[0])
            self.learner.load_task(images, labels)
            train_idx, test_idx = metaseg.split_indices(n, test_shots)
            iou = self._evaluate(train_idx, test_idx, labels, inner_batch_size, inner_iters, replacement, lr=lr, task_name=name,
                                 save_fine_tuned_checkpoints=save_fine_tuned_checkpoints,
                                 save_fine_tuned_checkpoints_dir=save_fine_tuned_checkpoints_dir, eval_sample_num=eval_sample_num,
                                 aug_rate=self.aug_rate if aug_rate is None else aug_rate, images=images, drop_rate=drop_rate)
            ious.append(iou)
            task_iou_map[name] = iou
        mean_iou = float(np.nanmean(ious))
        print("Mean IoU from train on {} images and evaluate on {} test images: {}".format(num_shots, test_shots, mean_iou))
        return mean_iou, task_iou_map

    def _evaluate_concurrently(self, sampled, num_shots, test_shots, inner_batch_size, inner_iters, replacement, lr, drop_rate):
        """evaluate's task loop over the lanes: every task of a group is fine-tuned from the same restored state on a learner of
        its own, the steps issued round-robin; host draws (example sampling, mini-batch schedule) stay in task order and the
        predictions / IoUs are taken task by task afterwards, so the result equals the sequential loop's."""
        import numpy as np
        from .metrics import iou as _iou
        L = self.learner
        lanes = [L] + self.lanes
        state = L.export_all()
        out = []
        for g0 in range(0, len(sampled), len(lanes)):
            group = []
            for lane, task in zip(lanes, sampled[g0:g0 + len(lanes)]):
                (images, labels), name = metaseg.sample_task([task], num_shots + test_shots, None, return_task_name=True)
                train_idx, test_idx = metaseg.split_indices(int(images.shape[0]), test_shots)
                schedule = [list(b) for b in metaseg.mini_batch_indices(len(train_idx), inner_batch_size, inner_iters, replacement)]
                if lane is not L:
                    lane.import_all(state)
                lane.load_task(images, labels)
                group.append((lane, name, images, labels, train_idx, test_idx, schedule))
            for j in range(max(len(g[6]) for g in group)):
                for lane, _, _, _, train_idx, _, schedule in group:
                    if j < len(schedule):
                        self._fine_tune_step([train_idx[i] for i in schedule[j]], j, lr, self.lr_scheduler, drop_rate, lane)
            for lane, name, _, labels, train_idx, test_idx, _ in group:
                preds = self._test_predictions(train_idx, test_idx, lane)
                lab = labels.detach().cpu().numpy() if hasattr(labels, "detach") else np.asarray(labels)
                class_iou = float(np.nanmean([_iou(preds[j], lab[test_idx[j]]) for j in range(len(test_idx))]))
                print("Mean task IoU: {}".format(class_iou))
                out.append((name, class_iou))
            L.import_all(state)
        return out

    def _fine_tune_step(self, idx, inner_iter, lr, lr_scheduler, drop_rate, L=None):
        """One fine-tuning step with the reference's feed precedence (reptile.py:265-276,459-469): (lr, drop_rate) together, else lr,
        else the scheduler's lr, else the model defaults."""
        L, wd = L or self.learner, self._pre_step_rate
        if lr is not None and drop_rate is not None:
            L.inner_step(idx, lr=lr, weight_decay_rate=wd, drop_rate=drop_rate)
        elif lr is not None:
            L.inner_step(idx, lr=lr, weight_decay_rate=wd)
        elif lr_scheduler is not None:
            L.inner_step(idx, lr=lr_scheduler.cur_lr(cur_step=inner_iter), weight_decay_rate=wd)
        else:
            L.inner_step(idx, weight_decay_rate=wd)

    def _evaluate(self, train_idx, test_idx, labels, inner_batch_size, inner_iters, replacement, lr=None, task_name=None,
                  save_fine_tuned_checkpoints=False, save_fine_tuned_checkpoints_dir=None, eval_sample_num=None, aug_rate=None,
                  images=None, drop_rate=None):
        """Evaluates a single task's train/test split (reptile.py:235-294): ALL global variables are restored afterwards."""
        import numpy as np
        from .metrics import iou as _iou
        L = self.learner
        old = L.export_all()
        inner_iter = 0
        if self.augmenter is None:
            schedule = metaseg.mini_batch_indices(len(train_idx), inner_batch_size, inner_iters, replacement)
        elif self.device_aug:   # same draws, pixels on the device: (recipes, positions in train_idx) per step
            shape = tuple(images.shape[1:]) if images is not None else (L.arch.image_size, L.arch.image_size, 3)
            keep_p = None if aug_rate is None else 1.0 - aug_rate
            schedule = metaseg.mini_batch_indices(len(train_idx), inner_batch_size, inner_iters, replacement,
                                                  visit=lambda i: (self.augmenter.plan(shape, keep_p), i))
        else:   # fine-tune on augmented copies of the support examples (reptile.py:261-262), built on the host
            if images is None:
                raise ValueError("_evaluate with augmentation needs the task's images")
            x, y = _to_numpy(images), _to_numpy(labels)
            schedule = metaseg.augmented_batches(x[train_idx], y[train_idx], inner_batch_size, inner_iters, replacement, self.augmenter,
                                                 aug_rate, pool=self.aug_pool)
        for inner_iter, b in enumerate(schedule):
            if self.augmenter is None:
                idx = [train_idx[i] for i in b]
            elif self.device_aug:
                idx = self._device_batch(L, [train_idx[i] for (_, i) in b], [r for (r, _) in b])
            else:
                L.load_task(b[0], b[1])
                idx = list(range(int(b[0].shape[0])))
            self._fine_tune_step(idx, inner_iter, lr, self.lr_scheduler, drop_rate)
        if save_fine_tuned_checkpoints:
            from .checkpoint import save_fine_tuned_checkpoint
            L.synchronize()
            save_fine_tuned_checkpoint(L.named_numpy(), save_fine_tuned_checkpoints_dir, task_name, eval_sample_num, inner_iter)
        if self.augmenter is not None and not self.device_aug:
            L.load_task(images, labels)   # the augmented batches replaced the resident task
        preds = self._test_predictions(train_idx, test_idx)
        lab = labels.detach().cpu().numpy() if hasattr(labels, "detach") else np.asarray(labels)
        class_iou = float(np.nanmean([_iou(preds[j], lab[test_idx[j]]) for j in range(len(test_idx))]))
        print("Mean task IoU: {}".format(class_iou))
        L.import_all(old)
        return class_iou

    # ------------------------------------------------------------------------------------------------ early stopping / k-shot curves
    ES_PATIENCE = 50          # _early_stopping_learn's default patience (reptile.py:444)
    K_SHOT_ES_MAX_STEPS = 500   # evaluate_k_shot_range's early-stopping horizon (reptile.py:429)

    def _early_stopping_learn(self, train_idx, val_idx, labels, inner_batch_size, min_steps, max_steps, replacement, lr_scheduler=None,
                              lr=None, drop_rate=None, patience=None, inner_iters=None, aug_rate=None, images=None):
        """Estimates the number of fine-tuning steps for a task (reptile.py:442-480): after EVERY step the validation examples are
        predicted and their mean IoU is shown to an EarlyStopper; all variables are restored.  Returns (best step count, best mIoU)."""
        import numpy as np
        from .hyperparam_search import EarlyStopper
        from .metrics import iou as _iou
        del inner_iters
        L = self.learner
        old = L.export_all()
        if lr_scheduler is not None and lr is not None:
            raise ValueError("Only lr_scheduler or lr should be speced. Not both.")
        stopper = EarlyStopper(self.ES_PATIENCE if patience is None else patience, min_steps=min_steps)
        lab = _to_numpy(labels)
        if self.augmenter is None:
            schedule = metaseg.mini_batch_indices(len(train_idx), inner_batch_size, max_steps, replacement)
        elif self.device_aug:   # drawn lazily batch by batch (the loop may stop early), pixels on the device
            shape = (L.arch.image_size, L.arch.image_size, 3)
            keep_p = None if aug_rate is None else 1.0 - aug_rate
            schedule = metaseg.mini_batch_indices(len(train_idx), inner_batch_size, max_steps, replacement,
                                                  visit=lambda i: (self.augmenter.plan(shape, keep_p), i))
        else:
            if images is None:
                raise ValueError("_early_stopping_learn with augmentation needs the task's images")
            x = _to_numpy(images)
            schedule = metaseg.lazy_augmented_batches(x[train_idx], lab[train_idx], inner_batch_size, max_steps, replacement,
                                                      self.augmenter, aug_rate)   # lazy: the loop may stop early
        for inner_iter, b in enumerate(schedule):
            if self.augmenter is None:
                idx = [train_idx[i] for i in b]
            elif self.device_aug:
                idx = self._device_batch(L, [train_idx[i] for (_, i) in b], [r for (r, _) in b])
            else:
                L.load_task(b[0], b[1])
                idx = list(range(int(b[0].shape[0])))
            self._fine_tune_step(idx, inner_iter, lr, lr_scheduler, drop_rate)
            if self.augmenter is not None and not self.device_aug:
                L.load_task(images, labels)   # predictions read the ORIGINAL examples
            preds = self._test_predictions(train_idx, val_idx)
            miou = np.nanmean([_iou(preds[j], lab[val_idx[j]]) for j in range(len(val_idx))])
            if not stopper.continue_training(miou, inner_iter + 1):
                break
        best_num_steps, best_iou = stopper.best_num_steps(), stopper.best_metric()
        print("Best iteration found: {}, with mean-IoU {}".format(best_num_steps, best_iou))
        L.import_all(old)
        return best_num_steps, best_iou

    def evaluate_with_early_stopping(self, dataset, num_classes=1, num_shots=5, inner_batch_size=8, min_steps=0, max_steps=80,
                                     replacement=False, eval_all_tasks=False, num_tasks_to_sample=20,
                                     test_shots=DEFAULT_NUM_TEST_EXAMPLES, lr: Optional[float] = None, drop_rate: Optional[float] = None,
                                     aug_rate: Optional[float] = None, eval_tasks_with_median_early_stopping_iterations: bool = False,
                                     **_unused_tf_handles):
        """reptile.py:296-391: per sampled task, split off `test_shots` validation examples and early-stop on them; optionally
        re-evaluate every task with the median step count.  Returns (task names, best step counts, IoUs)."""
        import numpy as np
        print("Evaluating {} meta-learning.".format(self.meta_fn))
        if eval_all_tasks:
            sampled = dataset
        else:
            random.shuffle(dataset)
            sampled = dataset[:num_tasks_to_sample]
        print("Evaluating {} {}-shot tasks.".format(len(sampled), num_shots))
        task_names, ious = [], []
        if min_steps != max_steps:
            num_steps = []
            for task in sampled:
                (images, labels), name = metaseg.sample_task([task], num_shots + test_shots, None, return_task_name=True)
                task_names.append(name)
                self.learner.load_task(images, labels)
                train_idx, test_idx = metaseg.split_indices(int(images.shape[0]), test_shots)
                steps, miou = self._early_stopping_learn(train_idx, test_idx, labels, inner_batch_size, min_steps=min_steps,
                                                         max_steps=max_steps, replacement=replacement, lr_scheduler=self.lr_scheduler,
                                                         lr=lr, drop_rate=drop_rate, aug_rate=aug_rate, images=images)
                ious.append(miou)
                num_steps.append(steps)
            estimated = int(np.median(num_steps))
        else:
            estimated = min_steps
            num_steps = [estimated] * len(sampled)
        if eval_tasks_with_median_early_stopping_iterations or min_steps == max_steps:
            print("Estimated best number of steps {}".format(estimated))
            mean_iou, task_iou_map = self.evaluate(sampled, num_classes=num_classes, num_shots=num_shots, inner_batch_size=inner_batch_size,
                                                   inner_iters=estimated, replacement=replacement, eval_all_tasks=eval_all_tasks,
                                                   num_tasks_to_sample=num_tasks_to_sample, test_shots=test_shots, lr=lr,
                                                   drop_rate=drop_rate, aug_rate=aug_rate)
            task_names, ious = list(task_iou_map.keys()), list(task_iou_map.values())
        else:
            mean_iou = np.nanmean(ious)
        print("Evaluated {} task/s".format(len(sampled)))
        print("Mean IoU from train on {} images and evaluate on {} test images: {}".format(num_shots, test_shots, mean_iou))
        return task_names, num_steps, ious

    def evaluate_k_shot_range(self, task, k_range, iter_range=None, test_samples=20, early_stopping_min_val_samples=5,
                              esimate_inner_iters_with_early_stoppping: bool = True, **params):
        """k-shot learning results of one task over a range of k (reptile.py:411-440): the first max(k)+test_samples examples are
        shuffled and split once; for each k the first k training examples are used -- with at least 10 of them, 20 % are held out to
        early-stop the step count (which then STAYS in `params` for the following, smaller-than-10 ks too: reference behaviour)."""
        iter_range = DEFAULT_ITER_RANGE if iter_range is None else iter_range
        (images, labels), name = metaseg.sample_task([task], max(k_range) + test_samples, None, return_task_name=True)
        self.learner.load_task(images, labels)
        training, test_idx = metaseg.split_indices(int(images.shape[0]), test_samples)
        mious = []
        for i, k in enumerate(k_range):
            print("Evaluating {}-shot learning".format(k))
            train_idx = training[:k]
            if esimate_inner_iters_with_early_stoppping:
                if k >= early_stopping_min_val_samples * 2:
                    val_shots = int(0.2 * k)
                    print("Split training dataset into {} train shots and {} val shots for early stopping to estimate number of "
                          "steps.".format(k - val_shots, val_shots))
                    order = list(train_idx)
                    random.shuffle(order)
                    d_tr, d_val = order[:-val_shots], order[-val_shots:]
                    steps, _ = self._early_stopping_learn(d_tr, d_val, labels, min_steps=1, max_steps=self.K_SHOT_ES_MAX_STEPS,
                                                          images=images, **params)
                    params["inner_iters"] = steps
            else:
                params["inner_iters"] = iter_range[i]
            mious.append(self._evaluate(train_idx, test_idx, labels, images=images, **params))
        print("Evaluated task {} over k-range {}".format(name, k_range))
        return mious

    def evaluate_m_k_shot_ranges_all_tasks(self, tasks, k_range, m, inner_batch_size, inner_iters, replacement, lr=None, test_samples=20,
                                           iter_range=None, aug_rate: float = 0.5, **_unused_tf_handles):
        """reptile.py:393-409: m repetitions of evaluate_k_shot_range per task -> (flat list of ks, flat list of IoUs)."""
        iter_range = DEFAULT_ITER_RANGE if iter_range is None else iter_range
        assert len(iter_range) == len(k_range)
        params = {"inner_batch_size": inner_batch_size, "inner_iters": inner_iters, "replacement": replacement, "lr": lr, "aug_rate": aug_rate}
        ks, results = [], []
        for task in tasks:
            for _ in range(m):
                res = self.evaluate_k_shot_range(task, k_range=k_range, iter_range=iter_range, test_samples=test_samples, **params)
                print("k-shot results {}".format({k: r for k, r in zip(k_range, res)}))
                results.extend(res)
                ks.extend(k_range)
        return ks, results

    def _test_predictions(self, train_idx, test_idx, L=None):
        """reptile.py:482-524: transductive -> all test images in one inference-mode batch; otherwise one call per test image on the
        batch [train images..., that test image], keeping the last prediction."""
        L = L or self.learner
        if self._transductive:
            return L.predict_resident(list(test_idx), training=False).cpu().numpy()
        out = []
        for t in test_idx:
            out.append(L.predict_resident(list(train_idx) + [t], training=False)[-1].cpu().numpy())
        return out


class FOMLIS(Gecko):
    """First-order MAML for image segmentation (reference class of the same name, reptile.py:569-663)."""

    meta_fn = "FOMAML"
    _train_aug_rate_from_self = True   # reptile.py:654,661 pass aug_rate=self.aug_rate

    def __init__(self, *args, train_shots: Optional[int] = None, tail_shots: Optional[int] = None,
                 sample_train_val_with_replacement: bool = False, **kwargs):
        super().__init__(*args, **kwargs)
        self.train_shots = train_shots - tail_shots if tail_shots is not None else train_shots
        self.tail_shots = tail_shots
        self.sample_train_val_with_replacement = bool(sample_train_val_with_replacement)
        if self.sample_train_val_with_replacement:
            if tail_shots is None or train_shots is None:
                raise ValueError("sample_train_val_with_replacement needs train_shots and tail_shots (reptile.py:657-658)")
            print("Sampling train val with replacement.")
        print("Specializing meta-learner to FOMAML.")

    def _npr(self, rng):
        """numpy generator of the with-replacement draws: the global `np.random` in reference mode (metaseg.py:313-318), a private
        stream derived from the task's generator in per-task mode."""
        import numpy as np
        return None if rng is None else np.random.RandomState(rng.getrandbits(32))

    def _task_batches(self, n_shots, inner_batch_size, inner_iters, replacement, rng):
        if self.sample_train_val_with_replacement:
            return metaseg.fomaml_batch_indices(n_shots, self.tail_shots, inner_batch_size, inner_iters, replacement, rng,
                                                with_replacement_train_shots=self.train_shots, npr=self._npr(rng))
        return metaseg.fomaml_batch_indices(n_shots, self.tail_shots, inner_batch_size, inner_iters, replacement, rng)

    def _steps_per_batch(self, lr) -> int:
        return 1

    def _step(self, idx, j, lr, L=None):
        L = L or self.learner
        wd = self._pre_step_rate
        if lr is not None:  # reptile.py:639-641
            L.inner_step(idx, lr=lr, weight_decay_rate=wd)
        else:  # :642-643 (the scheduler is never consulted: quirk E9)
            L.inner_step(idx, weight_decay_rate=wd)

    def train_step(self, dataset, num_classes=1, num_shots=5, inner_batch_size=8, inner_iters=8, replacement=False, meta_step_size=0.1,
                   meta_batch_size=1, verbose=False, lr=None, **_unused_tf_handles):
        self._run_meta_batch(dataset, num_shots, inner_batch_size, inner_iters, replacement, meta_step_size, meta_batch_size, lr, True)
