"""Entry point with the reference's command line (run_metasegnet.py:28-210, flags of meta_learners/args.py) on the MI355X
inner-loop engine.

    python run_metasegnet.py --image_size 224 --rsd 2 4 --sgd --foml --foml-tail 5 --train-shots 10 --meta-batch 8 \
        --meta-iters 100 --checkpoint ckpt --synthetic-tasks 64
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 run_metasegnet.py ...   (tasks sharded 1/GPU)

Writes the reference's checkpoint directory layout (mliis_amd/checkpoint.py) and `<checkpoint>/meta-test_results.json`.
--optimize_update_hyperparms_on_val_set: update-hyperparameter search on the validation tasks (mliis_amd/hyperparam_search.py: the
early-stopping harness of the reference + a built-in GP / expected-improvement optimiser in place of scikit-optimize);
--run_k_shot_learning_curves_experiment: k-shot learning curves on the FP-k tasks (or on synthetic tasks with --k-shot-range).
--augment / --aug_rate: the reference's augmentation of the inner-loop batches -- same draws of the operations and their parameters
(mliis_amd/augment.py); the pixel work runs on the device (csrc/augment.hip), or in numpy / scipy on the host with --augment-on-host
(draw-identical noise fields too).  Checkpoints: numpy .npz or TensorFlow TensorBundle files (--checkpoint-format tf; restoring takes either).  Data: --data-dir with FSS-1000 TFRecord-GZIP shards
(mliis_amd/tfrecord.py, no TensorFlow needed) or --synthetic-tasks N.
"""
import datetime
import json
import os
import random
import sys

# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); concurrent task lanes (--concurrent-tasks)
# want one queue each next to torch's own streams.  Read when the HIP runtime loads, so set before `import torch`.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from mliis_amd import checkpoint as ckpt  # noqa: E402
from mliis_amd.args import argument_parser, augment_mode, evaluate_kwargs, hyper_search_kwargs, make_lr_scheduler, model_kwargs, train_kwargs  # noqa: E402


def _dataset(args, device, rank):
    """(train, val, test) task lists (run_metasegnet.py:79-98); val is None when empty."""
    from mliis_amd.metaseg import DeviceTask, synthetic_task
    if not args.synthetic_tasks:
        if not args.data_dir:
            raise SystemExit("pass --data-dir <FSS-1000 TFRecord-GZIP shards> or --synthetic-tasks N")
        from mliis_amd import tfrecord
        if args.run_k_shot_learning_curves_experiment:
            test, _ = tfrecord.read_fp_k_shot_dataset(args.data_dir, image_size=args.image_size)
            return None, None, test
        ids = tfrecord.fp_k_test_task_ids() if args.fp_k_test_set else "fss"
        train, val, test, _, _, _ = tfrecord.read_fss_1000_dataset(args.data_dir, num_val_tasks=args.num_val_tasks, test_task_ids=ids,
                                                                   image_size=args.image_size)
        return train, (val or None), test
    n_ex = max(args.train_shots or 0, args.shots + 5)
    if args.run_k_shot_learning_curves_experiment:
        n_ex = max(n_ex, max(args.k_shot_range or [400]) + args.k_shot_test_samples)
    tasks = []
    for i in range(args.synthetic_tasks):
        x, y = synthetic_task(n_ex, args.image_size, seed=i)
        tasks.append(DeviceTask("synthetic_{:04d}".format(i), torch.from_numpy(x).to(device), torch.from_numpy(y).to(device)))
    n_test = max(1, len(tasks) // 4)
    n_val = min(args.num_val_tasks, max(0, len(tasks) - n_test - 1))
    train, val, test = tasks[:len(tasks) - n_test - n_val], tasks[len(tasks) - n_test - n_val:len(tasks) - n_test], tasks[-n_test:]
    return train, (val or None), test


def _validate_datasets(args, train_set, val_set, test_set):
    """utils/util.py:124-130."""
    if not args.pretrained and not args.run_k_shot_learning_curves_experiment:
        assert train_set is not None and len(train_set) > 0, "Training set must have examples."
    assert len(test_set) > 0, "Test set must have examples."
    if args.eval_val_tasks and val_set is not None and len(val_set) == 0:
        raise ValueError("Val set has no tasks to evaluate")


def _evaluate(args, learner, lr_scheduler, train_set, val_set, test_set, aug_pool=None, lanes=()):
    """The evaluation half of the reference's main (run_metasegnet.py:135-206), on rank 0."""
    import copy
    from mliis_amd.eval import evaluate_gecko, optimize_update_hyperparams, run_k_shot_learning_curves_experiment
    from mliis_amd.reptile import SingleRank
    from mliis_amd.train import train_gecko
    ek = evaluate_kwargs(args)
    ek["aug_pool"] = aug_pool
    os.makedirs(args.checkpoint, exist_ok=True)
    if args.optimize_update_hyperparms_on_val_set:
        print("Optimizing the update routine hyperparams on the val set")
        assert val_set is not None and len(val_set) > 0, "Dev set has no tasks"
        keep = ek["save_fine_tuned_checkpoints"]
        ek["save_fine_tuned_checkpoints"] = False
        splits = 1 if args.fss_1000 else 4
        estimated_lr, estimated_steps = optimize_update_hyperparams(
            learner, val_set, lr_scheduler=lr_scheduler, serially_eval_all_tasks=args.serially_eval_all_test_tasks,
            num_configs_to_sample=args.num_configs_to_sample, save_dir=args.checkpoint, results_csv_name=args.uho_results_csv_name,
            num_train_val_data_splits_to_sample_per_config=splits, max_steps=args.max_steps, min_steps=args.min_steps, **ek,
            **hyper_search_kwargs(args))
        ek["save_fine_tuned_checkpoints"] = keep
        ek["eval_inner_iters"] = estimated_steps
        ek["lr"] = estimated_lr
        if args.meta_fine_tune_steps_on_train_val > 0:   # NB (as in the reference): train_kwargs' meta_iters is NOT replaced by that count
            print("Fine-tuning meta-learned init for {} meta-steps with optimized hyperparameters.".format(args.meta_fine_tune_steps_on_train_val))
            tp = train_kwargs(args)
            tp["inner_iters"] = estimated_steps
            tp["lr"] = estimated_lr
            tp["meta_step_size"] = tp["meta_step_size_final"]
            train_gecko(learner, list(train_set) + list(val_set), test_set,
                        os.path.join(args.checkpoint, "fine-tuned_on_train_val_with_optimized_update_hyperparams"), lr_scheduler=lr_scheduler,
                        augment=augment_mode(args), dist=SingleRank(), seed=args.seed, checkpoint_format=args.checkpoint_format, aug_pool=aug_pool,
                        **tp)   # this rank alone (the process group is gone by now): all tasks, no collective
    del ek["eval_tasks_with_median_early_stopping_iterations"]
    if args.run_k_shot_learning_curves_experiment:
        kk = copy.copy(ek)
        del kk["save_fine_tuned_checkpoints"]
        del kk["save_fine_tuned_checkpoints_dir"]
        run_k_shot_learning_curves_experiment(learner, test_set, lr_scheduler=lr_scheduler, iter_range=args.k_shot_iter_range,
                                              k_range=args.k_shot_range, test_samples=args.k_shot_test_samples,
                                              csv_outpath=os.path.join(args.checkpoint, "k-shot-results.csv"), **kk)
        return
    mean_train_iou = None
    if train_set and not args.skip_train_task_eval:
        print("Evaluating {}-shot learning on training tasks.".format(args.shots))
        keep = ek["save_fine_tuned_checkpoints"]
        ek["save_fine_tuned_checkpoints"] = args.save_fine_tuned_checkpoints_train
        mean_train_iou, _ = evaluate_gecko(learner, train_set, lr_scheduler=lr_scheduler, serially_eval_all_tasks=False, lanes=lanes, **ek)
        ek["save_fine_tuned_checkpoints"] = keep
    name = "test"
    if args.eval_val_tasks:
        test_set, name = val_set, "val"
    print("Evaluating {}-shot learning on meta-{} tasks.".format(args.shots, name))
    mean_test_iou, task_name_iou_map = evaluate_gecko(learner, test_set, lr_scheduler=lr_scheduler, lanes=lanes,
                                                      serially_eval_all_tasks=args.serially_eval_all_test_tasks, **ek)
    print("Evaluated meta-{} tasks:".format(name))
    print(task_name_iou_map)
    if mean_train_iou is not None:
        print("Mean meta-train IoU: {}".format(mean_train_iou))
    # Do NOT change this print (it's used to grep logs):
    print("Mean IoU over all meta-test tasks: {}".format(mean_test_iou))
    out = os.path.join(args.checkpoint, "meta-test_results.json")
    with open(out, "w") as f:
        json.dump(task_name_iou_map, f)
    print("Wrote results to {}".format(out))


def main(argv=None, learner_factory=None, device=None):
    """argv: command line (default sys.argv[1:]).  learner_factory / device: test seam -- any object with the Learner protocol
    (tests drive the whole program on the CPU oracle learner); the product path always builds mliis_amd.learner.Learner on the GPU."""
    start = datetime.datetime.now()
    print("Experiment started at: {}".format(start))
    args = argument_parser().parse_args(argv)
    random.seed(args.seed)
    aug_pool = None
    if args.augment and args.augment_on_host and args.augment_workers != 0:   # forked workers: created before anything initialises the GPU
        from mliis_amd.augment import AugmentPool
        aug_pool = AugmentPool(None if args.augment_workers < 0 else args.augment_workers)
        print("Augmentation pixel work on {} worker processes.".format(aug_pool.workers))
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            torch.cuda.set_device(local)   # (device_count() does not initialise the GPU; this and everything after it does)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    device = torch.device("cuda", local) if device is None else torch.device(device)
    if learner_factory is None:
        from mliis_amd.learner import Learner
    else:
        Learner = learner_factory
    from mliis_amd.reptile import Dist
    from mliis_amd.train import train_gecko

    print("Defining model architecture:")
    print("Using loss {}".format(args.loss_name))
    # (multi-rank: the rank goes into the key of the device mask generator only -- same weights everywhere, different drop-connect draws)
    rank_kw = dict(rng_stream=rank) if (world > 1 and learner_factory is None) else {}
    learner = Learner(device=device, **model_kwargs(args), **rank_kw)
    lr_scheduler = make_lr_scheduler(args)
    print("Model contains {} trainable parameters.".format(learner.n_trainable))
    print("Meta-learning with algorithm:\n{}".format("FOMAML" if args.foml else "Reptile"))
    train_set, val_set, test_set = _dataset(args, device, rank)
    _validate_datasets(args, train_set, val_set, test_set)

    lanes = []
    if args.concurrent_tasks > 1 and not args.augment:   # the augmented path uploads host batches step by step: one lane
        lanes = [Learner(device=device, **dict(model_kwargs(args), seed=args.seed + 1000 * k), **rank_kw) for k in range(1, args.concurrent_tasks)]

    if args.restore_efficient_net_weights_from is not None and not args.pretrained:
        path = ckpt.latest_checkpoint(args.restore_efficient_net_weights_from)
        print("Restoring from checkpoint {}".format(path))
        learner.load_named(ckpt.load(path), strict=False, prefixes=[learner.feature_extractor_name])
    if not args.pretrained:
        print("Meta-training...")
        if args.continue_training_from_checkpoint is not None:
            path = ckpt.latest_checkpoint(args.continue_training_from_checkpoint)
            print("Continuing meta-training from checkpoint: {}".format(path))
            learner.load_named(ckpt.load(path))
        train_gecko(learner, train_set, val_set or test_set, args.checkpoint, lr_scheduler=lr_scheduler, augment=augment_mode(args), dist=Dist(),
                    seed=args.seed, checkpoint_format=args.checkpoint_format, aug_pool=aug_pool, lanes=lanes, **train_kwargs(args))
    else:
        path = ckpt.latest_checkpoint(args.checkpoint)
        print("Restoring from checkpoint: {}".format(path))
        if args.do_not_restore_final_layer_weights:
            learner.load_named(ckpt.load(path), strict=False, exclude_prefix=learner.final_layer_scope)
        else:
            learner.load_named(ckpt.load(path))

    if world > 1:
        # Meta-training is the only multi-rank phase.  The process group ends HERE, on every rank, before rank 0 evaluates alone:
        # the other ranks then simply exit instead of sitting in an RCCL barrier whose watchdog a long evaluation would trip.
        import torch.distributed as dist
        learner.synchronize()
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        _evaluate(args, learner, lr_scheduler, train_set, val_set, test_set, aug_pool, lanes)
    for ln in lanes:
        ln.close()
    if aug_pool is not None:
        aug_pool.close()
    end = datetime.datetime.now()
    print("Experiment finished at: {}, taking {}".format(end, end - start))


if __name__ == "__main__":
    main()
