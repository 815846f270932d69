"""Entry point with the reference's command line (run_metasegnet.py:28-210, flags of meta_learners/args.py) on the MI355X
inner-loop engine.

    python run_metasegnet.py --image_size 224 --rsd 2 4 --sgd --foml --foml-tail 5 --train-shots 10 --meta-batch 8 \
        --meta-iters 100 --checkpoint ckpt --synthetic-tasks 64
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 run_metasegnet.py ...   (tasks sharded 1/GPU)

Writes the reference's checkpoint directory layout (mliis_amd/checkpoint.py) and `<checkpoint>/meta-test_results.json`.
Out of scope in this build (clear errors): UHO hyper-parameter search, k-shot learning-curve experiment.
--augment / --aug_rate: the reference's host numpy augmentation of the inner-loop batches (mliis_amd/augment.py, same draws).  Checkpoints: numpy .npz or TensorFlow TensorBundle files (--checkpoint-format tf; restoring takes either).  Data: --data-dir with FSS-1000 TFRecord-GZIP shards
(mliis_amd/tfrecord.py, no TensorFlow needed) or --synthetic-tasks N.
"""
import datetime
import json
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from mliis_amd import checkpoint as ckpt  # noqa: E402
from mliis_amd.args import argument_parser, evaluate_kwargs, make_lr_scheduler, model_kwargs, train_kwargs  # noqa: E402


def _dataset(args, device, rank):
    from mliis_amd.metaseg import DeviceTask, synthetic_task
    if not args.synthetic_tasks:
        if not args.data_dir:
            raise SystemExit("pass --data-dir <FSS-1000 TFRecord-GZIP shards> or --synthetic-tasks N")
        from mliis_amd import tfrecord
        ids = tfrecord.fp_k_test_task_ids() if args.fp_k_test_set else "fss"
        train, val, test, _, _, _ = tfrecord.read_fss_1000_dataset(args.data_dir, num_val_tasks=args.num_val_tasks, test_task_ids=ids,
                                                                   image_size=args.image_size)
        if not train or not test:
            raise ValueError("Train / test set has no tasks to evaluate")          # utils/util.py:124-130
        return train, (val or test)
    n_ex = max(args.train_shots or 0, args.shots + 5)
    tasks = []
    for i in range(args.synthetic_tasks):
        x, y = synthetic_task(n_ex, args.image_size, seed=i)
        tasks.append(DeviceTask("synthetic_{:04d}".format(i), torch.from_numpy(x).to(device), torch.from_numpy(y).to(device)))
    n_test = max(1, len(tasks) // 4)
    return tasks[:-n_test], tasks[-n_test:]


def main():
    start = datetime.datetime.now()
    print("Experiment started at: {}".format(start))
    args = argument_parser().parse_args()
    if args.optimize_update_hyperparms_on_val_set or args.run_k_shot_learning_curves_experiment:
        raise NotImplementedError("UHO search / k-shot learning curves are experiment harnesses outside the hot path (SURVEY.md 8(f)-4)")
    random.seed(args.seed)
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    device = torch.device("cuda", local)
    from mliis_amd.learner import Learner
    from mliis_amd.reptile import Dist
    from mliis_amd.train import train_gecko

    print("Defining model architecture:")
    print("Using loss {}".format(args.loss_name))
    learner = Learner(device=device, **model_kwargs(args))
    lr_scheduler = make_lr_scheduler(args)
    print("Model contains {} trainable parameters.".format(learner.n_trainable))
    print("Meta-learning with algorithm:\n{}".format("FOMAML" if args.foml else "Reptile"))
    train_set, test_set = _dataset(args, device, rank)

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
        train_gecko(learner, train_set, test_set, args.checkpoint, lr_scheduler=lr_scheduler, augment=args.augment, dist=Dist(),
                    seed=args.seed, checkpoint_format=args.checkpoint_format, **train_kwargs(args))
    else:
        path = ckpt.latest_checkpoint(args.checkpoint)
        print("Restoring from checkpoint: {}".format(path))
        if args.do_not_restore_final_layer_weights:
            learner.load_named(ckpt.load(path), strict=False, exclude_prefix=learner.final_layer_scope)
        else:
            learner.load_named(ckpt.load(path))

    if rank == 0:
        from mliis_amd.eval import evaluate_gecko
        ek = evaluate_kwargs(args)
        print("Evaluating {}-shot learning on meta-test tasks.".format(args.shots))
        mean_test_iou, task_name_iou_map = evaluate_gecko(learner, test_set, lr_scheduler=lr_scheduler,
                                                          serially_eval_all_tasks=args.serially_eval_all_test_tasks, **ek)
        print("Evaluated meta-test tasks:")
        print(task_name_iou_map)
        # Do NOT change this print (it's used to grep logs):
        print("Mean IoU over all meta-test tasks: {}".format(mean_test_iou))
        os.makedirs(args.checkpoint, exist_ok=True)
        out = os.path.join(args.checkpoint, "meta-test_results.json")
        with open(out, "w") as f:
            json.dump(task_name_iou_map, f)
        print("Wrote results to {}".format(out))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    end = datetime.datetime.now()
    print("Experiment finished at: {}, taking {}".format(end, end - start))


if __name__ == "__main__":
    main()
