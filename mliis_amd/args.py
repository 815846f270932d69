"""Command-line surface of run_metasegnet.py: the same flag names, types and defaults as the reference's
meta_learners/args.py:16-118 (pinned by tests/golden/host_logic.json -> argparse_defaults / argparse_runsh_like), and the kwargs
builders of args.py:121-236 re-stated for the device learner.  Flags whose feature is outside the hot-path scope are accepted
(so existing command lines parse) and rejected with a clear error only if they would change behaviour.
"""
from __future__ import annotations

import argparse
from functools import partial

from .lr_schedulers import supported_learning_rate_schedulers

SUPPORTED_MODELS = {"efficientlab"}
SUPPORTED_SEARCH_ALGS = {"GP"}

# (flag, kwargs) in the reference's order
_FLAGS = [
    ("--fine-tune-task", dict(type=str, default=None)),
    ("--fine-tuned-checkpoint", dict(type=str, default=None)),
    ("--pretrained", dict(action="store_true", default=False)),
    ("--seed", dict(type=int, default=0)),
    ("--checkpoint", dict(default="model_checkpoint")),
    ("--classes", dict(type=int, default=1)),
    ("--shots", dict(type=int, default=5)),
    ("--train-shots", dict(type=int, default=5)),
    ("--inner-batch", dict(type=int, default=8)),
    ("--inner-iters", dict(type=int, default=8)),
    ("--replacement", dict(action="store_true")),
    ("--learning-rate", dict(type=float, default=1e-3)),
    ("--meta-step", dict(type=float, default=0.1)),
    ("--meta-step-final", dict(type=float, default=0.1)),
    ("--meta-batch", dict(type=int, default=5)),
    ("--meta-iters", dict(type=int, default=400000)),
    ("--eval-batch", dict(type=int, default=8)),
    ("--eval-iters", dict(type=int, default=4)),
    ("--eval-samples", dict(type=int, default=200)),
    ("--eval-interval", dict(type=int, default=10)),
    ("--weight-decay", dict(type=float, default=1)),
    ("--transductive", dict(action="store_true")),
    ("--foml", dict(action="store_true")),
    ("--foml-tail", dict(type=int, default=None)),
    ("--sgd", dict(action="store_true")),
    ("--n_unet_encoding_stacks", dict(type=int, default=4)),
    ("--data-dir", dict()),
    ("--loss_name", dict(default="cross_entropy")),
    ("--save_fine_tuned_checkpoints", dict(action="store_true")),
    ("--save_fine_tuned_checkpoints_train", dict(action="store_true")),
    ("--save_fine_tuned_checkpoints_dir", dict(default="/tmp/checkpoints/fine-tuned")),
    ("--model_name", dict(default="efficientlab")),
    ("--start_num_feature_maps_power", dict(type=int, default=5)),
    ("--restore_efficient_net_weights_from", dict(type=str, default=None)),
    ("--spatial_pyramid_pooling", dict(action="store_true")),
    ("--skip_decoding", dict(action="store_true")),
    ("--rsd", dict(type=int, nargs="+")),
    ("--feature_extractor_name", dict(type=str, default="efficientnet-b0")),
    ("--learning_rate_scheduler", dict(type=str, default="fixed")),
    ("--step_decay_rate", dict(type=float, default=0.5)),
    ("--decay_after_n_steps", dict(type=int, default=5)),
    ("--l2", dict(action="store_true")),
    ("--l1", dict(action="store_true")),
    ("--darc1", dict(action="store_true")),
    ("--augment", dict(action="store_true")),
    ("--final_layer_dropout_rate", dict(type=float, default=0.0)),
    ("--image_size", dict(type=int, default=320)),
    ("--label_smoothing", dict(type=float, default=0.0)),
    ("--continue_training_from_checkpoint", dict(default=None)),
    ("--fss_1000", dict(action="store_true")),
    ("--num_val_tasks", dict(type=int, default=0)),
    ("--eval_val_tasks", dict(action="store_true")),
    ("--serially_eval_all_test_tasks", dict(action="store_true")),
    ("--optimize_update_hyperparms_on_val_set", dict(action="store_true")),
    ("--num_configs_to_sample", dict(type=int, default=100)),
    ("--meta_fine_tune_steps_on_train_val", dict(type=int, default=0)),
    ("--uho_outer_iters", dict(type=int, default=2)),
    ("--lr_search_range_low", dict(type=float, default=0.0005)),
    ("--lr_search_range_high", dict(type=float, default=0.05)),
    ("--drop_rate_search_range_low", dict(type=float, default=0.2)),
    ("--drop_rate_search_range_high", dict(type=float, default=0.2)),
    ("--aug_rate_search_range_low", dict(type=float, default=0.5)),
    ("--aug_rate_search_range_high", dict(type=float, default=0.5)),
    ("--batch_size_search_range_low", dict(type=int, default=8)),
    ("--batch_size_search_range_high", dict(type=int, default=8)),
    ("--run_k_shot_learning_curves_experiment", dict(action="store_true")),
    ("--fp_k_test_set", dict(action="store_true")),
    ("--disable_rsd_residual_connections", dict(action="store_true")),
    ("--do_not_restore_final_layer_weights", dict(action="store_true")),
    ("--eval_tasks_with_median_early_stopping_iterations", dict(action="store_true")),
    ("--min_steps", dict(type=int, default=0)),
    ("--max_steps", dict(type=int, default=80)),
    ("--k_shot_iter_range", dict(nargs="+", type=int, default=None)),
    ("--sample_foml_train_val_with_replacement", dict(action="store_true")),
    ("--aug_rate", dict(type=float, default=0.5)),
    ("--uho_results_csv_name", dict(type=str, default="val-set_hyper_param_search_results.csv")),
    ("--uho_estimator", dict(type=str, default="GP")),
]
# extensions of this build (not in the reference)
_EXT = [
    ("--synthetic-tasks", dict(type=int, default=0, help="use N synthetic tasks instead of --data-dir (no dataset needed)")),
    ("--no-hip-graph", dict(action="store_true", help="launch inner steps eagerly instead of replaying a captured HIP graph")),
    ("--k-shot-range", dict(nargs="+", type=int, default=None,
                            help="ks of --run_k_shot_learning_curves_experiment (default: the reference's 1 5 10 50 100 200 400)")),
    ("--k-shot-test-samples", dict(type=int, default=20, help="held-out examples per task in the k-shot experiment (reference: 20)")),
    ("--skip-train-task-eval", dict(action="store_true",
                                    help="skip the evaluation pass over the meta-TRAIN tasks that the reference always runs before the test tasks")),
    ("--augment-on-host", dict(action="store_true",
                               help="with --augment: compute the augmented pixels in numpy / scipy on the host, draw- and pixel-identical to "
                                    "the reference -- REQUIRED for reference-parity runs.  Default: same scalar draws, pixel work on the device "
                                    "(csrc/augment.hip): Philox noise fields, Keys-cubic rotation instead of scipy's prefiltered B-spline, "
                                    "constant-mode holes by coordinate range -- statistically, not pixel-, identical")),
    ("--augment-workers", dict(type=int, default=-1,
                               help="worker processes for the pixel half of --augment (-1: host cores - 1, 0: inline like the reference)")),
    ("--concurrent-tasks", dict(type=int, default=1,
                                help="adapt this many tasks of a meta-batch at once on separate learners / streams (same meta-update; "
                                     "4 is the optimum on MI355X, needs meta_batch_size / ranks >= 2 to matter)")),
    ("--matmul-precision", dict(choices=["fp32", "fp32-native", "bf16", "fp8", "bf16-storage"], default="fp32",
                                help="operand precision of the matrix cores in the dense convs (bf16: fp32 tensors rounded on the fly, fp32 accumulation); "
                                     "bf16-storage: bf16 operands and the expanded MBConv tensors (z0, z1, a1, their gradients) as bf16 in HBM during "
                                     "training steps, fp32 statistics / accumulation / weights (BASELINE configs[3])")),
    ("--checkpoint-format", dict(choices=["npz", "tf"], default="npz",
                                 help="tensor container of written checkpoints: numpy .npz or a TensorFlow TensorBundle (.index/.data)")),
]


def argument_parser(extensions: bool = True) -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    for flag, kw in _FLAGS + (_EXT if extensions else []):
        p.add_argument(flag, **kw)
    return p


def _max_shots(a) -> int:
    """Examples a task can make resident at once: training / evaluation shots, or the whole k-shot pool."""
    n = max(16, a.train_shots or 0, a.shots + 5)
    if getattr(a, "run_k_shot_learning_curves_experiment", False):
        n = max(n, max(getattr(a, "k_shot_range", None) or [400]) + getattr(a, "k_shot_test_samples", 20))
    return n


def augment_mode(a):
    """False | True (host pixels, --augment-on-host) | "device" -- the `augment` argument of the meta-learners."""
    if not a.augment:
        return False
    return True if getattr(a, "augment_on_host", False) else "device"


def model_kwargs(a) -> dict:
    """Keyword arguments of mliis_amd.learner.Learner from parsed flags (reference: args.py:121-160)."""
    a.model_name = a.model_name.lower()
    if a.model_name not in SUPPORTED_MODELS:
        raise ValueError("Model name must be in the set: {} but is {}".format(SUPPORTED_MODELS, a.model_name))
    return dict(feature_extractor_name=a.feature_extractor_name, image_size=a.image_size, rsd=a.rsd or None,
                learning_rate=a.learning_rate, optimizer="sgd" if a.sgd else "adam", l2=a.l2, l1=a.l1, darc1=a.darc1,
                dice=("dice" in a.loss_name), label_smoothing=a.label_smoothing, final_layer_dropout_rate=a.final_layer_dropout_rate,
                spatial_pyramid_pooling=a.spatial_pyramid_pooling, skip_decoding=a.skip_decoding, seed=a.seed,
                use_graph=not getattr(a, "no_hip_graph", False), max_shots=_max_shots(a),
                matmul_precision=getattr(a, "matmul_precision", "fp32"),
                augment_batch_capacity=(max(16, a.inner_batch, a.eval_batch, getattr(a, "batch_size_search_range_high", 0) or 0)
                                        if augment_mode(a) == "device" else 0))
    # --disable_rsd_residual_connections is a no-op in the reference too (kwarg name mismatch)


def _meta_fn(a):
    from .reptile import FOMLIS, Gecko
    if a.foml:
        return partial(FOMLIS, train_shots=a.train_shots, tail_shots=a.foml_tail,
                       sample_train_val_with_replacement=a.sample_foml_train_val_with_replacement)
    return Gecko


def train_kwargs(a) -> dict:
    if a.learning_rate_scheduler not in supported_learning_rate_schedulers:
        raise ValueError("Learning rate scheduler, {}, not in supported set: {}".format(a.learning_rate_scheduler,
                                                                                       supported_learning_rate_schedulers.keys()))
    return dict(num_classes=a.classes, num_shots=a.shots, train_shots=(a.train_shots or None), inner_batch_size=a.inner_batch,
                inner_iters=a.inner_iters, replacement=a.replacement, meta_step_size=a.meta_step, meta_step_size_final=a.meta_step_final,
                meta_batch_size=a.meta_batch, meta_iters=a.meta_iters, eval_inner_batch_size=a.eval_batch, eval_inner_iters=a.eval_iters,
                eval_interval=a.eval_interval, weight_decay_rate=a.weight_decay, transductive=a.transductive, meta_fn=_meta_fn(a),
                aug_rate=a.aug_rate)


def evaluate_kwargs(a) -> dict:
    return dict(num_classes=a.classes, num_shots=a.shots, eval_inner_batch_size=a.eval_batch, eval_inner_iters=a.eval_iters,
                replacement=a.replacement, weight_decay_rate=a.weight_decay, num_samples=a.eval_samples, transductive=a.transductive,
                meta_fn=_meta_fn(a), augment=augment_mode(a), lr=None, aug_rate=a.aug_rate,
                eval_tasks_with_median_early_stopping_iterations=a.eval_tasks_with_median_early_stopping_iterations,
                save_fine_tuned_checkpoints=a.save_fine_tuned_checkpoints, save_fine_tuned_checkpoints_dir=a.save_fine_tuned_checkpoints_dir)


def hyper_search_kwargs(a) -> dict:
    """args.py:163-176."""
    from .hyperparam_search import SUPPORTED_SEARCH_ALGS
    assert a.uho_estimator in SUPPORTED_SEARCH_ALGS, "{} not in supported hyperparam search algs {}".format(a.uho_estimator, SUPPORTED_SEARCH_ALGS)
    return dict(lr_search_range_low=a.lr_search_range_low, lr_search_range_high=a.lr_search_range_high,
                drop_rate_search_range_low=a.drop_rate_search_range_low, drop_rate_search_range_high=a.drop_rate_search_range_high,
                aug_rate_search_range_low=a.aug_rate_search_range_low, aug_rate_search_range_high=a.aug_rate_search_range_high,
                batch_size_search_range_low=a.batch_size_search_range_low, batch_size_search_range_high=a.batch_size_search_range_high,
                estimator=a.uho_estimator)


def make_lr_scheduler(a):
    """run_metasegnet.py:55-65: the scheduler horizon is eval_inner_iters."""
    cls = supported_learning_rate_schedulers[a.learning_rate_scheduler]
    if cls is None:
        return None
    kw = {"decay_rate": a.step_decay_rate, "decay_after_n_steps": a.decay_after_n_steps} if "step" in a.learning_rate_scheduler else {}
    return cls(a.learning_rate, a.eval_iters, **kw)
