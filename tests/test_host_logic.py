"""Product host logic vs golden vectors generated from the reference's own numpy/Python code
(tests/golden/make_host_logic_golden.py)."""
import random

import numpy as np
import pytest

from mliis_amd import lr_schedulers as LR
from mliis_amd import metaseg as MS
from mliis_amd import metrics as MT


def test_cosine_and_step(golden):
    a = golden["cosine"]
    s = LR.CosineLRScheduler(*a["args"])
    assert [s.cur_lr(t) for t in range(len(a["lr"]))] == pytest.approx(a["lr"], rel=1e-15, abs=1e-18)
    for key in ("step", "step_b"):
        a = golden[key]
        s = LR.StepDecay(*a["args"])
        assert [s.cur_lr(t) for t in range(len(a["lr"]))] == pytest.approx(a["lr"], rel=1e-15)
    assert sorted(LR.supported_learning_rate_schedulers) == golden["schedulers"]
    assert LR.supported_learning_rate_schedulers["fixed"] is None


def test_mini_batches_bit_exact(golden):
    for c in golden["mini_batches"]:
        random.seed(c["seed"])
        got = [list(b) for b in MS.mini_batch_indices(c["n"], c["batch"], c["num_batches"], c["replacement"])]
        assert got == c["batches"], c
        random.seed(c["seed"])
        got2 = [[s[0] for s in b] for b in MS._mini_batches([(i, i) for i in range(c["n"])], c["batch"], c["num_batches"], c["replacement"])]
        assert got2 == c["batches"]


def test_mini_batches_private_rng_matches_global(golden):
    c = golden["mini_batches"][0]
    got = [list(b) for b in MS.mini_batch_indices(c["n"], c["batch"], c["num_batches"], False, rng=random.Random(c["seed"]))]
    assert got == c["batches"]


def test_empty_and_ragged():
    with pytest.raises(ValueError):
        list(MS.mini_batch_indices(0, 8, 1))
    with pytest.raises(ValueError):
        list(MS._mini_batches([], 8, 1))
    # 1 sample, batch 4: pure wrap-around duplicates
    assert [list(b) for b in MS.mini_batch_indices(1, 4, 2)] == [[0] * 4, [0] * 4]


def test_split(golden):
    for c in golden["split"]:
        random.seed(c["seed"])
        tr, te = MS.split_indices(c["n"], c["test_shots"])
        assert tr == c["train"] and te == c["test"]
        random.seed(c["seed"])
        tr2, te2 = MS._split_train_test_segmentation([(i, i) for i in range(c["n"])], c["test_shots"])
        assert [s[0] for s in tr2] == c["train"] and [s[0] for s in te2] == c["test"]


def test_fomaml_schedule(golden):
    for c in golden["foml_batches"]:
        random.seed(c["seed"])
        got = MS.fomaml_batch_indices(c["n"], c["tail"], c["batch"], c["inner_iters"])
        assert got == c["batches"]
        assert len(got) == c["inner_iters"] and len(got[-1]) == c["tail"]


def test_fomaml_schedule_with_replacement(golden):
    """--sample_foml_train_val_with_replacement: head / tail are numpy draws with replacement (metaseg.py:313-318), bit-exact against
    the reference's own function for seeded `random` + `np.random`."""
    for c in golden["foml_with_replacement"]:
        random.seed(c["seed"])
        np.random.seed(c["np_seed"])
        got = MS.fomaml_batch_indices(c["n"], c["tail"], c["batch"], c["inner_iters"], with_replacement_train_shots=c["train"])
        assert got == c["batches"], c
        np.random.seed(c["np_seed"])
        tr, te = MS._sample_train_test_segmentation_with_replacement(list(range(c["n"])), c["train"], c["tail"])
        assert tr == c["head"] and te == c["batches"][-1]
        # a private RandomState with the same seed gives the same draws (per-task mode)
        random.seed(c["seed"])
        got2 = MS.fomaml_batch_indices(c["n"], c["tail"], c["batch"], c["inner_iters"], with_replacement_train_shots=c["train"],
                                       npr=np.random.RandomState(c["np_seed"]))
        assert got2 == c["batches"]


def test_iou_measure_ci95_earlystopper(golden):
    for c in golden["iou"]:
        assert MT.iou(np.array(c["pred"], np.float32), np.array(c["label"], np.float32)) == pytest.approx(c["iou"], rel=1e-12)
    with pytest.raises(ValueError):
        MT.iou(np.zeros((1, 2, 2, 2)), np.zeros((1, 2, 2, 2)))
    with pytest.raises(ValueError):
        MT.iou(np.zeros((2, 2, 2)), np.zeros((2, 3, 2)))
    for c in golden["measure"]:
        tp, tn, fp, fn = MT.measure(np.array(c["y"], np.float32), np.array(c["pred"], np.float32))
        assert (tp, tn, fp, fn) == (c["tp"], c["tn"], c["fp"], c["fn"])
        assert MT.iou_img(tp, fp, fn) == pytest.approx(c["iou_img"])
    for c in golden["ci95"]:
        assert MT.ci95(c["a"]) == pytest.approx(c["v"], rel=1e-12, abs=1e-15)
    for c in golden["early_stopper"]:
        es = MT.EarlyStopper(patience=c["patience"], min_steps=c["min_steps"])
        assert [es.continue_training(m, i + 1) for i, m in enumerate(c["metrics"])] == c["continue"]
        assert es.best_num_steps() == c["best_num_steps"] and es.best_metric() == c["best_metric"]


def test_synthetic_task_shape_and_determinism():
    a, la = MS.synthetic_task(3, 32, seed=1)
    b, lb = MS.synthetic_task(3, 32, seed=1)
    assert a.shape == (3, 32, 32, 3) and la.shape == (3, 32, 32, 2)
    assert (a == b).all() and (la == lb).all()
    assert set(np.unique(la)) <= {0.0, 1.0} and (la.sum(-1) == 1).all()
    assert a.min() >= 0 and a.max() <= 255
